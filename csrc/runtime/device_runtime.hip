// Device runtime of the member engine: what the engine owns on a GPU besides kernels.
//
//   Arena        one region of HBM (hipMalloc) or of pinned host memory (hipHostMalloc), handed out
//                by bumping an offset. An engine's KV pool, the fixed-layout argument blocks of its
//                decode graphs and its staging buffers are carved from arenas sized at start-up:
//                288 GB of HBM is planned once, nothing on the serving path calls hipMalloc.
//   Stream/Event thin owners of hipStream_t / hipEvent_t (one stream per engine, so the members of
//                an ensemble have steps in flight at the same time).
//   GraphCache   hipGraph executables by integer key: begin/end a THREAD-LOCAL stream capture,
//                instantiate, replay. Thread-local because other host threads keep making HIP
//                calls while one engine records (another engine's step loop, the HTTP front end,
//                a process group's watchdog); under the global mode any such call fails the capture.
//
// C++ classes with a flat C ABI on top (ctypes binds it: llm_consensus_amd/runtime.py). Host code
// only; compiled with hipcc for the HIP runtime API. The reference has no counterpart: its
// "device" is an HTTPS endpoint (/root/reference/src/main.rs:82-86).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#define LC_EXPORT extern "C" __attribute__((visibility("default")))

namespace lcrt {

class Arena {
 public:
  Arena(size_t bytes, bool pinned) : cap_(bytes), pinned_(pinned) {
    hipError_t e = pinned ? hipHostMalloc(&base_, bytes, hipHostMallocDefault) : hipMalloc(&base_, bytes);
    if (e != hipSuccess) {
      base_ = nullptr;
      err_ = static_cast<int>(e);
      (void)hipGetLastError();
    }
  }
  ~Arena() {
    if (!base_) return;
    if (pinned_) (void)hipHostFree(base_);
    else (void)hipFree(base_);
  }
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;

  bool ok() const { return base_ != nullptr; }
  int error() const { return err_; }

  // nullptr when the arena is exhausted (the caller sized it: that is a planning error, not a
  // reason to fall back to the general allocator).
  void* alloc(size_t bytes, size_t align) {
    if (!base_ || align == 0 || (align & (align - 1))) return nullptr;
    std::lock_guard<std::mutex> g(mu_);
    const size_t at = (used_ + align - 1) & ~(align - 1);
    if (at > cap_ || bytes > cap_ - at) return nullptr;
    used_ = at + bytes;
    if (used_ > peak_) peak_ = used_;
    ++count_;
    return static_cast<char*>(base_) + at;
  }
  size_t mark() {
    std::lock_guard<std::mutex> g(mu_);
    return used_;
  }
  // Gives back everything allocated after `mark` (stack discipline).
  bool release(size_t mark) {
    std::lock_guard<std::mutex> g(mu_);
    if (mark > used_) return false;
    used_ = mark;
    return true;
  }
  size_t capacity() const { return cap_; }
  size_t used() const { return used_; }
  size_t peak() const { return peak_; }
  size_t allocations() const { return count_; }
  void* base() const { return base_; }

 private:
  void* base_ = nullptr;
  size_t cap_ = 0, used_ = 0, peak_ = 0, count_ = 0;
  bool pinned_ = false;
  int err_ = 0;
  std::mutex mu_;
};

class Stream {
 public:
  explicit Stream(int priority) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least urgent, hi = most urgent (numerically smaller)
    int p = priority > 0 ? hi : (priority < 0 ? lo : 0);
    if (hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, p) != hipSuccess) {
      s_ = nullptr;
      (void)hipGetLastError();
    }
  }
  ~Stream() {
    if (s_) (void)hipStreamDestroy(s_);
  }
  Stream(const Stream&) = delete;
  Stream& operator=(const Stream&) = delete;
  hipStream_t get() const { return s_; }

 private:
  hipStream_t s_ = nullptr;
};

class Event {
 public:
  explicit Event(bool timing) {
    if (hipEventCreateWithFlags(&e_, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) {
      e_ = nullptr;
      (void)hipGetLastError();
    }
  }
  ~Event() {
    if (e_) (void)hipEventDestroy(e_);
  }
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
  hipEvent_t get() const { return e_; }

 private:
  hipEvent_t e_ = nullptr;
};

class GraphCache {
 public:
  ~GraphCache() { clear(); }

  int begin(hipStream_t s) {
    if (!s) return -1;   // the legacy default stream cannot be captured
    return static_cast<int>(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  }

  // Ends the capture on `s` and stores the executable under `key` (replacing an older one).
  // Returns the number of nodes (> 0), or -(HIP error) on failure.
  long long end(hipStream_t s, long long key) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (e != hipSuccess || !g) {
      (void)hipGetLastError();
      return e != hipSuccess ? -static_cast<long long>(e) : -1;
    }
    size_t nodes = 0;
    (void)hipGraphGetNodes(g, nullptr, &nodes);
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || !x) {
      (void)hipGetLastError();
      return e != hipSuccess ? -static_cast<long long>(e) : -1;
    }
    std::lock_guard<std::mutex> lk(mu_);
    auto it = execs_.find(key);
    if (it != execs_.end()) {
      (void)hipGraphExecDestroy(it->second);
      it->second = x;
    } else {
      execs_.emplace(key, x);
    }
    return static_cast<long long>(nodes ? nodes : 1);
  }

  // Ends a capture that went wrong and throws its graph away.
  void abort(hipStream_t s) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
  }

  int launch(long long key, hipStream_t s) {
    hipGraphExec_t x = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = execs_.find(key);
      if (it == execs_.end()) return -1;
      x = it->second;
    }
    launches_.fetch_add(1, std::memory_order_relaxed);
    return static_cast<int>(hipGraphLaunch(x, s));
  }

  bool erase(long long key) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = execs_.find(key);
    if (it == execs_.end()) return false;
    (void)hipGraphExecDestroy(it->second);
    execs_.erase(it);
    return true;
  }

  void clear() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : execs_) (void)hipGraphExecDestroy(kv.second);
    execs_.clear();
  }

  size_t size() {
    std::lock_guard<std::mutex> lk(mu_);
    return execs_.size();
  }
  unsigned long long launches() const { return launches_.load(std::memory_order_relaxed); }

 private:
  std::mutex mu_;
  std::unordered_map<long long, hipGraphExec_t> execs_;
  std::atomic<unsigned long long> launches_{0};
};

}  // namespace lcrt

using namespace lcrt;

// ---- arenas ----
LC_EXPORT void* lc_rt_arena_create(long long bytes, int pinned) {
  if (bytes <= 0) return nullptr;
  Arena* a = new Arena(static_cast<size_t>(bytes), pinned != 0);
  if (!a->ok()) {
    delete a;
    return nullptr;
  }
  return a;
}
LC_EXPORT void lc_rt_arena_destroy(void* a) { delete static_cast<Arena*>(a); }
LC_EXPORT void* lc_rt_arena_alloc(void* a, long long bytes, long long align) {
  return (a && bytes >= 0) ? static_cast<Arena*>(a)->alloc(static_cast<size_t>(bytes), static_cast<size_t>(align)) : nullptr;
}
LC_EXPORT long long lc_rt_arena_mark(void* a) { return static_cast<long long>(static_cast<Arena*>(a)->mark()); }
LC_EXPORT int lc_rt_arena_release(void* a, long long mark) {
  return static_cast<Arena*>(a)->release(static_cast<size_t>(mark)) ? 0 : -1;
}
// out[0..3] = capacity, used, peak, number of allocations
LC_EXPORT void lc_rt_arena_stats(void* a, long long* out) {
  Arena* p = static_cast<Arena*>(a);
  out[0] = static_cast<long long>(p->capacity());
  out[1] = static_cast<long long>(p->used());
  out[2] = static_cast<long long>(p->peak());
  out[3] = static_cast<long long>(p->allocations());
}
LC_EXPORT int lc_rt_mem_info(long long* free_bytes, long long* total_bytes) {
  size_t f = 0, t = 0;
  hipError_t e = hipMemGetInfo(&f, &t);
  *free_bytes = static_cast<long long>(f);
  *total_bytes = static_cast<long long>(t);
  return static_cast<int>(e);
}

// ---- streams, events, copies ----
LC_EXPORT void* lc_rt_stream_create(int priority) {
  Stream* s = new Stream(priority);
  if (!s->get()) {
    delete s;
    return nullptr;
  }
  return s;
}
LC_EXPORT void lc_rt_stream_destroy(void* s) { delete static_cast<Stream*>(s); }
LC_EXPORT void* lc_rt_stream_handle(void* s) { return static_cast<Stream*>(s)->get(); }
LC_EXPORT int lc_rt_stream_sync(void* raw_stream) {
  return static_cast<int>(hipStreamSynchronize(static_cast<hipStream_t>(raw_stream)));
}
LC_EXPORT void* lc_rt_event_create(int timing) {
  Event* e = new Event(timing != 0);
  if (!e->get()) {
    delete e;
    return nullptr;
  }
  return e;
}
LC_EXPORT void lc_rt_event_destroy(void* e) { delete static_cast<Event*>(e); }
LC_EXPORT int lc_rt_event_record(void* e, void* raw_stream) {
  return static_cast<int>(hipEventRecord(static_cast<Event*>(e)->get(), static_cast<hipStream_t>(raw_stream)));
}
LC_EXPORT int lc_rt_event_sync(void* e) { return static_cast<int>(hipEventSynchronize(static_cast<Event*>(e)->get())); }
// 1 = done, 0 = still pending, < 0 = error
LC_EXPORT int lc_rt_event_query(void* e) {
  hipError_t r = hipEventQuery(static_cast<Event*>(e)->get());
  if (r == hipSuccess) return 1;
  if (r == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  return -static_cast<int>(r);
}
LC_EXPORT int lc_rt_stream_wait_event(void* raw_stream, void* e) {
  return static_cast<int>(hipStreamWaitEvent(static_cast<hipStream_t>(raw_stream), static_cast<Event*>(e)->get(), 0));
}
LC_EXPORT float lc_rt_event_elapsed_ms(void* start, void* stop) {
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, static_cast<Event*>(start)->get(), static_cast<Event*>(stop)->get()) != hipSuccess) {
    (void)hipGetLastError();
    return -1.f;
  }
  return ms;
}
// kind: 0 host -> device, 1 device -> host, 2 device -> device
LC_EXPORT int lc_rt_copy_async(void* dst, const void* src, long long bytes, int kind, void* raw_stream) {
  if (bytes <= 0) return 0;
  const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  return static_cast<int>(hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), k, static_cast<hipStream_t>(raw_stream)));
}

// ---- graphs ----
LC_EXPORT void* lc_rt_graphs_create() { return new GraphCache(); }
LC_EXPORT void lc_rt_graphs_destroy(void* g) { delete static_cast<GraphCache*>(g); }
LC_EXPORT int lc_rt_graphs_begin(void* g, void* raw_stream) {
  return static_cast<GraphCache*>(g)->begin(static_cast<hipStream_t>(raw_stream));
}
LC_EXPORT long long lc_rt_graphs_end(void* g, void* raw_stream, long long key) {
  return static_cast<GraphCache*>(g)->end(static_cast<hipStream_t>(raw_stream), key);
}
LC_EXPORT void lc_rt_graphs_abort(void* g, void* raw_stream) {
  static_cast<GraphCache*>(g)->abort(static_cast<hipStream_t>(raw_stream));
}
LC_EXPORT int lc_rt_graphs_launch(void* g, long long key, void* raw_stream) {
  return static_cast<GraphCache*>(g)->launch(key, static_cast<hipStream_t>(raw_stream));
}
LC_EXPORT int lc_rt_graphs_erase(void* g, long long key) { return static_cast<GraphCache*>(g)->erase(key) ? 0 : -1; }
LC_EXPORT long long lc_rt_graphs_size(void* g) { return static_cast<long long>(static_cast<GraphCache*>(g)->size()); }
LC_EXPORT long long lc_rt_graphs_launches(void* g) {
  return static_cast<long long>(static_cast<GraphCache*>(g)->launches());
}
