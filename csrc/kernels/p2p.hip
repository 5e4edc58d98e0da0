// Peer-to-peer collectives over xGMI for one tensor/expert-parallel group (2, 4 or 8 ranks,
// one process per GPU): every rank maps the other ranks' exchange regions (HIP IPC) and the
// kernels below read and write them directly. No RCCL call is involved, so the kernels can sit
// inside a captured hipGraph like any other launch, and the all-reduce carries its consumer
// with it: sum of the ranks' partial products + residual add + RMSNorm in one launch
// (two all-reduces per layer are the latency floor of a tensor-parallel decode step).
//
// The reference has no collective of any kind (/root/reference/src/main.rs has one remote
// model behind an HTTPS call); this is the tensor-parallel judge's analogue of its fan-in.
//
// Region of one rank (allocated uncached / fine-grained, zeroed once):
//
//   header   epoch (calls completed), done (ticket of the running call), error
//   flags1   [W][MAXB] u32   flags1[s][b] == e : rank s published block b's rows of call e
//   flags2   [W][MAXB] u32   same for the second phase of the two-shot form
//   stage    2 slots         this rank's contribution to call e lives in slot e & 1
//   red      2 slots         two-shot: the rows this rank reduced
//
// Protocol of one call (epoch e = header.epoch + 1, identical on all ranks because the ranks
// issue the same calls in the same order):
//   publish   every block writes its rows into the own stage slot with write-through stores at
//             system scope (sc0 sc1), every wave drains its stores (s_waitcnt vmcnt(0)), the
//             block synchronises, and one lane per peer stores e into flags1[rank][b] OF THAT
//             PEER's region, so pollers poll their own memory.
//   consume   one lane per peer polls the own flags1[peer][b] (relaxed, system scope, bounded
//             by a wall-clock timeout) until it reads e, the block synchronises, and the rows
//             are read from the peers' stage slots with sc0 sc1 loads (no cache between the
//             fabric and the registers), summed in fp32 IN RANK ORDER (every rank adds the same
//             bf16 values in the same order, so all ranks hold bit-identical results and
//             sample identical tokens).
//   finish    the block that draws the last ticket resets `done` and stores epoch = e.
// Slot reuse needs no second barrier: a rank writes slot e & 1 again in call e + 2, after its
// call e + 1 has ended; in call e + 1 it saw every peer's flag e + 1, which that peer stored
// in its own call e + 1, i.e. after the peer's call e (the last reader of the slot) had ended.
// Blocks are independent: block b of a rank waits only for block b of the peers, which is
// why every rank must launch a call with the same geometry (same T, d => same grid).
// Residency: a waiting block holds its CU slot. With one rank per GPU a block only ever waits
// for another GPU, and blocks are dispatched in index order, so the lowest unfinished index of
// every rank is always resident and the call progresses whatever the grid size. When several
// ranks share ONE GPU (tests, or two small groups packed on a device) their waiting blocks
// compete for the same slots, so all of them must fit at once: grids are capped (default
// kGridCap blocks, rows beyond that are walked by a block-stride loop; 8 ranks x 64 blocks of
// 256 threads = the 512 blocks a chip holds at two blocks per CU, which is what the reduction
// kernels' register budget admits). Ranks that share a GPU
// with each other AND with kernels that need a whole CU (160 KiB of LDS, all registers of a
// SIMD) should set the cap lower still, so that free CUs always remain (lc_p2p_comm_create).
#include "common.h"

#include <cstring>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 1024;
constexpr int kHeaderBytes = 256;
constexpr int kThreads = 256;
constexpr int kGridCap = 64;

struct P2PRegion {   // byte offsets inside every rank's region (same on all ranks)
  uint32_t flags1, flags2, stage, red;
  uint32_t stage_bytes, red_bytes;   // per slot
};

struct P2PPeers {
  char* base[kMaxRanks];
};

struct Header {
  uint32_t epoch, done, error, pad;
};

LC_DEVICE uint32_t load_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
LC_DEVICE void store_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// aux 17 = sc0 | sc1: system scope, write-through / cache-bypassing.
LC_DEVICE __amdgpu_buffer_rsrc_t rsrc_of(char* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
}
LC_DEVICE u32x4 load16_sys(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17);
}
LC_DEVICE void store16_sys(__amdgpu_buffer_rsrc_t r, uint32_t off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 17);
}

LC_DEVICE void unpack8(u32x4 v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = __uint_as_float(v[j] << 16);
    f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
  }
}
LC_DEVICE u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
  return v;
}

// Publishes "this block's rows of call `epoch` are in place" to every peer. Call with all
// threads; the stores it covers must have been issued by this block's threads before.
template <int W>
LC_DEVICE void publish(const P2PPeers& peers, uint32_t flags_off, int rank, uint32_t epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x < W && static_cast<int>(threadIdx.x) != rank) {
    uint32_t* f = reinterpret_cast<uint32_t*>(peers.base[threadIdx.x] + flags_off) + rank * kMaxBlocks + blockIdx.x;
    store_sys(f, epoch);
  }
}

// Waits until every peer has published block b of call `epoch` (or the timeout passes: then the
// error word is set and false is returned to all threads). `timeout` is in 10 ns ticks.
template <int W>
LC_DEVICE bool wait_peers(char* self, uint32_t flags_off, int rank, uint32_t epoch, unsigned long long timeout,
                          int* shared_ok) {
  // A communicator that has timed out once stays failed: later calls do not wait again (the
  // host reads the error word after its next synchronisation and raises).
  if (threadIdx.x == 0) *shared_ok = load_sys(&reinterpret_cast<Header*>(self)->error) == 0 ? 1 : 0;
  __syncthreads();
  if (*shared_ok != 0 && threadIdx.x < W && static_cast<int>(threadIdx.x) != rank) {
    const uint32_t* f = reinterpret_cast<const uint32_t*>(self + flags_off) + threadIdx.x * kMaxBlocks + blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // flags only grow; "at least this call" (wrap-safe) also accepts a peer that publishes
    // without consuming and therefore runs ahead (publish_only below)
    while (static_cast<int32_t>(load_sys(f) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
        store_sys(&reinterpret_cast<Header*>(self)->error, 1u + threadIdx.x);
        *shared_ok = 0;
        break;
      }
    }
  }
  __syncthreads();
  return *shared_ok != 0;
}

// Last block of the call advances the epoch (all blocks have read it at their start).
LC_DEVICE void finish_call(char* self, uint32_t epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    Header* h = reinterpret_cast<Header*>(self);
    const uint32_t t = __hip_atomic_fetch_add(&h->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1 == gridDim.x) {
      __hip_atomic_store(&h->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&h->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct ReduceArgs {
  const float* part;      // [S, T, d] fp32 split-K slabs of the local partial product, or null
  const bf16_t* xin;      // [T, d] bf16 local partial product (when part is null)
  bf16_t* h;              // [T, d] residual stream, updated in place (null: plain all-reduce)
  const bf16_t* w;        // [d] RMSNorm weight (with h)
  bf16_t* y;              // [T, d] RMSNorm(h) * w, or the plain sum
  int S, T, d, rank;
  float eps;
  unsigned long long timeout;
  int publish_only;       // stage + flag the local rows, skip the wait and the reduction (y, h untouched)
  size_t slab_stride;     // elements between two slabs of `part` (T * d, or more when the T rows are a
                          // range of a taller slab stack)
};

// The local partial row chunk (8 elements) as fp32.
LC_DEVICE void local_chunk(const ReduceArgs& a, int r, int c, float* f) {
  const size_t off = static_cast<size_t>(r) * a.d + c * 8;
  if (a.part != nullptr) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 0.f;
    for (int s = 0; s < a.S; ++s) {
      const float4* p = reinterpret_cast<const float4*>(a.part + static_cast<size_t>(s) * a.slab_stride + off);
      const float4 lo = p[0], hi = p[1];
      f[0] += lo.x; f[1] += lo.y; f[2] += lo.z; f[3] += lo.w;
      f[4] += hi.x; f[5] += hi.y; f[6] += hi.z; f[7] += hi.w;
    }
  } else {
    unpack8(*reinterpret_cast<const u32x4*>(a.xin + off), f);
  }
}

// What the epilogue of a row needs besides the reduced delta: the residual row and the norm
// weight. Loaded BEFORE the wait for the peers, so their latency overlaps the hand-off.
template <int MAXC>
struct RowOperands {
  u32x4 h[MAXC], w[MAXC];
};

template <int MAXC>
LC_DEVICE void load_row_operands(const ReduceArgs& a, int r, RowOperands<MAXC>& ro) {
  if (a.h == nullptr) return;
  const int nchunk = a.d >> 3;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = threadIdx.x + k * kThreads;
    if (c < nchunk) {
      ro.h[k] = *reinterpret_cast<const u32x4*>(a.h + static_cast<size_t>(r) * a.d + c * 8);
      ro.w[k] = *reinterpret_cast<const u32x4*>(a.w + c * 8);
    }
  }
}

// Residual add + RMSNorm of one row held as MAXC chunks per thread (acc = the all-reduced
// delta). Writes h and y. `red` = 16 floats of LDS.
template <int MAXC>
LC_DEVICE void row_epilogue(const ReduceArgs& a, int r, float (&acc)[MAXC][8], const RowOperands<MAXC>& ro,
                            float* red) {
  const int nchunk = a.d >> 3;
  if (a.h == nullptr) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = threadIdx.x + k * kThreads;
      if (c < nchunk) *reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(r) * a.d + c * 8) = pack8(acc[k]);
    }
    return;
  }
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = threadIdx.x + k * kThreads;
    if (c < nchunk) {
      float hv[8];
      unpack8(ro.h[k], hv);
      // the delta enters the residual stream rounded to bf16, as a stored all-reduce result would
      float dv[8];
      unpack8(pack8(acc[k]), dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] += dv[j];
      const u32x4 packed = pack8(hv);
      *reinterpret_cast<u32x4*>(a.h + static_cast<size_t>(r) * a.d + c * 8) = packed;
      unpack8(packed, acc[k]);                       // normalise what was stored
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += acc[k][j] * acc[k][j];
    }
  }
  ss = block_sum(ss, red);
  const float inv = rsqrtf(ss / static_cast<float>(a.d) + a.eps);
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = threadIdx.x + k * kThreads;
    if (c < nchunk) {
      float wv[8], o[8];
      unpack8(ro.w[k], wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = acc[k][j] * inv * wv[j];
      *reinterpret_cast<u32x4*>(a.y + static_cast<size_t>(r) * a.d + c * 8) = pack8(o);
    }
  }
}

// One-shot all-reduce (+ residual + RMSNorm): every rank reads every peer's rows.
template <int W, int MAXC>
__global__ __launch_bounds__(kThreads, 2) void p2p_allreduce_kernel(P2PPeers peers, P2PRegion reg, ReduceArgs a) {
  __shared__ float red[16];
  __shared__ int ok;
  char* self = peers.base[a.rank];
  const uint32_t epoch = reinterpret_cast<const Header*>(self)->epoch + 1;
  const uint32_t slot_off = reg.stage + (epoch & 1u) * reg.stage_bytes;
  const int nchunk = a.d >> 3;
  RowOperands<MAXC> ro;
  if (static_cast<int>(blockIdx.x) < a.T && !a.publish_only) load_row_operands<MAXC>(a, blockIdx.x, ro);
  // ---- publish this block's rows
  {
    const __amdgpu_buffer_rsrc_t mine = rsrc_of(self + slot_off, reg.stage_bytes);
    for (int r = blockIdx.x; r < a.T; r += gridDim.x)
      for (int c = threadIdx.x; c < nchunk; c += kThreads) {
        float f[8];
        local_chunk(a, r, c, f);
        store16_sys(mine, static_cast<uint32_t>((static_cast<size_t>(r) * a.d + c * 8) * 2), pack8(f));
      }
  }
  publish<W>(peers, reg.flags1, a.rank, epoch);
  // ---- consume
  if (!a.publish_only && wait_peers<W>(self, reg.flags1, a.rank, epoch, a.timeout, &ok)) {
    for (int r = blockIdx.x; r < a.T; r += gridDim.x) {
      if (r != static_cast<int>(blockIdx.x)) load_row_operands<MAXC>(a, r, ro);
      u32x4 raw[W][MAXC];
#pragma unroll
      for (int s = 0; s < W; ++s) {
        const __amdgpu_buffer_rsrc_t src = rsrc_of(peers.base[s] + slot_off, reg.stage_bytes);
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          const int c = threadIdx.x + k * kThreads;
          if (c < nchunk) raw[s][k] = load16_sys(src, static_cast<uint32_t>((static_cast<size_t>(r) * a.d + c * 8) * 2));
        }
      }
      float acc[MAXC][8];
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
        if (static_cast<int>(threadIdx.x) + k * kThreads < nchunk) {
#pragma unroll
          for (int s = 0; s < W; ++s) {
            float f[8];
            unpack8(raw[s][k], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] += f[j];
          }
        }
      }
      row_epilogue<MAXC>(a, r, acc, ro, red);
    }
  }
  finish_call(self, epoch);
}

// Two-shot form for long messages: rank q reduces the rows r with (r / rows_per_rank) == q from
// all peers (reduce-scatter), publishes the sums, and every rank then gathers the sums of the
// rows it does not own (all-gather) and applies residual + norm to all rows. Per rank
// 2 (W-1)/W of the message crosses the fabric instead of W-1 times.
template <int W, int MAXC>
__global__ __launch_bounds__(kThreads, 2) void p2p_allreduce2_kernel(P2PPeers peers, P2PRegion reg, ReduceArgs a) {
  __shared__ float red[16];
  __shared__ int ok;
  char* self = peers.base[a.rank];
  const uint32_t epoch = reinterpret_cast<const Header*>(self)->epoch + 1;
  const uint32_t slot_off = reg.stage + (epoch & 1u) * reg.stage_bytes;
  const uint32_t red_off = reg.red + (epoch & 1u) * reg.red_bytes;
  const int nchunk = a.d >> 3;
  const int rpr = (a.T + W - 1) / W;                         // rows per owner
  {
    const __amdgpu_buffer_rsrc_t mine = rsrc_of(self + slot_off, reg.stage_bytes);
    for (int r = blockIdx.x; r < a.T; r += gridDim.x)
      for (int c = threadIdx.x; c < nchunk; c += kThreads) {
        float f[8];
        local_chunk(a, r, c, f);
        store16_sys(mine, static_cast<uint32_t>((static_cast<size_t>(r) * a.d + c * 8) * 2), pack8(f));
      }
  }
  publish<W>(peers, reg.flags1, a.rank, epoch);
  bool good = wait_peers<W>(self, reg.flags1, a.rank, epoch, a.timeout, &ok);
  // ---- reduce the rows this rank owns (same row -> block map as the publish), publish the sums
  if (good) {
    const __amdgpu_buffer_rsrc_t myred = rsrc_of(self + red_off, reg.red_bytes);
    for (int r = blockIdx.x; r < a.T; r += gridDim.x) {
      if (r / rpr != a.rank) continue;
      const uint32_t out_off = static_cast<uint32_t>((static_cast<size_t>(r - a.rank * rpr) * a.d) * 2);
      for (int c = threadIdx.x; c < nchunk; c += kThreads) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 raw[W];
#pragma unroll
        for (int s = 0; s < W; ++s)
          raw[s] = load16_sys(rsrc_of(peers.base[s] + slot_off, reg.stage_bytes),
                              static_cast<uint32_t>((static_cast<size_t>(r) * a.d + c * 8) * 2));
#pragma unroll
        for (int s = 0; s < W; ++s) {
          float f[8];
          unpack8(raw[s], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        store16_sys(myred, out_off + c * 16, pack8(acc));
      }
    }
  }
  publish<W>(peers, reg.flags2, a.rank, epoch);
  good = wait_peers<W>(self, reg.flags2, a.rank, epoch, a.timeout, &ok) && good;
  // ---- gather every row's sum from its owner, residual + norm
  if (good) {
    for (int r = blockIdx.x; r < a.T; r += gridDim.x) {
      const int owner = r / rpr;
      const __amdgpu_buffer_rsrc_t src = rsrc_of(peers.base[owner] + red_off, reg.red_bytes);
      const uint32_t row_off = static_cast<uint32_t>((static_cast<size_t>(r - owner * rpr) * a.d) * 2);
      RowOperands<MAXC> ro;
      load_row_operands<MAXC>(a, r, ro);
      float acc[MAXC][8];
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int c = threadIdx.x + k * kThreads;
        if (c < nchunk) {
          unpack8(load16_sys(src, row_off + c * 16), acc[k]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
        }
      }
      row_epilogue<MAXC>(a, r, acc, ro, red);
    }
  }
  finish_call(self, epoch);
}

struct ExchangeArgs {
  const char* src;   // local input: nblk blocks of m bytes
  char* out;         // W blocks of m bytes: block s = what rank s contributed
  uint32_t m;        // bytes per block
  uint32_t m_pad;    // bytes per block inside the staging slot (m rounded up to 16)
  int nblk;          // 1: all-gather (everyone reads block 0 of every peer); W: all-to-all (rank q reads block q)
  int rank;
  unsigned long long timeout;
  int publish_only;  // stage + flag only (see lc_p2p_allreduce)
};

LC_DEVICE uint32_t load4_sys(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 17);
}
LC_DEVICE void store4_sys(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, 17);
}

// All-gather / equal-split all-to-all of raw bytes. Block b owns the 16-byte chunks j with
// (j / chunks_per_block) == b of EVERY block, on the publishing and the reading side. Blocks of
// m bytes that are not a multiple of 16 (a handful of counters, odd row widths) take the
// narrow form: 4-byte units through the staging slot, whose blocks are m_pad apart; the last
// unit of a block whose length is not a multiple of 4 is assembled bytewise, so neither the
// input nor the output is touched beyond m bytes per block.
template <int W>
__global__ __launch_bounds__(kThreads) void p2p_exchange_kernel(P2PPeers peers, P2PRegion reg, ExchangeArgs a) {
  __shared__ int ok;
  char* self = peers.base[a.rank];
  const uint32_t epoch = reinterpret_cast<const Header*>(self)->epoch + 1;
  const uint32_t slot_off = reg.stage + (epoch & 1u) * reg.stage_bytes;
  const bool wide = (a.m & 15u) == 0;
  const uint32_t unit = wide ? 16u : 4u;
  const uint32_t chunks = (a.m + unit - 1) / unit;
  const uint32_t cpb = (chunks + gridDim.x - 1) / gridDim.x;
  const uint32_t j0 = blockIdx.x * cpb, j1 = min(chunks, j0 + cpb);
  {
    const __amdgpu_buffer_rsrc_t mine = rsrc_of(self + slot_off, reg.stage_bytes);
    for (int t = 0; t < a.nblk; ++t)
      for (uint32_t j = j0 + threadIdx.x; j < j1; j += kThreads) {
        if (wide) {
          store16_sys(mine, t * a.m_pad + j * 16, *reinterpret_cast<const u32x4*>(a.src + static_cast<size_t>(t) * a.m + j * 16));
        } else {
          const char* p = a.src + static_cast<size_t>(t) * a.m + j * 4;
          uint32_t v = 0;
          const uint32_t nb = min(4u, a.m - j * 4);
          if (nb == 4 && (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
            v = *reinterpret_cast<const uint32_t*>(p);
          } else {
            for (uint32_t k = 0; k < nb; ++k) v |= static_cast<uint32_t>(static_cast<uint8_t>(p[k])) << (8 * k);
          }
          store4_sys(mine, t * a.m_pad + j * 4, v);
        }
      }
  }
  publish<W>(peers, reg.flags1, a.rank, epoch);
  if (!a.publish_only && wait_peers<W>(self, reg.flags1, a.rank, epoch, a.timeout, &ok)) {
    const uint32_t sel = a.nblk == 1 ? 0u : static_cast<uint32_t>(a.rank) * a.m_pad;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const __amdgpu_buffer_rsrc_t src = rsrc_of(peers.base[s] + slot_off, reg.stage_bytes);
      for (uint32_t j = j0 + threadIdx.x; j < j1; j += kThreads) {
        if (wide) {
          *reinterpret_cast<u32x4*>(a.out + static_cast<size_t>(s) * a.m + j * 16) = load16_sys(src, sel + j * 16);
        } else {
          const uint32_t v = load4_sys(src, sel + j * 4);
          char* q = a.out + static_cast<size_t>(s) * a.m + j * 4;
          const uint32_t nb = min(4u, a.m - j * 4);
          if (nb == 4 && (reinterpret_cast<uintptr_t>(q) & 3) == 0) {
            *reinterpret_cast<uint32_t*>(q) = v;
          } else {
            for (uint32_t k = 0; k < nb; ++k) q[k] = static_cast<char>((v >> (8 * k)) & 0xffu);
          }
        }
      }
    }
  }
  finish_call(self, epoch);
}

struct CountedArgs {
  const char* src;        // W blocks of cap rows: block t holds the rows bound for rank t
  char* out;              // W blocks of cap rows: block s receives the rows rank s sent here
  const int32_t* counts;  // [W * per] rows per (destination rank, slot); a destination's rows = sum of its `per` slots
  int32_t* recv_counts;   // [W * per] out: entry s * per + e = what rank s reports for (this rank, slot e); may be null
  uint32_t row_bytes;     // multiple of 16
  uint32_t cap;           // rows per block
  int per, rank;
  unsigned long long timeout;
};

constexpr uint32_t kCountHeader = 4096;   // bytes at the head of the staging slot that carry the counts

// All-to-all of row blocks with device-side row counts: only the rows that exist cross the
// fabric, although every (source, destination) pair owns `cap` rows of both buffers. The
// publisher puts its whole count table at the head of its staging slot; a reader sums the `per`
// entries addressed to it to learn how many rows to fetch from that peer, and hands the entries
// on as recv_counts. Nothing about the counts is known to the host: an expert-parallel MoE layer
// of any size is plain kernel launches (capturable, no read-back). Rows of `out` beyond the
// counts are left untouched.
template <int W>
__global__ __launch_bounds__(kThreads) void p2p_exchange_counted_kernel(P2PPeers peers, P2PRegion reg, CountedArgs a) {
  __shared__ int ok;
  __shared__ uint32_t rows_to[W], rows_from[W];
  char* self = peers.base[a.rank];
  const uint32_t epoch = reinterpret_cast<const Header*>(self)->epoch + 1;
  const uint32_t slot_off = reg.stage + (epoch & 1u) * reg.stage_bytes;
  const uint32_t block_bytes = a.cap * a.row_bytes;
  const uint32_t cpr = a.row_bytes >> 4;                               // 16-byte chunks per row
  const uint32_t chunks = a.cap * cpr;                                 // per block, at most
  const uint32_t cpb = (chunks + gridDim.x - 1) / gridDim.x;
  const uint32_t j0 = blockIdx.x * cpb, j1 = min(chunks, j0 + cpb);
  const __amdgpu_buffer_rsrc_t mine = rsrc_of(self + slot_off, reg.stage_bytes);
  if (threadIdx.x < W) {
    uint32_t n = 0;
    for (int e = 0; e < a.per; ++e) n += static_cast<uint32_t>(max(0, a.counts[threadIdx.x * a.per + e]));
    rows_to[threadIdx.x] = min(n, a.cap);
  }
  // every block publishes the count table (same values; a reader may be served by any of them)
  for (uint32_t i = threadIdx.x; i < static_cast<uint32_t>(W * a.per); i += kThreads)
    store4_sys(mine, i * 4, static_cast<uint32_t>(a.counts[i]));
  __syncthreads();
  for (int t = 0; t < W; ++t) {
    const uint32_t lim = min(j1, rows_to[t] * cpr);
    for (uint32_t j = j0 + threadIdx.x; j < lim; j += kThreads)
      store16_sys(mine, kCountHeader + t * block_bytes + j * 16,
                  *reinterpret_cast<const u32x4*>(a.src + static_cast<size_t>(t) * block_bytes + j * 16));
  }
  publish<W>(peers, reg.flags1, a.rank, epoch);
  if (wait_peers<W>(self, reg.flags1, a.rank, epoch, a.timeout, &ok)) {
    if (threadIdx.x < W) {
      const __amdgpu_buffer_rsrc_t src = rsrc_of(peers.base[threadIdx.x] + slot_off, reg.stage_bytes);
      uint32_t n = 0;
      for (int e = 0; e < a.per; ++e) {
        const int32_t c = static_cast<int32_t>(load4_sys(src, (a.rank * a.per + e) * 4));
        n += static_cast<uint32_t>(max(0, c));
        if (blockIdx.x == 0 && a.recv_counts != nullptr) a.recv_counts[threadIdx.x * a.per + e] = c;
      }
      rows_from[threadIdx.x] = min(n, a.cap);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const __amdgpu_buffer_rsrc_t src = rsrc_of(peers.base[s] + slot_off, reg.stage_bytes);
      const uint32_t lim = min(j1, rows_from[s] * cpr);
      for (uint32_t j = j0 + threadIdx.x; j < lim; j += kThreads)
        *reinterpret_cast<u32x4*>(a.out + static_cast<size_t>(s) * block_bytes + j * 16) =
            load16_sys(src, kCountHeader + a.rank * block_bytes + j * 16);
    }
  }
  finish_call(self, epoch);
}

struct CommState {
  int world, rank;
  P2PPeers peers;
  P2PRegion reg;
  size_t region_bytes;
  unsigned long long timeout;
  int grid_cap;
};

P2PRegion layout(uint32_t stage_bytes, uint32_t red_bytes, size_t* total) {
  P2PRegion r;
  r.flags1 = kHeaderBytes;
  r.flags2 = r.flags1 + kMaxRanks * kMaxBlocks * 4;
  r.stage = r.flags2 + kMaxRanks * kMaxBlocks * 4;
  r.stage_bytes = stage_bytes;
  r.red = r.stage + 2 * stage_bytes;
  r.red_bytes = red_bytes;
  *total = static_cast<size_t>(r.red) + 2 * static_cast<size_t>(red_bytes);
  return r;
}

int grid_for_rows(int T, int cap) { return std::max(1, std::min(T, cap)); }

}  // namespace

// ---- host side: regions, handles, communicator objects --------------------------------------
// Size of a rank's exchange region for `stage_bytes` per staging slot (rounded up to 4 KiB).
LC_EXPORT long long lc_p2p_region_bytes(long long stage_bytes) {
  size_t total = 0;
  const uint32_t sb = static_cast<uint32_t>((stage_bytes + 4095) / 4096 * 4096);
  layout(sb, sb, &total);
  return static_cast<long long>(total);
}

// Allocates and zeroes a region on the current device. Uncached (fine-grained) memory: remote
// and local accesses see each other's writes without cache maintenance. Writes the 64-byte IPC
// handle of the allocation to `handle_out` (for other processes).
LC_EXPORT int lc_p2p_alloc(long long bytes, void** ptr, void* handle_out) {
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, static_cast<size_t>(bytes), hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, static_cast<size_t>(bytes), hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) return static_cast<int>(e);
  // Only the header and the flag arrays carry state between calls; the staging slots are
  // written before they are read.
  const size_t control = kHeaderBytes + 2 * static_cast<size_t>(kMaxRanks) * kMaxBlocks * 4;
  e = hipMemset(p, 0, std::min(control, static_cast<size_t>(bytes)));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess && handle_out != nullptr) {
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e == hipSuccess) std::memcpy(handle_out, &h, sizeof(h));
  }
  if (e != hipSuccess) {
    (void)hipFree(p);
    return static_cast<int>(e);
  }
  *ptr = p;
  return 0;
}

LC_EXPORT int lc_p2p_open(const void* handle, void** ptr) {
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  return static_cast<int>(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}

LC_EXPORT int lc_p2p_close(void* ptr) { return static_cast<int>(hipIpcCloseMemHandle(ptr)); }
LC_EXPORT int lc_p2p_free(void* ptr) { return static_cast<int>(hipFree(ptr)); }
LC_EXPORT int lc_p2p_handle_bytes() { return static_cast<int>(sizeof(hipIpcMemHandle_t)); }

// A communicator = this rank's view of the group's regions. `bases[world]` are the addresses of
// all ranks' regions in THIS process (own allocation at index `rank`, peers through lc_p2p_open
// or, for ranks that are threads of one process, the peers' own pointers).
LC_EXPORT void* lc_p2p_comm_create(int world, int rank, void* const* bases, long long stage_bytes,
                                   double timeout_s, int grid_cap) {
  if ((world != 2 && world != 4 && world != 8) || rank < 0 || rank >= world) return nullptr;
  CommState* c = new CommState();
  c->world = world;
  c->rank = rank;
  for (int i = 0; i < kMaxRanks; ++i) c->peers.base[i] = i < world ? static_cast<char*>(bases[i]) : nullptr;
  const uint32_t sb = static_cast<uint32_t>((stage_bytes + 4095) / 4096 * 4096);
  c->reg = layout(sb, sb, &c->region_bytes);
  c->timeout = static_cast<unsigned long long>(timeout_s * 1e8);   // s_memrealtime ticks at 100 MHz
  c->grid_cap = grid_cap > 0 ? std::min(grid_cap, kMaxBlocks) : kGridCap;
  return c;
}

LC_EXPORT void lc_p2p_comm_destroy(void* comm) { delete static_cast<CommState*>(comm); }

// Error word of this rank's region (0 = fine; 1 + peer = timed out waiting for that peer).
// Synchronises the device.
LC_EXPORT int lc_p2p_comm_error(void* comm) {
  CommState* c = static_cast<CommState*>(comm);
  Header h;
  if (hipMemcpy(&h, c->peers.base[c->rank], sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return static_cast<int>(h.error);
}

LC_EXPORT long long lc_p2p_comm_capacity(void* comm) { return static_cast<CommState*>(comm)->reg.stage_bytes; }

// How long the kernels launched from now on wait for a peer (the first-contact check of a new
// group uses a few seconds; a serving group a limit that outlasts a peer's longest other step).
LC_EXPORT void lc_p2p_comm_set_timeout(void* comm, double timeout_s) {
  static_cast<CommState*>(comm)->timeout = static_cast<unsigned long long>(timeout_s * 1e8);
}

#define LC_P2P_BY_WORLD(KERNEL, ...)                                           \
  switch (c->world) {                                                          \
    case 2: KERNEL(2, __VA_ARGS__); break;                                     \
    case 4: KERNEL(4, __VA_ARGS__); break;                                     \
    default: KERNEL(8, __VA_ARGS__); break;                                    \
  }

// y = RMSNorm(h += sum over ranks of the local partial product) * w   (h, w given), or
// y = sum over ranks (h null). The local partial is either S fp32 slabs [S, T, d] (`part`;
// `slab_stride` elements apart, 0 = T * d) or a bf16 [T, d] tensor (`xin`). two_shot selects the reduce-scatter + all-gather form. Every
// rank must make the same call (same T, d, two_shot).
// publish_only (one-shot form only) is for measurements where the peers of the rank under
// test merely have to play their part of the protocol: they stage their rows, raise their
// flags and leave, without waiting for anybody or holding a CU while they do.
LC_EXPORT int lc_p2p_allreduce(void* comm, const void* part, int S, long long slab_stride, const void* xin, void* h,
                               const void* w, void* y, int T, int d, float eps, int two_shot, int publish_only,
                               hipStream_t stream) {
  CommState* c = static_cast<CommState*>(comm);
  if (T <= 0) return 0;
  if (d % 8 || d > 8 * kThreads * 8 || (part == nullptr) == (xin == nullptr) || (part && S < 1) || (h && !w)) return -1;
  if (static_cast<size_t>(T) * d * 2 > c->reg.stage_bytes) return -2;
  ReduceArgs a{static_cast<const float*>(part), static_cast<const bf16_t*>(xin), static_cast<bf16_t*>(h),
               static_cast<const bf16_t*>(w), static_cast<bf16_t*>(y), S, T, d, c->rank, eps, c->timeout,
               publish_only, slab_stride > 0 ? static_cast<size_t>(slab_stride) : static_cast<size_t>(T) * d};
  if (publish_only && two_shot) return -1;
  const dim3 grid(grid_for_rows(T, c->grid_cap)), block(kThreads);
  const int maxc = d <= 8 * kThreads ? 1 : (d <= 32 * kThreads ? 4 : 8);
#define LC_LAUNCH1(W_, MAXC_) p2p_allreduce_kernel<W_, MAXC_><<<grid, block, 0, stream>>>(c->peers, c->reg, a)
#define LC_LAUNCH2(W_, MAXC_) p2p_allreduce2_kernel<W_, MAXC_><<<grid, block, 0, stream>>>(c->peers, c->reg, a)
#define LC_PICK(W_, MAXC_)                     \
  do {                                         \
    if (two_shot) LC_LAUNCH2(W_, MAXC_);       \
    else LC_LAUNCH1(W_, MAXC_);                \
  } while (0)
  if (maxc == 1) {
    LC_P2P_BY_WORLD(LC_PICK, 1)
  } else if (maxc == 4) {
    LC_P2P_BY_WORLD(LC_PICK, 4)
  } else {
    LC_P2P_BY_WORLD(LC_PICK, 8)
  }
#undef LC_PICK
#undef LC_LAUNCH1
#undef LC_LAUNCH2
  LC_RETURN_LAST_ERROR();
}

// All-gather (nblk = 1: src holds m bytes, out world * m) or equal-split all-to-all (nblk =
// world: src and out hold world blocks of m bytes). Any m > 0.
LC_EXPORT int lc_p2p_exchange(void* comm, const void* src, void* out, long long m, int all_to_all,
                              int publish_only, hipStream_t stream) {
  CommState* c = static_cast<CommState*>(comm);
  if (m <= 0) return 0;
  const int nblk = all_to_all ? c->world : 1;
  const long long m_pad = (m + 15) / 16 * 16;
  if (static_cast<size_t>(m_pad) * nblk > c->reg.stage_bytes) return -2;
  ExchangeArgs a{static_cast<const char*>(src), static_cast<char*>(out), static_cast<uint32_t>(m),
                 static_cast<uint32_t>(m_pad), nblk, c->rank, c->timeout, publish_only};
  const long long chunks = m_pad / 16;
  const int blocks = static_cast<int>(std::max<long long>(1, std::min<long long>(c->grid_cap, (chunks + 1023) / 1024)));
  const dim3 grid(blocks), block(kThreads);
#define LC_LAUNCHX(W_, unused) p2p_exchange_kernel<W_><<<grid, block, 0, stream>>>(c->peers, c->reg, a)
  LC_P2P_BY_WORLD(LC_LAUNCHX, 0)
#undef LC_LAUNCHX
  LC_RETURN_LAST_ERROR();
}

// Rows a counted exchange can carry per (source, destination) pair for rows of `row_bytes`.
LC_EXPORT long long lc_p2p_counted_cap(void* comm, long long row_bytes) {
  CommState* c = static_cast<CommState*>(comm);
  if (row_bytes <= 0 || c->reg.stage_bytes <= kCountHeader) return 0;
  return static_cast<long long>((c->reg.stage_bytes - kCountHeader) / (static_cast<size_t>(c->world) * row_bytes));
}

// All-to-all of row blocks with device-side counts (see p2p_exchange_counted_kernel). src / out:
// world blocks of cap rows of row_bytes (a multiple of 16); counts: int32 [world * per] on the
// device; recv_counts: int32 [world * per] out (or null).
LC_EXPORT int lc_p2p_exchange_counted(void* comm, const void* src, void* out, long long row_bytes, int cap,
                                      const void* counts, int per, void* recv_counts, hipStream_t stream) {
  CommState* c = static_cast<CommState*>(comm);
  if (cap <= 0) return 0;
  if (row_bytes <= 0 || row_bytes % 16 || per < 1 || static_cast<size_t>(c->world) * per * 4 > kCountHeader) return -1;
  if (kCountHeader + static_cast<size_t>(c->world) * cap * row_bytes > c->reg.stage_bytes) return -2;
  CountedArgs a{static_cast<const char*>(src), static_cast<char*>(out), static_cast<const int32_t*>(counts),
                static_cast<int32_t*>(recv_counts), static_cast<uint32_t>(row_bytes), static_cast<uint32_t>(cap), per,
                c->rank, c->timeout};
  const long long chunks = static_cast<long long>(cap) * (row_bytes / 16);
  const int blocks = static_cast<int>(std::max<long long>(1, std::min<long long>(c->grid_cap, (chunks + 1023) / 1024)));
  const dim3 grid(blocks), block(kThreads);
#define LC_LAUNCHC(W_, unused) p2p_exchange_counted_kernel<W_><<<grid, block, 0, stream>>>(c->peers, c->reg, a)
  LC_P2P_BY_WORLD(LC_LAUNCHC, 0)
#undef LC_LAUNCHC
  LC_RETURN_LAST_ERROR();
}
