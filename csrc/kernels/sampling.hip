// Token sampling and log-prob scoring over fp32 logits [R, V] (V up to ~152k).
//
//   lc_sample         temperature -> top-k -> top-p -> categorical draw; returns the token
//                     and its log-probability under the distribution it was drawn from.
//                     temperature == 0 selects the arg-max (log-prob under softmax(logits)).
//                     Top-k / top-p use threshold bisection (count / mass above a value)
//                     instead of sorting the vocabulary: a pass is one coalesced sweep.
//   lc_logprob_gather log-softmax over the vocabulary, gathered at given target ids: the
//                     constrained decision used for verdicts ("Good" vs "NeedsRefinement",
//                     the two words the reference's parser accepts,
//                     /root/reference/src/main.rs:144-148) and multiple-choice answers.
//
// One 1024-thread block per row. Randomness is counter-based (seed, step) so a chain's
// tokens do not depend on which other sequences share its batch.
#include "common.h"

namespace {

constexpr int kThreads = 1024;

struct ArgMax {
  float v;
  int i;
};

LC_DEVICE ArgMax argmax_combine(ArgMax a, ArgMax b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

LC_DEVICE ArgMax block_argmax(ArgMax x, float* sv, int* si) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ArgMax o{__shfl_xor(x.v, off, 64), __shfl_xor(x.i, off, 64)};
    x = argmax_combine(x, o);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) {
    sv[wid] = x.v;
    si[wid] = x.i;
  }
  __syncthreads();
  ArgMax r{sv[0], si[0]};
  for (int i = 1; i < nw; ++i) r = argmax_combine(r, ArgMax{sv[i], si[i]});
  return r;
}

// mass and count of entries with z >= thr, where z = logit * inv_t and weights exp(z - zmax)
LC_DEVICE void mass_above(const float* __restrict__ row, int V, float inv_t, float zmax, float thr,
                          float* red, float* mass, float* count) {
  float m = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float z = row[i] * inv_t;
    if (z >= thr) {
      m += __expf(z - zmax);
      c += 1.f;
    }
  }
  *mass = block_sum(m, red);
  *count = block_sum(c, red);
}

__global__ __launch_bounds__(kThreads) void sample_kernel(
    const float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ row_index,
    const float* __restrict__ temperature, const float* __restrict__ top_p, const int32_t* __restrict__ top_k,
    const uint64_t* __restrict__ seeds, const int32_t* __restrict__ steps, int32_t* __restrict__ out_token,
    float* __restrict__ out_logprob) {
  __shared__ float red[16];
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ float partial[kThreads];
  __shared__ int chosen;
  const int r = blockIdx.x;
  const float* row = logits + static_cast<size_t>(row_index ? row_index[r] : r) * ld;
  const float T = temperature[r];
  const bool greedy = !(T > 0.f);
  const float inv_t = greedy ? 1.f : 1.f / T;

  // pass 1: arg-max of z
  ArgMax am{-INFINITY, 0x7fffffff};
  for (int i = threadIdx.x; i < V; i += blockDim.x) am = argmax_combine(am, ArgMax{row[i] * inv_t, i});
  am = block_argmax(am, sv, si);
  const float zmax = am.v;
  // pass 2: partition function
  float total, cnt_all;
  mass_above(row, V, inv_t, zmax, -INFINITY, red, &total, &cnt_all);
  if (greedy) {
    if (threadIdx.x == 0) {
      out_token[r] = am.i;
      out_logprob[r] = -__logf(total);  // z_max - zmax - log(sum)
    }
    return;
  }

  float thr = -INFINITY;  // kept set = { z >= thr }
  float kept_mass = total;
  const int k = top_k ? top_k[r] : 0;
  if (k > 0 && k < V) {
    // largest threshold that still keeps >= k entries
    float lo = zmax - 60.f, hi = zmax;  // count(lo) >= k assumed (entries 60 nats below max carry no mass)
    float m_lo, c_lo;
    mass_above(row, V, inv_t, zmax, lo, red, &m_lo, &c_lo);
    if (c_lo >= static_cast<float>(k)) {
      for (int it = 0; it < 30; ++it) {
        const float mid = 0.5f * (lo + hi);
        float m, c;
        mass_above(row, V, inv_t, zmax, mid, red, &m, &c);
        if (c >= static_cast<float>(k)) {
          lo = mid;
          m_lo = m;
          if (c == static_cast<float>(k)) break;
        } else {
          hi = mid;
        }
      }
      thr = lo;
      kept_mass = m_lo;
    }
  }
  const float p = top_p ? top_p[r] : 1.f;
  if (p < 1.f) {
    // largest threshold whose kept mass is still >= p * (mass after top-k)
    const float target = p * kept_mass;
    float lo = fmaxf(thr, zmax - 60.f), hi = zmax;
    float m_lo, c_lo;
    mass_above(row, V, inv_t, zmax, lo, red, &m_lo, &c_lo);
    for (int it = 0; it < 30; ++it) {
      const float mid = 0.5f * (lo + hi);
      float m, c;
      mass_above(row, V, inv_t, zmax, mid, red, &m, &c);
      if (m >= target) {
        lo = mid;
        m_lo = m;
        c_lo = c;
      } else {
        hi = mid;
      }
    }
    thr = lo;
    kept_mass = m_lo;
  }

  // draw: u * kept_mass located by an index-ordered scan (contiguous chunk per thread)
  const float u = rng_uniform(seeds[r], static_cast<uint32_t>(steps ? steps[r] : 0), 0);
  const int chunk = (V + blockDim.x - 1) / blockDim.x;
  const int b0 = threadIdx.x * chunk, b1 = min(V, b0 + chunk);
  float mine = 0.f;
  for (int i = b0; i < b1; ++i) {
    const float z = row[i] * inv_t;
    if (z >= thr) mine += __expf(z - zmax);
  }
  partial[threadIdx.x] = mine;
  if (threadIdx.x == 0) chosen = -1;
  __syncthreads();
  if (threadIdx.x < 64) {
    // wave 0 scans the 1024 partials: 16 per lane
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += partial[threadIdx.x * 16 + j];
    float incl = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(incl, off, 64);
      if (static_cast<int>(threadIdx.x) >= off) incl += o;
    }
    const float grand = __shfl(incl, 63, 64);
    const float target = u * grand;
    float before = incl - acc;
    // the lane whose span contains the target
    const bool here = (target >= before) && (target < incl || threadIdx.x == 63);
    const unsigned long long ball = __ballot(here);
    const int owner = ball ? __ffsll(static_cast<long long>(ball)) - 1 : 63;
    if (static_cast<int>(threadIdx.x) == owner) {
      float run = before;
      int pick = threadIdx.x * 16 + 15;
      for (int j = 0; j < 16; ++j) {
        const float nxt = run + partial[threadIdx.x * 16 + j];
        if (target < nxt) {
          pick = threadIdx.x * 16 + j;
          break;
        }
        run = nxt;
      }
      // skip empty trailing chunks (when the target equals the grand total)
      while (pick > 0 && partial[pick] == 0.f) --pick;
      chosen = pick;
      float base = 0.f;
      for (int j = 0; j < pick; ++j) base += partial[j];
      partial[0] = fmaxf(target - base, 0.f);  // residual inside the chosen chunk
      partial[1] = grand;
    }
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) == chosen) {
    const float resid = partial[0], grand = partial[1];
    float run = 0.f;
    int tok = -1, last_kept = -1;
    float ztok = 0.f;
    for (int i = b0; i < b1; ++i) {
      const float z = row[i] * inv_t;
      if (z >= thr) {
        last_kept = i;
        run += __expf(z - zmax);
        if (resid < run) {
          tok = i;
          ztok = z;
          break;
        }
      }
    }
    if (tok < 0) {
      tok = last_kept >= 0 ? last_kept : am.i;
      ztok = row[tok] * inv_t;
    }
    out_token[r] = tok;
    out_logprob[r] = (ztok - zmax) - __logf(grand);
  }
}

// out[r, c] = log_softmax(logits[r])[ids[r*id_stride + c]]
__global__ __launch_bounds__(kThreads) void logprob_gather_kernel(
    const float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ row_index,
    const int32_t* __restrict__ ids, int id_stride, int C, float* __restrict__ out) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* row = logits + static_cast<size_t>(row_index ? row_index[r] : r) * ld;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, row[i]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) s += __expf(row[i] - mx);
  s = block_sum(s, red);
  const float logz = mx + __logf(s);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    int id = ids[static_cast<size_t>(r) * id_stride + c];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[static_cast<size_t>(r) * C + c] = row[id] - logz;
  }
}

}  // namespace

LC_EXPORT int lc_sample(const void* logits, int V, int ld, const void* row_index, const void* temperature,
                        const void* top_p, const void* top_k, const void* seeds, const void* steps,
                        void* out_token, void* out_logprob, int R, hipStream_t stream) {
  if (R <= 0) return 0;
  sample_kernel<<<dim3(R), dim3(kThreads), 0, stream>>>(
      static_cast<const float*>(logits), V, ld, static_cast<const int32_t*>(row_index),
      static_cast<const float*>(temperature), static_cast<const float*>(top_p),
      static_cast<const int32_t*>(top_k), static_cast<const uint64_t*>(seeds),
      static_cast<const int32_t*>(steps), static_cast<int32_t*>(out_token), static_cast<float*>(out_logprob));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_logprob_gather(const void* logits, int V, int ld, const void* row_index, const void* ids,
                                int id_stride, int C, void* out, int R, hipStream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  logprob_gather_kernel<<<dim3(R), dim3(kThreads), 0, stream>>>(
      static_cast<const float*>(logits), V, ld, static_cast<const int32_t*>(row_index),
      static_cast<const int32_t*>(ids), id_stride, C, static_cast<float*>(out));
  LC_RETURN_LAST_ERROR();
}
