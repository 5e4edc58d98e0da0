// Token sampling and log-prob scoring over fp32 logits [R, V] (V up to ~152k).
//
//   lc_sample         temperature -> top-k -> top-p -> categorical draw; returns the token
//                     and its log-probability under the distribution it was drawn from.
//                     temperature == 0 selects the arg-max (log-prob under softmax(logits)).
//                     Top-k / top-p never sort the vocabulary: the kept set is {z >= thr}
//                     and thr is found by a 12-way threshold search (per-thread LDS
//                     histograms), each sweep one coalesced 16-byte-per-lane pass over the row.
//                     The top-p cut is DEFINED by that search: the largest threshold of the
//                     grid refined kSearchPassesP times inside [max - 30, max] whose kept mass
//                     still reaches top_p (at most 1.4e-3 below the exact sort-based cut; the
//                     host oracle ops/reference.py::top_p_threshold implements the same rule).
//   lc_logprob_gather log-softmax over the vocabulary, gathered at given target ids: the
//                     constrained decision used for verdicts ("Good" vs "NeedsRefinement",
//                     the two words the reference's parser accepts,
//                     /root/reference/src/main.rs:144-148) and multiple-choice answers.
//
// One 1024-thread block per row. Randomness is counter-based (seed, step) so a chain's
// tokens do not depend on which other sequences share its batch. The inverse-CDF draw
// walks the vocabulary in index order (wave-contiguous segments + wave prefix scans).
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kBins = 12;          // threshold-search fan-out per sweep (12 x 1024 floats of LDS)
constexpr int kGroups = 8;         // groups per wave segment in the draw (see the draw phase)
constexpr int kSearchPasses = 6;    // top-k: 30 / 12^6 ~ 1e-5 resolution in logit/T units (separates the k-th entry)
constexpr int kSearchPassesP = 4;   // top-p: 30 / 12^4 ~ 1.4e-3; entries that close to the cut carry ~1e-6 of the mass each

struct ArgMax {
  float v;
  int i;
};

LC_DEVICE ArgMax argmax_combine(ArgMax a, ArgMax b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

// Visits every element of the row once as (value, index); 16-byte loads when the row is
// 16-byte aligned and V % 4 == 0, scalar loads otherwise.
template <typename F>
LC_DEVICE void for_each_strided(const float* __restrict__ row, int V, bool vec, F&& f) {
  if (vec) {
    // Eight independent 16-byte loads are issued before any of them is consumed: with one
    // block per row a CU has few waves, and a load-use-load chain leaves it latency-bound at
    // a fraction of what the L2 can deliver.
    constexpr int U = 8, STEP = kThreads * 4;
    int i = threadIdx.x * 4;
    for (; i + (U - 1) * STEP < V; i += U * STEP) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(row + i + u * STEP);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f(v[u][0], i + u * STEP);
        f(v[u][1], i + u * STEP + 1);
        f(v[u][2], i + u * STEP + 2);
        f(v[u][3], i + u * STEP + 3);
      }
    }
    for (; i < V; i += STEP) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
      f(v[0], i);
      f(v[1], i + 1);
      f(v[2], i + 2);
      f(v[3], i + 3);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += kThreads) f(row[i], i);
  }
}

__global__ __launch_bounds__(kThreads) void sample_kernel(
    const float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ row_index,
    const float* __restrict__ temperature, const float* __restrict__ top_p, const int32_t* __restrict__ top_k,
    const uint64_t* __restrict__ seeds, const int32_t* __restrict__ steps, int32_t* __restrict__ out_token,
    float* __restrict__ out_logprob) {
  __shared__ float red[16];
  __shared__ float sv[kWaves];
  __shared__ int si[kWaves];
  __shared__ float multi[kWaves][kBins];
  extern __shared__ float hist[];  // [kBins][kThreads]
  __shared__ float seg_sum[kWaves * kGroups];
  __shared__ float bc[2];
  __shared__ int bci[1];
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* row = logits + static_cast<size_t>(row_index ? row_index[r] : r) * ld;
  const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  const float T = temperature[r];
  const bool greedy = !(T > 0.f);
  const float inv_t = greedy ? 1.f : 1.f / T;

  // ---- pass 1: arg-max of z = logit / T and, in the same sweep, the partition function ----
  // Every thread keeps a running maximum m and the sum of exp(z - m) over what it has seen
  // (rescaled whenever m grows); the block then rebases the per-thread sums on the global max.
  ArgMax am{-INFINITY, 0x7fffffff};
  float run_m = -INFINITY, run_s = 0.f;
  for_each_strided(row, V, vec, [&](float x, int i) {
    const float z = x * inv_t;
    am = argmax_combine(am, ArgMax{z, i});
    if (z > run_m) {
      run_s = run_s * __expf(run_m - z) + 1.f;     // exp(-inf) = 0 on the first element
      run_m = z;
    } else if (z > -INFINITY) {                    // -inf pads (ragged vocabulary shards) carry no mass
      run_s += __expf(z - run_m);
    }
  });
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    am = argmax_combine(am, ArgMax{__shfl_xor(am.v, off, 64), __shfl_xor(am.i, off, 64)});
  if (lane == 0) {
    sv[wid] = am.v;
    si[wid] = am.i;
  }
  __syncthreads();
  am = ArgMax{sv[0], si[0]};
  for (int i = 1; i < kWaves; ++i) am = argmax_combine(am, ArgMax{sv[i], si[i]});
  const float zmax = am.v;
  const float total = block_sum(run_m > -INFINITY ? run_s * __expf(run_m - zmax) : 0.f, red);
  // A row without a single finite comparable entry (all NaN: a poisoned forward pass) leaves the
  // arg-max index at its sentinel. It must never be used as an index or handed to the caller.
  if (am.i < 0 || am.i >= V) am.i = 0;
  if (greedy) {
    if (threadIdx.x == 0) {
      out_token[r] = am.i;
      out_logprob[r] = -__logf(total);
    }
    return;
  }

  // ---- threshold search: kept set = { z >= thr } ----
  float thr = -INFINITY;
  float kept_mass = total;
  const int k = top_k ? top_k[r] : 0;
  const float p = top_p ? top_p[r] : 1.f;
  const bool use_k = k > 0 && k < V, use_p = p < 1.f;
  for (int phase = 0; phase < 2; ++phase) {
    if (phase == 0 ? !use_k : !use_p) continue;
    // Entries more than 30 below the max carry < 1e-13 of the mass each: start there.
    float lo = fmaxf(thr, zmax - 30.f), hi = zmax;
    const float target = phase == 0 ? static_cast<float>(k) : p * kept_mass;
    bool found = false;
    // Mass (or count) of the entries at or above `hi`. While hi is still the row maximum those
    // entries are binned with the rest (top bin); once the search has moved to an interior
    // interval they are known from the sweep that excluded them and are skipped: a sweep after
    // the first touches its histogram only for the ~1/12 of the candidates still in range.
    float above = 0.f;
    bool hi_is_max = true;
    const int passes = phase == 0 ? kSearchPasses : kSearchPassesP;
    for (int it = 0; it < passes; ++it) {
      // Per-thread histogram over kBins equal sub-intervals of [lo, hi): thread t owns column t
      // of hist[kBins][kThreads] (conflict-free, no atomics), so suffix sums give the mass (or
      // count) above each threshold t_j = lo + w*j.
      const float w = (hi - lo) * (1.f / kBins);
      const float inv_w = w > 0.f ? 1.f / w : 0.f;
#pragma unroll
      for (int j = 0; j < kBins; ++j) hist[j * kThreads + threadIdx.x] = 0.f;
      for_each_strided(row, V, vec, [&](float x, int) {
        const float z = x * inv_t;
        if (z >= lo && (hi_is_max || z < hi)) {
          int b = static_cast<int>((z - lo) * inv_w);
          b = b < kBins - 1 ? b : kBins - 1;
          hist[b * kThreads + threadIdx.x] += phase == 0 ? 1.f : __expf(z - zmax);
        }
      });
      float bins[kBins];
#pragma unroll
      for (int j = 0; j < kBins; ++j) bins[j] = wave_sum(hist[j * kThreads + threadIdx.x]);
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kBins; ++j) multi[wid][j] = bins[j];
      }
      __syncthreads();
      // every thread walks the thresholds from the top; the result is block-uniform
      int best = -1;
      float cum = above, best_v = 0.f, above_best = above;
      for (int j = kBins - 1; j >= 0 && best < 0; --j) {
        const float before = cum;
        for (int wv = 0; wv < kWaves; ++wv) cum += multi[wv][j];
        if (cum >= target) {
          best = j;
          best_v = cum;
          above_best = before;
        }
      }
      if (best < 0) break;  // even {z >= lo} misses the target: keep the previous edge
      const float nlo = lo + w * static_cast<float>(best);
      if (best < kBins - 1) {
        hi = lo + w * static_cast<float>(best + 1);
        hi_is_max = false;
        above = above_best;
      }
      lo = nlo;
      thr = lo;
      found = true;
      if (phase == 0 && best_v == target) break;
    }
    if (found && phase == 0 && use_p) {
      // top-p is relative to the mass that survived top-k
      float m = 0.f;
      for_each_strided(row, V, vec, [&](float x, int) {
        const float z = x * inv_t;
        m += z >= thr ? __expf(z - zmax) : 0.f;
      });
      kept_mass = block_sum(m, red);
    }
  }

  // ---- draw: locate u * kept_mass in the index-ordered CDF of the kept set ----
  const float u = rng_uniform(seeds[r], static_cast<uint32_t>(steps ? steps[r] : 0), 0);
  // Each wave owns a contiguous segment of the vocabulary, cut into kGroups groups; the sweep below
  // leaves one mass per (wave, group), thread 0 picks the group that holds the target and only
  // that group (1/128 of the row) is walked again, by its wave, with prefix scans.
  const int gsz = (((V + kWaves - 1) / kWaves + kGroups - 1) / kGroups + 255) & ~255;   // multiple of 256
  const int seg = gsz * kGroups;
  const int s0 = wid * seg, s1 = min(V, s0 + seg);
  auto kept_w = [&](int i) -> float {
    if (i >= s1) return 0.f;
    const float z = row[i] * inv_t;
    return z >= thr ? __expf(z - zmax) : 0.f;
  };
  // weights of elements i .. i+3 (i is a multiple of 4): one 16-byte load when the row allows it
  auto kept_w4 = [&](int i) -> f32x4 {
    if (vec && i < s1) {   // V % 4 == 0 and segment bounds are multiples of 4: the quad is whole
      const f32x4 x = *reinterpret_cast<const f32x4*>(row + i);
      f32x4 w;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float z = x[q] * inv_t;
        w[q] = z >= thr ? __expf(z - zmax) : 0.f;
      }
      return w;
    }
    return f32x4{kept_w(i), kept_w(i + 1), kept_w(i + 2), kept_w(i + 3)};
  };
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    float mine = 0.f;
    const int g1 = min(s1, s0 + (g + 1) * gsz);
    for (int i = s0 + g * gsz + lane * 4; i < g1; i += 256) {
      const f32x4 w4 = kept_w4(i);
      mine += w4[0] + w4[1] + w4[2] + w4[3];
    }
    mine = wave_sum(mine);
    if (lane == 0) seg_sum[wid * kGroups + g] = mine;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float grand = 0.f;
    for (int w2 = 0; w2 < kWaves * kGroups; ++w2) grand += seg_sum[w2];
    const float target = u * grand;
    float run = 0.f;
    int owner = 0;
    for (int w2 = 0; w2 < kWaves * kGroups; ++w2) {
      if (seg_sum[w2] > 0.f) {
        owner = w2;  // last non-empty group so far: fallback when target == grand
        if (target < run + seg_sum[w2]) break;
      }
      run += seg_sum[w2];
    }
    run = 0.f;
    for (int w2 = 0; w2 < owner; ++w2) run += seg_sum[w2];
    bc[0] = fmaxf(target - run, 0.f);
    bc[1] = grand;
    bci[0] = owner;
  }
  __syncthreads();
  if (wid == bci[0] / kGroups) {
    const float resid = bc[0], grand = bc[1];
    float run = 0.f;
    int tok = -1, last_kept = -1;
    const int gb = s0 + (bci[0] % kGroups) * gsz, ge = min(s1, gb + gsz);
    for (int base = gb; base < ge && tok < 0; base += 256) {
      const int i = base + lane * 4;
      const f32x4 w4 = kept_w4(i);
      const float w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];
      const float mysum = w0 + w1 + w2 + w3;
      float incl = mysum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const float before = run + incl - mysum;
      const bool hit = mysum > 0.f && resid < run + incl;
      const unsigned long long hm = __ballot(hit);
      const unsigned long long km = __ballot(mysum > 0.f);
      if (km) {
        // last kept element seen so far (fallback when rounding never crosses resid)
        const int hl = 63 - __clzll(static_cast<long long>(km));
        int lk = -1;
        if (lane == hl) {
          const float ws[4] = {w0, w1, w2, w3};
          for (int q = 0; q < 4; ++q)
            if (ws[q] > 0.f) lk = i + q;
        }
        last_kept = __shfl(lk, hl, 64);
      }
      if (hm) {
        const int first = __ffsll(static_cast<long long>(hm)) - 1;
        int mytok = -1;
        if (lane == first) {
          float c = before;
          const float ws[4] = {w0, w1, w2, w3};
          for (int q = 0; q < 4; ++q) {
            if (ws[q] > 0.f) {
              mytok = i + q;
              c += ws[q];
              if (resid < c) break;
            }
          }
        }
        tok = __shfl(mytok, first, 64);
      }
      run += __shfl(incl, 63, 64);
    }
    if (tok < 0 || tok >= V) tok = (last_kept >= 0 && last_kept < V) ? last_kept : am.i;
    if (lane == 0) {
      out_token[r] = tok;
      out_logprob[r] = (row[tok] * inv_t - zmax) - __logf(grand);
    }
  }
}

// out[r, c] = log_softmax(logits[r])[ids[r*id_stride + c]]
__global__ __launch_bounds__(kThreads) void logprob_gather_kernel(
    const float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ row_index,
    const int32_t* __restrict__ ids, int id_stride, int C, float* __restrict__ out) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* row = logits + static_cast<size_t>(row_index ? row_index[r] : r) * ld;
  const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  // online max + rescaled sum in one pass
  float mx = -INFINITY, s = 0.f;
  for_each_strided(row, V, vec, [&](float x, int) {
    if (x > mx) {
      s = s * __expf(mx - x) + 1.f;
      mx = x;
    } else {
      s += __expf(x - mx);
    }
  });
  const float bmx = block_max(mx, red);
  s = block_sum(mx == -INFINITY ? 0.f : s * __expf(mx - bmx), red);
  const float logz = bmx + __logf(s);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    int id = ids[static_cast<size_t>(r) * id_stride + c];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[static_cast<size_t>(r) * C + c] = row[id] - logz;
  }
}

// Vocabulary-parallel form: this rank holds columns [vocab_lo, vocab_lo + V) of the logits.
// out[r, c] = logits[r][ids[r, c] - vocab_lo] when the id falls into this shard, else 0;
// out[r, C] = log-sum-exp of the shard's row; columns C + 1 .. Cp - 1 (padding that keeps a row
// a multiple of 16 bytes for the exchange) are zero. The ranks exchange these Cp numbers per row
// instead of the logits and logprob_merge_kernel finishes the log-softmax.
__global__ __launch_bounds__(kThreads) void logit_targets_shard_kernel(
    const float* __restrict__ logits, int V, int ld, const int32_t* __restrict__ row_index,
    const int32_t* __restrict__ ids, int id_stride, int C, int Cp, int vocab_lo, float* __restrict__ out) {
  __shared__ float red[16];
  const int r = blockIdx.x;
  const float* row = logits + static_cast<size_t>(row_index ? row_index[r] : r) * ld;
  const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  float mx = -INFINITY, s = 0.f;
  for_each_strided(row, V, vec, [&](float x, int) {
    if (x > mx) {
      s = s * __expf(mx - x) + 1.f;
      mx = x;
    } else {
      s += __expf(x - mx);
    }
  });
  const float bmx = block_max(mx, red);
  s = block_sum(mx == -INFINITY ? 0.f : s * __expf(mx - bmx), red);
  const float lse = bmx + __logf(s);
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    float v = 0.f;
    if (c < C) {
      const int id = ids[static_cast<size_t>(r) * id_stride + c] - vocab_lo;
      if (id >= 0 && id < V) v = row[id];
    } else if (c == C) {
      v = lse;
    }
    out[static_cast<size_t>(r) * Cp + c] = v;
  }
}

// parts [W, R, Cp] from logit_targets_shard_kernel of W ranks -> out[r, c] = sum_w parts[w, r, c]
// - logsumexp_w parts[w, r, C]. One thread per (row, target).
__global__ void logprob_merge_kernel(const float* __restrict__ parts, int W, int R, int C, int Cp,
                                     float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * C) return;
  const int r = i / C, c = i % C;
  float mx = -INFINITY;
  for (int w = 0; w < W; ++w) mx = fmaxf(mx, parts[(static_cast<size_t>(w) * R + r) * Cp + C]);
  float z = 0.f, v = 0.f;
  for (int w = 0; w < W; ++w) {
    const float* p = parts + (static_cast<size_t>(w) * R + r) * Cp;
    z += __expf(p[C] - mx);
    v += p[c];
  }
  out[i] = v - (mx + __logf(z));
}

}  // namespace

LC_EXPORT int lc_logit_targets_shard(const void* logits, int V, int ld, const void* row_index, const void* ids,
                                     int id_stride, int C, int Cp, int vocab_lo, void* out, int R,
                                     hipStream_t stream) {
  if (R <= 0) return 0;
  if (C <= 0 || Cp <= C) return -1;
  logit_targets_shard_kernel<<<dim3(R), dim3(kThreads), 0, stream>>>(
      static_cast<const float*>(logits), V, ld, static_cast<const int32_t*>(row_index),
      static_cast<const int32_t*>(ids), id_stride, C, Cp, vocab_lo, static_cast<float*>(out));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_logprob_merge(const void* parts, int W, int R, int C, int Cp, void* out, hipStream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  if (W < 1 || Cp <= C) return -1;
  logprob_merge_kernel<<<dim3((R * C + 255) / 256), dim3(256), 0, stream>>>(static_cast<const float*>(parts), W, R, C,
                                                                           Cp, static_cast<float*>(out));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_sample(const void* logits, int V, int ld, const void* row_index, const void* temperature,
                        const void* top_p, const void* top_k, const void* seeds, const void* steps,
                        void* out_token, void* out_logprob, int R, hipStream_t stream) {
  if (R <= 0) return 0;
  sample_kernel<<<dim3(R), dim3(kThreads), kBins * kThreads * sizeof(float), stream>>>(
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
