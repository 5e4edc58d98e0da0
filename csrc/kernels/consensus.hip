// Device-side consensus reducers (batched over requests).
//
//   majority vote  : votes [R, N] int32 (class id or -1) -> winner / count / valid /
//                    agreement / unanimous. One wave per request; lane c owns class c and
//                    counts its votes with a 64-bit ballot + popcount per 64 voters.
//   cosine medoid  : embeddings [n, d] fp32, n <= 64 -> Gram matrix E*E^T on the exact-fp32
//                    MFMA (v_mfma_f32_32x32x2_f32), cosine row sums, arg-max.
//   judge reduce   : scores [R, C, J] -> best mean candidate + "every judge agrees" flag.
//
// The reference's only reducer is "all votes Good" (/root/reference/src/main.rs:266, :323);
// its device analogue is the unanimous flag of the two-class majority vote.
#include "common.h"

namespace {

// tie: 0 = lowest class id, 1 = class of the earliest voter, 2 = larger weight sum (then lower id).
// weights [N] or null: with tie 0 / 1 the ranking is by weight sum, then count, then the tie
// rule (weighted plurality); with tie 2 by count, then weight sum. Same keys, same float
// summation order (voter order) as lc::majority_vote on the host.
__global__ void majority_vote_kernel(const int32_t* __restrict__ votes, const float* __restrict__ weights,
                                     int R, int N, int num_classes,
                                     int tie, int32_t* __restrict__ winner, int32_t* __restrict__ count,
                                     int32_t* __restrict__ valid, float* __restrict__ agreement,
                                     int32_t* __restrict__ unanimous) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const int32_t* row = votes + static_cast<size_t>(r) * N;
  int my_count = 0, my_first = 0x7fffffff, n_valid = 0;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    const int v = i < N ? row[i] : -1;
    const bool ok = v >= 0 && v < num_classes;
    n_valid += __popcll(__ballot(ok));
    for (int c = 0; c < num_classes; ++c) {
      const unsigned long long m = __ballot(ok && v == c);
      if (lane == c && m) {
        my_count += __popcll(m);
        my_first = min(my_first, base + __ffsll(static_cast<long long>(m)) - 1);
      }
    }
  }
  float my_w = static_cast<float>(my_count);
  if (weights != nullptr) {
    my_w = 0.f;
    if (lane < num_classes && my_count > 0)
      for (int i = 0; i < N; ++i)
        if (row[i] == lane) my_w += weights[i];
  }
  const bool weighted_primary = weights != nullptr && tie != 2;
  // arg-max over lanes (classes) on (key1, key2), then lower class or earlier voter
  int best_c = lane < num_classes && my_count > 0 ? lane : -1;
  int best_n = lane < num_classes ? my_count : 0;
  int best_f = my_first;
  float best_k1 = weighted_primary ? my_w : static_cast<float>(my_count);
  float best_k2 = weighted_primary ? static_cast<float>(my_count) : (tie == 2 ? my_w : 0.f);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int oc = __shfl_xor(best_c, off, 64), on = __shfl_xor(best_n, off, 64),
              of = __shfl_xor(best_f, off, 64);
    const float ok1 = __shfl_xor(best_k1, off, 64), ok2 = __shfl_xor(best_k2, off, 64);
    bool take = false;
    if (oc >= 0) {
      if (best_c < 0) take = true;
      else if (ok1 != best_k1) take = ok1 > best_k1;
      else if (ok2 != best_k2) take = ok2 > best_k2;
      else take = tie == 1 ? (of < best_f) : (oc < best_c);
    }
    if (take) {
      best_c = oc;
      best_n = on;
      best_f = of;
      best_k1 = ok1;
      best_k2 = ok2;
    }
  }
  if (lane == 0) {
    winner[r] = best_c;
    count[r] = best_c >= 0 ? best_n : 0;
    valid[r] = n_valid;
    agreement[r] = n_valid > 0 ? static_cast<float>(best_n) / static_cast<float>(n_valid) : 0.f;
    unanimous[r] = (n_valid > 0 && best_n == n_valid) ? 1 : 0;
  }
}

// One block (4 waves) per request. Wave w accumulates the Gram tile over the slice
// k = w, w+4, ... (pairs of columns) and the four partial tiles are summed through LDS.
// n <= 32 uses one 32x32 tile; n <= 64 uses 2x2 tiles.
__global__ __launch_bounds__(256) void cosine_medoid_kernel(const float* __restrict__ E, int n, int d,
                                                            int32_t* __restrict__ out_index,
                                                            float* __restrict__ out_score,
                                                            float* __restrict__ out_mean, int stride_req) {
  __shared__ float gram[64][65];
  __shared__ float rowsum[64];
  const float* e = E + static_cast<size_t>(blockIdx.x) * stride_req;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nt = n > 32 ? 2 : 1;
  for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) (&gram[0][0])[i] = 0.f;
  __syncthreads();
  for (int ti = 0; ti < nt; ++ti)
    for (int tj = ti; tj < nt; ++tj) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int ri = ti * 32 + r, rj = tj * 32 + r;
      for (int k0 = wave * 2; k0 < d; k0 += 8) {
        const int k = k0 + h;
        const float a = (ri < n && k < d) ? e[static_cast<size_t>(ri) * d + k] : 0.f;
        const float b = (rj < n && k < d) ? e[static_cast<size_t>(rj) * d + k] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      // C layout: col = lane&31 (j), row = (reg&3) + 8*(reg>>2) + 4*h (i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int i = ti * 32 + (g & 3) + 8 * (g >> 2) + 4 * h, j = tj * 32 + r;
        atomicAdd(&gram[i][j], acc[g]);
      }
    }
  __syncthreads();
  if (nt == 2)
    for (int idx = threadIdx.x; idx < 32 * 32; idx += blockDim.x) {
      const int i = 32 + idx / 32, j = idx % 32;  // mirror the lower-left tile
      gram[i][j] = gram[j][i];
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int i = threadIdx.x;
    float s = 0.f;
    if (i < n) {
      const float ni = gram[i][i];
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const float den = ni * gram[j][j];
        s += den > 0.f ? gram[i][j] * rsqrtf(den) : 0.f;
      }
    }
    rowsum[i] = s;
    // arg-max (first maximum) and total
    float bv = i < n ? s : -INFINITY;
    int bi = i < n ? i : 0x7fffffff;
    float tot = i < n ? s : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
      tot += __shfl_xor(tot, off, 64);
    }
    if (i == 0) {
      out_index[blockIdx.x] = n > 0 ? bi : -1;
      out_score[blockIdx.x] = n > 1 ? bv / static_cast<float>(n - 1) : 1.f;
      out_mean[blockIdx.x] = n > 1 ? tot / static_cast<float>(n * (n - 1)) : 1.f;
    }
  }
}

__global__ void judge_reduce_kernel(const float* __restrict__ scores, int R, int C, int J,
                                    int32_t* __restrict__ winner, float* __restrict__ score,
                                    int32_t* __restrict__ unanimous) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* s = scores + static_cast<size_t>(r) * C * J;
  // lane c: mean score of candidate c
  float mean = -INFINITY;
  if (lane < C) {
    float t = 0.f;
    for (int j = 0; j < J; ++j) t += s[lane * J + j];
    mean = t / static_cast<float>(J);
  }
  float bv = mean;
  int bi = lane < C ? lane : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  // lane j (strided): does judge j rank the winner first?
  int agree = 1;
  for (int j = lane; j < J; j += 64) {
    int top = 0;
    for (int c = 1; c < C; ++c)
      if (s[c * J + j] > s[top * J + j]) top = c;
    if (top != bi) agree = 0;
  }
  const unsigned long long dis = __ballot(agree == 0);
  if (lane == 0) {
    winner[r] = bi;
    score[r] = bv;
    unanimous[r] = dis == 0 ? 1 : 0;
  }
}

// Answer extraction on token ids: for each generated sequence find the LAST occurrence of
// the marker token string (e.g. the tokens of "Answer:") and classify the first token among
// the next `window` tokens that is one of the C choice tokens; -1 if there is no marker or no
// choice token follows it. With an empty marker the whole sequence is the window and the
// LAST choice token wins. One wave per row; lanes test positions in parallel and the wave
// keeps the largest matching position.
__global__ void answer_extract_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                      int R, int ld, const int32_t* __restrict__ marker, int P,
                                      const int32_t* __restrict__ choice_ids, int C, int window,
                                      int32_t* __restrict__ cls) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const int32_t* row = ids + static_cast<size_t>(r) * ld;
  const int n = min(lens[r], ld);
  auto class_of = [&](int tok) {
    for (int c = 0; c < C; ++c)
      if (choice_ids[c] == tok) return c;
    return -1;
  };
  if (P == 0) {
    int best = -1;  // position of the last choice token
    for (int p = lane; p < n; p += 64)
      if (class_of(row[p]) >= 0) best = p;
    best = wave_max_int(best);
    if (lane == 0) cls[r] = best >= 0 ? class_of(row[best]) : -1;
    return;
  }
  int last = -1;
  for (int p = lane; p + P <= n; p += 64) {
    bool ok = true;
    for (int j = 0; j < P && ok; ++j) ok = row[p + j] == marker[j];
    if (ok) last = p;
  }
  last = wave_max_int(last);
  if (lane == 0) {
    int out = -1;
    if (last >= 0) {
      const int lo = last + P, hi = min(n, lo + window);
      for (int p = lo; p < hi && out < 0; ++p) out = class_of(row[p]);
    }
    cls[r] = out;
  }
}

}  // namespace

LC_EXPORT int lc_answer_extract(const void* ids, const void* lens, int R, int ld, const void* marker, int P,
                                const void* choice_ids, int C, int window, void* cls, hipStream_t stream) {
  if (R <= 0) return 0;
  if (C < 1 || P < 0 || ld < 1) return -1;
  answer_extract_kernel<<<dim3((R + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const int32_t*>(ids), static_cast<const int32_t*>(lens), R, ld,
      static_cast<const int32_t*>(marker), P, static_cast<const int32_t*>(choice_ids), C, window,
      static_cast<int32_t*>(cls));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_majority_vote(const void* votes, int R, int N, int num_classes, int tie, void* winner,
                               void* count, void* valid, void* agreement, void* unanimous,
                               const void* weights, hipStream_t stream) {
  if (R <= 0) return 0;
  if (num_classes < 1 || num_classes > 64 || tie < 0 || tie > 2) return -1;
  majority_vote_kernel<<<dim3((R + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const int32_t*>(votes), static_cast<const float*>(weights), R, N, num_classes, tie,
      static_cast<int32_t*>(winner),
      static_cast<int32_t*>(count), static_cast<int32_t*>(valid), static_cast<float*>(agreement),
      static_cast<int32_t*>(unanimous));
  LC_RETURN_LAST_ERROR();
}

// E holds n_req matrices of [n, d] (stride_req floats apart).
LC_EXPORT int lc_cosine_medoid(const void* E, int n_req, int n, int d, long long stride_req, void* out_index,
                               void* out_score, void* out_mean, hipStream_t stream) {
  if (n_req <= 0) return 0;
  if (n < 1 || n > 64) return -1;
  cosine_medoid_kernel<<<dim3(n_req), dim3(256), 0, stream>>>(
      static_cast<const float*>(E), n, d, static_cast<int32_t*>(out_index), static_cast<float*>(out_score),
      static_cast<float*>(out_mean), static_cast<int>(stride_req));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_judge_reduce(const void* scores, int R, int C, int J, void* winner, void* score,
                              void* unanimous, hipStream_t stream) {
  if (R <= 0) return 0;
  if (C < 1 || C > 64) return -1;
  judge_reduce_kernel<<<dim3((R + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const float*>(scores), R, C, J, static_cast<int32_t*>(winner), static_cast<float*>(score),
      static_cast<int32_t*>(unanimous));
  LC_RETURN_LAST_ERROR();
}
