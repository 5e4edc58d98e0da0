// Mixture-of-experts routing for Mixtral-style layers: router logits + softmax + top-k,
// permutation of token rows into expert order (histogram + prefix sum in LDS), and the
// weighted combine of expert outputs. The expert FFN itself is lc_gemm_grouped (gemm.hip),
// which gathers rows through the permutation, so no permuted activation copy is written.
// No counterpart in the reference (remote model, /root/reference/src/main.rs:82-86).
#include "common.h"

namespace {

constexpr int kMaxExperts = 16;
constexpr int kMaxTopK = 4;

// Softmax over the experts' logits, top-k, renormalise; run by one thread per token. `logit`
// may live in LDS or registers; loops have compile-time bounds so nothing is indexed dynamically
// in registers (that would go through scratch memory).
LC_DEVICE void route_token(const float* logit, int E, int topk, int32_t* __restrict__ ids_out,
                           float* __restrict__ w_out, float* __restrict__ logits_out) {
  float lg[kMaxExperts];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) {
    // HF Mixtral rounds router logits to the activation dtype before the softmax.
    lg[e] = e < E ? bf16_to_f32(f32_to_bf16(logit[e])) : -INFINITY;
    if (e < E && logits_out) logits_out[e] = lg[e];
    mx = fmaxf(mx, lg[e]);
  }
  float p[kMaxExperts], sum = 0.f;
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) {
    p[e] = e < E ? __expf(lg[e] - mx) : -1.f;      // -1: never chosen
    sum += e < E ? p[e] : 0.f;
  }
  float w[kMaxTopK], wsum = 0.f;
  int ids[kMaxTopK];
  unsigned used = 0;
#pragma unroll
  for (int k = 0; k < kMaxTopK; ++k) {
    if (k < topk) {
      int best = 0;
      float bp = -2.f;
#pragma unroll
      for (int e = 0; e < kMaxExperts; ++e) {
        const bool free_e = !((used >> e) & 1u);
        if (free_e && p[e] > bp) {       // strict: the lowest index wins a tie
          bp = p[e];
          best = e;
        }
      }
      used |= 1u << best;
      ids[k] = best;
      w[k] = bp / sum;
      wsum += w[k];
    }
  }
#pragma unroll
  for (int k = 0; k < kMaxTopK; ++k) {
    if (k < topk) {
      ids_out[k] = ids[k];
      w_out[k] = w[k] / wsum;
    }
  }
}

// logits[e] = x . Wr[e]. WAVES waves share a token: decode steps have a few dozen tokens, and
// with one wave per token the whole router is 16 workgroups walking 8 dependent rounds of loads
// (42 us at 64 tokens). Four waves per token issue all of a token's loads at once.
template <int WAVES>
__global__ __launch_bounds__(256) void moe_router_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wr,
                                                         int T, int d, int E, int topk,
                                                         int32_t* __restrict__ topk_ids, float* __restrict__ topk_w,
                                                         float* __restrict__ router_logits) {
  constexpr int TPB = 4 / WAVES;                      // tokens per 256-thread block
  __shared__ float part[4][kMaxExperts];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * TPB + wave / WAVES;
  const int sub = wave % WAVES;                       // this wave's share of the row
  float acc[kMaxExperts];
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) acc[e] = 0.f;
  if (t < T) {
    const bf16_t* xr = x + static_cast<size_t>(t) * d;
    for (int c = sub * 64 + lane; c < d / 8; c += 64 * WAVES) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + c * 8);
      float xf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = bf16_to_f32(static_cast<bf16_t>(xv[j]));
#pragma unroll
      for (int e = 0; e < kMaxExperts; ++e) {
        if (e < E) {
          const bf16x8 wv = *reinterpret_cast<const bf16x8*>(Wr + static_cast<size_t>(e) * d + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[e] += xf[j] * bf16_to_f32(static_cast<bf16_t>(wv[j]));
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e)
    if (e < E) acc[e] = wave_sum(acc[e]);
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < kMaxExperts; ++e) part[wave][e] = acc[e];
  }
  __syncthreads();
  if (lane == 0 && sub == 0 && t < T) {
    float logit[kMaxExperts];
#pragma unroll
    for (int e = 0; e < kMaxExperts; ++e) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) v += part[wave + q][e];
      logit[e] = v;
    }
    route_token(logit, E, topk, topk_ids + static_cast<size_t>(t) * topk, topk_w + static_cast<size_t>(t) * topk,
                router_logits ? router_logits + static_cast<size_t>(t) * E : nullptr);
  }
}

// Single block. Builds offsets[E+1] (rows per expert, prefix-summed), sorted_src[slot] =
// source token row, and slot_of[t*topk+k] = slot, for A = T*topk assignments.
// Deterministic placement: assignment a goes after all assignments a' < a of the same expert.
// Each thread owns a contiguous chunk of assignments; its per-expert counts are prefix-scanned
// over the block (wave scan + per-wave totals), which also yields the per-expert totals, so the
// ids are read from memory once and the block synchronises once.
__global__ __launch_bounds__(1024) void moe_permute_kernel(const int32_t* __restrict__ topk_ids, int A,
                                                           int topk, int E, int32_t* __restrict__ offsets,
                                                           int32_t* __restrict__ sorted_src,
                                                           int32_t* __restrict__ slot_of) {
  __shared__ int chunk_cnt[kMaxExperts][1024 / 64];   // inclusive per-wave totals
  constexpr int kKeep = 4;                            // ids kept in registers per thread
  const int per = (A + blockDim.x - 1) / blockDim.x;
  const int a0 = threadIdx.x * per, a1 = min(A, a0 + per);
  int kept[kKeep];
  int mine[kMaxExperts];
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) mine[e] = 0;
#pragma unroll
  for (int i = 0; i < kKeep; ++i) kept[i] = (i < per && a0 + i < a1) ? topk_ids[a0 + i] : -1;
#pragma unroll
  for (int i = 0; i < kKeep; ++i) {
#pragma unroll
    for (int q = 0; q < kMaxExperts; ++q) mine[q] += (q == kept[i]) ? 1 : 0;
  }
  for (int a = a0 + kKeep; a < a1; ++a) {
    const int e = topk_ids[a];
#pragma unroll
    for (int q = 0; q < kMaxExperts; ++q) mine[q] += (q == e) ? 1 : 0;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int excl[kMaxExperts];
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) {
    int incl = mine[e];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    excl[e] = incl - mine[e];
    if (lane == 63) chunk_cnt[e][wid] = incl;
  }
  __syncthreads();
  // excl[e] becomes this thread's first slot for expert e: expert base + rows of earlier waves +
  // rows of earlier lanes
  int run = 0;
#pragma unroll
  for (int e = 0; e < kMaxExperts; ++e) {
    int before = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
      const int c = chunk_cnt[e][w];
      before += w < wid ? c : 0;
      total += c;
    }
    excl[e] += run + before;
    if (threadIdx.x == 0 && e < E) offsets[e] = run;
    run += e < E ? total : 0;
  }
  if (threadIdx.x == 0) offsets[E] = run;
  auto place = [&](int a, int e) {
    int slot = 0;
#pragma unroll
    for (int q = 0; q < kMaxExperts; ++q)
      if (q == e) slot = excl[q]++;
    sorted_src[slot] = a / topk;
    slot_of[a] = slot;
  };
#pragma unroll
  for (int i = 0; i < kKeep; ++i)
    if (kept[i] >= 0) place(a0 + i, kept[i]);
  for (int a = a0 + kKeep; a < a1; ++a) place(a, topk_ids[a]);
}

// out[t,:] = sum_k w[t,k] * Y[slot_of[t*topk+k], :]   (fp32 accumulate, bf16 store). WAVES waves
// share a token row (decode steps: few tokens, all of a row's loads in flight at once).
template <int WAVES>
__global__ __launch_bounds__(256) void moe_combine_kernel(const bf16_t* __restrict__ Y,
                                                          const int32_t* __restrict__ slot_of,
                                                          const float* __restrict__ topk_w, bf16_t* __restrict__ out,
                                                          int T, int d, int topk) {
  constexpr int TPB = 4 / WAVES;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * TPB + wave / WAVES;
  const int sub = wave % WAVES;
  if (t >= T) return;
  int slots[kMaxTopK];
  float ws[kMaxTopK];
#pragma unroll
  for (int k = 0; k < kMaxTopK; ++k) {
    slots[k] = k < topk ? slot_of[static_cast<size_t>(t) * topk + k] : 0;
    ws[k] = k < topk ? topk_w[static_cast<size_t>(t) * topk + k] : 0.f;
  }
  for (int c = sub * 64 + lane; c < d / 8; c += 64 * WAVES) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kMaxTopK; ++k) {
      if (k < topk) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(Y + static_cast<size_t>(slots[k]) * d + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += ws[k] * bf16_to_f32(static_cast<bf16_t>(v[j]));
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = static_cast<short>(f32_to_bf16(acc[j]));
    *reinterpret_cast<bf16x8*>(out + static_cast<size_t>(t) * d + c * 8) = o;
  }
}

// ---- expert-parallel exchange (fixed capacity, no host round trip) -------------------------
// ep ranks, `per` experts each. A rank routes its own token slice, sorts the (token, k) pairs
// by expert (moe_permute_kernel) and ships the rows of experts [r*per, (r+1)*per) to rank r.
// Every (source, destination) pair owns `cap` rows of a [ep, cap, d] buffer, so the exchange
// is an all-to-all with equal splits whatever the routing is; how many of the rows are real
// travels beside them as per-expert counts.
//
// ep_pack: one wave per send row. Row (r, j), j < rows routed to rank r, is a copy of the
// token row of the j-th pair of that destination (pairs of one destination are contiguous
// in expert-sorted order). send_counts[r*per + e] = pairs for expert r*per + e.
// slot_back[a] = row of the RETURNED buffer (same [ep, cap, d] layout) that holds the expert
// output of pair a: the combine kernel reads it through this map.
__global__ void ep_pack_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ sorted_src,
                               const int32_t* __restrict__ offsets, const int32_t* __restrict__ slot_of,
                               int per, int ep, int cap, int A, int d, bf16_t* __restrict__ send,
                               int32_t* __restrict__ send_counts, int32_t* __restrict__ slot_back) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gtid < ep * per) send_counts[gtid] = offsets[gtid + 1] - offsets[gtid];
  if (gtid < A) {
    const int slot = slot_of[gtid];
    int r = 0;
    while (r + 1 < ep && slot >= offsets[(r + 1) * per]) ++r;
    slot_back[gtid] = r * cap + (slot - offsets[r * per]);
  }
  if (w >= ep * cap) return;
  const int r = w / cap, j = w - r * cap;
  const int start = offsets[r * per], cnt = offsets[(r + 1) * per] - start;
  if (j >= cnt) return;
  const bf16_t* src = x + static_cast<size_t>(sorted_src[start + j]) * d;
  bf16_t* dst = send + static_cast<size_t>(w) * d;
  for (int c = lane * 8; c < d; c += 64 * 8)
    *reinterpret_cast<bf16x8*>(dst + c) = *reinterpret_cast<const bf16x8*>(src + c);
}

// ep_recv_index: single block. recv_counts[s*per + e] = rows source s sent for local expert e;
// they sit in the receive buffer at rows s*cap + (sum of that source's counts for experts < e)
// onwards. Builds what the grouped GEMM wants: offsets_local[per+1] over the experts and, for
// every position of the expert-sorted order, the receive-buffer row it reads (row_index).
// Positions beyond the last real row point at row 0.
__global__ __launch_bounds__(256) void ep_recv_index_kernel(const int32_t* __restrict__ recv_counts, int ep,
                                                            int per, int cap, int32_t* __restrict__ row_index,
                                                            int32_t* __restrict__ offsets_local) {
  __shared__ int seg_base[kMaxExperts * 64];   // [e][s]: first sorted position of source s's rows for expert e
  __shared__ int src_cum[64 * (kMaxExperts + 1)];  // [s][e]: rows of source s for experts < e
  __shared__ int total;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < per; ++e) {
      offsets_local[e] = run;
      for (int s = 0; s < ep; ++s) {
        seg_base[e * ep + s] = run;
        run += recv_counts[s * per + e];
      }
    }
    offsets_local[per] = run;
    total = run;
    for (int s = 0; s < ep; ++s) {
      int c = 0;
      for (int e = 0; e < per; ++e) {
        src_cum[s * (per + 1) + e] = c;
        c += recv_counts[s * per + e];
      }
      src_cum[s * (per + 1) + per] = c;
    }
  }
  __syncthreads();
  const int rows = ep * cap;
  for (int p = total + threadIdx.x; p < rows; p += blockDim.x) row_index[p] = 0;
  for (int w = threadIdx.x; w < rows; w += blockDim.x) {
    const int s = w / cap, j = w - s * cap;
    const int* cum = src_cum + s * (per + 1);
    if (j >= cum[per]) continue;
    int e = 0;
    while (e + 1 < per && j >= cum[e + 1]) ++e;
    row_index[seg_base[e * ep + s] + (j - cum[e])] = w;
  }
}

// out[row_index[p], :] = y[p, :] for p < *total (one wave per row).
__global__ void scatter_rows_kernel(const bf16_t* __restrict__ y, const int32_t* __restrict__ row_index,
                                    const int32_t* __restrict__ total, int cap_rows, int d,
                                    bf16_t* __restrict__ out) {
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= cap_rows || p >= *total) return;
  const bf16_t* src = y + static_cast<size_t>(p) * d;
  bf16_t* dst = out + static_cast<size_t>(row_index[p]) * d;
  for (int c = lane * 8; c < d; c += 64 * 8)
    *reinterpret_cast<bf16x8*>(dst + c) = *reinterpret_cast<const bf16x8*>(src + c);
}

}  // namespace

LC_EXPORT int lc_ep_pack(const void* x, const void* sorted_src, const void* offsets, const void* slot_of, int per,
                         int ep, int cap, int A, int d, void* send, void* send_counts, void* slot_back,
                         hipStream_t stream) {
  if (A <= 0 || cap <= 0) return 0;
  if (d % 8 || ep < 1 || ep > 64 || per < 1 || ep * per > kMaxExperts) return -1;
  const int waves = std::max(ep * cap, (std::max(A, ep * per) + 63) / 64);
  ep_pack_kernel<<<dim3((waves + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const bf16_t*>(x), static_cast<const int32_t*>(sorted_src), static_cast<const int32_t*>(offsets),
      static_cast<const int32_t*>(slot_of), per, ep, cap, A, d, static_cast<bf16_t*>(send),
      static_cast<int32_t*>(send_counts), static_cast<int32_t*>(slot_back));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_ep_recv_index(const void* recv_counts, int ep, int per, int cap, void* row_index,
                               void* offsets_local, hipStream_t stream) {
  if (ep < 1 || ep > 64 || per < 1 || per > kMaxExperts || cap <= 0) return -1;
  ep_recv_index_kernel<<<dim3(1), dim3(256), 0, stream>>>(static_cast<const int32_t*>(recv_counts), ep, per, cap,
                                                         static_cast<int32_t*>(row_index),
                                                         static_cast<int32_t*>(offsets_local));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_scatter_rows(const void* y, const void* row_index, const void* total, int cap_rows, int d,
                              void* out, hipStream_t stream) {
  if (cap_rows <= 0) return 0;
  if (d % 8) return -1;
  scatter_rows_kernel<<<dim3((cap_rows + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const bf16_t*>(y), static_cast<const int32_t*>(row_index), static_cast<const int32_t*>(total),
      cap_rows, d, static_cast<bf16_t*>(out));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_moe_router(const void* x, const void* Wr, int T, int d, int E, int topk, void* topk_ids,
                            void* topk_w, void* router_logits, hipStream_t stream) {
  if (T <= 0) return 0;
  if (E > kMaxExperts || topk > kMaxTopK || topk > E || d % 8) return -1;
#define LC_ARGS                                                                             \
  static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(Wr), T, d, E, topk,             \
      static_cast<int32_t*>(topk_ids), static_cast<float*>(topk_w), static_cast<float*>(router_logits)
  if (T <= 1024) moe_router_kernel<4><<<dim3(T), dim3(256), 0, stream>>>(LC_ARGS);
  else moe_router_kernel<1><<<dim3((T + 3) / 4), dim3(256), 0, stream>>>(LC_ARGS);
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_moe_permute(const void* topk_ids, int T, int topk, int E, void* offsets, void* sorted_src,
                             void* slot_of, hipStream_t stream) {
  if (T <= 0) return 0;
  if (E > kMaxExperts) return -1;
  // small steps: fewer waves to scan and to synchronise
  const int threads = T * topk <= 256 ? 256 : 1024;
  moe_permute_kernel<<<dim3(1), dim3(threads), 0, stream>>>(static_cast<const int32_t*>(topk_ids), T * topk, topk,
                                                         E, static_cast<int32_t*>(offsets),
                                                         static_cast<int32_t*>(sorted_src),
                                                         static_cast<int32_t*>(slot_of));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_moe_combine(const void* Y, const void* slot_of, const void* topk_w, void* out, int T, int d,
                             int topk, hipStream_t stream) {
  if (T <= 0) return 0;
  if (d % 8) return -1;
  if (topk > kMaxTopK) return -1;
#define LC_ARGS                                                                                             \
  static_cast<const bf16_t*>(Y), static_cast<const int32_t*>(slot_of), static_cast<const float*>(topk_w),   \
      static_cast<bf16_t*>(out), T, d, topk
  if (T <= 1024) moe_combine_kernel<4><<<dim3(T), dim3(256), 0, stream>>>(LC_ARGS);
  else moe_combine_kernel<1><<<dim3((T + 3) / 4), dim3(256), 0, stream>>>(LC_ARGS);
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}
