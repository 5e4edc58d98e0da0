// Memory-bound row kernels: embedding gather (also the row gather and the device-side token
// feed), RMSNorm (+ fused residual add and split-K slab sum), KV block copy, embedding pooling,
// small fp32 helpers (accumulate, column padding, column concatenation). SwiGLU is a GEMM epilogue. All use 16-byte per-lane accesses and one wave per
// row; none of them exists in the reference (its model is remote,
// /root/reference/src/main.rs:82-86).
#include "common.h"

namespace {

constexpr int kMaxChunksPerLane = 16;  // rows up to 16*64*8 = 8192 elements stay in registers

// x[t,:] = E[ids[t],:]. With a token feed, rows whose feed_rows[t] >= 0 take their id from
// feed_tok[feed_rows[t]] instead: the token a row of the PREVIOUS decode step sampled, which is
// still on the device when this step is launched ahead of that step's results.
__global__ void embed_gather_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ E,
                                    bf16_t* __restrict__ x, int T, int d, int vocab,
                                    const int32_t* __restrict__ feed_tok, const int32_t* __restrict__ feed_rows) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  int id = ids[row];
  if (feed_rows != nullptr) {
    const int src = feed_rows[row];
    if (src >= 0) id = feed_tok[src];
  }
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4* src = reinterpret_cast<const u32x4*>(E + static_cast<size_t>(id) * d);
  u32x4* dst = reinterpret_cast<u32x4*>(x + static_cast<size_t>(row) * d);
  for (int c = lane; c < d / 8; c += 64) dst[c] = src[c];
}

// Residual add + RMSNorm. The addend is either `delta` (bf16 [T, d]) or, when `part` is
// given, the sum of S fp32 slabs [S][T][d] left un-reduced by a split-K GEMM (the slab sum
// happens here instead of in a kernel of its own). With an addend, h += addend (h is the
// bf16 residual stream, updated in place); then y = h * rsqrt(mean(h^2) + eps) * w.
// WAVES_PER_ROW = 1: one wave per row, CH chunks of 8 per lane (large T, bandwidth-bound).
// WAVES_PER_ROW = 4: one 256-thread block per row (decode: few rows, latency-bound, so the
// row is spread over 4x the lanes and the gain vector is fetched before the reduction).
template <int CH, int WAVES_PER_ROW>
__global__ __launch_bounds__(256) void rmsnorm_kernel(bf16_t* __restrict__ h, const bf16_t* __restrict__ delta,
                                                      const float* __restrict__ part, int S,
                                                      const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                      int T, int d, float eps) {
  __shared__ float red[4];
  constexpr int LANES = 64 * WAVES_PER_ROW;
  const int row = WAVES_PER_ROW == 1 ? blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : blockIdx.x;
  const int lane = WAVES_PER_ROW == 1 ? (threadIdx.x & 63) : threadIdx.x;
  if (row >= T) return;
  const int nchunk = d / 8;
  const size_t slab = static_cast<size_t>(T) * d;
  bf16_t* hrow = h + static_cast<size_t>(row) * d;
  const bf16_t* drow = delta ? delta + static_cast<size_t>(row) * d : nullptr;
  const float* prow = part ? part + static_cast<size_t>(row) * d : nullptr;
  bf16x8 v[CH], g[CH];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane + i * LANES;
    if (c < nchunk) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + c * 8);
      g[i] = *reinterpret_cast<const bf16x8*>(w + c * 8);
      if (prow) {
        // Eight slabs' loads at a time are issued before the first add: a loop of load-then-add
        // pays one L2 round trip per slab, and at decode sizes this kernel is all round trips.
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        for (int sp0 = 0; sp0 < S; sp0 += 8) {
          f32x4 pl[8], ph[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (sp0 + u < S) {
              pl[u] = *reinterpret_cast<const f32x4*>(prow + (sp0 + u) * slab + c * 8);
              ph[u] = *reinterpret_cast<const f32x4*>(prow + (sp0 + u) * slab + c * 8 + 4);
            }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (sp0 + u < S) {
              lo += pl[u];
              hi += ph[u];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = static_cast<short>(f32_to_bf16(bf16_to_f32(static_cast<bf16_t>(a[j])) +
                                                bf16_to_f32(f32_to_bf16(lo[j]))));
          a[j + 4] = static_cast<short>(f32_to_bf16(bf16_to_f32(static_cast<bf16_t>(a[j + 4])) +
                                                    bf16_to_f32(f32_to_bf16(hi[j]))));
        }
        *reinterpret_cast<bf16x8*>(hrow + c * 8) = a;
      } else if (drow) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(drow + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          a[j] = static_cast<short>(f32_to_bf16(bf16_to_f32(static_cast<bf16_t>(a[j])) +
                                                bf16_to_f32(static_cast<bf16_t>(b[j]))));
        *reinterpret_cast<bf16x8*>(hrow + c * 8) = a;
      }
      v[i] = a;  // normalise exactly what the residual stream now holds
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf16_to_f32(static_cast<bf16_t>(a[j]));
        ss += f * f;
      }
    }
  }
  ss = wave_sum(ss);
  if constexpr (WAVES_PER_ROW > 1) {
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
  }
  const float scale = rsqrtf(ss / static_cast<float>(d) + eps);
  bf16_t* yrow = y + static_cast<size_t>(row) * d;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane + i * LANES;
    if (c < nchunk) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = static_cast<short>(f32_to_bf16(bf16_to_f32(static_cast<bf16_t>(v[i][j])) * scale *
                                              bf16_to_f32(static_cast<bf16_t>(g[i][j]))));
      *reinterpret_cast<bf16x8*>(yrow + c * 8) = o;
    }
  }
}

// Copy-on-write of KV blocks: for every (src, dst) pair copy `block_bytes` bytes in each
// of the `n_planes` cache planes (planes are `plane_stride` bytes apart).
__global__ void copy_blocks_kernel(char* __restrict__ base, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, size_t block_bytes,
                                   size_t plane_stride) {
  const int pair = blockIdx.x;
  const size_t plane = blockIdx.y;
  const u32x4* s = reinterpret_cast<const u32x4*>(base + plane * plane_stride + src[pair] * block_bytes);
  u32x4* d = reinterpret_cast<u32x4*>(base + plane * plane_stride + dst[pair] * block_bytes);
  for (size_t i = threadIdx.x; i < block_bytes / 16; i += blockDim.x) d[i] = s[i];
}

// Pooled sentence embedding, one block per sequence, 8 columns (16 B) per lane and load.
// Rows [begin[s], end[s]) of `hidden` belong to sequence s.
// mode 0: mean over the rows, L2-normalised; mode 1: last row, L2-normalised; mode 2: plain
// fp32 sum of the rows (a prompt prefilled in several chunks adds its chunk sums and
// normalises once at the end: l2_normalize_kernel).
__global__ __launch_bounds__(256) void embed_pool_kernel(const bf16_t* __restrict__ hidden,
                                                         const int32_t* __restrict__ begin,
                                                         const int32_t* __restrict__ end,
                                                         float* __restrict__ out, int d, int mode) {
  __shared__ float red[16];
  const int s = blockIdx.x;
  const int r0 = begin[s], r1 = end[s];
  const int n = r1 - r0;
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n > 0) {
      for (int r = (mode == 1 ? r1 - 1 : r0); r < r1; ++r) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(hidden + static_cast<size_t>(r) * d + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32(static_cast<bf16_t>(v[j]));
      }
      if (mode == 0) {
        const float inv_n = 1.f / static_cast<float>(n);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv_n;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      out[static_cast<size_t>(s) * d + c + j] = acc[j];
      ss += acc[j] * acc[j];
    }
  }
  if (mode == 2) return;
  ss = block_sum(ss, red);
  const float inv = ss > 0.f ? rsqrtf(ss) : 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) out[static_cast<size_t>(s) * d + c] *= inv;
}

// y[r, :] = x[r, :] / |x[r, :]| (fp32 rows; zero rows stay zero). One block per row.
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int d) {
  __shared__ float red[16];
  const float* xr = x + static_cast<size_t>(blockIdx.x) * d;
  float ss = 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) ss += xr[c] * xr[c];
  ss = block_sum(ss, red);
  const float inv = ss > 0.f ? rsqrtf(ss) : 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) y[static_cast<size_t>(blockIdx.x) * d + c] = xr[c] * inv;
}

// y[i] += x[i] (fp32): chunk sums of a pooled embedding that spans several prefill steps.
__global__ void accumulate_f32_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    y[i] += x[i];
}

// out[r, c] = x[r, c] for c < n, `fill` for n <= c < n_pad (fp32 rows widened to a common width).
__global__ void pad_cols_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int n, int n_pad,
                                    float fill) {
  const long long total = static_cast<long long>(R) * n_pad;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int r = static_cast<int>(i / n_pad), c = static_cast<int>(i % n_pad);
    out[i] = c < n ? x[static_cast<size_t>(r) * n + c] : fill;
  }
}

// out[r, w * c + j] = parts[w, r, j] for w * c + j < total (units of 2 bytes): the column
// all-gather's [W, R, c] blocks laid side by side, ragged last shard cut off.
__global__ void concat_cols_kernel(const uint16_t* __restrict__ parts, int W, int R, int c, int total,
                                   uint16_t* __restrict__ out) {
  const long long n = static_cast<long long>(R) * total;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int r = static_cast<int>(i / total), col = static_cast<int>(i % total);
    const int w = col / c, j = col % c;
    out[i] = parts[(static_cast<size_t>(w) * R + r) * c + j];
  }
}

}  // namespace

LC_EXPORT int lc_concat_cols(const void* parts, int W, int R, int c_units, int total_units, void* out,
                             hipStream_t stream) {
  if (R <= 0 || total_units <= 0) return 0;
  if (W < 1 || c_units < 1 || total_units > W * c_units) return -1;
  const long long n = static_cast<long long>(R) * total_units;
  const int grid = static_cast<int>(std::min<long long>((n + 255) / 256, 8192));
  concat_cols_kernel<<<dim3(grid), dim3(256), 0, stream>>>(static_cast<const uint16_t*>(parts), W, R, c_units,
                                                          total_units, static_cast<uint16_t*>(out));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_accumulate_f32(void* y, const void* x, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = static_cast<int>(std::min<long long>((n + 255) / 256, 1024));
  accumulate_f32_kernel<<<dim3(grid), dim3(256), 0, stream>>>(static_cast<float*>(y), static_cast<const float*>(x), n);
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_pad_cols_f32(const void* x, void* out, int R, int n, int n_pad, float fill, hipStream_t stream) {
  if (R <= 0 || n_pad <= 0) return 0;
  if (n > n_pad) return -1;
  const long long total = static_cast<long long>(R) * n_pad;
  const int grid = static_cast<int>(std::min<long long>((total + 255) / 256, 4096));
  pad_cols_f32_kernel<<<dim3(grid), dim3(256), 0, stream>>>(static_cast<const float*>(x), static_cast<float*>(out), R,
                                                           n, n_pad, fill);
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_embed_gather(const void* ids, const void* E, void* x, int T, int d, int vocab,
                              const void* feed_tok, const void* feed_rows, hipStream_t stream) {
  if (T <= 0) return 0;
  if (d % 8 || ((feed_tok == nullptr) != (feed_rows == nullptr))) return -1;
  embed_gather_kernel<<<dim3((T + 3) / 4), dim3(256), 0, stream>>>(
      static_cast<const int32_t*>(ids), static_cast<const bf16_t*>(E), static_cast<bf16_t*>(x), T, d, vocab,
      static_cast<const int32_t*>(feed_tok), static_cast<const int32_t*>(feed_rows));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_rmsnorm(void* h, const void* delta, const void* w, void* y, int T, int d, float eps,
                         const void* part, int S, hipStream_t stream) {
  if (T <= 0) return 0;
  if (d % 8 || d > kMaxChunksPerLane * 64 * 8 || (part && (S < 1 || delta))) return -1;
#define LC_RMS(CH, WPR, GRID)                                                                   \
  rmsnorm_kernel<CH, WPR><<<dim3(GRID), dim3(256), 0, stream>>>(                                 \
      static_cast<bf16_t*>(h), static_cast<const bf16_t*>(delta), static_cast<const float*>(part), S, \
      static_cast<const bf16_t*>(w), static_cast<bf16_t*>(y), T, d, eps)
  if (T <= 1024) {  // few rows: a block per row
    if (d <= 1 * 2048) LC_RMS(1, 4, T);
    else if (d <= 2 * 2048) LC_RMS(2, 4, T);
    else LC_RMS(4, 4, T);
  } else {
    const int grid = (T + 3) / 4;
    if (d <= 2 * 512) LC_RMS(2, 1, grid);
    else if (d <= 4 * 512) LC_RMS(4, 1, grid);
    else if (d <= 8 * 512) LC_RMS(8, 1, grid);
    else LC_RMS(16, 1, grid);
  }
#undef LC_RMS
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_copy_blocks(void* base, const void* src, const void* dst, int n_pairs,
                             long long block_bytes, long long plane_stride, int n_planes,
                             hipStream_t stream) {
  if (n_pairs <= 0) return 0;
  if (block_bytes % 16) return -1;
  copy_blocks_kernel<<<dim3(n_pairs, n_planes), dim3(256), 0, stream>>>(
      static_cast<char*>(base), static_cast<const int32_t*>(src), static_cast<const int32_t*>(dst),
      static_cast<size_t>(block_bytes), static_cast<size_t>(plane_stride));
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_embed_pool(const void* hidden, const void* begin, const void* end, void* out, int n_seqs, int d,
                            int mode, hipStream_t stream) {
  if (n_seqs <= 0) return 0;
  if (d % 8 || mode < 0 || mode > 2) return -1;
  embed_pool_kernel<<<dim3(n_seqs), dim3(256), 0, stream>>>(
      static_cast<const bf16_t*>(hidden), static_cast<const int32_t*>(begin), static_cast<const int32_t*>(end),
      static_cast<float*>(out), d, mode);
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_l2_normalize(const void* x, void* y, int rows, int d, hipStream_t stream) {
  if (rows <= 0) return 0;
  l2_normalize_kernel<<<dim3(rows), dim3(256), 0, stream>>>(static_cast<const float*>(x), static_cast<float*>(y), d);
  LC_RETURN_LAST_ERROR();
}
