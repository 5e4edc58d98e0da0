// bf16 MFMA GEMMs for Y[M,N] = X[M,K] * W[N,K]^T on gfx950.
//
// Weights are stored pre-shuffled in MFMA-fragment order ("packed"):
//     Wp[((nt*KT + kt)*64 + lane)*8 + j] = W[nt*16 + (lane&15)][kt*32 + 8*(lane>>4) + j]
// so one wave-wide 16-byte load (or one global_load_lds_dwordx4) fetches exactly one
// 16(n) x 32(k) operand fragment as 1 KiB of contiguous memory, and the LDS image of
// a staged tile is lane-linear (conflict-free ds_read_b128, no swizzle needed).
//
// Orientation: the weight fragment is the MFMA A operand (rows = n) and the
// activation fragment the B operand (cols = m), i.e. the kernels compute Y^T tiles.
// A lane then owns 4 consecutive n of one row m: an 8-byte bf16 store.
//
// Kernels of this file:
//   * gemm_tile_kernel    -- 128x128x64 block tile, 4 waves (2x2), both operands staged
//     global->LDS with global_load_lds_dwordx4, double-buffered: outputs whose width is not
//     a multiple of 256 (tiny models); everything else goes to gemm256.hip / gemm_panel.hip.
//   * gemm_grouped_kernel -- the same tile per expert over ragged row groups (MoE).
//   * splitk_reduce_kernel -- sums the fp32 slabs of a split-K GEMM.
// Epilogues: bias add, bf16 or fp32 store, fused SwiGLU (gate/up n-tiles interleaved).
#include "gemm_common.h"

namespace {

// ---------------------------------------------------------------------------------
// Large-M kernel. Block tile 128(n) x 128(m) x 64(k); waves 2(n) x 2(m), 64x64 each.
// LDS per stage: W 8 n-tiles x 2 k-tiles x 1 KiB + X 8 m-tiles x 2 k-tiles x 1 KiB = 32 KiB.
// ---------------------------------------------------------------------------------
constexpr int GT_BN = 128, GT_BM = 128, GT_BK = 64;
constexpr int GT_STAGE_BYTES = 32 * 1024;

LC_DEVICE void gt_stage(char* stage, const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ X,
                        int nt0, int m0, int m_hi, const int32_t* __restrict__ row_index, int kt0, int KT,
                        int K, int wave, int lane) {
  // 32 one-KiB pieces per stage, 8 per wave: pieces [0,16) are W, [16,32) are X.
  // X rows are clamped to m_hi-1 (tail tiles) and optionally gathered through row_index.
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = wave * 8 + i;
    char* dst = stage + p * 1024;  // wave-uniform; the DMA adds lane*16
    if (p < 16) {
      const int nt = p >> 1, kt = p & 1;
      const bf16_t* src = Wp + (static_cast<size_t>(nt0 + nt) * KT + (kt0 + kt)) * 512 + lane * 8;
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, 0);
    } else {
      const int q = p - 16;
      const int mt = q >> 1, kt = q & 1;
      int row = m0 + mt * 16 + (lane & 15);
      row = row < m_hi ? row : m_hi - 1;
      if (row_index) row = row_index[row];
      const bf16_t* src = X + static_cast<size_t>(row) * K + (kt0 + kt) * 32 + 8 * (lane >> 4);
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, 0);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_tile_kernel(const bf16_t* __restrict__ Wp,
                                                        const bf16_t* __restrict__ X,
                                                        void* __restrict__ Y,
                                                        const bf16_t* __restrict__ bias, int M, int N,
                                                        int K, int ldy, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave >> 1, wm = wave & 1;

  // XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous run of
  // tiles (bijective form for grids that are not a multiple of 8).
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  // m fastest: consecutive tiles reuse the same 128-column weight panel from L2.
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * GT_BM, nt0 = tn * (GT_BN / 16);
  const int KT = K / 32;
  const int nk = K / GT_BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gt_stage(smem, Wp, X, nt0, m0, M, nullptr, 0, KT, K, wave, lane);
  __syncthreads();  // emits s_waitcnt vmcnt(0) for the LDS-DMA before the barrier

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * GT_STAGE_BYTES;
    if (t + 1 < nk)
      gt_stage(smem + ((t + 1) & 1) * GT_STAGE_BYTES, Wp, X, nt0, m0, M, nullptr, (t + 1) * 2, KT, K, wave, lane);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(cur + ((wn * 4 + i) * 2 + kt) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(cur + 16 * 1024 + ((wm * 4 + j) * 2 + kt) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  const int nbase = tn * GT_BN + wn * 64, mbase = m0 + wm * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[i + 1][j], Y, bias, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Y, bias, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      }
    }
}


// ---------------------------------------------------------------------------------
// Grouped (MoE) variant of the tile kernel: rows [offsets[e], offsets[e+1]) of the
// expert-sorted activation matrix multiply expert e's packed weights. Blocks are laid
// out as (m-tile slot, n-tile); a block finds its expert by walking the E+1 offsets and
// exits if its slot is past the last used tile. `row_index` (optional) gathers source
// rows, so the permutation of tokens into expert order never materialises a copy.
// ---------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const bf16_t* __restrict__ Wp_all,
                                                           long long expert_stride,
                                                           const bf16_t* __restrict__ X,
                                                           const int32_t* __restrict__ row_index,
                                                           const int32_t* __restrict__ offsets, int E,
                                                           void* __restrict__ Y, int N, int K, int ldy,
                                                           int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave >> 1, wm = wave & 1;
  const int tn = blockIdx.x % tiles_n;
  int slot = blockIdx.x / tiles_n;
  int e = 0, m0 = 0, m_hi = 0;
  bool found = false;
  for (e = 0; e < E; ++e) {
    const int lo = offsets[e], hi = offsets[e + 1];
    const int nt = (hi - lo + GT_BM - 1) / GT_BM;
    if (slot < nt) {
      m0 = lo + slot * GT_BM;
      m_hi = hi;
      found = true;
      break;
    }
    slot -= nt;
  }
  if (!found) return;
  const bf16_t* Wp = Wp_all + static_cast<size_t>(e) * expert_stride;
  const int nt0 = tn * (GT_BN / 16);
  const int KT = K / 32, nk = K / GT_BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gt_stage(smem, Wp, X, nt0, m0, m_hi, row_index, 0, KT, K, wave, lane);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * GT_STAGE_BYTES;
    if (t + 1 < nk)
      gt_stage(smem + ((t + 1) & 1) * GT_STAGE_BYTES, Wp, X, nt0, m0, m_hi, row_index, (t + 1) * 2, KT, K, wave, lane);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(cur + ((wn * 4 + i) * 2 + kt) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(cur + 16 * 1024 + ((wm * 4 + j) * 2 + kt) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const int nbase = tn * GT_BN + wn * 64, mbase = m0 + wm * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[i + 1][j], Y, nullptr, mbase + j * 16, nbase + i * 16, m_hi, N, ldy, lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Y, nullptr, mbase + j * 16, nbase + i * 16, m_hi, N, ldy, lane, 0);
      }
    }
}

// Sums S fp32 slabs [S, M, N] (+bias) and writes bf16 (mode 0), fp32 (mode 1) or
// SwiGLU over gate/up n-tiles interleaved at 16 columns (mode 2, out [M, N/2]).
__global__ void splitk_reduce_kernel(const float* __restrict__ part, const bf16_t* __restrict__ bias,
                                     void* __restrict__ Y, int S, int M, int N, int ldy, int mode) {
  const size_t total4 = static_cast<size_t>(M) * (N / 4);
  const size_t slab = static_cast<size_t>(M) * N;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total4;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const size_t m = i / (N / 4);
    const int n = static_cast<int>(i % (N / 4)) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + m * N + n);
    for (int s = 1; s < S; ++s) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(part + s * slab + m * N + n);
      v += p;
    }
    if (bias)
      for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(bias[n + r]);
    if (mode == 0) {
      u32x2 o;
      o[0] = pack_bf16x2(v[0], v[1]);
      o[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(Y) + m * ldy + n) = o;
    } else if (mode == 1) {
      *reinterpret_cast<f32x4*>(static_cast<float*>(Y) + m * ldy + n) = v;
    } else {
      // gate tile = even 16-column tile, up tile = the next one
      const int tile = n >> 4;
      if (tile & 1) continue;
      f32x4 u = *reinterpret_cast<const f32x4*>(part + m * N + n + 16);
      for (int s = 1; s < S; ++s) u += *reinterpret_cast<const f32x4*>(part + s * slab + m * N + n + 16);
      if (bias)
        for (int r = 0; r < 4; ++r) u[r] += bf16_to_f32(bias[n + 16 + r]);
      u32x2 o;
      o[0] = pack_bf16x2(silu(v[0]) * u[0], silu(v[1]) * u[1]);
      o[1] = pack_bf16x2(silu(v[2]) * u[2], silu(v[3]) * u[3]);
      *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(Y) + m * ldy + (tile >> 1) * 16 + (n & 15)) = o;
    }
  }
}

}  // namespace

// Y = X * W^T with a 128x128 tile. Requirements: K % 64 == 0, n_pad % 128 == 0 (the
// packed weight buffer holds n_pad/16 n-tiles; rows >= N are zero), ldy = row stride of Y.
LC_EXPORT int lc_gemm_tile(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K,
                           int n_pad, int ldy, int epi, hipStream_t stream) {
  if (M <= 0) return 0;
  if (K % 64 || n_pad % 128 || N > n_pad) return -1;
  const int tiles_m = (M + GT_BM - 1) / GT_BM, tiles_n = n_pad / GT_BN;
  const dim3 grid(tiles_m * tiles_n), block(256);
  const size_t lds = 2 * GT_STAGE_BYTES;
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
  if (epi == EPI_BF16)
    gemm_tile_kernel<EPI_BF16><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else if (epi == EPI_F32)
    gemm_tile_kernel<EPI_F32><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else if (epi == EPI_SWIGLU)
    gemm_tile_kernel<EPI_SWIGLU><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else
    return -2;
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_splitk_reduce(const void* part, const void* bias, void* Y, int S, int M, int N, int ldy,
                               int mode, hipStream_t stream) {
  if (M <= 0) return 0;
  if (N % 32) return -1;
  const size_t total4 = static_cast<size_t>(M) * (N / 4);
  const int grid = static_cast<int>(std::min<size_t>((total4 + 255) / 256, 2048));
  splitk_reduce_kernel<<<dim3(grid), dim3(256), 0, stream>>>(static_cast<const float*>(part),
                                                             static_cast<const bf16_t*>(bias), Y, S, M, N,
                                                             ldy, mode);
  LC_RETURN_LAST_ERROR();
}

// Grouped GEMM over E experts. offsets (device, int32 [E+1]) delimit each expert's rows
// in the expert-sorted order; total_rows bounds the launch (tiles beyond the used ones
// exit immediately). Output rows are in sorted order.
LC_EXPORT int lc_gemm_grouped(const void* Wp_all, long long expert_stride, const void* X,
                              const void* row_index, const void* offsets, int E, void* Y, int total_rows,
                              int N, int K, int n_pad, int ldy, int epi, hipStream_t stream) {
  if (total_rows <= 0) return 0;
  if (K % 64 || n_pad % 128 || N > n_pad || E < 1) return -1;
  const int tiles_n = n_pad / GT_BN;
  const int max_tiles_m = (total_rows + GT_BM - 1) / GT_BM + E;
  const dim3 grid(max_tiles_m * tiles_n), block(256);
  const size_t lds = 2 * GT_STAGE_BYTES;
  auto w = static_cast<const bf16_t*>(Wp_all);
  auto x = static_cast<const bf16_t*>(X);
  auto ri = static_cast<const int32_t*>(row_index);
  auto off = static_cast<const int32_t*>(offsets);
  if (epi == EPI_BF16)
    gemm_grouped_kernel<EPI_BF16><<<grid, block, lds, stream>>>(w, expert_stride, x, ri, off, E, Y, N, K, ldy, tiles_n);
  else if (epi == EPI_SWIGLU)
    gemm_grouped_kernel<EPI_SWIGLU><<<grid, block, lds, stream>>>(w, expert_stride, x, ri, off, E, Y, N, K, ldy, tiles_n);
  else
    return -2;
  LC_RETURN_LAST_ERROR();
}
