// GEMM for one 256-row activation panel against a wide weight matrix (M <= 256, N large):
// the gate/up and vocabulary projections of a decode step that carries ~256 sequences
// (K-chain self-consistency). At this M a weight element is used 256 times, so the product is
// MFMA- and HBM-bound at once: streaming kernels that hold 16 m-tiles per wave drown in LDS
// reads, and the 256x256 tile kernel covers N/256 < 256 CUs. This kernel uses a
// 128(n) x 256(m) x 64(k) tile, so N/128 workgroups fill the chip with ONE tile each.
//
// Operand conventions as in gemm.hip / gemm256.hip (packed weights = MFMA A operand,
// activations = B operand, both staged with LDS-DMA into lane-linear fragment images).
//
// Wave (wn = w&1, wm = w>>1) owns 64 n x 64 m = 4 x 4 accumulator fragments. A K-tile is
// 48 pieces of 1 KiB (W: 8 n-tiles x 2 k-tiles in fragment order; X: 32 groups of 8 rows x 128 B,
// whole cache lines, XOR-swizzled); wave w moves the two W pieces of n-tile w and X pieces w,
// w+8, w+16, w+24. Three LDS stages form a ring
// with a prefetch distance of two K-tiles, and one barrier per K-tile does both jobs:
//
//     wait vmcnt(6)      this wave's pieces of K-tile t have landed (K-tile t+1 stays in flight)
//     barrier            => K-tile t is visible to everybody, and everybody is done with t-1
//     issue K-tile t+2   into the stage K-tile t-1 occupied
//     32 MFMAs           two halves (m-lo, m-hi); the m-hi activation fragments are read
//                        while the m-lo MFMAs execute
//
// The DMA queue is never drained inside the loop.
#include "gemm_common.h"

namespace {

constexpr int GM_BN = 128, GM_BM = 256, GM_BK = 64;
constexpr int GM_WBYTES = 16 * 1024, GM_STAGE = 48 * 1024, GM_NSTAGE = 3;

struct GmCtx {
  const bf16_t* __restrict__ Wp;
  const bf16_t* __restrict__ X;
  char* smem;
  int nt0, M, K, KT, nk, wave, lane;
};

LC_DEVICE void gm_issue_tile(const GmCtx& c, int k_tile) {
  if (k_tile >= c.nk) return;
  char* stage = c.smem + (k_tile % GM_NSTAGE) * GM_STAGE;
  const int kt0 = k_tile * 2;
  {
    const bf16_t* src = c.Wp + (static_cast<size_t>(c.nt0 + c.wave) * c.KT + kt0) * 512 + c.lane * 8;
    char* dst = stage + (c.wave * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src + 512), LC_LDS_PTR(dst + 1024), 16, 0, 0);
  }
  // Activations: 32 pieces of 8 rows x 128 B (the K-tile's full width, i.e. whole cache lines).
  // The LDS image of a DMA is lane-linear, so the XOR swizzle that keeps the fragment reads
  // off each other's banks is applied on the source side: lane (rr = l>>3, lc = l&7) fetches
  // 16-byte chunk lc^rr of row rr, and chunk c of row rr is later read at position c^rr.
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = q * 8 + c.wave;
    const int rr = c.lane >> 3, lc = c.lane & 7;
    int row = piece * 8 + rr;
    row = row < c.M ? row : c.M - 1;
    const bf16_t* src = c.X + static_cast<size_t>(row) * c.K + k_tile * GM_BK + ((lc ^ rr) << 3);
    char* dst = stage + GM_WBYTES + piece * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, 0);
  }
}

// Byte offset inside the X image of the B fragment (m-tile mt, k-tile kt) for this lane.
LC_DEVICE int gm_x_off(int mt, int kt, int lane) {
  const int r = lane & 15, rr = r & 7;
  const int piece = mt * 2 + (r >> 3);
  const int chunk = kt * 4 + (lane >> 4);
  return piece * 1024 + rr * 128 + ((chunk ^ rr) << 4);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_m256_kernel(const bf16_t* __restrict__ Wp,
                                                        const bf16_t* __restrict__ X, void* __restrict__ Y,
                                                        const bf16_t* __restrict__ bias, int M, int N, int K,
                                                        int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  GmCtx c{Wp, X, smem, static_cast<int>(blockIdx.x) * (GM_BN / 16), M, K, K / 32, K / GM_BK, wave, lane};

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4][2], b0[2][2], b1[2][2];

  gm_issue_tile(c, 0);
  gm_issue_tile(c, 1);

  for (int t = 0; t < c.nk; ++t) {
    if (t + 1 < c.nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    gm_issue_tile(c, t + 2);

    const char* stage = smem + (t % GM_NSTAGE) * GM_STAGE;
    const char* wimg = stage + (wn * 4 * 2) * 1024 + lane * 16;
    const char* ximg = stage + GM_WBYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) a[i][kt] = *reinterpret_cast<const bf16x8*>(wimg + (i * 2 + kt) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        b0[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + gm_x_off(wm * 4 + j, kt, lane));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        b1[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + gm_x_off(wm * 4 + 2 + j, kt, lane));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kt], b0[j][kt], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kt], b1[j][kt], acc[i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    // every LDS read of this K-tile has been consumed by an MFMA above, so passing the next
    // barrier implies this wave is done with the stage
  }

  const int nbase = blockIdx.x * GM_BN + wn * 64, mbase = wm * 64;
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

}  // namespace

// Y[M<=256, N] = X * W^T. Requirements: K % 64 == 0, K >= 128, n_pad % 128 == 0.
LC_EXPORT int lc_gemm_m256(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K,
                           int n_pad, int ldy, int epi, hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > GM_BM || K % 64 || K < 128 || n_pad % GM_BN || N > n_pad) return -1;
  const dim3 grid(n_pad / GM_BN), block(512);
  const size_t lds = GM_NSTAGE * GM_STAGE;
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_m256_kernel<EPI_BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_m256_kernel<EPI_F32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_m256_kernel<EPI_SWIGLU>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    attr_set = true;
  }
  if (epi == EPI_BF16)
    gemm_m256_kernel<EPI_BF16><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy);
  else if (epi == EPI_F32)
    gemm_m256_kernel<EPI_F32><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy);
  else if (epi == EPI_SWIGLU)
    gemm_m256_kernel<EPI_SWIGLU><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy);
  else
    return -2;
  LC_RETURN_LAST_ERROR();
}
