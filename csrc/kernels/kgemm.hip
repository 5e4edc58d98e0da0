// Decode GEMM for small M (<= 64) and narrow outputs: Y[M,N] = X[M,K] * W[N,K]^T where
// N/128 is far below the 256 CUs of the chip (attention QKV / O / MLP-down at batch 64).
//
// Instead of splitting K across workgroups (fp32 partial slabs + a reduce launch), the K
// range is split across the 8 waves of ONE workgroup and reduced through LDS:
//   * a workgroup owns NTW consecutive 16-column n-tiles (16..64 output columns), so even
//     N = 4096 yields 256 workgroups of 8 waves;
//   * wave w streams the packed weight fragments of those n-tiles for k-tiles
//     [w*KT/8, (w+1)*KT/8) straight into VGPRs (nontemporal, read exactly once) together
//     with the matching activation fragments (L2-resident, shared by all workgroups);
//   * there is no LDS staging and no barrier in the streaming loop, so the compiler's own
//     counted vmcnt keeps loads in flight continuously;
//   * the 8 partial accumulator sets are summed through LDS (tree, 3 barriers) and wave 0..
//     apply the epilogue (bias / bf16 / fp32 / fused SwiGLU).
// Same packed-weight layout and MFMA orientation as gemm.hip (weights = A operand).
#include "gemm_common.h"

namespace {

template <int MT, int NTW, int EPI>
__global__ __launch_bounds__(512) void kgemm_kernel(const bf16_t* __restrict__ Wp,
                                                    const bf16_t* __restrict__ X, void* __restrict__ Y,
                                                    const bf16_t* __restrict__ bias, int M, int N, int K,
                                                    int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = NTW * MT;  // accumulator fragments per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KT = K / 32;
  const int per = (KT + 7) / 8;
  const int kt_lo = wave * per, kt_hi = min(KT, kt_lo + per);
  const int nt0 = blockIdx.x * NTW;

  f32x4 acc[NTW][MT];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane row pointers of the activation fragments (rows clamped for the M tail)
  const bf16_t* xrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int row = m * 16 + (lane & 15);
    row = row < M ? row : M - 1;
    xrow[m] = X + static_cast<size_t>(row) * K + 8 * (lane >> 4);
  }
  const bf16_t* wbase = Wp + static_cast<size_t>(nt0) * KT * 512 + lane * 8;

  auto load = [&](bf16x8 (&wf)[NTW], bf16x8 (&xf)[MT], int kt) {
#pragma unroll
    for (int n = 0; n < NTW; ++n)
      wf[n] = __builtin_nontemporal_load(
          reinterpret_cast<const bf16x8*>(wbase + (static_cast<size_t>(n) * KT + kt) * 512));
#pragma unroll
    for (int m = 0; m < MT; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xrow[m] + kt * 32);
  };
  auto fma = [&](const bf16x8 (&wf)[NTW], const bf16x8 (&xf)[MT]) {
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
  };

  // 4-deep register ring, statically indexed (unrolled by 4)
  bf16x8 w0[NTW], w1[NTW], w2[NTW], w3[NTW], x0[MT], x1[MT], x2[MT], x3[MT];
  int kt = kt_lo;
  if (kt + 0 < kt_hi) load(w0, x0, kt + 0);
  if (kt + 1 < kt_hi) load(w1, x1, kt + 1);
  if (kt + 2 < kt_hi) load(w2, x2, kt + 2);
  for (; kt < kt_hi; kt += 4) {
    if (kt + 3 < kt_hi) load(w3, x3, kt + 3);
    fma(w0, x0);
    if (kt + 4 < kt_hi) load(w0, x0, kt + 4);
    if (kt + 1 < kt_hi) fma(w1, x1);
    if (kt + 5 < kt_hi) load(w1, x1, kt + 5);
    if (kt + 2 < kt_hi) fma(w2, x2);
    if (kt + 6 < kt_hi) load(w2, x2, kt + 6);
    if (kt + 3 < kt_hi) fma(w3, x3);
  }

  // ---- cross-wave reduction through LDS: 8 -> 4 -> 2 -> 1 ----
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [4 waves][NF][64 lanes]
#pragma unroll
  for (int half = 4; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int f = 0; f < NF; ++f) red[((wave - half) * NF + f) * 64 + lane] = acc[f / MT][f % MT];
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f / MT][f % MT] += red[(wave * NF + f) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (n & 1) continue;
        store_tile<EPI>(acc[n][m], &acc[(n + 1) % NTW][m], Y, bias, m * 16, (nt0 + n) * 16, M, N, ldy, lane, 0);
      } else {
        store_tile<EPI>(acc[n][m], nullptr, Y, bias, m * 16, (nt0 + n) * 16, M, N, ldy, lane, 0);
      }
    }
}

template <int MT, int NTW>
int launch_k(int epi, const bf16_t* Wp, const bf16_t* X, void* Y, const bf16_t* bias, int M, int N, int K,
             int ldy, int n_pad, hipStream_t stream) {
  const dim3 grid(n_pad / (16 * NTW)), block(512);
  const size_t lds = 4 * NTW * MT * 64 * sizeof(f32x4);
  if (epi == EPI_BF16)
    kgemm_kernel<MT, NTW, EPI_BF16><<<grid, block, lds, stream>>>(Wp, X, Y, bias, M, N, K, ldy);
  else if (epi == EPI_F32)
    kgemm_kernel<MT, NTW, EPI_F32><<<grid, block, lds, stream>>>(Wp, X, Y, bias, M, N, K, ldy);
  else if constexpr (NTW >= 2)
    kgemm_kernel<MT, NTW, EPI_SWIGLU><<<grid, block, lds, stream>>>(Wp, X, Y, bias, M, N, K, ldy);
  else
    return -3;
  return static_cast<int>(hipGetLastError());
}

}  // namespace

// M <= 64, K % 32 == 0, n_pad % 128 == 0. ntw = n-tiles (16 columns) per workgroup: 1, 2 or 4.
LC_EXPORT int lc_kgemm(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K,
                       int n_pad, int ldy, int epi, int ntw, hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 64 || K % 32 || n_pad % 128 || N > n_pad) return -1;
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
  const int mt = (M + 15) / 16;
#define LC_K(MT_)                                                                          \
  if (ntw == 1) return launch_k<MT_, 1>(epi, w, x, Y, b, M, N, K, ldy, n_pad, stream);     \
  if (ntw == 2) return launch_k<MT_, 2>(epi, w, x, Y, b, M, N, K, ldy, n_pad, stream);     \
  if (ntw == 4) return launch_k<MT_, 4>(epi, w, x, Y, b, M, N, K, ldy, n_pad, stream);     \
  return -1;
  if (mt <= 1) { LC_K(1) }
  if (mt <= 2) { LC_K(2) }
  LC_K(4)
#undef LC_K
}
