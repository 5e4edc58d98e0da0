// Large-M bf16 GEMM, 256(n) x 256(m) x 64(k) block tile, FOUR waves of 128 x 128 each: one wave
// per SIMD, its 8 x 8 accumulator fragments (256 registers) in AGPRs, and every memory
// instruction placed by hand between MFMAs.
//
// Against the eight-wave form (gemm256.hip: 64 x 128 per wave, two waves per SIMD taking turns)
// a wave tile twice as large reads each activation fragment from LDS for twice as many MFMAs:
// 64 KiB of LDS fragment reads per K-tile and CU instead of 128, one barrier per K-tile instead of
// four. The price is that nothing but this wave's own instruction stream hides its memory
// latencies, so the stream is laid out as eight groups per k-step,
//     [one or two memory instructions] [8 MFMAs],
// and every operand is requested a whole phase (64 MFMAs) or more before its first use:
//
//   K-tile t (weight fragment sets A0[c], A1[c], c = t & 1), phase k0, per group i of 8:
//       activation fragment i of K-tile t, k-step 1 -> bX[1][i]     (LDS -> registers)
//       8 MFMAs on A0[c][i]
//       weight fragment i of K-tile t+2, k-step 0 -> A0[c][i]       (global -> registers, in place)
//   wait: A1[c] and X(t+1) have landed, own LDS reads retired; BARRIER
//       => X(t+1) is visible to all four waves, and nobody reads stage t % 4 any more
//   phase k1, per group i:
//       (groups 0-3) two LDS-DMA instructions of the activations of K-tile t+4 -> stage t % 4
//       activation fragment i of K-tile t+1, k-step 0 -> bX[0][i]
//       8 MFMAs on A1[c][i]
//       weight fragment i of K-tile t+2, k-step 1 -> A1[c][i]
//   wait: A0[c ^ 1] has landed, LDS reads retired
//
// so every request has two K-tiles (~3 us) to land. Measured against gemm256 on one MI355X
// (benchmarks/micro/gemm256w_bench.hip, 8192 rows, random data, results bit-identical):
// K = 14336 (down projection) 645 vs 666 us (1.49 vs 1.44 PFLOP/s); at K = 4096 the longer
// prologue (four activation stages, two weight sets) costs more than the loop gains: gate/up 1360 vs
// 1311 us, QKV 293 vs 290, O 194 vs 198. ops.linear therefore takes this kernel for K >= 8192 only.
//
// Operand conventions as in gemm256.hip: packed weights (1-KiB MFMA fragments) are the A operand and
// never pass through LDS, activations are the B operand, staged as whole 128-byte rows into an
// XOR-swizzled image. The order of accumulation (k-step 0 before k-step 1 of every K-tile) is
// the same, so the results equal gemm256's bit for bit.
#include "gemm_common.h"

namespace {

constexpr int GW_BN = 256, GW_BM = 256, GW_BK = 64;
constexpr int GW_XSTAGE = 32 * 1024, GW_NS = 4;
constexpr int GW_LDS = GW_NS * GW_XSTAGE;
constexpr int GW_BAND = 4;

LC_DEVICE int gw_x_off(int kt, int lane) {
  const int r = lane & 15, rr = r & 7;
  return (r >> 3) * 1024 + rr * 128 + (((kt * 4 + (lane >> 4)) ^ rr) << 4);
}

#define GW_PIN() __builtin_amdgcn_sched_barrier(0)

template <int EPI>
__global__ __launch_bounds__(256) void gemm256w_kernel(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ X,
                                                       void* __restrict__ Y, const bf16_t* __restrict__ bias, int M,
                                                       int N, int K, int ldy, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 1, wm = wave >> 1;

  // XCD-aware tile order, as in gemm256.hip: each XCD gets a contiguous run of tiles, ordered in
  // bands of GW_BAND n-tiles so that the tiles an XCD runs at once share panels through its L2.
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int band = tile / (GW_BAND * tiles_m), idx = tile % (GW_BAND * tiles_m);
  const int bw = min(GW_BAND, tiles_n - band * GW_BAND);
  const int tn = band * GW_BAND + idx % bw, tm = idx / bw;
  const int m0 = tm * GW_BM;
  const int KT = K / 32, nk = K / GW_BK;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- weights: scalar base (advances 2 KiB per K-tile) + one 32-bit lane offset per n-fragment
  const char* wrow = reinterpret_cast<const char*>(Wp) + static_cast<size_t>(tn * 16 + wn * 8) * KT * 1024;
  uint32_t woff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) woff[i] = static_cast<uint32_t>(i) * KT * 1024u + lane * 16u;
  bf16x8 A0[2][8], A1[2][8], bX[2][8];
  auto wload0 = [&](bf16x8& v, int i, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(woff[i]), "s"(base) : "memory");
  };
  auto wload1 = [&](bf16x8& v, int i, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(v) : "v"(woff[i]), "s"(base) : "memory");
  };
  // ---- activations: this wave stages row tiles q*4 + wave (16 rows each), two 8-row pieces per tile
  const int rr = lane >> 3, lc = lane & 7;
  const bf16_t* xsrc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      int row = m0 + (q * 4 + wave) * 16 + hf * 8 + rr;
      row = row < M ? row : M - 1;
      xsrc[q][hf] = X + static_cast<size_t>(row) * K + ((lc ^ rr) << 3);
    }
  auto xissue = [&](int k_tile, int q) {          // row tile q*4 + wave of K-tile k_tile (clamped) into its stage
    const int kk = k_tile < nk ? k_tile : nk - 1;
    char* dst = smem + (k_tile % GW_NS) * GW_XSTAGE + ((q * 4 + wave) * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(xsrc[q][0] + kk * GW_BK), LC_LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(xsrc[q][1] + kk * GW_BK), LC_LDS_PTR(dst + 1024), 16, 0, 0);
  };
  const int xoff0 = gw_x_off(0, lane), xoff1 = gw_x_off(1, lane);
  auto xfrag = [&](int k_tile, int j, int xoff) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(smem + (k_tile % GW_NS) * GW_XSTAGE + ((wm * 8 + j) * 2) * 1024 + xoff);
  };

  // ---- prologue: the activations of K-tiles 0-3 and the weight fragments of K-tiles 0 and 1
#pragma unroll
  for (int kt = 0; kt < GW_NS; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) xissue(kt, q);
#pragma unroll
  for (int i = 0; i < 8; ++i) wload0(A0[0][i], i, wrow);
#pragma unroll
  for (int i = 0; i < 8; ++i) wload1(A1[0][i], i, wrow);
#pragma unroll
  for (int i = 0; i < 8; ++i) wload0(A0[1][i], i, wrow + (nk > 1 ? 2048 : 0));
#pragma unroll
  for (int i = 0; i < 8; ++i) wload1(A1[1][i], i, wrow + (nk > 1 ? 2048 : 0));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) bX[0][j] = xfrag(0, j, xoff0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // The MFMAs are written out with the accumulator tied to an AGPR tuple ("+a"): with all 256
  // AGPRs taken by accumulators the compiler's own choice was the untied form plus a copy of
  // every accumulator tile through one spare tuple, per MFMA.
#define GW_MFMA8(AF, BF, I)                                                                          \
  _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                      \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][j]) : "v"(AF[I]), "v"(BF[j]))

  // Requests per K-tile and wave, in order: phase k0: 8 weight loads; phase k1: 8 LDS-DMA interleaved
  // with the first 4 of 8 weight loads, then the other 4. Everything is requested two K-tiles
  // before it is used.
#define GW_KTILE(T, CUR)                                                                             \
  do {                                                                                               \
    const int t_ = (T);                                                                              \
    const int t2_ = t_ + 2 < nk ? t_ + 2 : nk - 1;                                                   \
    const char* wnext = wrow + static_cast<size_t>(t2_) * 2048;                                      \
    /* ---- phase k0: fragment i of A0[CUR] is requested again (K-tile t+2) once its MFMAs are out */ \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
      bX[1][i] = xfrag(t_, i, xoff1);                                                                \
      GW_PIN();                                                                                      \
      GW_MFMA8(A0[CUR], bX[0], i);                                                                   \
      GW_PIN();                                                                                      \
      wload0(A0[CUR][i], i, wnext);                                                                  \
    }                                                                                                \
    /* A1[CUR] (K-tile t): requested in phase k1 of K-tile t-2; since then 8 + 16 + 8 requests.  */  \
    /* X(t+1), requested in K-tile t-3, is older.                                                */  \
    asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");                                     \
    __builtin_amdgcn_s_barrier();                                                                    \
    asm volatile("" ::: "memory");                                                                   \
    /* ---- phase k1 */                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
      if (i < 4) xissue(t_ + GW_NS, i);                                                              \
      bX[0][i] = xfrag(t_ + 1, i, xoff0);                                                            \
      GW_PIN();                                                                                      \
      GW_MFMA8(A1[CUR], bX[1], i);                                                                   \
      GW_PIN();                                                                                      \
      wload1(A1[CUR][i], i, wnext);                                                                  \
    }                                                                                                \
    /* A0[CUR ^ 1] (K-tile t+1): requested in phase k0 of K-tile t-1; since then 16 + 8 + 16 */      \
    asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");                                     \
  } while (0)

  int t = 0;
  for (; t + 1 < nk; t += 2) {
    GW_KTILE(t, 0);
    GW_KTILE(t + 1, 1);
  }
  if (t < nk) GW_KTILE(t, 0);
#undef GW_KTILE
  // the last MFMAs (which the compiler does not see as such) must have written their accumulators
  // before the epilogue reads them: 16 passes of 4 cycles
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped requests of the last K-tile
  // The registers of requests that are never consumed stay claimed until the wait above.
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(A0[0][i]), "v"(A1[0][i]), "v"(A0[1][i]), "v"(A1[1][i]));

  const int nbase = tn * GW_BN + wn * 128, mbase = m0 + wm * 128;
  // Epilogue through LDS: stored straight from the accumulator layout a store instruction would
  // touch 16 rows with 32 bytes each. Every wave transposes its 128 x 128 sub-tile through a
  // private 32-KiB region of the (now idle) ring and writes whole rows: 16 bytes per lane, four
  // rows of 256 contiguous bytes per instruction (SwiGLU: eight rows of 128 bytes).
  if ((ldy & 7) == 0 && (tn + 1) * GW_BN <= N) {        // uniform: the whole block tile is inside Y
    __builtin_amdgcn_s_barrier();                        // every wave is past its last ring read, every DMA has landed
    asm volatile("" ::: "memory");
    constexpr int CH = EPI == EPI_SWIGLU ? 8 : 16;       // 16-byte chunks per staged row
    constexpr int ROWB = CH * 16;
    char* mine = smem + wave * (128 * 256);
    const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = j * 16 + r16;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u32x2 o;
        int chunk;
        if constexpr (EPI == EPI_SWIGLU) {
          if (i & 1) continue;
          const int nb = nbase + i * 16 + 4 * g;
          float gt[4], up[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gt[r] = acc[i][j][r] + (bias ? bf16_to_f32(bias[nb + r]) : 0.f);
            up[r] = acc[i + 1][j][r] + (bias ? bf16_to_f32(bias[nb + 16 + r]) : 0.f);
          }
          o[0] = pack_bf16x2(silu(gt[0]) * up[0], silu(gt[1]) * up[1]);
          o[1] = pack_bf16x2(silu(gt[2]) * up[2], silu(gt[3]) * up[3]);
          chunk = (i >> 1) * 2 + (g >> 1);
        } else {
          const int nb = nbase + i * 16 + 4 * g;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + (bias ? bf16_to_f32(bias[nb + r]) : 0.f);
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
          chunk = i * 2 + (g >> 1);
        }
        // chunk c of row m lives at position c ^ (m mod CH): column accesses spread over the banks
        *reinterpret_cast<u32x2*>(mine + m * ROWB + ((chunk ^ (m & (CH - 1))) << 4) + (g & 1) * 8) = o;
      }
    }
    constexpr int RPI = 64 / CH;                         // rows per store instruction
    const int c16 = lane & (CH - 1), rsub = lane / CH;
    const int ncol = (EPI == EPI_SWIGLU ? tn * (GW_BN / 2) + wn * 64 : nbase) + c16 * 8;
#pragma unroll
    for (int p = 0; p < 128 / RPI; ++p) {
      const int m = p * RPI + rsub;
      const u32x4 v = *reinterpret_cast<const u32x4*>(mine + m * ROWB + ((c16 ^ (m & (CH - 1))) << 4));
      if (mbase + m < M)
        *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(Y) + static_cast<size_t>(mbase + m) * ldy + ncol) = v;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[i + 1][j], Y, bias, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Y, bias, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      }
    }
}

}  // namespace

// Y = X * W^T with the four-wave 256 x 256 tile. Requirements: K % 64 == 0, K >= 192, n_pad % 256 == 0.
LC_EXPORT int lc_gemm256w(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K, int n_pad,
                          int ldy, int epi, hipStream_t stream) {
  if (M <= 0) return 0;
  if (K % 64 || K < GW_NS * GW_BK || n_pad % 256 || N > n_pad) return -1;
  const int tiles_m = (M + GW_BM - 1) / GW_BM, tiles_n = n_pad / GW_BN;
  const dim3 grid(tiles_m * tiles_n), block(256);
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256w_kernel<EPI_BF16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256w_kernel<EPI_SWIGLU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS);
  });
  if (epi == EPI_BF16) gemm256w_kernel<EPI_BF16><<<grid, block, GW_LDS, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else if (epi == EPI_SWIGLU)
    gemm256w_kernel<EPI_SWIGLU><<<grid, block, GW_LDS, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else return -2;
  LC_RETURN_LAST_ERROR();
}
