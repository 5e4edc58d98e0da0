// Skinny GEMM for decode steps: ONE activation panel of BM <= 256 rows against a weight
// matrix that is streamed from HBM exactly once. Y[M, N] = X[M, K] * W[N, K]^T.
//
// At these shapes the weights dominate the traffic (a 7-8B model streams ~15 GB per decode
// step) and the job is to keep enough loads in flight per CU to saturate HBM while the
// MFMA work rides along. The kernel is a 128(n) x BM(m) x 64(k) tile whose K loop runs over
// a deep ring of LDS stages filled by LDS-DMA (global_load_lds, 16 B per lane):
//
//     BM    stage (W 16 KiB + X BM*128 B)   stages   K-tiles in flight   wave grid (n x m)   frags / wave
//     256   48 KiB                          3        2                   2 x 4               4 nt x 4 mt
//     128   32 KiB                          4        3                   4 x 2               2 nt x 4 mt
//      64   24 KiB                          6        5                   4 x 2               2 nt x 2 mt
//      32   20 KiB                          7        6                   4 x 2               2 nt x 1 mt
//
// so at BM = 64 a CU has up to 5 x 24 KiB = 120 KiB of loads outstanding, without spending a
// single VGPR on staging. One barrier per K-tile does both jobs of the ring:
//
//     wait vmcnt(P*(D-1))   this wave's P pieces of K-tile t have landed; D-1 tiles stay in flight
//     barrier               => K-tile t is visible to everybody, and everybody is done with t-1
//     issue K-tile t+D      into the stage K-tile t-1 occupied
//     MFMAs of K-tile t
//
// Operand conventions as in gemm.hip / gemm256.hip: packed weights (MFMA-fragment order,
// 1-KiB pieces) are the MFMA A operand; activations are the B operand, staged as 8-row x
// 128-byte pieces (whole cache lines) into an XOR-swizzled image. blockIdx.y selects a K
// split (EPI_F32 slabs, summed by the consumer or by lc_splitk_reduce) so that narrow
// outputs still fill the chip.
#include "gemm_common.h"
#include <type_traits>

namespace {

constexpr int GP_BN = 128, GP_BK = 64, GP_WBYTES = 16 * 1024;
// Cache policy of the weight loads: non-temporal. Every weight byte is read exactly once per
// launch while the activation panel is re-read by every workgroup, so the weights should pass
// through L2 without displacing it: measured 41.6 vs 48.5 us (5.65 vs 4.84 TB/s) on a
// 28672 x 4096 matrix at 32 rows, and 4-9 % on the narrower projections.
constexpr int GP_NT = 2;

// DEEP = true: the ring takes (almost) all of a CU's LDS, one workgroup per CU, most loads in
// flight -- best when the GEMM has the GPU to itself. DEEP = false: a ring of three stages
// (<= 72 KiB at BM <= 64) so two workgroups fit on a CU -- for ensembles whose member engines
// run on concurrent streams and fill each other's gaps.
template <int BM, bool DEEP>
struct PanelShape {
  static constexpr int STAGE = GP_WBYTES + BM * 128;
  static constexpr int NSTAGE = !DEEP ? 3 : (BM >= 256 ? 3 : (BM >= 128 ? 4 : (BM >= 64 ? 6 : 7)));
  static constexpr int DEPTH = NSTAGE - 1;                 // K-tiles in flight
  static constexpr int WN = BM >= 256 ? 2 : 4, WM = 8 / WN;  // wave grid
  static constexpr int NT_W = 8 / WN;                      // n-tiles per wave
  static constexpr int MT_W = (BM / 16) / WM;              // m-tiles per wave
  static constexpr int XP = BM / 8 >= 8 ? BM / 64 : 1;     // X pieces a wave issues per K-tile
  static constexpr int P = 2 + XP;                         // DMA instructions per wave per K-tile
  static constexpr bool PIPE = BM >= 256;                  // fragments double-buffered in registers
};

struct GpCtx {
  const bf16_t* __restrict__ Wp;
  const bf16_t* __restrict__ X;
  char* smem;
  int nt0, M, K, KT, kt_lo, nk, wave, lane;
};

template <int BM, bool DEEP, bool UNCHECKED = false>
LC_DEVICE void gp_issue_tile(const GpCtx& c, int t) {
  using S = PanelShape<BM, DEEP>;
  if (!UNCHECKED && t >= c.nk) return;
  char* stage = c.smem + (t % S::NSTAGE) * S::STAGE;
  const int k_tile = c.kt_lo + t;
  {
    const bf16_t* src = c.Wp + (static_cast<size_t>(c.nt0 + c.wave) * c.KT + k_tile * 2) * 512 + c.lane * 8;
    char* dst = stage + (c.wave * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, GP_NT);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src + 512), LC_LDS_PTR(dst + 1024), 16, 0, GP_NT);
  }
  // Activations: pieces of 8 rows x 128 B. The LDS image of a DMA is lane-linear, so the XOR
  // swizzle that keeps the fragment reads off each other's banks is applied on the source
  // side: lane (rr = l>>3, lc = l&7) fetches 16-byte chunk lc^rr of row rr; chunk c of row rr
  // is later read at position c^rr. With fewer than 8 pieces (BM = 32) waves 4-7 repeat the
  // pieces of waves 0-3, so that every wave issues the same number of loads.
  const int rr = c.lane >> 3, lc = c.lane & 7;
#pragma unroll
  for (int q = 0; q < S::XP; ++q) {
    const int piece = (q * 8 + c.wave) % (BM / 8);
    int row = piece * 8 + rr;
    row = row < c.M ? row : c.M - 1;
    const bf16_t* src = c.X + static_cast<size_t>(row) * c.K + k_tile * GP_BK + ((lc ^ rr) << 3);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(stage + GP_WBYTES + piece * 1024), 16, 0, 0);
  }
}

// Byte offset inside the X image of the B fragment (m-tile mt, k-tile kt) for this lane.
LC_DEVICE int gp_x_off(int mt, int kt, int lane) {
  const int r = lane & 15, rr = r & 7;
  return (mt * 2 + (r >> 3)) * 1024 + rr * 128 + (((kt * 4 + (lane >> 4)) ^ rr) << 4);
}

// s_waitcnt vmcnt(n) needs an immediate: one case per number of K-tiles still in flight.
template <int P>
LC_DEVICE void gp_wait_tiles_in_flight(int tiles) {
  switch (tiles) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 1) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 2) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 3) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 4) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 5) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * 6) : "memory"); break;
  }
}

template <int BM, bool DEEP, int EPI>
__global__ __launch_bounds__(512) void gemm_panel_kernel(const bf16_t* __restrict__ Wp,
                                                         const bf16_t* __restrict__ X, void* __restrict__ Y,
                                                         const bf16_t* __restrict__ bias, int M, int N, int K,
                                                         int ldy, int nk_per_split) {
  using S = PanelShape<BM, DEEP>;
  static_assert(S::P * S::DEPTH <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % S::WN, wm = wave / S::WN;
  GpCtx c{Wp, X, smem, static_cast<int>(blockIdx.x) * (GP_BN / 16), M, K, K / 32,
          static_cast<int>(blockIdx.y) * nk_per_split, nk_per_split, wave, lane};

  f32x4 acc[S::NT_W][S::MT_W];
#pragma unroll
  for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (S::PIPE) {
    // BM = 256: a K-tile costs a wave 32 MFMAs, 16 fragment reads and 6 DMA issues, and with two
    // waves per SIMD in step at the barrier those three phases would run one after the other.
    // Here the fragments live in two register sets: while the MFMAs of K-tile t run from one set,
    // the reads of K-tile t+1 fill the other, and the DMA issues sit between them. A stage is
    // free as soon as its fragments are in registers, so all NSTAGE stages hold tiles in flight.
    bf16x8 fa[2][S::NT_W][2], fb[2][S::MT_W][2];
    auto read_frags = [&](int t, int set) __attribute__((always_inline)) {
      const char* stage = smem + (t % S::NSTAGE) * S::STAGE;
      const char* wimg = stage + (wn * S::NT_W * 2) * 1024 + lane * 16;
      const char* ximg = stage + GP_WBYTES;
#pragma unroll
      for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          fa[set][i][kt] = *reinterpret_cast<const bf16x8*>(wimg + (i * 2 + kt) * 1024);
#pragma unroll
      for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          fb[set][j][kt] = *reinterpret_cast<const bf16x8*>(ximg + gp_x_off(wm * S::MT_W + j, kt, lane));
    };
    // MAIN steps (t + NSTAGE < nk) have nothing conditional in them: one basic block, so the
    // scheduler can be told to spread the fragment reads and the DMA issues between the MFMAs.
    auto step = [&](int t, auto set_c, auto main_c) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      constexpr bool MAIN = decltype(main_c)::value;
      // K-tile t+1 has landed (this wave's pieces; the barrier extends that to everybody's), and
      // this wave's reads of K-tile t are complete, so after the barrier stage t % NSTAGE is free.
      const int after = c.nk - 2 - t;    // K-tiles beyond t+1 that exist
      if constexpr (MAIN) {
        gp_wait_tiles_in_flight<S::P>(S::NSTAGE - 2);
      } else {
        if (after >= 0) gp_wait_tiles_in_flight<S::P>(after < S::NSTAGE - 2 ? after : S::NSTAGE - 2);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the fragment reads issued one step ago
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MAIN || after >= 0) read_frags(t + 1, set ^ 1);
      if (MAIN || t + S::NSTAGE < c.nk) gp_issue_tile<BM, DEEP, true>(c, t + S::NSTAGE);
      if constexpr (!MAIN) __builtin_amdgcn_s_setprio(1);   // (a priority change would pin the schedule)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
          for (int i = 0; i < S::NT_W; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][i][kt], fb[set][j][kt], acc[i][j], 0, 0, 0);
      if constexpr (!MAIN) __builtin_amdgcn_s_setprio(0);
      if constexpr (MAIN) {
        // One memory instruction (DMA issue or fragment read; both count as LDS instructions
        // for the scheduler), then one MFMA early on and two towards the end, where the reads
        // that the next step waits for should already be on their way.
        constexpr int N_MFMA = 2 * S::MT_W * S::NT_W, N_MEM = 2 * (S::MT_W + S::NT_W) + S::P;
        constexpr int N_DOUBLE = N_MFMA - N_MEM;     // groups that get two MFMAs
        static_assert(N_DOUBLE >= 0 && N_DOUBLE <= N_MEM, "between one and two MFMAs per memory instruction");
#pragma unroll
        for (int g = 0; g < N_MEM - N_DOUBLE; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x130, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#pragma unroll
        for (int g = 0; g < N_DOUBLE; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x130, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int t = 0; t < S::NSTAGE; ++t) gp_issue_tile<BM, DEEP>(c, t);
    gp_wait_tiles_in_flight<S::P>(c.nk - 1 < S::NSTAGE - 1 ? c.nk - 1 : S::NSTAGE - 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0);
    int t = 0;
    for (; t + 1 + S::NSTAGE < c.nk; t += 2) {
      step(t, I0{}, std::true_type{});
      step(t + 1, I1{}, std::true_type{});
    }
    for (; t < c.nk; t += 2) {
      step(t, I0{}, std::false_type{});
      if (t + 1 < c.nk) step(t + 1, I1{}, std::false_type{});
    }
  } else {
#pragma unroll
  for (int t = 0; t < S::DEPTH; ++t) gp_issue_tile<BM, DEEP>(c, t);

  for (int t = 0; t < c.nk; ++t) {
    const int behind = c.nk - 1 - t;   // K-tiles after t that exist at all
    gp_wait_tiles_in_flight<S::P>(behind < S::DEPTH - 1 ? behind : S::DEPTH - 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    gp_issue_tile<BM, DEEP>(c, t + S::DEPTH);

    const char* stage = smem + (t % S::NSTAGE) * S::STAGE;
    const char* wimg = stage + (wn * S::NT_W * 2) * 1024 + lane * 16;
    const char* ximg = stage + GP_WBYTES;
    bf16x8 a[S::NT_W][2], b[S::MT_W][2];
#pragma unroll
    for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) a[i][kt] = *reinterpret_cast<const bf16x8*>(wimg + (i * 2 + kt) * 1024);
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        b[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + gp_x_off(wm * S::MT_W + j, kt, lane));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
        for (int i = 0; i < S::NT_W; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kt], b[j][kt], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    // every LDS read of this K-tile has been consumed by an MFMA above, so passing the next
    // barrier implies this wave is done with the stage
  }
  }

  const int nbase = blockIdx.x * GP_BN + wn * (S::NT_W * 16), mbase = wm * (S::MT_W * 16);
  const size_t slab = (EPI == EPI_F32) ? static_cast<size_t>(blockIdx.y) * M * ldy : 0;
  const bf16_t* eb = gridDim.y > 1 ? nullptr : bias;   // split-K: the reducer adds the bias
#pragma unroll
  for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[(i + 1) % S::NT_W][j], Y, eb, mbase + j * 16, nbase + i * 16, M, N, ldy,
                        lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Y, eb, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, slab);
      }
    }
}

template <int BM, bool DEEP>
int launch_panel(int epi, const bf16_t* w, const bf16_t* x, void* Y, const bf16_t* b, int M, int N, int K,
                 int n_pad, int ldy, int splits, hipStream_t stream) {
  using S = PanelShape<BM, DEEP>;
  const dim3 grid(n_pad / GP_BN, splits), block(512);
  const int lds = S::NSTAGE * S::STAGE;
  const int nk = (K / GP_BK) / splits;
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_kernel<BM, DEEP, EPI_BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_kernel<BM, DEEP, EPI_F32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_kernel<BM, DEEP, EPI_SWIGLU>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (epi == EPI_BF16)
    gemm_panel_kernel<BM, DEEP, EPI_BF16><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, nk);
  else if (epi == EPI_F32)
    gemm_panel_kernel<BM, DEEP, EPI_F32><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, nk);
  else if (epi == EPI_SWIGLU)
    gemm_panel_kernel<BM, DEEP, EPI_SWIGLU><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, nk);
  else
    return -2;
  return static_cast<int>(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// 160-column workgroups. A matrix of 257-320 column tiles of 128 (Qwen2.5-7B's gate/up projection:
// 37888 columns = 296 tiles) needs two rounds of one-workgroup-per-CU on 256 CUs, and the second
// round keeps 40 CUs busy for as long as the first kept all of them: 59 us where the 271 MB of
// weights would stream in ~45. With ten n-tiles (five gate/up pairs) per workgroup the same matrix
// is 237 workgroups, one round. Up to 64 rows waves 0-4 compute one pair each against all rows
// (MFMA time is far from binding there) and waves 5-7 only stage; every wave stages n-tile `wave`,
// waves 5 and 6 also stage n-tiles 8 and 9, so their K-tiles count two more DMA instructions (P2).
// The last workgroup's n-tiles past the end of the matrix are staged from the last valid one
// and never stored.
template <int BM>
struct Panel5Shape {
  static constexpr int NT = 10, WBYTES = NT * 2048, STAGE = WBYTES + BM * 128;
  static constexpr int NSTAGE = BM >= 256 ? 3 : (BM >= 128 ? 4 : (BM >= 64 ? 5 : 6));   // 156 / 144 / 140 / 144 KiB
  static constexpr int DEPTH = NSTAGE - 1;
  static constexpr int MT_W = BM / 16;
  static constexpr int XP = BM >= 64 ? BM / 64 : 1;         // X pieces a wave stages per K-tile
  static constexpr int P = 2 + XP, P2 = 4 + XP;             // DMA instructions per wave and K-tile
  // From 128 rows on, 32-64 MFMAs per K-tile on each of five waves no longer hide behind the weight
  // stream, so the rows are split over all eight waves instead (one or two row tiles x ten n-tiles
  // per wave): Qwen2.5-7B gate/up at 128 rows 59.9 -> 54.6 us, at 256 rows 104 -> 85-88 us
  static constexpr bool MSPLIT = BM >= 128;
  static constexpr int NT_C = MSPLIT ? NT : 2;              // n-tiles a computing wave covers
  static constexpr int MT_C = MSPLIT ? BM / 128 : MT_W;     // ... and row tiles
};

template <int BM, int EPI>
__global__ __launch_bounds__(512) void gemm_panel5_kernel(const bf16_t* __restrict__ Wp,
                                                          const bf16_t* __restrict__ X, void* __restrict__ Y,
                                                          const bf16_t* __restrict__ bias, int M, int N, int K,
                                                          int ldy, int nt_total) {
  using S = Panel5Shape<BM>;
  static_assert(S::P2 * S::DEPTH <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KT = K / 32, nk = K / GP_BK;
  const int nt0 = static_cast<int>(blockIdx.x) * S::NT;
  const bool two = wave == 5 || wave == 6;
  const int t_a = min(nt0 + wave, nt_total - 1), t_b = min(nt0 + wave + 3, nt_total - 1);
  const bf16_t* wsrc_a = Wp + static_cast<size_t>(t_a) * KT * 512 + lane * 8;
  const bf16_t* wsrc_b = Wp + static_cast<size_t>(t_b) * KT * 512 + lane * 8;
  const int rr = lane >> 3, lc = lane & 7;
  const bf16_t* xsrc[S::XP];
  int xdst[S::XP];
#pragma unroll
  for (int q = 0; q < S::XP; ++q) {
    const int piece = (q * 8 + wave) % (BM / 8);
    int xrow = piece * 8 + rr;
    xrow = xrow < M ? xrow : M - 1;
    xsrc[q] = X + static_cast<size_t>(xrow) * K + ((lc ^ rr) << 3);
    xdst[q] = S::WBYTES + piece * 1024;
  }
  auto issue = [&](int t) {
    if (t >= nk) return;
    char* stage = smem + (t % S::NSTAGE) * S::STAGE;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(wsrc_a + t * 1024), LC_LDS_PTR(stage + wave * 2048), 16, 0, GP_NT);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(wsrc_a + t * 1024 + 512), LC_LDS_PTR(stage + wave * 2048 + 1024),
                                     16, 0, GP_NT);
    if (two) {
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(wsrc_b + t * 1024), LC_LDS_PTR(stage + (wave + 3) * 2048), 16, 0,
                                       GP_NT);
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(wsrc_b + t * 1024 + 512),
                                       LC_LDS_PTR(stage + (wave + 3) * 2048 + 1024), 16, 0, GP_NT);
    }
#pragma unroll
    for (int q = 0; q < S::XP; ++q)
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(xsrc[q] + t * GP_BK), LC_LDS_PTR(stage + xdst[q]), 16, 0, 0);
  };

  f32x4 acc[S::NT_C][S::MT_C];
#pragma unroll
  for (int i = 0; i < S::NT_C; ++i)
#pragma unroll
    for (int j = 0; j < S::MT_C; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool computes = S::MSPLIT || wave < 5;
  const int nt_w = S::MSPLIT ? 0 : wave * 2;                // first n-tile (within the workgroup) of this wave
  const int mt_w = S::MSPLIT ? wave * S::MT_C : 0;          // first row tile
#pragma unroll
  for (int t = 0; t < S::DEPTH; ++t) issue(t);

  for (int t = 0; t < nk; ++t) {
    const int behind = nk - 1 - t;
    const int flight = behind < S::DEPTH - 1 ? behind : S::DEPTH - 1;
    if (two) gp_wait_tiles_in_flight<S::P2>(flight);
    else gp_wait_tiles_in_flight<S::P>(flight);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + S::DEPTH);
    if (computes) {
      const char* stage = smem + (t % S::NSTAGE) * S::STAGE;
      const char* wimg = stage + nt_w * 2048 + lane * 16;
      const char* ximg = stage + S::WBYTES;
      // one 32-deep k-step at a time keeps the fragment registers of a step few
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 a[S::NT_C], b[S::MT_C];
#pragma unroll
        for (int i = 0; i < S::NT_C; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wimg + i * 2048 + kt * 1024);
#pragma unroll
        for (int j = 0; j < S::MT_C; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(ximg + gp_x_off(mt_w + j, kt, lane));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < S::MT_C; ++j)
#pragma unroll
          for (int i = 0; i < S::NT_C; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  if (!computes) return;
  const int nbase = blockIdx.x * (S::NT * 16) + nt_w * 16;
#pragma unroll
  for (int j = 0; j < S::MT_C; ++j)
#pragma unroll
    for (int i = 0; i < S::NT_C; i += 2) {
      const int m0 = (mt_w + j) * 16;
      if constexpr (EPI == EPI_SWIGLU) {
        store_tile<EPI>(acc[i][j], &acc[i + 1][j], Y, bias, m0, nbase + i * 16, M, N, ldy, lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Y, bias, m0, nbase + i * 16, M, N, ldy, lane, 0);
        store_tile<EPI>(acc[i + 1][j], nullptr, Y, bias, m0, nbase + (i + 1) * 16, M, N, ldy, lane, 0);
      }
    }
}

template <int BM>
int launch_panel5(int epi, const bf16_t* w, const bf16_t* x, void* Y, const bf16_t* b, int M, int N, int K,
                  int n_pad, int ldy, hipStream_t stream) {
  using S = Panel5Shape<BM>;
  const int nt_total = n_pad / 16;
  const dim3 grid((nt_total + S::NT - 1) / S::NT), block(512);
  const int lds = S::NSTAGE * S::STAGE;
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel5_kernel<BM, EPI_BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel5_kernel<BM, EPI_SWIGLU>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (epi == EPI_BF16)
    gemm_panel5_kernel<BM, EPI_BF16><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, nt_total);
  else if (epi == EPI_SWIGLU)
    gemm_panel5_kernel<BM, EPI_SWIGLU><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, nt_total);
  else
    return -2;
  return static_cast<int>(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Grouped form for MoE decode steps: blockIdx.z selects an expert; the expert's rows are
// offsets[e] .. offsets[e+1] of the expert-sorted token list (optionally gathered from X through
// row_index), its weights start at Wp_all + e * expert_stride. A decode step routes a few dozen
// tokens to each expert, so every expert's weights are streamed once against one small panel,
// at the same HBM rate as the dense kernel, instead of through 128x128 tiles built for prefill.
// The host passes the total row count, which bounds what any one expert can receive.
template <int BM, int EPI>
__global__ __launch_bounds__(512) void gemm_panel_grouped_kernel(const bf16_t* __restrict__ Wp_all,
                                                                 long long expert_stride,
                                                                 const bf16_t* __restrict__ X,
                                                                 const int32_t* __restrict__ row_index,
                                                                 const int32_t* __restrict__ offsets,
                                                                 void* __restrict__ Y, int N, int K, int ldy) {
  using S = PanelShape<BM, true>;
  static_assert(S::P * S::DEPTH <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e = blockIdx.z;
  // An expert with more than BM rows is covered by further row blocks (blockIdx.y), each of
  // which streams the weights again; the host sizes BM at about twice the mean rows per expert
  // so that this is the exception.
  const int r0 = offsets[e] + static_cast<int>(blockIdx.y) * BM;
  const int M = min(offsets[e + 1] - r0, BM);
  if (M <= 0) return;                       // block-uniform: no (more) rows routed to this expert
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % S::WN, wm = wave / S::WN;
  const bf16_t* Wp = Wp_all + static_cast<size_t>(e) * expert_stride;
  const int KT = K / 32, nk = K / GP_BK, nt0 = static_cast<int>(blockIdx.x) * (GP_BN / 16);

  // Source of this lane's 16-byte chunk in each activation piece it stages (fixed over K).
  const int rr = lane >> 3, lc = lane & 7;
  const bf16_t* xsrc[S::XP];
#pragma unroll
  for (int q = 0; q < S::XP; ++q) {
    const int piece = (q * 8 + wave) % (BM / 8);
    int row = piece * 8 + rr;
    row = row < M ? row : M - 1;
    const int grow = row_index ? row_index[r0 + row] : r0 + row;
    xsrc[q] = X + static_cast<size_t>(grow) * K + ((lc ^ rr) << 3);
  }
  auto issue = [&](int t) {
    if (t >= nk) return;
    char* stage = smem + (t % S::NSTAGE) * S::STAGE;
    const bf16_t* src = Wp + (static_cast<size_t>(nt0 + wave) * KT + t * 2) * 512 + lane * 8;
    char* dst = stage + (wave * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, GP_NT);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src + 512), LC_LDS_PTR(dst + 1024), 16, 0, GP_NT);
#pragma unroll
    for (int q = 0; q < S::XP; ++q) {
      const int piece = (q * 8 + wave) % (BM / 8);
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(xsrc[q] + t * GP_BK),
                                       LC_LDS_PTR(stage + GP_WBYTES + piece * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[S::NT_W][S::MT_W];
#pragma unroll
  for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < S::DEPTH; ++t) issue(t);

  for (int t = 0; t < nk; ++t) {
    const int behind = nk - 1 - t;
    gp_wait_tiles_in_flight<S::P>(behind < S::DEPTH - 1 ? behind : S::DEPTH - 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + S::DEPTH);
    const char* stage = smem + (t % S::NSTAGE) * S::STAGE;
    const char* wimg = stage + (wn * S::NT_W * 2) * 1024 + lane * 16;
    const char* ximg = stage + GP_WBYTES;
    bf16x8 a[S::NT_W][2], b[S::MT_W][2];
#pragma unroll
    for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) a[i][kt] = *reinterpret_cast<const bf16x8*>(wimg + (i * 2 + kt) * 1024);
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        b[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + gp_x_off(wm * S::MT_W + j, kt, lane));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < S::MT_W; ++j)
#pragma unroll
        for (int i = 0; i < S::NT_W; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kt], b[j][kt], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }

  const int nbase = blockIdx.x * GP_BN + wn * (S::NT_W * 16), mbase = wm * (S::MT_W * 16);
  // rows of this expert start at r0 in the output; SwiGLU halves the column count
  bf16_t* Ye = static_cast<bf16_t*>(Y) + static_cast<size_t>(r0) * ldy;
#pragma unroll
  for (int i = 0; i < S::NT_W; ++i)
#pragma unroll
    for (int j = 0; j < S::MT_W; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[(i + 1) % S::NT_W][j], Ye, nullptr, mbase + j * 16, nbase + i * 16, M, N, ldy,
                        lane, 0);
      } else {
        store_tile<EPI>(acc[i][j], nullptr, Ye, nullptr, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      }
    }
}

template <int BM>
int launch_panel_grouped(int epi, const bf16_t* w, long long stride, const bf16_t* x, const int32_t* ri,
                         const int32_t* off, int E, int total_rows, void* Y, int N, int K, int n_pad, int ldy,
                         hipStream_t stream) {
  using S = PanelShape<BM, true>;
  const dim3 grid(n_pad / GP_BN, (total_rows + BM - 1) / BM, E), block(512);
  const int lds = S::NSTAGE * S::STAGE;
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_grouped_kernel<BM, EPI_BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_grouped_kernel<BM, EPI_SWIGLU>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (epi == EPI_BF16)
    gemm_panel_grouped_kernel<BM, EPI_BF16><<<grid, block, lds, stream>>>(w, stride, x, ri, off, Y, N, K, ldy);
  else if (epi == EPI_SWIGLU)
    gemm_panel_grouped_kernel<BM, EPI_SWIGLU><<<grid, block, lds, stream>>>(w, stride, x, ri, off, Y, N, K, ldy);
  else
    return -2;
  return static_cast<int>(hipGetLastError());
}

}  // namespace

// Y[M<=256, N] = X * W^T. Requirements: K % 64 == 0, (K/64) % splits == 0, n_pad % 128 == 0.
// splits > 1 requires epi == EPI_F32 and Y = fp32 workspace [splits, M, ldy]. deep = 0 selects the
// three-stage ring (two workgroups per CU) for M <= 128; deep = 2 the 160-column workgroups
// (splits == 1, bf16 or SwiGLU epilogue; n_pad need not be a multiple of 160).
LC_EXPORT int lc_gemm_panel(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K,
                            int n_pad, int ldy, int epi, int splits, int deep, hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 256 || K % 64 || n_pad % GP_BN || N > n_pad || splits < 1 || (K / GP_BK) % splits) return -1;
  if (splits > 1 && epi != EPI_F32) return -1;
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
#define LC_PANEL(BM_)                                                                            \
  return deep ? launch_panel<BM_, true>(epi, w, x, Y, b, M, N, K, n_pad, ldy, splits, stream)    \
              : launch_panel<BM_, false>(epi, w, x, Y, b, M, N, K, n_pad, ldy, splits, stream)
  // Wide outputs whose 128-column tiles need a second, mostly empty round on the 256 CUs take
  // 160-column workgroups instead when that saves the round (deep = 2 asks for it; the caller
  // knows whether the GEMM has the GPU to itself).
  if (deep == 2) {
    if (splits != 1 || epi == EPI_F32) return -1;
    if (M <= 32) return launch_panel5<32>(epi, w, x, Y, b, M, N, K, n_pad, ldy, stream);
    if (M <= 64) return launch_panel5<64>(epi, w, x, Y, b, M, N, K, n_pad, ldy, stream);
    if (M <= 128) return launch_panel5<128>(epi, w, x, Y, b, M, N, K, n_pad, ldy, stream);
    return launch_panel5<256>(epi, w, x, Y, b, M, N, K, n_pad, ldy, stream);
  }
  if (M <= 32) LC_PANEL(32);
  if (M <= 64) LC_PANEL(64);
  if (M <= 128) LC_PANEL(128);
  return launch_panel<256, true>(epi, w, x, Y, b, M, N, K, n_pad, ldy, splits, stream);
#undef LC_PANEL
}

// Grouped skinny GEMM over E experts (see gemm_panel_grouped_kernel). total_rows <= 256 bounds the
// rows any single expert can receive. Same row/offset conventions as lc_gemm_grouped.
LC_EXPORT int lc_gemm_panel_grouped(const void* Wp_all, long long expert_stride, const void* X,
                                    const void* row_index, const void* offsets, int E, void* Y, int total_rows,
                                    int N, int K, int n_pad, int ldy, int epi, hipStream_t stream) {
  if (total_rows <= 0) return 0;
  if (K % 64 || n_pad % GP_BN || N > n_pad || E < 1) return -1;
  auto w = static_cast<const bf16_t*>(Wp_all);
  auto x = static_cast<const bf16_t*>(X);
  auto ri = static_cast<const int32_t*>(row_index);
  auto off = static_cast<const int32_t*>(offsets);
  // Panel height: about twice the mean rows per expert (the routing of a batch is never even).
  const int want = std::min(total_rows, std::max(32, 2 * ((total_rows + E - 1) / E)));
#define LC_PG(BM_) \
  return launch_panel_grouped<BM_>(epi, w, expert_stride, x, ri, off, E, total_rows, Y, N, K, n_pad, ldy, stream)
  if (want <= 32) LC_PG(32);
  if (want <= 64) LC_PG(64);
  if (want <= 128) LC_PG(128);
  LC_PG(256);
#undef LC_PG
}
