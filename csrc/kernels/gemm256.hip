// Large-M bf16 GEMM, 256(n) x 256(m) x 64(k) block tile, 8 waves, requests that stay in flight
// across barriers (counted s_waitcnt vmcnt, raw s_barrier).
//
// Same operand conventions as gemm.hip: packed weights (MFMA-fragment order) are the MFMA A
// operand, activations the B operand. Wave (wn = w&3, wm = w>>2) owns 64 n x 128 m of the tile =
// 4 x 8 accumulator fragments (128 VGPRs). Each phase of a K-tile is
//     R (fragment reads, requests, waits) | barrier | M (16 MFMAs) | barrier
// and waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in M while its
// partner is in R. The request queue is never drained inside the loop.
//
// Two main loops (SCHED):
//   2 (default)  weights straight from global memory into registers (they are stored as fragments),
//                one K-tile ahead; activations through three 32-KiB LDS stages filled by LDS-DMA as
//                whole 128-byte rows in an XOR-swizzled image; two phases per K-tile, one per k-step.
//                Described at the loop.
//   0            both operands through LDS (LC_GEMM256_SCHED=0, kept for comparison runs): a K-tile
//                is four 16-KiB "halves" (W0, W1, X0, X1), each wave moves one 2-KiB slice of a
//                half with two LDS-DMA instructions, one half per phase, running ahead of the math:
//                    P1(t): X1(t+1)   P2(t): W0(t+2)   P3(t): W1(t+2)   P4(t): X0(t+2)
//                with the phases P1 (n-lo, m-lo)  P2 (n-hi, m-lo)  P3 (n-hi, m-hi)  P4 (n-lo, m-hi),
//                so all 8 weight fragments are read in P1 and the activation fragments in P1 and P2.
//                A region is re-staged only after a barrier that follows the retirement
//                (lgkmcnt(0)) of its last readers in BOTH groups; in P4's R part a wave waits with
//                vmcnt(6): everything but the three newest halves -- i.e. all of K-tile t+1 -- has
//                landed; two barriers later the leading group reads it.
#include <stdlib.h>

#include "gemm_common.h"

namespace {

constexpr int G2_BN = 256, G2_BM = 256, G2_BK = 64;
constexpr int G2_STAGE = 64 * 1024;   // W image 32 KiB + X image 32 KiB
constexpr int G2_XOFF = 32 * 1024;
constexpr int G2_BAND = 4;

struct G2Ctx {
  const bf16_t* __restrict__ Wp;
  const bf16_t* __restrict__ X;
  char* smem;
  int nt0, m0, M, K, KT, nk, wave, lane;
  int kt_lo = 0;   // first K-tile of this block's K split
};

// Issues this wave's two 1-KiB pieces of half `which` (0:W0 1:W1 2:X0 3:X1) of K-tile kt_tile.
LC_DEVICE void g2_issue_half(const G2Ctx& c, int k_tile, int which) {
  if (k_tile >= c.nk) return;
  char* stage = c.smem + (k_tile & 1) * G2_STAGE;
  const int kt0 = (c.kt_lo + k_tile) * 2;
  const int tile = (which & 1) * 8 + c.wave;  // 16-row tile inside the 256-row operand panel
  if (which < 2) {
    const bf16_t* src = c.Wp + (static_cast<size_t>(c.nt0 + tile) * c.KT + kt0) * 512 + c.lane * 8;
    char* dst = stage + (tile * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src + 512), LC_LDS_PTR(dst + 1024), 16, 0, 0);
  } else {
    // Two pieces of 8 rows x 128 B (whole cache lines: the K-tile's full width). The LDS image
    // of a DMA is lane-linear, so the XOR swizzle that spreads the fragment reads over the banks
    // is applied on the source side: lane (rr = l>>3, lc = l&7) fetches 16-byte chunk lc^rr of
    // row rr; chunk c of row rr is then read at position c^rr (g2_x_off).
    const int rr = c.lane >> 3, lc = c.lane & 7;
    int row0 = c.m0 + tile * 16 + rr, row1 = row0 + 8;
    row0 = row0 < c.M ? row0 : c.M - 1;
    row1 = row1 < c.M ? row1 : c.M - 1;
    const int koff = kt0 * 32 + ((lc ^ rr) << 3);
    char* dst = stage + G2_XOFF + (tile * 2) * 1024;
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(c.X + static_cast<size_t>(row0) * c.K + koff), LC_LDS_PTR(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(c.X + static_cast<size_t>(row1) * c.K + koff), LC_LDS_PTR(dst + 1024), 16, 0, 0);
  }
}

// Byte offset, inside one m-tile's 2-KiB image, of this lane's B fragment for k-tile kt.
LC_DEVICE int g2_x_off(int kt, int lane) {
  const int r = lane & 15, rr = r & 7;
  return (r >> 3) * 1024 + rr * 128 + (((kt * 4 + (lane >> 4)) ^ rr) << 4);
}

// ---- SCHED 2: phases cut by (k-step, m-half), weights in registers ----
// LDS: three X stages of 32 KiB (rows 0-127 / 128-255 of a K-tile in two 16-KiB halves); the
// allocation is 128 KiB because the epilogue stages the output tile through it.
constexpr int G2B_XSTAGE = 32 * 1024;
constexpr int G2B_LDS = 128 * 1024;

// 32 MFMAs of one k-step: all four n-fragments against all eight m-fragments.
#define G2B_MFMA32(AFR, BFR)                                                                          \
  do {                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int jh = 0; jh < 8; jh += 4)                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
        _Pragma("unroll") for (int j = jh; j < jh + 4; ++j)                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AFR[i], BFR[j], acc[i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                    \
  } while (0)

#define G2_MFMA_QUADRANT(NLO, BFR, MLO)                                                              \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int kt = 0; kt < 2; ++kt)                                                 \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
          acc[(NLO) + i][(MLO) + j] =                                                                \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(NLO) + i][kt], BFR[j][kt], acc[(NLO) + i][(MLO) + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)

#define G2_BARRIER()                         \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)

// GROUPED: the rows are the expert-sorted token list of a MoE layer; expert e owns rows
// offsets[e] .. offsets[e+1] and the weights at Wp + e * expert_stride. m-tiles are numbered
// through all experts (tiles_m is an upper bound: total_rows / 256 + E); a block walks the E+1
// offsets to find its expert and leaves if its m-tile is past the last one in use.
template <int EPI, bool GROUPED = false, int SCHED = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(const bf16_t* __restrict__ Wp,
                                                      const bf16_t* __restrict__ X, void* __restrict__ Y,
                                                      const bf16_t* __restrict__ bias, int M, int N, int K,
                                                      int ldy, int tiles_m, int tiles_n,
                                                      long long expert_stride = 0,
                                                      const int32_t* __restrict__ offsets = nullptr, int E = 0,
                                                      int nk_per_split = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 3, wm = wave >> 2;

  // XCD-aware remap (blocks b and b+8 share an XCD): each XCD gets a contiguous run of tiles.
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  // Tiles are ordered in bands of G2_BAND n-tiles, n fastest inside a band: the ~32 tiles an
  // XCD runs at once then form an 8(m) x 4(n) patch that shares 8 activation and 4 weight
  // panels through that XCD's L2 instead of streaming 32 different activation panels.
  const int band = tile / (G2_BAND * tiles_m), idx = tile % (G2_BAND * tiles_m);
  const int bw = min(G2_BAND, tiles_n - band * G2_BAND);
  const int tn = band * G2_BAND + idx % bw, tm = idx / bw;

  int m0 = tm * G2_BM;
  if constexpr (GROUPED) {
    int gm = tm, e = 0, lo = 0, hi = 0;
    bool found = false;
    for (e = 0; e < E; ++e) {
      lo = offsets[e];
      hi = offsets[e + 1];
      const int nt = (hi - lo + G2_BM - 1) / G2_BM;
      if (gm < nt) {
        found = true;
        break;
      }
      gm -= nt;
    }
    if (!found) return;                      // block-uniform, before any barrier
    Wp += static_cast<size_t>(e) * expert_stride;
    m0 = lo + gm * G2_BM;
    M = hi;                                  // rows of this expert end here (absolute row index)
  }
  G2Ctx c{Wp, X, smem, tn * (G2_BN / 16), m0, M, K, K / 32, K / G2_BK, wave, lane};
  if (nk_per_split > 0) {   // split-K (blockIdx.y): fp32 slabs, summed by the consumer or by lc_splitk_reduce
    c.nk = nk_per_split;
    c.kt_lo = static_cast<int>(blockIdx.y) * nk_per_split;
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (SCHED == 2) {
    // SCHED 2. A K-tile is consumed in two phases of 32 MFMAs, one k-step (32 deep) of all four
    // n-fragments against all eight m-fragments each (P1 = k0, P2 = k1), so every accumulator
    // receives k0 before k1 of each K-tile (results equal SCHED 0 bit for bit), a phase reads
    // eight activation fragments from LDS, and there are four barrier-to-barrier slots per K-tile
    // instead of eight (each slot carries ~75 cycles of barrier and hand-over cost on top of its
    // MFMAs). The WEIGHT fragments never pass through LDS: the packed weights are stored as
    // ready-made MFMA fragments, so a wave loads its own four n-fragments of one k-step per phase
    // straight from global memory into registers (one coalesced KiB per instruction), one K-tile
    // ahead, double-buffered. LDS carries the activations only (three 32-KiB stages; the two
    // halves of X of K-tile t+2 are requested in P1 and P2 of K-tile t into the stage K-tile t-1
    // used): 128 KiB of fragment reads and 32 KiB of LDS-DMA writes per K-tile and CU instead of
    // 192 + 64. That matters because the LDS traffic of one wave slows the MFMAs of its SIMD
    // partner (barrier-to-barrier slots measured 393-704 time units where their 16 MFMAs need
    // ~330). Every request is issued unconditionally (indices are clamped
    // at the end of K) so that the number of memory operations between a request and its first
    // use is the same on every path, and all waits are counted by hand: left to the compiler,
    // every wait for a register load becomes vmcnt(0), because it treats a pending
    // global_load ... lds as a FLAT access whose return order is unknown.
    bf16x8 A0[2][4], A1[2][4], bX[8];
    const int xoff0 = g2_x_off(0, lane), xoff1 = g2_x_off(1, lane);
    const bf16_t* wbase = c.Wp + (static_cast<size_t>(c.nt0 + wn * 4) * c.KT + c.kt_lo * 2) * 512 + lane * 8;
    const size_t wnt = static_cast<size_t>(c.KT) * 512;      // elements between consecutive n-tiles
    // register loads from inline assembly: the compiler neither waits for them nor moves them
    auto wfrag = [&](int i, int kidx) -> bf16x8 {            // fragment of n-tile i, 32-deep k index kidx
      bf16x8 v;
      const bf16_t* ptr = wbase + i * wnt + static_cast<size_t>(kidx) * 512;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(ptr) : "memory");
      return v;
    };
    auto xissue = [&](int k_tile, int half, int xs) {        // X rows [128 half, +128) of K-tile k_tile
      const int kk = k_tile < c.nk ? k_tile : c.nk - 1;
      const int kt0 = (c.kt_lo + kk) * 2;
      const int tile = half * 8 + wave;
      const int rr = lane >> 3, lc = lane & 7;
      int row0 = c.m0 + tile * 16 + rr, row1 = row0 + 8;
      row0 = row0 < c.M ? row0 : c.M - 1;
      row1 = row1 < c.M ? row1 : c.M - 1;
      const int koff = kt0 * 32 + ((lc ^ rr) << 3);
      char* dst = smem + xs * G2B_XSTAGE + (tile * 2) * 1024;
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(c.X + static_cast<size_t>(row0) * c.K + koff), LC_LDS_PTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(c.X + static_cast<size_t>(row1) * c.K + koff), LC_LDS_PTR(dst + 1024), 16, 0, 0);
    };
    xissue(0, 0, 0);
    xissue(0, 1, 0);
    xissue(1, 0, 1);
    xissue(1, 1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A0[0][i] = wfrag(i, 0);
      A1[0][i] = wfrag(i, 1);
    }
    // Everything: the first K-tile consumes the weight fragments requested last. With X(1) in
    // front of them, the counted waits of the loop hold from its first K-tile on.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) G2_BARRIER();
    int xs = 0;
#define G2C_KTILE(T, CUR, NXT)                                                                          \
    do {                                                                                                \
      const int t_ = (T);                                                                               \
      const char* ximg = smem + xs * G2B_XSTAGE + (wm * 8 * 2) * 1024;                                  \
      const int xs2 = xs == 0 ? 2 : xs - 1;                                                             \
      const int kn = (t_ + 1 < c.nk ? t_ + 1 : c.nk - 1) * 2;                                           \
      /* P1 (k0, all m) */                                                                              \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                     \
        bX[j] = *reinterpret_cast<const bf16x8*>(ximg + j * 2048 + xoff0);                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) A0[NXT][i] = wfrag(i, kn);                          \
      xissue(t_ + 2, 0, xs2);                                                                           \
      /* A0[CUR]: requested at the start of P1 of K-tile t-1, 2 + 6 + 6 = 14 operations ago */          \
      asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");                                      \
      G2_BARRIER();                                                                                     \
      G2B_MFMA32(A0[CUR], bX);                                                                          \
      G2_BARRIER();                                                                                     \
      /* P2 (k1, all m) */                                                                              \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                     \
        bX[j] = *reinterpret_cast<const bf16x8*>(ximg + j * 2048 + xoff1);                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) A1[NXT][i] = wfrag(i, kn + 1);                      \
      xissue(t_ + 2, 1, xs2);                                                                           \
      /* X(t+1): second half requested in P2 of K-tile t-1: 4 + 6 + 4 = 14 operations ago (the  */      \
      /* two requests above included); A1[CUR]: requested in P2 of K-tile t-1 as well            */      \
      asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");                                      \
      G2_BARRIER();                                                                                     \
      G2B_MFMA32(A1[CUR], bX);                                                                          \
      G2_BARRIER();                                                                                     \
      xs = xs == 2 ? 0 : xs + 1;                                                                        \
    } while (0)
    int t = 0;
    for (; t + 1 < c.nk; t += 2) {
      G2C_KTILE(t, 0, 1);
      G2C_KTILE(t + 1, 1, 0);
    }
    if (t < c.nk) G2C_KTILE(t, 0, 1);                       // odd number of K-tiles
#undef G2C_KTILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clamped requests of the last K-tiles
    // The last K-tile's register loads are never consumed. Their destination registers must still
    // belong to them until the wait above: a register the compiler considered free would be handed
    // to another value and then overwritten when the load lands. These empty statements are that
    // claim (a "use" after the wait).
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("" ::"v"(A0[0][i]), "v"(A0[1][i]), "v"(A1[0][i]), "v"(A1[1][i]));
  } else {
  bf16x8 a[4][2], b0[4][2], b1[4][2];
  const int xoff[2] = {g2_x_off(0, lane), g2_x_off(1, lane)};

  // ---- prologue: K-tile 0 completely, plus the three halves of K-tile 1 that steady state
  //      expects to be in flight at the top of a K-tile ----
#pragma unroll
  for (int h = 0; h < 4; ++h) g2_issue_half(c, 0, h);
  g2_issue_half(c, 1, 0);
  g2_issue_half(c, 1, 1);
  g2_issue_half(c, 1, 2);
  if (c.nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // Stagger: waves 4-7 (the SIMD partners of waves 0-3) run half a phase behind, so one
  // partner's LDS reads / DMA issue ("R") always overlap the other's 16 MFMAs ("M").
  if (wave >= 4) G2_BARRIER();

  for (int t = 0; t < c.nk; ++t) {
    const char* wimg = smem + (t & 1) * G2_STAGE + (wn * 4 * 2) * 1024 + lane * 16;
    const char* ximg = smem + (t & 1) * G2_STAGE + G2_XOFF + (wm * 8 * 2) * 1024;
    // ---------------- P1 ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) a[i][kt] = *reinterpret_cast<const bf16x8*>(wimg + (i * 2 + kt) * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) b0[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + j * 2048 + xoff[kt]);
    g2_issue_half(c, t + 1, 3);
    // Reads are retired BEFORE the barrier: the W regions are re-staged by the other wave
    // group in the very next slot.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G2_BARRIER();
    G2_MFMA_QUADRANT(0, b0, 0);
    G2_BARRIER();
    // ---------------- P2 ----------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        b1[j][kt] = *reinterpret_cast<const bf16x8*>(ximg + (4 + j) * 2048 + xoff[kt]);
    g2_issue_half(c, t + 2, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G2_BARRIER();
    G2_MFMA_QUADRANT(2, b0, 0);
    G2_BARRIER();
    // ---------------- P3 ----------------
    g2_issue_half(c, t + 2, 1);
    G2_BARRIER();
    G2_MFMA_QUADRANT(2, b1, 4);
    G2_BARRIER();
    // ---------------- P4 ----------------
    g2_issue_half(c, t + 2, 2);
    // The wait sits before P4's FIRST barrier: the lagging group's wait must be over one
    // barrier before the leading group starts reading K-tile t+1.
    if (t + 2 < c.nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_BARRIER();
    G2_MFMA_QUADRANT(0, b1, 4);
    G2_BARRIER();
  }
  }
  if (wave < 4) G2_BARRIER();  // balance the stagger barrier

  const int nbase = tn * G2_BN + wn * 64, mbase = m0 + wm * 128;
  // Epilogue through LDS. Stored straight from the accumulator layout, a wave's store
  // instruction would touch 16 rows with 32 bytes each; the stores of a tile then cost as much
  // as ten K-tiles of math. Instead every wave transposes its sub-tile through a private LDS
  // region (the ring is idle by now) and writes whole rows of it: 16 bytes per lane, 8 (plain)
  // or 16 (SwiGLU) rows of 128 or 64 contiguous bytes per instruction.
  if constexpr (EPI != EPI_F32) {
    if ((ldy & 7) == 0 && (tn + 1) * G2_BN <= N) {   // uniform: the whole block tile is inside Y
      G2_BARRIER();                                    // every wave is past its last ring read
      // Staged rows are 128 bytes: a wave's 64 columns, or with SwiGLU the 32 + 32 output
      // columns of two waves that are neighbours in n (they share a region and split its rows).
      constexpr int ROWB = 128, CH = 8;
      char* mine = smem + (EPI == EPI_SWIGLU ? (wm * 2 + (wn >> 1)) : wave) * (128 * ROWB);
      const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = j * 16 + r16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
            chunk = (wn & 1) * 4 + (i >> 1) * 2 + (g >> 1);
          } else {
            const int nb = nbase + i * 16 + 4 * g;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + (bias ? bf16_to_f32(bias[nb + r]) : 0.f);
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            chunk = i * 2 + (g >> 1);
          }
          // chunk c of row m lives at position c ^ (m mod CH): column accesses spread over banks
          *reinterpret_cast<u32x2*>(mine + m * ROWB + ((chunk ^ (m & (CH - 1))) << 4) + (g & 1) * 8) = o;
        }
      }
      constexpr int RPI = 64 / CH;                         // rows per store instruction
      const int c16 = lane & (CH - 1), rsub = lane / CH;
      const int ncol = (EPI == EPI_SWIGLU ? tn * (G2_BN / 2) + (wn >> 1) * 64 : nbase) + c16 * 8;
      constexpr int PASSES = EPI == EPI_SWIGLU ? 8 : 16;
      const int p0 = EPI == EPI_SWIGLU ? (wn & 1) * 8 : 0;
      if constexpr (EPI == EPI_SWIGLU) G2_BARRIER();       // the partner wave's half rows are in place
#pragma unroll
      for (int pp = 0; pp < PASSES; ++pp) {
        const int p = p0 + pp;
        const int m = p * RPI + rsub;
        const u32x4 v = *reinterpret_cast<const u32x4*>(mine + m * ROWB + ((c16 ^ (m & (CH - 1))) << 4));
        if (mbase + m < M)
          *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(Y) + static_cast<size_t>(mbase + m) * ldy + ncol) = v;
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (EPI == EPI_SWIGLU) {
        if (i & 1) continue;
        store_tile<EPI>(acc[i][j], &acc[i + 1][j], Y, bias, mbase + j * 16, nbase + i * 16, M, N, ldy, lane, 0);
      } else {
        const size_t slab = (EPI == EPI_F32) ? static_cast<size_t>(blockIdx.y) * M * ldy : 0;
        store_tile<EPI>(acc[i][j], nullptr, Y, gridDim.y > 1 ? nullptr : bias, mbase + j * 16, nbase + i * 16, M, N,
                        ldy, lane, slab);
      }
    }
}


}  // namespace

// Main-loop form: 2 = weights in registers, activations through LDS (default); LC_GEMM256_SCHED=0
// selects the older form with both operands through LDS (kept for comparison runs).
static int g2_sched() {
  static const int v = getenv("LC_GEMM256_SCHED") ? atoi(getenv("LC_GEMM256_SCHED")) : 2;
  return v;
}

// Y = X * W^T with the 256x256 pipelined tile. Requirements: K % 64 == 0, n_pad % 256 == 0.
LC_EXPORT int lc_gemm256_splitk(const void* Wp, const void* X, void* Y, int M, int N, int K, int n_pad, int ldy,
                                int splits, hipStream_t stream);

LC_EXPORT int lc_gemm256(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K,
                         int n_pad, int ldy, int epi, hipStream_t stream) {
  if (M <= 0) return 0;
  if (K % 64 || n_pad % 256 || N > n_pad) return -1;
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = n_pad / G2_BN;
  const dim3 grid(tiles_m * tiles_n), block(512);
  const size_t lds = 2 * G2_STAGE;
  auto w = static_cast<const bf16_t*>(Wp);
  auto x = static_cast<const bf16_t*>(X);
  auto b = static_cast<const bf16_t*>(bias);
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_F32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_SWIGLU>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  });
  if (g2_sched() == 2) {
    static LcDeviceOnce s2_once;
    s2_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_BF16, false, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_F32, false, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_SWIGLU, false, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
    });
    if (epi == EPI_BF16)
      gemm256_kernel<EPI_BF16, false, 2><<<grid, block, G2B_LDS, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
    else if (epi == EPI_F32)
      gemm256_kernel<EPI_F32, false, 2><<<grid, block, G2B_LDS, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
    else if (epi == EPI_SWIGLU)
      gemm256_kernel<EPI_SWIGLU, false, 2><<<grid, block, G2B_LDS, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
    else
      return -2;
    LC_RETURN_LAST_ERROR();
  }
  if (epi == EPI_BF16)
    gemm256_kernel<EPI_BF16><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else if (epi == EPI_F32)
    gemm256_kernel<EPI_F32><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else if (epi == EPI_SWIGLU)
    gemm256_kernel<EPI_SWIGLU><<<grid, block, lds, stream>>>(w, x, Y, b, M, N, K, ldy, tiles_m, tiles_n);
  else
    return -2;
  LC_RETURN_LAST_ERROR();
}

// Grouped form over E experts for prefill-sized MoE batches: X [total_rows, K] and Y are in
// expert-sorted row order, offsets [E+1] delimit the experts' rows. Requirements as lc_gemm256.
LC_EXPORT int lc_gemm256_grouped(const void* Wp_all, long long expert_stride, const void* X, const void* offsets,
                                 int E, void* Y, int total_rows, int N, int K, int n_pad, int ldy, int epi,
                                 hipStream_t stream) {
  if (total_rows <= 0) return 0;
  if (K % 64 || n_pad % 256 || N > n_pad || E < 1) return -1;
  const int tiles_m = (total_rows + G2_BM - 1) / G2_BM + E, tiles_n = n_pad / G2_BN;
  const dim3 grid(tiles_m * tiles_n), block(512);
  const size_t lds = 2 * G2_STAGE;
  auto w = static_cast<const bf16_t*>(Wp_all);
  auto x = static_cast<const bf16_t*>(X);
  auto off = static_cast<const int32_t*>(offsets);
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_BF16, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_SWIGLU, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  });
  if (g2_sched() == 2) {
    static LcDeviceOnce s2_once;
    s2_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_BF16, true, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_SWIGLU, true, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
    });
    if (epi == EPI_BF16)
      gemm256_kernel<EPI_BF16, true, 2><<<grid, block, G2B_LDS, stream>>>(w, x, Y, nullptr, total_rows, N, K, ldy,
                                                                          tiles_m, tiles_n, expert_stride, off, E);
    else if (epi == EPI_SWIGLU)
      gemm256_kernel<EPI_SWIGLU, true, 2><<<grid, block, G2B_LDS, stream>>>(w, x, Y, nullptr, total_rows, N, K, ldy,
                                                                            tiles_m, tiles_n, expert_stride, off, E);
    else
      return -2;
    LC_RETURN_LAST_ERROR();
  }
  if (epi == EPI_BF16)
    gemm256_kernel<EPI_BF16, true><<<grid, block, lds, stream>>>(w, x, Y, nullptr, total_rows, N, K, ldy, tiles_m,
                                                                 tiles_n, expert_stride, off, E);
  else if (epi == EPI_SWIGLU)
    gemm256_kernel<EPI_SWIGLU, true><<<grid, block, lds, stream>>>(w, x, Y, nullptr, total_rows, N, K, ldy, tiles_m,
                                                                   tiles_n, expert_stride, off, E);
  else
    return -2;
  LC_RETURN_LAST_ERROR();
}

// Split-K form for a few hundred to a few thousand rows against a narrow output, where the
// 256x256 tiles alone would leave most CUs idle: Y = fp32 slabs [splits, M, ldy] (no bias).
LC_EXPORT int lc_gemm256_splitk(const void* Wp, const void* X, void* Y, int M, int N, int K, int n_pad, int ldy,
                                int splits, hipStream_t stream) {
  if (M <= 0) return 0;
  if (K % 64 || n_pad % 256 || N > n_pad || splits < 1 || (K / G2_BK) % splits) return -1;
  const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = n_pad / G2_BN;
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  const size_t lds = 2 * G2_STAGE;
  static LcDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_F32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  });
  if (g2_sched() == 2) {
    static LcDeviceOnce s2_once;
    s2_once.run([&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI_F32, false, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, G2B_LDS);
    });
    gemm256_kernel<EPI_F32, false, 2><<<grid, block, G2B_LDS, stream>>>(
        static_cast<const bf16_t*>(Wp), static_cast<const bf16_t*>(X), Y, nullptr, M, N, K, ldy, tiles_m, tiles_n, 0,
        nullptr, 0, (K / G2_BK) / splits);
    LC_RETURN_LAST_ERROR();
  }
  gemm256_kernel<EPI_F32><<<grid, block, lds, stream>>>(static_cast<const bf16_t*>(Wp), static_cast<const bf16_t*>(X), Y,
                                                        nullptr, M, N, K, ldy, tiles_m, tiles_n, 0, nullptr, 0,
                                                        (K / G2_BK) / splits);
  LC_RETURN_LAST_ERROR();
}
