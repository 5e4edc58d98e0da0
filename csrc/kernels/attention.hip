// Paged attention for gfx950: prefill (causal, GQA, optional sliding window), decode
// (one query token per sequence, split-KV), shared-prefix decode (all chains of a
// sequence group attend to their common prompt blocks in ONE pass, 32 query rows per
// wave on the MFMA), and the log-sum-exp merge of partial results.
//
// KV cache layout (per layer):  K  [num_blocks][Hkv][BS][Dh]   (token-major rows)
//                               Vt [num_blocks][Hkv][Dh][BS]   (V stored transposed)
// With V transposed, both MFMA operands of both products are k-contiguous 16-byte
// loads straight from HBM/L2 into VGPRs: no LDS, no transposing reads.
//
// One wave owns 32 query rows. Per 32-key tile it computes S^T = K * Q^T with
// v_mfma_f32_32x32x16_bf16 (A = K rows, B = Q^T), so the query row lives on the LANE
// (column) and a lane holds 16 of its 32 scores: softmax needs 15 in-lane ops plus one
// cross-half shuffle. The 32x32 score tile then feeds O^T += Vt * P^T directly as the
// B operand of the next MFMA (accumulator registers 8s..8s+7, packed to bf16, are
// k-step s). MFMA row i of S^T is mapped to key k0 + pi(i), pi = swap bits 2 and 3,
// which makes the keys of k-step s / lane-half h the contiguous run 16s+8h .. +7 that
// the Vt fragment load wants.
//
// Nothing here has a counterpart in the reference (remote model,
// /root/reference/src/main.rs:82-86); shared-prefix decode exists because the
// reference's evaluate prompts and K-chain self-consistency share prompt prefixes.
#include "common.h"

namespace {

constexpr float kNegBig = -1.0e30f;

struct TileCtx {
  const bf16_t* __restrict__ Kc;
  const bf16_t* __restrict__ Vt;
  const int32_t* __restrict__ bt;  // this sequence's block table
  int Hkv, kvh, lbs, BS;           // lbs = log2(BS)
  float scale2;                    // softmax scale * log2(e)
};

// Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2. Workgroups
// come in groups of `gx` that read the same K/V tiles; this returns the logical index
// (group * gx + member) for a physical id such that the members of a group are all on one XCD
// while consecutive GROUPS still go round-robin over the XCDs. Spreading by groups matters: the
// work per group is uneven (ragged prompts, empty group slots), and a mapping that hands each XCD
// a contiguous run of groups would put a whole long sequence on one eighth of the chip.
LC_DEVICE int xcd_group_remap(int id, int gx, int n_groups) {
  const int full = (n_groups & ~7) * gx;
  if (id >= full) return id;
  const int x = id & 7, k = id >> 3;
  return (x + 8 * (k / gx)) * gx + k % gx;
}

template <int DH>
struct AttnState {
  f32x16 o[DH / 32];  // O^T blocks: row d = blk*32 + (reg&3) + 8*(reg>>2) + 4*h, col q = lane&31
  float m, l;         // running max (log2 domain) and this lane-half's partial sum
};

// One 32-key tile for the wave's 32 query rows, operands already in registers: S^T = K * Q^T,
// online softmax, O^T += Vt * P^T. Row visibility: lo_q <= key < hi_q.
template <int DH>
LC_DEVICE void process_tile(AttnState<DH>& st, const bf16x8 (&qf)[DH / 16], const bf16x8 (&kf)[DH / 16],
                            const bf16x8 (&vf)[DH / 32][2], int k0, int lo_q, int hi_q, float scale2, int lane) {
  const int h = lane >> 5;
  f32x16 sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < DH / 16; ++kk)
    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], sc, 0, 0, 0);

  // ---- online softmax (row q = lane&31; this lane holds keys k0 + 16*(reg>>3) + 8h + (reg&7)) ----
  // Only tiles that straddle a row's visibility bounds (the causal diagonal, a sliding-window
  // edge, the ragged tail) need per-key masks; the test is wave-uniform.
  const bool interior = __all(k0 >= lo_q && k0 + 32 <= hi_q);
  float tmax = kNegBig;
  if (interior) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      sc[g] *= scale2;
      tmax = fmaxf(tmax, sc[g]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int kidx = k0 + 16 * (g >> 3) + 8 * h + (g & 7);
      sc[g] = (kidx >= lo_q && kidx < hi_q) ? sc[g] * scale2 : kNegBig;
      tmax = fmaxf(tmax, sc[g]);
    }
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  const float m_new = fmaxf(st.m, tmax);
  // Rescaling the accumulators costs Dh/2 multiplies per lane; once the running maxima have
  // settled (no lane's maximum moved) it is skipped for the whole wave.
  if (__any(m_new > st.m)) {
    const float alpha = fast_exp2(st.m - m_new);
    st.l *= alpha;
#pragma unroll
    for (int b = 0; b < DH / 32; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) st.o[b][g] *= alpha;
    st.m = m_new;
  }
  float psum = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    // masked scores sit at kNegBig: 2^(kNegBig - m) is 0 unless the whole row is still masked
    // (m == kNegBig), which the explicit test covers
    const float p = sc[g] > 0.5f * kNegBig ? fast_exp2(sc[g] - st.m) : 0.f;
    sc[g] = p;
    psum += p;
  }
  st.l += psum;

  // ---- O^T += Vt * P^T ----
  bf16x8 pf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 packed;
#pragma unroll
    for (int j = 0; j < 4; ++j) packed[j] = pack_bf16x2(sc[8 * s + 2 * j], sc[8 * s + 2 * j + 1]);
    pf[s] = __builtin_bit_cast(bf16x8, packed);
  }
#pragma unroll
  for (int b = 0; b < DH / 32; ++b)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      st.o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[b][s], pf[s], st.o[b], 0, 0, 0);
}

// Processes keys [kv_lo, kv_hi) for the wave's 32 rows. Row visibility: lo_q <= key < hi_q.
template <int DH>
LC_DEVICE void attend_range(AttnState<DH>& st, const bf16x8 (&qf)[DH / 16], const TileCtx& c, int kv_lo,
                            int kv_hi, int lo_q, int hi_q, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int pi_r = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);
  const size_t kblk_stride = static_cast<size_t>(c.Hkv) * c.BS * DH;
  // Block-table entries live in a VGPR (lane i holds entry blk_base + i) and are fetched by
  // lane shuffle: looking them up in memory per tile would put a second dependent global
  // round trip in front of every K/V load. Reloaded every 64 blocks.
  const int last_blk = (kv_hi - 1) >> c.lbs;
  int blk_base = (kv_lo & ~31) >> c.lbs;
  int btv = c.bt[min(blk_base + lane, last_blk)];
  for (int k0 = kv_lo & ~31; k0 < kv_hi; k0 += 32) {
    if (((k0 + 31) >> c.lbs) - blk_base > 63) {
      blk_base = k0 >> c.lbs;
      btv = c.bt[min(blk_base + lane, last_blk)];
    }
    // ---- S^T = K * Q^T ----
    int key = k0 + pi_r;
    key = key < kv_hi ? key : kv_hi - 1;
    const bf16_t* kp = c.Kc + static_cast<size_t>(__shfl(btv, (key >> c.lbs) - blk_base, 64)) * kblk_stride +
                       (static_cast<size_t>(c.kvh) * c.BS + (key & (c.BS - 1))) * DH + 8 * h;
    bf16x8 kf[DH / 16];
#pragma unroll
    for (int kk = 0; kk < DH / 16; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(kp + 16 * kk);
    // Vt fragments for this tile: chunk (s, h) = keys k0 + 16s + 8h .. +7
    bf16x8 vf[DH / 32][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int key8 = k0 + 16 * s + 8 * h;
      const int last8 = (kv_hi - 1) & ~7;
      key8 = key8 < kv_hi ? key8 : last8;
      const bf16_t* vp = c.Vt + static_cast<size_t>(__shfl(btv, (key8 >> c.lbs) - blk_base, 64)) * kblk_stride +
                         (static_cast<size_t>(c.kvh) * DH + r) * c.BS + (key8 & (c.BS - 1));
#pragma unroll
      for (int b = 0; b < DH / 32; ++b)
        vf[b][s] = *reinterpret_cast<const bf16x8*>(vp + static_cast<size_t>(b) * 32 * c.BS);
    }
    process_tile<DH>(st, qf, kf, vf, k0, lo_q, hi_q, c.scale2, lane);
  }
}

template <int DH>
LC_DEVICE void init_state(AttnState<DH>& st) {
#pragma unroll
  for (int b = 0; b < DH / 32; ++b)
#pragma unroll
    for (int g = 0; g < 16; ++g) st.o[b][g] = 0.f;
  st.m = kNegBig;
  st.l = 0.f;
}

template <int DH>
LC_DEVICE void load_q(bf16x8 (&qf)[DH / 16], const bf16_t* qrow, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < DH / 16; ++kk) {
    if (qrow) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + 16 * kk + 8 * h);
    else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

// Final normalisation + store. bf16 rows (final output) or fp32 rows + lse2 (partials).
template <int DH>
LC_DEVICE void finish_and_store(AttnState<DH>& st, bf16_t* out_bf16, float* out_f32, float* out_lse,
                                int lane) {
  const int h = lane >> 5;
  const float l = st.l + __shfl_xor(st.l, 32, 64);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (out_lse && h == 0) *out_lse = l > 0.f ? st.m + log2f(l) : kNegBig;
#pragma unroll
  for (int b = 0; b < DH / 32; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = b * 32 + 8 * g4 + 4 * h;
      const float v0 = st.o[b][4 * g4 + 0] * inv, v1 = st.o[b][4 * g4 + 1] * inv;
      const float v2 = st.o[b][4 * g4 + 2] * inv, v3 = st.o[b][4 * g4 + 3] * inv;
      if (out_bf16) {
        u32x2 o;
        o[0] = pack_bf16x2(v0, v1);
        o[1] = pack_bf16x2(v2, v3);
        *reinterpret_cast<u32x2*>(out_bf16 + d) = o;
      } else if (out_f32) {
        *reinterpret_cast<f32x4*>(out_f32 + d) = f32x4{v0, v1, v2, v3};
      }
    }
}

// ------------------------------------------------------------------ prefill
// one workgroup per (pair of q tiles, kv_head x ceil(G/4), seq); a workgroup is 4 waves = up to 4 of the G query heads
// that share a kv head, 32 query rows each (256 threads leave each wave the ~180 VGPRs a
// Dh = 128 tile needs). The 32-key K/V tiles are staged ONCE per workgroup in LDS by LDS-DMA
// and every wave reads its MFMA fragments from there, instead of each wave pulling the same
// 16 KiB per tile through the vector L1. A ring of PF_NSTAGE
// stages keeps two tiles in flight; one barrier per tile does both jobs of the ring (tile t
// visible to all, tile t-1 released), exactly as in gemm_panel.hip.
//
// LDS images (per stage): K = 32 keys x Dh*2 bytes, rows whole; V^T = Dh rows x 64 bytes (32
// keys). A DMA writes lane-linear, so the bank swizzle is applied to the SOURCE chunk a lane
// fetches: K position (key, c) holds chunk c ^ (key & (CH-1)) of that key, V position (d, c)
// holds chunk c ^ ((d >> 2) & 3) of row d. Fragment reads apply the same XOR.
constexpr int PF_NSTAGE = 4;

template <int DH>
struct PrefillLds {
  static constexpr int CH = DH / 8;              // 16-byte chunks per key row
  static constexpr int KEYS_PER_PIECE = 64 / CH;  // a DMA instruction moves 64 chunks
  static constexpr int K_PIECES = 32 / KEYS_PER_PIECE;
  static constexpr int V_PIECES = DH / 16;       // 16 d-rows x 4 chunks per piece
  static constexpr int PIECES = K_PIECES + V_PIECES;
  static constexpr int K_BYTES = 32 * DH * 2;
  static constexpr int STAGE = 2 * K_BYTES;
};

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_prefill_lds_kernel(
    const bf16_t* __restrict__ Q, int q_stride, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ cu_q,
    const int32_t* __restrict__ ctx_lens, bf16_t* __restrict__ out, int Hq, int Hkv, int lbs, float scale2,
    int window, int gx, int gy) {
  using L = PrefillLds<DH>;
  constexpr int PPW = L::PIECES / 4;   // DMA instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = Hq / Hkv, nsub = (G + 3) / 4;
  // One-dimensional launch: the workgroups of one (sequence, kv head) share K/V tiles and are
  // placed on one XCD.
  const int logical = xcd_group_remap(blockIdx.x, gx, gridDim.x / gx);
  const int bx = logical % gx, by = (logical / gx) % gy, s = logical / (gx * gy);
  const int kvh = by / nsub, sub = by % nsub;
  const int q0 = cu_q[s], q_len = cu_q[s + 1] - q0;
  // A workgroup takes the query tiles bx and nq-1-bx of its sequence, the long one first: under
  // the causal mask tile j attends j+1 key tiles, so every pair costs the same and the fixed
  // cost of a workgroup (block table, first tiles in flight) is paid once per nq+1 key tiles.
  const int nq = (q_len + 31) >> 5;
  const int t_heavy = nq - 1 - bx;
  if (t_heavy < bx) return;
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool has_head = sub * 4 + g < G;   // a wave without a head still moves its share of K/V
  const int head = kvh * G + min(sub * 4 + g, G - 1);
  const int ctx = ctx_lens[s];
  const int base_pos = ctx - q_len;
  const int BS = 1 << lbs;
  const int32_t* bt = block_tables + static_cast<size_t>(s) * max_blocks;
  const size_t kblk_stride = static_cast<size_t>(Hkv) * BS * DH;
  const int r = lane & 31, h = lane >> 5;
  const int pi_r = (r & ~0xC) | ((r & 4) << 1) | ((r & 8) >> 1);

  // The block-table entries this workgroup needs go to LDS first. Looking them up in global
  // memory while tiles are in flight would make every DMA address wait on vmcnt(0) -- i.e.
  // drain the ring once per piece; LDS reads are counted separately (lgkmcnt).
  int32_t* bt_s = reinterpret_cast<int32_t*>(smem + PF_NSTAGE * L::STAGE);
  {
    const int kv_top = base_pos + min(t_heavy * 32 + 31, q_len - 1) + 1;
    for (int b = threadIdx.x; b <= ((kv_top - 1) >> lbs); b += 256) bt_s[b] = bt[b];
  }
  __syncthreads();

  // Per query tile of the pair (0 = the long one): rows, visibility bounds, key-tile range.
  const bool two = t_heavy != bx;
  int row_i[2], hi_qs[2], lo_qs[2], kv_his[2], t_los[2], nts[2];
  bool valids[2];
  bf16x8 qf[2][DH / 16];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int tile = p == 0 ? t_heavy : bx;
    row_i[p] = tile * 32 + r;
    valids[p] = has_head && row_i[p] < q_len && (p == 0 || two);
    const int pos = base_pos + row_i[p];
    hi_qs[p] = valids[p] ? pos + 1 : 0;
    lo_qs[p] = (valids[p] && window > 0 && pos + 1 > window) ? pos + 1 - window : 0;
    kv_his[p] = base_pos + min(tile * 32 + 31, q_len - 1) + 1;
    int kv_lo = 0;
    if (window > 0) {
      const int first_pos = base_pos + tile * 32;
      kv_lo = first_pos + 1 > window ? first_pos + 1 - window : 0;
    }
    t_los[p] = kv_lo >> 5;
    nts[p] = (p == 0 || two) ? ((kv_his[p] + 31) >> 5) - t_los[p] : 0;   // 32-key tiles [t_lo, t_lo + nt)
    // both tiles' query rows are fetched up front: nothing but LDS-DMA is issued once the ring runs
    load_q<DH>(qf[p], valids[p] ? Q + static_cast<size_t>(q0 + row_i[p]) * q_stride + head * DH : nullptr, lane);
  }
  const int nt0 = nts[0], total = nts[0] + nts[1];

  // The ring runs through both query tiles without draining: virtual tile v < nt0 belongs to
  // the long query tile, the rest to the short one.
  auto issue_tile = [&](int v) {
    if (v >= total) return;
    const int p = v >= nt0 ? 1 : 0;
    const int kv_hi = p ? kv_his[1] : kv_his[0];
    const int k0 = ((p ? t_los[1] : t_los[0]) + (p ? v - nt0 : v)) << 5;
    char* stage = smem + (v % PF_NSTAGE) * L::STAGE;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = q * 4 + g;
      if (piece < L::K_PIECES) {
        const int kl = piece * L::KEYS_PER_PIECE + lane / L::CH, lc = lane % L::CH;   // key in tile, position
        int key = k0 + kl;
        key = key < kv_hi ? key : kv_hi - 1;
        const bf16_t* src = Kc + static_cast<size_t>(bt_s[key >> lbs]) * kblk_stride +
                            (static_cast<size_t>(kvh) * BS + (key & (BS - 1))) * DH + ((lc ^ (kl & (L::CH - 1))) << 3);
        __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(stage + piece * 1024), 16, 0, 0);
      } else {
        const int vp = piece - L::K_PIECES;
        const int d = vp * 16 + (lane >> 2), lc = lane & 3;
        int key8 = k0 + 8 * (lc ^ ((d >> 2) & 3));
        const int last8 = (kv_hi - 1) & ~7;
        key8 = key8 < kv_hi ? key8 : last8;
        const bf16_t* src = Vt + static_cast<size_t>(bt_s[key8 >> lbs]) * kblk_stride +
                            (static_cast<size_t>(kvh) * DH + d) * BS + (key8 & (BS - 1));
        __builtin_amdgcn_global_load_lds(LC_GLOBAL_PTR(src), LC_LDS_PTR(stage + L::K_BYTES + vp * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_in_flight = [&](int tiles) {   // at most `tiles` younger tiles stay in flight
    if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
  };
  static_assert(PF_NSTAGE >= 3 && PF_NSTAGE <= 5, "wait_in_flight covers up to three younger tiles");

  auto run_tiles = [&](const bf16x8 (&q)[DH / 16], int v0, int n, int t_lo, int lo_q, int hi_q, bool valid,
                       int row) {
    AttnState<DH> st;
    init_state<DH>(st);
    for (int v = v0; v < v0 + n; ++v) {
      wait_in_flight(min(PF_NSTAGE - 2, total - 1 - v));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_tile(v + PF_NSTAGE - 1);
      const char* stage = smem + (v % PF_NSTAGE) * L::STAGE;
      const int k0 = (t_lo + v - v0) << 5;
      bf16x8 kf[DH / 16], vf[DH / 32][2];
#pragma unroll
      for (int kk = 0; kk < DH / 16; ++kk)
        kf[kk] = *reinterpret_cast<const bf16x8*>(stage + pi_r * (DH * 2) + (((2 * kk + h) ^ (pi_r & (L::CH - 1))) << 4));
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int b = 0; b < DH / 32; ++b) {
          const int d = b * 32 + r;
          vf[b][sx] = *reinterpret_cast<const bf16x8*>(stage + L::K_BYTES + d * 64 + (((2 * sx + h) ^ ((d >> 2) & 3)) << 4));
        }
      process_tile<DH>(st, q, kf, vf, k0, lo_q, hi_q, scale2, lane);
    }
    bf16_t* orow = valid ? out + static_cast<size_t>(q0 + row) * (Hq * DH) + head * DH : nullptr;
    finish_and_store<DH>(st, orow, nullptr, nullptr, lane);
  };

#pragma unroll
  for (int v = 0; v < PF_NSTAGE - 1; ++v) issue_tile(v);
  run_tiles(qf[0], 0, nt0, t_los[0], lo_qs[0], hi_qs[0], valids[0], row_i[0]);
  if (two) run_tiles(qf[1], nt0, nts[1], t_los[1], lo_qs[1], hi_qs[1], valids[1], row_i[1]);
}

// ------------------------------------------------------------------ decode
// One query token per sequence. grid (kv_head, seq, split); block = 1 wave; MFMA columns
// 0..G-1 carry the G query heads of this kv head. Keys [kv_begin[b], ctx_len[b]) are cut
// into `nsplit` ranges. With partials == nullptr (nsplit == 1) the bf16 output is final.
template <int DH>
LC_DEVICE void decode_one_wave(const bf16_t* __restrict__ Q, int q_stride, const bf16_t* __restrict__ Kc,
                               const bf16_t* __restrict__ Vt, const int32_t* __restrict__ block_tables,
                               int max_blocks, const int32_t* __restrict__ ctx_lens,
                               const int32_t* __restrict__ kv_begin, bf16_t* __restrict__ out,
                               float* __restrict__ part_o, float* __restrict__ part_lse, int np_total, int Hq,
                               int Hkv, int lbs, float scale2, int window, int kvh, int b, int split, int nsplit,
                               int lane, bool merge_prefix) {
  const int G = Hq / Hkv;
  const int q = lane & 31;
  const bool valid = q < G;
  const int head = kvh * G + q;
  const int ctx = ctx_lens[b];
  int begin = kv_begin ? kv_begin[b] : 0;
  if (window > 0 && ctx > window) begin = max(begin, ctx - window);
  // split the range on 32-key boundaries
  const int ntile = (ctx - (begin & ~31) + 31) / 32;
  const int per = (ntile + nsplit - 1) / nsplit;
  const int lo = max(begin, (begin & ~31) + split * per * 32);
  const int hi = min(ctx, (begin & ~31) + (split + 1) * per * 32);

  bf16x8 qf[DH / 16];
  load_q<DH>(qf, valid ? Q + static_cast<size_t>(b) * q_stride + head * DH : nullptr, lane);
  TileCtx c{Kc, Vt, block_tables + static_cast<size_t>(b) * max_blocks, Hkv, kvh, lbs, 1 << lbs, scale2};
  AttnState<DH> st;
  init_state<DH>(st);
  if (merge_prefix) {
    // nsplit == 1 and the shared-prefix pass has ALREADY run: this wave holds the row's whole
    // private range, so it folds the prefix partials (slots 1 .. np_total-1: normalised fp32 rows
    // + log-sum-exp) into its own state and writes the final bf16 row: no partial slot for the
    // private range, no merge launch. The first prefix slot is requested before the private
    // keys are walked, so its round trip overlaps theirs.
    const int h = lane >> 5;
    const bool has_prefix = valid && begin > 0 && (!kv_begin || kv_begin[b] > 0);
    const size_t slot = (static_cast<size_t>(b) * Hq + (valid ? head : 0)) * np_total;
    f32x4 pre[DH / 32][4];
    float pre_lse = kNegBig;
    if (has_prefix) {
      pre_lse = part_lse[slot + 1];
#pragma unroll
      for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          pre[blk][g4] = *reinterpret_cast<const f32x4*>(part_o + (slot + 1) * DH + blk * 32 + 8 * g4 + 4 * h);
    } else {
#pragma unroll
      for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) pre[blk][g4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (lo < hi) attend_range<DH>(st, qf, c, lo, hi, valid ? lo : 0, valid ? hi : 0, lane);
    const float l_all = st.l + __shfl_xor(st.l, 32, 64);
    float mx = fmaxf(st.m, pre_lse);
    if (has_prefix)
      for (int p = 2; p < np_total; ++p) mx = fmaxf(mx, part_lse[slot + p]);
    const float a0 = fast_exp2(st.m - mx);
    const float w1 = pre_lse > 0.5f * kNegBig ? fast_exp2(pre_lse - mx) : 0.f;
    float den = l_all * a0 + w1;
#pragma unroll
    for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) st.o[blk][4 * g4 + r4] = st.o[blk][4 * g4 + r4] * a0 + w1 * pre[blk][g4][r4];
    if (has_prefix)
      for (int p = 2; p < np_total; ++p) {
        const float lp = part_lse[slot + p];
        if (lp <= 0.5f * kNegBig) continue;
        const float w = fast_exp2(lp - mx);
        den += w;
#pragma unroll
        for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part_o + (slot + p) * DH + blk * 32 + 8 * g4 + 4 * h);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) st.o[blk][4 * g4 + r4] += w * v[r4];
          }
      }
    // finish_and_store divides by the sum over both lane halves of st.l
    st.l = 0.5f * den;
    finish_and_store<DH>(st, valid ? out + static_cast<size_t>(b) * (Hq * DH) + head * DH : nullptr, nullptr,
                         nullptr, lane);
    return;
  }
  if (lo < hi) attend_range<DH>(st, qf, c, lo, hi, valid ? lo : 0, valid ? hi : 0, lane);
  if (part_o) {
    const size_t slot = (static_cast<size_t>(b) * Hq + head) * np_total + split;
    finish_and_store<DH>(st, nullptr, valid ? part_o + slot * DH : nullptr, valid ? part_lse + slot : nullptr,
                         lane);
    // Rows outside every shared-prefix group get no prefix partials: mark those slots empty
    // here (the prefix kernel never visits such rows).
    if (kv_begin && kv_begin[b] == 0 && split == 0 && valid && lane < 32)
      for (int p = nsplit; p < np_total; ++p) part_lse[slot - split + p] = kNegBig;
  } else {
    finish_and_store<DH>(st, valid ? out + static_cast<size_t>(b) * (Hq * DH) + head * DH : nullptr, nullptr,
                         nullptr, lane);
  }
}

template <int DH>
__global__ __launch_bounds__(64) void attn_decode_kernel(const bf16_t* __restrict__ Q, int q_stride,
                                   const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
                                   const int32_t* __restrict__ block_tables, int max_blocks,
                                   const int32_t* __restrict__ ctx_lens, const int32_t* __restrict__ kv_begin,
                                   bf16_t* __restrict__ out, float* __restrict__ part_o,
                                   float* __restrict__ part_lse, int np_total, int Hq, int Hkv, int lbs,
                                   float scale2, int window, int merge_prefix) {
  decode_one_wave<DH>(Q, q_stride, Kc, Vt, block_tables, max_blocks, ctx_lens, kv_begin, out, part_o, part_lse,
                      np_total, Hq, Hkv, lbs, scale2, window, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z,
                      threadIdx.x & 63, merge_prefix != 0);
}

// Multi-wave variant: the block's key range is cut into NW sub-ranges, one per wave, and the
// NW partial results are merged through LDS by log-sum-exp before anything is written. With
// nsplit == 1 this produces the final bf16 rows directly: no partial slots in HBM and no
// separate merge kernel in the decode step. Requires G = Hq/Hkv <= 8.
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_decode_mw_kernel(
    const bf16_t* __restrict__ Q, int q_stride, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ kv_begin, bf16_t* __restrict__ out, float* __restrict__ part_o,
    float* __restrict__ part_lse, int np_total, int Hq, int Hkv, int lbs, float scale2, int window) {
  __shared__ float sm_o[NW][8][DH + 8];   // +8: the 8 rows x 2 lane halves written at once hit disjoint banks
  __shared__ float sm_m[NW][8], sm_l[NW][8];
  const int kvh = blockIdx.x, b = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = Hq / Hkv;
  const int q = lane & 31, hh = lane >> 5;
  const bool valid = q < G;
  const int head = kvh * G + q;
  const int ctx = ctx_lens[b];
  int begin = kv_begin ? kv_begin[b] : 0;
  if (window > 0 && ctx > window) begin = max(begin, ctx - window);
  const int base = begin & ~31;
  const int ntile = (ctx - base + 31) / 32;
  const int per_w = (ntile + nsplit * NW - 1) / (nsplit * NW);   // 32-key tiles per wave
  const int part = split * NW + wave;
  const int lo = max(begin, base + part * per_w * 32);
  const int hi = min(ctx, base + (part + 1) * per_w * 32);

  bf16x8 qf[DH / 16];
  load_q<DH>(qf, valid ? Q + static_cast<size_t>(b) * q_stride + head * DH : nullptr, lane);
  TileCtx c{Kc, Vt, block_tables + static_cast<size_t>(b) * max_blocks, Hkv, kvh, lbs, 1 << lbs, scale2};
  AttnState<DH> st;
  init_state<DH>(st);
  if (lo < hi) attend_range<DH>(st, qf, c, lo, hi, valid ? lo : 0, valid ? hi : 0, lane);

  const float l_all = st.l + __shfl_xor(st.l, 32, 64);
  if (valid) {
    if (hh == 0) {
      sm_m[wave][q] = st.m;
      sm_l[wave][q] = l_all;
    }
#pragma unroll
    for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(&sm_o[wave][q][blk * 32 + 8 * g4 + 4 * hh]) =
            f32x4{st.o[blk][4 * g4 + 0], st.o[blk][4 * g4 + 1], st.o[blk][4 * g4 + 2], st.o[blk][4 * g4 + 3]};
  }
  __syncthreads();
  for (int e = threadIdx.x; e < G * DH; e += 64 * NW) {
    const int qq = e / DH, d = e % DH;
    float mx = kNegBig;
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = fmaxf(mx, sm_m[w][qq]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float sc = exp2f(sm_m[w][qq] - mx);
      num += sc * sm_o[w][qq][d];
      den += sc * sm_l[w][qq];
    }
    const float val = den > 0.f ? num / den : 0.f;
    const int hd = kvh * G + qq;
    if (part_o) {
      const size_t slot = (static_cast<size_t>(b) * Hq + hd) * np_total + split;
      part_o[slot * DH + d] = val;
      if (d == 0) {
        part_lse[slot] = den > 0.f ? mx + log2f(den) : kNegBig;
        if (kv_begin && kv_begin[b] == 0 && split == 0)   // no prefix partials for this row
          for (int p = nsplit; p < np_total; ++p) part_lse[slot + p] = kNegBig;
      }
    } else {
      out[static_cast<size_t>(b) * (Hq * DH) + hd * DH + d] = f32_to_bf16(val);
    }
  }
}

// ------------------------------------------------------------------ shared-prefix decode
// Group g = `count` consecutive batch rows starting at `first` whose first
// `shared_blocks` KV blocks are identical. Rows of the MFMA tile = (chain, q-head in the
// kv head's GQA group): count*G rows, 32 per wave. Every row sees every prefix key, so
// the prefix K/V is read once per kv head for the whole group instead of once per chain.
// grid (row_tile, kv_head, group*nsplit + split); block = 1 wave. Writes partial slots
// [slot0 + split].
template <int DH>
__global__ __launch_bounds__(64) void attn_prefix_kernel(const bf16_t* __restrict__ Q, int q_stride,
                                   const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
                                   const int32_t* __restrict__ block_tables, int max_blocks,
                                   const int32_t* __restrict__ group_first,
                                   const int32_t* __restrict__ group_count,
                                   const int32_t* __restrict__ group_shared_blocks, int nsplit,
                                   float* __restrict__ part_o, float* __restrict__ part_lse, int np_total,
                                   int slot0, int Hq, int Hkv, int lbs, float scale2) {
  const int grp = blockIdx.z / nsplit, split = blockIdx.z % nsplit;
  const int kvh = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int G = Hq / Hkv;
  const int first = group_first[grp], count = group_count[grp];
  const int plen = group_shared_blocks[grp] << lbs;
  const int row = blockIdx.x * 32 + (lane & 31);
  if (blockIdx.x * 32 >= count * G) return;
  const bool valid = row < count * G;
  const int chain = row / G, qh = row % G;
  const int b = first + chain;
  const int head = kvh * G + qh;
  const int ntile = plen / 32;  // BS >= 32 keeps plen a multiple of 32; smaller BS handled by masking
  const int per = ((plen + 31) / 32 + nsplit - 1) / nsplit;
  const int lo = split * per * 32;
  const int hi = min(plen, (split + 1) * per * 32);
  (void)ntile;

  bf16x8 qf[DH / 16];
  load_q<DH>(qf, valid ? Q + static_cast<size_t>(b) * q_stride + head * DH : nullptr, lane);
  TileCtx c{Kc, Vt, block_tables + static_cast<size_t>(first) * max_blocks, Hkv, kvh, lbs, 1 << lbs, scale2};
  AttnState<DH> st;
  init_state<DH>(st);
  if (lo < hi) attend_range<DH>(st, qf, c, lo, hi, valid ? lo : 0, valid ? hi : 0, lane);
  const size_t slot = valid ? (static_cast<size_t>(b) * Hq + head) * np_total + slot0 + split : 0;
  finish_and_store<DH>(st, nullptr, valid ? part_o + slot * DH : nullptr, valid ? part_lse + slot : nullptr, lane);
}

// Multi-wave form of the shared-prefix pass: the NW waves of a workgroup take consecutive
// pieces of the split's key range and their partial results are merged by log-sum-exp in LDS,
// so a group's prefix costs one partial slot per row in HBM instead of one per piece (with five
// pieces of a 512-key prompt the slots, written here and read again by the merge kernel, were
// more traffic than the prefix keys themselves).
template <int DH, int NW>
LC_DEVICE void prefix_multi_wave(
    const bf16_t* __restrict__ Q, int q_stride, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_count, const int32_t* __restrict__ group_shared_blocks, int nsplit,
    float* __restrict__ part_o, float* __restrict__ part_lse, int np_total, int slot0, int Hq, int Hkv, int lbs,
    float scale2, char* pm_smem, int bx, int kvh, int bz) {
  static_assert(NW == 4, "the merge below is a two-level tree over four waves");
  constexpr int NV = DH / 2;                       // accumulator values per lane
  float* sm = reinterpret_cast<float*>(pm_smem);   // [2 slots][NV + 2 values][64 lanes]
  const int grp = bz / nsplit, split = bz % nsplit;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = Hq / Hkv;
  const int first = group_first[grp], count = group_count[grp];
  const int plen = group_shared_blocks[grp] << lbs;
  if (bx * 32 >= count * G) return;     // block-uniform: an empty group slot
  const int q = lane & 31;
  const int row = bx * 32 + q;
  const bool valid = row < count * G;
  const int b = first + row / G;
  const int head = kvh * G + row % G;
  const int per = ((plen + 31) / 32 + nsplit * NW - 1) / (nsplit * NW);   // 32-key tiles per wave
  const int part = split * NW + wave;
  const int lo = min(plen, part * per * 32);
  const int hi = min(plen, (part + 1) * per * 32);

  bf16x8 qf[DH / 16];
  load_q<DH>(qf, valid ? Q + static_cast<size_t>(b) * q_stride + head * DH : nullptr, lane);
  TileCtx c{Kc, Vt, block_tables + static_cast<size_t>(first) * max_blocks, Hkv, kvh, lbs, 1 << lbs, scale2};
  AttnState<DH> st;
  init_state<DH>(st);
  if (lo < hi) attend_range<DH>(st, qf, c, lo, hi, valid ? lo : 0, valid ? hi : 0, lane);

  // Merge of the four waves' states by log-sum-exp, as a tree: a lane of one wave and the same
  // lane of another hold the same (row, dims) positions, so a state is handed over lane to lane
  // through LDS (lane-major: no bank conflicts) without looking at its layout. Waves 2, 3 ->
  // waves 0, 1; wave 1 -> wave 0, which normalises and writes the partial slot. Two states of
  // 16.5 KiB in LDS (the all-to-all form needed four in row-major order, 65 KiB, which kept the
  // workgroups of a launch at two per CU).
  auto put = [&](int s_) {
    float* p = sm + s_ * (NV + 2) * 64 + lane;
#pragma unroll
    for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
      for (int g = 0; g < 16; ++g) p[(blk * 16 + g) * 64] = st.o[blk][g];
    p[NV * 64] = st.m;
    p[(NV + 1) * 64] = st.l;
  };
  auto take = [&](int s_) {
    const float* p = sm + s_ * (NV + 2) * 64 + lane;
    const float m2 = p[NV * 64], l2 = p[(NV + 1) * 64];
    const float mx = fmaxf(st.m, m2);
    const float a1 = exp2f(st.m - mx), a2 = exp2f(m2 - mx);
    st.l = st.l * a1 + l2 * a2;
#pragma unroll
    for (int blk = 0; blk < DH / 32; ++blk)
#pragma unroll
      for (int g = 0; g < 16; ++g) st.o[blk][g] = st.o[blk][g] * a1 + p[(blk * 16 + g) * 64] * a2;
    st.m = mx;
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) take(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    take(0);
    const size_t slot = valid ? (static_cast<size_t>(b) * Hq + head) * np_total + slot0 + split : 0;
    finish_and_store<DH>(st, nullptr, valid ? part_o + slot * DH : nullptr, valid ? part_lse + slot : nullptr, lane);
  }
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void attn_prefix_mw_kernel(
    const bf16_t* __restrict__ Q, int q_stride, const bf16_t* __restrict__ Kc, const bf16_t* __restrict__ Vt,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_count, const int32_t* __restrict__ group_shared_blocks, int nsplit,
    float* __restrict__ part_o, float* __restrict__ part_lse, int np_total, int slot0, int Hq, int Hkv, int lbs,
    float scale2) {
  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  prefix_multi_wave<DH, NW>(Q, q_stride, Kc, Vt, block_tables, max_blocks, group_first, group_count,
                            group_shared_blocks, nsplit, part_o, part_lse, np_total, slot0, Hq, Hkv, lbs, scale2,
                            pm_smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------ merge
// out[b, head, :] = sum_p w_p * part_o[b, head, p, :],  w_p = 2^(lse_p - max) / sum.
// One wave per (b, head); lanes cover Dh.
template <int DH>
__global__ void attn_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_lse,
                                  bf16_t* __restrict__ out, int np_total, int n_rows) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const float* lse = part_lse + static_cast<size_t>(row) * np_total;
  float mx = kNegBig;
  for (int p = 0; p < np_total; ++p) mx = fmaxf(mx, lse[p]);
  float acc[DH / 64] = {0.f};
  float wsum = 0.f;
  for (int p = 0; p < np_total; ++p) {
    const float l = lse[p];
    if (l <= kNegBig * 0.5f) continue;
    const float w = exp2f(l - mx);
    wsum += w;
    const float* o = part_o + (static_cast<size_t>(row) * np_total + p) * DH;
#pragma unroll
    for (int i = 0; i < DH / 64; ++i) acc[i] += w * o[lane + 64 * i];
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
#pragma unroll
  for (int i = 0; i < DH / 64; ++i)
    out[static_cast<size_t>(row) * DH + lane + 64 * i] = f32_to_bf16(acc[i] * inv);
}

__global__ void fill_f32_kernel(float* __restrict__ p, float v, size_t n) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

int ilog2(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

}  // namespace

LC_EXPORT int lc_attn_prefill(const void* Q, int q_stride, const void* Kc, const void* Vt,
                              const void* block_tables, int max_blocks, const void* cu_q,
                              const void* ctx_lens, void* out, int n_seqs, int max_q_len, int Hq, int Hkv,
                              int Dh, int block_size, float scale, int window, hipStream_t stream) {
  if (n_seqs <= 0 || max_q_len <= 0) return 0;
  if (Hq % Hkv || (Dh != 64 && Dh != 128) || block_size % 8 || (block_size & (block_size - 1))) return -1;
  const int G = Hq / Hkv;
  const int gx = ((max_q_len + 31) / 32 + 1) / 2, gy = Hkv * ((G + 3) / 4);
  const dim3 grid(static_cast<unsigned>(gx) * gy * n_seqs), block(256);
  const float scale2 = scale * 1.4426950408889634f;
  const int lbs = ilog2(block_size);
#define LC_ARGS                                                                                          \
  static_cast<const bf16_t*>(Q), q_stride, static_cast<const bf16_t*>(Kc), static_cast<const bf16_t*>(Vt), \
      static_cast<const int32_t*>(block_tables), max_blocks, static_cast<const int32_t*>(cu_q),          \
      static_cast<const int32_t*>(ctx_lens), static_cast<bf16_t*>(out), Hq, Hkv, lbs, scale2, window, gx, gy
  if (max_blocks > 8192) return -1;   // the block-table slice kept in LDS
  const size_t bt_bytes = static_cast<size_t>(max_blocks) * sizeof(int32_t);
  if (Dh == 128)
    attn_prefill_lds_kernel<128><<<grid, block, PF_NSTAGE * PrefillLds<128>::STAGE + bt_bytes, stream>>>(LC_ARGS);
  else
    attn_prefill_lds_kernel<64><<<grid, block, PF_NSTAGE * PrefillLds<64>::STAGE + bt_bytes, stream>>>(LC_ARGS);
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}

// Decode attention. part_o/part_lse may be null when nsplit == 1 and no prefix pass is
// used; otherwise partial slot `split` of np_total is written and lc_attn_merge finishes.
LC_EXPORT int lc_attn_decode(const void* Q, int q_stride, const void* Kc, const void* Vt,
                             const void* block_tables, int max_blocks, const void* ctx_lens,
                             const void* kv_begin, void* out, void* part_o, void* part_lse, int np_total,
                             int nsplit, int n_seqs, int Hq, int Hkv, int Dh, int block_size, float scale,
                             int window, int waves, int merge_prefix, hipStream_t stream) {
  if (n_seqs <= 0) return 0;
  // merge_prefix: the shared-prefix pass has run; one wave per (row, kv head) walks the private keys,
  // folds the prefix slots 1 .. np_total-1 in and writes the final rows
  if (merge_prefix && (waves != 1 || nsplit != 1 || !part_o || !part_lse || !kv_begin || np_total < 2)) return -1;
  if (Hq % Hkv || Hq / Hkv > 32 || (Dh != 64 && Dh != 128) || block_size % 8 ||
      (block_size & (block_size - 1)) || nsplit < 1 || (!part_o && nsplit != 1) || nsplit > np_total)
    return -1;
  if (waves != 1 && ((waves != 2 && waves != 4) || Hq / Hkv > 8)) return -1;
  const dim3 grid(Hkv, n_seqs, nsplit), block(64);
  const float scale2 = scale * 1.4426950408889634f;
  const int lbs = ilog2(block_size);
#define LC_ARGS                                                                                          \
  static_cast<const bf16_t*>(Q), q_stride, static_cast<const bf16_t*>(Kc), static_cast<const bf16_t*>(Vt), \
      static_cast<const int32_t*>(block_tables), max_blocks, static_cast<const int32_t*>(ctx_lens),      \
      static_cast<const int32_t*>(kv_begin), static_cast<bf16_t*>(out), static_cast<float*>(part_o),     \
      static_cast<float*>(part_lse), np_total, Hq, Hkv, lbs, scale2, window
  if (waves == 4) {
    if (Dh == 128) attn_decode_mw_kernel<128, 4><<<grid, dim3(256), 0, stream>>>(LC_ARGS);
    else attn_decode_mw_kernel<64, 4><<<grid, dim3(256), 0, stream>>>(LC_ARGS);
  } else if (waves == 2) {
    if (Dh == 128) attn_decode_mw_kernel<128, 2><<<grid, dim3(128), 0, stream>>>(LC_ARGS);
    else attn_decode_mw_kernel<64, 2><<<grid, dim3(128), 0, stream>>>(LC_ARGS);
  } else if (Dh == 128) {
    attn_decode_kernel<128><<<grid, block, 0, stream>>>(LC_ARGS, merge_prefix);
  } else {
    attn_decode_kernel<64><<<grid, block, 0, stream>>>(LC_ARGS, merge_prefix);
  }
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_attn_prefix(const void* Q, int q_stride, const void* Kc, const void* Vt,
                             const void* block_tables, int max_blocks, const void* group_first,
                             const void* group_count, const void* group_shared_blocks, int n_groups,
                             int max_group_rows, int nsplit, void* part_o, void* part_lse, int np_total,
                             int slot0, int Hq, int Hkv, int Dh, int block_size, float scale, int waves,
                             hipStream_t stream) {
  if (n_groups <= 0 || max_group_rows <= 0) return 0;
  if (Hq % Hkv || (Dh != 64 && Dh != 128) || block_size % 8 || (block_size & (block_size - 1)) ||
      nsplit < 1 || slot0 + nsplit > np_total || (waves != 1 && waves != 4))
    return -1;
  const int G = Hq / Hkv;
  const dim3 grid((max_group_rows * G + 31) / 32, Hkv, n_groups * nsplit), block(64);
  const float scale2 = scale * 1.4426950408889634f;
  const int lbs = ilog2(block_size);
#define LC_ARGS                                                                                          \
  static_cast<const bf16_t*>(Q), q_stride, static_cast<const bf16_t*>(Kc), static_cast<const bf16_t*>(Vt), \
      static_cast<const int32_t*>(block_tables), max_blocks, static_cast<const int32_t*>(group_first),   \
      static_cast<const int32_t*>(group_count), static_cast<const int32_t*>(group_shared_blocks), nsplit, \
      static_cast<float*>(part_o), static_cast<float*>(part_lse), np_total, slot0, Hq, Hkv, lbs, scale2
  if (waves == 4) {
    const size_t lds = sizeof(float) * 2 * (Dh / 2 + 2) * 64;
    if (Dh == 128) attn_prefix_mw_kernel<128, 4><<<grid, dim3(256), lds, stream>>>(LC_ARGS);
    else attn_prefix_mw_kernel<64, 4><<<grid, dim3(256), lds, stream>>>(LC_ARGS);
  } else if (Dh == 128) {
    attn_prefix_kernel<128><<<grid, block, 0, stream>>>(LC_ARGS);
  } else {
    attn_prefix_kernel<64><<<grid, block, 0, stream>>>(LC_ARGS);
  }
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_attn_merge(const void* part_o, const void* part_lse, void* out, int np_total, int n_rows,
                            int Dh, hipStream_t stream) {
  if (n_rows <= 0) return 0;
  if (Dh != 64 && Dh != 128) return -1;
  const dim3 grid((n_rows + 3) / 4), block(256);
  if (Dh == 128)
    attn_merge_kernel<128><<<grid, block, 0, stream>>>(static_cast<const float*>(part_o),
                                                       static_cast<const float*>(part_lse),
                                                       static_cast<bf16_t*>(out), np_total, n_rows);
  else
    attn_merge_kernel<64><<<grid, block, 0, stream>>>(static_cast<const float*>(part_o),
                                                      static_cast<const float*>(part_lse),
                                                      static_cast<bf16_t*>(out), np_total, n_rows);
  LC_RETURN_LAST_ERROR();
}

LC_EXPORT int lc_fill_f32(void* p, float v, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = static_cast<int>(std::min<long long>((n + 255) / 256, 2048));
  fill_f32_kernel<<<dim3(grid), dim3(256), 0, stream>>>(static_cast<float*>(p), v, static_cast<size_t>(n));
  LC_RETURN_LAST_ERROR();
}
