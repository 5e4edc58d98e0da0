// RoPE + paged KV-cache write, fused. Input is the QKV projection output
// qkv[T, (Hq + 2*Hkv) * Dh] (bias already applied by the GEMM epilogue). Q heads are
// rotated in place; K heads are rotated and written to their cache slot
// K[blk][Hkv][BS][Dh]; V heads are written transposed into Vt[blk][Hkv][Dh][BS].
// cos/sin come from a host-built fp32 table [max_pos][Dh/2] (on-device trig would make
// this VALU-bound). Rotation is the "rotate-half" convention of HF Llama/Mistral/Qwen2:
// pairs (i, i + Dh/2).
//
// The projection may arrive un-reduced: `part` = S fp32 slabs [S][T][ld] from a split-K
// GEMM (+ optional bias). The slabs are summed here, so the separate reduction kernel and
// its launch disappear from the decode step; Q is then written (rotated, bf16) into `qkv`.
#include "common.h"

namespace {

template <int DH>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(
    bf16_t* __restrict__ qkv, const int32_t* __restrict__ positions, const int32_t* __restrict__ slot_mapping,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, bf16_t* __restrict__ Kc,
    bf16_t* __restrict__ Vt, int T, int Hq, int Hkv, int BS, int max_pos, const float* __restrict__ part, int S,
    const bf16_t* __restrict__ bias, int skip_v) {
  constexpr int HALF = DH / 2;
  constexpr int QUADS = HALF / 4;  // work item = 4 consecutive rotation pairs (or 8 V elements)
  const int t = blockIdx.x;
  int pos = positions[t];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  const int slot = slot_mapping[t];
  const int blk = slot / BS, off = slot % BS;
  const int nheads = Hq + 2 * Hkv;
  const int ld = nheads * DH;
  const size_t slab = static_cast<size_t>(T) * ld;
  bf16_t* row = qkv + static_cast<size_t>(t) * ld;
  const float* prow = part ? part + static_cast<size_t>(t) * ld : nullptr;

  // four consecutive elements of this token's projection row, rounded to bf16 precision
  // (the rounding point of a bf16 GEMM epilogue) and widened to fp32
  auto load4 = [&](int col) -> f32x4 {
    f32x4 v;
    if (prow) {
      // All slabs' loads are issued before the first add: a loop of load-then-add would pay one
      // L2 round trip per slab, and this kernel is nothing but round trips.
      constexpr int MAX_S = 8;
      f32x4 ps[MAX_S];
#pragma unroll
      for (int sp = 0; sp < MAX_S; ++sp)
        if (sp < S) ps[sp] = *reinterpret_cast<const f32x4*>(prow + sp * slab + col);
      v = ps[0];
#pragma unroll
      for (int sp = 1; sp < MAX_S; ++sp)
        if (sp < S) v += ps[sp];
      for (int sp = MAX_S; sp < S; ++sp) v += *reinterpret_cast<const f32x4*>(prow + sp * slab + col);
      if (bias) {
        const u32x2 bb = *reinterpret_cast<const u32x2*>(bias + col);
        v[0] += bf16_to_f32(static_cast<bf16_t>(bb[0] & 0xffff));
        v[1] += bf16_to_f32(static_cast<bf16_t>(bb[0] >> 16));
        v[2] += bf16_to_f32(static_cast<bf16_t>(bb[1] & 0xffff));
        v[3] += bf16_to_f32(static_cast<bf16_t>(bb[1] >> 16));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = bf16_to_f32(f32_to_bf16(v[j]));
    } else {
      const u32x2 xx = *reinterpret_cast<const u32x2*>(row + col);
      v[0] = bf16_to_f32(static_cast<bf16_t>(xx[0] & 0xffff));
      v[1] = bf16_to_f32(static_cast<bf16_t>(xx[0] >> 16));
      v[2] = bf16_to_f32(static_cast<bf16_t>(xx[1] & 0xffff));
      v[3] = bf16_to_f32(static_cast<bf16_t>(xx[1] >> 16));
    }
    return v;
  };
  auto store4 = [](bf16_t* dst, const f32x4& v) {
    u32x2 o;
    o[0] = pack_bf16x2(v[0], v[1]);
    o[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(dst) = o;
  };

  // Work items of a token: (Hq + Hkv) * QUADS rotations (pairs (i, i + HALF) of Q and K heads) and
  // Hkv * DH / 4 transposed V scatters, in one index space so that at decode sizes every thread
  // has at most one item and all of a token's loads are in flight together. gridDim.y blocks
  // share a token: decode steps have few tokens, and one block per token would leave most CUs
  // idle while every thread walks several dependent load chains.
  const int tid = blockIdx.y * blockDim.x + threadIdx.x, nthr = gridDim.y * blockDim.x;
  const int n_rot = (Hq + Hkv) * QUADS;
  const int n_v = (slot < 0 || skip_v) ? 0 : Hkv * (DH / 4);
  for (int it = tid; it < n_rot + n_v; it += nthr) {
    if (it < n_rot) {
      const int hd = it / QUADS, i = (it % QUADS) * 4;
      if (hd >= Hq && slot < 0) continue;
      const f32x4 a = load4(hd * DH + i), b = load4(hd * DH + i + HALF);
      const f32x4 c = *reinterpret_cast<const f32x4*>(cos_t + static_cast<size_t>(pos) * HALF + i);
      const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_t + static_cast<size_t>(pos) * HALF + i);
      f32x4 ra, rb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[j] = a[j] * c[j] - b[j] * sn[j];
        rb[j] = b[j] * c[j] + a[j] * sn[j];
      }
      bf16_t* dst = hd < Hq ? row + hd * DH
                            : Kc + ((static_cast<size_t>(blk) * Hkv + (hd - Hq)) * BS + off) * DH;
      store4(dst + i, ra);
      store4(dst + i + HALF, rb);
    } else {
      const int iv = it - n_rot;
      const int vh = iv / (DH / 4), i = (iv % (DH / 4)) * 4;
      const f32x4 v = load4((Hq + Hkv + vh) * DH + i);
      bf16_t* dst = Vt + ((static_cast<size_t>(blk) * Hkv + vh) * DH + i) * BS + off;
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[static_cast<size_t>(j) * BS] = f32_to_bf16(v[j]);
    }
  }
}

// V write for prefill-sized steps. The cache keeps V transposed (Vt[blk][Hkv][DH][BS]) so that the
// attention kernels read keys along the fastest axis; written token by token that is one 2-byte
// store per element, BS*2 bytes apart. Here a block takes 64 consecutive tokens of one KV head
// through LDS and writes, per head dimension, runs of 8 tokens as 16-byte stores (eight lanes
// cover a 128-byte line when the 64 tokens fill one cache block). A run whose slots are not
// consecutive, 8-aligned and inside one block (sequence boundaries, odd prompt lengths) falls
// back to element stores.
template <int DH>
__global__ __launch_bounds__(256) void v_transpose_write_kernel(const bf16_t* __restrict__ qkv,
                                                                const int32_t* __restrict__ slot_mapping,
                                                                bf16_t* __restrict__ Vt, int T, int Hq, int Hkv,
                                                                int BS) {
  constexpr int TOK = 64, LDT = DH + 8;   // padded row: 8-token column reads spread over the banks
  __shared__ bf16_t tile[TOK * LDT];
  const int t0 = blockIdx.x * TOK, vh = blockIdx.y;
  const int ld = (Hq + 2 * Hkv) * DH;
  const bf16_t* src = qkv + static_cast<size_t>(Hq + Hkv + vh) * DH;
  constexpr int CHUNKS = DH / 8;
  for (int it = threadIdx.x; it < TOK * CHUNKS; it += 256) {
    const int r = it / CHUNKS, c = it % CHUNKS;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t0 + r < T) v = *reinterpret_cast<const bf16x8*>(src + static_cast<size_t>(t0 + r) * ld + c * 8);
    *reinterpret_cast<bf16x8*>(tile + r * LDT + c * 8) = v;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < DH * (TOK / 8); it += 256) {
    const int d = it / (TOK / 8), g = it % (TOK / 8);
    const int r0 = g * 8;
    if (t0 + r0 >= T) continue;
    int sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sl[j] = t0 + r0 + j < T ? slot_mapping[t0 + r0 + j] : -1;
    bool run = sl[0] >= 0 && (sl[0] & 7) == 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) run = run && sl[j] == sl[0] + j;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = static_cast<short>(tile[(r0 + j) * LDT + d]);
    if (run) {   // BS is a multiple of 8, so an aligned run of 8 slots stays inside one block
      const int blk = sl[0] / BS, off = sl[0] % BS;
      *reinterpret_cast<bf16x8*>(Vt + ((static_cast<size_t>(blk) * Hkv + vh) * DH + d) * BS + off) = o;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (sl[j] < 0) continue;
        const int blk = sl[j] / BS, off = sl[j] % BS;
        Vt[((static_cast<size_t>(blk) * Hkv + vh) * DH + d) * BS + off] = static_cast<bf16_t>(o[j]);
      }
    }
  }
}

}  // namespace

LC_EXPORT int lc_rope_kv_write(void* qkv, const void* positions, const void* slot_mapping,
                               const void* cos_t, const void* sin_t, void* Kc, void* Vt, int T, int Hq,
                               int Hkv, int Dh, int block_size, int max_pos, const void* part, int S,
                               const void* bias, hipStream_t stream) {
  if (T <= 0) return 0;
  if ((Dh != 64 && Dh != 128) || (part && S < 1)) return -1;
  const dim3 grid(T, T <= 512 ? 4 : 1), block(256);
  // prefill-sized steps with a finished (bf16) projection: V goes through the transposing kernel
  const int skip_v = (T > 512 && !part && block_size % 8 == 0) ? 1 : 0;
#define LC_ARGS                                                                                       \
  static_cast<bf16_t*>(qkv), static_cast<const int32_t*>(positions),                                  \
      static_cast<const int32_t*>(slot_mapping), static_cast<const float*>(cos_t),                    \
      static_cast<const float*>(sin_t), static_cast<bf16_t*>(Kc), static_cast<bf16_t*>(Vt), T, Hq, Hkv, \
      block_size, max_pos, static_cast<const float*>(part), S, static_cast<const bf16_t*>(bias), skip_v
  if (Dh == 128) rope_kv_write_kernel<128><<<grid, block, 0, stream>>>(LC_ARGS);
  else rope_kv_write_kernel<64><<<grid, block, 0, stream>>>(LC_ARGS);
#undef LC_ARGS
  if (skip_v) {
    const dim3 vgrid((T + 63) / 64, Hkv);
    if (Dh == 128)
      v_transpose_write_kernel<128><<<vgrid, block, 0, stream>>>(
          static_cast<const bf16_t*>(qkv), static_cast<const int32_t*>(slot_mapping), static_cast<bf16_t*>(Vt), T, Hq,
          Hkv, block_size);
    else
      v_transpose_write_kernel<64><<<vgrid, block, 0, stream>>>(
          static_cast<const bf16_t*>(qkv), static_cast<const int32_t*>(slot_mapping), static_cast<bf16_t*>(Vt), T, Hq,
          Hkv, block_size);
  }
  LC_RETURN_LAST_ERROR();
}
