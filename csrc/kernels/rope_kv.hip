// RoPE + paged KV-cache write, fused. Input is the QKV projection output
// qkv[T, (Hq + 2*Hkv) * Dh] (bias already applied by the GEMM epilogue). Q heads are
// rotated in place; K heads are rotated and written to their cache slot
// K[blk][Hkv][BS][Dh]; V heads are written transposed into Vt[blk][Hkv][Dh][BS].
// cos/sin come from a host-built fp32 table [max_pos][Dh/2] (on-device trig would make
// this VALU-bound). Rotation is the "rotate-half" convention of HF Llama/Mistral/Qwen2:
// pairs (i, i + Dh/2).
#include "common.h"

namespace {

template <int DH>
__global__ void rope_kv_write_kernel(bf16_t* __restrict__ qkv, const int32_t* __restrict__ positions,
                                     const int32_t* __restrict__ slot_mapping,
                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                     bf16_t* __restrict__ Kc, bf16_t* __restrict__ Vt, int T, int Hq,
                                     int Hkv, int BS, int max_pos) {
  constexpr int HALF = DH / 2;
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  int pos = positions[t];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  const int slot = slot_mapping[t];
  const int blk = slot / BS, off = slot % BS;
  bf16_t* row = qkv + static_cast<size_t>(t) * (Hq + 2 * Hkv) * DH;
  const int nheads = Hq + 2 * Hkv;
  for (int hd = wave; hd < nheads; hd += nwave) {
    bf16_t* x = row + hd * DH;
    if (hd < Hq + Hkv) {
      // rotate (Q in place; K into the cache)
      for (int i = lane; i < HALF; i += 64) {
        const float c = cos_t[static_cast<size_t>(pos) * HALF + i], s = sin_t[static_cast<size_t>(pos) * HALF + i];
        const float a = bf16_to_f32(x[i]), b = bf16_to_f32(x[i + HALF]);
        const bf16_t ra = f32_to_bf16(a * c - b * s), rb = f32_to_bf16(b * c + a * s);
        if (hd < Hq) {
          x[i] = ra;
          x[i + HALF] = rb;
        } else if (slot >= 0) {
          bf16_t* dst = Kc + ((static_cast<size_t>(blk) * Hkv + (hd - Hq)) * BS + off) * DH;
          dst[i] = ra;
          dst[i + HALF] = rb;
        }
      }
    } else if (slot >= 0) {
      const int vh = hd - Hq - Hkv;
      bf16_t* dst = Vt + (static_cast<size_t>(blk) * Hkv + vh) * DH * BS + off;
      for (int i = lane; i < DH; i += 64) dst[static_cast<size_t>(i) * BS] = x[i];
    }
  }
}

}  // namespace

LC_EXPORT int lc_rope_kv_write(void* qkv, const void* positions, const void* slot_mapping,
                               const void* cos_t, const void* sin_t, void* Kc, void* Vt, int T, int Hq,
                               int Hkv, int Dh, int block_size, int max_pos, hipStream_t stream) {
  if (T <= 0) return 0;
  if (Dh != 64 && Dh != 128) return -1;
  const dim3 grid(T), block(256);
#define LC_ARGS                                                                                       \
  static_cast<bf16_t*>(qkv), static_cast<const int32_t*>(positions),                                  \
      static_cast<const int32_t*>(slot_mapping), static_cast<const float*>(cos_t),                    \
      static_cast<const float*>(sin_t), static_cast<bf16_t*>(Kc), static_cast<bf16_t*>(Vt), T, Hq, Hkv, \
      block_size, max_pos
  if (Dh == 128) rope_kv_write_kernel<128><<<grid, block, 0, stream>>>(LC_ARGS);
  else rope_kv_write_kernel<64><<<grid, block, 0, stream>>>(LC_ARGS);
#undef LC_ARGS
  LC_RETURN_LAST_ERROR();
}
