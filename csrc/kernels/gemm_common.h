// Pieces shared by the GEMM translation units: epilogue modes and the accumulator-tile
// store (bias, bf16/fp32 output, fused SwiGLU over gate/up tiles interleaved at 16 columns).
#pragma once
#include "common.h"

namespace {

enum Epilogue : int {
  EPI_BF16 = 0,    // Y bf16 [M, ldy]  (+bias)
  EPI_F32 = 1,     // Y f32  [M, ldy]  (+bias); with split-K: slab blockIdx.y of [S, M, ldy], no bias
  EPI_SWIGLU = 2,  // n-tiles alternate gate/up; Y bf16 [M, N/2] = silu(gate)*up (+bias before act)
};

LC_DEVICE float silu(float x) { return x / (1.f + __expf(-x)); }

// Stores one 16(n) x 16(m) accumulator tile. `n0`/`m0` are the tile origin.
template <int EPI>
LC_DEVICE void store_tile(const f32x4& acc, const f32x4* acc_up, void* __restrict__ Y,
                          const bf16_t* __restrict__ bias, int m0, int n0, int M, int N, int ldy,
                          int lane, size_t slab_off) {
  const int m = m0 + (lane & 15);
  const int nb = n0 + 4 * (lane >> 4);
  if (m >= M) return;
  if constexpr (EPI == EPI_SWIGLU) {
    // n0 is the gate tile's origin in the interleaved space; output column = n0/2 + ...
    if (nb >= N) return;
    const int ob = (n0 >> 1) + 4 * (lane >> 4);
    float g[4], u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g[r] = acc[r];
      u[r] = (*acc_up)[r];
      if (bias) {
        g[r] += bf16_to_f32(bias[nb + r]);
        u[r] += bf16_to_f32(bias[nb + 16 + r]);
      }
    }
    u32x2 o;
    o[0] = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
    o[1] = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
    *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(Y) + static_cast<size_t>(m) * ldy + ob) = o;
  } else {
    if (nb >= N) return;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + ((bias && nb + r < N) ? bf16_to_f32(bias[nb + r]) : 0.f);
    if constexpr (EPI == EPI_BF16) {
      bf16_t* dst = static_cast<bf16_t*>(Y) + static_cast<size_t>(m) * ldy + nb;
      if (nb + 3 < N) {
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(dst) = o;
      } else {
        for (int r = 0; r < 4 && nb + r < N; ++r) dst[r] = f32_to_bf16(v[r]);
      }
    } else {
      float* dst = static_cast<float*>(Y) + slab_off + static_cast<size_t>(m) * ldy + nb;
      if (nb + 3 < N) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        for (int r = 0; r < 4 && nb + r < N; ++r) dst[r] = v[r];
      }
    }
  }
}

}  // namespace
