// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernel library.
// Wave width is 64 everywhere; MFMA fragments follow the gfx950 lane maps:
//   v_mfma_f32_16x16x32_bf16: A lane l -> A[row l&15][k 8*(l>>4)+j], B lane l -> B[k 8*(l>>4)+j][col l&15],
//                             C/D col = l&15, row = 4*(l>>4)+reg.
//   v_mfma_f32_32x32x16_bf16: A lane l -> A[row l&31][k 8*(l>>5)+j], B lane l -> B[k 8*(l>>5)+j][col l&31],
//                             C/D col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LC_WAVE 64
#define LC_DEVICE __device__ __forceinline__

#define LC_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LC_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

LC_DEVICE float bf16_to_f32(bf16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 lc_bf16v2;
typedef __attribute__((ext_vector_type(2))) float lc_f32v2;

// fp32 -> bf16, round-to-nearest-even, in hardware (v_cvt_pk_bf16_f32 converts two at a time).
LC_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
  const lc_f32v2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, lc_bf16v2));
}

LC_DEVICE bf16_t f32_to_bf16(float f) { return static_cast<bf16_t>(pack_bf16x2(f, 0.f) & 0xffffu); }

// 2^x as the bare v_exp_f32 (no denormal-range fix-up: results below 2^-126 flush to zero,
// which is what a softmax weight that small should do anyway).
LC_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

LC_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

LC_DEVICE float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

LC_DEVICE int wave_max_int(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide reductions for blocks of up to 16 waves; `scratch` needs 16 floats.
LC_DEVICE float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}

LC_DEVICE float block_max(float v, float* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = scratch[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
  return t;
}

// Counter-based RNG: one 64-bit mix per (stream, step, lane-counter) triple. Results
// depend only on the arguments, so sampling is reproducible regardless of batch
// composition or launch geometry.
LC_DEVICE uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

LC_DEVICE float rng_uniform(uint64_t seed, uint32_t step, uint32_t counter) {
  uint64_t z = mix64(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
  z = mix64(z ^ ((static_cast<uint64_t>(step) << 32) | counter));
  return (static_cast<float>(z >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
}

// Per-device, thread-safe "do this once" for launch attributes (the dynamic-LDS limit of a
// kernel is a per-device property, and launchers are called from several host threads: one
// per engine stream, one per loopback rank). The guarded call must be idempotent: two threads
// that race on a device's first launch may both run it.
struct LcDeviceOnce {
  std::atomic<uint8_t> done[64];
  template <class F>
  void run(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (done[dev].load(std::memory_order_acquire)) return;
    f();
    done[dev].store(1, std::memory_order_release);
  }
};

#define LC_EXPORT extern "C" __attribute__((visibility("default")))

// Launch-error reporting: every launcher returns hipGetLastError() as int so the
// Python side can raise; nothing aborts the process.
#define LC_RETURN_LAST_ERROR() return static_cast<int>(hipGetLastError())
