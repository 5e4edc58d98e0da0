// MFMA issue-rate ceiling on gfx950: waves that do nothing but independent MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak benchmarks/micro/mfma_peak.hip && build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_wave_iter, int threads, int blocks, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  launch(iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(s);
  launch(iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  const double flops = flop_per_wave_iter * iters * (threads / 64) * blocks;
  std::printf("%-28s threads %3d blocks %4d  %8.3f ms  %6.3f PFLOP/s\n", name, threads, blocks, ms, flops / ms / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 512 * sizeof(float));
  const int iters = 20000;
  for (int blocks : {256, 512, 2048}) {
    for (int threads : {256, 512}) {
      run("16x16x32 x8 acc", [&](int it) { k16<8><<<blocks, threads>>>(out, it); }, 8 * 16384.0, threads, blocks, iters);
      run("16x16x32 x32 acc", [&](int it) { k16<32><<<blocks, threads>>>(out, it / 4); }, 32 * 16384.0 / 4, threads, blocks, iters);
      run("32x32x16 x4 acc", [&](int it) { k32<4><<<blocks, threads>>>(out, it); }, 4 * 32768.0, threads, blocks, iters);
    }
  }
  hipFree(out);
  return 0;
}
