// Stand-alone comparison of two main loops for the prefill GEMM on one 256 x 256 x 64 tile:
//   lc_gemm256   8 waves, 64(n) x 128(m) per wave, two waves per SIMD (csrc/kernels/gemm256.hip)
//   gemm256w     4 waves, 128 x 128 per wave, ONE wave per SIMD, accumulators in AGPRs, every
//                memory instruction placed by hand between MFMAs
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsrc/kernels benchmarks/micro/gemm256w_bench.hip
//        csrc/kernels/gemm256.hip csrc/kernels/gemm256w.hip -o /tmp/gemm256w_bench
// Run:   gemm256w_bench M N K   (random data; checks 4096 outputs against a host dot product)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

extern "C" int lc_gemm256w(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K, int n_pad,
                           int ldy, int epi, hipStream_t stream);
extern "C" int lc_gemm256(const void* Wp, const void* X, void* Y, const void* bias, int M, int N, int K, int n_pad,
                          int ldy, int epi, hipStream_t stream);

static float bf2f(uint16_t h) {
  uint32_t u = static_cast<uint32_t>(h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return static_cast<uint16_t>(u >> 16);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 14336;
  const int epi = argc > 4 ? atoi(argv[4]) : 0;
  const int KT = K / 32;
  std::vector<uint16_t> hw(static_cast<size_t>(N) * K), hx(static_cast<size_t>(M) * K);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (static_cast<float>(s >> 40) / 16777216.0f - 0.5f);
  };
  for (auto& v : hw) v = f2bf(rnd() * 0.1f);
  for (auto& v : hx) v = f2bf(rnd());
  // hw is the PACKED image: fragment (nt, kt) = 512 elements, lane l element j = W[nt*16 + (l&15)][kt*32 + 8*(l>>4) + j]
  auto w_at = [&](int n, int k) {
    const int nt = n / 16, kt = k / 32, l = (n % 16) + 16 * ((k % 32) / 8), j = k % 8;
    return bf2f(hw[(static_cast<size_t>(nt) * KT + kt) * 512 + l * 8 + j]);
  };
  uint16_t *dw, *dx, *dy0, *dy1;
  const int ldy = epi == 2 ? N / 2 : N;
  hipMalloc(&dw, hw.size() * 2);
  hipMalloc(&dx, hx.size() * 2);
  hipMalloc(&dy0, static_cast<size_t>(M) * ldy * 2);
  hipMalloc(&dy1, static_cast<size_t>(M) * ldy * 2);
  hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dy0, 0, static_cast<size_t>(M) * ldy * 2);
  hipMemset(dy1, 0, static_cast<size_t>(M) * ldy * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double flop = 2.0 * M * N * K;
  for (int which = 0; which < 2; ++which) {
    auto run = [&]() {
      return which == 0 ? lc_gemm256(dw, dx, dy0, nullptr, M, N, K, N, ldy, epi, nullptr)
                        : lc_gemm256w(dw, dx, dy1, nullptr, M, N, K, N, ldy, epi, nullptr);
    };
    for (int i = 0; i < 10; ++i)
      if (int rc = run()) { printf("launch %d failed: %d\n", which, rc); return 1; }
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    const int reps = 5, per = 40;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0);
      for (int i = 0; i < per; ++i) run();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = fminf(best, ms / per);
      tot += ms / per;
    }
    printf("{\"kernel\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"epi\": %d, \"us_best\": %.1f, \"us_mean\": %.1f, \"PF_mean\": %.3f}\n",
           which == 0 ? "gemm256" : "gemm256w", M, N, K, epi, best * 1e3, tot / reps * 1e3, flop / (tot / reps * 1e-3) / 1e15);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  std::vector<uint16_t> y0(static_cast<size_t>(M) * ldy), y1(y0.size());
  hipMemcpy(y0.data(), dy0, y0.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(y1.data(), dy1, y1.size() * 2, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < y0.size(); ++i) diff += y0[i] != y1[i];
  double worst = 0;
  if (epi == 0)
    for (int t = 0; t < 4096; ++t) {
      const int m = (t * 7919) % M, n = (t * 104729) % N;
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += static_cast<double>(w_at(n, k)) * bf2f(hx[static_cast<size_t>(m) * K + k]);
      worst = fmax(worst, fabs(acc - bf2f(y1[static_cast<size_t>(m) * ldy + n])) / (fabs(acc) + 1.0));
    }
  printf("{\"outputs_differing_from_gemm256\": %zu, \"of\": %zu, \"worst_rel_err_vs_host\": %.4g}\n", diff, y0.size(), worst);
  return (worst < 0.02) ? 0 : 2;
}
