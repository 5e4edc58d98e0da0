// Seedable PRNG for proposer / refiner selection.
//
// The reference picks the proposer and the refiner with an unseeded thread_rng()
// (/root/reference/src/main.rs:229, :272), so its runs cannot be replayed. Every
// random choice in this engine goes through this generator so a session is a pure
// function of (seed, member replies).
#pragma once
#include <cstdint>

namespace lc {

class SplitMix64 {
 public:
  explicit SplitMix64(uint64_t seed = 0x9E3779B97F4A7C15ull) : s_(seed) {}
  uint64_t next() {
    uint64_t z = (s_ += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // Uniform integer in [0, n) without modulo bias (Lemire's multiply-shift with rejection).
  uint64_t below(uint64_t n) {
    if (n <= 1) return 0;
    uint64_t x = next();
    __uint128_t m = static_cast<__uint128_t>(x) * n;
    uint64_t lo = static_cast<uint64_t>(m);
    if (lo < n) {
      uint64_t t = (0 - n) % n;
      while (lo < t) {
        x = next();
        m = static_cast<__uint128_t>(x) * n;
        lo = static_cast<uint64_t>(m);
      }
    }
    return static_cast<uint64_t>(m >> 64);
  }
  double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t state() const { return s_; }
  void reseed(uint64_t seed) { s_ = seed; }

 private:
  uint64_t s_;
};

}  // namespace lc
