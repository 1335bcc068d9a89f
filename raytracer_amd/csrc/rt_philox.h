// Philox4x32-10 counter-based RNG (Salmon et al., SC'11; Random123 constants), host + device.
// Stream convention (shared with the Node harness and the CPU oracle):
//   key = (seed_lo, seed_hi), counter = (draw >> 2, a, b, domain), U = (word[draw & 3] >> 8) * 2^-24.
// U has 24 bits, so it is exact in fp32 and fp64 alike and `U < threshold` can be decided on
// the integer k = word >> 8 against ceil(threshold * 2^24) with no rounding at all.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

struct Philox4 { uint32_t w[4]; };

RT_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3; return o;
}

// One stream; the current block is cached so that four consecutive draws cost one Philox call.
struct RngStream {
  uint32_t k0, k1, a, b, domain, draw, block_idx;
  Philox4 block;
  RT_HD void init(uint64_t seed, uint32_t domain_, uint32_t a_, uint32_t b_, uint32_t first_draw = 0) {
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32); a = a_; b = b_; domain = domain_;
    draw = first_draw; block_idx = 0xFFFFFFFFu;
  }
  // 24-bit integer k; U = k * 2^-24
  RT_HD uint32_t next_k() {
    uint32_t bi = draw >> 2;
    if (bi != block_idx) { block = philox4x32_10(bi, a, b, domain, k0, k1); block_idx = bi; }
    uint32_t sel = draw & 3;
    uint32_t w = sel == 0 ? block.w[0] : sel == 1 ? block.w[1] : sel == 2 ? block.w[2] : block.w[3];
    draw++;
    return w >> 8;
  }
  template <typename T> RT_HD T next() { return (T)next_k() * (T)(1.0 / 16777216.0); }
};

// threshold for `U < x` on the integer draw: k < ceil(x * 2^24); NaN (undefined) -> never true
inline uint32_t rt_threshold_u24(double x) {
  if (!(x == x)) return 0;
  if (x <= 0) return 0;
  if (x >= 1) return 1u << 24;
  double s = x * 16777216.0;
  uint32_t k = (uint32_t)s;
  return ((double)k < s) ? k + 1 : k;
}
