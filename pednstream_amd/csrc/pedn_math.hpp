// pedn_math.hpp -- device-side arithmetic primitives of the HIP engine (gfx950).
//
// Everything here must be bit-identical to what the reference computes on the host through numpy/libm:
//   * pedn_powf   : glibc 2.35 powf algorithm (numpy float32 scalar power == libm powf; link.py:212,317)
//   * pedn_exp    : glibc 2.35 exp (FMA build) restated (softmax, path_finder.py:585)
//   * Philox4x32-10 keyed (seed, replica, link, t, site) -> binomial / normal  (RNG contract)
// Only IEEE-754 +,-,*,/,sqrt,floor on binary32/binary64 are used; the translation unit is compiled with
// -ffp-contract=off so that no multiply-add is fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pedn {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

// ---- glibc powf tables (__powf_log2_data, __exp2f_data) ---------------------------------------------------
__device__ const double kLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
__device__ const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// log2(x) in binary64 as glibc's powf computes it (log2_inline); x > 0 finite
__device__ __forceinline__ double pedn_powf_log2(float x) {
  uint32_t ix = f2u(x);
  if (ix < 0x00800000u) {
    ix = f2u(x * 0x1p23f);
    ix -= 23u << 23;
  }
  uint32_t tmp = ix - 0x3f330000u;
  int i = (int)((tmp >> 19) & 15u);
  uint32_t top = tmp & 0xff800000u;
  uint32_t iz = ix - top;
  int k = (int32_t)top >> 23;
  double invc = kLog2Tab[i][0], logc = kLog2Tab[i][1];
  double z = (double)u2f(iz);
  double r = z * invc - 1.0;
  double y0 = logc + (double)k;
  double r2 = r * r;
  double yy = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
  double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
  double r4 = r2 * r2;
  double q = 0x1.71547652ab82bp0 * r + y0;
  q = p * r2 + q;
  return yy * r4 + q;
}

// 2^(ylogx) rounded to binary32 as glibc's powf does (exp2_inline)
__device__ __forceinline__ float pedn_powf_exp2(double ylogx) {
  if (ylogx <= -150.0) return 0.0f;
  double kd = ylogx + 0x1.8p+47;
  uint64_t ki = d2u(kd);
  kd -= 0x1.8p+47;
  double rr = ylogx - kd;
  uint64_t t = kExp2Tab[ki & 31u];
  t += ki << 47;
  double s = u2d(t);
  double zz = 0x1.c6af84b912394p-5 * rr + 0x1.ebfce50fac4f3p-3;
  double rr2 = rr * rr;
  double e = 0x1.62e42ff0c52d6p-1 * rr + 1.0;
  e = zz * rr2 + e;
  e = e * s;
  return (float)e;
}

// x >= 0 finite, y > 0 finite
__device__ inline float pedn_powf(float x, float y) {
  if (x == 0.0f) return 0.0f;
  return pedn_powf_exp2((double)y * pedn_powf_log2(x));
}

// x**2 and x**3 of the diffusion weights (link.py:211-212) share one logarithm
__device__ inline void pedn_powf_2_3(float x, float& p2, float& p3) {
  if (x == 0.0f) { p2 = p3 = 0.0f; return; }
  const double lg = pedn_powf_log2(x);
  p2 = pedn_powf_exp2(2.0 * lg);
  p3 = pedn_powf_exp2(3.0 * lg);
}

// ---- exp ------------------------------------------------------------------------------------------------------
// glibc 2.35 __exp_fma restated (N = 128 table, degree-5 polynomial, the shipped binary's fused multiply-adds); see
// oracle/pedn_oracle.c pw_exp for the derivation.  Main path 2^-54 <= |x| < 512; the softmax never leaves it.
__device__ const uint64_t kExpTab[128][2] = {
#include "exp_table.inc"
};

// main path of pedn_exp without its branches: valid when `special` comes back false (2^-54 <= |x| < 512); several of these in
// a row stay one straight-line block, so their table look-ups and dependent chains overlap
__device__ __forceinline__ double pedn_exp_main(double x, bool& special) {
  const uint32_t abstop = (uint32_t)(d2u(x) >> 52) & 0x7ffu;
  special = abstop - 0x3c9u >= 0x3fu;
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  double kd = fma(x, InvLn2N, Shift);
  uint64_t ki = d2u(kd);
  kd -= Shift;
  double r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
  uint64_t idx = ki & 127u;
  double tail = u2d(kExpTab[idx][0]);
  uint64_t sbits = kExpTab[idx][1] + (ki << 45);
  double r2 = r * r;
  double tmp = fma(r2 * r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r));
  double scale = u2d(sbits);
  return fma(scale, tmp, scale);
}

__device__ inline double pedn_exp(double x) {
  uint32_t abstop = (uint32_t)(d2u(x) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
    if (x != x) return x;
    if (abstop >= 0x409u) return (d2u(x) >> 63) ? 0.0 : __longlong_as_double(0x7ff0000000000000ll);
    return exp(x);  // 512 <= |x| < 1024: not reachable from the softmax, not bit-pinned
  }
  bool special;
  return pedn_exp_main(x, special);
}

// ---- RNG contract -----------------------------------------------------------------------------------------------
struct RngKey {
  uint32_t k0, k1;  // seed lo / hi
  uint32_t replica, link, t, site;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply (v_mad_u64_u32) per product instead of separate mul_hi / mul_lo
    uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ void rng_words(const RngKey& k, uint32_t call, uint32_t w[4]) {
  w[0] = k.t; w[1] = k.link; w[2] = k.site | (call << 8); w[3] = k.replica;
  philox4x32_10(w, k.k0, k.k1);
}

// Irwin-Hall(8) over the eight 16-bit halves of ONE Philox call; 0x1.3988e1409212ep-16 = sqrt(1.5) * 2^-16
__device__ inline double rng_z(const RngKey& k) {
  uint32_t w[4];
  rng_words(k, 0u, w);
  uint32_t s = (w[0] & 0xffffu) + (w[0] >> 16) + (w[1] & 0xffffu) + (w[1] >> 16) + (w[2] & 0xffffu) + (w[2] >> 16) + (w[3] & 0xffffu) + (w[3] >> 16);
  return (double)((int)s - 262140) * 0x1.3988e1409212ep-16;
}

// rng_binomial with Philox call 0 of the key already drawn into w (by a caller that had a load in flight to draw it under)
__device__ inline double rng_binomial_w(long long n, double p, const RngKey& k, uint32_t (&w)[4], int meanfield) {
  if (n <= 0 || p <= 0.0) return 0.0;
  if (meanfield) return floor((double)n * p);
  if (p >= 1.0) return (double)n;
  if (n <= 16) {
    const uint32_t thr = (uint32_t)floor(p * 65536.0);
    const int nn = (int)n;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
      cnt += (i < nn) && (h < thr);
    }
    if (nn > 8) {
      rng_words(k, 1u, w);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t h = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        cnt += (i + 8 < nn) && (h < thr);
      }
    }
    return (double)cnt;
  }
  const uint32_t hs = (w[0] & 0xffffu) + (w[0] >> 16) + (w[1] & 0xffffu) + (w[1] >> 16) + (w[2] & 0xffffu) + (w[2] >> 16) + (w[3] & 0xffffu) + (w[3] >> 16);
  const double z = (double)((int)hs - 262140) * 0x1.3988e1409212ep-16;  // rng_z
  double mean = (double)n * p;
  double sd = sqrt(mean * (1.0 - p));
  double x = floor(mean + sd * z + 0.5);
  if (x < 0.0) x = 0.0;
  if (x > (double)n) x = (double)n;
  return x;
}

__device__ inline double rng_binomial(long long n, double p, const RngKey& k, int meanfield) {
  if (n <= 0 || p <= 0.0) return 0.0;
  if (meanfield) return floor((double)n * p);
  if (p >= 1.0) return (double)n;
  // both branches start from Philox call 0: drawn once, ahead of the branch (lanes of one wave usually take both)
  uint32_t w[4];
  rng_words(k, 0u, w);
  return rng_binomial_w(n, p, k, w, 0);
}

}  // namespace pedn
