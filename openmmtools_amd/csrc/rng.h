// Counter-based RNG (Philox4x32-10) and the deterministic f64 exp used by the
// Metropolis tests.  Shared by host and device code of libremd_hip.so.
//
// RNG stream spec: key = (seed_lo, seed_hi);
//   counter = (a, b, (u32)t, stream ^ ((u32)(t >> 32) << 8))
// where (a, b, t) are stream-specific (see the REMD_STREAM_* comments).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define REMD_HD __host__ __device__ __forceinline__
#else
#define REMD_HD static inline
#endif

#define REMD_STREAM_SWAP_ALL  1u  // a,b = attempt index lo,hi ; t = iteration
#define REMD_STREAM_NEIGHBOR  2u  // a = 0: offset draw; a = 1+state_i: pair uniform ; t = iteration
#define REMD_STREAM_SAMS      3u  // a = replica ; t = iteration
#define REMD_STREAM_VELOCITY  4u  // a = atom, b = global replica ; t = iteration
#define REMD_STREAM_OU        5u  // a = atom, b = global replica ; t = global O-substep counter
#define REMD_STREAM_METROPOLIS 7u // a = index of the '}' inside the step program, b = global replica ; t = global step
#define REMD_STREAM_BAROSTAT  6u  // a = 0: volume draw, 1: acceptance uniform ; b = global replica ; t = attempt counter

struct philox4 { uint32_t w[4]; };

REMD_HD uint32_t remd_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

REMD_HD philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                              uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = remd_mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = remd_mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    philox4 o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

REMD_HD philox4 remd_philox(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t t) {
    return philox4x32_10(a, b, (uint32_t)t, stream ^ ((uint32_t)(t >> 32) << 8),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 53-bit uniform in [0,1) from two words
REMD_HD double remd_u53(uint32_t hi, uint32_t lo) {
    uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)m * (1.0 / 9007199254740992.0);
}
// 23-bit uniform strictly inside (0,1), exactly representable in f32
REMD_HD float remd_u23(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// Deterministic exp(x) for x <= 0 (swap acceptance): only IEEE fma/mul/add/rint, so the
// host oracle and the gfx950 kernel produce bit-identical results.
REMD_HD double remd_exp_det(double x) {
    if (!(x > -700.0)) return 0.0;
    if (x > 0.0) x = 0.0;
    double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                 // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);     // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);    // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);    // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);     // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);    // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);    // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);    // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int64_t ki = (int64_t)k;                           // -1010 <= ki <= 0
    union { uint64_t u; double d; } s;
    s.u = (uint64_t)(ki + 1023) << 52;
    return p * s.d;
}
