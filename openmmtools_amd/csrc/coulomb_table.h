// Ewald direct-space force from a table in r^2 (force-only evaluations of the cluster-pair kernels, forces.hip).
//
//   F_i = fr (x_j - x_i),   fr = -q_i q_j G(u),   G(u) = (erfc(alpha r) / r + 2 alpha / sqrt(pi) exp(-alpha^2 u)) / u,  u = r^2
//
// (the derivative of OpenMM's NonbondedForce Ewald direct-space term, q_i q_j erfc(alpha r) / r; f64 restatement:
// oracle/forcefield.py).  Evaluated directly this costs v_rsq + v_exp + v_rcp (quarter rate) and ~20 more VALU instructions per
// lane pair; the table costs a bit-field extract, an and, a convert, one ds_read_b128 and three FMAs.
//
// Bins are logarithmic: the key of u is its IEEE-754 exponent followed by the CTAB_M leading mantissa bits, so a bin spans
// 2^-CTAB_M of its octave and G (close to a power law) has the same relative curvature in every bin.  Per bin one cubic in
// t = (the remaining CTAB_SHIFT mantissa bits, as an integer), interpolating G at the four Chebyshev nodes of the bin (f64 on
// the host); with CTAB_M = 5 the interpolation error is < 3e-8 of G (tests/test_coulomb_table.py), below the 1.5e-7 (absolute,
// in erfc) of the Abramowitz & Stegun 7.1.26 form it replaces and far below the 1e-5 u_kl contract.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define CTAB_M      5
#define CTAB_SHIFT  (23 - CTAB_M)
#define CTAB_MASK   ((1u << CTAB_SHIFT) - 1u)
#define CTAB_EMIN   (-8)          // r^2 below 2^-8 nm^2 (r < 0.0625 nm) is clamped: closer than any pair that is not excluded

struct coulomb_table_host {
    int key0 = 0, n = 0; float umin = 0.f;
    std::vector<float> c;         // [n][4]
};

static inline double ctab_G(double alpha, double u)
{
    const double r = std::sqrt(u);
    return (std::erfc(alpha * r) / r + 2.0 * alpha / std::sqrt(M_PI) * std::exp(-alpha * alpha * u)) / u;
}

// bins from 2^CTAB_EMIN up to (and including) the bin that holds rcc2
static inline coulomb_table_host ctab_build(double alpha, double rcc2)
{
    coulomb_table_host T;
    T.umin = std::ldexp(1.0f, CTAB_EMIN);
    uint32_t b0, b1; float f1 = (float)rcc2;
    std::memcpy(&b0, &T.umin, 4); std::memcpy(&b1, &f1, 4);
    T.key0 = (int)(b0 >> CTAB_SHIFT);
    T.n = (int)(b1 >> CTAB_SHIFT) - T.key0 + 1;
    if (T.n < 1) T.n = 1;
    T.c.resize((size_t)4 * T.n);
    for (int k = 0; k < T.n; ++k) {
        const uint32_t bits = (uint32_t)(T.key0 + k) << CTAB_SHIFT;
        float u0f; std::memcpy(&u0f, &bits, 4);
        const uint32_t bits_next = (uint32_t)(T.key0 + k + 1) << CTAB_SHIFT;
        float u1f; std::memcpy(&u1f, &bits_next, 4);
        const double u0 = u0f, h = (double)u1f - u0;
        // Newton form through the Chebyshev nodes of [0, 1], expanded to the power basis in t
        double t[4], y[4];
        for (int i = 0; i < 4; ++i) { t[i] = 0.5 - 0.5 * std::cos((2 * i + 1) * M_PI / 8.0); y[i] = -ctab_G(alpha, u0 + t[i] * h); }
        double d[4] = { y[0], y[1], y[2], y[3] };
        for (int lev = 1; lev < 4; ++lev) for (int i = 3; i >= lev; --i) d[i] = (d[i] - d[i - 1]) / (t[i] - t[i - lev]);
        double p[4] = { d[3], 0, 0, 0 };       // Horner expansion of d0 + (t-t0)(d1 + (t-t1)(d2 + (t-t2) d3))
        int deg = 0;
        for (int lev = 2; lev >= 0; --lev) {
            double q[4] = { 0, 0, 0, 0 };
            for (int i = 0; i <= deg; ++i) { q[i + 1] += p[i]; q[i] -= t[lev] * p[i]; }
            q[0] += d[lev];
            ++deg;
            for (int i = 0; i <= deg; ++i) p[i] = q[i];
        }
        const double s = std::ldexp(1.0, -CTAB_SHIFT);       // t = (mantissa remainder) * 2^-CTAB_SHIFT
        T.c[4 * k + 0] = (float)p[0];
        T.c[4 * k + 1] = (float)(p[1] * s);
        T.c[4 * k + 2] = (float)(p[2] * s * s);
        T.c[4 * k + 3] = (float)(p[3] * s * s * s);
    }
    return T;
}

// the device arithmetic, restated for the host (test hook remd_test_coulomb_table): -G(u) in f32
static inline float ctab_eval_host(const coulomb_table_host& T, float u)
{
    uint32_t bits; std::memcpy(&bits, &u, 4);
    const int k = (int)(bits >> CTAB_SHIFT) - T.key0;
    const float tf = (float)(bits & CTAB_MASK);
    const float* c = &T.c[4 * (size_t)k];
    return std::fmaf(tf, std::fmaf(tf, std::fmaf(tf, c[3], c[2]), c[1]), c[0]);
}
