// Langevin splitting integrator kernels (gfx950): V / R / O substeps, constraints,
// Maxwell-Boltzmann velocity draw, kinetic-energy reduction, CM-motion removal.
//
// Reference semantics (restated):
//   openmmtools/integrators.py:1404-1423  R: x += (dt/n_R) v ; constrain x ; v += (x - x1)/(dt/n_R) ; constrain v
//   openmmtools/integrators.py:1425-1446  V: v += (dt/n_V) f/m ; constrain v
//   openmmtools/integrators.py:1448-1460  O: v = a v + b sigma xi ; constrain v,  a = exp(-gamma h), b = sqrt(1-a^2),
//                                            h = dt/max(1,n_O) (:1142-1146), sigma = sqrt(kT/m) (:1314)
//   openmmtools/mcmc.py:710-711           setVelocitiesToTemperature (+ velocity constraints)
//   openmmtools/integrators.py:1313       addUpdateContextState (CMMotionRemover acts here, once per step)
//
// Design: one thread owns one *constraint unit* (a rigid water, an X-H cluster, or a free
// atom), keeps its <= 4 atoms' x and v in registers and runs the whole chain of substeps
// between two force evaluations without touching HBM in between.  All replicas of the
// rank are covered by one launch (blockIdx.y = replica).
#include "remd_internal.h"
#include "rng.h"
#include <set>
#include "pair_math.h"
#include "listed_terms.h"
#include "nocutoff_pair.h"

#define UNIT_FREE   0
#define UNIT_SETTLE 1
#define UNIT_SHAKE  2
#define MAX_TOK 24
#define CHAIN_BIN_COLS 256       // mesh columns (nx) a chain workgroup can bin in its LDS counters

struct chain_prog {
    int n;                 // tokens in this chain
    char tok[MAX_TOK];     // 'V','R','O','C' (C = subtract centre-of-mass velocity)
    int o_index[MAX_TOK];  // for 'O': index of this O inside the step program
    long long step[MAX_TOK]; // global step index of each token (a chain may hold tokens left pending by the previous step)
    float hV, hR;          // dt/n_V, dt/n_R
    float hVg[4];          // multiple-time-step splittings: dt / (V tokens of force group g); tokens '0'..'3'
    const long long* Fg[4]; // forces of force group g ([R][3][Npad] fixed point, like the all-forces accumulator)
    float a, b;            // OU coefficients
    int nO;
    int accumulate_momentum;  // after the chain, add sum(m v) into cmm buffer cmm_w
    int cmm_w, cmm_r;         // double-buffered momentum accumulators: 'C' reads cmm_r and clears the other one
    int zero_force;           // the chain ends with stale forces (an R after its last V): clear them for the next evaluation
    int m_buf;                // token 'M' (inside a chain): every workgroup of the replica publishes its partial sum(m v) as epoch-tagged 64-bit
    unsigned int m_epoch;     // words (m_epoch-th exchange of this handle, d_chain_sync) and reads its siblings': the 'C' behind it has the sum
    int measure;              // bit 0: heat (kinetic-energy change of the O substeps), bit 1: kinetic part of the shadow work (V, R substeps)
};

// token t of the program, from registers: a dynamic index into the kernel-argument array is a scalar memory load per token on the
// chain's critical path (the chain is a handful of wavefronts waiting for one thing after another: profiles/r05_15_chain_segments.txt);
// the six words are loaded once with the other arguments
__device__ __forceinline__ char chain_tok(const chain_prog& prog, int t)
{
    static_assert(MAX_TOK == 24, "six 32-bit words of tokens");
    const unsigned int* w = reinterpret_cast<const unsigned int*>(prog.tok);
    const unsigned int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5];
    const int q = t >> 2;
    const unsigned int x = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : q == 3 ? w3 : q == 4 ? w4 : w5;
    return (char)((x >> ((t & 3) * 8)) & 0xffu);
}

// state of one constraint unit between the segments of a chain (registers)
struct unit_regs { float3 x[4], v[4]; float im[4]; float heat, shadow;
    int shake_it;                            // most Newton updates a position solve of this unit needed in this launch (X-H clusters)
    float cmx, cmy, cmz; int have_cm;        // centre-of-mass velocity from an 'M' token of this launch, for the 'C' that follows it
#ifdef CHAIN_STAMPS
    unsigned long long* stamps; unsigned long long t_last;     // tools/chain_segments.py: per-token wall-clock of workgroup (0, 0)
#endif
};

struct settle_const { float mO, mH, ra, rb, rc, dOH, dHH; };

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// One wavefront per SIMD at best (a few hundred waves per launch): the chain kernel is bound by the LATENCY of its
// dependent arithmetic, so the 1-ulp hardware reciprocal / square root / reciprocal square root replace the IEEE
// expansions (~10 dependent instructions each) of '/', sqrtf and rsqrtf; fp32 SETTLE is ~1e-7 relative either way.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// Analytic SETTLE (Miyamoto & Kollman 1992) in coordinates relative to the old O position:
// p0[] old (constrained) positions relative to A0 (p0[0] = 0), p1[] unconstrained new
// positions relative to A0.  Returns constrained new positions (relative to A0) in p1.
__device__ __forceinline__ void settle_positions(const settle_const& sc, const float3* p0, float3* p1)
{
    const float3 b0 = p0[1], c0 = p0[2];
    const float M = sc.mO + 2.f * sc.mH;
    const float3 d0 = (p1[0] * sc.mO + p1[1] * sc.mH + p1[2] * sc.mH) * frcp(M);
    const float3 a1 = p1[0] - d0, b1 = p1[1] - d0, c1 = p1[2] - d0;
    float3 Z = cross3(b0, c0);
    float3 X = cross3(a1, Z);
    float3 Y = cross3(Z, X);
    X = X * frsq(dot3(X, X)); Y = Y * frsq(dot3(Y, Y)); Z = Z * frsq(dot3(Z, Z));
    const float xb0 = dot3(X, b0), yb0 = dot3(Y, b0);
    const float xc0 = dot3(X, c0), yc0 = dot3(Y, c0);
    const float za1 = dot3(Z, a1);
    const float xb1 = dot3(X, b1), yb1 = dot3(Y, b1), zb1 = dot3(Z, b1);
    const float xc1 = dot3(X, c1), yc1 = dot3(Y, c1), zc1 = dot3(Z, c1);
    const float sinphi = za1 * frcp(sc.ra);
    const float cosphi = fsqrt(fmaxf(0.f, 1.f - sinphi * sinphi));
    const float sinpsi = (zb1 - zc1) * frcp(2.f * sc.rc * cosphi);
    const float cospsi = fsqrt(fmaxf(0.f, 1.f - sinpsi * sinpsi));
    const float ya2 = sc.ra * cosphi;
    const float xb2 = -sc.rc * cospsi;
    const float yb2 = -sc.rb * cosphi - sc.rc * sinpsi * sinphi;
    const float yc2 = -sc.rb * cosphi + sc.rc * sinpsi * sinphi;
    const float alpha = xb2 * (xb0 - xc0) + yb0 * yb2 + yc0 * yc2;
    const float beta  = xb2 * (yc0 - yb0) + xb0 * yb2 + xc0 * yc2;
    const float gamma = xb0 * yb1 - xb1 * yb0 + xc0 * yc1 - xc1 * yc0;
    const float al2be2 = alpha * alpha + beta * beta;
    const float sintheta = (alpha * gamma - beta * fsqrt(fmaxf(0.f, al2be2 - gamma * gamma))) * frcp(al2be2);
    const float costheta = fsqrt(fmaxf(0.f, 1.f - sintheta * sintheta));
    const float xa3 = -ya2 * sintheta, ya3 = ya2 * costheta, za3 = za1;
    const float xb3 = xb2 * costheta - yb2 * sintheta, yb3 = xb2 * sintheta + yb2 * costheta, zb3 = zb1;
    const float xc3 = -xb2 * costheta - yc2 * sintheta, yc3 = -xb2 * sintheta + yc2 * costheta, zc3 = zc1;
    p1[0] = X * xa3 + Y * ya3 + Z * za3 + d0;
    p1[1] = X * xb3 + Y * yb3 + Z * zb3 + d0;
    p1[2] = X * xc3 + Y * yc3 + Z * zc3 + d0;
}

// Analytic velocity constraint for a rigid triangle: remove the relative velocity along the
// three bonds by solving the 3x3 Lagrange-multiplier system (Cramer's rule).
__device__ __forceinline__ void settle_velocities(float imA, float imB, float imC, const float3* p, float3* v)
{
    float3 eAB = p[1] - p[0], eBC = p[2] - p[1], eCA = p[0] - p[2];
    eAB = eAB * frsq(dot3(eAB, eAB)); eBC = eBC * frsq(dot3(eBC, eBC)); eCA = eCA * frsq(dot3(eCA, eCA));
    const float dAB = dot3(v[1] - v[0], eAB), dBC = dot3(v[2] - v[1], eBC), dCA = dot3(v[0] - v[2], eCA);
    const float cAB_BC = dot3(eAB, eBC), cAB_CA = dot3(eAB, eCA), cBC_CA = dot3(eBC, eCA);
    const float m00 = imA + imB,        m01 = -cAB_BC * imB,  m02 = -cAB_CA * imA;
    const float m10 = -cAB_BC * imB,    m11 = imB + imC,      m12 = -cBC_CA * imC;
    const float m20 = -cAB_CA * imA,    m21 = -cBC_CA * imC,  m22 = imC + imA;
    const float det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
    const float idet = frcp(det);
    const float tAB = (dAB * (m11 * m22 - m12 * m21) - m01 * (dBC * m22 - m12 * dCA) + m02 * (dBC * m21 - m11 * dCA)) * idet;
    const float tBC = (m00 * (dBC * m22 - m12 * dCA) - dAB * (m10 * m22 - m12 * m20) + m02 * (m10 * dCA - dBC * m20)) * idet;
    const float tCA = (m00 * (m11 * dCA - dBC * m21) - m01 * (m10 * dCA - dBC * m20) + dAB * (m10 * m21 - m11 * m20)) * idet;
    v[0] = v[0] + (eAB * tAB - eCA * tCA) * imA;
    v[1] = v[1] + (eBC * tBC - eAB * tAB) * imB;
    v[2] = v[2] + (eCA * tCA - eBC * tBC) * imC;
}

// K x K linear solve (K <= 3: the constraints of one X-H star cluster), Cramer's rule, everything in registers
template <int K>
__device__ __forceinline__ void solve_small(const float (&A)[3][3], const float (&b)[3], float (&x)[3])
{
    if (K == 1) {
        x[0] = b[0] * frcp(A[0][0]); x[1] = 0.f; x[2] = 0.f;
    } else if (K == 2) {
        const float idet = frcp(A[0][0] * A[1][1] - A[0][1] * A[1][0]);
        x[0] = (b[0] * A[1][1] - A[0][1] * b[1]) * idet;
        x[1] = (A[0][0] * b[1] - b[0] * A[1][0]) * idet;
        x[2] = 0.f;
    } else {
        const float c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][0] * A[2][2] - A[1][2] * A[2][0], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
        const float idet = frcp(A[0][0] * c00 - A[0][1] * c01 + A[0][2] * c02);
        x[0] = (b[0] * c00 - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) + A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2])) * idet;
        x[1] = (A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * c01 + A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0])) * idet;
        x[2] = (A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) + b[0] * c02) * idet;
    }
}

// Position constraints of a star cluster (central atom 0 bonded to atoms 1..NAT-1): p0 old constrained positions, p1
// unconstrained new positions (both relative to the old central atom).  The SHAKE displacements act along the OLD bond
// vectors r0_q with one multiplier per bond; instead of Gauss-Seidel sweeps over the bonds (6-10 sweeps, data dependent,
// and the one wavefront holding the solute's clusters used to set the duration of the whole integrator launch) the K x K
// system  |s_q + sum_p B_qp lam_p r0_p|^2 = d_q^2,  B_qp = 1/m_0 + delta_qp / m_q,  is solved by Newton iterations with the
// exact Jacobian (quadratic convergence: a half step moves bond lengths by < 1 %, so two iterations reach fp32 round-off).
// Round 6: the iteration runs until every bond of the cluster is within the integrator's constraint tolerance (integrators.py:1416-1418:
// addConstrainPositions works to getConstraintTolerance(), a RELATIVE distance error) -- | |r|^2 - d^2 | <= 2 tol d^2 -- with tol no
// smaller than what fp32 lengths can hold (the caller passes max(tol, 2e-7)) and at most SHAKE_MAX_IT updates; before it was a fixed
// three updates whatever the tolerance.  Returns the number of updates made, SHAKE_MAX_IT + 1 when the bound was reached unconverged.
// NAT is a compile-time constant so that every array lives in registers (no scratch).
#define SHAKE_MAX_IT 8
template <int NAT>
__device__ __forceinline__ int shake_positions(const float* im, const float* d, float tol, const float3* p0, float3* p1)
{
    constexpr int K = NAT - 1;
    float3 r0[3], sv[3];
    float lam[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        r0[q] = q < K ? p0[q + 1] - p0[0] : f3(0, 0, 0);
        sv[q] = q < K ? p1[q + 1] - p1[0] : f3(0, 0, 0);
    }
    const float tol2 = 2.f * tol;
    int it = 0;
    for (;; ++it) {
        float3 acc = f3(0, 0, 0);                                   // im0 * sum_p lam_p r0_p (the central atom's share)
#pragma unroll
        for (int p = 0; p < K; ++p) acc = acc + r0[p] * (lam[p] * im[0]);
        float J[3][3], g[3], dl[3];
        bool converged = true;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < K) {
                const float3 cur = sv[q] + acc + r0[q] * (lam[q] * im[q + 1]);
                const float d2 = d[q] * d[q];
                g[q] = d2 - dot3(cur, cur);
                converged = converged && fabsf(g[q]) <= tol2 * d2;
#pragma unroll
                for (int p = 0; p < 3; ++p) J[q][p] = p < K ? 2.f * dot3(cur, r0[p]) * (im[0] + (p == q ? im[q + 1] : 0.f)) : 0.f;
            } else {
                g[q] = 0.f;
#pragma unroll
                for (int p = 0; p < 3; ++p) J[q][p] = p == q ? 1.f : 0.f;
            }
        }
        if (converged) break;
        if (it == SHAKE_MAX_IT) { it = SHAKE_MAX_IT + 1; break; }
        solve_small<K>(J, g, dl);
#pragma unroll
        for (int q = 0; q < K; ++q) lam[q] += dl[q];
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
        p1[0] = p1[0] - r0[q] * (lam[q] * im[0]);
        p1[q + 1] = p1[q + 1] + r0[q] * (lam[q] * im[q + 1]);
    }
    return it;
}

// Velocity constraints of a star cluster: the multipliers solve a K x K LINEAR system exactly (no iteration):
//   sum_p (1/m_0 r_q.r_p + delta_qp r_q.r_q / m_q) mu_p = r_q . (v_q - v_0)
template <int NAT>
__device__ __forceinline__ void shake_velocities(const float* im, float /*tol*/, const float3* p, float3* v)
{
    constexpr int K = NAT - 1;
    float3 r[3];
    float A[3][3], b[3], mu[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) r[q] = q < K ? p[q + 1] - p[0] : f3(0, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        b[q] = q < K ? dot3(r[q], v[q + 1] - v[0]) : 0.f;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
            A[q][pp] = (q < K && pp < K) ? dot3(r[q], r[pp]) * (im[0] + (pp == q ? im[q + 1] : 0.f)) : (pp == q ? 1.f : 0.f);
    }
    solve_small<K>(A, b, mu);
#pragma unroll
    for (int q = 0; q < K; ++q) {
        v[0] = v[0] + r[q] * (mu[q] * im[0]);
        v[q + 1] = v[q + 1] - r[q] * (mu[q] * im[q + 1]);
    }
}

template <int TYPE, int NAT>
__device__ __forceinline__ void constrain_v(const settle_const& sc, const float* im, float tol, float3* v, const float3* x)
{
    if (TYPE == UNIT_SETTLE) {
        float3 p[3] = { f3(0, 0, 0), x[1] - x[0], x[2] - x[0] };
        settle_velocities(im[0], im[1], im[2], p, v);
    } else if (TYPE == UNIT_SHAKE) {
        float3 p[NAT];
#pragma unroll
        for (int k = 0; k < NAT; ++k) p[k] = x[k] - x[0];
        shake_velocities<NAT>(im, tol, p, v);
    }
}

__device__ __forceinline__ float3 gaussian3(uint64_t seed, uint32_t stream, uint32_t atom, uint32_t replica, uint64_t t)
{
    philox4 w = remd_philox(seed, stream, atom, replica, t);
    const float r1 = fsqrt(-2.f * __logf(remd_u23(w.w[0])));
    const float r2 = fsqrt(-2.f * __logf(remd_u23(w.w[2])));
    float s1, c1, s2, c2;
    __sincosf(6.2831853071795865f * remd_u23(w.w[1]), &s1, &c1);
    __sincosf(6.2831853071795865f * remd_u23(w.w[3]), &s2, &c2);
    (void)s2;
    return f3(r1 * c1, r1 * s1, r2 * c2);
}

// the whole chain of substeps for one constraint unit, everything in registers
template <int TYPE, int NAT, bool LOADS>
__device__ __forceinline__ float3 run_unit(const chain_prog& prog, const int* idx, const float* dist, const settle_const& sc,
                                          float tol, int Npad, float4* __restrict__ P, float4* __restrict__ V,
                                          const long long* F, long long* Fw, const float* __restrict__ invmass, float kT,
                                          uint32_t rg, uint64_t seed, const long long* __restrict__ cmm_r, float inv_total_mass,
                                          const remd_chain_bins& bins, int r,
                                          unit_regs& S, int t0, int t1, bool first, bool last,
                                          float4* __restrict__ Xold, float4* __restrict__ Vold)
{
    float3 (&x)[4] = S.x; float3 (&v)[4] = S.v;
    float (&im)[4] = S.im;
    // kinetic energy of this unit's atoms (integrators.py:1141: 0.5 m v^2 summed over the degrees of freedom)
    auto unit_ke = [&]() { float ke = 0.f;
#pragma unroll
        for (int k = 0; k < NAT; ++k) ke += 0.5f * dot3(v[k], v[k]) * frcp(im[k]);
        return ke; };
    if (first) {
        S.heat = 0.f; S.shadow = 0.f;
        if (LOADS) {
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                const float4 p = P[idx[k]], w = V[idx[k]];
                x[k] = f3(p.x, p.y, p.z); v[k] = f3(w.x, w.y, w.z);
                im[k] = invmass[idx[k]];
            }
        }
#ifdef CHAIN_STAMPS
        if (S.stamps) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + 21], now - S.t_last); S.t_last = now; }
#endif
    }
    for (int t = t0; t < t1; ++t) {
        const char tok = chain_tok(prog, t);
        const bool is_v = tok == 'V' || (tok >= '0' && tok <= '3');
        const float ke0 = ((prog.measure & 1) && tok == 'O') || ((prog.measure & 2) && (is_v || tok == 'R')) ? unit_ke() : 0.f;
        if (tok == '{') {
            // Metropolization starts: remember x and v (integrators.py:1539-1542)
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                Xold[idx[k]] = make_float4(x[k].x, x[k].y, x[k].z, 0.f);
                Vold[idx[k]] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
            }
        } else if (tok == 'V' || (tok >= '0' && tok <= '3')) {
            // all forces, or the forces of one force group with that group's share of the time step (integrators.py:1437-1440)
            const long long* Ft = tok == 'V' ? F : prog.Fg[tok - '0'] + (size_t)r * 3 * Npad;
            const float hv = tok == 'V' ? prog.hV : prog.hVg[tok - '0'];
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                const float s = hv * im[k] * (1.0f / 4294967296.0f);
                v[k].x += s * (float)Ft[idx[k]];
                v[k].y += s * (float)Ft[Npad + idx[k]];
                v[k].z += s * (float)Ft[2 * Npad + idx[k]];
            }
#ifdef CHAIN_STAMPS
            if (S.stamps && t == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + 22], now - S.t_last); S.t_last = now; }
#endif
            constrain_v<TYPE, NAT>(sc, im, tol, v, x);
        } else if (tok == 'R') {
            if (TYPE == UNIT_FREE) {
#pragma unroll
                for (int k = 0; k < NAT; ++k) x[k] = x[k] + v[k] * prog.hR;
            } else {
                // relative coordinates (origin = old position of atom 0) keep fp32 precision
                float3 p0[NAT], p1[NAT], q[NAT];
#pragma unroll
                for (int k = 0; k < NAT; ++k) {
                    p0[k] = x[k] - x[0];
                    p1[k] = p0[k] + v[k] * prog.hR;
                    q[k] = p1[k];
                }
                if (TYPE == UNIT_SETTLE) settle_positions(sc, p0, p1);
                else S.shake_it = max(S.shake_it, shake_positions<NAT>(im, dist, tol, p0, p1));
                const float ih = frcp(prog.hR);
                const float3 org = x[0];
#pragma unroll
                for (int k = 0; k < NAT; ++k) {
                    v[k] = v[k] + (p1[k] - q[k]) * ih;          // integrators.py:1417
                    x[k] = org + p1[k];
                }
                constrain_v<TYPE, NAT>(sc, im, tol, v, x);
            }
        } else if (tok == 'O') {
            const uint64_t cnt = (uint64_t)prog.step[t] * (uint64_t)prog.nO + (uint64_t)prog.o_index[t];
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                const float3 xi = gaussian3(seed, REMD_STREAM_OU, (uint32_t)idx[k], rg, cnt);
                const float sig = prog.b * fsqrt(kT * im[k]);
                v[k].x = prog.a * v[k].x + sig * xi.x;
                v[k].y = prog.a * v[k].y + sig * xi.y;
                v[k].z = prog.a * v[k].z + sig * xi.z;
            }
            constrain_v<TYPE, NAT>(sc, im, tol, v, x);
        } else if (tok == 'C') {
            // CMMotionRemover: v -= P/M with P accumulated by the previous chain or by the 'M' token in front (read at the
            // coherence point: other workgroups added to it by atomics during this launch)
            float sx, sy, sz;
            if (S.have_cm) { sx = S.cmx; sy = S.cmy; sz = S.cmz; }
            else {
                sx = (float)__hip_atomic_load(&cmm_r[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (1.0f / 4294967296.0f) * inv_total_mass;
                sy = (float)__hip_atomic_load(&cmm_r[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (1.0f / 4294967296.0f) * inv_total_mass;
                sz = (float)__hip_atomic_load(&cmm_r[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (1.0f / 4294967296.0f) * inv_total_mass;
            }
#pragma unroll
            for (int k = 0; k < NAT; ++k) { v[k].x -= sx; v[k].y -= sy; v[k].z -= sz; }
        }
#ifdef CHAIN_STAMPS
        if (S.stamps) { const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + min(t, 33)], now - S.t_last); S.t_last = now; }
#endif
        if ((prog.measure & 1) && tok == 'O') S.heat += unit_ke() - ke0;                           // :1448-1460
        if ((prog.measure & 2) && (is_v || tok == 'R')) S.shadow += unit_ke() - ke0;                // :1409-1446 (kinetic part)
    }
    float3 mom = f3(0, 0, 0);
    if (!last) return mom;
#pragma unroll
    for (int k = 0; k < NAT; ++k) {
        P[idx[k]] = make_float4(x[k].x, x[k].y, x[k].z, 0.f);
        V[idx[k]] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
        mom = mom + v[k] * frcp(im[k]);
        if (prog.zero_force) { Fw[idx[k]] = 0; Fw[Npad + idx[k]] = 0; Fw[2 * Npad + idx[k]] = 0; }
        if (bins.count) {
            // these positions are final for the force evaluation that follows: bin the atom by its PME mesh column here, so
            // that no binning launch sits between the integrator and the spreading pass (the order inside a bin is
            // irrelevant: charges and forces are fixed-point sums)
            float u; int kx;
            remd_pme_scaled1(x[k].x, bins.box[4 * r], bins.nx, u, kx);
            if (kx >= bins.nx) kx -= bins.nx;
            const int slot = atomicAdd(&bins.count[(size_t)r * bins.nx + kx], 1);
            if (slot < bins.cap) {
                const size_t e = ((size_t)r * bins.nx + kx) * bins.cap + slot;
                bins.atoms[e] = make_float4(x[k].x, x[k].y, x[k].z, __int_as_float(idx[k]));
                if (bins.q) bins.q[e] = bins.param[idx[k]].x;        // (state-independent charges only, see remd_pme_chain_bins)
            } else atomicCAS(bins.err, 0u, 2u);
        }
#ifdef CHAIN_STAMPS
        if (S.stamps) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + 24 + k], now - S.t_last); S.t_last = now; }
#endif
    }
#ifdef CHAIN_STAMPS
    if (S.stamps) { const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[35], now - S.t_last); S.t_last = now; }
#endif
    return mom;
}

template <bool EARLY>
__device__ __forceinline__ void integrate_chain_body(chain_prog prog, int n_units, const int4* __restrict__ unit_atoms,
                            const unsigned char* __restrict__ unit_type, const float* __restrict__ shake_dist,
                            settle_const sc, float tol,
                            int Npad, float4* __restrict__ pos, float4* __restrict__ vel,
                            long long* force, const float* __restrict__ invmass,
                            const int64_t* __restrict__ labels, const double* __restrict__ beta,
                            int r_begin, uint64_t seed, long long* __restrict__ cmm, float inv_total_mass,
                            unsigned int* join_flag, unsigned int join_seq, remd_chain_bins bins,
                            unsigned long long* chain_slots, unsigned int* chain_sync_err,
                            unsigned long long* own_time, long long* __restrict__ work, float4* __restrict__ xold, float4* __restrict__ vold,
                            remd_fold_args fold, const unsigned int* __restrict__ noise_id)
{
    // EARLY: what does not depend on the forces travels while this workgroup waits for them (or, when they are complete already, all
    // at once instead of table -> type -> distances and label -> beta one round trip after the other): the unit table, positions,
    // velocities, masses, the state's temperature.  Positions and velocities are written by the previous chain launch of this stream
    // only.  (Not in the two-per-CU compilation: the longer live ranges are 34 more spilled registers there.)
    const int uidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    int4 a4 = make_int4(-1, -1, -1, -1);
    int type = UNIT_FREE;
    float dist[3] = { 0.f, 0.f, 0.f };
    float kT = 0.f;
    uint32_t rg = 0u;
    unit_regs S;
    S.have_cm = 0; S.shake_it = 0;
    if (EARLY) {
        if (uidx < n_units) {
            a4 = unit_atoms[uidx]; type = (int)unit_type[uidx];
            dist[0] = shake_dist[uidx * 3]; dist[1] = shake_dist[uidx * 3 + 1]; dist[2] = shake_dist[uidx * 3 + 2];
        }
        kT = frcp((float)beta[labels[r_begin + r]]);       // fp32 state: 1 ulp of kT is below its own rounding
        rg = noise_id ? noise_id[r] : (uint32_t)(r_begin + r);
        const int e_idx[4] = { a4.x, a4.y, a4.z, a4.w };
        const float4* Pe = pos + (size_t)r * Npad;
        const float4* Ve = vel + (size_t)r * Npad;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e_idx[k] >= 0) {
                const float4 p = Pe[e_idx[k]], w = Ve[e_idx[k]];
                S.x[k] = f3(p.x, p.y, p.z); S.v[k] = f3(w.x, w.y, w.z); S.im[k] = invmass[e_idx[k]];
            }
        }
    }
    if (fold.done) {
        // remd_fold_args: wait until every workgroup of the direct-space stream's last launch has counted itself done (their force
        // atomics are complete by then)
        // one counter per replica, each on its own cache line (done[16 r]): a replica's chain workgroups need that replica's forces
        // only, and a few hundred arrivals on ONE address serialise at the memory side (~35 ns each: profiles/r04_p_*)
        if (threadIdx.x == 0) {
            const unsigned int* word = fold.done + 16 * blockIdx.y;
            long long n = 0;
            while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fold.target) < 0) {
                // (a fault is already raised -- this propagation is run again whatever happens from here: do not wait a second time;
                // without this every step behind a poll that ran out waits for its own time-out, seconds each; looked at every 256th poll
                // only, from the 256th on: a wait of ordinary length never pays for the extra load)
                if ((n & 255) == 255 && __hip_atomic_load(chain_sync_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(2);
                if (++n > (1ll << 25)) { atomicCAS(chain_sync_err, 0u, 1u); break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (join_flag) {
        // the forces of the direct-space stream: poll its "done" flag here instead of behind a cross-stream event (remd_ctx::d_sync)
        if (threadIdx.x == 0) {
            long long n = 0;
            while ((int)(__hip_atomic_load(join_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - join_seq) < 0) {
                if ((n & 255) == 255 && __hip_atomic_load(join_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // (as above)
                __builtin_amdgcn_s_sleep(2);
                if (++n > (1ll << 25)) { atomicCAS(join_flag + 1, 0u, 1u); break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // profiling: this launch's OWN time, from "forces complete" seen to the end, for workgroup (0, 0) (the launch duration a
    // profiler reports also holds the wait for the direct-space stream in the prologue)
    const unsigned long long own_t0 = own_time ? wall_clock64() : 0ull;
    float3 mom = f3(0, 0, 0);
    const int cmm_r_eff = prog.cmm_r, cmm_w_eff = prog.cmm_w;
    if (uidx == 0 && cmm_r_eff >= 0) {
        // this chain consumes accumulator cmm_r: clear the OTHER buffer (its sum was consumed one step ago)
        long long* o = cmm + ((size_t)(1 - cmm_r_eff) * gridDim.y + r) * 4;
        o[0] = 0; o[1] = 0; o[2] = 0;
    }
    if (!EARLY) {
        if (uidx < n_units) a4 = unit_atoms[uidx];
        if (a4.x >= 0) type = (int)unit_type[uidx];
        if (a4.x >= 0 && type == UNIT_SHAKE) { dist[0] = shake_dist[uidx * 3]; dist[1] = shake_dist[uidx * 3 + 1]; dist[2] = shake_dist[uidx * 3 + 2]; }
        kT = frcp((float)beta[labels[r_begin + r]]);
        rg = noise_id ? noise_id[r] : (uint32_t)(r_begin + r);
    }
    const bool active = a4.x >= 0;                              // padding units do nothing (but take part in the 'M' barrier)
    if (!active) type = UNIT_FREE;
    const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
    float4* P = pos + (size_t)r * Npad;
    float4* V = vel + (size_t)r * Npad;
    const long long* F = force + (size_t)r * 3 * Npad;
    long long* Fw = force + (size_t)r * 3 * Npad;
    const long long* cr = cmm + ((size_t)max(cmm_r_eff, 0) * gridDim.y + r) * 4;
#ifdef CHAIN_STAMPS
    S.stamps = (own_time && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? own_time + 2 : nullptr;
    S.t_last = own_t0;
#endif
    // mesh-column bins: by the workgroup (below, behind the token program) when the columns fit its LDS counters, else atom by atom
    // at the end of run_unit
#ifdef CHAIN_UNIT_BINS
    const bool wg_bins = false;
#else
    const bool wg_bins = bins.count != nullptr && bins.nx <= CHAIN_BIN_COLS;
#endif
    remd_chain_bins unit_bins = bins;
    if (wg_bins) unit_bins.count = nullptr;
#ifdef CHAIN_STAMPS
    if (S.stamps) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + 20], now - S.t_last); S.t_last = now; }
#endif
    // segments of the token program, split at 'M' (momentum sum + barrier over the replica's workgroups)
    for (int t0 = 0;;) {
        int t1 = t0;
        while (t1 < prog.n && chain_tok(prog, t1) != 'M') ++t1;
        const bool first = t0 == 0, last = t1 == prog.n;
        if (active) {
#define RUN(TY, NA) mom = run_unit<TY, NA, !EARLY>(prog, idx, dist, sc, tol, Npad, P, V, F, Fw, invmass, kT, rg, seed, cr, inv_total_mass, unit_bins, r, S, t0, t1, first, last, xold ? xold + (size_t)r * Npad : nullptr, vold ? vold + (size_t)r * Npad : nullptr)
            if (type == UNIT_SETTLE) RUN(UNIT_SETTLE, 3);
            else if (type == UNIT_FREE) { if (a4.y < 0) RUN(UNIT_FREE, 1); else RUN(UNIT_FREE, 4); }
            else if (a4.z < 0) RUN(UNIT_SHAKE, 2);
            else if (a4.w < 0) RUN(UNIT_SHAKE, 3);
            else RUN(UNIT_SHAKE, 4);
#undef RUN
        }
        if (last) break;
        {
            // 'M': sum(m v) of the velocities as they are now, over all workgroups of this replica (all of them are resident: the launcher
            // checks the grid size).  One exchange through device memory: a workgroup publishes its three fixed-point partial sums as
            // 64-bit words that carry the epoch of this barrier in their low 16 bits (48-bit payload), in the epoch's parity half of
            // chain_slots [2][replica][workgroup][3]; every workgroup then reads all words of its replica, spinning on a word until its
            // tag is this epoch's.  A word is rewritten two epochs later, which its writer cannot reach before every reader has published
            // the epoch in between, i.e. is done with this one.  (Before: atomics into one accumulator + an arrival counter + a read-back
            // = three dependent round trips, 5.1 us of the headline step's 27 us chain; the sum is the same integer.)
            __shared__ long long s_pm[4][3];
            __shared__ unsigned long long s_tot[3];
            float3 pm = f3(0, 0, 0);
            if (active) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (idx[k] >= 0) pm = pm + S.v[k] * frcp(S.im[k]);
            }
            for (int off = 32; off > 0; off >>= 1) {
                pm.x += __shfl_xor(pm.x, off); pm.y += __shfl_xor(pm.y, off); pm.z += __shfl_xor(pm.z, off);
            }
            if ((threadIdx.x & 63) == 0) {
                long long* w = s_pm[threadIdx.x >> 6];
                w[0] = (long long)((double)pm.x * 4294967296.0); w[1] = (long long)((double)pm.y * 4294967296.0); w[2] = (long long)((double)pm.z * 4294967296.0);
            }
            if (threadIdx.x < 3) s_tot[threadIdx.x] = 0ull;
            __syncthreads();
            const unsigned long long tag = (unsigned long long)(prog.m_epoch & 0xffffu);
            unsigned long long* slots = chain_slots + ((size_t)(prog.m_epoch & 1u) * gridDim.y + r) * gridDim.x * 3;
            if (threadIdx.x < 3) {
                const long long t = s_pm[0][threadIdx.x] + s_pm[1][threadIdx.x] + s_pm[2][threadIdx.x] + s_pm[3][threadIdx.x];
                if (t >= (1ll << 46) || t < -(1ll << 46)) atomicCAS(chain_sync_err, 0u, 7u);       // (a fault of its own: the host then sums with two launches, the polled waits stay)
                __hip_atomic_store(&slots[blockIdx.x * 3 + threadIdx.x], ((unsigned long long)t << 16) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int n_words = 3 * (int)gridDim.x;
            if (threadIdx.x < 255 && (int)threadIdx.x < n_words) {
                long long part = 0;
                for (int q = threadIdx.x; q < n_words; q += 255) {          // (255 = 0 mod 3: a thread stays on one component)
                    unsigned long long w;
                    long long n = 0;
                    while (((w = __hip_atomic_load(&slots[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffull) != tag) {
                        if ((n & 255) == 255 && __hip_atomic_load(chain_sync_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // (as in the prologue)
                        __builtin_amdgcn_s_sleep(1);
                        if (++n > (1ll << 25)) { atomicCAS(chain_sync_err, 0u, 3u); break; }
                    }
                    part += (long long)w >> 16;
                }
                atomicAdd(&s_tot[threadIdx.x % 3], (unsigned long long)part);
            }
            __syncthreads();
            S.cmx = (float)(long long)s_tot[0] * (1.0f / 4294967296.0f) * inv_total_mass;
            S.cmy = (float)(long long)s_tot[1] * (1.0f / 4294967296.0f) * inv_total_mass;
            S.cmz = (float)(long long)s_tot[2] * (1.0f / 4294967296.0f) * inv_total_mass;
            S.have_cm = 1;
            __syncthreads();                                    // (s_pm / s_tot may be written again by a second 'M' of this launch)
#ifdef CHAIN_STAMPS
            if (S.stamps) { const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + min(t1, 33)], now - S.t_last); S.t_last = now; }
#endif
        }
        t0 = t1 + 1;
    }
    if (wg_bins) {
        // The final positions of this workgroup's atoms, binned by PME mesh column for the spreading pass that follows (the order inside
        // a bin is irrelevant: charges and forces are fixed-point sums).  One atomic with a returned slot per ATOM made the end of the chain
        // three dependent rounds per unit (each behind the stores issued before it: one in-order counter) on ~35 atoms per counter;
        // here the atoms take ranks in LDS counters, the workgroup reserves one range per occupied column with ONE returning atomic,
        // and every atom stores at base + rank.
        __shared__ int s_bin_cnt[CHAIN_BIN_COLS], s_bin_base[CHAIN_BIN_COLS];
        for (int c = threadIdx.x; c < bins.nx; c += blockDim.x) s_bin_cnt[c] = 0;
        __syncthreads();
        int col[4] = { 0, 0, 0, 0 }, rank[4] = { 0, 0, 0, 0 };
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (idx[k] >= 0) {
                    float u; int kx;
                    remd_pme_scaled1(S.x[k].x, bins.box[4 * r], bins.nx, u, kx);
                    if (kx >= bins.nx) kx -= bins.nx;
                    col[k] = kx; rank[k] = atomicAdd(&s_bin_cnt[kx], 1);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < bins.nx; c += blockDim.x) {
            const int n = s_bin_cnt[c];
            if (n > 0) s_bin_base[c] = atomicAdd(&bins.count[(size_t)r * bins.nx + c], n);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (idx[k] >= 0) {
                    const int slot = s_bin_base[col[k]] + rank[k];
                    if (slot < bins.cap) {
                        const size_t e = ((size_t)r * bins.nx + col[k]) * bins.cap + slot;
                        bins.atoms[e] = make_float4(S.x[k].x, S.x[k].y, S.x[k].z, __int_as_float(idx[k]));
                        if (bins.q) bins.q[e] = bins.param[idx[k]].x;        // (state-independent charges only, see remd_pme_chain_bins)
                    } else atomicCAS(bins.err, 0u, 2u);
                }
            }
        }
#ifdef CHAIN_STAMPS
        if (S.stamps) { const unsigned long long now = wall_clock64(); atomicAdd(&S.stamps[1 + 27], now - S.t_last); S.t_last = now; }
#endif
    }
    if (prog.accumulate_momentum) {
        // wavefront shuffle reduction, one fixed-point atomic per wave (integer => order-independent sum)
        for (int off = 32; off > 0; off >>= 1) {
            mom.x += __shfl_xor(mom.x, off); mom.y += __shfl_xor(mom.y, off); mom.z += __shfl_xor(mom.z, off);
        }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* c = reinterpret_cast<unsigned long long*>(cmm + ((size_t)cmm_w_eff * gridDim.y + r) * 4);
            atomicAdd(&c[0], (unsigned long long)(long long)((double)mom.x * 4294967296.0));
            atomicAdd(&c[1], (unsigned long long)(long long)((double)mom.y * 4294967296.0));
            atomicAdd(&c[2], (unsigned long long)(long long)((double)mom.z * 4294967296.0));
        }
    }
    if (prog.measure && work) {
        // heat / kinetic shadow work of this launch: wavefront sums, one fixed-point atomic per wave (order-independent)
        float hq = active ? S.heat : 0.f, sw = active ? S.shadow : 0.f;
        for (int off = 32; off > 0; off >>= 1) { hq += __shfl_xor(hq, off); sw += __shfl_xor(sw, off); }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* w = reinterpret_cast<unsigned long long*>(work + 4 * (size_t)r);
            if (prog.measure & 1) atomicAdd(&w[0], (unsigned long long)(long long)((double)hq * 16777216.0));
            if (prog.measure & 2) atomicAdd(&w[1], (unsigned long long)(long long)((double)sw * 16777216.0));
        }
    }
    // most Newton updates any X-H position solve of this handle has needed (remd_get_constraint_stats): d_sync[3], next to the fault
    // word.  The word only ever grows and stops changing after the first steps, so a lane looks before it writes (no atomic otherwise).
    if (type == UNIT_SHAKE && S.shake_it > 0 &&
        (unsigned int)S.shake_it > __hip_atomic_load(chain_sync_err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(chain_sync_err + 1, (unsigned int)S.shake_it);
    if (own_time && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        atomicAdd(&own_time[0], wall_clock64() - own_t0);
        atomicAdd(&own_time[1], 1ull);
    }
}

// Two compilations of the same body.  The integrator's working set under the widest unit type is 256 VGPRs + AGPRs = one wavefront per
// SIMD, one workgroup per CU; a grid larger than the chip then runs in rounds (DHFR x 16: 512 workgroups).  The second compilation is held
// to two wavefronts per SIMD (256 registers in all: the four-atom X-H path spills, the water path -- 238 -- does not) and takes such grids
// in one round.
#define CHAIN_PARAMS chain_prog prog, int n_units, const int4* __restrict__ unit_atoms, const unsigned char* __restrict__ unit_type, const float* __restrict__ shake_dist, settle_const sc, float tol, int Npad, float4* __restrict__ pos, float4* __restrict__ vel, long long* force, const float* __restrict__ invmass, const int64_t* __restrict__ labels, const double* __restrict__ beta, int r_begin, uint64_t seed, long long* __restrict__ cmm, float inv_total_mass, unsigned int* join_flag, unsigned int join_seq, remd_chain_bins bins, unsigned long long* chain_slots, unsigned int* chain_sync_err, unsigned long long* own_time, long long* __restrict__ work, float4* __restrict__ xold, float4* __restrict__ vold, remd_fold_args fold, const unsigned int* __restrict__ noise_id
#define CHAIN_ARGS prog, n_units, unit_atoms, unit_type, shake_dist, sc, tol, Npad, pos, vel, force, invmass, labels, beta, r_begin, seed, cmm, inv_total_mass, join_flag, join_seq, bins, chain_slots, chain_sync_err, own_time, work, xold, vold, fold, noise_id
__global__ __launch_bounds__(256)
void integrate_chain_kernel(CHAIN_PARAMS) { integrate_chain_body<true>(CHAIN_ARGS); }
__global__ __launch_bounds__(256, 2)
void integrate_chain2_kernel(CHAIN_PARAMS) { integrate_chain_body<false>(CHAIN_ARGS); }

// Maxwell-Boltzmann velocities (mcmc.py:710-711): v = sqrt(kT/m) xi, then velocity constraints.
template <int TYPE, int NAT>
__device__ __forceinline__ void assign_unit(const int* idx, const settle_const& sc, float tol, const float4* __restrict__ P,
                                            float4* __restrict__ V, const float* __restrict__ invmass, float kT, uint32_t rg,
                                            uint64_t seed, int64_t iteration)
{
    float3 x[NAT], v[NAT];
    float im[NAT];
#pragma unroll
    for (int k = 0; k < NAT; ++k) {
        const float4 p = P[idx[k]];
        x[k] = f3(p.x, p.y, p.z);
        im[k] = invmass[idx[k]];
        const float3 xi = gaussian3(seed, REMD_STREAM_VELOCITY, (uint32_t)idx[k], rg, (uint64_t)iteration);
        const float sig = fsqrt(kT * im[k]);
        v[k] = xi * sig;
    }
    constrain_v<TYPE, NAT>(sc, im, tol, v, x);
#pragma unroll
    for (int k = 0; k < NAT; ++k) V[idx[k]] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
}

__global__ __launch_bounds__(256)
void assign_velocities_kernel(int n_units, const int4* __restrict__ unit_atoms,
                              const unsigned char* __restrict__ unit_type, settle_const sc, float tol,
                              int Npad, const float4* __restrict__ pos, float4* __restrict__ vel,
                              const float* __restrict__ invmass, const int64_t* __restrict__ labels,
                              const double* __restrict__ beta, int r_begin, uint64_t seed, int64_t iteration,
                              const unsigned int* __restrict__ noise_id)
{
    const int uidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (uidx >= n_units) return;
    const int4 a4 = unit_atoms[uidx];
    if (a4.x < 0) return;
    const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
    const int type = unit_type[uidx];
    const float4* P = pos + (size_t)r * Npad;
    float4* V = vel + (size_t)r * Npad;
    const float kT = frcp((float)beta[labels[r_begin + r]]);       // fp32 state: 1 ulp of kT is below its own rounding
    const uint32_t rg = noise_id ? noise_id[r] : (uint32_t)(r_begin + r);
    if (type == UNIT_SETTLE) assign_unit<UNIT_SETTLE, 3>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration);
    else if (type == UNIT_FREE) { if (a4.y < 0) assign_unit<UNIT_FREE, 1>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration);
                                  else assign_unit<UNIT_FREE, 4>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration); }
    else if (a4.z < 0) assign_unit<UNIT_SHAKE, 2>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration);
    else if (a4.w < 0) assign_unit<UNIT_SHAKE, 3>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration);
    else assign_unit<UNIT_SHAKE, 4>(idx, sc, tol, P, V, invmass, kT, rg, seed, iteration);
}

// KE = sum 1/2 m v^2 per replica: per-lane partial -> wave shuffle -> LDS -> one value per block,
// blocks of one replica are summed in fixed order by the last stage (deterministic).
__global__ __launch_bounds__(256)
void kinetic_energy_kernel(int N, int Npad, const float4* __restrict__ vel, const float* __restrict__ mass,
                           double* __restrict__ ke)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* V = vel + (size_t)r * Npad;
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float4 v = V[i];
        acc += 0.5 * (double)mass[i] * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ke[r] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void check_finite_kernel(int N, int Npad, const float4* __restrict__ pos, const float4* __restrict__ vel, int* __restrict__ nan_flag)
{
    const int r = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 p = pos[(size_t)r * Npad + i], v = vel[(size_t)r * Npad + i];
    const float s = p.x + p.y + p.z + v.x + v.y + v.z;
    if (!isfinite(s)) nan_flag[r] = 1;
}

// ---------------------------------------------------------------------------------------------
struct unit_tables {
    int n_units = 0;
    int4* d_atoms = nullptr; unsigned char* d_type = nullptr; float* d_dist = nullptr;
    settle_const sc{};
};
static handle_table<unit_tables> g_units;

int remd_build_constraints(remd_ctx* h, const remd_system_desc* d)
{
    unit_tables& ut = g_units[h];
    if (ut.d_atoms) { hipFree(ut.d_atoms); hipFree(ut.d_type); hipFree(ut.d_dist); ut = unit_tables(); }
    const int N = d->n_atoms;
    std::vector<char> used(N, 0);
    std::vector<int4> atoms; std::vector<unsigned char> type; std::vector<float> dist;
    for (int s = 0; s < d->n_settle; ++s) {
        const int* a = d->settle_atoms + 3 * s;
        atoms.push_back(make_int4(a[0], a[1], a[2], -1)); type.push_back(UNIT_SETTLE);
        dist.push_back(0); dist.push_back(0); dist.push_back(0);
        for (int k = 0; k < 3; ++k) { if (a[k] < 0 || a[k] >= N || used[a[k]]) return remd_fail(h, -3, "bad SETTLE triple"); used[a[k]] = 1; }
    }
    while (atoms.size() % 64) { atoms.push_back(make_int4(-1, -1, -1, -1)); type.push_back(UNIT_SETTLE); dist.push_back(0); dist.push_back(0); dist.push_back(0); }
    for (int s = 0; s < d->n_shake; ++s) {
        const int* a = d->shake_atoms + 4 * s;
        atoms.push_back(make_int4(a[0], a[1], a[2], a[3])); type.push_back(UNIT_SHAKE);
        for (int k = 0; k < 3; ++k) dist.push_back((float)d->shake_dist[3 * s + k]);
        for (int k = 0; k < 4; ++k) if (a[k] >= 0) { if (a[k] >= N || used[a[k]]) return remd_fail(h, -3, "bad SHAKE cluster"); used[a[k]] = 1; }
    }
    auto pad64 = [&](unsigned char ty) {
        // units of one kind fill whole wavefronts: no wave mixes SETTLE, SHAKE and free-atom code paths
        while (atoms.size() % 64) { atoms.push_back(make_int4(-1, -1, -1, -1)); type.push_back(ty); dist.push_back(0); dist.push_back(0); dist.push_back(0); }
    };
    pad64(UNIT_SHAKE);
    // unconstrained atoms: FOUR to a unit while they last, the remainder one each (a unit = a thread; on DHFR -- 7023 waters, 790 X-H
    // clusters, 464 free atoms -- single-atom units made 8336 units = 33 workgroups per replica, and at one workgroup per CU (this
    // kernel's registers) 16 x 33 = 528 workgroups are THREE rounds of the chip; four to a unit: 7988 units = 32 workgroups, 512 = two rounds)
    {
        std::vector<int> fr;
        for (int i = 0; i < N; ++i) if (!used[i]) fr.push_back(i);
        size_t q = 0;
        for (; q + 4 <= fr.size(); q += 4) {
            atoms.push_back(make_int4(fr[q], fr[q + 1], fr[q + 2], fr[q + 3])); type.push_back(UNIT_FREE);
            dist.push_back(0); dist.push_back(0); dist.push_back(0);
        }
        for (; q < fr.size(); ++q) {
            atoms.push_back(make_int4(fr[q], -1, -1, -1)); type.push_back(UNIT_FREE);
            dist.push_back(0); dist.push_back(0); dist.push_back(0);
        }
    }
    ut.n_units = (int)atoms.size();
    REMD_CHECK(h, hipMalloc(&ut.d_atoms, sizeof(int4) * ut.n_units));
    REMD_CHECK(h, hipMalloc(&ut.d_type, ut.n_units));
    REMD_CHECK(h, hipMalloc(&ut.d_dist, sizeof(float) * 3 * ut.n_units));
    REMD_CHECK(h, hipMemcpy(ut.d_atoms, atoms.data(), sizeof(int4) * ut.n_units, hipMemcpyHostToDevice));
    REMD_CHECK(h, hipMemcpy(ut.d_type, type.data(), ut.n_units, hipMemcpyHostToDevice));
    REMD_CHECK(h, hipMemcpy(ut.d_dist, dist.data(), sizeof(float) * 3 * ut.n_units, hipMemcpyHostToDevice));
    h->n_settle = d->n_settle; h->n_shake = d->n_shake;
    int n_con = 3 * d->n_settle;
    for (int s = 0; s < d->n_shake; ++s) for (int k = 1; k < 4; ++k) if (d->shake_atoms[4 * s + k] >= 0) n_con++;
    h->n_dof = 3 * N - n_con - (d->cmm_frequency > 0 ? 3 : 0);
    if (d->n_settle > 0) {
        const int* a = d->settle_atoms;
        const double mO = d->mass[a[0]], mH = d->mass[a[1]];
        const double rc = 0.5 * d->settle_dHH;
        const double t = sqrt(d->settle_dOH * d->settle_dOH - rc * rc);
        const double ra = 2.0 * mH * t / (mO + 2.0 * mH);
        ut.sc.mO = (float)mO; ut.sc.mH = (float)mH; ut.sc.ra = (float)ra; ut.sc.rb = (float)(t - ra);
        ut.sc.rc = (float)rc; ut.sc.dOH = (float)d->settle_dOH; ut.sc.dHH = (float)d->settle_dHH;
    }
    return 0;
}

void remd_free_constraints(remd_ctx* h)
{
    unit_tables* it = g_units.find(h);
    if (!it) return;
    if (it->d_atoms) { hipFree(it->d_atoms); hipFree(it->d_type); hipFree(it->d_dist); }
    g_units.erase(h);
}

int remd_parse_splitting(remd_ctx* h, const char* splitting, std::vector<char>& tokens, int& nV, int& nR, int& nO, int* nVg)
{
    // integrators.py:1474-1537: space-separated, case-insensitive V/R/O tokens, Metropolization braces, and force-group
    // suffixes V0, V1, ...: with more than one distinct group the splitting is a multiple-time-step one, every V must name its
    // group (:1527-1529) and takes dt / (occurrences of that group) with that group's forces only (:1437-1438); with one group
    // (or none) every V uses all forces and dt / (number of V tokens) (:1440, :1535)
    tokens.clear(); nV = nR = nO = 0;
    std::string s(splitting ? splitting : "");
    std::vector<int> vgroup;             // per V token: group index or -1 (no suffix)
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && s[i] == ' ') ++i;
        if (i >= s.size()) break;
        size_t j = i; while (j < s.size() && s[j] != ' ') ++j;
        std::string tok = s.substr(i, j - i);
        for (auto& c : tok) c = (char)toupper(c);
        if (tok[0] == 'V' && tok.find_first_not_of("0123456789", 1) == std::string::npos) {
            int g = -1;
            if (tok.size() > 1) {
                if (tok.size() > 3) return remd_fail(h, -3, "force group of '" + tok + "' out of range");
                g = atoi(tok.c_str() + 1);
                if (g > 31) return remd_fail(h, -3, "OpenMM only allows up to 32 force groups (integrators.py:1346-1347)");
            }
            tokens.push_back('V'); vgroup.push_back(g); nV++;
        }
        else if (tok == "R") { tokens.push_back('R'); nR++; }
        else if (tok == "O") { tokens.push_back('O'); nO++; }
        else if (tok == "{" || tok == "}") tokens.push_back(tok[0]);      // Metropolization of the substeps in between (:1539-1557)
        else return remd_fail(h, -3, "unsupported splitting token '" + tok + "' (supported: V V<group> R O { })");
        i = j;
    }
    {
        std::set<int> groups;
        for (int g : vgroup) if (g >= 0) groups.insert(g);
        int counts[4] = {0, 0, 0, 0};
        if (groups.size() > 1) {
            if (!nVg) return remd_fail(h, -3, "multiple-time-step splittings are set with remd_set_integrator");
            size_t v = 0;
            for (auto& c : tokens) if (c == 'V') {
                const int g = vgroup[v++];
                if (g < 0) return remd_fail(h, -3, "a multiple-time-step splitting must name the force group of every V (integrators.py:1527-1529)");
                if (g > 3) return remd_fail(h, -3, "force groups above 3 are not supported in multiple-time-step splittings");
                c = (char)('0' + g); counts[g]++;
            }
        }
        if (nVg) for (int g = 0; g < 4; ++g) nVg[g] = counts[g];
    }
    if (tokens.empty()) return remd_fail(h, -3, "empty splitting string");
    if (nR == 0 || nV == 0) return remd_fail(h, -3, "splitting needs at least one R and one V (integrators.py:1376-1385)");
    int depth = 0;
    for (char c : tokens) {
        if (c == '{') { if (++depth > 1) return remd_fail(h, -3, "nested '{' in the splitting string"); }
        else if (c == '}') { if (--depth < 0) return remd_fail(h, -3, "'}' without '{' in the splitting string"); }
        else if (c == 'O' && depth > 0) return remd_fail(h, -3, "O substeps cannot be Metropolized (integrators.py:1387-1401)");
    }
    if (depth != 0) return remd_fail(h, -3, "'{' without '}' in the splitting string");
    return 0;
}

remd_chain_bins remd_pme_chain_bins(remd_ctx* h);
static void launch_chain(remd_ctx* h, const unit_tables& ut, const chain_prog& prog, bool bin_for_pme = false)
{
    // mesh-column bins from the chain's epilogue: in a handle that runs as ONE block (the binning launch would sit on the step's only
    // critical path); the blocks of a phased propagation bin with a launch of their own, which runs beside the other block's kernels while
    // the chain is the serial part of both (24 x alanine dipeptide: 17.5 -> 18.2 it/s; one block of 8 x CB7:B2: 13.2 -> 12.2 the other way;
    // profiles/r06_45).  REMD_PME_CHAINBIN=0 / 1 pins it (bit-identical either way: the order inside a bin is irrelevant).
    static const int chainbin_env = getenv("REMD_PME_CHAINBIN") ? atoi(getenv("REMD_PME_CHAINBIN")) : -1;
    // (a block whose chain grid is several rounds of the chip -- 64 x DHFR: 2 048 workgroups -- keeps them too: there the epilogue is
    // amortised over the rounds and the binning launch is the dearer one, 3.38 -> 3.46 s per iteration of 128 x DHFR without this bound)
    const bool chain_bins = chainbin_env >= 0 ? chainbin_env != 0
                                              : (h->parent == nullptr || (long long)((ut.n_units + 255) / 256) * h->R > 1024);
    const remd_chain_bins bins = (bin_for_pme && chain_bins) ? remd_pme_chain_bins(h) : remd_chain_bins();
    remd_prof_scope ps(h, "integrate_chain");
    dim3 grid((ut.n_units + 255) / 256, h->R);
    // The two-per-CU compilation is OPT-IN (REMD_CHAIN_TWO=1).  It takes a grid larger than the chip in one round (DHFR x 16: -1.3 % of the
    // converged step), but its workgroups hold the WHOLE register file of their CUs while they poll for the forces in the prologue, and a
    // direct-space stream that still has workgroups to place (one workgroup per work item, the scatter) then never gets a slot: the poll runs
    // out after seconds (seen at the end of round 5 with 128 alanine replicas = 384 workgroups, and as a 4.8 s stall of the tuner's
    // one-per-item candidate on DHFR).  One workgroup per CU leaves 160 registers per lane for the kernels the chain waits for.
    static const char* two_env = getenv("REMD_CHAIN_TWO");
    const bool two = two_env != nullptr && atoi(two_env) != 0;
    auto kern = two ? integrate_chain2_kernel : integrate_chain_kernel;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, h->stream, prog, ut.n_units, ut.d_atoms, ut.d_type,
                       ut.d_dist, ut.sc, (float)fmax(h->constraint_tol, REMD_CONSTRAINT_TOL_FLOOR), h->Npad, h->d_pos, h->d_vel, h->d_force,
                       h->d_invmass, h->d_labels, h->d_beta, h->r_begin, h->seed, h->d_cmm,
                       (float)(h->total_mass > 0 ? 1.0 / h->total_mass : 0.0),
                       h->join_deferred ? h->d_sync + 1 : (unsigned int*)nullptr, h->join_deferred, bins, h->d_chain_sync, h->d_sync + 2,
                       (h->profiling == 2 || (h->profiling == 1 && h->prof_filter.find("integrate_chain") != std::string::npos)) ? h->d_chain_own : (unsigned long long*)nullptr,
                       h->d_work, h->d_xold, h->d_vold, h->fold_pending ? h->fold : remd_fold_args(), h->d_noise_id);
    h->join_deferred = 0; h->fold_pending = false;
    if (bins.count) h->cbins_ready = true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Heat, shadow work and Metropolization (integrators.py:1175-1204, 1404-1460, 1539-1557).  The kinetic-energy changes of the
// substeps are summed inside the chain kernel (fixed point, per replica); the potential-energy change of an R substep needs the
// energy at the positions before and after it: remd_run_steps evaluates energies there (the evaluation behind an R also
// provides the forces of the V that follows, so a step of "V R O R V" costs two evaluations with energies instead of one without).
#define WORK_SCALE 16777216.0        // 2^24: 6e-8 kJ/mol
__global__ void work_pe_kernel(int R, const double* __restrict__ U, double* __restrict__ pe_prev, long long* __restrict__ work, int accumulate)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (accumulate) work[4 * r + 1] += (long long)((U[r] - pe_prev[r]) * WORK_SCALE);          // :1420-1423 (potential part)
    pe_prev[r] = U[r];
}
// '}' (:1544-1557): accept = step(exp(-shadow_work / kT) - uniform); trials++; on rejection x = xold, v = -vold; shadow_work = 0
__global__ void metropolis_kernel(int R, int r_begin, uint64_t seed, long long gstep, int brace, const int64_t* __restrict__ labels,
                                  const double* __restrict__ beta, long long* __restrict__ work, int* __restrict__ accept,
                                  const unsigned int* __restrict__ noise_id)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double sw = (double)work[4 * r + 1] / WORK_SCALE;
    const philox4 w = remd_philox(seed, REMD_STREAM_METROPOLIS, (uint32_t)brace, noise_id ? noise_id[r] : (uint32_t)(r_begin + r), (uint64_t)gstep);
    const double u = remd_u53(w.w[2], w.w[3]);
    const int acc = (exp(-sw * beta[labels[r_begin + r]]) - u >= 0.0) ? 1 : 0;
    accept[r] = acc;
    work[4 * r + 2] += 1;
    if (!acc) work[4 * r + 3] += 1;
    work[4 * r + 1] = 0;
}
__global__ __launch_bounds__(256)
void metropolis_restore_kernel(int N, int Npad, const int* __restrict__ accept, float4* __restrict__ pos, float4* __restrict__ vel,
                               const float4* __restrict__ xold, const float4* __restrict__ vold)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (i >= N || accept[r]) return;
    const size_t o = (size_t)r * Npad + i;
    pos[o] = xold[o];
    const float4 v = vold[o];
    vel[o] = make_float4(-v.x, -v.y, -v.z, 0.f);
}
int remd_work_buffers(remd_ctx* h)
{
    if (h->work_R == h->R && h->d_work) return 0;
    if (h->d_work) { hipFree(h->d_work); hipFree(h->d_pe_prev); hipFree(h->d_xold); hipFree(h->d_vold); hipFree(h->d_accept); }
    h->d_work = nullptr;
    REMD_CHECK(h, hipMalloc(&h->d_work, sizeof(long long) * 4 * h->R));
    REMD_CHECK(h, hipMalloc(&h->d_pe_prev, sizeof(double) * h->R));
    REMD_CHECK(h, hipMalloc(&h->d_xold, sizeof(float4) * (size_t)h->R * h->Npad));
    REMD_CHECK(h, hipMalloc(&h->d_vold, sizeof(float4) * (size_t)h->R * h->Npad));
    REMD_CHECK(h, hipMalloc(&h->d_accept, sizeof(int) * h->R));
    REMD_CHECK(h, hipMemsetAsync(h->d_work, 0, sizeof(long long) * 4 * h->R, h->stream));
    h->work_R = h->R;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Resident small-system path (round 3).  For systems of up to 1024 atoms without constraints, mesh or listed terms (the
// reference's HarmonicOscillator and LennardJonesFluid test systems: BASELINE configs 1 and 2) an MD step of the regular path
// is a chain of ~8 dependent launches of a few microseconds each and the GPU idles in between (LJ fluid, 16 replicas: 30 us
// per step for 8 k atoms).  mcmc.py:700-719 is ONE integrator.step(n_steps) per move, so the MI355X-first shape of that is
// ONE launch per propagation: a workgroup owns a replica, a thread owns an atom (x, v, f, 1/m and the pair parameters stay in
// registers for all n_steps), positions and a Verlet list live in LDS:
//   * neighbour list: all pairs inside r_c + skin, FULL list (every pair from both sides: no atomics, every atom sums its
//     forces in list order -- deterministic), rebuilt by an all-pairs pass over the LDS positions whenever any atom has moved
//     more than skin / 2 since the last build (a workgroup vote at every evaluation): the list is a superset of the pairs
//     inside r_c at all times, the cutoff test in the force loop is exact;
//   * force evaluation where the splitting string needs one (a V after an R), with the pair arithmetic of the regular
//     kernels (pair_math.h: LJ + switch, soft-core for alchemical / non-alchemical pairs at the replica's lambda);
//   * V / R / O / centre-of-mass removal as in the chain kernel, same Philox streams (atom, global replica, global O-substep
//     counter): the trajectories follow the regular path to fp32 summation order.
// Two workgroup barriers per force evaluation, nothing else between steps.
struct resident_prog {
    int n; char tok[MAX_TOK]; int o_index[MAX_TOK];
    float hV, hR, a, b; int nO;
    int n_steps, cmm_frequency; long long gstep0, first_step;
};
struct resident_sys {
    int N, Npad, method, alch, n_ext, list_cap;
    nb_params p;
    float skin, ext_K, ext_x0, inv_total_mass;
    const float4* param; const float* rep_lam; const int* ext_atoms; const float* invmass; const float* box;
    const int64_t* labels; const double* beta; int r_begin; uint64_t seed; const unsigned int* noise_id;
    unsigned int* err;
};

#ifndef RES_UNROLL
#define RES_UNROLL 2
#endif
#define RES_FSCALE 8192.f       // LDS force accumulators: 32-bit fixed point, 2^-13 kJ/mol/nm (|F| < 2.6e5 kJ/mol/nm: r > 0.19 nm for argon)

// LJ + switch of one pair without branches: the soft-core form  x = 1 / (sc + (r / sigma)^6),  U = lam eps4 x (x - 1)  IS
// Lennard-Jones for sc = 0, lam = 1 (alchemy.py:1383-1388 with softcore_c = 6), so alchemical / non-alchemical pairs differ from
// the rest only in two selected constants; the switching polynomial is evaluated at x clamped to [0, 1].  Hardware reciprocals
// (1 ulp).  Returns dU/dr / r, so that F_i = fr * (x_j - x_i).
template <bool ALCH>
__device__ __forceinline__ float resident_pair(const nb_params& p, float r2, float4 pi, float4 pj, float lam_a, float sc)
{
    const float inv_r = __builtin_amdgcn_rsqf(r2), r = r2 * inv_r;
    const float sig = pi.y + pj.y, eps4 = pi.z * pj.z;
    const bool soft = ALCH && ((pi.w != pj.w) || pi.w > 1.5f);      // (w = 2: annihilate_sterics, pair_math.h)
    const float lam = soft ? lam_a : 1.f, s0 = soft ? sc : 0.f;
    const float is2 = __builtin_amdgcn_rcpf(sig * sig);
    const float q2 = r2 * is2, t = q2 * q2 * q2;
    const float x = __builtin_amdgcn_rcpf(s0 + t);
    float U = lam * eps4 * x * (x - 1.f);
    float dUdr = lam * eps4 * (2.f * x - 1.f) * (-x * x * 6.f * t * inv_r);
    const float xs = fminf(fmaxf((r - p.rs) * p.inv_sw, 0.f), 1.f);          // no switch: inv_sw = 0
    const float Sw = 1.f + xs * xs * xs * (-10.f + xs * (15.f - 6.f * xs));
    const float dS = xs * xs * (-30.f + xs * (60.f - 30.f * xs)) * p.inv_sw;
    dUdr = Sw * dUdr + U * dS;
    return dUdr * inv_r;
}

template <bool ALCH>
__global__ __launch_bounds__(1024)
void resident_md_kernel(resident_prog prog, resident_sys S, float4* __restrict__ pos, float4* __restrict__ vel)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = S.N, T = blockDim.x, tid = threadIdx.x, r = blockIdx.x, nw = T >> 6;
    float4* s_pos = reinterpret_cast<float4*>(smem);                      // [T] positions (w: unused)
    float4* s_par = s_pos + T;                                            // [T] pair parameters of every atom
    float* s_red = reinterpret_cast<float*>(s_par + T);                   // [16][4] wavefront partial sums; the last words:
    int* s_vote = reinterpret_cast<int*>(s_red + 60);                     // [2] "somebody left its skin / 2 sphere", by evaluation parity
    int* s_np = reinterpret_cast<int*>(s_red + 62);                       // pairs in the list
    int* s_f = reinterpret_cast<int*>(s_red + 64);                        // [3][T] fixed-point force accumulators
    unsigned long long* s_fl = reinterpret_cast<unsigned long long*>(s_f + 3 * T + (T & 1));   // [3][T] the same in 64 bits: contributions too large for the fast path
    unsigned int* s_pairs = reinterpret_cast<unsigned int*>(s_fl + 3 * T); // [cap] i | j << 16, i < j
    const bool active = tid < N;
    float4* P = pos + (size_t)r * S.Npad;
    float4* V = vel + (size_t)r * S.Npad;
    float3 x = f3(0, 0, 0), v = f3(0, 0, 0), f = f3(0, 0, 0), xref = f3(0, 0, 0);
    float im = 0.f;
    float4 par = make_float4(0, 0, 0, 0);
    if (active) {
        const float4 p4 = P[tid], v4 = V[tid];
        x = f3(p4.x, p4.y, p4.z); v = f3(v4.x, v4.y, v4.z);
        im = S.invmass[tid];
        if (S.method >= 0) par = S.param[tid];
    }
    s_par[tid] = par;
    s_f[tid] = 0; s_f[T + tid] = 0; s_f[2 * T + tid] = 0;
    s_fl[tid] = 0ull; s_fl[T + tid] = 0ull; s_fl[2 * T + tid] = 0ull;
    bool ext = false;
    for (int k = 0; k < S.n_ext; ++k) ext |= (S.ext_atoms[k] == tid);
    const float Lx = S.box[4 * r], Ly = S.box[4 * r + 1], Lz = S.box[4 * r + 2];
    const float iLx = Lx > 0.f ? 1.f / Lx : 0.f, iLy = Ly > 0.f ? 1.f / Ly : 0.f, iLz = Lz > 0.f ? 1.f / Lz : 0.f;
    float lam_a = 1.f, sc = 0.f;
    if (ALCH) { lam_a = S.rep_lam[4 * r]; sc = S.rep_lam[4 * r + 1]; }
    const float kT = frcp((float)S.beta[S.labels[S.r_begin + r]]);
    const uint32_t rg = S.noise_id ? S.noise_id[r] : (uint32_t)(S.r_begin + r);
    const float rl = S.p.rc + S.skin, rl2 = rl * rl, half_skin2 = 0.25f * S.skin * S.skin;
    bool have_list = false, forces_valid = false;
    int n_eval = 0;
    if (tid == 0) { s_vote[0] = 0; s_vote[1] = 0; *s_np = 0; }
    __syncthreads();

    auto evaluate = [&]() {
        // publish the positions; rebuild the list if any atom has left its skin / 2 sphere.  The vote rides on the barrier that
        // publishes the positions (two words used in turn: the one of the next evaluation is cleared behind this barrier)
        const float3 d = x - xref;
        const bool moved = !have_list || (active && dot3(d, d) > half_skin2);
        const int par_e = n_eval & 1;
        s_pos[tid] = make_float4(x.x, x.y, x.z, 0.f);
        if (moved) s_vote[par_e] = 1;
        __syncthreads();
        const int rebuild = s_vote[par_e];
        if (tid == 0) s_vote[par_e ^ 1] = 0;
        ++n_eval;
        f = f3(0, 0, 0);
        if (S.method >= 0) {
            if (rebuild) {
                if (tid == 0) *s_np = 0;
                __syncthreads();
                if (active) {
                    for (int j = 0; j < N; ++j) {                        // wave-uniform j: LDS broadcast reads
                        const float4 q = s_pos[j];
                        float dx = q.x - x.x, dy = q.y - x.y, dz = q.z - x.z;
                        dx -= Lx * rintf(dx * iLx); dy -= Ly * rintf(dy * iLy); dz -= Lz * rintf(dz * iLz);
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 < rl2 && j > tid) {                       // every pair once; the order of the list does not matter
                            const int slot = atomicAdd(s_np, 1);         // (forces are integer sums)
                            if (slot < S.list_cap) s_pairs[slot] = (unsigned int)tid | ((unsigned int)j << 16);
                        }
                    }
                    xref = x;
                }
                have_list = true;
                __syncthreads();
                if (tid == 0 && *s_np > S.list_cap) atomicCAS(S.err, 0u, 4u);
            }
            {
                // a thread takes pairs tid, tid + T, ...: the same number for every lane (an atom-per-lane loop runs as long as the
                // busiest atom of the wavefront: 16 slots for 10 neighbours on average), RES_UNROLL pairs per trip with all their
                // LDS reads in flight
                const int np = min(*s_np, S.list_cap);
                for (int k = tid; k < np; k += RES_UNROLL * T) {
                    unsigned int w[RES_UNROLL]; float4 qi[RES_UNROLL], qj[RES_UNROLL], pi[RES_UNROLL], pj[RES_UNROLL];
#pragma unroll
                    for (int u = 0; u < RES_UNROLL; ++u) w[u] = s_pairs[min(k + u * T, np - 1)];
#pragma unroll
                    for (int u = 0; u < RES_UNROLL; ++u) {
                        const int i = w[u] & 0xffffu, j = w[u] >> 16;
                        qi[u] = s_pos[i]; qj[u] = s_pos[j]; pi[u] = s_par[i]; pj[u] = s_par[j];
                    }
#pragma unroll
                    for (int u = 0; u < RES_UNROLL; ++u) {
                        const int i = w[u] & 0xffffu, j = w[u] >> 16;
                        float dx = qj[u].x - qi[u].x, dy = qj[u].y - qi[u].y, dz = qj[u].z - qi[u].z;
                        dx -= Lx * rintf(dx * iLx); dy -= Ly * rintf(dy * iLy); dz -= Lz * rintf(dz * iLz);
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 < S.p.rc2 && k + u * T < np) {
                            const float fr = resident_pair<ALCH>(S.p, r2, pi[u], pj[u], lam_a, sc) * RES_FSCALE;
                            const float ax = fr * dx, ay = fr * dy, az = fr * dz;                                    // F_i = fr (x_j - x_i)
                            if (fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fabsf(az)) < 134217728.f) {                       // 2^27: sixteen of them fit 32 bits
                                const int fx = __float2int_rn(ax), fy = __float2int_rn(ay), fz = __float2int_rn(az);
                                atomicAdd(&s_f[i], fx); atomicAdd(&s_f[T + i], fy); atomicAdd(&s_f[2 * T + i], fz);
                                atomicAdd(&s_f[j], -fx); atomicAdd(&s_f[T + j], -fy); atomicAdd(&s_f[2 * T + j], -fz);
                            } else {
                                // a pair deep inside the repulsive core (an unminimised start): 64-bit accumulators, rare and slow
                                const long long fx = (long long)ax, fy = (long long)ay, fz = (long long)az;
                                atomicAdd(&s_fl[i], (unsigned long long)fx); atomicAdd(&s_fl[T + i], (unsigned long long)fy); atomicAdd(&s_fl[2 * T + i], (unsigned long long)fz);
                                atomicAdd(&s_fl[j], (unsigned long long)(-fx)); atomicAdd(&s_fl[T + j], (unsigned long long)(-fy)); atomicAdd(&s_fl[2 * T + j], (unsigned long long)(-fz));
                            }
                        }
                    }
                }
            }
            __syncthreads();                                 // every pair is in; nobody reads s_pos any more
            {
                const long long lx = (long long)s_fl[tid], ly = (long long)s_fl[T + tid], lz = (long long)s_fl[2 * T + tid];
                f = f3((float)s_f[tid], (float)s_f[T + tid], (float)s_f[2 * T + tid]);
                if (lx | ly | lz) {
                    f = f + f3((float)lx, (float)ly, (float)lz);
                    s_fl[tid] = 0ull; s_fl[T + tid] = 0ull; s_fl[2 * T + tid] = 0ull;
                }
                f = f * (1.f / RES_FSCALE);
            }
            s_f[tid] = 0; s_f[T + tid] = 0; s_f[2 * T + tid] = 0;      // (the next accumulation starts behind the next publication barrier)
        } else {
            __syncthreads();
        }
        if (ext) { f.x -= S.ext_K * (x.x - S.ext_x0); f.y -= S.ext_K * x.y; f.z -= S.ext_K * x.z; }
        forces_valid = true;
    };

    for (int s = 0; s < prog.n_steps; ++s) {
        const long long gstep = prog.gstep0 + s;
        if (prog.cmm_frequency > 0 && ((prog.first_step + s) % prog.cmm_frequency) == 0) {
            // integrators.py:1313: CMMotionRemover at the top of a step: v -= sum(m v) / M; fixed-order sums (deterministic)
            float3 pm = active ? v * frcp(im) : f3(0, 0, 0);
            for (int off = 32; off > 0; off >>= 1) { pm.x += __shfl_xor(pm.x, off); pm.y += __shfl_xor(pm.y, off); pm.z += __shfl_xor(pm.z, off); }
            if ((tid & 63) == 0) { s_red[3 * (tid >> 6)] = pm.x; s_red[3 * (tid >> 6) + 1] = pm.y; s_red[3 * (tid >> 6) + 2] = pm.z; }
            __syncthreads();
            float3 tot = f3(0, 0, 0);
            for (int w = 0; w < nw; ++w) tot = tot + f3(s_red[3 * w], s_red[3 * w + 1], s_red[3 * w + 2]);
            __syncthreads();
            if (active) v = v - tot * S.inv_total_mass;
        }
        for (int t = 0; t < prog.n; ++t) {
            const char tok = prog.tok[t];
            if (tok == 'V') {
                if (!forces_valid) evaluate();
                v = v + f * (prog.hV * im);
            } else if (tok == 'R') {
                x = x + v * prog.hR;
                forces_valid = false;
            } else {
                const uint64_t cnt = (uint64_t)gstep * (uint64_t)prog.nO + (uint64_t)prog.o_index[t];
                const float3 xi = gaussian3(S.seed, REMD_STREAM_OU, (uint32_t)tid, rg, cnt);
                const float sig = prog.b * fsqrt(kT * im);
                v = f3(prog.a * v.x + sig * xi.x, prog.a * v.y + sig * xi.y, prog.a * v.z + sig * xi.z);
            }
        }
    }
    if (active) {
        P[tid] = make_float4(x.x, x.y, x.z, 0.f);
        V[tid] = make_float4(v.x, v.y, v.z, 0.f);
    }
}

void remd_launch_join_wait(remd_ctx* h);
int remd_nb_resident_info(remd_ctx* h, int* ok, int* method, int* has_alch, nb_params* p, const float4** param, const float** rep_lam);
void remd_nb_invalidate_sort(remd_ctx* h);

// returns 1 when the propagation was run by the resident kernel, 0 when the system / request is not one it covers, < 0 on error
static int remd_run_steps_resident(remd_ctx* h, const std::vector<char>& tokens, int nV, int nR, int nO,
                                   int64_t iteration, int64_t first_step, int n_steps)
{
    const bool enabled = !(getenv("REMD_RESIDENT") && atoi(getenv("REMD_RESIDENT")) == 0);      // read per call: the parity tests switch it
    if (!enabled || h->no_resident) return 0;
    if (h->N > 1024 || h->n_settle > 0 || h->n_shake > 0 || h->n_bonds > 0 || h->n_angles > 0 || h->n_torsions > 0) return 0;
    if (h->baro_frequency > 0 || h->profiling == 2 || (int)tokens.size() > MAX_TOK || n_steps < 1) return 0;
    if (h->measure_heat || h->measure_shadow) return 0;
    for (char c : tokens) if (c != 'V' && c != 'R' && c != 'O') return 0;
    int ok = 0, method = -1, alch = 0; nb_params p{}; const float4* param = nullptr; const float* rep_lam = nullptr;
    int rc = remd_nb_resident_info(h, &ok, &method, &alch, &p, &param, &rep_lam);
    if (rc) return rc;
    if (!ok) return 0;
    resident_sys S{};
    S.N = h->N; S.Npad = h->Npad; S.method = method; S.alch = alch; S.n_ext = h->n_ext; S.p = p;
    // skin: a fifth of the cutoff, at most what keeps r_c + skin inside half the smallest box edge (minimum image)
    double lmin = 1e30;
    for (int r = 0; r < h->R; ++r) for (int k = 0; k < 3; ++k) lmin = std::min(lmin, h->box_host.size() >= (size_t)3 * (r + 1) ? h->box_host[3 * r + k] : 1e30);
    S.skin = method >= 0 ? (float)std::max(0.0, std::min(0.2 * p.rc, 0.5 * lmin - p.rc - 1e-3)) : 0.f;
    if (method >= 0 && !(0.5 * lmin > p.rc)) return 0;
    S.ext_K = (float)h->ext_K; S.ext_x0 = (float)h->ext_x0; S.inv_total_mass = (float)(h->total_mass > 0 ? 1.0 / h->total_mass : 0.0);
    S.param = param; S.rep_lam = rep_lam; S.ext_atoms = h->d_ext_atoms; S.invmass = h->d_invmass; S.box = h->d_box;
    S.labels = h->d_labels; S.beta = h->d_beta; S.r_begin = h->r_begin; S.seed = h->seed; S.err = h->d_sync + 2; S.noise_id = h->d_noise_id;
    const int T = std::max(64, (h->N + 63) / 64 * 64);
    // pair-list capacity from the LDS that is left: positions + parameters (32 B per thread), partial sums / flags, force accumulators
    const size_t fixed = (size_t)T * 32 + 64 * sizeof(float) + (size_t)T * 12 + 8 + (size_t)T * 24;
    const size_t lds_max = 144 * 1024;
    S.list_cap = method >= 0 ? (int)std::min<size_t>(32768, (lds_max - fixed) / 4) : 0;
    if (getenv("REMD_RESIDENT_CAP")) S.list_cap = std::max(1, std::min(S.list_cap, atoi(getenv("REMD_RESIDENT_CAP"))));      // test hook: provoke the overflow path
    if (method >= 0 && S.list_cap < 4 * h->N && !getenv("REMD_RESIDENT_CAP")) return 0;
    const size_t lds = fixed + (size_t)S.list_cap * 4;
    resident_prog prog{};
    prog.n = (int)tokens.size();
    int oidx = 0;
    for (int t = 0; t < prog.n; ++t) { prog.tok[t] = tokens[t]; prog.o_index[t] = tokens[t] == 'O' ? oidx++ : 0; }
    prog.hV = (float)(h->dt / (nV > 0 ? nV : 1)); prog.hR = (float)(h->dt / (nR > 0 ? nR : 1));
    const double hO = h->dt / (nO > 0 ? nO : 1);
    prog.a = (float)exp(-h->gamma * hO); prog.b = (float)sqrt(1.0 - exp(-2.0 * h->gamma * hO)); prog.nO = nO > 0 ? nO : 1;
    prog.n_steps = n_steps; prog.cmm_frequency = h->cmm_frequency;
    prog.gstep0 = (long long)iteration * (long long)h->n_steps + first_step; prog.first_step = first_step;
    remd_launch_join_wait(h);
    remd_prof_scope ps(h, "resident_md");
    if (alch) {
        REMD_CHECK(h, hipFuncSetAttribute((const void*)resident_md_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        hipLaunchKernelGGL(resident_md_kernel<true>, dim3(h->R), dim3(T), lds, h->stream, prog, S, h->d_pos, h->d_vel);
    } else {
        REMD_CHECK(h, hipFuncSetAttribute((const void*)resident_md_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        hipLaunchKernelGGL(resident_md_kernel<false>, dim3(h->R), dim3(T), lds, h->stream, prog, S, h->d_pos, h->d_vel);
    }
    REMD_CHECK(h, hipGetLastError());
    h->forces_valid = false; h->force_zeroed = false;
    remd_nb_invalidate_sort(h);            // the atoms moved n_steps without the regular path's evaluation counter seeing it
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Resident small-molecule MD (round 6): the reference's vacuum test systems (AlanineDipeptideVacuum 22 atoms, HostGuestVacuum 156,
// TolueneVacuum 15; testsystems.py:3352-3388) are three dependent launches per MD step on the regular path -- NoCutoff pair sum, listed
// terms, integrator chain -- of which the chain alone is 25 us of fixed latency for 22 atoms (rocprofv3, profiles/r06_43): 39 us per step
// whatever the size.  As for the Lennard-Jones fluids above, ONE launch per propagation: a workgroup owns a replica; a thread owns a
// constraint unit (x, v, 1/m of its <= 4 atoms in registers for all n_steps, the unit arithmetic of the chain kernel: X-H clusters,
// rigid waters, free atoms) AND, for the pair sum, an atom; positions, pair parameters and the fixed-point force accumulators live in LDS.
// A force evaluation is: units publish their positions and clear their atoms' accumulators | barrier | every atom sums its partners in
// ascending order (nocutoff_pair.h: the arithmetic and the order of nocutoff_kernel), exceptions, listed terms (listed_forces_body, the
// accumulators being an LDS address) | barrier.  Every contribution is converted to fixed point exactly as on the regular path and integer
// sums do not depend on their order; with the same Philox streams and the same centre-of-mass sum (per-wavefront fp32 partial sums in
// unit order, then integers) the trajectory follows the regular path to fp32 rounding (one step: velocities within 1 ulp, positions equal;
// the compiler contracts the long expressions of the two kernels differently; tools/experiments/resident_mol_diff.py), like the
// Lennard-Jones kernel above (tests/test_nocutoff.py::test_resident_small_molecule_kernel_follows_the_regular_launches).
// Measured (profiles/r06_43_small_molecule_systems.txt): 24 x AlanineDipeptideVacuum 39 -> 23 us per MD step; a step is then the latency
// of its seven tokens at one wavefront per SIMD (~1 us each, X-H Newton iterations) + one evaluation.  From ~100 atoms on one workgroup
// per replica loses against the regular launches, which spread the listed terms over the chip (CB7:B2 in vacuum, 156 atoms: 72 against
// 61 us per step) -- the kernel takes systems of up to RESIDENT_MOL_MAX_ATOMS atoms.
struct resident_mol_sys {
    int N, Npad, n_units, words, n_exc;
    const float4* nb_param; const unsigned int* excl; const int* exc_atoms; const float4* exc_par;
    const int4* unit_atoms; const unsigned char* unit_type; const float* shake_dist; settle_const sc; float tol;
    const float* invmass; const int64_t* labels; const double* beta; int r_begin; uint64_t seed; const unsigned int* noise_id;
    float inv_total_mass; unsigned int* shake_stat;
    listed_tables L; int n_listed;
};

// one token of the step program on the registers of a unit: the V / R / O branches of run_unit, expression for expression
__device__ __forceinline__ char resident_tok(const resident_prog& prog, int t)
{
    static_assert(MAX_TOK == 24, "six 32-bit words of tokens");
    const unsigned int* w = reinterpret_cast<const unsigned int*>(prog.tok);
    const unsigned int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5];
    const int q = t >> 2;
    const unsigned int x = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : q == 3 ? w3 : q == 4 ? w4 : w5;
    return (char)((x >> ((t & 3) * 8)) & 0xffu);
}

template <int TYPE, int NAT>
__device__ __forceinline__ void resident_mol_token(char tok, const resident_prog& prog, int o_index, long long gstep, const int* idx, const float* dist,
                                                   const settle_const& sc, float tol, const long long* F, int Fs, float kT, uint32_t rg, uint64_t seed,
                                                   unit_regs& S)
{
    float3 (&x)[4] = S.x; float3 (&v)[4] = S.v;
    float (&im)[4] = S.im;
    if (tok == 'V') {
        const float hv = prog.hV;
#pragma unroll
        for (int k = 0; k < NAT; ++k) {
            const float s = hv * im[k] * (1.0f / 4294967296.0f);
            v[k].x += s * (float)F[idx[k]];
            v[k].y += s * (float)F[Fs + idx[k]];
            v[k].z += s * (float)F[2 * Fs + idx[k]];
        }
        constrain_v<TYPE, NAT>(sc, im, tol, v, x);
    } else if (tok == 'R') {
        if (TYPE == UNIT_FREE) {
#pragma unroll
            for (int k = 0; k < NAT; ++k) x[k] = x[k] + v[k] * prog.hR;
        } else {
            float3 p0[NAT], p1[NAT], q[NAT];
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                p0[k] = x[k] - x[0];
                p1[k] = p0[k] + v[k] * prog.hR;
                q[k] = p1[k];
            }
            if (TYPE == UNIT_SETTLE) settle_positions(sc, p0, p1);
            else S.shake_it = max(S.shake_it, shake_positions<NAT>(im, dist, tol, p0, p1));
            const float ih = frcp(prog.hR);
            const float3 org = x[0];
#pragma unroll
            for (int k = 0; k < NAT; ++k) {
                v[k] = v[k] + (p1[k] - q[k]) * ih;
                x[k] = org + p1[k];
            }
            constrain_v<TYPE, NAT>(sc, im, tol, v, x);
        }
    } else if (tok == 'O') {
        const uint64_t cnt = (uint64_t)gstep * (uint64_t)prog.nO + (uint64_t)o_index;
#pragma unroll
        for (int k = 0; k < NAT; ++k) {
            const float3 xi = gaussian3(seed, REMD_STREAM_OU, (uint32_t)idx[k], rg, cnt);
            const float sig = prog.b * fsqrt(kT * im[k]);
            v[k].x = prog.a * v[k].x + sig * xi.x;
            v[k].y = prog.a * v[k].y + sig * xi.y;
            v[k].z = prog.a * v[k].z + sig * xi.z;
        }
        constrain_v<TYPE, NAT>(sc, im, tol, v, x);
    }
}

#define RESIDENT_MOL_T 256
#define RESIDENT_MOL_MAX_ATOMS 64
__global__ __launch_bounds__(RESIDENT_MOL_T)
void resident_mol_kernel(resident_prog prog, resident_mol_sys S, float4* __restrict__ pos, float4* __restrict__ vel)
{
    __shared__ float4 s_pos[RESIDENT_MOL_T], s_par[RESIDENT_MOL_T];
    __shared__ long long s_F[3 * RESIDENT_MOL_T];
    __shared__ long long s_pm[RESIDENT_MOL_T / 64][3];
    constexpr int Fs = RESIDENT_MOL_T;
    const int tid = threadIdx.x, r = blockIdx.x, N = S.N;
    float4* P = pos + (size_t)r * S.Npad;
    float4* V = vel + (size_t)r * S.Npad;
    int4 a4 = make_int4(-1, -1, -1, -1);
    int type = UNIT_FREE;
    float dist[3] = { 0.f, 0.f, 0.f };
    if (tid < S.n_units) {
        a4 = S.unit_atoms[tid]; type = (int)S.unit_type[tid];
        dist[0] = S.shake_dist[tid * 3]; dist[1] = S.shake_dist[tid * 3 + 1]; dist[2] = S.shake_dist[tid * 3 + 2];
    }
    const bool active = a4.x >= 0;
    if (!active) type = UNIT_FREE;
    const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
    unit_regs U;
    U.have_cm = 0; U.shake_it = 0; U.heat = 0.f; U.shadow = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        U.x[k] = f3(0, 0, 0); U.v[k] = f3(0, 0, 0); U.im[k] = 0.f;
        if (idx[k] >= 0) {
            const float4 p = P[idx[k]], w = V[idx[k]];
            U.x[k] = f3(p.x, p.y, p.z); U.v[k] = f3(w.x, w.y, w.z); U.im[k] = S.invmass[idx[k]];
        }
    }
    s_par[tid] = tid < N ? S.nb_param[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    s_pos[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float kT = frcp((float)S.beta[S.labels[S.r_begin + r]]);
    const uint32_t rg = S.noise_id ? S.noise_id[r] : (uint32_t)(S.r_begin + r);
    bool forces_valid = false;
    __syncthreads();

    auto evaluate = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (idx[k] >= 0) {
                s_pos[idx[k]] = make_float4(U.x[k].x, U.x[k].y, U.x[k].z, 0.f);
                s_F[idx[k]] = 0; s_F[Fs + idx[k]] = 0; s_F[2 * Fs + idx[k]] = 0;
            }
        }
        __syncthreads();
        if (tid < N) {
            const float4 xi = s_pos[tid], pi = s_par[tid];
            float fx = 0.f, fy = 0.f, fz = 0.f;
            double e = 0.0;
            const unsigned int* mrow = S.excl + (size_t)tid * S.words;
            for (int j0 = 0; j0 < N; j0 += 32) {
                const unsigned int m = mrow[j0 >> 5];
                const int jn = min(32, N - j0);
                for (int k = 0; k < jn; ++k) {
                    const int j = j0 + k;
                    if (j == tid || ((m >> k) & 1u)) continue;
                    nocutoff_pair<false>(xi, pi, s_pos[j], s_par[j], fx, fy, fz, e);
                }
            }
            add_force(s_F, Fs, tid, fx, fy, fz);
        }
        for (int t = tid; t < S.n_exc; t += RESIDENT_MOL_T) {
            const int i = S.exc_atoms[2 * t], j = S.exc_atoms[2 * t + 1];
            const float4 par = S.exc_par[t];
            const float3 d = sub3(ld3(s_pos, j), ld3(s_pos, i));
            double e = 0.0;
            const float fr = nocutoff_exception<false>(par, d, e);
            add_force(s_F, Fs, i, fr * d.x, fr * d.y, fr * d.z);
            add_force(s_F, Fs, j, -fr * d.x, -fr * d.y, -fr * d.z);
        }
        // (every lane of a wavefront takes part in listed_forces_body's reduction over the lanes of one atom)
        for (int base = 0; base < S.n_listed; base += RESIDENT_MOL_T)
            listed_forces_body(S.L, Fs, s_pos, (const float*)nullptr, s_F, base + tid, 0);
        __syncthreads();
        forces_valid = true;
    };

    for (int s = 0; s < prog.n_steps; ++s) {
        const long long gstep = prog.gstep0 + s;
        if (prog.cmm_frequency > 0 && ((prog.first_step + s) % prog.cmm_frequency) == 0) {
            // CMMotionRemover at the top of a step (integrators.py:1313): the sum of the chain kernel -- fp32 over a unit's atoms and the
            // units of a wavefront, then fixed point
            float3 pm = f3(0, 0, 0);
            if (active) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (idx[k] >= 0) pm = pm + U.v[k] * frcp(U.im[k]);
            }
            for (int off = 32; off > 0; off >>= 1) { pm.x += __shfl_xor(pm.x, off); pm.y += __shfl_xor(pm.y, off); pm.z += __shfl_xor(pm.z, off); }
            if ((tid & 63) == 0) {
                long long* w = s_pm[tid >> 6];
                w[0] = (long long)((double)pm.x * 4294967296.0); w[1] = (long long)((double)pm.y * 4294967296.0); w[2] = (long long)((double)pm.z * 4294967296.0);
            }
            __syncthreads();
            long long tot[3] = { 0, 0, 0 };
            for (int w = 0; w < RESIDENT_MOL_T / 64; ++w) { tot[0] += s_pm[w][0]; tot[1] += s_pm[w][1]; tot[2] += s_pm[w][2]; }
            __syncthreads();
            const float sx = (float)tot[0] * (1.0f / 4294967296.0f) * S.inv_total_mass;
            const float sy = (float)tot[1] * (1.0f / 4294967296.0f) * S.inv_total_mass;
            const float sz = (float)tot[2] * (1.0f / 4294967296.0f) * S.inv_total_mass;
#pragma unroll
            for (int k = 0; k < 4; ++k) { U.v[k].x -= sx; U.v[k].y -= sy; U.v[k].z -= sz; }
        }
        int o_index = 0;
        for (int t = 0; t < prog.n; ++t) {
            // (the token from six registers loaded with the kernel arguments, the O counter kept here: a dynamic index into the argument
            //  arrays is a scalar memory load per token on a path that is all latency, see chain_tok)
            const char tok = resident_tok(prog, t);
            if (tok == 'V' && !forces_valid) evaluate();
            if (tok == 'R') forces_valid = false;
            const int o_now = o_index;
            if (tok == 'O') ++o_index;
            if (active) {
#define RUN(TY, NA) resident_mol_token<TY, NA>(tok, prog, o_now, gstep, idx, dist, S.sc, S.tol, s_F, Fs, kT, rg, S.seed, U)
                if (type == UNIT_SETTLE) RUN(UNIT_SETTLE, 3);
                else if (type == UNIT_FREE) { if (a4.y < 0) RUN(UNIT_FREE, 1); else RUN(UNIT_FREE, 4); }
                else if (a4.z < 0) RUN(UNIT_SHAKE, 2);
                else if (a4.w < 0) RUN(UNIT_SHAKE, 3);
                else RUN(UNIT_SHAKE, 4);
#undef RUN
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (idx[k] >= 0) {
            P[idx[k]] = make_float4(U.x[k].x, U.x[k].y, U.x[k].z, 0.f);
            V[idx[k]] = make_float4(U.v[k].x, U.v[k].y, U.v[k].z, 0.f);
        }
    }
    if (type == UNIT_SHAKE && U.shake_it > 0 &&
        (unsigned int)U.shake_it > __hip_atomic_load(S.shake_stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(S.shake_stat, (unsigned int)U.shake_it);
}

// returns 1 when the propagation was run by the resident small-molecule kernel, 0 when the system / request is not one it covers
static int remd_run_steps_resident_mol(remd_ctx* h, const std::vector<char>& tokens, int nV, int nR, int nO,
                                       int64_t iteration, int64_t first_step, int n_steps)
{
    const bool enabled = !(getenv("REMD_RESIDENT") && atoi(getenv("REMD_RESIDENT")) == 0);      // read per call: the parity tests switch it
    if (!enabled || h->no_resident) return 0;
    if (!h->nocutoff || h->gbsa || h->n_regions > 0 || h->nb_method != REMD_NB_NONE || h->n_ext > 0) return 0;
    const unit_tables& ut = g_units[h];
    if (h->N > RESIDENT_MOL_MAX_ATOMS || ut.n_units > RESIDENT_MOL_T || ut.n_units < 1) return 0;
    if (getenv("REMD_RESIDENT_MOL") && atoi(getenv("REMD_RESIDENT_MOL")) == 0) return 0;
    if (h->baro_frequency > 0 || h->profiling == 2 || (int)tokens.size() > MAX_TOK || n_steps < 1) return 0;
    if (h->measure_heat || h->measure_shadow) return 0;
    for (char c : tokens) if (c != 'V' && c != 'R' && c != 'O') return 0;
    resident_mol_sys S{};
    S.N = h->N; S.Npad = h->Npad; S.n_units = ut.n_units;
    if (remd_nocutoff_info(h, &S.nb_param, &S.excl, &S.words, &S.n_exc, &S.exc_atoms, &S.exc_par)) return 0;
    S.unit_atoms = ut.d_atoms; S.unit_type = ut.d_type; S.shake_dist = ut.d_dist; S.sc = ut.sc;
    S.tol = (float)fmax(h->constraint_tol, REMD_CONSTRAINT_TOL_FLOOR);
    S.invmass = h->d_invmass; S.labels = h->d_labels; S.beta = h->d_beta; S.r_begin = h->r_begin; S.seed = h->seed; S.noise_id = h->d_noise_id;
    S.inv_total_mass = (float)(h->total_mass > 0 ? 1.0 / h->total_mass : 0.0);
    S.shake_stat = h->d_sync + 3;
    listed_tables L{};
    L.n_bonds = h->n_bonds; L.n_angles = h->n_angles; L.n_torsions = h->n_torsions;
    L.bond_atoms = h->d_bond_atoms; L.bond_params = h->d_bond_params;
    L.angle_atoms = h->d_angle_atoms; L.angle_params = h->d_angle_params;
    L.torsion_atoms = h->d_torsion_atoms; L.torsion_params = h->d_torsion_params;
    S.n_listed = L.n_bonds + L.n_angles + L.n_torsions;
    if (S.n_listed > 0 && h->d_aterm && h->n_aterm > 0) { L.aterm = h->d_aterm; L.n_aterm = h->n_aterm; S.n_listed = h->n_aterm; }
    S.L = L;
    resident_prog prog{};
    prog.n = (int)tokens.size();
    int oidx = 0;
    for (int t = 0; t < prog.n; ++t) { prog.tok[t] = tokens[t]; prog.o_index[t] = tokens[t] == 'O' ? oidx++ : 0; }
    prog.hV = (float)(h->dt / (nV > 0 ? nV : 1)); prog.hR = (float)(h->dt / (nR > 0 ? nR : 1));
    const double hO = h->dt / (nO > 0 ? nO : 1);
    prog.a = (float)exp(-h->gamma * hO); prog.b = (float)sqrt(1.0 - exp(-2.0 * h->gamma * hO)); prog.nO = nO > 0 ? nO : 1;
    prog.n_steps = n_steps; prog.cmm_frequency = h->cmm_frequency;
    prog.gstep0 = (long long)iteration * (long long)h->n_steps + first_step; prog.first_step = first_step;
    remd_launch_join_wait(h);
    remd_prof_scope ps(h, "resident_md");
    hipLaunchKernelGGL(resident_mol_kernel, dim3(h->R), dim3(RESIDENT_MOL_T), 0, h->stream, prog, S, h->d_pos, h->d_vel);
    REMD_CHECK(h, hipGetLastError());
    h->forces_valid = false; h->force_zeroed = false;
    return 1;
}

// Runs n_steps of the token program.  Tokens are grouped into chains that need no new
// force evaluation; a 'V' after an 'R' forces a force evaluation first.
//
// One loop body = one MD step's launches: the chain(s) around the centre-of-mass removal and the force evaluation on two streams.
// (Round 2 also captured the body into a hipGraph and replayed it: bit-identical and no faster on ROCm 7.2 -- the floor of a step is
// the dependent chain of kernels, not the host -- so the capture path was removed in round 3; DESIGN.md section 7b has the numbers.)
void remd_launch_join_wait(remd_ctx* h);
void remd_nb_tune_step(remd_ctx* h, int steps_left_in_call);

// The state remd_run_steps keeps between the MD steps of one call, as an object: begin() = everything in front of the step loop,
// step(s) = one MD step's launches, end() = the flush behind the last step.  One handle runs begin / step ... / end by itself
// (remd_run_steps); round 6: SEVERAL handles of one device take turns step by step from one host thread (remd_run_steps_many: the
// chain of one group of replicas beside the force kernels of another) -- nothing in here synchronises with the host.
struct step_runner {
    remd_ctx* h = nullptr; const unit_tables* ut = nullptr; const std::vector<char>* tokens = nullptr;
    int64_t first_step = 0; int n_steps = 0;
    chain_prog base{}, cur{};
    bool mts = false; unsigned group_mask[4] = {0u, 0u, 0u, 0u}; bool group_valid[4] = {false, false, false, false};
    bool shadow = false, pe_valid = false, zeroed_by_chain = false, device_waits_ok = false, merge_cmm = false;
    int cmm_w = 0; long long gstep0 = 0;
    bool done_by_resident = false;       // the resident small-system kernel took the whole call: step() / end() do nothing

    int begin(remd_ctx* h_, const std::vector<char>& tokens_, int nV, int nR, int nO, int64_t iteration, int64_t first_step_, int n_steps_)
    {
        h = h_; tokens = &tokens_; first_step = first_step_; n_steps = n_steps_;
        ut = &g_units[h];
        if (ut->n_units == 0) return remd_fail(h, -3, "no system set");
        {
            const int rr = remd_run_steps_resident(h, tokens_, nV, nR, nO, iteration, first_step, n_steps);
            if (rr < 0) return rr;
            if (rr != 0) { done_by_resident = true; return 0; }
            const int rm = remd_run_steps_resident_mol(h, tokens_, nV, nR, nO, iteration, first_step, n_steps);
            if (rm < 0) return rm;
            if (rm != 0) { done_by_resident = true; return 0; }
        }
        base = chain_prog{};
        base.hV = (float)(h->dt / (nV > 0 ? nV : 1));
        base.hR = (float)(h->dt / (nR > 0 ? nR : 1));
        const double hO = h->dt / (nO > 0 ? nO : 1);                 // integrators.py:1142
        base.a = (float)exp(-h->gamma * hO);                         // :1143
        base.b = (float)sqrt(1.0 - exp(-2.0 * h->gamma * hO));       // :1146
        base.nO = nO > 0 ? nO : 1;
        base.cmm_r = -1; base.cmm_w = 0; base.zero_force = 0;
        // multiple-time-step program: one force array per force group that the splitting names, evaluated when a V of the group
        // comes up and the positions have changed since its last evaluation
        for (char c : tokens_) mts |= (c >= '0' && c <= '3');
        if (mts) {
            const size_t nf = (size_t)h->R * 3 * h->Npad;
            for (int g = 0; g < 4; ++g) {
                base.hVg[g] = h->nVg[g] > 0 ? (float)(h->dt / h->nVg[g]) : 0.f;
                for (int c = 0; c < 6; ++c) if (h->fgroup[c] == g) group_mask[g] |= 1u << c;
                if (h->nVg[g] > 0 && (!h->d_force_g[g] || h->force_g_n != nf)) {
                    if (h->d_force_g[g]) { REMD_CHECK(h, hipStreamSynchronize(h->stream)); hipFree(h->d_force_g[g]); h->d_force_g[g] = nullptr; }
                    REMD_CHECK(h, hipMalloc(&h->d_force_g[g], sizeof(long long) * nf));
                }
                base.Fg[g] = h->d_force_g[g];
            }
            h->force_g_n = nf;
            unsigned named = 0u;
            for (int g = 0; g < 4; ++g) if (h->nVg[g] > 0) named |= group_mask[g];
            for (int c = 0; c < 6; ++c)
                if (h->fgroup[c] > 3 || !(named & (1u << c)))
                    return remd_fail(h, -3, "multiple-time-step splitting: a force class sits in a force group that no V of the splitting names "
                                            "(its forces would never act); groups 0-3 are supported");
        }
        int n_braces = 0;
        for (char c : tokens_) n_braces += (c == '}');
        shadow = h->measure_shadow || n_braces > 0;                         // a Metropolized program measures shadow work (:1117-1119)
        base.measure = (h->measure_heat ? 1 : 0) | (shadow ? 2 : 0);
        if (base.measure) { int rcw = remd_work_buffers(h); if (rcw) return rcw; }
        pe_valid = false;                  // d_pe_prev holds U at the current positions
        cur = base; cur.n = 0;
        cmm_w = 0;                         // accumulator the next momentum sum goes to
        zeroed_by_chain = false;
        gstep0 = (long long)iteration * (long long)h->n_steps + first_step;
        // the centre-of-mass motion remover needs sum(m v) over the whole replica between two tokens of a step: either two launches
        // (the first ends with the sum) or one launch with a barrier over the replica's workgroups in device memory ('M' token) --
        // every workgroup of the grid must then be resident at once, hence the bound on the grid
        const bool merge_env = !(getenv("REMD_CHAIN_MERGE") && atoi(getenv("REMD_CHAIN_MERGE")) == 0);
        const long long chain_blocks = (long long)((ut->n_units + 255) / 256) * h->R;
        // (the same bound holds for the join polled in the chain's prologue: spinning workgroups of a grid larger than the chip
        // holds at once could keep the direct-space stream's last launches from ever being dispatched)
        device_waits_ok = chain_blocks <= 1024 && !h->no_device_waits;
        merge_cmm = merge_env && device_waits_ok && h->profiling != 2 && !h->lean_waits && !h->no_chain_barrier && !h->no_chain_merge;
        const long long sync_key = (long long)h->R * 1000003ll + ut->n_units;
        if (merge_cmm && (!h->d_chain_sync || h->chain_sync_key != sync_key)) {      // slots of THIS grid shape
            if (h->d_chain_sync) { REMD_CHECK(h, hipStreamSynchronize(h->stream)); hipFree(h->d_chain_sync); h->d_chain_sync = nullptr; }
            h->chain_sync_key = sync_key;
            const size_t slot_bytes = sizeof(unsigned long long) * 2 * (size_t)h->R * (size_t)((ut->n_units + 255) / 256) * 3;   // [2][R][workgroups][3]
            REMD_CHECK(h, hipMalloc((void**)&h->d_chain_sync, slot_bytes));
            REMD_CHECK(h, hipMemsetAsync(h->d_chain_sync, 0, slot_bytes, h->stream));
            h->chain_sync_epoch = 0;
        }
        if (h->cmm_frequency > 0)
            hipMemsetAsync(h->d_cmm, 0, sizeof(long long) * 4 * 2 * h->R, h->stream);      // both accumulators, once per call
        return 0;
    }

    void flush(bool accumulate, bool bin_for_pme = false)      // bin_for_pme: a force evaluation follows this launch directly
    {
        if (cur.n == 0 && !accumulate) return;
        cur.accumulate_momentum = accumulate ? 1 : 0;
        cur.cmm_w = cmm_w;
        // forces are stale after an R that follows the chain's last V: let the chain clear them (saves a memset)
        bool seenR = false, staleAtEnd = false;
        for (int t = 0; t < cur.n; ++t) { if (cur.tok[t] == 'R') seenR = true; if (cur.tok[t] == 'V') seenR = false; }
        staleAtEnd = seenR && !mts;          // (the per-group arrays of a multiple-time-step program are cleared before their evaluation)
        cur.zero_force = staleAtEnd ? 1 : 0;
        if (staleAtEnd) zeroed_by_chain = true;
        launch_chain(h, *ut, cur, bin_for_pme);
        cur = base; cur.n = 0;
    }
    void push(char tok, int oidx, long long step)
    {
        if (cur.n == MAX_TOK) flush(false);
        cur.tok[cur.n] = tok; cur.o_index[cur.n] = oidx; cur.step[cur.n] = step; cur.n++;
    }
    int evaluate_with_energy(bool accumulate)
    {
        // energies (and forces) at the current positions; accumulate: add U - U_prev to the shadow work (:1420-1423)
        flush(false);
        h->force_zeroed = zeroed_by_chain;
        zeroed_by_chain = false;
        int rc = remd_compute_forces(h, true);
        if (rc) return rc;
        hipLaunchKernelGGL(work_pe_kernel, dim3((h->R + 63) / 64), dim3(64), 0, h->stream, h->R, h->d_potential, h->d_pe_prev, h->d_work, accumulate ? 1 : 0);
        pe_valid = true;
        return 0;
    }

    int step(int s)
    {
        if (done_by_resident) return 0;
        const long long gstep = gstep0 + s;
        // integrators.py:1313 addUpdateContextState: CMMotionRemover fires at the top of a step
        if (h->cmm_frequency > 0 && ((first_step + s) % h->cmm_frequency) == 0) {
            bool pending_reads_cmm = false;
            for (int t = 0; t < cur.n; ++t) pending_reads_cmm |= (cur.tok[t] == 'C');
            if (pending_reads_cmm) flush(false);   // it must see its own accumulator before the next sum starts
            if (merge_cmm && cur.n > 0 && cur.n + 2 <= MAX_TOK) {
                // ONE launch: pending tokens, 'M' (sum(m v) into buffer cmm_w + barrier over the replica's workgroups), 'C', ...
                push('M', 0, gstep);
                cur.m_buf = cmm_w; cur.m_epoch = ++h->chain_sync_epoch;
            } else {
                flush(true);                    // finishes pending tokens and accumulates sum(m v) into buffer cmm_w
            }
            push('C', 0, gstep);
            cur.cmm_r = cmm_w;                  // this chain subtracts P/M from that buffer and clears the other one
            cmm_w = 1 - cmm_w;
        }
        // Monte Carlo barostat (NPT states): acts in the same updateContextState slot, every baro_frequency-th step
        if (h->baro_frequency > 0 && (++h->baro_steps % h->baro_frequency) == 0) {
            flush(false);
            h->force_zeroed = zeroed_by_chain;
            zeroed_by_chain = false;
            int rc = remd_barostat_attempt(h);
            if (rc) return rc;
            for (bool& gv : group_valid) gv = false;
        }
        int oidx = 0, brace = 0;
        for (char tok : *tokens) {
            if (tok == '{') { push('{', 0, gstep); continue; }
            if (tok == '}') {
                flush(false);
                hipLaunchKernelGGL(metropolis_kernel, dim3((h->R + 63) / 64), dim3(64), 0, h->stream, h->R, h->r_begin, h->seed, gstep, brace++,
                                   h->d_labels, h->d_beta, h->d_work, h->d_accept, h->d_noise_id);
                hipLaunchKernelGGL(metropolis_restore_kernel, dim3((h->N + 255) / 256, h->R), dim3(256), 0, h->stream, h->N, h->Npad, h->d_accept,
                                   h->d_pos, h->d_vel, h->d_xold, h->d_vold);
                h->forces_valid = false; pe_valid = false;         // rejected replicas are back at their old positions
                for (bool& gv : group_valid) gv = false;
                continue;
            }
            if (tok == 'R' && shadow && !pe_valid) { int rc = evaluate_with_energy(false); if (rc) return rc; }
            if (tok >= '0' && tok <= '3' && !group_valid[tok - '0']) {
                // forces of this force group at the current positions, into the group's own array
                const int g = tok - '0';
                const bool has_mesh = (group_mask[g] >> REMD_FG_RECIPROCAL) & 1u;
                flush(false, has_mesh);
                long long* all_forces = h->d_force;
                h->d_force = h->d_force_g[g];
                h->force_zeroed = false; zeroed_by_chain = false;
                h->defer_join_ok = device_waits_ok && !h->lean_waits;
                int rc = remd_compute_forces(h, false, group_mask[g]);
                h->defer_join_ok = false;
                h->d_force = all_forces;
                h->forces_valid = false; h->force_zeroed = false;      // (the all-forces accumulator was not touched)
                if (rc) return rc;
                group_valid[g] = true;
            }
            if (tok == 'V' && !h->forces_valid) {
                flush(false, true);
                h->force_zeroed = zeroed_by_chain;
                zeroed_by_chain = false;
                h->defer_join_ok = device_waits_ok && !h->lean_waits;   // the next main-stream launch is the chain holding this V
                int rc = remd_compute_forces(h, false);
                h->defer_join_ok = false;
                if (rc) return rc;
            }
            push(tok, tok == 'O' ? oidx : 0, gstep);
            if (tok == 'O') oidx++;
            if (tok == 'R') {
                h->forces_valid = false;
                for (bool& gv : group_valid) gv = false;
                if (shadow) { int rc = evaluate_with_energy(true); if (rc) return rc; }
            }
        }
        return 0;
    }

    int end()
    {
        if (done_by_resident) return 0;
        flush(false);
        remd_launch_join_wait(h);
        h->force_zeroed = zeroed_by_chain;
        REMD_CHECK(h, hipGetLastError());
        return 0;
    }
};

// workgroups of one integrator-chain launch of this handle (one per 256 constraint units per replica)
long long remd_chain_blocks(remd_ctx* h)
{
    const unit_tables* ut = g_units.find(h);
    return ut ? (long long)((ut->n_units + 255) / 256) * h->R : 0;
}

int remd_run_steps(remd_ctx* h, const std::vector<char>& tokens, int nV, int nR, int nO,
                   int64_t iteration, int64_t first_step, int n_steps)
{
    step_runner sr;
    int rc = sr.begin(h, tokens, nV, nR, nO, iteration, first_step, n_steps);
    if (rc) return rc;
    for (int s = 0; s < n_steps && !sr.done_by_resident; ++s) {
        remd_nb_tune_step(h, n_steps - s);
        rc = sr.step(s); if (rc) return rc;
    }
    return sr.end();
}

// Several handles of ONE device, one host thread, the MD steps of the handles taking turns: handle 0's step s, handle 1's step s, ...
// Each handle keeps its own pair of streams, so the integrator chain of one group of replicas (a few hundred wavefronts waiting for one
// dependent thing after another) runs beside the pair and mesh kernels of another.  Nothing here waits for the device.
int remd_run_steps_many(remd_ctx** hs, int n, int64_t iteration, int64_t first_step, int n_steps)
{
    std::vector<step_runner> sr((size_t)n);
    for (int i = 0; i < n; ++i) {
        hipSetDevice(hs[i]->device);
        int rc = sr[i].begin(hs[i], hs[i]->tokens, hs[i]->nV, hs[i]->nR, hs[i]->nO, iteration, first_step, n_steps);
        if (rc) return rc;
    }
    for (int s = 0; s < n_steps; ++s)
        for (int i = 0; i < n; ++i) {
            if (sr[i].done_by_resident) continue;
            remd_nb_tune_step(hs[i], n_steps - s);
            int rc = sr[i].step(s); if (rc) return rc;
        }
    for (int i = 0; i < n; ++i) { int rc = sr[i].end(); if (rc) return rc; }
    return 0;
}


int remd_assign_velocities(remd_ctx* h, int64_t iteration)
{
    const unit_tables& ut = g_units[h];
    remd_prof_scope ps(h, "assign_velocities");
    dim3 grid((ut.n_units + 255) / 256, h->R);
    hipLaunchKernelGGL(assign_velocities_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, ut.d_type, ut.sc,
                       (float)fmax(h->constraint_tol, REMD_CONSTRAINT_TOL_FLOOR), h->Npad, h->d_pos, h->d_vel, h->d_invmass, h->d_labels,
                       h->d_beta, h->r_begin, h->seed, iteration, h->d_noise_id);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

int remd_kinetic_energy(remd_ctx* h)
{
    remd_prof_scope ps(h, "kinetic_energy");
    hipLaunchKernelGGL(kinetic_energy_kernel, dim3(h->R), dim3(256), 0, h->stream, h->N, h->Npad, h->d_vel, h->d_mass, h->d_kinetic);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

int remd_check_finite(remd_ctx* h)
{
    REMD_CHECK(h, hipMemsetAsync(h->d_nan, 0, sizeof(int) * h->R, h->stream));
    dim3 grid((h->N + 255) / 256, h->R);
    hipLaunchKernelGGL(check_finite_kernel, grid, dim3(256), 0, h->stream, h->N, h->Npad, h->d_pos, h->d_vel, h->d_nan);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// FIRE minimisation (MultiStateSampler.minimize, multistatesampler.py:611-647, _minimize_replica :1351-1434, with the
// reference's FIREMinimizationIntegrator, integrators.py:2290-2469), all local replicas at once.  One reference step:
//   converged if |f| / ndof <= ftol (:2377-2386);  x0 = x, v0 = v, E0 = U(x) (:2392-2394);
//   v += dt/2 f/m;  x += dt v;  constrain x;  v += dt/2 f(x_new)/m + (x - x1)/dt;  constrain v (:2397-2402);
//   dE = U(x_new) - E0;  P = f.v;  v = (1 - alpha) v + alpha f/|f| |v| (:2404-2421);
//   restart (x = x0, v = v0, P = -1) unless dE < 0 (:2423-2431);  converged if dt <= 1e-5 timestep (:2433-2437);
//   P > 0: N_neg += 1, beyond N_min: dt = min(dt f_inc, dt_max), alpha *= f_alpha (:2439-2449);
//   P < 0: N_neg = 0, dt *= f_dec, v = 0, alpha = alpha_start (:2451-2458).
// Three kernels around one energy + force evaluation per step; the per-replica scalars are double buffered.
struct fire_rep { float dt, alpha; int n_neg, converged; double E, f2; };     // E, f2 = U and sum f^2 at the current x
struct fire_consts { float dt_max, f_inc, f_dec, alpha0, f_alpha, dt_min, ftol, ndof; int n_min; };

template <int TYPE, int NAT>
__device__ __forceinline__ void fire_move_unit(const int* idx, const float* dist, const settle_const& sc, float tol, int Npad,
                                               float4* __restrict__ P, float4* __restrict__ V, const long long* __restrict__ F,
                                               float4* __restrict__ X0, float4* __restrict__ V0, long long* __restrict__ F0,
                                               const float* __restrict__ invmass, float dt)
{
    float3 x[NAT], v[NAT];
    float im[NAT];
#pragma unroll
    for (int k = 0; k < NAT; ++k) {
        const float4 p = P[idx[k]], w = V[idx[k]];
        X0[idx[k]] = p; V0[idx[k]] = w;
        const long long fx = F[idx[k]], fy = F[Npad + idx[k]], fz = F[2 * Npad + idx[k]];
        F0[idx[k]] = fx; F0[Npad + idx[k]] = fy; F0[2 * Npad + idx[k]] = fz;
        x[k] = f3(p.x, p.y, p.z); v[k] = f3(w.x, w.y, w.z);
        im[k] = invmass[idx[k]];
        const float s = 0.5f * dt * im[k] * (1.0f / 4294967296.0f);
        v[k].x += s * (float)fx; v[k].y += s * (float)fy; v[k].z += s * (float)fz;
    }
    if (TYPE == UNIT_FREE) {
#pragma unroll
        for (int k = 0; k < NAT; ++k) x[k] = x[k] + v[k] * dt;
    } else {
        float3 p0[NAT], p1[NAT], q[NAT];
#pragma unroll
        for (int k = 0; k < NAT; ++k) { p0[k] = x[k] - x[0]; p1[k] = p0[k] + v[k] * dt; q[k] = p1[k]; }
        if (TYPE == UNIT_SETTLE) settle_positions(sc, p0, p1);
        else (void)shake_positions<NAT>(im, dist, tol, p0, p1);
        const float ih = frcp(dt);
        const float3 org = x[0];
#pragma unroll
        for (int k = 0; k < NAT; ++k) { v[k] = v[k] + (p1[k] - q[k]) * ih; x[k] = org + p1[k]; }
    }
#pragma unroll
    for (int k = 0; k < NAT; ++k) {
        P[idx[k]] = make_float4(x[k].x, x[k].y, x[k].z, 0.f);
        V[idx[k]] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
    }
}

__global__ __launch_bounds__(256)
void fire_move_kernel(int n_units, const int4* __restrict__ unit_atoms, const unsigned char* __restrict__ unit_type,
                      const float* __restrict__ shake_dist, settle_const sc, float tol, int Npad, float4* __restrict__ pos,
                      float4* __restrict__ vel, const long long* __restrict__ force, float4* __restrict__ x0,
                      float4* __restrict__ v0, long long* __restrict__ f0, const float* __restrict__ invmass,
                      const fire_rep* __restrict__ state)
{
    const int uidx = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    const fire_rep s = state[r];
    if (s.converged || uidx >= n_units) return;
    const int4 a4 = unit_atoms[uidx];
    if (a4.x < 0) return;
    const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
    const int type = unit_type[uidx];
    float dist[3] = { 0.f, 0.f, 0.f };
    if (type == UNIT_SHAKE) { dist[0] = shake_dist[uidx * 3]; dist[1] = shake_dist[uidx * 3 + 1]; dist[2] = shake_dist[uidx * 3 + 2]; }
    const size_t o = (size_t)r * Npad, of = (size_t)r * 3 * Npad;
#define RUN(TY, NA) fire_move_unit<TY, NA>(idx, dist, sc, tol, Npad, pos + o, vel + o, force + of, x0 + o, v0 + o, f0 + of, invmass, s.dt)
    if (type == UNIT_SETTLE) RUN(UNIT_SETTLE, 3);
    else if (type == UNIT_FREE) { if (a4.y < 0) RUN(UNIT_FREE, 1); else RUN(UNIT_FREE, 4); }
    else if (a4.z < 0) RUN(UNIT_SHAKE, 2);
    else if (a4.w < 0) RUN(UNIT_SHAKE, 3);
    else RUN(UNIT_SHAKE, 4);
#undef RUN
}

// second half kick with the new forces, velocity constraints, and the per-workgroup partial sums of f.f, v.v, f.v
template <int TYPE, int NAT>
__device__ __forceinline__ void fire_finish_unit(const int* idx, const settle_const& sc, float tol, int Npad,
                                                 const float4* __restrict__ P, float4* __restrict__ V, const long long* __restrict__ F,
                                                 const float* __restrict__ invmass, float dt, bool kick, double* sums)
{
    float3 x[NAT], v[NAT], f[NAT];
    float im[NAT];
#pragma unroll
    for (int k = 0; k < NAT; ++k) {
        const float4 p = P[idx[k]], w = V[idx[k]];
        x[k] = f3(p.x, p.y, p.z); v[k] = f3(w.x, w.y, w.z);
        im[k] = invmass[idx[k]];
        f[k] = f3((float)F[idx[k]] * (1.0f / 4294967296.0f), (float)F[Npad + idx[k]] * (1.0f / 4294967296.0f),
                  (float)F[2 * Npad + idx[k]] * (1.0f / 4294967296.0f));
        if (kick) v[k] = v[k] + f[k] * (0.5f * dt * im[k]);
    }
    if (kick) {
        constrain_v<TYPE, NAT>(sc, im, tol, v, x);
#pragma unroll
        for (int k = 0; k < NAT; ++k) V[idx[k]] = make_float4(v[k].x, v[k].y, v[k].z, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NAT; ++k) { sums[0] += (double)dot3(f[k], f[k]); sums[1] += (double)dot3(v[k], v[k]); sums[2] += (double)dot3(f[k], v[k]); }
}

__global__ __launch_bounds__(256)
void fire_finish_kernel(int n_units, const int4* __restrict__ unit_atoms, const unsigned char* __restrict__ unit_type,
                        settle_const sc, float tol, int Npad, const float4* __restrict__ pos, float4* __restrict__ vel,
                        const long long* __restrict__ force, const float* __restrict__ invmass,
                        const fire_rep* __restrict__ state, int kick, double* __restrict__ partial /*[R][gridDim.x][3]*/)
{
    const int uidx = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    const fire_rep s = state[r];
    double sums[3] = { 0.0, 0.0, 0.0 };
    const int4 a4 = (uidx < n_units) ? unit_atoms[uidx] : make_int4(-1, -1, -1, -1);
    if (a4.x >= 0 && !(s.converged && kick)) {
        const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
        const int type = unit_type[uidx];
        const size_t o = (size_t)r * Npad, of = (size_t)r * 3 * Npad;
#define RUN(TY, NA) fire_finish_unit<TY, NA>(idx, sc, tol, Npad, pos + o, vel + o, force + of, invmass, s.dt, kick != 0, sums)
        if (type == UNIT_SETTLE) RUN(UNIT_SETTLE, 3);
        else if (type == UNIT_FREE) { if (a4.y < 0) RUN(UNIT_FREE, 1); else RUN(UNIT_FREE, 4); }
        else if (a4.z < 0) RUN(UNIT_SHAKE, 2);
        else if (a4.w < 0) RUN(UNIT_SHAKE, 3);
        else RUN(UNIT_SHAKE, 4);
#undef RUN
    }
    __shared__ double s_part[4][3];
    for (int q = 0; q < 3; ++q) {
        double v = sums[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_part[w][threadIdx.x];         // fixed order => reproducible
        partial[((size_t)r * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = v;
    }
}

// scalar FIRE logic of every replica (computed redundantly by every thread from the partial sums) + the per-atom update
__global__ __launch_bounds__(256)
void fire_update_kernel(int n_units, const int4* __restrict__ unit_atoms, int Npad, float4* __restrict__ pos, float4* __restrict__ vel,
                        long long* __restrict__ force, const float4* __restrict__ x0, const float4* __restrict__ v0,
                        const long long* __restrict__ f0, const double* __restrict__ potential, const double* __restrict__ partial,
                        int nblk, fire_consts c, const fire_rep* __restrict__ cur, fire_rep* __restrict__ nxt, int init)
{
    const int uidx = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    const fire_rep s = cur[r];
    double f2 = 0.0, v2 = 0.0, fv = 0.0;
    for (int b = 0; b < nblk; ++b) { f2 += partial[((size_t)r * nblk + b) * 3]; v2 += partial[((size_t)r * nblk + b) * 3 + 1]; fv += partial[((size_t)r * nblk + b) * 3 + 2]; }
    fire_rep n = s;
    if (init) {
        // before the first step: forces and energy at the start positions, convergence test of the first step (:2377-2386)
        n.E = potential[r]; n.f2 = f2;
        n.converged = (sqrt(f2) / (double)c.ndof <= (double)c.ftol) ? 1 : 0;
        if (uidx == 0) nxt[r] = n;
        return;
    }
    if (s.converged) { if (uidx == 0) nxt[r] = s; return; }
    const double E_new = potential[r];
    const bool restart = !(E_new - s.E < 0.0);                                   // :2423-2427, NaN-safe
    const float fmag = (float)sqrt(f2), vmag = (float)sqrt(v2);
    const float P = restart ? -1.f : (float)fv;
    const int4 a4 = (uidx < n_units) ? unit_atoms[uidx] : make_int4(-1, -1, -1, -1);
    if (a4.x >= 0) {
        const int idx[4] = { a4.x, a4.y, a4.z, a4.w };
        const size_t o = (size_t)r * Npad, of = (size_t)r * 3 * Npad;
        for (int k = 0; k < 4; ++k) {
            const int i = idx[k];
            if (i < 0) break;
            if (restart) {
                pos[o + i] = x0[o + i];
                force[of + i] = f0[of + i]; force[of + Npad + i] = f0[of + Npad + i]; force[of + 2 * Npad + i] = f0[of + 2 * Npad + i];
            }
            float4 w = restart ? v0[o + i] : vel[o + i];
            if (!restart && fmag > 0.f) {
                const float sf = s.alpha * vmag / fmag * (1.0f / 4294967296.0f);    // alpha |v| / |f| on the fixed-point force
                w.x = (1.f - s.alpha) * w.x + sf * (float)force[of + i];
                w.y = (1.f - s.alpha) * w.y + sf * (float)force[of + Npad + i];
                w.z = (1.f - s.alpha) * w.z + sf * (float)force[of + 2 * Npad + i];
            }
            if (P < 0.f) w = make_float4(0.f, 0.f, 0.f, 0.f);                     // :2455
            vel[o + i] = w;
        }
    }
    if (uidx == 0) {
        if (!restart) { n.E = E_new; n.f2 = f2; }
        if (s.dt <= c.dt_min) n.converged = 1;                                    // :2433-2437
        if (P > 0.f) {
            n.n_neg = s.n_neg + 1;
            if (n.n_neg > c.n_min) { n.dt = fminf(s.dt * c.f_inc, c.dt_max); n.alpha = s.alpha * c.f_alpha; }
        }
        if (P < 0.f) { n.n_neg = 0; n.dt = s.dt * c.f_dec; n.alpha = c.alpha0; }
        // convergence test at the top of the next step (:2377-2386), on the forces the next step starts from
        if (sqrt(n.f2) / (double)c.ndof <= (double)c.ftol) n.converged = 1;
        nxt[r] = n;
    }
}

int remd_minimize_impl(remd_ctx* h, double tolerance, int max_iterations, int32_t* converged_out, int32_t* n_iter_out)
{
    const unit_tables& ut = g_units[h];
    if (ut.n_units == 0) return remd_fail(h, -3, "no system set");
    const int R = h->R, Npad = h->Npad;
    const dim3 grid((ut.n_units + 255) / 256, R);
    const int nblk = (int)grid.x;
    float4 *x0 = nullptr, *v0 = nullptr; long long* f0 = nullptr; double* partial = nullptr; fire_rep* st = nullptr;
    REMD_CHECK(h, hipMalloc(&x0, sizeof(float4) * (size_t)R * Npad));
    REMD_CHECK(h, hipMalloc(&v0, sizeof(float4) * (size_t)R * Npad));
    REMD_CHECK(h, hipMalloc(&f0, sizeof(long long) * 3 * (size_t)R * Npad));
    REMD_CHECK(h, hipMalloc(&partial, sizeof(double) * 3 * (size_t)R * nblk));
    REMD_CHECK(h, hipMalloc(&st, sizeof(fire_rep) * 2 * (size_t)R));
    auto cleanup = [&]() { hipFree(x0); hipFree(v0); hipFree(f0); hipFree(partial); hipFree(st); };
    const float timestep = 0.001f;                               // 1 fs (integrators.py:2318)
    fire_consts c{};
    c.dt_max = 0.010f; c.f_inc = 1.1f; c.f_dec = 0.5f; c.alpha0 = 0.1f; c.f_alpha = 0.99f; c.n_min = 5;
    c.dt_min = 1.0e-5f * timestep; c.ftol = (float)tolerance; c.ndof = 3.0f * (float)h->N;
    std::vector<fire_rep> init(2 * (size_t)R);
    for (auto& s : init) { s.dt = timestep; s.alpha = c.alpha0; s.n_neg = 0; s.converged = 0; s.E = 0.0; s.f2 = 0.0; }
    REMD_CHECK(h, hipMemcpyAsync(st, init.data(), sizeof(fire_rep) * init.size(), hipMemcpyHostToDevice, h->stream));
    // "velocities should be set to zero before using this integrator" (integrators.py:2341)
    REMD_CHECK(h, hipMemsetAsync(h->d_vel, 0, sizeof(float4) * (size_t)R * Npad, h->stream));
    const float tol = (float)fmax(h->constraint_tol, REMD_CONSTRAINT_TOL_FLOOR);
    int cur = 0, rc = 0, it = 0;
    h->forces_valid = false; h->force_zeroed = false;
    if ((rc = remd_compute_forces(h, true))) { cleanup(); return rc; }
    hipLaunchKernelGGL(fire_finish_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, ut.d_type, ut.sc, tol, Npad, h->d_pos,
                       h->d_vel, h->d_force, h->d_invmass, st, 0, partial);
    hipLaunchKernelGGL(fire_update_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, Npad, h->d_pos, h->d_vel, h->d_force,
                       x0, v0, f0, h->d_potential, partial, nblk, c, st, st + R, 1);
    cur = 1;
    std::vector<fire_rep> host(R);
    const int limit = max_iterations > 0 ? max_iterations : 200000;
    bool all_done = false;
    while (it < limit && !all_done) {
        const int chunk = std::min(50, limit - it);              // the reference polls 'converged' every 50 steps (:1407-1409)
        for (int k = 0; k < chunk; ++k, ++it) {
            fire_rep* S = st + (size_t)cur * R;
            fire_rep* Nx = st + (size_t)(1 - cur) * R;
            hipLaunchKernelGGL(fire_move_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, ut.d_type, ut.d_dist, ut.sc, tol,
                               Npad, h->d_pos, h->d_vel, h->d_force, x0, v0, f0, h->d_invmass, S);
            h->forces_valid = false; h->force_zeroed = false;
            if ((rc = remd_compute_forces(h, true))) { cleanup(); return rc; }
            hipLaunchKernelGGL(fire_finish_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, ut.d_type, ut.sc, tol, Npad,
                               h->d_pos, h->d_vel, h->d_force, h->d_invmass, S, 1, partial);
            hipLaunchKernelGGL(fire_update_kernel, grid, dim3(256), 0, h->stream, ut.n_units, ut.d_atoms, Npad, h->d_pos, h->d_vel,
                               h->d_force, x0, v0, f0, h->d_potential, partial, nblk, c, S, Nx, 0);
            cur = 1 - cur;
        }
        hipMemcpyAsync(host.data(), st + (size_t)cur * R, sizeof(fire_rep) * R, hipMemcpyDeviceToHost, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess) { cleanup(); return remd_fail(h, -2, "minimize: device error"); }
        all_done = true;
        for (int r = 0; r < R; ++r) all_done = all_done && host[r].converged;
        if (max_iterations > 0) all_done = false;               // a fixed number of steps was asked for
    }
    hipMemcpyAsync(host.data(), st + (size_t)cur * R, sizeof(fire_rep) * R, hipMemcpyDeviceToHost, h->stream);
    hipStreamSynchronize(h->stream);
    if (converged_out) for (int r = 0; r < R; ++r) converged_out[r] = host[r].converged;
    if (n_iter_out) *n_iter_out = it;
    h->forces_valid = false; h->force_zeroed = false;
    cleanup();
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
