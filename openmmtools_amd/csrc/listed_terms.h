// Listed terms of a force-only evaluation (harmonic bonds, angles, periodic torsions, non-zero 1-4 exceptions, Ewald exclusion
// correction), one term per thread.  Shared by forces.hip (listed_forces_kernel) and pme.hip (round 4: in the mode in which the
// direct-space stream is the critical one the terms ride as extra workgroups of the spreading launch on the mesh stream --
// one dependent 13 us launch less).  Functional forms: OpenMM HarmonicBondForce / HarmonicAngleForce / PeriodicTorsionForce /
// NonbondedForce exceptions as the reference's test systems build them (testsystems.py:3504-3517); f64 restatement:
// oracle/forcefield.py.
#pragma once
#include "remd_internal.h"

__device__ __forceinline__ void add_force(long long* __restrict__ F, int Npad, int i, float fx, float fy, float fz)
{
    unsigned long long* U = reinterpret_cast<unsigned long long*>(F);
    atomicAdd(&U[i],            remd_f2fix(fx));
    atomicAdd(&U[Npad + i],     remd_f2fix(fy));
    atomicAdd(&U[2 * Npad + i], remd_f2fix(fz));
}


__device__ __forceinline__ float3 ld3(const float4* P, int i) { const float4 p = P[i]; return make_float3(p.x, p.y, p.z); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 scl3(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 add3(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float dotf(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 crs3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}



// Lennard-Jones part of an exception between an alchemical and a non-alchemical atom: the factory moves it to a
// lambda_sterics-controlled CustomBondForce with the soft-core expression of the pair interactions, no cutoff, no switch
// (alchemy.py:1836-1851, 1985-1998, expression :1374-1380 with softcore_c = 6).  lam_a = lambda^a, sc = alpha (1 - lambda)^b.
// Returns (U, dU/dr).
__device__ __forceinline__ float2 softcore_exception(float lam_a, float sc, float sig, float eps, float r2, float inv_r)
{
    const float is2 = 1.f / (sig * sig);
    const float t = r2 * r2 * r2 * is2 * is2 * is2;
    const float x = 1.f / (sc + t);
    const float e4 = 4.f * eps * lam_a;
    return make_float2(e4 * x * (x - 1.f), e4 * (2.f * x - 1.f) * (-x * x * 6.f * t * inv_r));
}


// All short "listed" terms of one force evaluation in a single launch (force-only path): harmonic bonds, angles,
// periodic torsions, non-zero exceptions and the Ewald exclusion correction.  One term per thread.
__device__ __forceinline__
void listed_forces_body(const listed_tables& T, int Npad, const float4* __restrict__ pos, const float* __restrict__ box,
                        long long* __restrict__ force, int t, int r)
{
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    if (t < T.n_bonds) {
        const int i = T.bond_atoms[2 * t], j = T.bond_atoms[2 * t + 1];
        const float r0 = T.bond_params[2 * t], k = T.bond_params[2 * t + 1];
        const float3 d = sub3(ld3(P, j), ld3(P, i));
        const float len = sqrtf(dotf(d, d));
        const float fs = k * (len - r0) / len;
        add_force(F, Npad, i, fs * d.x, fs * d.y, fs * d.z);
        add_force(F, Npad, j, -fs * d.x, -fs * d.y, -fs * d.z);
        return;
    }
    t -= T.n_bonds;
    if (t < T.n_angles) {
        const int a = T.angle_atoms[3 * t], b = T.angle_atoms[3 * t + 1], c = T.angle_atoms[3 * t + 2];
        const float th0 = T.angle_params[2 * t], k = T.angle_params[2 * t + 1];
        const float3 v0 = sub3(ld3(P, a), ld3(P, b)), v1 = sub3(ld3(P, c), ld3(P, b));
        const float3 cp = crs3(v0, v1);
        const float rp = fmaxf(sqrtf(dotf(cp, cp)), 1e-6f);
        const float r20 = dotf(v0, v0), r21 = dotf(v1, v1);
        const float cosine = fminf(fmaxf(dotf(v0, v1) * rsqrtf(r20 * r21), -1.f), 1.f);
        const float dEdth = k * (acosf(cosine) - th0);
        const float3 fa = scl3(crs3(v0, cp), -dEdth / (r20 * rp));
        const float3 fc = scl3(crs3(cp, v1), -dEdth / (r21 * rp));
        add_force(F, Npad, a, fa.x, fa.y, fa.z);
        add_force(F, Npad, c, fc.x, fc.y, fc.z);
        add_force(F, Npad, b, -(fa.x + fc.x), -(fa.y + fc.y), -(fa.z + fc.z));
        return;
    }
    t -= T.n_angles;
    if (t < T.n_torsions) {
        const int a1 = T.torsion_atoms[4 * t], a2 = T.torsion_atoms[4 * t + 1], a3 = T.torsion_atoms[4 * t + 2], a4 = T.torsion_atoms[4 * t + 3];
        const float per = T.torsion_params[3 * t], phase = T.torsion_params[3 * t + 1], k = T.torsion_params[3 * t + 2];
        const float3 p1 = ld3(P, a1), p2 = ld3(P, a2), p3 = ld3(P, a3), p4 = ld3(P, a4);
        const float3 b1 = sub3(p2, p1), b2 = sub3(p3, p2), b3 = sub3(p4, p3);
        const float3 m = crs3(b1, b2), nn = crs3(b2, b3);
        const float m2 = fmaxf(dotf(m, m), 1e-12f), n2 = fmaxf(dotf(nn, nn), 1e-12f);
        const float lb2 = sqrtf(dotf(b2, b2));
        const float phi = atan2f(lb2 * dotf(b1, nn), dotf(m, nn));
        const float dEdphi = -k * per * sinf(per * phi - phase);
        const float3 g1 = scl3(m, -lb2 / m2);
        const float3 g4 = scl3(nn, lb2 / n2);
        const float s12 = dotf(b1, b2) / (lb2 * lb2), s32 = dotf(b3, b2) / (lb2 * lb2);
        const float3 g2 = add3(scl3(g1, -(1.f + s12)), scl3(g4, s32));
        const float3 g3 = add3(scl3(g4, -(1.f + s32)), scl3(g1, s12));
        add_force(F, Npad, a1, -dEdphi * g1.x, -dEdphi * g1.y, -dEdphi * g1.z);
        add_force(F, Npad, a2, -dEdphi * g2.x, -dEdphi * g2.y, -dEdphi * g2.z);
        add_force(F, Npad, a3, -dEdphi * g3.x, -dEdphi * g3.y, -dEdphi * g3.z);
        add_force(F, Npad, a4, -dEdphi * g4.x, -dEdphi * g4.y, -dEdphi * g4.z);
        return;
    }
    t -= T.n_torsions;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    if (t < T.n_exc) {
        const int i = T.exc_atoms[2 * t], j = T.exc_atoms[2 * t + 1];
        float qq = T.exc_params[3 * t];
        const float sig = T.exc_params[3 * t + 1], eps = T.exc_params[3 * t + 2];
        if (T.rep_lam && T.exc_alch[t] > 0) qq *= T.rep_lam[4 * r + 2];      // alchemy.py:1964-1966 exception offset
        float3 d = sub3(ld3(P, j), ld3(P, i));
        if (Lx > 0.f) { d.x -= Lx * rintf(d.x / Lx); d.y -= Ly * rintf(d.y / Ly); d.z -= Lz * rintf(d.z / Lz); }
        const float r2 = dotf(d, d);
        const float inv_r = rsqrtf(r2);
        float dUlj;
        if (T.rep_lam && T.exc_alch[t] == 1 && eps != 0.f) {
            dUlj = softcore_exception(T.rep_lam[4 * r], T.rep_lam[4 * r + 1], sig, eps, r2, inv_r).y;
        } else {
            const float s2 = sig * sig * inv_r * inv_r, s6 = s2 * s2 * s2;
            dUlj = 4.f * eps * s6 * (6.f - 12.f * s6) * inv_r;
        }
        const float fr = (dUlj - qq * inv_r * inv_r) * inv_r;
        add_force(F, Npad, i, fr * d.x, fr * d.y, fr * d.z);
        add_force(F, Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
        return;
    }
    t -= T.n_exc;
    if (t < T.n_excl) {
        const int i = T.excl_atoms[2 * t], j = T.excl_atoms[2 * t + 1];
        float qq = T.excl_qq[t];
        if (T.rep_lam) { const float le = T.rep_lam[4 * r + 2]; const int na = T.excl_alch[t]; qq *= (na == 2) ? le * le : (na == 1) ? le : 1.f; }
        float3 d = sub3(ld3(P, j), ld3(P, i));
        d.x -= Lx * rintf(d.x / Lx); d.y -= Ly * rintf(d.y / Ly); d.z -= Lz * rintf(d.z / Lz);
        const float r2 = dotf(d, d);
        const float inv_r = rsqrtf(r2);
        const float ar = T.alpha * r2 * inv_r;
        const float erf_ar = erff(ar);
        const float fr = -qq * (T.two_alpha_sqrtpi * __expf(-ar * ar) * inv_r - erf_ar * inv_r * inv_r) * inv_r;
        add_force(F, Npad, i, fr * d.x, fr * d.y, fr * d.z);
        add_force(F, Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
    }
}

