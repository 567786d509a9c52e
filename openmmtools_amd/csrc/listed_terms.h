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
#ifdef EXP_NOATOM          // knock-out probe (tools/build_variant.sh -DEXP_NOATOM): the arithmetic stays, the atomics go (results wrong on purpose)
    if (fx != 1.2345e33f) return;
#endif
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


// ---- the terms, one function per class: the forces on every atom of term t (slot order = the order of its atom indices).
// Shared by the two ways the launch is organised below, so that both add exactly the same fixed-point numbers.
// (named members, no arrays: a slot picked at run time from an array would put the array into scratch memory)
struct listed_out { int a0, a1, a2, a3; float f0x, f0y, f0z, f1x, f1y, f1z, f2x, f2y, f2z, f3x, f3y, f3z; };
#define LISTED_SET(o, k, X, Y, Z) do { (o).f##k##x = (X); (o).f##k##y = (Y); (o).f##k##z = (Z); } while (0)
__device__ __forceinline__ void listed_bond(const listed_tables& T, int t, const float4* __restrict__ P, listed_out& o)
{
    const int i = T.bond_atoms[2 * t], j = T.bond_atoms[2 * t + 1];
    const float r0 = T.bond_params[2 * t], k = T.bond_params[2 * t + 1];
    const float3 d = sub3(ld3(P, j), ld3(P, i));
    const float len = sqrtf(dotf(d, d));
    const float fs = k * (len - r0) / len;
    o.a0 = i; o.a1 = j;
    LISTED_SET(o, 0, fs * d.x, fs * d.y, fs * d.z);
    LISTED_SET(o, 1, -fs * d.x, -fs * d.y, -fs * d.z);
}
__device__ __forceinline__ void listed_angle(const listed_tables& T, int t, const float4* __restrict__ P, listed_out& o)
{
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
    o.a0 = a; o.a1 = b; o.a2 = c;
    LISTED_SET(o, 0, fa.x, fa.y, fa.z); LISTED_SET(o, 2, fc.x, fc.y, fc.z);
    LISTED_SET(o, 1, -(fa.x + fc.x), -(fa.y + fc.y), -(fa.z + fc.z));
}
__device__ __forceinline__ void listed_torsion(const listed_tables& T, int t, const float4* __restrict__ P, listed_out& o)
{
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
    o.a0 = a1; o.a1 = a2; o.a2 = a3; o.a3 = a4;
    LISTED_SET(o, 0, -dEdphi * g1.x, -dEdphi * g1.y, -dEdphi * g1.z);
    LISTED_SET(o, 1, -dEdphi * g2.x, -dEdphi * g2.y, -dEdphi * g2.z);
    LISTED_SET(o, 2, -dEdphi * g3.x, -dEdphi * g3.y, -dEdphi * g3.z);
    LISTED_SET(o, 3, -dEdphi * g4.x, -dEdphi * g4.y, -dEdphi * g4.z);
}
__device__ __forceinline__ void listed_exception(const listed_tables& T, int t, const float4* __restrict__ P, const float* __restrict__ box, int r,
                                                 listed_out& o)
{
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
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
    o.a0 = i; o.a1 = j;
    LISTED_SET(o, 0, fr * d.x, fr * d.y, fr * d.z);
    LISTED_SET(o, 1, -fr * d.x, -fr * d.y, -fr * d.z);
}
__device__ __forceinline__ void listed_exclusion(const listed_tables& T, int t, const float4* __restrict__ P, const float* __restrict__ box, int r,
                                                 listed_out& o)
{
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
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
    o.a0 = i; o.a1 = j;
    LISTED_SET(o, 0, fr * d.x, fr * d.y, fr * d.z);
    LISTED_SET(o, 1, -fr * d.x, -fr * d.y, -fr * d.z);
}

// All short "listed" terms of one force evaluation in a single launch (force-only path): harmonic bonds, angles,
// periodic torsions, non-zero exceptions and the Ewald exclusion correction.
//
// Two organisations of the same arithmetic.  One TERM per thread (T.aterm NULL): every atom of the term gets its three integer
// atomics -- 6 to 12 scattered 64-bit atomics per term, and scattered atomics are what that launch spends its time on (MI355X:
// ~45 G per second chip-wide, ONE PER 64-BYTE LINE an instruction touches; DHFR x 16: 180 us with them, 6 without,
// profiles/r06_23_atomics_knock_out.txt).  Round 6, the default: one (term, atom of the term) ENTRY per thread, the entries in the
// order of the atoms they act on (remd_ctx::d_aterm: class << 29 | slot << 27 | term, forces.hip: build_atom_terms): every thread
// evaluates the whole term and adds only its slot's force, so the 64 lanes of an atomic instruction fall on a handful of
// neighbouring atoms -- a few lines instead of 64 -- at the price of evaluating a torsion four times.  Integer sums do not
// depend on the order: the totals are the same fixed-point numbers.
__device__ __forceinline__
void listed_forces_body(const listed_tables& T, int Npad, const float4* __restrict__ pos, const float* __restrict__ box,
                        long long* __restrict__ force, int t, int r)
{
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    int cls = -1, slot = -1;                              // slot -1: this thread adds every slot of the term
    if (T.aterm) {
        // (no early return on this path: every lane takes part in the reduction over the lanes of one atom below)
        if (t < T.n_aterm) {
            const unsigned int ent = T.aterm[t];
            cls = (int)(ent >> 29); slot = (int)((ent >> 27) & 3u); t = (int)(ent & 0x7ffffffu);
            const int n_cls = cls == 0 ? T.n_bonds : cls == 1 ? T.n_angles : cls == 2 ? T.n_torsions : cls == 3 ? T.n_exc : T.n_excl;
            if (n_cls == 0) cls = -1;                     // the class is not part of this evaluation (force groups)
        }
    } else {
        if (t < T.n_bonds) cls = 0;
        else if ((t -= T.n_bonds) < T.n_angles) cls = 1;
        else if ((t -= T.n_angles) < T.n_torsions) cls = 2;
        else if ((t -= T.n_torsions) < T.n_exc) cls = 3;
        else if ((t -= T.n_exc) < T.n_excl) cls = 4;
        else return;
    }
    listed_out o;
    o.a0 = o.a1 = o.a2 = o.a3 = 0;
    LISTED_SET(o, 0, 0.f, 0.f, 0.f); LISTED_SET(o, 1, 0.f, 0.f, 0.f); LISTED_SET(o, 2, 0.f, 0.f, 0.f); LISTED_SET(o, 3, 0.f, 0.f, 0.f);
    int n = 0;
    if (cls == 0) { listed_bond(T, t, P, o); n = 2; }
    else if (cls == 1) { listed_angle(T, t, P, o); n = 3; }
    else if (cls == 2) { listed_torsion(T, t, P, o); n = 4; }
    else if (cls == 3) { listed_exception(T, t, P, box, r, o); n = 2; }
    else if (cls == 4) { listed_exclusion(T, t, P, box, r, o); n = 2; }
    if (T.aterm) {
        // (the members are pinned to registers first: left alone the compiler turns the selects below into selects of the members'
        //  ADDRESSES and keeps the struct in scratch memory)
        int a0 = o.a0, a1 = o.a1, a2 = o.a2, a3 = o.a3;
        float f0x = o.f0x, f0y = o.f0y, f0z = o.f0z, f1x = o.f1x, f1y = o.f1y, f1z = o.f1z;
        float f2x = o.f2x, f2y = o.f2y, f2z = o.f2z, f3x = o.f3x, f3y = o.f3y, f3z = o.f3z;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("" : "+v"(f0x), "+v"(f0y), "+v"(f0z), "+v"(f1x), "+v"(f1y), "+v"(f1z));
        asm volatile("" : "+v"(f2x), "+v"(f2y), "+v"(f2z), "+v"(f3x), "+v"(f3y), "+v"(f3z));
        const int lane = (int)(threadIdx.x & 63u);
        int a = slot == 0 ? a0 : slot == 1 ? a1 : slot == 2 ? a2 : a3;
        const float fx = slot == 0 ? f0x : slot == 1 ? f1x : slot == 2 ? f2x : f3x;
        const float fy = slot == 0 ? f0y : slot == 1 ? f1y : slot == 2 ? f2y : f3y;
        const float fz = slot == 0 ? f0z : slot == 1 ? f1z : slot == 2 ? f2z : f3z;
        if (cls < 0) a = -1 - lane;                       // idle lane: a segment of its own, nothing to add
        // The entries are in atom order, so the lanes of one atom are neighbours: their fixed-point forces are summed along the
        // run (a segmented scan over the wavefront) and the run's last lane issues the atom's one atomic triple.  Atomics of
        // several lanes on ONE address are serialised at the L2 (measured: the entries with an atomic each are slower than the
        // term-per-thread launch), while the runs' last lanes hit distinct, neighbouring atoms.
        unsigned long long vx = cls < 0 ? 0ull : remd_f2fix(fx), vy = cls < 0 ? 0ull : remd_f2fix(fy), vz = cls < 0 ? 0ull : remd_f2fix(fz);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int au = __shfl_up(a, d);
            const unsigned long long ux = __shfl_up(vx, d), uy = __shfl_up(vy, d), uz = __shfl_up(vz, d);
            if (lane >= d && au == a) { vx += ux; vy += uy; vz += uz; }
        }
        const int an = __shfl_down(a, 1);
        if (cls >= 0 && (lane == 63 || an != a)) {
            unsigned long long* U = reinterpret_cast<unsigned long long*>(F);
            atomicAdd(&U[a], vx); atomicAdd(&U[Npad + a], vy); atomicAdd(&U[2 * Npad + a], vz);
        }
        return;
    }
    add_force(F, Npad, o.a0, o.f0x, o.f0y, o.f0z);
    add_force(F, Npad, o.a1, o.f1x, o.f1y, o.f1z);
    if (n > 2) add_force(F, Npad, o.a2, o.f2x, o.f2y, o.f2z);
    if (n > 3) add_force(F, Npad, o.a3, o.f3x, o.f3y, o.f3z);
}
