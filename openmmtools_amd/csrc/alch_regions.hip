// General alchemical regions (gfx950): the custom forces AbsoluteAlchemicalFactory._alchemically_modify_NonbondedForce builds
// (alchemy/alchemy.py:1539-2038) when the one-region / exact-PME fast path of forces.hip does not cover the request -- several
// regions with their own lambdas, pairs of interacting regions, soft-core electrostatics of the 'direct-space' / 'coulomb' PME
// treatments and of the reaction-field treatments, any soft-core exponents.  include/remd_hip.h (remd_set_alchemical_regions)
// states the force split and the expressions; the f64 restatement is oracle/alchemical_regions.py.
//
// Design: alchemical atoms are few (tens among thousands), so the custom forces are ONE launch over the (alchemical atom, atom)
// candidates per force evaluation -- n_alch x N distance tests, a bit per candidate for everything static (self, excluded pairs,
// the second side of an alchemical/alchemical pair, regions that do not interact) -- in front of the fork of a force evaluation.
// Forces add to the fixed-point accumulators of forces.hip (the alchemical atom's own force is summed over the wavefront first:
// lanes of one wavefront share it).  Energies: a second kernel with one workgroup per (state, replica) and a fixed order of
// summation; it serves the potential of a replica's own state (slot EP_REGION of the energy partials) and, over all K states, the
// region part of u_kl (the `alch` term of assemble_ukl_kernel).
#include "remd_internal.h"
#include "listed_terms.h"
#include <cmath>
#include <algorithm>
#include <cstring>

struct region_consts {
    float rc2, rs, inv_sw;                 // cutoff^2, sterics switch (rs < 0: none), 1 / (rc - rs)
    float rs_e, inv_sw_e;                  // electrostatics switch (rs_e < 0: none)
    float alpha_e, two_alpha_sqrtpi_e, krf, crf;
    int elec, n_cls, n_reg1, words, N, Npad, n_alch, n_exc;
    // exact PME treatment: the Ewald split of the handle (erfc to rcc), per region the self term, the net charge, the environment's
    // net charge and the coefficient of the neutralising background (E = coeff Q^2 / V)
    int consistent_exc;                    // the exceptions' electrostatics with the pairs' g (consistent_exceptions)
    int n_bonds, n_angles, n_torsions;     // alchemically softened bonded terms (lambda_bonds / lambda_angles / lambda_torsions of their region)
    int exact; float alpha_x, two_alpha_sqrtpi_x, rcc2;
    double self_x[4], q_x[4], q_env, plasma;
};

struct region_tables {
    region_consts c{};
    int n_regions = 0, K = 0;
    std::vector<double> softcore; std::vector<int> annihilate;
    struct cls { int kind, a, b, P; };     // kind 0: (environment, a); 1: (a, a); 2: (a, b) interacting; P = region whose soft-core constants apply
    std::vector<cls> classes;
    std::vector<double> ls, le;            // [K][n_regions]
    int* d_alch = nullptr;                 // [n_alch] atom
    float4* d_atom = nullptr;              // [N] q sqrt(k_e), sigma / 2, 2 sqrt(eps), region (bits)
    unsigned int* d_skip = nullptr;        // [n_alch][words] candidates that are never evaluated
    int* d_cls_of = nullptr;               // [(n + 1)^2] class of a pair of regions
    int* d_exc_atoms = nullptr; float4* d_exc_par = nullptr;      // exceptions: (k_e qq, sigma, 4 eps, class bits)
    float4* d_state_cls = nullptr;         // [K][n_cls][2]: (l^a, alpha (1 - l)^b, l^d, beta (1 - l)^e), (c, f, 0, 0)
    int* d_own = nullptr; std::vector<int> own_host;
    double* d_epart = nullptr; size_t epart_n = 0;          // [R][columns][n_alch] energy partials
    // softened bonded terms: atoms, parameters (the last entry of a term = its region as float bits), per state the regions' lambdas
    int* d_bonded_atoms = nullptr; float* d_bonded_par = nullptr;      // bonds [n][2] + angles [n][3] + torsions [n][4]; [n][3] + [n][3] + [n][4]
    float* d_state_bl = nullptr;           // [K][3][n_regions]
    std::vector<double> bl;                // host: [3][K][n_regions]
    // exact PME treatment (<= 4 charged regions: the slots of one float4 per replica, which the mesh kernels index by the atom's code)
    unsigned int* d_corr = nullptr;        // [n_alch][words] skipped candidates that still get the Ewald correction -qq erf(alpha r) / r
    float4* d_param_pme = nullptr;         // [Npad] the mesh kernels' charges: reference charges of the alchemical atoms, w = 8 + region slot
    float* d_rep_le = nullptr;             // [R][4] lambda_electrostatics of the regions at each replica's state (or the probe's)
    std::vector<float> rep_le_host;
    float* d_state_le = nullptr;           // [K][4]
    bool have_override = false; float le_override[4] = {1.f, 1.f, 1.f, 1.f};
    // a deep copy of the descriptor of remd_set_alchemical_regions: the blocks of a phased propagation (api.hip) are set up from it
    struct desc_store {
        remd_alch_regions_desc d{};
        std::vector<int32_t> region_of_atom, annihilate, interactions, exception_atoms, bond_atoms, bond_region, angle_atoms, angle_region, torsion_atoms, torsion_region;
        std::vector<double> softcore, charge, sigma, epsilon, exception_params, bond_params, angle_params, torsion_params;
        void assign(const remd_alch_regions_desc* s)
        {
            d = *s;
            auto cpi = [](std::vector<int32_t>& v, const int32_t*& p, size_t n) { if (p && n) { v.assign(p, p + n); p = v.data(); } else { v.clear(); p = nullptr; } };
            auto cpd = [](std::vector<double>& v, const double*& p, size_t n) { if (p && n) { v.assign(p, p + n); p = v.data(); } else { v.clear(); p = nullptr; } };
            const size_t N = (size_t)d.n_atoms, n = (size_t)d.n_regions;
            cpi(region_of_atom, d.region_of_atom, N); cpd(softcore, d.softcore, 8 * n); cpi(annihilate, d.annihilate, 2 * n);
            cpi(interactions, d.interactions, 2 * (size_t)d.n_interactions);
            cpd(charge, d.charge, N); cpd(sigma, d.sigma, N); cpd(epsilon, d.epsilon, N);
            cpi(exception_atoms, d.exception_atoms, 2 * (size_t)d.n_exceptions); cpd(exception_params, d.exception_params, 3 * (size_t)d.n_exceptions);
            cpi(bond_atoms, d.bond_atoms, 2 * (size_t)d.n_bonds); cpd(bond_params, d.bond_params, 2 * (size_t)d.n_bonds); cpi(bond_region, d.bond_region, (size_t)d.n_bonds);
            cpi(angle_atoms, d.angle_atoms, 3 * (size_t)d.n_angles); cpd(angle_params, d.angle_params, 2 * (size_t)d.n_angles); cpi(angle_region, d.angle_region, (size_t)d.n_angles);
            cpi(torsion_atoms, d.torsion_atoms, 4 * (size_t)d.n_torsions); cpd(torsion_params, d.torsion_params, 3 * (size_t)d.n_torsions); cpi(torsion_region, d.torsion_region, (size_t)d.n_torsions);
        }
    } store;
    bool have_bonded_lambdas = false;
};
static handle_table<region_tables> g_reg;

static inline float host_int_as_float(int v) { float f; memcpy(&f, &v, sizeof f); return f; }
template <typename T> static void dfree(T*& p) { if (p) { hipFree(p); p = nullptr; } }
template <typename T>
static int upload(remd_ctx* h, T*& dptr, const std::vector<T>& host)
{
    dfree(dptr);
    if (host.empty()) return 0;
    REMD_CHECK(h, hipMalloc(&dptr, sizeof(T) * host.size()));
    REMD_CHECK(h, hipMemcpy(dptr, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    return 0;
}

__device__ __forceinline__ void region_switch(float rs, float inv_sw, float r, float& U, float& dUdr)
{
    if (rs >= 0.f && r > rs) {
        const float x = (r - rs) * inv_sw;
        const float S = 1.f + x * x * x * (-10.f + x * (15.f - 6.f * x));
        const float dS = x * x * (-30.f + x * (60.f - 30.f * x)) * inv_sw;
        dUdr = S * dUdr + U * dS;
        U *= S;
    }
}

// soft-core Lennard-Jones (alchemy.py:1383-1388): U = l^a eps4 x (x - 1), x = (sigma / reff)^6 = (alpha (1 - l)^b + (r / sigma)^c)^(-6/c)
__device__ __forceinline__ void region_sterics(float4 A, float4 B, float sig, float eps4, float r, float inv_r, float& U, float& dUdr)
{
    const float t = powf(r / sig, B.x);
    const float base = A.y + t;
    const float x = powf(base, -6.f / B.x);
    U = A.x * eps4 * x * (x - 1.f);
    dUdr = A.x * eps4 * (2.f * x - 1.f) * (-6.f * x / base * t * inv_r);
}

// soft-core electrostatics (alchemy.py:1425-1430, 1434, 1505-1507, 1534-1536): U = l^d qq g(reff), reff = sigma (beta (1 - l)^e + (r / sigma)^f)^(1/f),
// g(x) = erfc(alpha x) / x + krf x^2 - crf
__device__ __forceinline__ void region_electrostatics(float4 A, float4 B, float alpha, float two_alpha_sqrtpi, float krf, float crf,
                                                      float sig, float qq, float r, float inv_r, float& U, float& dUdr)
{
    const float t = powf(r / sig, B.y);
    const float base = A.w + t;
    const float reff = sig * powf(base, 1.f / B.y);
    const float inv = 1.f / reff;
    float g, dg;
    if (alpha > 0.f) {
        const float ar = alpha * reff;
        const float ec = erfcf(ar);
        g = ec * inv; dg = -(ec * inv + two_alpha_sqrtpi * expf(-ar * ar)) * inv;
    } else { g = inv; dg = -inv * inv; }
    g += krf * reff * reff - crf; dg += 2.f * krf * reff;
    U = A.z * qq * g;
    dUdr = A.z * qq * dg * (reff / base * t * inv_r);
}

// lambda_electrostatics of the region an atom (or an exception) belongs to, exact PME treatment: slot = region - 1 of the replica's float4
__device__ __forceinline__ float region_le(const float* __restrict__ le4, int g) { return g > 0 ? le4[g - 1] : 1.f; }

// one candidate pair (alchemical atom a, atom j): energy and dU/dr / r; false: nothing to add
// exact PME treatment (le4 != NULL): the alchemical atom's direct-space Ewald term with the scaled charges, to the Coulomb range of the split
__device__ __forceinline__ bool region_pair(const region_consts& c, const float4* __restrict__ cls_tab, const int* __restrict__ cls_of,
                                            float4 pa, float4 pj, float3 d, float& U, float& fr, const float* __restrict__ le4 = nullptr)
{
    const float r2 = dotf(d, d);
    if (r2 >= (le4 ? fmaxf(c.rc2, c.rcc2) : c.rc2)) return false;
    const int ga = __float_as_int(pa.w), gj = __float_as_int(pj.w);
    const int k = cls_of[ga * c.n_reg1 + gj];
    const float inv_r = rsqrtf(r2), r = r2 * inv_r;
    const float sig = pa.y + pj.y, eps4 = pa.z * pj.z, qq = pa.x * pj.x;
    U = 0.f; float dU = 0.f;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A;
    if (k >= 0) { A = cls_tab[2 * k]; B = cls_tab[2 * k + 1]; }
    if (k >= 0 && eps4 != 0.f && r2 < c.rc2) {
        region_sterics(A, B, sig, eps4, r, inv_r, U, dU);
        region_switch(c.rs, c.inv_sw, r, U, dU);
    }
    if (k >= 0 && c.elec && qq != 0.f) {
        float Ue, dUe;
        region_electrostatics(A, B, c.alpha_e, c.two_alpha_sqrtpi_e, c.krf, c.crf, sig, qq, r, inv_r, Ue, dUe);
        region_switch(c.rs_e, c.inv_sw_e, r, Ue, dUe);
        U += Ue; dU += dUe;
    }
    if (le4 && qq != 0.f && r2 < c.rcc2) {
        const float L = region_le(le4, ga) * region_le(le4, gj) * qq;
        const float ar = c.alpha_x * r, ec = erfcf(ar);
        U += L * ec * inv_r;
        dU -= L * (ec * inv_r + c.two_alpha_sqrtpi_x * expf(-ar * ar)) * inv_r;
    }
    fr = dU * inv_r;
    return true;
}

// exact PME treatment: the reciprocal sum holds every pair; an excluded one (and a pair of two regions that do not interact,
// alchemy.py:1663-1672) is taken out again: -qq erf(alpha r) / r with the scaled charges, at any distance
__device__ __forceinline__ void region_ewald_correction(const region_consts& c, float4 pa, float4 pj, float3 d, const float* __restrict__ le4, float& U, float& fr)
{
    const float r2 = dotf(d, d);
    const float inv_r = rsqrtf(r2), r = r2 * inv_r;
    const float L = region_le(le4, __float_as_int(pa.w)) * region_le(le4, __float_as_int(pj.w)) * pa.x * pj.x;
    const float ar = c.alpha_x * r, ef = erff(ar);
    U = -L * ef * inv_r;
    fr = -L * (c.two_alpha_sqrtpi_x * expf(-ar * ar) * inv_r - ef * inv_r * inv_r) * inv_r;
}

// exception t: soft-core sterics without cutoff or switch, electrostatics l^d qq / reff (alchemy.py:1374-1380, 1434, 1456-1461);
// exact PME treatment: the charge product times the lambda of the region that owns the exception (its parameter offset, :1978-1982)
__device__ __forceinline__ void region_exception(const region_consts& c, const float4* __restrict__ cls_tab, float4 par, float3 d, float& U, float& fr,
                                                 const float* __restrict__ le4 = nullptr)
{
    const int w = __float_as_int(par.w), k = w & 0xffff;
    const float4 A = cls_tab[2 * k], B = cls_tab[2 * k + 1];
    const float r2 = dotf(d, d);
    const float inv_r = rsqrtf(r2), r = r2 * inv_r;
    U = 0.f; float dU = 0.f;
    if (par.z != 0.f) region_sterics(A, B, par.y, par.z, r, inv_r, U, dU);
    if (c.elec && par.x != 0.f) {
        float Ue, dUe;
        if (c.consistent_exc) region_electrostatics(A, B, c.alpha_e, c.two_alpha_sqrtpi_e, c.krf, c.crf, par.y, par.x, r, inv_r, Ue, dUe);
        else region_electrostatics(A, B, 0.f, 0.f, 0.f, 0.f, par.y, par.x, r, inv_r, Ue, dUe);
        U += Ue; dU += dUe;
    }
    if (le4 && par.x != 0.f) {
        const float L = region_le(le4, w >> 16) * par.x;
        U += L * inv_r; dU -= L * inv_r * inv_r;
    }
    fr = dU * inv_r;
}

__device__ __forceinline__ float3 region_min_image(float3 d, float Lx, float Ly, float Lz)
{
    if (Lx > 0.f) { d.x -= Lx * rintf(d.x / Lx); d.y -= Ly * rintf(d.y / Ly); d.z -= Lz * rintf(d.z / Lz); }      // (no box: a NoCutoff system)
    return d;
}

// ---- softened bonded terms: lambda x the reference's harmonic bond / harmonic angle / periodic torsion (alchemy.py:1180, 1261, 1341) -------------
struct region_bonded { const int* atoms; const float* par; const float* lam; int n_regions; };
// term t of the three classes laid end to end; returns lambda x energy, adds lambda x force when F is given
__device__ __forceinline__ float region_bonded_term(const region_consts& c, const region_bonded& B, int t, const float4* __restrict__ P,
                                                    long long* __restrict__ F)
{
    if (t < c.n_bonds) {
        const int i = B.atoms[2 * t], j = B.atoms[2 * t + 1];
        const float r0 = B.par[3 * t], k = B.par[3 * t + 1];
        const float lam = B.lam[__float_as_int(B.par[3 * t + 2]) - 1];
        const float3 d = sub3(ld3(P, j), ld3(P, i));
        const float len = sqrtf(dotf(d, d));
        if (F) {
            const float fs = lam * k * (len - r0) / len;
            add_force(F, c.Npad, i, fs * d.x, fs * d.y, fs * d.z);
            add_force(F, c.Npad, j, -fs * d.x, -fs * d.y, -fs * d.z);
        }
        return lam * 0.5f * k * (len - r0) * (len - r0);
    }
    t -= c.n_bonds;
    const int* A = B.atoms + 2 * c.n_bonds;
    const float* Q = B.par + 3 * c.n_bonds;
    if (t < c.n_angles) {
        const int a = A[3 * t], b = A[3 * t + 1], cc = A[3 * t + 2];
        const float th0 = Q[3 * t], k = Q[3 * t + 1];
        const float lam = B.lam[B.n_regions + __float_as_int(Q[3 * t + 2]) - 1];
        const float3 v0 = sub3(ld3(P, a), ld3(P, b)), v1 = sub3(ld3(P, cc), ld3(P, b));
        const float3 cp = crs3(v0, v1);
        const float rp = fmaxf(sqrtf(dotf(cp, cp)), 1e-6f);
        const float r20 = dotf(v0, v0), r21 = dotf(v1, v1);
        const float cosine = fminf(fmaxf(dotf(v0, v1) * rsqrtf(r20 * r21), -1.f), 1.f);
        const float dth = acosf(cosine) - th0;
        if (F) {
            const float dEdth = lam * k * dth;
            const float3 fa = scl3(crs3(v0, cp), -dEdth / (r20 * rp)), fc = scl3(crs3(cp, v1), -dEdth / (r21 * rp));
            add_force(F, c.Npad, a, fa.x, fa.y, fa.z);
            add_force(F, c.Npad, cc, fc.x, fc.y, fc.z);
            add_force(F, c.Npad, b, -(fa.x + fc.x), -(fa.y + fc.y), -(fa.z + fc.z));
        }
        return lam * 0.5f * k * dth * dth;
    }
    t -= c.n_angles;
    A += 3 * c.n_angles; Q += 3 * c.n_angles;
    const int a1 = A[4 * t], a2 = A[4 * t + 1], a3 = A[4 * t + 2], a4 = A[4 * t + 3];
    const float per = Q[4 * t], phase = Q[4 * t + 1], k = Q[4 * t + 2];
    const float lam = B.lam[2 * B.n_regions + __float_as_int(Q[4 * t + 3]) - 1];
    const float3 b1 = sub3(ld3(P, a2), ld3(P, a1)), b2 = sub3(ld3(P, a3), ld3(P, a2)), b3 = sub3(ld3(P, a4), ld3(P, a3));
    const float3 m = crs3(b1, b2), nn = crs3(b2, b3);
    const float m2 = fmaxf(dotf(m, m), 1e-12f), n2 = fmaxf(dotf(nn, nn), 1e-12f);
    const float lb2 = sqrtf(dotf(b2, b2));
    const float phi = atan2f(lb2 * dotf(b1, nn), dotf(m, nn));
    const float arg = per * phi - phase;
    if (F) {
        const float dEdphi = -lam * k * per * sinf(arg);
        const float3 g1 = scl3(m, -lb2 / m2), g4 = scl3(nn, lb2 / n2);
        const float s12 = dotf(b1, b2) / (lb2 * lb2), s32 = dotf(b3, b2) / (lb2 * lb2);
        const float3 g2 = add3(scl3(g1, -(1.f + s12)), scl3(g4, s32)), g3 = add3(scl3(g4, -(1.f + s32)), scl3(g1, s12));
        add_force(F, c.Npad, a1, -dEdphi * g1.x, -dEdphi * g1.y, -dEdphi * g1.z);
        add_force(F, c.Npad, a2, -dEdphi * g2.x, -dEdphi * g2.y, -dEdphi * g2.z);
        add_force(F, c.Npad, a3, -dEdphi * g3.x, -dEdphi * g3.y, -dEdphi * g3.z);
        add_force(F, c.Npad, a4, -dEdphi * g4.x, -dEdphi * g4.y, -dEdphi * g4.z);
    }
    return lam * k * (1.f + cosf(arg));
}

// forces: workgroup = (alchemical atom a, chunk of 1024 atoms j); the first workgroups of a replica also take the exceptions.
// One candidate in ten is inside the cutoff, and a wavefront that holds ONE of them pays the whole soft-core arithmetic (powf, erfc):
// the candidates are therefore tested first (distance, static bits) and the survivors compacted -- per wavefront by ballot + prefix count,
// so that the order (and with it every float sum) is fixed -- before 256 threads evaluate them densely: the launch's cost per force
// evaluation of 8 x CB7:B2 (34 alchemical atoms) fell from 21.6 to 5.6 us under 'direct-space', 13.3 to 8.0 under the exact PME treatment
// (tools/bench_configs.py 4d / 4r against 4, profiles/r06_40_general_regions_config4.txt).
#define REGION_CHUNK 1024
__global__ __launch_bounds__(256)
void region_forces_kernel(region_consts c, const int* __restrict__ alch, const float4* __restrict__ atom, const unsigned int* __restrict__ skip,
                          const int* __restrict__ cls_of, const float4* __restrict__ state_cls, const int* __restrict__ own,
                          const int* __restrict__ exc_atoms, const float4* __restrict__ exc_par,
                          const float4* __restrict__ pos, const float* __restrict__ box, long long* __restrict__ force,
                          const unsigned int* __restrict__ corr, const float* __restrict__ rep_le, region_bonded bonded)
{
    __shared__ int s_list[4][REGION_CHUNK / 4];            // per wavefront: its survivors, j | correction flag << 30
    __shared__ int s_count[4];
    const int r = blockIdx.y;
    const float* le4 = rep_le ? rep_le + 4 * r : nullptr;
    const int nchunk = (c.N + REGION_CHUNK - 1) / REGION_CHUNK;
    const int ia = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    const float4* P = pos + (size_t)r * c.Npad;
    long long* F = force + (size_t)r * 3 * c.Npad;
    const float4* cls_tab = state_cls + (size_t)own[r] * c.n_cls * 2;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const int a = alch[ia];
    const float4 pa = atom[a];
    const float3 xa = ld3(P, a);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float rmax2 = le4 ? fmaxf(c.rc2, c.rcc2) : c.rc2;
    const int j1 = min(c.N, (chunk + 1) * REGION_CHUNK);
    int n_mine = 0;
    for (int j0 = chunk * REGION_CHUNK + wave * 64; j0 < j1; j0 += 256) {
        const int j = j0 + lane;
        int keep = 0;
        if (j < j1) {
            const size_t word = (size_t)ia * c.words + (j >> 5);
            const bool skipped = (skip[word] >> (j & 31)) & 1u;
            const bool fix = skipped && le4 && ((corr[word] >> (j & 31)) & 1u);
            if (!skipped || fix) {
                const float3 d = region_min_image(sub3(ld3(P, j), xa), Lx, Ly, Lz);
                if (fix || dotf(d, d) < rmax2) keep = j | (fix ? (1 << 30) : 0) | (1 << 31);
            }
        }
        const unsigned long long m = __ballot(keep != 0);
        if (keep) s_list[wave][n_mine + __popcll(m & ((1ull << lane) - 1ull))] = keep;
        n_mine += __popcll(m);
    }
    if (lane == 0) s_count[wave] = n_mine;
    __syncthreads();
    float fx = 0.f, fy = 0.f, fz = 0.f;
    const int n0 = s_count[0], n1 = n0 + s_count[1], n2 = n1 + s_count[2], n3 = n2 + s_count[3];
    for (int t = threadIdx.x; t < n3; t += 256) {
        const int e = t < n0 ? s_list[0][t] : t < n1 ? s_list[1][t - n0] : t < n2 ? s_list[2][t - n1] : s_list[3][t - n2];
        const int j = e & 0x3fffffff;
        const float3 d = region_min_image(sub3(ld3(P, j), xa), Lx, Ly, Lz);
        const float4 pj = atom[j];
        float U, fr;
        if (e & (1 << 30)) {
            if (pa.x * pj.x == 0.f) continue;
            region_ewald_correction(c, pa, pj, d, le4, U, fr);
        } else if (!region_pair(c, cls_tab, cls_of, pa, pj, d, U, fr, le4)) continue;
        fx += fr * d.x; fy += fr * d.y; fz += fr * d.z;
        add_force(F, c.Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
    }
    for (int off = 32; off > 0; off >>= 1) { fx += __shfl_xor(fx, off); fy += __shfl_xor(fy, off); fz += __shfl_xor(fz, off); }
    if (lane == 0 && (fx != 0.f || fy != 0.f || fz != 0.f)) add_force(F, c.Npad, a, fx, fy, fz);
    // exceptions: spread over the workgroups of the replica
    for (int t = blockIdx.x * 256 + threadIdx.x; t < c.n_exc; t += gridDim.x * 256) {
        const int i = exc_atoms[2 * t], j = exc_atoms[2 * t + 1];
        float3 d = sub3(ld3(P, j), ld3(P, i));
        if (Lx > 0.f) d = region_min_image(d, Lx, Ly, Lz);
        float U, fr;
        region_exception(c, cls_tab, exc_par[t], d, U, fr, le4);
        add_force(F, c.Npad, i, fr * d.x, fr * d.y, fr * d.z);
        add_force(F, c.Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
    }
    // softened bonded terms, likewise
    bonded.lam += (size_t)own[r] * 3 * bonded.n_regions;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < c.n_bonds + c.n_angles + c.n_torsions; t += gridDim.x * 256)
        region_bonded_term(c, bonded, t, P, F);
}

// energies: workgroup = (alchemical atom, state column, replica) -- its candidates, and for the first atom's workgroup also the exceptions and
// the softened bonded terms -- writes one f64 partial; region_energy_sum_kernel adds a (column, replica)'s partials in a fixed order.  state =
// own[r] when `own` is given (the replica's potential), else the column.  (One workgroup per (column, replica) looping over all n_alch x N
// candidates took 0.9 - 1.4 ms on CB7:B2: eight workgroups on the chip in the own-state launches; rocprofv3, profiles/r06_42.)
__global__ __launch_bounds__(256)
void region_energy_kernel(region_consts c, const int* __restrict__ alch, const float4* __restrict__ atom, const unsigned int* __restrict__ skip,
                          const int* __restrict__ cls_of, const float4* __restrict__ state_cls, const int* __restrict__ own,
                          const int* __restrict__ exc_atoms, const float4* __restrict__ exc_par,
                          const float4* __restrict__ pos, const float* __restrict__ box, double* __restrict__ part /*[R][cols][n_alch]*/,
                          const unsigned int* __restrict__ corr, const float* __restrict__ rep_le, region_bonded bonded)
{
    // rep_le (exact PME treatment): the electrostatic terms at the replicas' lambdas ride along; NULL: sterics / custom electrostatics only
    __shared__ double s_part[4];
    const int ia = blockIdx.x, col = blockIdx.y, r = blockIdx.z;
    const float* le4 = rep_le ? rep_le + 4 * r : nullptr;
    const int state = own ? own[r] : col;
    const float4* P = pos + (size_t)r * c.Npad;
    const float4* cls_tab = state_cls + (size_t)state * c.n_cls * 2;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    double e = 0.0;
    const int a = alch[ia];
    const float4 pa = atom[a];
    const float3 xa = ld3(P, a);
    for (int j = threadIdx.x; j < c.N; j += 256) {
        const size_t word = (size_t)ia * c.words + (j >> 5);
        const bool skipped = (skip[word] >> (j & 31)) & 1u;
        if (skipped && (!le4 || !((corr[word] >> (j & 31)) & 1u))) continue;
        const float3 d = region_min_image(sub3(ld3(P, j), xa), Lx, Ly, Lz);
        float U, fr;
        if (skipped) {
            const float4 pj = atom[j];
            if (pa.x * pj.x != 0.f) { region_ewald_correction(c, pa, pj, d, le4, U, fr); e += (double)U; }
        } else if (region_pair(c, cls_tab, cls_of, pa, atom[j], d, U, fr, le4)) e += (double)U;
    }
    if (ia == 0) {
        for (int t = threadIdx.x; t < c.n_exc; t += 256) {
            float3 d = sub3(ld3(P, exc_atoms[2 * t + 1]), ld3(P, exc_atoms[2 * t]));
            if (Lx > 0.f) d = region_min_image(d, Lx, Ly, Lz);
            float U, fr;
            region_exception(c, cls_tab, exc_par[t], d, U, fr, le4);
            e += (double)U;
        }
        bonded.lam += (size_t)state * 3 * bonded.n_regions;
        for (int t = threadIdx.x; t < c.n_bonds + c.n_angles + c.n_torsions; t += 256) e += (double)region_bonded_term(c, bonded, t, P, nullptr);
    }
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) part[((size_t)r * gridDim.y + col) * c.n_alch + ia] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(64)
void region_energy_sum_kernel(region_consts c, int cols, const double* __restrict__ part, const float* __restrict__ rep_le, const float* __restrict__ box,
                              double* __restrict__ out, int out_stride, int out_offset)
{
    const int col = blockIdx.x, r = blockIdx.y;
    const double* p = part + ((size_t)r * cols + col) * c.n_alch;
    double e = 0.0;
    for (int t = threadIdx.x; t < c.n_alch; t += 64) e += p[t];
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if (threadIdx.x == 0) {
        if (rep_le) {
            // Ewald self terms of the regions' scaled charges and their part of the neutralising background (the environment's are with
            // the handle's constants: forces.hip const_energy_kernel)
            const float* le4 = rep_le + 4 * r;
            double Q = c.q_env;
            for (int g = 0; g < 4; ++g) { const double l = (double)le4[g]; e += l * l * c.self_x[g]; Q += l * c.q_x[g]; }
            const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
            if (V > 0) e += c.plasma * (Q * Q - c.q_env * c.q_env) / V;
        }
        out[(size_t)r * out_stride + out_offset + col] = e;
    }
}

// ---------------------------------------------------------------------------------------------------
void remd_regions_release(remd_ctx* h)
{
    region_tables* t = g_reg.find(h);
    if (t) {
        dfree(t->d_alch); dfree(t->d_atom); dfree(t->d_skip); dfree(t->d_cls_of); dfree(t->d_exc_atoms); dfree(t->d_exc_par);
        dfree(t->d_state_cls); dfree(t->d_own); dfree(t->d_epart);
        dfree(t->d_corr); dfree(t->d_param_pme); dfree(t->d_rep_le); dfree(t->d_state_le);
        dfree(t->d_bonded_atoms); dfree(t->d_bonded_par); dfree(t->d_state_bl);
        g_reg.erase(h);
    }
    h->n_regions = 0; h->regions_exact = 0;
}

int remd_set_alchemical_regions(remd_handle h, const remd_alch_regions_desc* d)
{
    if (!h) return remd_fail(h, -1, "remd_set_alchemical_regions: NULL handle");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    remd_regions_release(h);
    h->config_version++;
    h->forces_valid = false;
    if (!d || d->n_regions == 0) return 0;
    // (a block of a phased propagation reads the descriptor its parent keeps: api.hip phase_children)
    const remd_desc_store* store = h->parent ? h->parent->sysdesc : h->sysdesc;
    if (!h->has_system || !store || !store->valid) return remd_fail(h, -2, "remd_set_alchemical_regions: call remd_set_system first");
    const int N = h->N, n = d->n_regions;
    if (d->n_atoms != N) return remd_fail(h, -1, "remd_set_alchemical_regions: n_atoms differs from the system's");
    if (n < 0 || n > 64 || !d->region_of_atom || !d->softcore || !d->annihilate || !d->charge || !d->sigma || !d->epsilon ||
        d->n_interactions < 0 || (d->n_interactions > 0 && !d->interactions) || d->n_exceptions < 0 || (d->n_exceptions > 0 && (!d->exception_atoms || !d->exception_params)))
        return remd_fail(h, -1, "remd_set_alchemical_regions: bad arguments");
    if (h->nb_method == REMD_NB_NONE && !h->nocutoff) return remd_fail(h, -3, "alchemical regions need a NonbondedForce");
    if (store->d.n_alch != 0) return remd_fail(h, -3, "alchemical regions: the descriptor of remd_set_system must be the factory's NonbondedForce (n_alch = 0)");
    region_tables& t = g_reg[h];
    t.n_regions = n;
    t.softcore.assign(d->softcore, d->softcore + 8 * (size_t)n);
    t.annihilate.assign(d->annihilate, d->annihilate + 2 * (size_t)n);
    for (int g = 0; g < n; ++g) {
        const double* s = &t.softcore[8 * (size_t)g];
        if (!(s[4] > 0) || !(s[7] > 0)) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: softcore_c and softcore_f must be positive"); }
    }
    // classes of pairs of regions
    std::vector<int> cls_of((size_t)(n + 1) * (n + 1), -1);
    t.classes.clear();
    for (int g = 1; g <= n; ++g) {
        cls_of[g] = cls_of[(size_t)g * (n + 1)] = (int)t.classes.size(); t.classes.push_back({0, g, g, g});
        cls_of[(size_t)g * (n + 1) + g] = (int)t.classes.size(); t.classes.push_back({1, g, g, g});
    }
    const bool exact = d->exact_pme != 0;
    if (exact && (h->nb_method != REMD_NB_PME || n > 4)) {
        remd_regions_release(h);
        return remd_fail(h, exact && n > 4 ? -3 : -1, n > 4 ? "alchemical regions: more than four regions under the exact PME treatment are not supported" : "alchemical regions: exact_pme needs a PME system");
    }
    std::vector<char> interacting((size_t)(n + 1) * (n + 1), 0);
    for (int k = 0; k < d->n_interactions; ++k) {
        const int a = d->interactions[2 * k], b = d->interactions[2 * k + 1];
        if (a < 1 || b < 1 || a > n || b > n || a == b) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: bad pair of interacting regions"); }
        interacting[(size_t)a * (n + 1) + b] = interacting[(size_t)b * (n + 1) + a] = 1;
        if (exact) continue;                // exact PME: the pair sees each other's scaled charges; no sterics (tables zeroed, alchemy.py:1886-1911)
        if (cls_of[(size_t)a * (n + 1) + b] >= 0) continue;
        cls_of[(size_t)a * (n + 1) + b] = cls_of[(size_t)b * (n + 1) + a] = (int)t.classes.size(); t.classes.push_back({2, a, b, b});
    }
    // atoms
    std::vector<int> alch;
    std::vector<float4> atom(N);
    const double sqk = sqrt(REMD_ONE_4PI_EPS0);
    for (int i = 0; i < N; ++i) {
        const int g = d->region_of_atom[i];
        if (g < 0 || g > n) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: region index out of range"); }
        if (g > 0) alch.push_back(i);
        if (!(d->sigma[i] > 0) && g > 0) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: sigma must be positive (the factory sets 0 to 0.1 nm, alchemy.py:1638-1648)"); }
        atom[i] = make_float4((float)(d->charge[i] * sqk), (float)(0.5 * d->sigma[i]), (float)(2.0 * sqrt(d->epsilon[i])), host_int_as_float(g));
    }
    if (alch.empty()) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: no alchemical atom"); }
    const int na = (int)alch.size(), words = (N + 31) / 32;
    std::vector<int> ord(N, -1);
    for (int k = 0; k < na; ++k) ord[alch[k]] = k;
    std::vector<unsigned int> skip((size_t)na * words, 0u), corr(exact ? (size_t)na * words : 0, 0u);
    auto set_skip = [&](int ia, int j) { skip[(size_t)ia * words + (j >> 5)] |= 1u << (j & 31); };
    // exact PME treatment: an excluded pair's share of the reciprocal sum is taken out by the custom-forces launch (once per pair)
    auto set_corr = [&](int ia, int j) { if (exact && j != alch[ia] && !(d->region_of_atom[j] > 0 && j < alch[ia])) corr[(size_t)ia * words + (j >> 5)] |= 1u << (j & 31); };
    for (int ia = 0; ia < na; ++ia) {
        const int a = alch[ia], ga = d->region_of_atom[a];
        for (int j = 0; j < N; ++j) {
            const int gj = d->region_of_atom[j];
            // itself; an alchemical/alchemical pair is taken from its lower atom; regions that do not interact
            if (j == a || (gj > 0 && j < a)) set_skip(ia, j);
            else if (cls_of[(size_t)ga * (n + 1) + gj] < 0) {
                if (!exact) set_skip(ia, j);
                else if (!interacting[(size_t)ga * (n + 1) + gj]) { set_skip(ia, j); set_corr(ia, j); }       // alchemy.py:1663-1672
            }
            // environment atoms without sigma cannot enter the mixing rule: they have neither epsilon nor (in the factory's system) a way to interact
            else if (gj == 0 && !(d->sigma[j] > 0) && (d->epsilon[j] != 0.0 || (d->electrostatics && d->charge[j] != 0.0))) {
                remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: sigma must be positive (the factory sets 0 to 0.1 nm, alchemy.py:1638-1648)");
            }
        }
    }
    // every exception of the system is an exclusion of the custom forces (alchemy.py:1944-1947)
    const remd_system_desc& sd = store->d;
    for (int e = 0; e < sd.n_exceptions; ++e) {
        const int i = sd.exception_atoms[2 * e], j = sd.exception_atoms[2 * e + 1];
        if (ord[i] >= 0) { set_skip(ord[i], j); set_corr(ord[i], j); }
        if (ord[j] >= 0) { set_skip(ord[j], i); set_corr(ord[j], i); }
    }
    // the exceptions that became custom bonds
    std::vector<int> ea; std::vector<float4> ep;
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
        if (i < 0 || j < 0 || i >= N || j >= N || i == j) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: bad exception pair"); }
        const int gi = d->region_of_atom[i], gj = d->region_of_atom[j];
        const double qq = d->exception_params[3 * e], sg = d->exception_params[3 * e + 1], eps = d->exception_params[3 * e + 2];
        if ((gi == 0 && gj == 0) || (eps == 0.0 && (qq == 0.0 || !(d->electrostatics || exact)))) continue;
        if (!(sg > 0)) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: exception sigma must be positive"); }
        // an exception between atoms of two regions belongs to the FIRST region's (environment, region) bond force: the factory's loop meets
        // it there as "only one alchemical" and zeroes it before the second region's turn (alchemy.py:1972-1976, 1992-2006)
        const int cl = (gi > 0 && gj > 0 && gi != gj) ? cls_of[std::min(gi, gj)] : cls_of[(size_t)gi * (n + 1) + gj];
        const int owner = (gi > 0 && gj > 0) ? std::min(gi, gj) : std::max(gi, gj);       // whose lambda_electrostatics the charge product's offset carries
        ea.push_back(i); ea.push_back(j);
        ep.push_back(make_float4((float)(qq * REMD_ONE_4PI_EPS0), (float)sg, (float)(4.0 * eps), host_int_as_float(cl | (owner << 16))));
    }
    region_consts& c = t.c;
    // the NonbondedForce's cutoff and switch (h->cutoff is the range of the Ewald direct-space sum, which a rebalanced split stretches)
    // (NoCutoff: every pair, no switch -- the custom forces copy the NonbondedForce's method, alchemy.py:1793-1796)
    const double rcut = h->nocutoff ? 1e18 : sd.cutoff, rsw = h->nocutoff ? -1.0 : sd.switch_distance;
    c.rc2 = (float)std::min(rcut * rcut, 3e38);
    c.rs = rsw > 0 && rsw < rcut ? (float)rsw : -1.f;
    c.inv_sw = c.rs >= 0.f ? (float)(1.0 / (rcut - rsw)) : 0.f;
    c.elec = (d->electrostatics && !exact) ? 1 : 0;
    c.exact = exact ? 1 : 0;
    c.consistent_exc = (d->consistent_exceptions && c.elec) ? 1 : 0;
    for (int g = 0; g < 4; ++g) c.self_x[g] = c.q_x[g] = 0.0;
    c.q_env = 0.0; c.plasma = 0.0; c.alpha_x = c.two_alpha_sqrtpi_x = 0.f; c.rcc2 = 0.f;
    std::vector<float4> param_pme;
    if (exact) {
        // the handle's Ewald split (h->cutoff: the range of its direct-space sum)
        c.alpha_x = (float)h->ewald_alpha; c.two_alpha_sqrtpi_x = (float)(2.0 * h->ewald_alpha / sqrt(M_PI)); c.rcc2 = (float)(h->cutoff * h->cutoff);
        c.plasma = -REMD_ONE_4PI_EPS0 * M_PI / (2.0 * h->ewald_alpha * h->ewald_alpha);
        param_pme.assign(h->Npad, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int i = 0; i < N; ++i) {
            const int g = d->region_of_atom[i];
            if (g > 0) {
                c.self_x[g - 1] += -REMD_ONE_4PI_EPS0 * h->ewald_alpha / sqrt(M_PI) * d->charge[i] * d->charge[i];
                c.q_x[g - 1] += d->charge[i];
                param_pme[i] = make_float4((float)(d->charge[i] * sqk), 0.f, 0.f, (float)(8 + g - 1));
            } else {
                if (sd.charge[i] != d->charge[i]) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: the charges of the environment differ from the system's"); }
                c.q_env += sd.charge[i];
                param_pme[i] = make_float4((float)(sd.charge[i] * sqk), 0.f, 0.f, 0.f);
            }
            if (g > 0 && sd.charge[i] != 0.0) { remd_regions_release(h); return remd_fail(h, -1, "alchemical regions (exact PME): the system's descriptor must carry the alchemical atoms without charge"); }
        }
    }
    c.rs_e = c.elec && d->elec_switch_distance >= 0 && d->elec_switch_distance < rcut ? (float)d->elec_switch_distance : -1.f;
    c.inv_sw_e = c.rs_e >= 0.f ? (float)(1.0 / (rcut - d->elec_switch_distance)) : 0.f;
    c.alpha_e = (float)d->elec_alpha; c.two_alpha_sqrtpi_e = (float)(2.0 * d->elec_alpha / sqrt(M_PI));
    c.krf = (float)d->elec_krf; c.crf = (float)d->elec_crf;
    c.n_cls = (int)t.classes.size(); c.n_reg1 = n + 1; c.words = words; c.N = N; c.Npad = h->Npad; c.n_alch = na; c.n_exc = (int)ea.size() / 2;
    int rc;
    if ((rc = upload(h, t.d_alch, alch)) || (rc = upload(h, t.d_atom, atom)) || (rc = upload(h, t.d_skip, skip)) || (rc = upload(h, t.d_cls_of, cls_of)) ||
        (rc = upload(h, t.d_exc_atoms, ea)) || (rc = upload(h, t.d_exc_par, ep)) || (rc = upload(h, t.d_corr, corr)) ||
        (rc = upload(h, t.d_param_pme, param_pme))) { remd_regions_release(h); return rc; }
    {   // softened bonded terms
        if (d->n_bonds < 0 || d->n_angles < 0 || d->n_torsions < 0 || (d->n_bonds > 0 && (!d->bond_atoms || !d->bond_params || !d->bond_region)) ||
            (d->n_angles > 0 && (!d->angle_atoms || !d->angle_params || !d->angle_region)) || (d->n_torsions > 0 && (!d->torsion_atoms || !d->torsion_params || !d->torsion_region))) {
            remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: bad softened bonded terms");
        }
        std::vector<int> ba; std::vector<float> bp;
        auto take = [&](int cnt, int width, int npar, const int32_t* atoms, const double* par, const int32_t* reg) -> bool {
            for (int k = 0; k < cnt; ++k) {
                for (int q = 0; q < width; ++q) { const int a = atoms[width * k + q]; if (a < 0 || a >= N) return false; ba.push_back(a); }
                for (int q = 0; q < npar; ++q) bp.push_back((float)par[npar * k + q]);
                if (reg[k] < 1 || reg[k] > n) return false;
                bp.push_back(host_int_as_float(reg[k]));
            }
            return true;
        };
        if (!take(d->n_bonds, 2, 2, d->bond_atoms, d->bond_params, d->bond_region) || !take(d->n_angles, 3, 2, d->angle_atoms, d->angle_params, d->angle_region) ||
            !take(d->n_torsions, 4, 3, d->torsion_atoms, d->torsion_params, d->torsion_region)) {
            remd_regions_release(h); return remd_fail(h, -1, "alchemical regions: softened bonded term with a bad atom or region");
        }
        c.n_bonds = d->n_bonds; c.n_angles = d->n_angles; c.n_torsions = d->n_torsions;
        if ((rc = upload(h, t.d_bonded_atoms, ba)) || (rc = upload(h, t.d_bonded_par, bp))) { remd_regions_release(h); return rc; }
    }
    h->n_regions = n; h->regions_exact = exact ? 1 : 0;
    if (!h->parent) t.store.assign(d);
    return 0;
}

int remd_set_region_lambdas(remd_handle h, int K, int n_regions, const double* ls, const double* le)
{
    if (!h) return remd_fail(h, -1, "remd_set_region_lambdas: NULL handle");
    region_tables* tp = g_reg.find(h);
    if (!tp || h->n_regions == 0) return remd_fail(h, -2, "remd_set_region_lambdas: no alchemical regions on this handle");
    region_tables& t = *tp;
    if (K != h->K || n_regions != t.n_regions || !ls || !le) return remd_fail(h, -1, "remd_set_region_lambdas: K / n_regions differ from remd_set_states / remd_set_alchemical_regions");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    const int n = t.n_regions, C = t.c.n_cls;
    for (size_t k = 0; k < (size_t)K * n; ++k)
        if (!(ls[k] >= 0.0 && ls[k] <= 1.0 && le[k] >= 0.0 && le[k] <= 1.0)) return remd_fail(h, -1, "remd_set_region_lambdas: lambdas must be in [0, 1]");
    t.ls.assign(ls, ls + (size_t)K * n); t.le.assign(le, le + (size_t)K * n); t.K = K;
    std::vector<float4> tab((size_t)K * C * 2);
    for (int k = 0; k < K; ++k)
        for (int q = 0; q < C; ++q) {
            const region_tables::cls& cl = t.classes[q];
            const double* s = &t.softcore[8 * (size_t)(cl.P - 1)];
            double l_s, l_e;
            if (cl.kind == 0) { l_s = ls[(size_t)k * n + cl.a - 1]; l_e = le[(size_t)k * n + cl.a - 1]; }
            else if (cl.kind == 1) {
                l_s = t.annihilate[2 * (cl.a - 1)] ? ls[(size_t)k * n + cl.a - 1] : 1.0;
                l_e = t.annihilate[2 * (cl.a - 1) + 1] ? le[(size_t)k * n + cl.a - 1] : 1.0;
            } else { l_s = ls[(size_t)k * n + cl.a - 1] * ls[(size_t)k * n + cl.b - 1]; l_e = le[(size_t)k * n + cl.a - 1] * le[(size_t)k * n + cl.b - 1]; }
            tab[((size_t)k * C + q) * 2] = make_float4((float)pow(l_s, s[2]), (float)(s[0] * pow(1.0 - l_s, s[3])),
                                                       (float)pow(l_e, s[5]), (float)(s[1] * pow(1.0 - l_e, s[6])));
            tab[((size_t)k * C + q) * 2 + 1] = make_float4((float)s[4], (float)s[7], 0.f, 0.f);
        }
    int rc = upload(h, t.d_state_cls, tab);
    if (rc) return rc;
    std::vector<float> sle((size_t)K * 4, 1.f);
    for (int k = 0; k < K; ++k) for (int g = 0; g < n && g < 4; ++g) sle[4 * (size_t)k + g] = (float)le[(size_t)k * n + g];
    if ((rc = upload(h, t.d_state_le, sle))) return rc;
    t.rep_le_host.clear();
    t.bl.assign(3 * (size_t)K * n, 1.0);                      // until remd_set_region_bonded_lambdas says otherwise
    {
        std::vector<float> one(3 * (size_t)K * n, 1.f);
        if ((rc = upload(h, t.d_state_bl, one))) return rc;
    }
    h->config_version++;
    h->forces_valid = false;
    return 0;
}

int remd_set_region_bonded_lambdas(remd_handle h, int K, int n_regions, const double* lb, const double* la, const double* lt)
{
    if (!h) return remd_fail(h, -1, "remd_set_region_bonded_lambdas: NULL handle");
    region_tables* tp = g_reg.find(h);
    if (!tp || h->n_regions == 0) return remd_fail(h, -2, "remd_set_region_bonded_lambdas: no alchemical regions on this handle");
    region_tables& t = *tp;
    if (K != h->K || K != t.K || n_regions != t.n_regions) return remd_fail(h, -1, "remd_set_region_bonded_lambdas: call remd_set_region_lambdas first (same K, n_regions)");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    const int n = t.n_regions;
    const double* src[3] = {lb, la, lt};
    std::vector<float> tab(3 * (size_t)K * n, 1.f);           // [K][3][n]
    for (int q = 0; q < 3; ++q) for (int k = 0; k < K; ++k) for (int g = 0; g < n; ++g) {
        const double v = src[q] ? src[q][(size_t)k * n + g] : 1.0;
        if (!(v >= 0.0 && v <= 1.0)) return remd_fail(h, -1, "remd_set_region_bonded_lambdas: lambdas must be in [0, 1]");
        tab[((size_t)k * 3 + q) * n + g] = (float)v;
    }
    int rc = upload(h, t.d_state_bl, tab);
    if (rc) return rc;
    for (int q = 0; q < 3; ++q) for (int k = 0; k < K; ++k) for (int g = 0; g < n; ++g) t.bl[((size_t)q * K + k) * n + g] = (double)tab[((size_t)k * 3 + q) * n + g];
    t.have_bonded_lambdas = true;
    h->config_version++;
    h->forces_valid = false;
    return 0;
}

// the regions of `parent` on one of its blocks (api.hip phase_children): descriptor, the states' lambdas, the bonded lambdas
int remd_regions_clone(remd_ctx* parent, remd_ctx* child)
{
    region_tables* tp = g_reg.find(parent);
    if (!tp || parent->n_regions == 0) return 0;
    region_tables& t = *tp;
    if (t.store.d.n_regions != t.n_regions || t.K != parent->K || t.ls.size() != (size_t)t.K * t.n_regions)
        return remd_fail(parent, -2, "phases: the handle's alchemical regions are not complete (remd_set_region_lambdas)");
    int rc = remd_set_alchemical_regions(child, &t.store.d);
    if (!rc) rc = remd_set_region_lambdas(child, t.K, t.n_regions, t.ls.data(), t.le.data());
    if (!rc && t.have_bonded_lambdas && t.bl.size() == 3 * (size_t)t.K * t.n_regions) {
        const size_t m = (size_t)t.K * t.n_regions;
        rc = remd_set_region_bonded_lambdas(child, t.K, t.n_regions, t.bl.data(), t.bl.data() + m, t.bl.data() + 2 * m);
    }
    if (rc) return remd_fail(parent, rc, std::string("phases: ") + child->err);
    return 0;
}

static int region_own_states(remd_ctx* h, region_tables& t)
{
    std::vector<int> own(h->R);
    for (int r = 0; r < h->R; ++r) own[r] = h->labels.empty() ? 0 : (int)h->labels[h->r_begin + r];
    for (int r = 0; r < h->R; ++r) if (own[r] < 0 || own[r] >= t.K) return remd_fail(h, -1, "alchemical regions: a replica's state has no region lambdas");
    if (t.c.exact) {
        // lambda_electrostatics of the regions per replica: its own state's, or the probe's (u_kl passes)
        std::vector<float> rl(4 * (size_t)h->R, 1.f);
        for (int r = 0; r < h->R; ++r) for (int g = 0; g < t.n_regions; ++g)
            rl[4 * (size_t)r + g] = t.have_override ? t.le_override[g] : (float)t.le[(size_t)own[r] * t.n_regions + g];
        if (rl != t.rep_le_host || !t.d_rep_le) {
            if (t.rep_le_host.size() != rl.size() || !t.d_rep_le) { dfree(t.d_rep_le); REMD_CHECK(h, hipMalloc(&t.d_rep_le, sizeof(float) * rl.size())); }
            REMD_CHECK(h, hipMemcpyAsync(t.d_rep_le, rl.data(), sizeof(float) * rl.size(), hipMemcpyHostToDevice, h->stream));
            REMD_CHECK(h, hipStreamSynchronize(h->stream));
            t.rep_le_host = rl;
        }
    }
    if (own == t.own_host && t.d_own) return 0;          // uploaded only when the labels changed
    if (t.own_host.size() != own.size()) { dfree(t.d_own); REMD_CHECK(h, hipMalloc(&t.d_own, sizeof(int) * own.size())); }
    REMD_CHECK(h, hipMemcpyAsync(t.d_own, own.data(), sizeof(int) * own.size(), hipMemcpyHostToDevice, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    t.own_host = own;
    return 0;
}

static int region_energy_launch(remd_ctx* h, region_tables& t, int cols, const int* d_own, const float* d_rep_le, double* d_out, int out_stride, int out_offset)
{
    const size_t need = (size_t)h->R * cols * t.c.n_alch;
    if (need > t.epart_n) { dfree(t.d_epart); REMD_CHECK(h, hipMalloc(&t.d_epart, sizeof(double) * need)); t.epart_n = need; }
    const region_bonded rb{t.d_bonded_atoms, t.d_bonded_par, t.d_state_bl, t.n_regions};
    hipLaunchKernelGGL(region_energy_kernel, dim3(t.c.n_alch, cols, h->R), dim3(256), 0, h->stream, t.c, t.d_alch, t.d_atom, t.d_skip, t.d_cls_of,
                       t.d_state_cls, d_own, t.d_exc_atoms, t.d_exc_par, h->d_pos, h->d_box, t.d_epart, t.d_corr, d_rep_le, rb);
    hipLaunchKernelGGL(region_energy_sum_kernel, dim3(cols, h->R), dim3(64), 0, h->stream, t.c, cols, t.d_epart, d_rep_le, h->d_box, d_out, out_stride, out_offset);
    return 0;
}

// at the head of a force evaluation, on the stream everything else of the evaluation is ordered behind
int remd_regions_forces(remd_ctx* h, bool with_energy, int ep_slot)
{
    region_tables* tp = g_reg.find(h);
    if (!tp) return 0;
    region_tables& t = *tp;
    if (!t.d_state_cls || t.K != h->K) return remd_fail(h, -2, "alchemical regions: remd_set_region_lambdas has not been called for these states");
    int rc = region_own_states(h, t);
    if (rc) return rc;
    remd_prof_scope ps(h, "alch_regions");
    const int nchunk = (h->N + REGION_CHUNK - 1) / REGION_CHUNK;
    hipLaunchKernelGGL(region_forces_kernel, dim3(t.c.n_alch * nchunk, h->R), dim3(256), 0, h->stream, t.c, t.d_alch, t.d_atom, t.d_skip, t.d_cls_of,
                       t.d_state_cls, t.d_own, t.d_exc_atoms, t.d_exc_par, h->d_pos, h->d_box, h->d_force, t.d_corr, t.c.exact ? t.d_rep_le : (const float*)nullptr, region_bonded{t.d_bonded_atoms, t.d_bonded_par, t.d_state_bl, t.n_regions});
    if (with_energy) {
        if ((rc = region_energy_launch(h, t, 1, t.d_own, t.c.exact ? t.d_rep_le : (const float*)nullptr, h->d_epart, h->n_epart, ep_slot))) return rc;
    }
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

// the region terms at every state's lambdas: out[r][k]; *d_own = the replicas' own states on the device
int remd_regions_ukl(remd_ctx* h, double* d_out, const int** d_own)
{
    region_tables* tp = g_reg.find(h);
    if (!tp) return remd_fail(h, -2, "alchemical regions: none on this handle");
    region_tables& t = *tp;
    if (!t.d_state_cls || t.K != h->K) return remd_fail(h, -2, "alchemical regions: remd_set_region_lambdas has not been called for these states");
    int rc = region_own_states(h, t);
    if (rc) return rc;
    remd_prof_scope ps(h, "alch_ukl");
    // (exact PME treatment: the sterics only; the Coulomb part is the quadratic form of forces.hip)
    if ((rc = region_energy_launch(h, t, h->K, (const int*)nullptr, (const float*)nullptr, d_out, h->K, 0))) return rc;
    REMD_CHECK(h, hipGetLastError());
    *d_own = t.d_own;
    return 0;
}

// ---- exact PME treatment: what forces.hip / pme.hip ask ---------------------------------------------------------------------------------
// the mesh kernels' charge table and per-replica lambdas (NULL: not in this mode); pme.hip indexes the float4 of a replica by the atom's code
int remd_regions_pme_tables(remd_ctx* h, const float4** param, const float** rep_le)
{
    *param = nullptr; *rep_le = nullptr;
    if (!h->regions_exact) return 0;
    region_tables* t = g_reg.find(h);
    if (!t) return 0;
    *param = t->d_param_pme; *rep_le = t->d_rep_le;
    return 1;
}
// every replica at these lambda_electrostatics (u_kl probes); NULL: back to the replicas' own states.  n = regions.
int remd_regions_le_override(remd_ctx* h, const float* le, int* n, const float** d_state_le)
{
    region_tables* t = g_reg.find(h);
    if (!t || !t->c.exact) return remd_fail(h, -2, "alchemical regions: not under the exact PME treatment");
    t->have_override = le != nullptr;
    for (int g = 0; g < 4; ++g) t->le_override[g] = (le && g < t->n_regions) ? le[g] : 1.f;
    if (n) *n = t->n_regions;
    if (d_state_le) *d_state_le = t->d_state_le;
    return 0;
}

// lambda_electrostatics of region `g` (0-based) at state k, as remd_set_region_lambdas gave it (gbsa.hip: the alchemical particles' factor)
int remd_regions_state_le(remd_ctx* h, int k, int g, double* le)
{
    region_tables* t = g_reg.find(h);
    if (!t || t->K <= 0 || k < 0 || k >= t->K || g < 0 || g >= t->n_regions) return -1;
    *le = t->le[(size_t)k * t->n_regions + g];
    return 0;
}
