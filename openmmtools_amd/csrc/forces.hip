// Force / potential-energy kernels (gfx950): harmonic external force, bonded terms, nonbonded
// direct space (LJ + switching function, reaction-field or Ewald-direct Coulomb, soft-core
// alchemical sterics), exceptions, Ewald exclusion correction.
//
// Forces accumulate in 64-bit fixed point (integer atomics => bit-reproducible sums, scale 2^32);
// energies are written as per-block f64 partials and summed in a fixed order.
//
// Functional forms restated from the reference's system builders / OpenMM semantics:
//   testsystems.HarmonicOscillator      testsystems.py:779-786   U = K/2 ((x-x0)^2+y^2+z^2) + U0
//   NonbondedForce (LJ fluid)           testsystems.py:1978-2000 CutoffPeriodic, switch, dispersion correction
//   Amber explicit solvent              testsystems.py:3504-3517 PME, HBonds, rigid water, switch
//   alchemical soft-core sterics        alchemy/alchemy.py:1383-1388 (softcore_c = 6 closed form)
// The f64 restatement used for parity is oracle/md_oracle.py.
//
// Nonbonded design: atoms are kept in a spatial (Morton) order of whole molecules; 8-atom clusters of that order are
// paired through per-tile union lists with every cluster pair listed once (nonbonded_sci_kernel, Newton's third law);
// a list that outgrows its capacity, or a system without sortable molecules, runs on the 64-atom tile kernel
// (nonbonded_kernel: lane = atom i, wave-uniform j stream, each pair from both sides, exclusions as a bit window).
// The Monte Carlo barostat lives in barostat.hip, the pair arithmetic in pair_math.h.
#include "remd_internal.h"
#include "rng.h"
#include <cmath>
#include <algorithm>
#include <set>
#include <cstdlib>
#include <cstdio>

#define EP_EXT      0
#define EP_BOND     1
#define EP_ANGLE    2
#define EP_TORSION  3
#define EP_EXCEPT   4
#define EP_EXCLCORR 5
#define EP_PME      6
#define EP_CONST    7
#define EP_NB0      8      // first nonbonded block slot

#include "pair_math.h"
#include "listed_terms.h"

#define MAX_EXCL_WORDS 8

struct nb_tables {
    nb_params p{};
    int method = NB_LJ_ONLY;
    bool has_alch = false;
    float4* d_param = nullptr;            // [Npad] (q*sqrt(k_e), sigma/2, 2*sqrt(eps), alchemical flag)
    unsigned long long* d_mask = nullptr; // [Npad][excl_words]
    int n_exc = 0; int* d_exc_atoms = nullptr; float* d_exc_params = nullptr;     // nonzero exceptions
    int* d_exc_alch = nullptr; int* d_excl_alch = nullptr;   // number of alchemical atoms in each pair (0, 1, 2)
    double lam_e_override = -1.0;         // >= 0: evaluate every replica at this lambda_electrostatics (u_kl probes)
    double self_nn = 0, self_aa = 0, q_n = 0, q_a = 0;   // Ewald self / net-charge pieces (non-alchemical, alchemical)
    double* d_probe = nullptr; int probe_R = 0;            // [3][R] potentials at lambda_e = 0, 1/2, 1
    int n_excl = 0; int* d_excl_atoms = nullptr; float* d_excl_qq = nullptr;      // all excluded pairs (Ewald correction)
    float* d_rep_lam = nullptr;           // [R][4] per replica: lambda_s^a, alpha (1-lambda_s)^b, lambda_e, pad
    int rep_lam_R = 0;
    std::vector<double> state_lam_a, state_sc;   // per state, for the alchemical u_kl kernel
    double* d_state_lam = nullptr;        // [K][2]
    double* d_alch_ukl = nullptr;         // [R][K]
    int* d_own = nullptr;                 // [R] the replicas' own states (column of d_alch_ukl that d_potential already holds)
    int alch_R = 0, alch_K = 0;
    double disp_coeff = 0.0;              // E_disp = disp_coeff / V
    double self_energy = 0.0;             // Ewald self term, kJ/mol
    double net_charge_term = 0.0;         // neutralising-plasma coefficient: E = coeff / V
    std::vector<double> charge;           // original charges
    std::vector<char> is_alch;
    // spatial ordering: exclusion/constraint-connected groups (molecules) stay contiguous, groups are ordered
    // along a Morton curve of their cells every resort_interval force evaluations
    int n_groups = 0; int* d_grp_first = nullptr; int* d_grp_size = nullptr;
    int* d_order = nullptr;               // [R][Npad] sorted slot -> atom (-1: padding)
    float4* d_spos = nullptr;             // [R][Npad] positions in sorted order (refreshed every evaluation)
    float4* d_sposi = nullptr;            // [R][Npad] the same as 32-bit box fractions (bit patterns in x, y, z) + charge in w
    float4* d_lj_sposi = nullptr;         // [R][NLpad] the LJ sub-system's
    float4* d_sparam = nullptr;           // [R][Npad]
    unsigned long long* d_smask = nullptr;// [R][Npad][excl_words]
    float4* d_tile_c = nullptr; float4* d_tile_h = nullptr;   // [R][ntile] bounding-box centre / half extent
    float4* d_partial = nullptr; size_t partial_n = 0;      // [R][n_jsplit][Npad] nonbonded force partials
    // 8-atom cluster pair lists (sorted slot space): lane = (i atom, j atom) of an 8 x 8 cluster pair
    float4* d_cl_c = nullptr; float4* d_cl_h = nullptr;     // [R][ncl] cluster bounding boxes
    int cl_cap = 0; bool clusters = true;
    // LJ-active sub-system (atoms with eps != 0; 1/3 of a TIP3P box): own sorted order, clusters and pair list, so
    // that the main cluster kernel is Coulomb-only
    bool lj_split = false; int NL = 0, NLpad = 0, lj_words = 1, lj_cap = 0;
    int* d_lj_ord = nullptr;              // [Npad] atom -> LJ ordinal (-1: no LJ)
    unsigned long long* d_lj_mask = nullptr;   // [NLpad][lj_words] exclusion window in LJ-ordinal space
    int* d_lj_order = nullptr; float4* d_lj_spos = nullptr; float4* d_lj_sparam = nullptr; unsigned long long* d_lj_smask = nullptr;
    float4* d_lj_tile_c = nullptr; float4* d_lj_tile_h = nullptr; float4* d_lj_cl_c = nullptr; float4* d_lj_cl_h = nullptr;
    // Newton's-third-law path: per-tile union lists (jc | imask << 16) and near-diagonal exclusion words
    int sci_split = 12;     // list slices per tile (workgroups of 4 wavefronts); set per system in remd_build_nonbonded
    unsigned int* d_sci_list = nullptr; int* d_sci_count = nullptr; unsigned long long* d_excl = nullptr; int excl_W = 0;
    long long* d_sforce = nullptr; long long* d_lj_sforce = nullptr;   // [R][3][Npad] / [R][3][NLpad] forces in sorted slot space
    unsigned int* d_lj_sci_list = nullptr; int* d_lj_sci_count = nullptr; unsigned long long* d_lj_excl = nullptr; int lj_excl_W = 0;
    int sort_R = 0; int evals_since_sort = 1 << 30; int resort_interval = 40; bool sorting = true;
    std::vector<float> rep_lam_host;      // what d_rep_lam holds
    unsigned int* d_queue = nullptr;      // item queues of the resident-workgroup pair kernel: [0] Coulomb, [1] LJ, [2] workgroups done
    // How many workgroups of the pair kernel stay resident next to the mesh kernels (0 = one per item) is a balance between
    // the two streams that depends on the system (24 x alanine dipeptide: 2 per CU is 4 % faster than one per item, 8 x
    // host-guest: 7 % slower), so it is measured: during the first long remd_run_steps the candidates take turns over
    // segments of TUNE_SEG real MD steps timed with events; results are bit-identical under every choice (fixed-point sums).
    struct tune_seg { int cand; hipEvent_t a, b; };       // cand: index into g_tune_cands
    int nb_prio = 0;                      // current choice: 1 = the pair kernel runs at raised wave priority and the mesh kernels do not
    std::vector<tune_seg> tune_segs; int tune_state = 0 /* 0 measuring, 1 awaiting resolve, 2 done */, tune_left = 0, tune_next = 0;
    int nb_grid = 0;                      // current choice (workgroups; 0 = one per item)
    float sort_cell = 0.45f;              // Morton cell edge (nm) of the molecule sort (sort_hbits = 0)
    int sort_hbits = 0;                   // > 0: 2^sort_hbits cells per box edge along the Hilbert curve (round 6, the default)
    // Ewald direct-space force table of the force-only pair kernels (coulomb_table.h); REMD_NB_TABLE=0: Abramowitz & Stegun erfc
    float4* d_ctab = nullptr; bool use_table = false;
    unsigned int* d_pair_done = nullptr; unsigned int pair_done_target = 0;      // remd_fold_args: the scatter launch's done counter
    int* d_tile_of_rank = nullptr;        // [R][ntile] the main system's tiles by descending list length, refreshed with the molecule order
    int* d_sort_scratch = nullptr; size_t sort_scratch_n = 0;                    // sort_groups_large_kernel (more than 8191 molecules)
};
static handle_table<nb_tables> g_nb;

// cross-stream dependencies without the command processor (see remd_ctx::d_sync): one lane polls a flag in device memory
__global__ void remd_spin_wait_kernel(const unsigned int* flag, unsigned int seq, unsigned int* spin_out)
{
    if (threadIdx.x == 0) {
        long long n = 0;
        while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
            // a fault is raised already (this propagation is run again by the host whatever happens from here): no second wait -- else every
            // step behind a poll that ran out waits for its own time-out
            if ((n & 255) == 255 && __hip_atomic_load(spin_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(4);
            if (++n > (1ll << 25)) { atomicCAS(spin_out, 0u, 1u); break; }       // seconds: something upstream died; say so instead of hanging
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
void remd_launch_join_wait(remd_ctx* h)      // a deferred join nobody consumed: wait for it now
{
    if (h->fold_pending) {
        // no chain took it (remd_fold_args): the scatter is the direct-space stream's last launch -- an event is enough here
        // (end of a propagation, once per call)
        hipEventRecord(h->ev_join, h->stream2);
        hipStreamWaitEvent(h->stream, h->ev_join, 0);
        h->fold_pending = false;
    }
    if (!h->join_deferred) return;
    hipLaunchKernelGGL(remd_spin_wait_kernel, dim3(1), dim3(64), 0, h->stream, h->d_sync + 1, h->join_deferred, h->d_sync + 2);
    h->join_deferred = 0;
}
__global__ void remd_signal_kernel(unsigned int* flag, unsigned int seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double wave_sum(double e)
{
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    return e;
}

// block of 256 threads: deterministic sum, result valid in thread 0
__device__ __forceinline__ double block_sum_256(double e, double* s_part)
{
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = e;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

template <bool ENERGY>
__global__ __launch_bounds__(64)
void ext_force_kernel(int n_ext, const int* __restrict__ ext_atoms, float K, float x0, double U0,
                      int Npad, const float4* __restrict__ pos, long long* __restrict__ force,
                      double* __restrict__ epart, int n_epart)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n_ext; t += 64) {
        const int i = ext_atoms[t];
        const float4 p = pos[(size_t)r * Npad + i];
        const float dx = p.x - x0;
        add_force(force + (size_t)r * 3 * Npad, Npad, i, -K * dx, -K * p.y, -K * p.z);
        if (ENERGY) e += 0.5 * (double)K * ((double)dx * dx + (double)p.y * p.y + (double)p.z * p.z) + U0;
    }
    if (ENERGY) {
        e = wave_sum(e);
        if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_EXT] = e;
    }
}

// ---- bonded terms (float3 helpers: listed_terms.h) ------------------------------------------------
template <bool ENERGY>
__global__ __launch_bounds__(256)
void bond_kernel(int n, const int* __restrict__ atoms, const float* __restrict__ params, int Npad,
                 const float4* __restrict__ pos, long long* __restrict__ force, double* __restrict__ epart, int n_epart)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int i = atoms[2 * t], j = atoms[2 * t + 1];
        const float r0 = params[2 * t], k = params[2 * t + 1];
        const float3 d = sub3(ld3(P, j), ld3(P, i));
        const float len = sqrtf(dotf(d, d));
        const float dl = len - r0;
        const float fs = k * dl / len;               // F_i = +fs * d, F_j = -fs * d
        add_force(F, Npad, i, fs * d.x, fs * d.y, fs * d.z);
        add_force(F, Npad, j, -fs * d.x, -fs * d.y, -fs * d.z);
        if (ENERGY) e += 0.5 * (double)k * (double)dl * (double)dl;
    }
    if (ENERGY) { e = block_sum_256(e, s_part); if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_BOND] = e; }
}

template <bool ENERGY>
__global__ __launch_bounds__(256)
void angle_kernel(int n, const int* __restrict__ atoms, const float* __restrict__ params, int Npad,
                  const float4* __restrict__ pos, long long* __restrict__ force, double* __restrict__ epart, int n_epart)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int a = atoms[3 * t], b = atoms[3 * t + 1], c = atoms[3 * t + 2];
        const float th0 = params[2 * t], k = params[2 * t + 1];
        const float3 v0 = sub3(ld3(P, a), ld3(P, b));     // b -> a
        const float3 v1 = sub3(ld3(P, c), ld3(P, b));     // b -> c
        const float3 cp = crs3(v0, v1);
        const float rp = fmaxf(sqrtf(dotf(cp, cp)), 1e-6f);
        const float r20 = dotf(v0, v0), r21 = dotf(v1, v1);
        const float dt = dotf(v0, v1);
        const float cosine = fminf(fmaxf(dt * rsqrtf(r20 * r21), -1.f), 1.f);
        const float theta = acosf(cosine);
        const float dth = theta - th0;
        const float dEdth = k * dth;
        // dtheta/dx_a = (v0 x cp) / (|v0|^2 |cp|),  dtheta/dx_c = (cp x v1) / (|v1|^2 |cp|)
        const float3 fa = scl3(crs3(v0, cp), -dEdth / (r20 * rp));
        const float3 fc = scl3(crs3(cp, v1), -dEdth / (r21 * rp));
        add_force(F, Npad, a, fa.x, fa.y, fa.z);
        add_force(F, Npad, c, fc.x, fc.y, fc.z);
        add_force(F, Npad, b, -(fa.x + fc.x), -(fa.y + fc.y), -(fa.z + fc.z));
        if (ENERGY) e += 0.5 * (double)k * (double)dth * (double)dth;
    }
    if (ENERGY) { e = block_sum_256(e, s_part); if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_ANGLE] = e; }
}

template <bool ENERGY>
__global__ __launch_bounds__(256)
void torsion_kernel(int n, const int* __restrict__ atoms, const float* __restrict__ params, int Npad,
                    const float4* __restrict__ pos, long long* __restrict__ force, double* __restrict__ epart, int n_epart)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int a1 = atoms[4 * t], a2 = atoms[4 * t + 1], a3 = atoms[4 * t + 2], a4 = atoms[4 * t + 3];
        const float per = params[3 * t], phase = params[3 * t + 1], k = params[3 * t + 2];
        const float3 p1 = ld3(P, a1), p2 = ld3(P, a2), p3 = ld3(P, a3), p4 = ld3(P, a4);
        const float3 b1 = sub3(p2, p1), b2 = sub3(p3, p2), b3 = sub3(p4, p3);
        const float3 m = crs3(b1, b2), nn = crs3(b2, b3);
        const float m2 = fmaxf(dotf(m, m), 1e-12f), n2 = fmaxf(dotf(nn, nn), 1e-12f);
        const float lb2 = sqrtf(dotf(b2, b2));
        // IUPAC dihedral: phi = atan2(|b2| b1.(b2 x b3), (b1 x b2).(b2 x b3))
        const float phi = atan2f(lb2 * dotf(b1, nn), dotf(m, nn));
        const float arg = per * phi - phase;
        const float dEdphi = -k * per * sinf(arg);
        // gradient of phi (Blondel & Karplus 1996)
        const float3 g1 = scl3(m, -lb2 / m2);
        const float3 g4 = scl3(nn, lb2 / n2);
        const float s12 = dotf(b1, b2) / (lb2 * lb2), s32 = dotf(b3, b2) / (lb2 * lb2);
        const float3 g2 = add3(scl3(g1, -(1.f + s12)), scl3(g4, s32));     // = -g1 - s12 g1 + s32 g4
        const float3 g3 = add3(scl3(g4, -(1.f + s32)), scl3(g1, s12));
        add_force(F, Npad, a1, -dEdphi * g1.x, -dEdphi * g1.y, -dEdphi * g1.z);
        add_force(F, Npad, a2, -dEdphi * g2.x, -dEdphi * g2.y, -dEdphi * g2.z);
        add_force(F, Npad, a3, -dEdphi * g3.x, -dEdphi * g3.y, -dEdphi * g3.z);
        add_force(F, Npad, a4, -dEdphi * g4.x, -dEdphi * g4.y, -dEdphi * g4.z);
        if (ENERGY) e += (double)k * (1.0 + (double)cosf(arg));
    }
    if (ENERGY) { e = block_sum_256(e, s_part); if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_TORSION] = e; }
}

__global__ __launch_bounds__(256)
void listed_forces_kernel(listed_tables T, int Npad, const float4* __restrict__ pos, const float* __restrict__ box,
                          long long* __restrict__ force)
{
    listed_forces_body(T, Npad, pos, box, force, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}


// ---- spatial ordering ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned morton3(unsigned x, unsigned y, unsigned z)
{
    unsigned m = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) m |= ((x >> b) & 1u) << (3 * b) | ((y >> b) & 1u) << (3 * b + 1) | ((z >> b) & 1u) << (3 * b + 2);
    return m;
}

// index of cell (x, y, z) of a 2^bits cube along the 3-D Hilbert curve (Skilling's transpose algorithm): consecutive indices are
// face neighbours, so the molecules of a run of the order -- the 8-atom clusters and 64-atom tiles of the pair kernel -- fill
// compact blobs.  The Z-order curve used until round 6 jumps at every power-of-two boundary: on the headline system its clusters
// give 27 970 cluster-pair steps per replica where the Hilbert order on 16^3 cells gives 23 389 (tools/models/cluster_order_model.py).
__device__ __forceinline__ unsigned hilbert3(unsigned x, unsigned y, unsigned z, int bits)
{
    unsigned X[3] = { x, y, z };
    const unsigned M = 1u << (bits - 1);
    for (unsigned Q = M; Q > 1u; Q >>= 1) {
        const unsigned P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0u;
    for (unsigned Q = M; Q > 1u; Q >>= 1) if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    unsigned k = 0u;
    for (int b = bits - 1; b >= 0; --b) {
#pragma unroll
        for (int i = 0; i < 3; ++i) k = (k << 1) | ((X[i] >> b) & 1u);
    }
    return k;
}

// one workgroup per replica: rank the groups by (curve index of the cell of the group's first atom, group index), then
// lay their atoms out contiguously.  Keys are unique, so the order is deterministic.  The work arrays (key, first atom / size /
// offset by rank: 16 - 20 bytes per group) live in LDS up to 8191 groups (sort_groups_kernel) and in a global scratch buffer
// beyond (sort_groups_large_kernel, 64-bit keys; round 4: systems of more than 8191 molecules used to leave the cluster-pair
// path for the tile kernel).  The ranking is G comparisons per group -- every 40 evaluations, a few hundred microseconds for
// 30 k molecules.
template <typename KEY>
__device__ __forceinline__
void sort_groups_body(int G, int N, int Npad, const int* __restrict__ grp_first, const int* __restrict__ grp_size,
                      const float4* __restrict__ pos, const float* __restrict__ box, float cell, int* __restrict__ order,
                      KEY* key, int* r_first, int* r_size, int* r_off, int* s_part, int hbits)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    // hbits > 0: 2^hbits cells per edge, ordered along the Hilbert curve; 0: cells of edge `cell` along the Z-order curve (until round 6)
    const int ncx = hbits > 0 ? (1 << hbits) : max(1, min(63, (int)(Lx / cell))), ncy = hbits > 0 ? (1 << hbits) : max(1, min(63, (int)(Ly / cell))),
              ncz = hbits > 0 ? (1 << hbits) : max(1, min(63, (int)(Lz / cell)));
    constexpr int GBITS = sizeof(KEY) == 8 ? 32 : 13;
    for (int g = tid; g < G; g += 1024) {
        const float4 x = pos[(size_t)r * Npad + grp_first[g]];
        float fx = x.x / Lx, fy = x.y / Ly, fz = x.z / Lz;
        fx -= floorf(fx); fy -= floorf(fy); fz -= floorf(fz);
        const unsigned cx = min(ncx - 1, (int)(fx * ncx)), cy = min(ncy - 1, (int)(fy * ncy)), cz = min(ncz - 1, (int)(fz * ncz));
        key[g] = ((KEY)(hbits > 0 ? hilbert3(cx, cy, cz, hbits) : morton3(cx, cy, cz)) << GBITS) | (KEY)(unsigned)g;
    }
    __syncthreads();
    for (int g = tid; g < G; g += 1024) {
        const KEY k = key[g];
        int rank = 0;
        for (int o = 0; o < G; ++o) rank += (key[o] < k) ? 1 : 0;
        r_first[rank] = grp_first[g];
        r_size[rank] = grp_size[g];
    }
    __syncthreads();
    const int per = (G + 1023) / 1024;
    const int b = min(G, tid * per), e = min(G, b + per);
    int sum = 0;
    for (int k = b; k < e; ++k) sum += r_size[k];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = (tid > 0) ? s_part[tid - 1] : 0;
    for (int k = b; k < e; ++k) { r_off[k] = run; run += r_size[k]; }
    __syncthreads();
    int* O = order + (size_t)r * Npad;
    for (int k = tid; k < G; k += 1024) {
        const int st = r_off[k], f = r_first[k], n = r_size[k];
        for (int a = 0; a < n; ++a) O[st + a] = f + a;
    }
    for (int k = N + tid; k < Npad; k += 1024) O[k] = -1;
}

__global__ __launch_bounds__(1024)
void sort_groups_kernel(int G, int N, int Npad, const int* __restrict__ grp_first, const int* __restrict__ grp_size,
                        const float4* __restrict__ pos, const float* __restrict__ box, float cell, int* __restrict__ order, int hbits)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* key = reinterpret_cast<unsigned*>(smem);       // [G]
    int* r_first = reinterpret_cast<int*>(key + G);          // [G] by rank
    __shared__ int s_part[1024];
    sort_groups_body<unsigned>(G, N, Npad, grp_first, grp_size, pos, box, cell, order, key, r_first, r_first + G, r_first + 2 * G, s_part, hbits);
}

// scratch: [R][5 G] ints (the 64-bit keys first)
__global__ __launch_bounds__(1024)
void sort_groups_large_kernel(int G, int N, int Npad, const int* __restrict__ grp_first, const int* __restrict__ grp_size,
                              const float4* __restrict__ pos, const float* __restrict__ box, float cell, int* __restrict__ order,
                              int* __restrict__ scratch, int hbits)
{
    __shared__ int s_part[1024];
    int* base = scratch + (size_t)blockIdx.x * 5 * G;
    unsigned long long* key = reinterpret_cast<unsigned long long*>(base);
    int* r_first = base + 2 * G;
    sort_groups_body<unsigned long long>(G, N, Npad, grp_first, grp_size, pos, box, cell, order, key, r_first, r_first + G, r_first + 2 * G, s_part, hbits);
}

__global__ __launch_bounds__(256)
void gather_params_kernel(int Npad, int words, const int* __restrict__ order, const float4* __restrict__ param,
                          const unsigned long long* __restrict__ mask, float4* __restrict__ sparam,
                          unsigned long long* __restrict__ smask)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (k >= Npad) return;
    const int o = order[(size_t)r * Npad + k];
    sparam[(size_t)r * Npad + k] = (o >= 0) ? param[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < words; ++w)
        smask[((size_t)r * Npad + k) * words + w] = (o >= 0) ? mask[(size_t)o * words + w] : 0ull;
}

// LJ-active atoms in the order of the sorted slots (one workgroup per replica): flags -> exclusive scan -> compaction;
// parameters and exclusion windows are gathered from the LJ-ordinal tables built on the host.
__global__ __launch_bounds__(1024)
void compact_lj_kernel(int N, int Npad, int NL, int NLpad, int words, const int* __restrict__ order, const int* __restrict__ lj_ord,
                       const float4* __restrict__ param, const unsigned long long* __restrict__ lj_mask,
                       int* __restrict__ lj_order, float4* __restrict__ lj_sparam, unsigned long long* __restrict__ lj_smask)
{
    __shared__ int s_part[1024];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int* O = order + (size_t)r * Npad;
    const int per = (N + 1023) / 1024;
    const int b = min(N, tid * per), e = min(N, b + per);
    int cnt = 0;
    for (int k = b; k < e; ++k) cnt += (lj_ord[O[k]] >= 0) ? 1 : 0;
    s_part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int slot = (tid > 0) ? s_part[tid - 1] : 0;
    for (int k = b; k < e; ++k) {
        const int a = O[k];
        const int ord = lj_ord[a];
        if (ord >= 0) {
            lj_order[(size_t)r * NLpad + slot] = a;
            lj_sparam[(size_t)r * NLpad + slot] = param[a];
            for (int w = 0; w < words; ++w) lj_smask[((size_t)r * NLpad + slot) * words + w] = lj_mask[(size_t)ord * words + w];
            ++slot;
        }
    }
    for (int k = NL + tid; k < NLpad; k += 1024) {
        lj_order[(size_t)r * NLpad + k] = -1;
        lj_sparam[(size_t)r * NLpad + k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < words; ++w) lj_smask[((size_t)r * NLpad + k) * words + w] = 0ull;
    }
}

// positions into sorted order + bounding box of every 64-atom tile (relative to the tile's first atom, minimum image)
__device__ __forceinline__
void gather_positions_body(int tile, int nt, int r, int lane, int Npad, int Npad_pos, const int* __restrict__ order,
                           const float4* __restrict__ pos, const float* __restrict__ box, float4* __restrict__ spos,
                           float4* __restrict__ tile_c, float4* __restrict__ tile_h, float4* __restrict__ cl_c,
                           float4* __restrict__ cl_h, const float4* __restrict__ sparam = nullptr, float4* __restrict__ sposi = nullptr)
{
    const int k = tile * 64 + lane;
    const int o = order[(size_t)r * Npad + k];
    float4 x = (o >= 0) ? pos[(size_t)r * Npad_pos + o] : make_float4(0.f, 0.f, 0.f, 0.f);
    // .w = the atom's charge (q sqrt(k_e), sorted parameter table): the Coulomb-only pair kernel then needs ONE 16-byte load per
    // atom instead of two, which also frees the registers for a second entry of prefetch
    if (sparam) x.w = sparam[(size_t)r * Npad + k].x;
    spos[(size_t)r * Npad + k] = x;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    if (sposi) {
        // the same atom as 32-bit fractions of the box edges: a difference of two such integers wraps modulo 2^32, which IS the
        // minimum image (three subtractions instead of subtract / multiply / round / fused multiply-add per component in the
        // Coulomb-only pair loop); 2^-32 of an edge is far below the f32 resolution of the positions themselves
        auto frac = [](float v, float L) { float f = v / L; f -= floorf(f); return (unsigned int)(f * 4294967296.f); };
        const unsigned int fx = frac(x.x, Lx), fy = frac(x.y, Ly), fz = frac(x.z, Lz);
        sposi[(size_t)r * Npad + k] = make_float4(__uint_as_float(fx), __uint_as_float(fy), __uint_as_float(fz), x.w);
    }
    const float x0 = __shfl(x.x, 0), y0 = __shfl(x.y, 0), z0 = __shfl(x.z, 0);   // lane 0 of a tile is always a real atom
    float dx = x.x - x0, dy = x.y - y0, dz = x.z - z0;
    dx -= Lx * rintf(dx / Lx); dy -= Ly * rintf(dy / Ly); dz -= Lz * rintf(dz / Lz);
    if (o < 0) { dx = dy = dz = 0.f; }
    float lox = dx, hix = dx, loy = dy, hiy = dy, loz = dz, hiz = dz;
    for (int off = 32; off > 0; off >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, off)); hix = fmaxf(hix, __shfl_xor(hix, off));
        loy = fminf(loy, __shfl_xor(loy, off)); hiy = fmaxf(hiy, __shfl_xor(hiy, off));
        loz = fminf(loz, __shfl_xor(loz, off)); hiz = fmaxf(hiz, __shfl_xor(hiz, off));
    }
    if (lane == 0) {
        tile_c[(size_t)r * nt + tile] = make_float4(x0 + 0.5f * (lox + hix), y0 + 0.5f * (loy + hiy), z0 + 0.5f * (loz + hiz), 0.f);
        tile_h[(size_t)r * nt + tile] = make_float4(0.5f * (hix - lox), 0.5f * (hiy - loy), 0.5f * (hiz - loz), 0.f);
    }
    if (cl_c) {
        // 8-atom clusters: same construction inside each group of 8 lanes (relative to the group's first atom)
        const int g0 = lane & ~7;
        const float cx0 = __shfl(x.x, g0), cy0 = __shfl(x.y, g0), cz0 = __shfl(x.z, g0);
        const int o0 = __shfl(o, g0);
        float ex = x.x - cx0, ey = x.y - cy0, ez = x.z - cz0;
        ex -= Lx * rintf(ex / Lx); ey -= Ly * rintf(ey / Ly); ez -= Lz * rintf(ez / Lz);
        if (o < 0) { ex = ey = ez = 0.f; }
        float ax = ex, bx2 = ex, ay = ey, by2 = ey, az = ez, bz2 = ez;
        for (int off = 4; off > 0; off >>= 1) {
            ax = fminf(ax, __shfl_xor(ax, off)); bx2 = fmaxf(bx2, __shfl_xor(bx2, off));
            ay = fminf(ay, __shfl_xor(ay, off)); by2 = fmaxf(by2, __shfl_xor(by2, off));
            az = fminf(az, __shfl_xor(az, off)); bz2 = fmaxf(bz2, __shfl_xor(bz2, off));
        }
        if ((lane & 7) == 0) {
            const int ncl = nt * 8;
            const int c = tile * 8 + (lane >> 3);
            // an empty (all-padding) cluster is parked far away with a negative extent so that it never pairs
            cl_c[(size_t)r * ncl + c] = (o0 >= 0) ? make_float4(cx0 + 0.5f * (ax + bx2), cy0 + 0.5f * (ay + by2), cz0 + 0.5f * (az + bz2), 0.f)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            cl_h[(size_t)r * ncl + c] = (o0 >= 0) ? make_float4(0.5f * (bx2 - ax), 0.5f * (by2 - ay), 0.5f * (bz2 - az), 0.f)
                                                  : make_float4(-1e9f, -1e9f, -1e9f, 0.f);
        }
    }
}

__global__ __launch_bounds__(64)
void gather_positions_kernel(int Npad, int Npad_pos, const int* __restrict__ order, const float4* __restrict__ pos,
                             const float* __restrict__ box, float4* __restrict__ spos, float4* __restrict__ tile_c,
                             float4* __restrict__ tile_h, float4* __restrict__ cl_c, float4* __restrict__ cl_h)
{
    gather_positions_body(blockIdx.x, gridDim.x, blockIdx.y, threadIdx.x, Npad, Npad_pos, order, pos, box, spos, tile_c, tile_h, cl_c, cl_h);
}

// main system (first nt_a tiles of the grid) and LJ sub-system in one launch: one dependent launch less per evaluation
struct gather_args { int Npad; const int* order; float4* spos; float4* tile_c; float4* tile_h; float4* cl_c; float4* cl_h; const float4* sparam; float4* sposi; };
__global__ __launch_bounds__(64)
void gather_positions2_kernel(int nt_a, gather_args a, gather_args b, int Npad_pos, const float4* __restrict__ pos,
                              const float* __restrict__ box)
{
    const bool second = (int)blockIdx.x >= nt_a;
    const gather_args& g = second ? b : a;
    gather_positions_body(second ? blockIdx.x - nt_a : blockIdx.x, second ? gridDim.x - nt_a : nt_a, blockIdx.y, threadIdx.x, g.Npad, Npad_pos,
                          g.order, pos, box, g.spos, g.tile_c, g.tile_h, g.cl_c, g.cl_h, g.sparam, g.sposi);
}

// ---- Newton's-third-law path: super-cluster ("sci") lists -----------------------------------------------------------
// One list per 64-atom tile (= 8 consecutive i clusters): the ascending union of the j clusters that neighbour any of
// them, each entry carrying the 8-bit mask of the i clusters it pairs with.  Only cluster pairs with jc >= ic are listed,
// so every atom pair is evaluated once; the kernel accumulates the reaction on the j atoms in registers across the
// (up to 8) i clusters of an entry and adds it with one reduction + 8 integer atomics per entry.  Forces go to an
// accumulator in SORTED slot space, where the 8 atoms of a cluster are one aligned 64-byte line (the L2 atomic units
// are limited by line requests: scattering to the atoms' own slots costs 3-4 lines per cluster and made this kernel
// 93 us instead of ~55); scatter_sorted_forces_kernel folds it into the per-atom accumulator afterwards.

// sorted-slot force accumulator -> per-atom accumulator (integer atomics: the PME gather may be adding on the other
// stream); the slot is cleared for the next evaluation
__device__ __forceinline__
void scatter_sorted_forces_body(int Npad_a, const int* __restrict__ order_a, long long* __restrict__ sforce_a,
                                int Npad_b, const int* __restrict__ order_b, long long* __restrict__ sforce_b,
                                long long* __restrict__ force, int Npad_force, int k, int r)
{
    const bool second = k >= Npad_a;                      // main system slots first, then the LJ sub-system's
    if (second) k -= Npad_a;
    const int Npad_s = second ? Npad_b : Npad_a;
    if (k >= Npad_s) return;
    long long* S = (second ? sforce_b : sforce_a) + (size_t)r * 3 * Npad_s;
    const int o = (second ? order_b : order_a)[(size_t)r * Npad_s + k];
    unsigned long long* U = reinterpret_cast<unsigned long long*>(force + (size_t)r * 3 * Npad_force);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long long v = S[(size_t)c * Npad_s + k];
        if (v != 0) { S[(size_t)c * Npad_s + k] = 0; if (o >= 0) atomicAdd(&U[(size_t)c * Npad_force + o], (unsigned long long)v); }
    }
}

__global__ __launch_bounds__(256)
void scatter_sorted_forces_kernel(int Npad_a, const int* __restrict__ order_a, long long* __restrict__ sforce_a,
                                  int Npad_b, const int* __restrict__ order_b, long long* __restrict__ sforce_b,
                                  long long* __restrict__ force, int Npad_force, unsigned int* __restrict__ done = nullptr)
{
    scatter_sorted_forces_body(Npad_a, order_a, sforce_a, Npad_b, order_b, sforce_b, force, Npad_force, blockIdx.x * 256 + threadIdx.x, blockIdx.y);
    // remd_fold_args: every thread drains its OWN force atomics (s_waitcnt vmcnt(0): the compiler puts no wait in front of the
    // barrier -- the fence of __syncthreads is workgroup scope and gfx950 has a back-off barrier), then the barrier, then
    // the arrival.  The force sums are device-scope read-modify-writes, performed at the memory side and visible to every XCD once
    // acknowledged, so the arrival needs NO release fence: a device-scope release writes back the whole L2 of the XCD it runs on,
    // and 312 of them made this launch 28 us instead of 6 (profiles/r04_p_*)
    if (done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done + 16 * blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// lane <-> (i atom ii, j atom jj) of a cluster pair in the sci kernels.  SCI_LANES_IJ = 1 (round 4): ii in the LOW three lane
// bits, so the per-entry all-reduce of the reaction forces over the 8 i atoms is three DPP adds inside groups of 8 lanes
// (quad permutes + half-row mirror) instead of a row rotation and two cross-row swaps per component (27 -> 9 instructions per
// list entry); the exclusion words are ballots in the same lane order (build_excl_kernel).
#ifndef SCI_LANES_IJ
#define SCI_LANES_IJ 1
#endif
#if SCI_LANES_IJ
#define SCI_LANE_II(lane) ((lane) & 7)
#define SCI_LANE_JJ(lane) ((lane) >> 3)
#define SCI_LANE_OF(ii, jj) ((jj) * 8 + (ii))
#else
#define SCI_LANE_II(lane) ((lane) >> 3)
#define SCI_LANE_JJ(lane) ((lane) & 7)
#define SCI_LANE_OF(ii, jj) ((ii) * 8 + (jj))
#endif

__global__ __launch_bounds__(64)
void build_excl_kernel(int Npad, int ncl, int W, int words, const unsigned long long* __restrict__ smask,
                       unsigned long long* __restrict__ excl)
{
    const int ic = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const int ii = SCI_LANE_II(lane), jj = SCI_LANE_JJ(lane);          // bit = the pair kernel's lane of (i atom, j atom)
    const int half = 32 * words;
    const int i = ic * 8 + ii;
    for (int dj = 0; dj < W; ++dj) {
        const int j = (ic + dj) * 8 + jj;
        const int d = j - i + half;
        bool ex = (dj == 0 && jj <= ii);
        if (d >= 0 && d < 2 * half) ex = ex || ((smask[((size_t)r * Npad + i) * words + (d >> 6)] >> (d & 63)) & 1ull);
        const unsigned long long m = __ballot(ex);
        if (lane == 0) excl[((size_t)r * ncl + ic) * W + dj] = m;
    }
}

__device__ __forceinline__
void build_sci_list_body(int T, int r, int lane, int ncl, int cap, float rc2, const float4* __restrict__ cl_c,
                         const float4* __restrict__ cl_h, const float4* __restrict__ tile_c, const float4* __restrict__ tile_h,
                         const float* __restrict__ box, unsigned int* __restrict__ list, int* __restrict__ count)
{
    __shared__ int s_tiles[64];
    __shared__ float4 s_ci[8], s_hi[8];
    const int ntile = ncl >> 3;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const float iLx = 1.f / Lx, iLy = 1.f / Ly, iLz = 1.f / Lz;
    if (lane < 8) { s_ci[lane] = cl_c[(size_t)r * ncl + T * 8 + lane]; s_hi[lane] = cl_h[(size_t)r * ncl + T * 8 + lane]; }
    const float4 ct = tile_c[(size_t)r * ntile + T], ht = tile_h[(size_t)r * ntile + T];
    auto near = [&](const float4 ci, const float4 hi, const float4 cj, const float4 hj) {
        float bx = cj.x - ci.x, by = cj.y - ci.y, bz = cj.z - ci.z;
        bx -= Lx * rintf(bx * iLx); by -= Ly * rintf(by * iLy); bz -= Lz * rintf(bz * iLz);
        bx = fmaxf(0.f, fabsf(bx) - hi.x - hj.x); by = fmaxf(0.f, fabsf(by) - hi.y - hj.y); bz = fmaxf(0.f, fabsf(bz) - hi.z - hj.z);
        return (hi.x >= 0.f) && (bx * bx + by * by + bz * bz <= rc2);
    };
    unsigned int* L = list + ((size_t)r * ntile + T) * cap;
    int n = 0;
    for (int tbase = T; tbase < ntile; tbase += 64) {
        const int jt = tbase + lane;
        const bool thit = (jt < ntile) && near(ct, ht, tile_c[(size_t)r * ntile + jt], tile_h[(size_t)r * ntile + jt]);
        const unsigned long long tm = __ballot(thit);
        const int nhit = __popcll(tm);
        __syncthreads();                                   // (one wavefront: orders the LDS reuse between chunks)
        if (thit) s_tiles[__popcll(tm & ((1ull << lane) - 1ull))] = jt;
        __syncthreads();
        for (int q = 0; q < nhit; q += 8) {                // 8 hit tiles = 64 j clusters per pass, one per lane
            const int g = q + (lane >> 3);
            unsigned int imask = 0u;
            int jc = 0;
            if (g < nhit) {
                jc = s_tiles[g] * 8 + (lane & 7);
                const float4 cj = cl_c[(size_t)r * ncl + jc], hj = cl_h[(size_t)r * ncl + jc];   // empty clusters carry a negative extent
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (jc >= T * 8 + s && near(s_ci[s], s_hi[s], cj, hj)) imask |= 1u << s;
            }
            const unsigned long long m = __ballot(imask != 0u);
            if (imask != 0u) {
                const int slot = n + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cap) L[slot] = (unsigned int)jc | (imask << 16);
            }
            n += __popcll(m);
        }
    }
    if (lane == 0) count[(size_t)r * ntile + T] = n;      // n > cap is detected on the host side (fallback)
}

// tiles of every replica by descending list length (ties: ascending tile index): tile_of_rank[r][k] = the tile with the k-th longest list
__global__ __launch_bounds__(256)
void rank_tiles_kernel(int ntile, const int* __restrict__ count, int* __restrict__ tile_of_rank)
{
    const int r = blockIdx.y;
    const int* c = count + (size_t)r * ntile;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntile; t += gridDim.x * 256) {
        const int mine = c[t];
        int rank = 0;
        for (int q = 0; q < ntile; ++q) { const int o = c[q]; rank += (o > mine || (o == mine && q < t)) ? 1 : 0; }
        tile_of_rank[(size_t)r * ntile + rank] = t;
    }
}

__global__ __launch_bounds__(64)
void build_sci_list_kernel(int ncl, int cap, float rc2, const float4* __restrict__ cl_c, const float4* __restrict__ cl_h,
                           const float4* __restrict__ tile_c, const float4* __restrict__ tile_h,
                           const float* __restrict__ box, unsigned int* __restrict__ list, int* __restrict__ count)
{
    build_sci_list_body(blockIdx.x, blockIdx.y, threadIdx.x, ncl, cap, rc2, cl_c, cl_h, tile_c, tile_h, box, list, count);
}

struct sci_list_args { int ncl, cap; const float4* cl_c; const float4* cl_h; const float4* tile_c; const float4* tile_h; unsigned int* list; int* count; };
// (rc2_a / rc2_b: the main system of an Ewald method lists to the Coulomb range nb_params::rcc2, the LJ sub-system to the
// NonbondedForce cutoff)
__global__ __launch_bounds__(64)
void build_sci_list2_kernel(int nt_a, sci_list_args a, sci_list_args b, float rc2_a, float rc2_b, const float* __restrict__ box)
{
    const bool second = (int)blockIdx.x >= nt_a;
    const sci_list_args& g = second ? b : a;
    build_sci_list_body(second ? blockIdx.x - nt_a : blockIdx.x, blockIdx.y, threadIdx.x, g.ncl, g.cap, second ? rc2_b : rc2_a, g.cl_c, g.cl_h,
                        g.tile_c, g.tile_h, box, g.list, g.count);
}

// all-reduce over the lanes that differ in bit 3 / 4 / 5 of the lane id without the LDS crossbar (ds_bpermute: address
// arithmetic + a round trip per exchange): a row rotation by 8 (DPP) and the gfx950 row / half swaps
__device__ __forceinline__ float allsum_x8(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}
// all-reduce inside every group of 8 consecutive lanes: two quad permutes and the half-row mirror (DPP, no cross-row traffic)
__device__ __forceinline__ float allsum_low8(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}
__device__ __forceinline__ float allsum_x16(float v)
{
    float a = v, b = v;        // odd rows of a <-> even rows of b
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float allsum_x32(float v)
{
    float a = v, b = v;        // upper half of a <-> lower half of b
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// v where bit `lane` of the wave-uniform mask is set, else 0: one v_cndmask on the scalar mask
__device__ __forceinline__ float keep_where(unsigned long long m, float v)
{
    float o;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(m));
    return o;
}
__device__ __forceinline__ float max_sv(float s_uniform, float v)
{
    float o;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(o) : "s"(s_uniform), "v"(v));
    return o;
}
__device__ __forceinline__ float min_sv(float s_uniform, float v)
{
    float o;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(o) : "s"(s_uniform), "v"(v));
    return o;
}

// lane = (ii, jj) by SCI_LANE_II / SCI_LANE_JJ above: the lane holds atom ii of all 8 i clusters of its tile in registers and, per
// list entry, atom jj of the j cluster.
#define SCI_NW 4
// NW wavefronts per workgroup take consecutive slices of one tile's list and merge their i forces through LDS (fixed
// order), so the i atoms are flushed once per workgroup.
struct sci_args {
    int N, Npad, ncl, cap, W, Npad_force, ep_off, nsplit;
    const float4* spos; const float4* sparam; const unsigned long long* excl; const unsigned int* list; const int* count;
    long long* force;
    const float4* sposi;        // positions as 32-bit box fractions + charge (gather_positions_body), or NULL
    const int* tile_of_rank;    // [R][ntile] tiles of a replica by descending list length (rank_tiles_kernel), or NULL: item order = tile order
};
#define SCI_EWALD(M) ((M) == NB_EWALD || (M) == NB_EWALD_NOLJ)
// A/B switches of the round-4 pair-kernel changes (tools/build_variant.sh -DSCI_...=0)
#ifndef SCI_EARLY_TABLE
#define SCI_EARLY_TABLE 1
#endif
#ifndef SCI_PREFETCH2
#define SCI_PREFETCH2 1
#endif
#ifndef SCI_PACKQ
#define SCI_PACKQ 1
#endif
#ifndef SCI_INTCOORD
#define SCI_INTCOORD 1
#endif
// x / y components of the separation and of both force accumulators as register pairs: one v_pk_fma_f32 each instead of a packed
// multiply and two adds
#ifndef SCI_PKACC
#define SCI_PKACC 1
#endif
#ifndef SCI_TABIDX
#define SCI_TABIDX 1
#endif
#ifndef SCI_DPP_ASM
#define SCI_DPP_ASM 1
#endif
typedef float sci_v2f __attribute__((ext_vector_type(2)));
template <int METHOD, bool ENERGY, bool ALCH, int NW, bool TABLE = false, bool INTPOS = false>
__device__ __forceinline__
void nonbonded_sci_body(const nb_params& p, const sci_args& a, int item, const float* __restrict__ box,
                        const float* __restrict__ rep_lam, double* __restrict__ epart, int n_epart, int R,
                        const float4* ctab = nullptr)
{
    constexpr bool TAB = TABLE && !ENERGY && SCI_EWALD(METHOD);
    const float rcut2 = SCI_EWALD(METHOD) ? p.rcc2 : p.rc2;     // Coulomb range of the Ewald split / NonbondedForce cutoff
    const int N = a.N, Npad = a.Npad, ncl = a.ncl, cap = a.cap, W = a.W, Npad_force = a.Npad_force, ep_off = a.ep_off, nsplit = a.nsplit;
    const float4* __restrict__ spos = a.spos; const float4* __restrict__ sparam = a.sparam;
    const unsigned long long* __restrict__ excl = a.excl; const unsigned int* __restrict__ list = a.list;
    const int* __restrict__ count = a.count; long long* __restrict__ force = a.force;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntile = ncl >> 3;
    // work items are dispatched in index order and a tile's list is up to twice the average length: with the tiles in list-length
    // order (longest first, all replicas' longest before anybody's second) the long items start at the front of the launch and the
    // short ones fill its tail, instead of a long item that starts in the last round setting the duration of the launch
    int T, r, zsl;
    if (a.tile_of_rank) {
        const int nzg = nsplit / NW, per_rank = R * nzg;
        const int k = item / per_rank, rem = item - k * per_rank;
        r = rem / nzg; zsl = (rem - r * nzg) * NW + wv;
        T = a.tile_of_rank[(size_t)r * ntile + k];
    } else {
        T = item % ntile; r = (item / ntile) % R; zsl = (item / (ntile * R)) * NW + wv;
    }
    const int ii = SCI_LANE_II(lane), jj = SCI_LANE_JJ(lane);
    // force-only Coulomb-only kernel: positions as integer box fractions (sci_args::sposi), minimum image by wrap-around
    // (INTPOS: the caller also has integer positions for a system that keeps its parameter loads -- the LJ sub-system of a split launch)
    constexpr bool INTC = SCI_INTCOORD && !ENERGY && ((SCI_PACKQ && !ALCH && (METHOD == NB_EWALD_NOLJ || METHOD == NB_RF_NOLJ)) || INTPOS);
    const float4* __restrict__ P = (INTC ? a.sposi : spos) + (size_t)r * Npad;      // (the split launch always passes sposi)
    const float4* __restrict__ prm = sparam + (size_t)r * Npad;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const float iLx = 1.f / Lx, iLy = 1.f / Ly, iLz = 1.f / Lz;
    const float sLx = Lx * (1.f / 4294967296.f), sLy = Ly * (1.f / 4294967296.f), sLz = Lz * (1.f / 4294967296.f);
    float lam_a = 1.f, sc = 0.f, lam_e = 1.f;
    if (ALCH) { lam_a = rep_lam[4 * r]; sc = rep_lam[4 * r + 1]; lam_e = rep_lam[4 * r + 2]; }
    const int c_last = (N - 1) >> 3;                         // the only cluster that can mix real and padding atoms
    const unsigned long long valid_j = __builtin_amdgcn_ballot_w64(c_last * 8 + jj < N), valid_i = __builtin_amdgcn_ballot_w64(c_last * 8 + ii < N);

    // slice zsl takes the entries zsl, zsl + nsplit, ...: the near-diagonal entries (most i clusters per entry) come first
    // in a list, so contiguous slices would be unevenly loaded
    const int n_all = min(count[(size_t)r * ntile + T], cap);
    const int n = (n_all - zsl + nsplit - 1) / nsplit;
    if (NW == 1 && n <= 0) return;
    const unsigned int* L = list + ((size_t)r * ntile + T) * cap + zsl;

    // Coulomb-only main kernel of a split system: the charge rides in spos.w (gather_positions_body), no parameter loads
    constexpr bool PACKQ = SCI_PACKQ && (METHOD == NB_EWALD_NOLJ || METHOD == NB_RF_NOLJ) && !ALCH;
    constexpr bool PF2 = PACKQ && SCI_PREFETCH2;
    float4 xi[8], pi[8];
    float fix[8], fiy[8], fiz[8];
    sci_v2f fixy[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        fixy[s] = sci_v2f{ 0.f, 0.f };
        const int i = (T * 8 + s) * 8 + ii;
        xi[s] = P[i];
        if (PACKQ) pi[s] = make_float4(xi[s].w, 0.f, 0.f, 0.f); else pi[s] = prm[i];
        if (ALCH && pi[s].w != 0.f) pi[s].x *= lam_e;
        fix[s] = fiy[s] = fiz[s] = 0.f;
    }
    // exclusion words of this tile's near-diagonal pairs: lane s*8 + dj holds excl[ic_s][dj] (W <= 8), read back with readlane
    unsigned int ex_lo = 0u, ex_hi = 0u;
    if (W <= 8 && jj < W) {
        const unsigned long long m = excl[((size_t)r * ncl + T * 8 + ii) * W + jj];
        ex_lo = (unsigned int)m; ex_hi = (unsigned int)(m >> 32);
    }
    double e = 0.0;
    // j forces of 8 consecutive entries are parked in the lane group ii = entry & 7 and flushed together: 3 full-width
    // atomic instructions per 8 entries.  (Loads and atomics share one in-order wait counter, so every atomic issued
    // inside the entry loop makes the wait for the next prefetched j atoms a full L2 round trip.)
    float qfx = 0.f, qfy = 0.f, qfz = 0.f;
    int qj = 0;

    for (int base = 0; base < n; base += 64) {
        const int cnt = min(64, n - base);                       // list entries in this 64-chunk
        const unsigned int my_ent = (lane < cnt) ? L[(size_t)(base + lane) * nsplit] : 0u;
        int jn = (int)(__builtin_amdgcn_readlane(my_ent, 0) & 0xffffu);
        float4 gx = P[jn * 8 + jj], gp = PACKQ ? gx : prm[jn * 8 + jj];
        float4 gx2 = gx;                                         // PACKQ: the entry after the next one
        if (PF2) { jn = (int)(__builtin_amdgcn_readlane(my_ent, min(1, cnt - 1)) & 0xffffu); gx2 = P[jn * 8 + jj]; }
        for (int k = 0; k < cnt; ++k) {
            const unsigned int ent = __builtin_amdgcn_readlane(my_ent, k);
            const int jc = (int)(ent & 0xffffu);
            const unsigned int imask = ent >> 16;
            const float4 xj = gx;
            float4 pj = PACKQ ? make_float4(gx.w, 0.f, 0.f, 0.f) : gp;
            // prefetch the next entry's j atoms behind this entry's arithmetic (unconditional: the last entry of a chunk
            // re-reads itself, which keeps the loop free of a branch the register allocator would pin copies to); the packed
            // Coulomb-only kernel keeps two entries in flight
            if (PF2) {
                gx = gx2;
                jn = (int)(__builtin_amdgcn_readlane(my_ent, min(k + 2, cnt - 1)) & 0xffffu);
                gx2 = P[jn * 8 + jj];
            } else {
                jn = (int)(__builtin_amdgcn_readlane(my_ent, min(k + 1, cnt - 1)) & 0xffffu);
                gx = P[jn * 8 + jj]; if (!PACKQ) gp = prm[jn * 8 + jj];
            }
            if (ALCH && pj.w != 0.f) pj.x *= lam_e;
            const int j = jc * 8 + jj;
            float fjx = 0.f, fjy = 0.f, fjz = 0.f;
            sci_v2f fjxy = { 0.f, 0.f };
            bool touched = false;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (!((imask >> s) & 1u)) continue;              // wave-uniform
                const int ic = T * 8 + s;
                float dx, dy, dz;
                if (INTC) {
                    dx = (float)(int)(__float_as_uint(xj.x) - __float_as_uint(xi[s].x)) * sLx;
                    dy = (float)(int)(__float_as_uint(xj.y) - __float_as_uint(xi[s].y)) * sLy;
                    dz = (float)(int)(__float_as_uint(xj.z) - __float_as_uint(xi[s].z)) * sLz;
                } else {
                    dx = xj.x - xi[s].x; dy = xj.y - xi[s].y; dz = xj.z - xi[s].z;
                    dx -= Lx * rintf(dx * iLx); dy -= Ly * rintf(dy * iLy); dz -= Lz * rintf(dz * iLz);
                }
                const float r2 = dx * dx + dy * dy + dz * dz;
                // which lane pairs count is wave-uniform data (cutoff ballot, exclusion word, padding masks): scalar unit
                unsigned long long in = __builtin_amdgcn_ballot_w64(r2 < rcut2);
                // lanes beyond the cutoff evaluate at the cutoff (the table also has a lower end)
                // (SCI_TABIDX, the early-table kernel: only the lower end is clamped.  Lanes beyond the cutoff are discarded by the
                // select below whatever they read, and an LDS address past the workgroup's allocation reads as zero)
                constexpr bool EARLY = SCI_EARLY_TABLE && TAB && METHOD == NB_EWALD_NOLJ && !ALCH;
                float r2c = (EARLY && SCI_TABIDX) ? r2
                          : TAB ? __builtin_amdgcn_fmed3f(r2, p.ctab_umin, rcut2) : min_sv(rcut2, r2);
                // Coulomb-only kernel from the table: the LDS read is issued here, in front of the scalar mask chain and its
                // branches, so that its round trip overlaps them instead of stalling the polynomial that consumes it
                float4 tc = make_float4(0.f, 0.f, 0.f, 0.f); float ttf = 0.f;
                if (EARLY) {
                    if (SCI_TABIDX) {
                        // lower clamp, key = bits [18, 32) as a bit-field extract, LDS address = key * 16 + base as one shift-add:
                        // three instructions (the plain expression: a register copy of the bound, max, shift, mask, add)
                        typedef float sci_v4f __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(3))) const sci_v4f lds_v4f;
                        const unsigned int lds_base = (unsigned int)(__UINTPTR_TYPE__)(lds_v4f*)ctab;
                        unsigned int addr;
                        asm("v_max_f32_e32 %0, %2, %3\n\tv_bfe_u32 %1, %0, %5, %6\n\tv_lshl_add_u32 %1, %1, 4, %4"
                            : "=&v"(r2c), "=&v"(addr) : "s"(p.ctab_umin), "v"(r2), "s"(lds_base), "n"(CTAB_SHIFT), "n"(32 - CTAB_SHIFT));
                        const sci_v4f c4 = *(lds_v4f*)(__UINTPTR_TYPE__)addr;
                        tc = make_float4(c4.x, c4.y, c4.z, c4.w);
                    } else {
                        tc = ctab[__float_as_uint(r2c) >> CTAB_SHIFT];
                    }
                    ttf = (float)(__float_as_uint(r2c) & CTAB_MASK);
                }
                const int dj = jc - ic;
                if (dj < W) {                                    // wave-uniform: exclusions (and the diagonal) live here
                    unsigned long long m;
                    // (readlane returns int: without the unsigned cast bit 31 of the low word would smear over lanes 32..63)
                    if (W <= 8) m = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(ex_lo, SCI_LANE_OF(s, dj))
                                  | ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(ex_hi, SCI_LANE_OF(s, dj)) << 32);
                    else {
                        const unsigned long long mv = excl[((size_t)r * ncl + ic) * W + dj];
                        m = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)mv)
                          | ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(mv >> 32)) << 32);
                    }
                    in &= ~m;
                }
                if (jc == c_last) in &= valid_j;
                if (ic == c_last) in &= valid_i;
                if (in == 0ull) continue;                        // no atom pair of this cluster pair is inside the cutoff
                touched = true;
                float fr, ee;
                // evaluated for every lane (excluded pairs, even r2 = 0, produce garbage that the select below discards)
                if (EARLY) { fr = (pi[s].x * pj.x) * fmaf(ttf, fmaf(ttf, fmaf(ttf, tc.w, tc.z), tc.y), tc.x); ee = 0.f; }
                else pair_interaction<METHOD, ALCH, !ENERGY, TAB>(p, r2c, pi[s], pj, lam_a, sc, fr, false, ee, ctab);
                fr = keep_where(in, fr);
                if (SCI_PKACC) {
                    const sci_v2f dxy = { dx, dy }, fr2 = { fr, fr };
                    fixy[s] = __builtin_elementwise_fma(dxy, fr2, fixy[s]);
                    fjxy = __builtin_elementwise_fma(-dxy, fr2, fjxy);
                    fiz[s] = fmaf(dz, fr, fiz[s]);
                    fjz = fmaf(-dz, fr, fjz);
                } else {
                    const float tx = fr * dx, ty = fr * dy, tz = fr * dz;
                    fix[s] += tx; fiy[s] += ty; fiz[s] += tz;
                    fjx -= tx; fjy -= ty; fjz -= tz;
                }
                if (ENERGY) e += ((in >> lane) & 1ull) ? (double)ee : 0.0;
            }
            if (SCI_PKACC) { fjx = fjxy.x; fjy = fjxy.y; }
            if (touched) {
                // reaction on the j atoms: all-reduce over the 8 ii lanes (every lane ends up with the total of its jj)
                if (SCI_LANES_IJ && SCI_DPP_ASM) {
                    // three DPP adds per component inside groups of 8 lanes, written out: left to the compiler the x / y pair stays a
                    // packed value (register copies + v_mov_b32_dpp + v_pk_add_f32: ~21 instructions per entry instead of these 9).
                    // (a DPP operand must not be read within two instructions of the VALU write that produced it: the leading s_nop
                    // covers the accumulators written just before, and inside the block every value is read two instructions after
                    // its own update)
                    asm("s_nop 1\n\t"
                        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf"
                        : "+v"(fjx), "+v"(fjy), "+v"(fjz));
                } else if (SCI_LANES_IJ) {
                    fjx = allsum_low8(fjx); fjy = allsum_low8(fjy); fjz = allsum_low8(fjz);
                } else {
                    fjx = allsum_x8(fjx); fjy = allsum_x8(fjy); fjz = allsum_x8(fjz);
                    fjx = allsum_x16(fjx); fjy = allsum_x16(fjy); fjz = allsum_x16(fjz);
                    fjx = allsum_x32(fjx); fjy = allsum_x32(fjy); fjz = allsum_x32(fjz);
                }
            }
            const int eb = k & 7;
            if (ii == eb) { qfx = fjx; qfy = fjy; qfz = fjz; qj = j; }
            if (eb == 7 || k + 1 == cnt) {
                // lane group ii holds entry (k & ~7) + ii: 8 lanes = one 64-byte line of the sorted accumulator
                if (ii <= eb) add_force(force + (size_t)r * 3 * Npad_force, Npad_force, qj, qfx, qfy, qfz);
            }
        }
    }
    // i forces: reduce-scatter over the 8 jj lanes (7 exchanges per component); afterwards lane (ii, jj) holds the
    // total of atom ii of cluster s = jj, so all 64 lanes issue one atomic triple
    auto reduce_scatter = [&](float (&v)[8]) {
        float a[4], b[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float send = (jj & 4) ? v[q] : v[q + 4], keep = (jj & 4) ? v[q + 4] : v[q];
            a[q] = keep + __shfl_xor(send, SCI_LANES_IJ ? 32 : 4);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float send = (jj & 2) ? a[q] : a[q + 2], keep = (jj & 2) ? a[q + 2] : a[q];
            b[q] = keep + __shfl_xor(send, SCI_LANES_IJ ? 16 : 2);
        }
        const float send = (jj & 1) ? b[0] : b[1], keep = (jj & 1) ? b[1] : b[0];
        return keep + __shfl_xor(send, SCI_LANES_IJ ? 8 : 1);
    };
    if (SCI_PKACC) {
#pragma unroll
        for (int s = 0; s < 8; ++s) { fix[s] = fixy[s].x; fiy[s] = fixy[s].y; }
    }
    float fx = reduce_scatter(fix), fy = reduce_scatter(fiy), fz = reduce_scatter(fiz);
    if (NW > 1) {
        __shared__ float s_f[NW][3][64];
        s_f[wv][0][lane] = fx; s_f[wv][1][lane] = fy; s_f[wv][2][lane] = fz;
        __syncthreads();
        if (wv == 0) {
            fx = s_f[0][0][lane]; fy = s_f[0][1][lane]; fz = s_f[0][2][lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) { fx += s_f[w][0][lane]; fy += s_f[w][1][lane]; fz += s_f[w][2][lane]; }
        }
    }
    if (wv == 0) add_force(force + (size_t)r * 3 * Npad_force, Npad_force, (T * 8 + jj) * 8 + ii, fx, fy, fz);
    if (ENERGY) {
        e = wave_sum(e);
        if (lane == 0) epart[(size_t)r * n_epart + EP_NB0 + ep_off + T * nsplit + zsl] = e;
    }
}

// 4 wavefronts per SIMD (<= 128 VGPRs) for the hot variants; the rarely used ones that would spill keep 3
#define SCI_RELAXED(M) (M == NB_RF || M == NB_EWALD || (M == NB_LJ_ONLY && ALCH))
// the Ewald force table (coulomb_table.h) of a force-only launch: global -> LDS, once per workgroup
extern __shared__ __attribute__((aligned(16))) float4 s_ctab[];
__device__ __forceinline__ void stage_coulomb_table(const nb_params& p, const float4* __restrict__ ctab_g, int nthreads)
{
    for (int k = threadIdx.x; k < p.ctab_n; k += nthreads) s_ctab[k] = ctab_g[k];
    __syncthreads();
}

template <int METHOD, bool ENERGY, bool ALCH, int NW, bool TABLE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(SCI_RELAXED(METHOD) ? 2 : 4, SCI_RELAXED(METHOD) ? 3 : 4)))
void nonbonded_sci_kernel(nb_params p, sci_args a, const float* __restrict__ box, const float* __restrict__ rep_lam,
                          double* __restrict__ epart, int n_epart, int R, const float4* __restrict__ ctab_g)
{
    constexpr bool TAB = TABLE && !ENERGY && SCI_EWALD(METHOD);
    if (TAB) stage_coulomb_table(p, ctab_g, 64 * NW);
    nonbonded_sci_body<METHOD, ENERGY, ALCH, NW, TAB>(p, a, blockIdx.x, box, rep_lam, epart, n_epart, R, s_ctab - p.ctab_key0);
}

// main (Coulomb-only) system and LJ sub-system in one launch: the short LJ work items fill the tail of the main ones
template <int METHOD_A, int METHOD_B, bool ENERGY, bool ALCH, int NW, bool TABLE>
__global__ __launch_bounds__(64 * NW)
__attribute__((amdgpu_waves_per_eu((SCI_RELAXED(METHOD_A) || SCI_RELAXED(METHOD_B)) ? 2 : 4, (SCI_RELAXED(METHOD_A) || SCI_RELAXED(METHOD_B)) ? 3 : 4)))
void nonbonded_sci2_kernel(nb_params p, sci_args a, sci_args b, int n_items_a, int n_items, unsigned int* queue, const float* __restrict__ box,
                           const float* __restrict__ rep_lam, double* __restrict__ epart, int n_epart, int R, const float4* __restrict__ ctab_g)
{
    constexpr bool TAB = TABLE && !ENERGY && SCI_EWALD(METHOD_A);
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    if (!queue) {
        if ((int)blockIdx.x < n_items_a) {
            if (TAB) stage_coulomb_table(p, ctab_g, 64 * NW);
            nonbonded_sci_body<METHOD_A, ENERGY, ALCH, NW, TAB>(p, a, blockIdx.x, box, rep_lam, epart, n_epart, R, s_ctab - p.ctab_key0);
        }
        else nonbonded_sci_body<METHOD_B, ENERGY, ALCH, NW, false, true>(p, b, blockIdx.x - n_items_a, box, rep_lam, epart, n_epart, R);
        return;
    }
    if (TAB) stage_coulomb_table(p, ctab_g, 64 * NW);
    // gridDim.x < n_items: a resident set of workgroups pulls items from two queues (Coulomb items first, then the short LJ
    // items).  The grid size bounds the share of a CU's wave slots and registers this kernel holds while the mesh kernels of
    // the other stream want them: with one workgroup per item the XY pass got 30 % of its work done next to this kernel and
    // needed a 47 us tail of its own.  Forces are fixed-point sums and energy partials are per item, so the order in which
    // items are pulled does not change a bit of the result.
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = (int)atomicAdd(&queue[0], 1u);
        __syncthreads();
        const int item = s_item;
        if (item >= n_items_a) break;
        nonbonded_sci_body<METHOD_A, ENERGY, ALCH, NW, TAB>(p, a, item, box, rep_lam, epart, n_epart, R, s_ctab - p.ctab_key0);
        __syncthreads();                 // s_item and the merge buffer are reused
    }
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) s_item = (int)atomicAdd(&queue[1], 1u);
        __syncthreads();
        const int item = s_item;
        if (item >= n_items - n_items_a) break;
        nonbonded_sci_body<METHOD_B, ENERGY, ALCH, NW, false, true>(p, b, item, box, rep_lam, epart, n_epart, R);
        __syncthreads();
    }
    // the last workgroup to leave rewinds the queues for the next launch (everybody has seen them run dry by then)
    if (threadIdx.x == 0 && atomicAdd(&queue[2], 1u) == gridDim.x - 1) { queue[0] = 0u; queue[1] = 0u; queue[2] = 0u; }
}

// Workgroup = 4 wavefronts = 4 consecutive i tiles of one replica sharing one stream of j atoms: the j
// positions/parameters are staged through LDS in chunks of NB_CHUNK atoms with coalesced 16-byte loads and
// read back with wave-uniform (broadcast) ds_read_b128, so the inner loop never waits on global memory.
#define NB_CHUNK 128
#define NB_WAVES 4
template <int METHOD, bool ENERGY, bool ALCH>
__global__ __launch_bounds__(64 * NB_WAVES)
void nonbonded_kernel(nb_params p, int N, int Npad, const float4* __restrict__ pos,
                      const float4* __restrict__ param_all, const unsigned long long* __restrict__ mask_all,
                      const int* __restrict__ order, const float4* __restrict__ tile_c, const float4* __restrict__ tile_h,
                      const float* __restrict__ box, const float* __restrict__ rep_lam,
                      float4* __restrict__ partial, double* __restrict__ epart, int n_epart, int ntile)
{
    __shared__ float4 s_pos[NB_CHUNK];
    __shared__ float4 s_prm[NB_CHUNK];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int itile = blockIdx.x * NB_WAVES + wave;
    const int i0 = itile * 64;
    const int i = i0 + lane;
    const int js = blockIdx.y;
    const int r = blockIdx.z;
    // pos / param / mask are per-replica arrays in sorted (spatially ordered) slot space when order != nullptr
    const float4* __restrict__ P = pos + (size_t)r * Npad;
    const float4* __restrict__ param = order ? param_all + (size_t)r * Npad : param_all;
    const unsigned long long* __restrict__ mask = order ? mask_all + (size_t)r * Npad * p.excl_words : mask_all;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const float iLx = 1.f / Lx, iLy = 1.f / Ly, iLz = 1.f / Lz;
    float lam_a = 1.f, sc = 0.f, lam_e = 1.f;
    if (ALCH) { lam_a = rep_lam[4 * r]; sc = rep_lam[4 * r + 1]; lam_e = rep_lam[4 * r + 2]; }
    const bool active = (itile < ntile) && (i < N);
    const float4 xi = P[active ? i : 0];
    float4 pi = param[active ? i : 0];
    if (ALCH && pi.w != 0.f) pi.x *= lam_e;
    unsigned long long mk[MAX_EXCL_WORDS];
#pragma unroll
    for (int w = 0; w < MAX_EXCL_WORDS; ++w) mk[w] = (w < p.excl_words) ? mask[(size_t)(active ? i : 0) * p.excl_words + w] : 0ull;
    const int half = 32 * p.excl_words;

    const int tps = (ntile + p.n_jsplit - 1) / p.n_jsplit;
    const int t0 = js * tps, t1 = min(ntile, t0 + tps);
    const int jbeg = t0 * 64, jend = min(t1 * 64, N);
    float4 ci = make_float4(0.f, 0.f, 0.f, 0.f), hi = ci;
    const bool cull = order != nullptr && itile < ntile;
    if (cull) { ci = tile_c[(size_t)r * ntile + itile]; hi = tile_h[(size_t)r * ntile + itile]; }
    float fx = 0.f, fy = 0.f, fz = 0.f;
    double e = 0.0;
    const float rcut2 = SCI_EWALD(METHOD) ? p.rcc2 : p.rc2;
    for (int cb = jbeg; cb < jend; cb += NB_CHUNK) {
        __syncthreads();                                     // previous chunk fully consumed
        {
            const int jj = min(cb + (tid & (NB_CHUNK - 1)), jend - 1);
            if (tid < NB_CHUNK) s_pos[tid] = P[jj]; else s_prm[tid - NB_CHUNK] = param[jj];
        }
        __syncthreads();
        const int cn = min(NB_CHUNK, jend - cb);
        for (int tb = 0; tb < cn; tb += 64) {                // the chunk holds up to two j tiles
            const int jb = cb + tb;
            const int tn = min(64, cn - tb);
            const bool near = (jb + 63 >= i0 - half) && (jb <= i0 + 63 + half);
            if (cull) {
                // tile-level cull: minimum distance between the two bounding boxes (minimum image) beyond the cutoff
                const int jt = jb >> 6;
                const float4 cj = tile_c[(size_t)r * ntile + jt], hj = tile_h[(size_t)r * ntile + jt];
                float bx = cj.x - ci.x, by = cj.y - ci.y, bz = cj.z - ci.z;
                bx -= Lx * rintf(bx * iLx); by -= Ly * rintf(by * iLy); bz -= Lz * rintf(bz * iLz);
                bx = fmaxf(0.f, fabsf(bx) - hi.x - hj.x); by = fmaxf(0.f, fabsf(by) - hi.y - hj.y); bz = fmaxf(0.f, fabsf(bz) - hi.z - hj.z);
                if (bx * bx + by * by + bz * bz > rcut2) continue;
            }
#pragma unroll 4
            for (int u = 0; u < tn; ++u) {
                const int j = jb + u;
                const float4 xj = s_pos[tb + u];             // same address in every lane: LDS broadcast
                float4 pj = s_prm[tb + u];
                float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                dx -= Lx * rintf(dx * iLx); dy -= Ly * rintf(dy * iLy); dz -= Lz * rintf(dz * iLz);
                const float r2 = dx * dx + dy * dy + dz * dz;
                bool in = active && (r2 < rcut2);
                if (near) {
                    const int d = j - i + half;
                    if (d >= 0 && d < 2 * half) in = in && !((mk[d >> 6] >> (d & 63)) & 1ull);
                }
                if (in) {
                    if (ALCH && pj.w != 0.f) pj.x *= lam_e;
                    float fr, ee;
                    pair_interaction<METHOD, ALCH, !ENERGY>(p, r2, pi, pj, lam_a, sc, fr, false, ee);
                    fx += fr * dx; fy += fr * dy; fz += fr * dz;
                    if (ENERGY) e += 0.5 * (double)ee;
                }
            }
        }
    }
    if (active) {
        // every (i tile, j slice) wave owns one row of the partial-force buffer: plain coalesced stores, no atomics;
        // nb_reduce_kernel folds the slices into the fixed-point accumulator in a fixed order
        const int io = order ? order[(size_t)r * Npad + i] : i;      // back to the atom's own slot
        partial[((size_t)r * p.n_jsplit + js) * Npad + io] = make_float4(fx, fy, fz, 0.f);
    }
    if (ENERGY) {
        e = wave_sum(e);
        if (lane == 0 && itile < ntile) epart[(size_t)r * n_epart + EP_NB0 + itile * p.n_jsplit + js] = e;
    }
}

__global__ __launch_bounds__(256)
void nb_reduce_kernel(int N, int Npad, int nsplit, const float4* __restrict__ partial, long long* __restrict__ force)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (i >= N) return;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float4 f = partial[((size_t)r * nsplit + s) * Npad + i];
        fx += f.x; fy += f.y; fz += f.z;
    }
    add_force(force + (size_t)r * 3 * Npad, Npad, i, fx, fy, fz);
}

// 1-4 style exceptions with non-zero parameters: plain Coulomb + LJ, no cutoff, no switch
template <bool ENERGY>
__global__ __launch_bounds__(256)
void exception_kernel(int n, const int* __restrict__ atoms, const float* __restrict__ params,
                      const int* __restrict__ alch, const float* __restrict__ rep_lam, int Npad,
                      const float4* __restrict__ pos, const float* __restrict__ box,
                      long long* __restrict__ force, double* __restrict__ epart, int n_epart)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int i = atoms[2 * t], j = atoms[2 * t + 1];
        float qq = params[3 * t];
        const float sig = params[3 * t + 1], eps = params[3 * t + 2];
        if (rep_lam && alch[t] > 0) qq *= rep_lam[4 * r + 2];
        float3 d = sub3(ld3(P, j), ld3(P, i));
        if (Lx > 0.f) { d.x -= Lx * rintf(d.x / Lx); d.y -= Ly * rintf(d.y / Ly); d.z -= Lz * rintf(d.z / Lz); }
        const float r2 = dotf(d, d);
        const float inv_r = rsqrtf(r2);
        float Ulj, dUlj;
        if (rep_lam && alch[t] == 1 && eps != 0.f) {             // soft-core: listed_terms.h
            const float2 sc = softcore_exception(rep_lam[4 * r], rep_lam[4 * r + 1], sig, eps, r2, inv_r);
            Ulj = sc.x; dUlj = sc.y;
        } else {
            const float s2 = sig * sig * inv_r * inv_r, s6 = s2 * s2 * s2;
            Ulj = 4.f * eps * s6 * (s6 - 1.f); dUlj = 4.f * eps * s6 * (6.f - 12.f * s6) * inv_r;
        }
        const float U = Ulj + qq * inv_r;
        const float dUdr = dUlj - qq * inv_r * inv_r;
        const float fr = dUdr * inv_r;
        add_force(F, Npad, i, fr * d.x, fr * d.y, fr * d.z);
        add_force(F, Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
        if (ENERGY) e += (double)U;
    }
    if (ENERGY) { e = block_sum_256(e, s_part); if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_EXCEPT] = e; }
}

// Ewald correction for excluded pairs: the reciprocal sum contains them, so subtract qq erf(alpha r)/r
template <bool ENERGY>
__global__ __launch_bounds__(256)
void ewald_exclusion_kernel(int n, const int* __restrict__ atoms, const float* __restrict__ qq_arr,
                            const int* __restrict__ alch, const float* __restrict__ rep_lam, float alpha,
                            float two_alpha_sqrtpi, int Npad, const float4* __restrict__ pos, const float* __restrict__ box,
                            long long* __restrict__ force, double* __restrict__ epart, int n_epart)
{
    __shared__ double s_part[4];
    const int r = blockIdx.x;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int i = atoms[2 * t], j = atoms[2 * t + 1];
        float qq = qq_arr[t];
        if (rep_lam) { const float le = rep_lam[4 * r + 2]; const int na = alch[t]; qq *= (na == 2) ? le * le : (na == 1) ? le : 1.f; }
        float3 d = sub3(ld3(P, j), ld3(P, i));
        d.x -= Lx * rintf(d.x / Lx); d.y -= Ly * rintf(d.y / Ly); d.z -= Lz * rintf(d.z / Lz);
        const float r2 = dotf(d, d);
        const float inv_r = rsqrtf(r2);
        const float rr = r2 * inv_r;
        const float ar = alpha * rr;
        const float erf_ar = erff(ar);
        const float U = -qq * erf_ar * inv_r;
        const float dUdr = -qq * (two_alpha_sqrtpi * __expf(-ar * ar) * inv_r - erf_ar * inv_r * inv_r);
        const float fr = dUdr * inv_r;
        add_force(F, Npad, i, fr * d.x, fr * d.y, fr * d.z);
        add_force(F, Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
        if (ENERGY) e += (double)U;
    }
    if (ENERGY) { e = block_sum_256(e, s_part); if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_EXCLCORR] = e; }
}

// per-replica constants: dispersion correction, Ewald self energy, neutralising background.  Alchemical charges
// scale with lambda_electrostatics (exact PME treatment, alchemy.py:1897-1899): self = nn + l^2 aa, Q = Qn + l Qa.
__global__ void const_energy_kernel(int R, double disp_coeff, double self_nn, double self_aa, double q_n, double q_a,
                                    double plasma_pref, const float* __restrict__ rep_lam,
                                    const float* __restrict__ box, double* __restrict__ epart, int n_epart)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
    const double le = rep_lam ? (double)rep_lam[4 * r + 2] : 1.0;
    const double Q = q_n + le * q_a;
    double e = self_nn + le * le * self_aa;
    if (V > 0) e += (disp_coeff + plasma_pref * Q * Q) / V;
    epart[(size_t)r * n_epart + EP_CONST] = e;
}

// alchemical/non-alchemical soft-core pair energies for every state lambda: alch[r][k]
__global__ __launch_bounds__(256)
void alch_ukl_kernel(nb_params p, int N, int Npad, int n_alch, const int* __restrict__ alch_atoms,
                     const float4* __restrict__ pos, const float4* __restrict__ param, const float* __restrict__ box,
                     int K, const double* __restrict__ state_lam /*[K][2]*/, double* __restrict__ out /*[R][K]*/,
                     int n_exc, const int* __restrict__ exc_atoms, const float* __restrict__ exc_params, const int* __restrict__ exc_alch,
                     const unsigned long long* __restrict__ mask /*[Npad][excl_words], atom order*/)
{
    __shared__ double s_part[4];
    const int k = blockIdx.x, r = blockIdx.y;
    const float4* P = pos + (size_t)r * Npad;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const float lam_a = (float)state_lam[2 * k], sc = (float)state_lam[2 * k + 1];
    double e = 0.0;
    const int total = n_alch * N;
    for (int t = threadIdx.x; t < total; t += 256) {
        const int a = alch_atoms[t / N], j = t % N;
        const float4 pj = param[j];
        // alchemical/alchemical pairs are lambda-controlled only under annihilate_sterics (w = 2); each of them once
        if (pj.w != 0.f && (pj.w < 1.5f || j <= a)) continue;
        // excluded pairs (a region that cuts a molecule has bonded neighbours on both sides; their exceptions follow below)
        const int dd = j - a + 32 * p.excl_words;
        if (dd >= 0 && dd < 64 * p.excl_words && ((mask[(size_t)a * p.excl_words + (dd >> 6)] >> (dd & 63)) & 1ull)) continue;
        const float4 pa = param[a];
        const float4 xa = P[a], xj = P[j];
        float dx = xj.x - xa.x, dy = xj.y - xa.y, dz = xj.z - xa.z;
        dx -= Lx * rintf(dx / Lx); dy -= Ly * rintf(dy / Ly); dz -= Lz * rintf(dz / Lz);
        const float r2 = dx * dx + dy * dy + dz * dz;
        if (r2 >= p.rc2) continue;
        const float sig = pa.y + pj.y, eps4 = pa.z * pj.z;
        if (eps4 == 0.f) continue;
        const float inv_r = rsqrtf(r2), rr = r2 * inv_r;
        const float is2 = 1.f / (sig * sig);
        const float tt = r2 * r2 * r2 * is2 * is2 * is2;
        const float x = 1.f / (sc + tt);
        float U = lam_a * eps4 * x * (x - 1.f), dU = 0.f;
        switch_fn(p, rr, U, dU);
        e += (double)U;
    }
    // exceptions with one alchemical atom: the same soft-core, no cutoff, no switch (listed_terms.h)
    for (int t = threadIdx.x; t < n_exc; t += 256) {
        const float eps = exc_params[3 * t + 2];
        if (exc_alch[t] != 1 || eps == 0.f) continue;
        const float4 xa = P[exc_atoms[2 * t]], xj = P[exc_atoms[2 * t + 1]];
        float dx = xj.x - xa.x, dy = xj.y - xa.y, dz = xj.z - xa.z;
        if (Lx > 0.f) { dx -= Lx * rintf(dx / Lx); dy -= Ly * rintf(dy / Ly); dz -= Lz * rintf(dz / Lz); }
        const float r2 = dx * dx + dy * dy + dz * dz;
        e += (double)softcore_exception(lam_a, sc, exc_params[3 * t + 1], eps, r2, rsqrtf(r2)).x;
    }
    e = block_sum_256(e, s_part);
    if (threadIdx.x == 0) out[(size_t)r * K + k] = e;
}

// sums the partial slots of each replica in fixed order: lane-strided, then xor-shuffle tree
__global__ __launch_bounds__(64)
void reduce_energy_kernel(int n_epart, const double* __restrict__ epart, double* __restrict__ potential)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n_epart; t += 64) e += epart[(size_t)r * n_epart + t];
    e = wave_sum(e);
    if (threadIdx.x == 0) potential[r] = e;
}

// u_kl rows (states.py:1908-1917 with pressure=None; paralleltempering.py:206-215):
//   u[r][l] = beta_l * (U_r + E_alch[r][l] - E_alch[r][own_r] + econst_l)
//   U_r is the potential at the replica's OWN state (round 4: it used to leave the lambda_sterics-controlled pairs out, which
//   made it useless as "the potential energy of the replica": the barostat's and GHMC's Metropolis tests and the work
//   accumulators difference it); E_alch[r][l] are those pairs re-evaluated at every state's lambda_sterics.
//   NPT states add p_l V_r (states.py:1913-1914)
__global__ void assemble_ukl_kernel(int R, int K, const double* __restrict__ potential,
                                    const double* __restrict__ beta, const double* __restrict__ econst,
                                    const double* __restrict__ alch /*[R][K] or null*/, const int* __restrict__ own /*[R]*/,
                                    const double* __restrict__ pressure /*[K] or null*/, const float* __restrict__ box,
                                    double econst_vref, double* __restrict__ ukl_rows)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * K) return;
    const int r = t / K, l = t % K;
    const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
    // the per-state constants are long-range corrections ~ 1/V quoted at the volume econst_vref (0: volume independent)
    double U = potential[r] + econst[l] * ((econst_vref > 0.0 && V > 0.0) ? econst_vref / V : 1.0);
    if (alch) U += alch[t] - alch[(size_t)r * K + own[r]];
    if (pressure) U += pressure[l] * V;
    ukl_rows[t] = beta[l] * U;
}

// ---------------------------------------------------------------------------------------------------
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

void remd_free_nonbonded(remd_ctx* h)
{
    nb_tables* it = g_nb.find(h);
    if (!it) return;
    nb_tables& t = *it;
    dfree(t.d_param); dfree(t.d_mask); dfree(t.d_exc_atoms); dfree(t.d_exc_params); dfree(t.d_excl_atoms); dfree(t.d_excl_qq);
    dfree(t.d_exc_alch); dfree(t.d_excl_alch); dfree(t.d_probe);
    dfree(t.d_rep_lam); dfree(t.d_state_lam); dfree(t.d_alch_ukl); dfree(t.d_own);
    dfree(t.d_grp_first); dfree(t.d_grp_size); dfree(t.d_order); dfree(t.d_spos); dfree(t.d_sposi); dfree(t.d_lj_sposi); dfree(t.d_sparam); dfree(t.d_smask);
    dfree(t.d_tile_c); dfree(t.d_tile_h); dfree(t.d_partial); dfree(t.d_cl_c); dfree(t.d_cl_h);
    dfree(t.d_lj_ord); dfree(t.d_lj_mask); dfree(t.d_lj_order); dfree(t.d_lj_spos); dfree(t.d_lj_sparam); dfree(t.d_lj_smask);
    dfree(t.d_lj_tile_c); dfree(t.d_lj_tile_h); dfree(t.d_lj_cl_c); dfree(t.d_lj_cl_h);
    dfree(t.d_sci_list); dfree(t.d_sci_count); dfree(t.d_excl); dfree(t.d_lj_sci_list); dfree(t.d_lj_sci_count); dfree(t.d_lj_excl);
    dfree(t.d_tile_of_rank);
    dfree(t.d_sforce); dfree(t.d_lj_sforce); dfree(t.d_queue); dfree(t.d_ctab); dfree(t.d_pair_done); dfree(t.d_sort_scratch);
    for (auto& sg : t.tune_segs) { if (sg.a) hipEventDestroy(sg.a); if (sg.b) hipEventDestroy(sg.b); }
    g_nb.erase(h);
}

// long-range dispersion correction coefficient (E = coeff / V), OpenMM NonbondedForce convention:
// averages over the N(N+1)/2 multiset of particle pairs (self pairs included), switching-region
// integral included.  `eps` already has alchemical atoms zeroed when the system is alchemical.
static double dispersion_coefficient(int N, const std::vector<double>& sigma, const std::vector<double>& eps,
                                     double rc, double rs)
{
    std::map<std::pair<double, double>, long long> classes;
    for (int i = 0; i < N; ++i) classes[{sigma[i], eps[i]}]++;
    std::vector<std::pair<std::pair<double, double>, long long>> cl(classes.begin(), classes.end());
    auto integral_switch = [&](double sig) {
        // energy removed by the switch: int_{rs}^{rc} (1 - S(r)) (sig^12/r^12 - sig^6/r^6) r^2 dr, composite Simpson (f64)
        if (!(rs >= 0) || rs >= rc) return 0.0;
        const int n = 4000; const double hstep = (rc - rs) / n;
        auto f = [&](double r) {
            const double x = (r - rs) / (rc - rs);
            const double S = 1.0 + x * x * x * (-10.0 + x * (15.0 - 6.0 * x));
            const double s6 = pow(sig / r, 6);
            return (1.0 - S) * (s6 * s6 - s6) * r * r;
        };
        double acc = f(rs) + f(rc);
        for (int k = 1; k < n; ++k) acc += f(rs + k * hstep) * ((k & 1) ? 4.0 : 2.0);
        return acc * hstep / 3.0;
    };
    double sum1 = 0, sum2 = 0, sum3 = 0;
    for (size_t a = 0; a < cl.size(); ++a)
        for (size_t b = a; b < cl.size(); ++b) {
            const double count = (a == b) ? 0.5 * (double)cl[a].second * (double)(cl[a].second + 1)
                                          : (double)cl[a].second * (double)cl[b].second;
            const double sig = 0.5 * (cl[a].first.first + cl[b].first.first);
            const double e = sqrt(cl[a].first.second * cl[b].first.second);
            if (e == 0.0) continue;
            const double s6 = pow(sig, 6);
            sum1 += count * e * s6 * s6;
            sum2 += count * e * s6;
            sum3 += count * e * integral_switch(sig);
        }
    const double npairs = 0.5 * (double)N * (double)(N + 1);
    sum1 /= npairs; sum2 /= npairs; sum3 /= npairs;
    return 8.0 * N * (double)N * M_PI * (sum1 / (9.0 * pow(rc, 9)) - sum2 / (3.0 * pow(rc, 3)) + sum3);
}

// the (term, slot) entries of the listed terms in the order of the atoms they act on (listed_terms.h): class << 29 | slot << 27 | term
static int build_atom_terms(remd_ctx* h, int N, const std::vector<int>& ba, const std::vector<int>& aa, const std::vector<int>& ta,
                            const std::vector<int>& exc, int n_exc, const std::vector<int>& excl, int n_excl)
{
    if (h->d_aterm) { hipFree(h->d_aterm); h->d_aterm = nullptr; }
    h->n_aterm = 0;
    const size_t nterm[5] = { ba.size() / 2, aa.size() / 3, ta.size() / 4, (size_t)n_exc, (size_t)n_excl };
    const std::vector<int>* arr[5] = { &ba, &aa, &ta, &exc, &excl };
    const int width[5] = { 2, 3, 4, 2, 2 };
    for (int c = 0; c < 5; ++c) if (nterm[c] >= (1u << 27)) return 0;          // (the term-per-thread launch stays)
    std::vector<int> start(N + 1, 0);
    for (int c = 0; c < 5; ++c)
        for (size_t t = 0; t < nterm[c]; ++t)
            for (int s = 0; s < width[c]; ++s) start[(*arr[c])[t * width[c] + s] + 1]++;
    for (int i = 0; i < N; ++i) start[i + 1] += start[i];
    if (start[N] == 0) return 0;
    std::vector<unsigned int> ent(start[N]);
    std::vector<int> cur(start.begin(), start.end() - 1);
    for (int c = 0; c < 5; ++c)
        for (size_t t = 0; t < nterm[c]; ++t)
            for (int s = 0; s < width[c]; ++s)
                ent[cur[(*arr[c])[t * width[c] + s]]++] = ((unsigned int)c << 29) | ((unsigned int)s << 27) | (unsigned int)t;
    h->n_aterm = (int)ent.size();
    return upload(h, h->d_aterm, ent);
}

int remd_build_nonbonded(remd_ctx* h, const remd_system_desc* d)
{
    remd_free_nonbonded(h);
    remd_nocutoff_release(h);
    remd_gbsa_release(h);
    // NoCutoff (vacuum systems): the direct sum of nocutoff.hip; everything else treats the handle as one without a cutoff-based nonbonded force
    h->nb_method = d->nb_method == REMD_NB_NOCUTOFF ? REMD_NB_NONE : d->nb_method;
    // bonded tables live in the context
    {
        std::vector<int> ba(d->bond_atoms, d->bond_atoms + 2 * (size_t)d->n_bonds);
        std::vector<float> bp(2 * (size_t)d->n_bonds);
        for (size_t k = 0; k < bp.size(); ++k) bp[k] = (float)d->bond_params[k];
        std::vector<int> aa(d->angle_atoms, d->angle_atoms + 3 * (size_t)d->n_angles);
        std::vector<float> ap(2 * (size_t)d->n_angles);
        for (size_t k = 0; k < ap.size(); ++k) ap[k] = (float)d->angle_params[k];
        std::vector<int> ta(d->torsion_atoms, d->torsion_atoms + 4 * (size_t)d->n_torsions);
        std::vector<float> tp(3 * (size_t)d->n_torsions);
        for (size_t k = 0; k < tp.size(); ++k) tp[k] = (float)d->torsion_params[k];
        int rc;
        if ((rc = upload(h, h->d_bond_atoms, ba)) || (rc = upload(h, h->d_bond_params, bp)) ||
            (rc = upload(h, h->d_angle_atoms, aa)) || (rc = upload(h, h->d_angle_params, ap)) ||
            (rc = upload(h, h->d_torsion_atoms, ta)) || (rc = upload(h, h->d_torsion_params, tp))) return rc;
        h->n_bonds = d->n_bonds; h->n_angles = d->n_angles; h->n_torsions = d->n_torsions;
        for (int k = 0; k < 2 * d->n_bonds; ++k) if (ba[k] < 0 || ba[k] >= d->n_atoms) return remd_fail(h, -3, "bond atom index out of range");
        for (int k = 0; k < 3 * d->n_angles; ++k) if (aa[k] < 0 || aa[k] >= d->n_atoms) return remd_fail(h, -3, "angle atom index out of range");
        for (int k = 0; k < 4 * d->n_torsions; ++k) if (ta[k] < 0 || ta[k] >= d->n_atoms) return remd_fail(h, -3, "torsion atom index out of range");
    }
    h->n_alch = d->n_alch;
    if (d->nb_method == REMD_NB_NONE || d->nb_method == REMD_NB_NOCUTOFF) {
        const std::vector<int> ba(d->bond_atoms, d->bond_atoms + 2 * (size_t)d->n_bonds), aa(d->angle_atoms, d->angle_atoms + 3 * (size_t)d->n_angles),
                               ta(d->torsion_atoms, d->torsion_atoms + 4 * (size_t)d->n_torsions), none;
        int rc0 = build_atom_terms(h, d->n_atoms, ba, aa, ta, none, 0, none, 0);
        if (rc0 || d->nb_method == REMD_NB_NONE) return rc0;
        if (d->n_alch > 0) return remd_fail(h, -3, "NoCutoff: alchemical atoms go through remd_set_alchemical_regions (the descriptor's one-region path needs a cutoff method)");
        return remd_nocutoff_build(h, d);
    }
    if (d->nb_method != REMD_NB_CUTOFF_PERIODIC && d->nb_method != REMD_NB_PME) return remd_fail(h, -3, "unknown nonbonded method");
    if (!d->charge || !d->sigma || !d->epsilon) return remd_fail(h, -1, "nonbonded parameter arrays missing");
    if (!(d->cutoff > 0)) return remd_fail(h, -1, "cutoff must be positive");
    const int N = d->n_atoms;
    nb_tables& t = g_nb[h];
    t.is_alch.assign(N, 0);
    for (int a = 0; a < d->n_alch; ++a) {
        if (d->alch_atoms[a] < 0 || d->alch_atoms[a] >= N) return remd_fail(h, -3, "alchemical atom index out of range");
        t.is_alch[d->alch_atoms[a]] = 1;
    }
    t.has_alch = d->n_alch > 0;
    if (t.has_alch && d->softcore_c != 6.0) return remd_fail(h, -3, "only softcore_c = 6 is implemented");
    h->sc_alpha = d->softcore_alpha; h->sc_a = d->softcore_a; h->sc_b = d->softcore_b; h->sc_c = d->softcore_c;
    {
        std::vector<int> al(d->alch_atoms, d->alch_atoms + d->n_alch);
        int rc = upload(h, h->d_alch_atoms, al); if (rc) return rc;
    }
    bool any_charge = false;
    t.charge.assign(d->charge, d->charge + N);
    for (int i = 0; i < N; ++i) any_charge |= (d->charge[i] != 0.0);
    t.method = !any_charge ? NB_LJ_ONLY : (d->nb_method == REMD_NB_PME ? NB_EWALD : NB_RF);
    if (d->nb_method == REMD_NB_PME && !any_charge) t.method = NB_LJ_ONLY;
    const double sqk = sqrt(REMD_ONE_4PI_EPS0);
    std::vector<float4> prm(h->Npad, make_float4(0, 0, 0, 0));
    for (int i = 0; i < N; ++i)
        prm[i] = make_float4((float)(d->charge[i] * sqk), (float)(0.5 * d->sigma[i]), (float)(2.0 * sqrt(d->epsilon[i])),
                             t.is_alch[i] ? (h->annihilate_sterics ? 2.f : 1.f) : 0.f);
    int rc;
    if ((rc = upload(h, t.d_param, prm))) return rc;
    // exclusion window
    int maxd = 0;
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
        if (i < 0 || j < 0 || i >= N || j >= N || i == j) return remd_fail(h, -3, "bad exception pair");
        maxd = std::max(maxd, std::abs(i - j));
    }
    int words = std::max(1, (maxd + 32) / 32);          // window [-32w, 32w) must contain +-maxd
    if (words > MAX_EXCL_WORDS) return remd_fail(h, -3, "exclusions span more than 255 atom indices (not supported yet)");
    std::vector<unsigned long long> mk((size_t)h->Npad * words, 0ull);
    auto setbit = [&](int i, int j) { const int dd = j - i + 32 * words; mk[(size_t)i * words + (dd >> 6)] |= 1ull << (dd & 63); };
    for (int i = 0; i < N; ++i) setbit(i, i);
    std::vector<int> exc_atoms, excl_atoms, exc_alch, excl_alch; std::vector<float> exc_params, excl_qq;
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
        setbit(i, j); setbit(j, i);
        const double qq = d->exception_params[3 * e], sg = d->exception_params[3 * e + 1], ep = d->exception_params[3 * e + 2];
        if (qq != 0.0 || ep != 0.0) {
            // alchemy.py:1964-1990: electrostatic exceptions touching the region scale with lambda_electrostatics;
            // alchemical/alchemical sterics exceptions stay at full strength (annihilate_sterics=False); a sterics
            // exception between an alchemical and a non-alchemical atom is soft-core (round 4: softcore_exception,
            // listed_terms.h; the factory's CustomBondForce, alchemy.py:1836-1851) -- softcore_c = 6 like the pair kernel.
            // (1 marks the soft-core Lennard-Jones exceptions: one alchemical atom, or two under annihilate_sterics -- the
            // alchemical/alchemical CustomBondForce is lambda-controlled then, alchemy.py:1841-1846; any value > 0 scales the charges)
            exc_alch.push_back((t.is_alch[i] && t.is_alch[j] && h->annihilate_sterics) ? 1 : (int)t.is_alch[i] + (int)t.is_alch[j]);
            exc_atoms.push_back(i); exc_atoms.push_back(j);
            exc_params.push_back((float)(qq * REMD_ONE_4PI_EPS0)); exc_params.push_back((float)sg); exc_params.push_back((float)ep);
        }
        if (d->charge[i] != 0.0 && d->charge[j] != 0.0) {
            excl_alch.push_back((int)t.is_alch[i] + (int)t.is_alch[j]);
            excl_atoms.push_back(i); excl_atoms.push_back(j);
            excl_qq.push_back((float)(d->charge[i] * d->charge[j] * REMD_ONE_4PI_EPS0));
        }
    }
    if ((rc = upload(h, t.d_mask, mk))) return rc;
    t.n_exc = (int)exc_params.size() / 3;
    if ((rc = upload(h, t.d_exc_atoms, exc_atoms)) || (rc = upload(h, t.d_exc_params, exc_params)) || (rc = upload(h, t.d_exc_alch, exc_alch))) return rc;
    t.n_excl = (t.method == NB_EWALD) ? (int)excl_qq.size() : 0;
    if ((rc = upload(h, t.d_excl_atoms, excl_atoms)) || (rc = upload(h, t.d_excl_qq, excl_qq)) || (rc = upload(h, t.d_excl_alch, excl_alch))) return rc;
    {
        const std::vector<int> ba(d->bond_atoms, d->bond_atoms + 2 * (size_t)d->n_bonds), aa(d->angle_atoms, d->angle_atoms + 3 * (size_t)d->n_angles),
                               ta(d->torsion_atoms, d->torsion_atoms + 4 * (size_t)d->n_torsions);
        if ((rc = build_atom_terms(h, N, ba, aa, ta, exc_atoms, t.n_exc, excl_atoms, t.n_excl))) return rc;
    }

    nb_params& p = t.p;
    p.rc = (float)d->cutoff; p.rc2 = (float)(d->cutoff * d->cutoff);
    p.rs = (d->switch_distance > 0 && d->switch_distance < d->cutoff) ? (float)d->switch_distance : -1.f;
    p.inv_sw = p.rs >= 0 ? (float)(1.0 / (d->cutoff - d->switch_distance)) : 0.f;
    const double eps_s = d->rf_dielectric;
    p.krf = (float)((eps_s - 1.0) / (2.0 * eps_s + 1.0) / pow(d->cutoff, 3));
    p.crf = (float)(3.0 * eps_s / (2.0 * eps_s + 1.0) / d->cutoff);
    p.rs_c = -1.f; p.inv_sw_c = 0.f;
    if (h->rf_unshifted) {                  // remd_set_reaction_field: UnshiftedReactionFieldForce (forces.py:1110-1150)
        p.crf = 0.f;
        if (h->rf_switch_width > 0.0 && h->rf_switch_width < d->cutoff) { p.rs_c = (float)(d->cutoff - h->rf_switch_width); p.inv_sw_c = (float)(1.0 / h->rf_switch_width); }
    }
    p.alpha = (float)d->ewald_alpha; p.two_alpha_sqrtpi = (float)(2.0 * d->ewald_alpha / sqrt(M_PI));
    p.excl_words = words;
    // Ewald split (remd_set_coulomb_cutoff): the erfc tail may be summed beyond the NonbondedForce cutoff, with the alpha and
    // the mesh of the descriptor chosen for that range by the host; Lennard-Jones terms keep d->cutoff and its switch
    p.rcc2 = p.rc2;
    if (t.method == NB_EWALD && h->coulomb_cutoff > 0.0) {
        if (h->coulomb_cutoff < d->cutoff) return remd_fail(h, -1, "the Coulomb cutoff of remd_set_coulomb_cutoff is shorter than the NonbondedForce cutoff");
        p.rcc2 = (float)(h->coulomb_cutoff * h->coulomb_cutoff);
    }
    p.ctab_key0 = 0; p.ctab_n = 0; p.ctab_umin = 0.f;
    t.use_table = false;
    if (t.method == NB_EWALD && !(getenv("REMD_NB_TABLE") && atoi(getenv("REMD_NB_TABLE")) == 0)) {
        const coulomb_table_host T = ctab_build(d->ewald_alpha, (double)p.rcc2);
        std::vector<float4> tab(T.n);
        for (int k = 0; k < T.n; ++k) tab[k] = make_float4(T.c[4 * k], T.c[4 * k + 1], T.c[4 * k + 2], T.c[4 * k + 3]);
        if ((rc = upload(h, t.d_ctab, tab))) return rc;
        p.ctab_key0 = T.key0; p.ctab_n = T.n; p.ctab_umin = T.umin;
        t.use_table = true;
    }
    const int ntile = (N + 63) / 64;
    {
        p.n_jsplit = std::max(1, std::min(std::min(ntile, 16), 4));
    }
    h->cutoff = std::max(d->cutoff, (double)sqrtf(p.rcc2)); h->switch_dist = d->switch_distance; h->ewald_alpha = d->ewald_alpha;
    for (int k = 0; k < 3; ++k) h->grid[k] = d->pme_grid[k];

    // groups: connected components of exclusions + constraints, made contiguous in index space
    {
        std::vector<int> parent(N);
        for (int i = 0; i < N; ++i) parent[i] = i;
        auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
        auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) parent[std::max(a, b)] = std::min(a, b); };
        for (int e = 0; e < d->n_exceptions; ++e) unite(d->exception_atoms[2 * e], d->exception_atoms[2 * e + 1]);
        for (int w = 0; w < d->n_settle; ++w) { unite(d->settle_atoms[3 * w], d->settle_atoms[3 * w + 1]); unite(d->settle_atoms[3 * w], d->settle_atoms[3 * w + 2]); }
        for (int c = 0; c < d->n_shake; ++c) for (int k = 1; k < 4; ++k) if (d->shake_atoms[4 * c + k] >= 0) unite(d->shake_atoms[4 * c], d->shake_atoms[4 * c + k]);
        // the root is the smallest index of a component; extend every component to the span [root, max member]
        std::vector<int> hi(N);
        for (int i = 0; i < N; ++i) hi[i] = i;
        for (int i = 0; i < N; ++i) { const int rt = find(i); hi[rt] = std::max(hi[rt], i); }
        std::vector<int> first, size;
        int i = 0;
        while (i < N) {
            int end = hi[find(i)];
            for (int k = i; k <= end; ++k) end = std::max(end, hi[find(k)]);   // merge interleaved components
            first.push_back(i); size.push_back(end - i + 1);
            i = end + 1;
        }
        t.n_groups = (int)first.size();
        if ((rc = upload(h, t.d_grp_first, first)) || (rc = upload(h, t.d_grp_size, size))) return rc;
        t.sorting = true;
        t.clusters = !getenv("REMD_NB_TILES");        // test hook: the 64-atom tile kernel a list overflow falls back to
        // 8 slices where the mesh stream runs beside the pair kernel (its launch may become a resident set pulling items: 105.9 vs
        // 106.55 ms per 500 steps on the headline config), 12 elsewhere (LJ fluid, one workgroup per item: 66.6 vs 63.3 it/s).
        // Fixed per system, never per launch mode: the slices' fp32 partial sums enter the forces bit-wise.
        t.sci_split = (t.method == NB_EWALD && h->overlap && h->stream2) ? 8 : 12;
        // (round 6, last scan: systems of 64 tiles or more have work items enough without the finer slicing -- 4 slices: 8 x CB7:B2 (71 tiles)
        // 13.81 -> 14.65 it/s, 16 x DHFR (369) 2.25 -> 2.31, DHFR on an alchemical ladder 1.92 -> 2.04; the headline system (36 tiles) loses
        // 10 % with 4; profiles/r06_45.  A property of the system, not of the replica count: blocks and one-block runs stay bit-identical.)
        // (the plain kernel on CB7:B2 -- the general alchemical path, whose pair kernels see the environment only -- loses 2 % with 4: 13.3 -> 13.0;
        // so from 64 tiles on for the soft-core variant, from 256 on for the plain one)
        if (t.method == NB_EWALD && h->overlap && h->stream2 && (h->N + 63) / 64 >= (d->n_alch > 0 ? 64 : 256)) t.sci_split = 4;
        if (getenv("REMD_NB_SPLIT")) t.sci_split = std::max(4, atoi(getenv("REMD_NB_SPLIT")));      // experiment hook
        // the ranking of the sort is G comparisons per molecule on ONE workgroup per replica (0.9 ms on DHFR's 7 k molecules): from 2048
        // molecules on it runs every 160 evaluations instead of every 40 (the order decays slowly: 0.893 -> 0.868 ms per step on 16 x DHFR,
        // flat between 80 and 320, profiles/r06_36_resort_interval.txt; the headline system does not notice either way)
        if (t.n_groups >= 2048) t.resort_interval = 160;
        if (getenv("REMD_NB_RESORT")) t.resort_interval = std::max(1, atoi(getenv("REMD_NB_RESORT")));
    }

    // LJ-active sub-system: worthwhile when charges exist and most atoms carry no LJ (TIP3P hydrogens)
    {
        std::vector<int> ord(h->Npad, -1);
        int NL = 0;
        for (int i = 0; i < N; ++i) if (d->epsilon[i] != 0.0) ord[i] = NL++;
        t.lj_split = any_charge && NL > 0 && NL * 10 <= N * 6;
        if (t.lj_split) {
            t.NL = NL; t.NLpad = (NL + 63) / 64 * 64;
            int maxd_lj = 0;
            for (int e = 0; e < d->n_exceptions; ++e) {
                const int a = ord[d->exception_atoms[2 * e]], b2 = ord[d->exception_atoms[2 * e + 1]];
                if (a >= 0 && b2 >= 0) maxd_lj = std::max(maxd_lj, std::abs(a - b2));
            }
            t.lj_words = std::max(1, (maxd_lj + 32) / 32);
            if (t.lj_words > MAX_EXCL_WORDS) t.lj_split = false;
        }
        if (t.lj_split) {
            std::vector<unsigned long long> lm((size_t)t.NLpad * t.lj_words, 0ull);
            auto setb = [&](int a, int b2) { const int dd = b2 - a + 32 * t.lj_words; lm[(size_t)a * t.lj_words + (dd >> 6)] |= 1ull << (dd & 63); };
            for (int a = 0; a < t.NL; ++a) setb(a, a);
            for (int e = 0; e < d->n_exceptions; ++e) {
                const int a = ord[d->exception_atoms[2 * e]], b2 = ord[d->exception_atoms[2 * e + 1]];
                if (a >= 0 && b2 >= 0) { setb(a, b2); setb(b2, a); }
            }
            if ((rc = upload(h, t.d_lj_ord, ord)) || (rc = upload(h, t.d_lj_mask, lm))) return rc;
        }
    }

    // dispersion correction of the (possibly alchemically modified) NonbondedForce
    t.disp_coeff = 0.0;
    if (d->use_dispersion_correction) {
        std::vector<double> sg(d->sigma, d->sigma + N), ep(d->epsilon, d->epsilon + N);
        for (int i = 0; i < N; ++i) if (t.is_alch[i]) ep[i] = 0.0;     // alchemy.py: alchemical atoms carry eps = 0 in the NonbondedForce
        t.disp_coeff = dispersion_coefficient(N, sg, ep, d->cutoff, p.rs >= 0 ? d->switch_distance : -1.0);
    }
    t.self_energy = 0.0; t.net_charge_term = 0.0; t.self_nn = t.self_aa = t.q_n = t.q_a = 0.0;
    if (t.method == NB_EWALD) {
        double q2n = 0, q2a = 0;
        for (int i = 0; i < N; ++i) {
            if (t.is_alch[i]) { q2a += d->charge[i] * d->charge[i]; t.q_a += d->charge[i]; }
            else { q2n += d->charge[i] * d->charge[i]; t.q_n += d->charge[i]; }
        }
        const double pref = -REMD_ONE_4PI_EPS0 * d->ewald_alpha / sqrt(M_PI);
        t.self_nn = pref * q2n; t.self_aa = pref * q2a;
        t.self_energy = t.self_nn + t.self_aa;
        t.net_charge_term = -REMD_ONE_4PI_EPS0 * M_PI / (2.0 * d->ewald_alpha * d->ewald_alpha);   // times Q^2 / V
    }
    return 0;
}

// per-replica lambda parameters follow the replica's current state label
static int update_replica_lambdas(remd_ctx* h, nb_tables& t)
{
    if (!t.has_alch) return 0;
    if (t.rep_lam_R != h->R) { dfree(t.d_rep_lam); REMD_CHECK(h, hipMalloc(&t.d_rep_lam, sizeof(float) * 4 * h->R)); t.rep_lam_R = h->R; t.rep_lam_host.clear(); }
    std::vector<float> rl(4 * (size_t)h->R, 0.f);
    for (int r = 0; r < h->R; ++r) {
        const int64_t k = h->labels.empty() ? 0 : h->labels[h->r_begin + r];
        const double ls = h->lam_s.empty() ? 1.0 : h->lam_s[k];
        const double le = t.lam_e_override >= 0.0 ? t.lam_e_override : (h->lam_e.empty() ? 1.0 : h->lam_e[k]);
        rl[4 * r] = (float)pow(ls, h->sc_a);
        rl[4 * r + 1] = (float)(h->sc_alpha * pow(1.0 - ls, h->sc_b));
        rl[4 * r + 2] = (float)le;
    }
    // uploaded only when something changed (labels after a mix, a lambda override of the u_kl passes): the steady state of the
    // MD loop has no host copy and no synchronisation per force evaluation
    if (rl == t.rep_lam_host) return 0;
    REMD_CHECK(h, hipMemcpyAsync(t.d_rep_lam, rl.data(), sizeof(float) * rl.size(), hipMemcpyHostToDevice, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    t.rep_lam_host = rl;
    return 0;
}

// per-tile union lists of one (sub-)system
static void launch_list_build(remd_ctx* h, nb_tables& t, bool lj)
{
    const int ncl = lj ? t.NLpad / 8 : ((h->N + 63) / 64) * 8;
    hipLaunchKernelGGL(build_sci_list_kernel, dim3(ncl / 8, h->R), dim3(64), 0, h->stream, ncl, lj ? t.lj_cap : t.cl_cap,
                       (!lj && t.method == NB_EWALD) ? t.p.rcc2 : t.p.rc2,
                       lj ? t.d_lj_cl_c : t.d_cl_c, lj ? t.d_lj_cl_h : t.d_cl_h, lj ? t.d_lj_tile_c : t.d_tile_c, lj ? t.d_lj_tile_h : t.d_tile_h,
                       h->d_box, lj ? t.d_lj_sci_list : t.d_sci_list, lj ? t.d_lj_sci_count : t.d_sci_count);
}

// sorted order (refreshed every resort_interval evaluations), sorted positions + bounding boxes and the super-cluster
// lists (every evaluation) of the cluster-pair path
static int ensure_sorted(remd_ctx* h, nb_tables& t)
{
    if (!t.sorting || t.n_groups <= 0) return 0;
    const int ntile = (h->N + 63) / 64;
    const bool cl = t.clusters && ntile * 8 < 65536;
    if (t.sort_R != h->R) {
        dfree(t.d_order); dfree(t.d_spos); dfree(t.d_sparam); dfree(t.d_smask); dfree(t.d_tile_c); dfree(t.d_tile_h);
        const size_t n = (size_t)h->R * h->Npad;
        REMD_CHECK(h, hipMalloc(&t.d_order, sizeof(int) * n));
        REMD_CHECK(h, hipMalloc(&t.d_spos, sizeof(float4) * n));
        dfree(t.d_sposi);
        REMD_CHECK(h, hipMalloc(&t.d_sposi, sizeof(float4) * n));
        REMD_CHECK(h, hipMalloc(&t.d_sparam, sizeof(float4) * n));
        REMD_CHECK(h, hipMalloc(&t.d_smask, sizeof(unsigned long long) * n * t.p.excl_words));
        REMD_CHECK(h, hipMalloc(&t.d_tile_c, sizeof(float4) * (size_t)h->R * ntile));
        REMD_CHECK(h, hipMalloc(&t.d_tile_h, sizeof(float4) * (size_t)h->R * ntile));
        dfree(t.d_pair_done);
        REMD_CHECK(h, hipMalloc(&t.d_pair_done, sizeof(unsigned int) * 16 * h->R));       // one arrival counter per replica, 64 bytes apart
        REMD_CHECK(h, hipMemsetAsync(t.d_pair_done, 0, sizeof(unsigned int) * 16 * h->R, h->stream));
        t.pair_done_target = 0;
        dfree(t.d_cl_c); dfree(t.d_cl_h);
        dfree(t.d_sci_list); dfree(t.d_sci_count); dfree(t.d_excl); dfree(t.d_sforce); dfree(t.d_lj_sforce);
        dfree(t.d_lj_sci_list); dfree(t.d_lj_sci_count); dfree(t.d_lj_excl);
        const int ncl = ntile * 8;
        t.cl_cap = std::min(ncl, 4096);     // whole row for systems up to 32k atoms: a cluster that straddles a large molecule can neighbour half the box
        if (cl) {
            REMD_CHECK(h, hipMalloc(&t.d_cl_c, sizeof(float4) * (size_t)h->R * ncl));
            REMD_CHECK(h, hipMalloc(&t.d_cl_h, sizeof(float4) * (size_t)h->R * ncl));
            t.excl_W = 4 * t.p.excl_words + 1;
            REMD_CHECK(h, hipMalloc(&t.d_sci_list, sizeof(unsigned int) * (size_t)h->R * ntile * t.cl_cap));
            REMD_CHECK(h, hipMalloc(&t.d_sci_count, sizeof(int) * (size_t)h->R * ntile));
            dfree(t.d_tile_of_rank);
            REMD_CHECK(h, hipMalloc(&t.d_tile_of_rank, sizeof(int) * (size_t)h->R * ntile));
            REMD_CHECK(h, hipMalloc(&t.d_excl, sizeof(unsigned long long) * (size_t)h->R * ncl * t.excl_W));
            REMD_CHECK(h, hipMalloc(&t.d_sforce, sizeof(long long) * n * 3));
            REMD_CHECK(h, hipMemsetAsync(t.d_sforce, 0, sizeof(long long) * n * 3, h->stream));
            if (!t.d_queue) {
                REMD_CHECK(h, hipMalloc(&t.d_queue, 4 * sizeof(unsigned int)));
                REMD_CHECK(h, hipMemsetAsync(t.d_queue, 0, 4 * sizeof(unsigned int), h->stream));
            }
        }
        if (cl && t.lj_split) {
            dfree(t.d_lj_order); dfree(t.d_lj_spos); dfree(t.d_lj_sparam); dfree(t.d_lj_smask); dfree(t.d_lj_tile_c); dfree(t.d_lj_tile_h);
            dfree(t.d_lj_cl_c); dfree(t.d_lj_cl_h);
            const size_t nl = (size_t)h->R * t.NLpad;
            const int ncl_lj = t.NLpad / 8;
            t.lj_cap = std::min(ncl_lj, 4096);
            REMD_CHECK(h, hipMalloc(&t.d_lj_order, sizeof(int) * nl));
            REMD_CHECK(h, hipMalloc(&t.d_lj_spos, sizeof(float4) * nl));
            dfree(t.d_lj_sposi);
            REMD_CHECK(h, hipMalloc(&t.d_lj_sposi, sizeof(float4) * nl));
            REMD_CHECK(h, hipMalloc(&t.d_lj_sparam, sizeof(float4) * nl));
            REMD_CHECK(h, hipMalloc(&t.d_lj_smask, sizeof(unsigned long long) * nl * t.lj_words));
            REMD_CHECK(h, hipMalloc(&t.d_lj_tile_c, sizeof(float4) * (size_t)h->R * (t.NLpad / 64)));
            REMD_CHECK(h, hipMalloc(&t.d_lj_tile_h, sizeof(float4) * (size_t)h->R * (t.NLpad / 64)));
            REMD_CHECK(h, hipMalloc(&t.d_lj_cl_c, sizeof(float4) * (size_t)h->R * ncl_lj));
            REMD_CHECK(h, hipMalloc(&t.d_lj_cl_h, sizeof(float4) * (size_t)h->R * ncl_lj));
            t.lj_excl_W = 4 * t.lj_words + 1;
            REMD_CHECK(h, hipMalloc(&t.d_lj_sci_list, sizeof(unsigned int) * (size_t)h->R * (ncl_lj / 8) * t.lj_cap));
            REMD_CHECK(h, hipMalloc(&t.d_lj_sci_count, sizeof(int) * (size_t)h->R * (ncl_lj / 8)));
            REMD_CHECK(h, hipMalloc(&t.d_lj_excl, sizeof(unsigned long long) * (size_t)h->R * ncl_lj * t.lj_excl_W));
            REMD_CHECK(h, hipMalloc(&t.d_lj_sforce, sizeof(long long) * nl * 3));
            REMD_CHECK(h, hipMemsetAsync(t.d_lj_sforce, 0, sizeof(long long) * nl * 3, h->stream));
        }
        t.sort_R = h->R; t.evals_since_sort = 1 << 30;
    }
    const bool split = cl && t.lj_split;
    if (t.evals_since_sort >= t.resort_interval) {
        remd_prof_scope ps(h, "nb_sort");
        // cells of the molecule order: 2^b per box edge with an edge of ~0.11 nm (b from the longest edge of the first replica's box, as
        // the host mirrors it; 32 per edge on the headline system, 64 on DHFR: REMD_NB_HBITS scans, profiles/r06_35_hilbert_cells.txt), along the Hilbert curve.  REMD_NB_CURVE=morton: the
        // Z-order curve on cells of 0.45 nm, as until round 6.  A property of the handle: fixed at the first sort.
        if (t.sort_hbits == 0 && !(getenv("REMD_NB_CURVE") && getenv("REMD_NB_CURVE")[0] == 'm')) {
            double lmax = 0.0;
            for (int k = 0; k < 3 && (size_t)k < h->box_host.size(); ++k) lmax = std::max(lmax, h->box_host[k]);
            t.sort_hbits = lmax > 0.0 ? std::max(1, std::min(6, (int)lround(log2(lmax / 0.11)))) : 4;
            if (getenv("REMD_NB_HBITS")) t.sort_hbits = std::max(1, std::min(6, atoi(getenv("REMD_NB_HBITS"))));      // experiment hook
        }
        if (t.n_groups < 8192) {
            const size_t lds = sizeof(int) * 4 * (size_t)t.n_groups;
            hipLaunchKernelGGL(sort_groups_kernel, dim3(h->R), dim3(1024), lds, h->stream, t.n_groups, h->N, h->Npad, t.d_grp_first,
                               t.d_grp_size, h->d_pos, h->d_box, t.sort_cell, t.d_order, t.sort_hbits);
        } else {
            const size_t need = (size_t)h->R * 5 * t.n_groups;
            if (t.sort_scratch_n < need) {
                dfree(t.d_sort_scratch);
                REMD_CHECK(h, hipMalloc(&t.d_sort_scratch, sizeof(int) * need));
                t.sort_scratch_n = need;
            }
            hipLaunchKernelGGL(sort_groups_large_kernel, dim3(h->R), dim3(1024), 0, h->stream, t.n_groups, h->N, h->Npad, t.d_grp_first,
                               t.d_grp_size, h->d_pos, h->d_box, t.sort_cell, t.d_order, t.d_sort_scratch, t.sort_hbits);
        }
        hipLaunchKernelGGL(gather_params_kernel, dim3((h->Npad + 255) / 256, h->R), dim3(256), 0, h->stream, h->Npad, t.p.excl_words,
                           t.d_order, t.d_param, t.d_mask, t.d_sparam, t.d_smask);
        if (split)
            hipLaunchKernelGGL(compact_lj_kernel, dim3(h->R), dim3(1024), 0, h->stream, h->N, h->Npad, t.NL, t.NLpad, t.lj_words, t.d_order,
                               t.d_lj_ord, t.d_param, t.d_lj_mask, t.d_lj_order, t.d_lj_sparam, t.d_lj_smask);
        if (cl) {
            hipLaunchKernelGGL(build_excl_kernel, dim3(ntile * 8, h->R), dim3(64), 0, h->stream, h->Npad, ntile * 8, t.excl_W, t.p.excl_words,
                               t.d_smask, t.d_excl);
            if (split)
                hipLaunchKernelGGL(build_excl_kernel, dim3(t.NLpad / 8, h->R), dim3(64), 0, h->stream, t.NLpad, t.NLpad / 8, t.lj_excl_W,
                                   t.lj_words, t.d_lj_smask, t.d_lj_excl);
        }
        t.evals_since_sort = 0;
    }
    t.evals_since_sort++;
    remd_prof_scope ps(h, "nb_gather");
    if (split) {
        // main system + LJ sub-system in one gather launch and one list launch
        const int ntile_lj = t.NLpad / 64;
        gather_args ga{h->Npad, t.d_order, t.d_spos, t.d_tile_c, t.d_tile_h, t.d_cl_c, t.d_cl_h, t.d_sparam, t.d_sposi};
        gather_args gb{t.NLpad, t.d_lj_order, t.d_lj_spos, t.d_lj_tile_c, t.d_lj_tile_h, t.d_lj_cl_c, t.d_lj_cl_h, nullptr, t.d_lj_sposi};
        sci_list_args la{ntile * 8, t.cl_cap, t.d_cl_c, t.d_cl_h, t.d_tile_c, t.d_tile_h, t.d_sci_list, t.d_sci_count};
        sci_list_args lb{t.NLpad / 8, t.lj_cap, t.d_lj_cl_c, t.d_lj_cl_h, t.d_lj_tile_c, t.d_lj_tile_h, t.d_lj_sci_list, t.d_lj_sci_count};
        const float rc2_main = t.method == NB_EWALD ? t.p.rcc2 : t.p.rc2;
        // (one launch for both with an arrival-counter barrier between the phases was measured twice in round 4: with a release fence
        // before the arrival -- a device-scope release writes back the XCD's whole L2 -- 90 us instead of 11.6 + 12.7; with the boxes
        // as write-through device-scope stores and no fence 23.3 us, what the two launches take: profiles/r04_h_rejected.txt,
        // profiles/r04_r_fused_list_v2.txt)
        hipLaunchKernelGGL(gather_positions2_kernel, dim3(ntile + ntile_lj, h->R), dim3(64), 0, h->stream, ntile, ga, gb, h->Npad, h->d_pos, h->d_box);
        hipLaunchKernelGGL(build_sci_list2_kernel, dim3(ntile + ntile_lj, h->R), dim3(64), 0, h->stream, ntile, la, lb, rc2_main, t.p.rc2, h->d_box);
        // (the list lengths of the evaluation that re-sorted the molecules order the work items until the next re-sort: the geometry
        // of a tile changes slowly; REMD_NB_RANK=0: tile order)
        if (t.evals_since_sort == 1 && t.d_tile_of_rank && ntile <= 2048)
            hipLaunchKernelGGL(rank_tiles_kernel, dim3((ntile + 255) / 256, h->R), dim3(256), 0, h->stream, ntile, t.d_sci_count, t.d_tile_of_rank);
    } else {
        hipLaunchKernelGGL(gather_positions_kernel, dim3(ntile, h->R), dim3(64), 0, h->stream, h->Npad, h->Npad, t.d_order, h->d_pos, h->d_box,
                           t.d_spos, t.d_tile_c, t.d_tile_h, cl ? t.d_cl_c : (float4*)nullptr, cl ? t.d_cl_h : (float4*)nullptr);
        if (cl) launch_list_build(h, t, false);
    }
    static const bool debug = getenv("REMD_DEBUG") != nullptr;
    if (cl && t.evals_since_sort == 1 && (t.cl_cap < ntile * 8 || debug)) {
        // capacity check once per re-sort (the only host synchronisation of this path)
        std::vector<int> cnt((size_t)h->R * ntile);
        REMD_CHECK(h, hipMemcpyAsync(cnt.data(), t.d_sci_count, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        int mx = 0; for (int c : cnt) mx = std::max(mx, c);
        if (debug) {
            std::vector<unsigned int> ll((size_t)ntile * t.cl_cap);
            hipMemcpy(ll.data(), t.d_sci_list, sizeof(unsigned int) * ll.size(), hipMemcpyDeviceToHost);
            double ne = 0, nb = 0;
            for (int T = 0; T < ntile; ++T) for (int k = 0; k < std::min(cnt[T], t.cl_cap); ++k) { ne += 1; nb += __builtin_popcount(ll[(size_t)T * t.cl_cap + k] >> 16); }
            fprintf(stderr, "[remd] sci list (replica 0): %d tiles, %.1f entries per tile (max %d, cap %d), %.2f i clusters per entry\n",
                    ntile, ne / ntile, mx, t.cl_cap, nb / std::max(1.0, ne));
        }
        if (mx > t.cl_cap) t.clusters = false;        // fall back to the tile kernel
    }
    return 0;
}

// direct-space launch: the Newton's-third-law cluster-pair kernel over the super-cluster lists (main system and, for the
// Ewald / reaction-field methods, the LJ-only sub-system in the same launch), or the all-tile kernel when the system has no
// sortable groups or a list outgrew its capacity
template <int METHOD, bool ENERGY>
static void launch_nb(remd_ctx* h, nb_tables& t)
{
    const int ntile = (h->N + 63) / 64;
    if (t.sorting && t.clusters && t.d_order && t.d_sci_list && t.n_groups > 0 && ntile * 8 < 65536) {
        const int ncl = ntile * 8;
        const float* rl = t.has_alch ? t.d_rep_lam : (const float*)nullptr;
        const bool split = t.lj_split && t.d_lj_sci_list && (METHOD == NB_EWALD || METHOD == NB_RF);
        constexpr int MAIN = (METHOD == NB_EWALD) ? NB_EWALD_NOLJ : (METHOD == NB_RF) ? NB_RF_NOLJ : METHOD;
        const int ssplit = std::max(SCI_NW, std::min(16, t.sci_split / SCI_NW * SCI_NW));
        // (measured, profiles/r05_7_*: -1.2 % of a step on 8 x host-guest (71 tiles) and 16 x DHFR (369 tiles), +1 % on 24 x alanine dipeptide
        // (36 tiles: its launch is two rounds of the chip whatever the order) -- so from 64 tiles on; REMD_NB_RANK = 0 / 1 pins it)
        static const int rank_env = getenv("REMD_NB_RANK") ? atoi(getenv("REMD_NB_RANK")) : -1;
        // (rank_tiles_kernel compares every pair of a replica's tiles: measured up to 369 tiles; capped where the ranking would cost
        // more than the tail it removes)
        // (round 6, the blocks of a phased propagation: a block's launch is one round of the chip beside the other block's kernels, and the
        // longest lists first help there at 36 tiles too: 24 x alanine dipeptide +2.5 %, profiles/r06_45)
        const bool ranked = (rank_env < 0 ? ((ntile >= 64 || h->parent != nullptr) && ntile <= 2048) : rank_env != 0) && t.lj_split && t.d_lj_sci_list && t.d_tile_of_rank && (METHOD == NB_EWALD || METHOD == NB_RF);
        sci_args sa{h->N, h->Npad, ncl, t.cl_cap, t.excl_W, h->Npad, 0, ssplit, t.d_spos, t.d_sparam, t.d_excl, t.d_sci_list, t.d_sci_count, t.d_sforce,
                    t.d_sposi, ranked ? t.d_tile_of_rank : (const int*)nullptr};
        const int items_a = ntile * h->R * (ssplit / SCI_NW);
        if (split) {
            // one launch for both systems, one scatter for both sorted accumulators; the launch is either one workgroup per
            // work item or a resident set pulling items from a queue (t.nb_grid, chosen by timing: remd_nb_tune_step)
            sci_args sb{t.NL, t.NLpad, t.NLpad / 8, t.lj_cap, t.lj_excl_W, t.NLpad, ncl * 4, ssplit, t.d_lj_spos, t.d_lj_sparam, t.d_lj_excl,
                        t.d_lj_sci_list, t.d_lj_sci_count, t.d_lj_sforce, t.d_lj_sposi, nullptr};
            const int items = items_a + (t.NLpad / 64) * h->R * (ssplit / SCI_NW);
            const int env_grid = getenv("REMD_NB_PERSIST_GRID") ? atoi(getenv("REMD_NB_PERSIST_GRID")) : -1;
            const int persist_grid = env_grid >= 0 ? env_grid : t.nb_grid;
            const int grid = (h->pme_concurrent && persist_grid > 0) ? std::min(items, persist_grid) : items;
            // requested by remd_compute_forces (h->fold_pending): the scatter's workgroups count themselves done for the integrator
            // chain to poll (remd_fold_args) -- no signal launch behind it
            const bool fold = h->fold_pending && !ENERGY && t.d_pair_done;
            if (h->fold_pending && !fold) h->fold_pending = false;
            const bool tab = t.use_table && !ENERGY && SCI_EWALD(MAIN);
            const size_t tab_lds = tab ? sizeof(float4) * (size_t)t.p.ctab_n : 0;
#define LAUNCH_SCI2(ALCHF, TABF) hipLaunchKernelGGL((nonbonded_sci2_kernel<MAIN, NB_LJ_ONLY, ENERGY, ALCHF, SCI_NW, TABF>), dim3(grid), dim3(64 * SCI_NW), \
            tab_lds, h->stream, t.p, sa, sb, items_a, items, grid < items ? t.d_queue : (unsigned int*)nullptr, h->d_box, rl, h->d_epart, h->n_epart, h->R, \
            (const float4*)t.d_ctab)
            if (t.has_alch) { if (tab) LAUNCH_SCI2(true, true); else LAUNCH_SCI2(true, false); }
            else { if (tab) LAUNCH_SCI2(false, true); else LAUNCH_SCI2(false, false); }
#undef LAUNCH_SCI2
            const dim3 sgrid((h->Npad + t.NLpad + 255) / 256, h->R);
            if (fold) {
                t.pair_done_target += sgrid.x;           // per replica
                h->fold.done = t.d_pair_done; h->fold.target = t.pair_done_target;
            }
            hipLaunchKernelGGL(scatter_sorted_forces_kernel, sgrid, dim3(256), 0, h->stream, h->Npad,
                               t.d_order, t.d_sforce, t.NLpad, t.d_lj_order, t.d_lj_sforce, h->d_force, h->Npad, fold ? t.d_pair_done : (unsigned int*)nullptr);
            return;
        }
        const bool tab1 = t.use_table && !ENERGY && SCI_EWALD(METHOD);
        const size_t tab1_lds = tab1 ? sizeof(float4) * (size_t)t.p.ctab_n : 0;
#define LAUNCH_SCI(ALCHF, TABF) hipLaunchKernelGGL((nonbonded_sci_kernel<METHOD, ENERGY, ALCHF, SCI_NW, TABF>), dim3(items_a), dim3(64 * SCI_NW), tab1_lds, \
            h->stream, t.p, sa, h->d_box, rl, h->d_epart, h->n_epart, h->R, (const float4*)t.d_ctab)
        if (t.has_alch) { if (tab1) LAUNCH_SCI(true, true); else LAUNCH_SCI(true, false); }
        else { if (tab1) LAUNCH_SCI(false, true); else LAUNCH_SCI(false, false); }
#undef LAUNCH_SCI
        hipLaunchKernelGGL(scatter_sorted_forces_kernel, dim3((h->Npad + 255) / 256, h->R), dim3(256), 0, h->stream, h->Npad, t.d_order,
                           t.d_sforce, 0, (const int*)nullptr, (long long*)nullptr, h->d_force, h->Npad);
        return;
    }
    dim3 grid((ntile + NB_WAVES - 1) / NB_WAVES, t.p.n_jsplit, h->R);
    const size_t need = (size_t)h->R * t.p.n_jsplit * h->Npad;
    if (t.partial_n < need) { dfree(t.d_partial); if (hipMalloc(&t.d_partial, sizeof(float4) * need) != hipSuccess) return; t.partial_n = need; }
    const bool sorted = t.sorting && t.d_order && t.n_groups > 0;
    const float4* P = sorted ? t.d_spos : h->d_pos;
    const float4* prm = sorted ? t.d_sparam : t.d_param;
    const unsigned long long* mk = sorted ? t.d_smask : t.d_mask;
    const int* ord = sorted ? t.d_order : nullptr;
    if (t.has_alch)
        hipLaunchKernelGGL((nonbonded_kernel<METHOD, ENERGY, true>), grid, dim3(64 * NB_WAVES), 0, h->stream, t.p, h->N, h->Npad, P,
                           prm, mk, ord, t.d_tile_c, t.d_tile_h, h->d_box, t.d_rep_lam, t.d_partial, h->d_epart, h->n_epart, ntile);
    else
        hipLaunchKernelGGL((nonbonded_kernel<METHOD, ENERGY, false>), grid, dim3(64 * NB_WAVES), 0, h->stream, t.p, h->N, h->Npad, P,
                           prm, mk, ord, t.d_tile_c, t.d_tile_h, h->d_box, (const float*)nullptr, t.d_partial, h->d_epart, h->n_epart, ntile);
    hipLaunchKernelGGL(nb_reduce_kernel, dim3((h->N + 255) / 256, h->R), dim3(256), 0, h->stream, h->N, h->Npad, t.p.n_jsplit,
                       t.d_partial, h->d_force);
}

// after a device-side fault: partial sums a discarded evaluation left in the sorted accumulators must not reach the next one
void remd_nb_reset_accumulators(remd_ctx* h)
{
    nb_tables* t = g_nb.find(h);
    h->fold_pending = false;
    if (!t || h->R <= 0) return;
    if (t->d_sforce) hipMemsetAsync(t->d_sforce, 0, sizeof(long long) * 3 * (size_t)h->R * h->Npad, h->stream);
    if (t->d_lj_sforce) hipMemsetAsync(t->d_lj_sforce, 0, sizeof(long long) * 3 * (size_t)h->R * t->NLpad, h->stream);
}

int remd_nb_molecules(remd_ctx* h, const int** first, const int** size)
{
    nb_tables* it = g_nb.find(h);
    if (!it || it->n_groups <= 0) return 0;
    *first = it->d_grp_first; *size = it->d_grp_size;
    return it->n_groups;
}

const float* remd_nb_rep_lam(remd_ctx* h)
{
    if (h->regions_exact) {                   // general regions under the exact PME treatment: the regions' lambdas per replica (alch_regions.hip)
        const float4* prm; const float* le;
        if (remd_regions_pme_tables(h, &prm, &le)) return le;
    }
    nb_tables* it = g_nb.find(h);
    return (it && it->has_alch) ? it->d_rep_lam : nullptr;
}
const float4* remd_nb_param(remd_ctx* h)
{
    if (h->regions_exact) {                   // ... and the mesh kernels' charges: the alchemical atoms' too (the pair kernels see the environment's only)
        const float4* prm; const float* le;
        if (remd_regions_pme_tables(h, &prm, &le)) return prm;
    }
    return g_nb[h].d_param;
}

int remd_nb_required_epart(remd_ctx* h)
{
    const int ntile = (h->Npad + 63) / 64;
    // up to 4 slices per main cluster + 4 per LJ-sub-system cluster; the LAST slot belongs to the custom forces of general alchemical
    // regions (alch_regions.hip)
    return EP_NB0 + ntile * 8 * 4 + 8 + ntile * 8 * 4 + 1;
}

#define TUNE_SEG 40                       // one re-sort of the spatial order per segment (resort_interval)
#define TUNE_NC 6
// candidates: resident workgroups of the pair kernel (0 = one per item; 3 ... 2 per CU of the 256 CUs) and which stream's kernels
// run at raised wave priority (1: the pair kernel, 0: the mesh kernels).  Round 4: with the rebalanced Ewald split the
// direct-space stream is the critical path on the headline system and (0, pair priority) wins: 87.6 against 90.0 ms per 500 steps.
struct tune_cand { int grid, prio_pair; };
static const tune_cand g_tune_cands[TUNE_NC] = {{0, 1}, {0, 0}, {768, 0}, {640, 0}, {576, 0}, {512, 0}};
// called at the top of every eagerly launched MD step of remd_run_steps
void remd_nb_tune_step(remd_ctx* h, int steps_left_in_call)
{
    nb_tables* tp = g_nb.find(h);
    if (!tp || h->nb_method == REMD_NB_NONE) return;
    nb_tables& t = *tp;
    const bool fixed = getenv("REMD_NB_PERSIST_GRID") != nullptr;
    if (fixed || t.tune_state != 0 || !h->pme_concurrent || h->profiling == 2) return;
    if (t.tune_left == 0) {
        hipEvent_t boundary = nullptr;
        if (!t.tune_segs.empty() && !t.tune_segs.back().b) {            // close the open segment
            hipEventCreate(&boundary); hipEventRecord(boundary, h->stream);
            t.tune_segs.back().b = boundary;
        }
        if (t.tune_next == 2 * TUNE_NC) { t.tune_state = 1; t.nb_grid = 0; return; }      // two rounds of the candidates measured
        if (steps_left_in_call < TUNE_SEG) { t.nb_grid = 0; return; }           // resume in a later call
        nb_tables::tune_seg sg{t.tune_next % TUNE_NC, nullptr, nullptr};
        hipEventCreate(&sg.a); hipEventRecord(sg.a, h->stream);                // (each segment owns its pair of events)
        t.tune_segs.push_back(sg);
        t.nb_grid = g_tune_cands[sg.cand].grid; t.nb_prio = g_tune_cands[sg.cand].prio_pair;
        t.tune_next++;
        t.tune_left = TUNE_SEG;
    }
    t.tune_left--;
}
// after the stream has been synchronised: pick the fastest candidate
void remd_nb_tune_resolve(remd_ctx* h)
{
    nb_tables* tp = g_nb.find(h);
    if (!tp || tp->tune_state != 1) return;
    nb_tables& t = *tp;
    double ms[TUNE_NC] = {0};
    bool ok = true;
    for (auto& sg : t.tune_segs) {
        float e = 0.f;
        if (!sg.a || !sg.b || hipEventElapsedTime(&e, sg.a, sg.b) != hipSuccess) { ok = false; (void)hipGetLastError(); }
        ms[sg.cand] += e;
        if (sg.a) hipEventDestroy(sg.a);
        if (sg.b) hipEventDestroy(sg.b);
    }
    t.tune_segs.clear();
    int best = 0;
    for (int c = 1; c < TUNE_NC; ++c) if (ok && ms[c] < ms[best] * 0.995) best = c;     // ties go to the earlier candidate
    t.nb_grid = ok ? g_tune_cands[best].grid : 0; t.nb_prio = ok ? g_tune_cands[best].prio_pair : 0;
    t.tune_state = 2;
    if (getenv("REMD_NB_TUNE_VERBOSE"))
    {
        fprintf(stderr, "[remd] pair-kernel residency, ms per %d steps:", 2 * TUNE_SEG);
        for (int c = 0; c < TUNE_NC; ++c) fprintf(stderr, " %d%s: %.2f", g_tune_cands[c].grid, g_tune_cands[c].prio_pair ? "p" : "", ms[c]);
        fprintf(stderr, " -> %d workgroups\n", t.nb_grid);
    }
}
// the next force evaluation re-sorts the molecules (a propagation that is run again after a device-side failure starts in the
// sort phase a fresh handle would have: bit-identical trajectories; the resident small-system path moves atoms without
// evaluations being counted)
void remd_nb_invalidate_sort(remd_ctx* h)
{
    nb_tables* t = g_nb.find(h);
    if (t) t->evals_since_sort = 1 << 30;
}
// what the resident small-system kernel (integrate.hip) needs from the nonbonded tables: pair constants, per-atom parameters in
// ATOM order, the per-replica lambdas (refreshed here when the labels changed).  ok = 0: this system is not one it covers.
int remd_nb_resident_info(remd_ctx* h, int* ok, int* method, int* has_alch, nb_params* p, const float4** param, const float** rep_lam)
{
    *ok = 0; *method = -1; *has_alch = 0; *param = nullptr; *rep_lam = nullptr;
    if (h->nb_method == REMD_NB_NONE) { *ok = h->nocutoff ? 0 : 1; return 0; }      // e.g. the harmonic oscillator: external force only (a NoCutoff system: forces.hip + nocutoff.hip)
    if (h->n_regions > 0) return 0;                                       // general alchemical regions: forces.hip + alch_regions.hip
    nb_tables* t = g_nb.find(h);
    if (!t || t->method != NB_LJ_ONLY || t->n_exc != 0 || t->n_excl != 0 || h->n_exceptions != 0) return 0;
    int rc = update_replica_lambdas(h, *t);
    if (rc) return rc;
    *ok = 1; *method = t->method; *has_alch = t->has_alch ? 1 : 0; *p = t->p; *param = t->d_param;
    *rep_lam = t->has_alch ? t->d_rep_lam : nullptr;
    return 0;
}

int remd_compute_forces(remd_ctx* h, bool with_energy, unsigned class_mask)
{
    // class_mask: which force classes act (REMD_FG_* bits; everything unless a multiple-time-step splitting asks for the forces
    // of one force group, integrate.hip).  Energies are only defined for the full set.
    if (with_energy && (class_mask & 63u) != 63u) return remd_fail(h, -1, "energies need every force class");
    const bool do_ext = (class_mask >> REMD_FG_EXTERNAL) & 1u, do_bond = (class_mask >> REMD_FG_BOND) & 1u;
    const bool do_angle = (class_mask >> REMD_FG_ANGLE) & 1u, do_torsion = (class_mask >> REMD_FG_TORSION) & 1u;
    const bool do_nb = (class_mask >> REMD_FG_NONBONDED) & 1u, do_recip = (class_mask >> REMD_FG_RECIPROCAL) & 1u;
    if (!h->force_zeroed)
        REMD_CHECK(h, hipMemsetAsync(h->d_force, 0, sizeof(long long) * 3 * (size_t)h->Npad * h->R, h->stream));
    h->force_zeroed = false;
    if (with_energy)
        REMD_CHECK(h, hipMemsetAsync(h->d_epart, 0, sizeof(double) * (size_t)h->n_epart * h->R, h->stream));
    const int R = h->R;
#define LAUNCH_E(kern, ...) do { if (with_energy) hipLaunchKernelGGL(kern<true>, __VA_ARGS__); else hipLaunchKernelGGL(kern<false>, __VA_ARGS__); } while (0)
    // custom forces of general alchemical regions (alch_regions.hip): part of the direct-space nonbonded class, launched here -- in
    // front of the fork -- so that both branches of the evaluation are ordered behind them
    if (h->n_regions > 0 && do_nb) { int rcr = remd_regions_forces(h, with_energy, h->n_epart - 1); if (rcr) return rcr; }
    if (h->nocutoff && do_nb) { int rcn = remd_nocutoff_forces(h, with_energy, EP_NB0); if (rcn) return rcn; }
    if (h->gbsa && do_nb) { int rcg = remd_gbsa_forces(h, with_energy, EP_NB0 + 1); if (rcg) return rcg; }
    if (h->n_ext > 0 && do_ext) {
        remd_prof_scope ps(h, "ext_force");
        LAUNCH_E(ext_force_kernel, dim3(R), dim3(64), 0, h->stream, h->n_ext, h->d_ext_atoms, (float)h->ext_K, (float)h->ext_x0,
                 h->ext_U0, h->Npad, h->d_pos, h->d_force, h->d_epart, h->n_epart);
    }
    // Two branches between the integrator chains.  The reciprocal-space pipeline is the longer one, so IT stays on the main
    // stream directly behind the integrator; the direct-space launches go to the second stream (h->stream is swapped until
    // the join).  Fork and join are flags in device memory polled by kernels (remd_ctx::d_sync), or events when a handle has
    // fallen back to them (REMD_SYNC_EVENTS=1 / api.hip: remd_recover_device_flag).  Measured alternatives that lost their
    // A/B and were removed in round 3 (numbers in DESIGN.md 7b): the mesh branch on the second stream, the pair kernel ahead
    // of the listed terms, the listed terms in front of the pair kernel or on a third stream, scatter + listed terms + join
    // flag in one launch.
    bool forked = false, swapped = false;
    struct unswap { remd_ctx* h; bool* on; ~unswap() { if (*on) std::swap(h->stream, h->stream2); } } guard{h, &swapped};
    auto listed_terms = [&](int& total) {
        listed_tables T{};
        T.n_bonds = do_bond ? h->n_bonds : 0; T.n_angles = do_angle ? h->n_angles : 0; T.n_torsions = do_torsion ? h->n_torsions : 0;
        T.bond_atoms = h->d_bond_atoms; T.bond_params = h->d_bond_params;
        T.angle_atoms = h->d_angle_atoms; T.angle_params = h->d_angle_params;
        T.torsion_atoms = h->d_torsion_atoms; T.torsion_params = h->d_torsion_params;
        nb_tables* it = g_nb.find(h);
        if (it && h->nb_method != REMD_NB_NONE && do_nb) {
            nb_tables& t = *it;
            T.n_exc = t.n_exc; T.exc_atoms = t.d_exc_atoms; T.exc_params = t.d_exc_params;
            T.n_excl = t.n_excl; T.excl_atoms = t.d_excl_atoms; T.excl_qq = t.d_excl_qq;
            T.alpha = t.p.alpha; T.two_alpha_sqrtpi = t.p.two_alpha_sqrtpi;
            T.exc_alch = t.d_exc_alch; T.excl_alch = t.d_excl_alch; T.rep_lam = t.has_alch ? t.d_rep_lam : nullptr;
        }
        total = T.n_bonds + T.n_angles + T.n_torsions + T.n_exc + T.n_excl;
        // one (term, slot) entry per thread in atom order (REMD_LISTED_ATOMS=0: one term per thread)
        const bool atoms_env = !(getenv("REMD_LISTED_ATOMS") && atoi(getenv("REMD_LISTED_ATOMS")) == 0);
        T.n_aterm = 0; T.aterm = nullptr;
        if (atoms_env && total > 0 && h->d_aterm && h->n_aterm > 0) { T.aterm = h->d_aterm; T.n_aterm = h->n_aterm; total = h->n_aterm; }
        return T;
    };
    const bool listed_main_env = !(getenv("REMD_LISTED_MAIN") && atoi(getenv("REMD_LISTED_MAIN")) == 0);
    const bool listed_ride_env = !(getenv("REMD_LISTED_RIDE") && atoi(getenv("REMD_LISTED_RIDE")) == 0);
    bool listed_rode = false;
    if (h->nb_method != REMD_NB_NONE) {      // per-replica lambdas must be current before ANY kernel reads them
        nb_tables& t0 = g_nb[h];
        int rc0 = update_replica_lambdas(h, t0);
        if (rc0) return rc0;
        if (t0.method == NB_EWALD && h->overlap && h->stream2 && do_nb && do_recip) {
            // which stream's kernels run at raised wave priority: chosen together with the pair kernel's residency (remd_nb_tune_step)
            const int prio_env = getenv("REMD_NB_PRIO") ? atoi(getenv("REMD_NB_PRIO")) : -1;
            t0.p.prio = prio_env >= 0 ? (prio_env ? 1 : 0) : t0.nb_prio;
            h->mesh_prio_hi = !t0.p.prio;
            if (!h->sync_events) {
                // the first mesh launch (binning kernel, or the spreading pass when the chain binned the atoms) stores the
                // fork flag; a one-wavefront kernel at the head of the second stream polls it
                h->fork_seq_pending = ++h->sync_seq;
                hipLaunchKernelGGL(remd_spin_wait_kernel, dim3(1), dim3(64), 0, h->stream2, h->d_sync, h->sync_seq, h->d_sync + 2);
            } else {
                hipEventRecord(h->ev_fork, h->stream);
                hipStreamWaitEvent(h->stream2, h->ev_fork, 0);
            }
            // direct-space stream critical (t0.p.prio): the listed terms of a force-only evaluation leave it -- as extra workgroups of
            // the spreading launch (REMD_LISTED_RIDE=0: as a launch of their own behind the mesh launches)
            h->mesh_listed_total = 0;
            {
                // Round 6, measured and left OFF (REMD_PAIR_AFTER_XY=1 switches it on): when the plane pass keeps whole planes in LDS with ONE
                // workgroup per CU (DHFR: 128 x 128 planes, 134 KB) its workgroups cannot be placed beside resident pair workgroups -- the
                // pass crawls for the pair kernel's 600 us and then runs its 240 us alone (876 us in all).  Holding the pair kernel back
                // until the plane pass has ENDED (an event), with gather, lists and the listed terms in front of the wait, gives the plane
                // pass the chip (381 us beside the listed terms) -- and then the pair kernel and the inverse-z / gather pass stretch each
                // other to 690 us: 139.9 against 134.8 ms per 100 steps of 16 DHFR replicas, +30 % on host-guest and alanine.  At this
                // size the step is the SUM of its kernels' stand-alone times whatever the order (profiles/r06_12_dhfr_pair_after_plane_pass.txt).
                static const int env_pax = getenv("REMD_PAIR_AFTER_XY") ? atoi(getenv("REMD_PAIR_AFTER_XY")) : -1;
                h->pair_after_xy = !with_energy && (class_mask & 63u) == 63u && env_pax > 0;
                if (h->pair_after_xy && !h->ev_xy) hipEventCreateWithFlags(&h->ev_xy, hipEventDisableTiming);
                h->xy_recorded = false;
            }
            if (listed_main_env && listed_ride_env && !with_energy && !h->sync_events && t0.p.prio != 0 && !h->pair_after_xy) {
                int total = 0;
                h->mesh_listed = listed_terms(total);
                h->mesh_listed_total = total;
                listed_rode = total > 0;
            }
            rc0 = remd_pme_forces(h, with_energy, h->stream, 1);        // everything up to the inverse z transform + gather
            if (rc0) return rc0;
            std::swap(h->stream, h->stream2);
            swapped = true; forked = true;
        }
    }
    h->pme_concurrent = forked;
    if (!forked) { h->mesh_prio_hi = true; if (h->nb_method != REMD_NB_NONE) g_nb[h].p.prio = 0; }
    const bool merged = !with_energy;      // force-only evaluations: every listed term in one launch
    if (!merged && h->n_bonds > 0) {
        remd_prof_scope ps(h, "bonded");
        LAUNCH_E(bond_kernel, dim3(R), dim3(256), 0, h->stream, h->n_bonds, h->d_bond_atoms, h->d_bond_params, h->Npad,
                 h->d_pos, h->d_force, h->d_epart, h->n_epart);
    }
    if (!merged && h->n_angles > 0) {
        remd_prof_scope ps(h, "bonded");
        LAUNCH_E(angle_kernel, dim3(R), dim3(256), 0, h->stream, h->n_angles, h->d_angle_atoms, h->d_angle_params, h->Npad,
                 h->d_pos, h->d_force, h->d_epart, h->n_epart);
    }
    if (!merged && h->n_torsions > 0) {
        remd_prof_scope ps(h, "bonded");
        LAUNCH_E(torsion_kernel, dim3(R), dim3(256), 0, h->stream, h->n_torsions, h->d_torsion_atoms, h->d_torsion_params, h->Npad,
                 h->d_pos, h->d_force, h->d_epart, h->n_epart);
    }
    // listed terms of a force-only evaluation: ONE launch, behind the pair kernel (it then starts 20 us earlier, next to the
    // spreading pass: 118.9 -> 116.8 ms per 500 steps)
    // forked force-only evaluations: the listed terms go to the MAIN stream behind the mesh launches (they only need the positions
    // and add with the same integer atomics): since the Ewald split was rebalanced the direct-space stream is the critical path of
    // a step, and this takes a dependent 13 us launch off it (93.3 -> 89.5 ms per 500 steps).  REMD_LISTED_MAIN=0: behind the pair
    // kernel on the direct-space stream, as in round 3.
    // (only in the mode in which the direct-space stream is the critical one, chosen by the tuner together with the wave priority:
    // on a system whose mesh chain is the longer branch the extra work on the main stream costs what it saves here)
    const bool pax = forked && h->pair_after_xy;
    const bool listed_main = listed_main_env && forked && !with_energy && !h->sync_events && h->nb_method != REMD_NB_NONE && g_nb[h].p.prio != 0 && !pax;
    auto launch_listed = [&](hipStream_t lst) {
        if (listed_rode) return;                      // they rode in the spreading launch
        int total = 0;
        const listed_tables T = listed_terms(total);
        if (total > 0) {
            remd_prof_scope ps(h, "bonded");
            hipLaunchKernelGGL(listed_forces_kernel, dim3((total + 255) / 256, R), dim3(256), 0, lst, T, h->Npad, h->d_pos,
                               h->d_box, h->d_force);
        }
    };
    if (h->nb_method == REMD_NB_NONE) {
        if (merged) launch_listed(h->stream);
    } else {
        nb_tables& t = g_nb[h];
        int rc = do_nb ? ensure_sorted(h, t) : 0;
        if (rc) return rc;
        // (after the swap h->stream2 is the main stream: the listed terms queue up behind the mesh launches already enqueued there)
        if (merged && listed_main) launch_listed(h->stream2);
        // let the integrator chain that consumes this evaluation poll the scatter's done counter (remd_fold_args) instead of a flag
        // from a signal launch: only where that chain is certain to be the next launch on the main stream (remd_run_steps, plain
        // single-group programs) and in the mode in which the direct-space stream is the critical one; REMD_NB_FOLD=0: signal launch
        const bool fold_env = !(getenv("REMD_NB_FOLD") && atoi(getenv("REMD_NB_FOLD")) == 0);
        h->fold_pending = fold_env && listed_main && merged && h->defer_join_ok && do_nb && (class_mask & 63u) == 63u && t.sorting && t.clusters &&
                          t.lj_split && t.d_lj_sci_list && t.d_sci_list && h->profiling != 2;
        h->fold.done = nullptr;                  // (launch_nb fills remd_fold_args where it takes the request)
        bool listed_early = false;
        if (pax && h->xy_recorded) {
            if (merged) { launch_listed(h->stream); listed_early = true; }      // beside the spreading and the plane pass
            hipStreamWaitEvent(h->stream, h->ev_xy, 0); h->xy_recorded = false;
        }
        if (do_nb) {
            remd_prof_scope ps(h, "nonbonded");
            if (with_energy) {
                if (t.method == NB_LJ_ONLY) launch_nb<NB_LJ_ONLY, true>(h, t);
                else if (t.method == NB_RF) launch_nb<NB_RF, true>(h, t);
                else launch_nb<NB_EWALD, true>(h, t);
            } else {
                if (t.method == NB_LJ_ONLY) launch_nb<NB_LJ_ONLY, false>(h, t);
                else if (t.method == NB_RF) launch_nb<NB_RF, false>(h, t);
                else launch_nb<NB_EWALD, false>(h, t);
            }
        }
        h->fold_pending = h->fold_pending && h->fold.done != nullptr;
        if (merged && !listed_main && !listed_early) launch_listed(h->stream);
        if (!merged && t.n_exc > 0) {
            remd_prof_scope ps(h, "exceptions");
            LAUNCH_E(exception_kernel, dim3(R), dim3(256), 0, h->stream, t.n_exc, t.d_exc_atoms, t.d_exc_params, t.d_exc_alch,
                     t.has_alch ? t.d_rep_lam : (const float*)nullptr, h->Npad,
                     h->d_pos, h->d_box, h->d_force, h->d_epart, h->n_epart);
        }
        if (!merged && t.n_excl > 0) {
            remd_prof_scope ps(h, "exceptions");
            LAUNCH_E(ewald_exclusion_kernel, dim3(R), dim3(256), 0, h->stream, t.n_excl, t.d_excl_atoms, t.d_excl_qq, t.d_excl_alch,
                     t.has_alch ? t.d_rep_lam : (const float*)nullptr, t.p.alpha,
                     t.p.two_alpha_sqrtpi, h->Npad, h->d_pos, h->d_box, h->d_force, h->d_epart, h->n_epart);
        }
        if (t.method == NB_EWALD) {
            if (forked) {                                                       // join
                if (h->fold_pending) {
                    // nothing left to launch on the direct-space stream: the chain polls the scatter's done counter
                    std::swap(h->stream, h->stream2); swapped = false;
                } else if (!h->sync_events) {
                    // (the scatter's last workgroup publishing the join instead of this launch: its arrival counter costs more than the
                    // launch it saves, 14 us against 5.2 + 5.6 with two-level counters -- profiles/r04_h_rejected.txt)
                    hipLaunchKernelGGL(remd_signal_kernel, dim3(1), dim3(64), 0, h->stream, h->d_sync + 1, h->sync_seq);
                    std::swap(h->stream, h->stream2); swapped = false;
                    // inside remd_run_steps the launch that follows on the main stream is an integrator chain: it polls the
                    // flag in its prologue (no kernel of its own for the wait)
                    if (h->defer_join_ok && !with_energy) h->join_deferred = h->sync_seq;
                    else hipLaunchKernelGGL(remd_spin_wait_kernel, dim3(1), dim3(64), 0, h->stream, h->d_sync + 1, h->sync_seq, h->d_sync + 2);
                } else {
                    hipEventRecord(h->ev_join, h->stream);
                    std::swap(h->stream, h->stream2); swapped = false;
                    hipStreamWaitEvent(h->stream, h->ev_join, 0);
                }
                rc = remd_pme_forces(h, with_energy, h->stream, 2); if (rc) return rc;        // (the energy reduction of the mesh part)
            }
            else if (do_recip) { rc = remd_pme_forces(h, with_energy, h->stream); if (rc) return rc; }
        }
        if (with_energy)
            hipLaunchKernelGGL(const_energy_kernel, dim3((R + 63) / 64), dim3(64), 0, h->stream, R, t.disp_coeff, t.self_nn, t.self_aa,
                               t.q_n, t.q_a, t.net_charge_term, t.has_alch ? t.d_rep_lam : (const float*)nullptr, h->d_box,
                               h->d_epart, h->n_epart);
    }
#undef LAUNCH_E
    if (with_energy)
        hipLaunchKernelGGL(reduce_energy_kernel, dim3(R), dim3(64), 0, h->stream, h->n_epart, h->d_epart, h->d_potential);
    REMD_CHECK(h, hipGetLastError());
    h->forces_valid = true;
    return 0;
}

// u_kl with lambda_electrostatics states: every Coulomb term is bilinear in the charges and alchemical charges scale
// linearly, so U(lambda_e) = a + b l + c l^2 EXACTLY; three energy passes at l = 0, 1/2, 1 determine a, b, c per replica.
__global__ void assemble_ukl_poly_kernel(int R, int K, const double* __restrict__ probe /*[3][R]*/,
                                         const double* __restrict__ beta, const double* __restrict__ econst,
                                         const double* __restrict__ lam_e, const double* __restrict__ alch, const int* __restrict__ own,
                                         const double* __restrict__ pressure /*[K] or null*/, const float* __restrict__ box,
                                         double econst_vref, double* __restrict__ ukl_rows, double* __restrict__ potential)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * K) return;
    const int r = t / K, l = t % K;
    const double P0 = probe[r], Ph = probe[R + r], P1 = probe[2 * R + r];
    const double c = 2.0 * (P1 - P0) - 4.0 * (Ph - P0);
    const double b = (P1 - P0) - c;
    const double le = lam_e[l];
    const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
    // the probes ran at the replica's own lambda_sterics: at its own lambda_electrostatics the polynomial is the potential of
    // the replica's state (d_potential holds the last probe until here)
    if (l == own[r]) potential[r] = P0 + b * le + c * le * le;
    double U = P0 + b * le + c * le * le + econst[l] * ((econst_vref > 0.0 && V > 0.0) ? econst_vref / V : 1.0);
    if (alch) U += alch[t] - alch[(size_t)r * K + own[r]];
    if (pressure) U += pressure[l] * V;
    ukl_rows[t] = beta[l] * U;
}

// u_kl with SEVERAL regions' lambda_electrostatics under the exact PME treatment (alch_regions.hip): every Coulomb term is bilinear in the
// charges and a region's charges scale with its lambda, so U(l_1 .. l_n) = c0 + sum_x (b_x l_x + a_x l_x^2) + sum_{x<y} c_xy l_x l_y EXACTLY;
// probes: all 0; per region l_x = 1/2 and 1 (others 0); per pair l_x = l_y = 1  ->  (n + 1)(n + 2) / 2 energy passes.
__global__ void assemble_ukl_quad_kernel(int R, int K, int n, const double* __restrict__ probe /*[P][R]*/,
                                         const double* __restrict__ beta, const double* __restrict__ econst,
                                         const float* __restrict__ state_le /*[K][4]*/, const double* __restrict__ alch, const int* __restrict__ own,
                                         const double* __restrict__ pressure /*[K] or null*/, const float* __restrict__ box,
                                         double econst_vref, double* __restrict__ ukl_rows, double* __restrict__ potential)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * K) return;
    const int r = t / K, l = t % K;
    const double c0 = probe[r];
    double lin[4], a[4], b[4], le[4];
    for (int x = 0; x < n; ++x) {
        const double Ph = probe[(size_t)(1 + 2 * x) * R + r] - c0, P1 = probe[(size_t)(2 + 2 * x) * R + r] - c0;
        a[x] = 2.0 * P1 - 4.0 * Ph; b[x] = P1 - a[x]; lin[x] = P1;
        le[x] = (double)state_le[4 * l + x];
    }
    double U = c0;
    for (int x = 0; x < n; ++x) U += b[x] * le[x] + a[x] * le[x] * le[x];
    int p = 1 + 2 * n;
    for (int x = 0; x < n; ++x) for (int y = x + 1; y < n; ++y, ++p)
        U += (probe[(size_t)p * R + r] - c0 - lin[x] - lin[y]) * le[x] * le[y];
    const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
    // the probes ran at the replica's own lambda_sterics: at its own lambda_electrostatics the form is the potential of its state
    if (l == own[r]) potential[r] = U;
    U += econst[l] * ((econst_vref > 0.0 && V > 0.0) ? econst_vref / V : 1.0);
    if (alch) U += alch[t] - alch[(size_t)r * K + own[r]];
    if (pressure) U += pressure[l] * V;
    ukl_rows[t] = beta[l] * U;
}

int remd_assemble_ukl(remd_ctx* h, double* d_rows)
{
    const int n = h->R * h->K;
    const double* alch = nullptr;
    nb_tables* it = g_nb.find(h);
    bool poly = false;
    const int* d_own_states = nullptr;
    if (h->n_regions > 0 && (h->nb_method != REMD_NB_NONE || h->nocutoff)) {
        // general alchemical regions: the custom forces at every state's lambdas (d_potential holds them at the replicas' own)
        nb_tables& t = g_nb[h];
        it = &t;
        if (t.alch_R != h->R || t.alch_K != h->K) {
            dfree(t.d_alch_ukl); dfree(t.d_state_lam); dfree(t.d_own);
            REMD_CHECK(h, hipMalloc(&t.d_alch_ukl, sizeof(double) * (size_t)n));
            t.alch_R = h->R; t.alch_K = h->K;
        }
        int rc = remd_regions_ukl(h, t.d_alch_ukl, &d_own_states);
        if (rc) return rc;
        if (h->gbsa && (rc = remd_gbsa_ukl(h, t.d_alch_ukl))) return rc;       // implicit solvent with alchemical particles: its energy at every state's lambda
        alch = t.d_alch_ukl;
        if (h->regions_exact) {
            int nreg = 0; const float* d_state_le = nullptr;
            if ((rc = remd_regions_le_override(h, nullptr, &nreg, &d_state_le))) return rc;
            const int P = (nreg + 1) * (nreg + 2) / 2;
            if (t.probe_R != h->R * P) { dfree(t.d_probe); REMD_CHECK(h, hipMalloc(&t.d_probe, sizeof(double) * (size_t)P * h->R)); t.probe_R = h->R * P; }
            std::vector<std::vector<float>> probes;
            probes.push_back(std::vector<float>(4, 0.f));
            for (int x = 0; x < nreg; ++x) for (float v : {0.5f, 1.f}) { std::vector<float> q(4, 0.f); q[x] = v; probes.push_back(q); }
            for (int x = 0; x < nreg; ++x) for (int y = x + 1; y < nreg; ++y) { std::vector<float> q(4, 0.f); q[x] = q[y] = 1.f; probes.push_back(q); }
            for (int q = 0; q < P; ++q) {
                if ((rc = remd_regions_le_override(h, probes[q].data(), nullptr, nullptr))) return rc;
                rc = remd_compute_forces(h, true);
                if (rc) { remd_regions_le_override(h, nullptr, nullptr, nullptr); return rc; }
                REMD_CHECK(h, hipMemcpyAsync(t.d_probe + (size_t)q * h->R, h->d_potential, sizeof(double) * h->R, hipMemcpyDeviceToDevice, h->stream));
            }
            remd_regions_le_override(h, nullptr, nullptr, nullptr);
            h->forces_valid = false;        // the last pass used a probe's lambdas, not the replicas' own
            hipLaunchKernelGGL(assemble_ukl_quad_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->R, h->K, nreg, t.d_probe,
                               h->d_beta, h->d_econst, d_state_le, alch, d_own_states, h->baro_frequency > 0 ? h->d_pressure : (const double*)nullptr,
                               h->d_box, h->econst_vref, d_rows, h->d_potential);
            REMD_CHECK(h, hipGetLastError());
            return 0;
        }
    } else
    if (it && it->has_alch && h->nb_method != REMD_NB_NONE) {
        nb_tables& t = *it;
        if (t.alch_R != h->R || t.alch_K != h->K) {
            dfree(t.d_alch_ukl); dfree(t.d_state_lam); dfree(t.d_own);
            REMD_CHECK(h, hipMalloc(&t.d_alch_ukl, sizeof(double) * (size_t)n));
            REMD_CHECK(h, hipMalloc(&t.d_state_lam, sizeof(double) * 2 * h->K));
            REMD_CHECK(h, hipMalloc(&t.d_own, sizeof(int) * h->R));
            t.alch_R = h->R; t.alch_K = h->K;
        }
        std::vector<double> sl(2 * (size_t)h->K);
        for (int k = 0; k < h->K; ++k) {
            sl[2 * k] = pow(h->lam_s[k], h->sc_a);
            sl[2 * k + 1] = h->sc_alpha * pow(1.0 - h->lam_s[k], h->sc_b);
        }
        std::vector<int> own(h->R);
        for (int r = 0; r < h->R; ++r) own[r] = h->labels.empty() ? 0 : (int)h->labels[h->r_begin + r];
        REMD_CHECK(h, hipMemcpyAsync(t.d_state_lam, sl.data(), sizeof(double) * sl.size(), hipMemcpyHostToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(t.d_own, own.data(), sizeof(int) * own.size(), hipMemcpyHostToDevice, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        {
            remd_prof_scope ps(h, "alch_ukl");
            hipLaunchKernelGGL(alch_ukl_kernel, dim3(h->K, h->R), dim3(256), 0, h->stream, t.p, h->N, h->Npad, h->n_alch,
                               h->d_alch_atoms, h->d_pos, t.d_param, h->d_box, h->K, t.d_state_lam, t.d_alch_ukl,
                               t.n_exc, t.d_exc_atoms, t.d_exc_params, t.d_exc_alch, t.d_mask);
        }
        alch = t.d_alch_ukl; d_own_states = t.d_own;
        // lambda_electrostatics states need the polynomial only when alchemical atoms carry charge
        bool lam_e_varies = false;
        for (int k = 0; k < h->K; ++k) lam_e_varies |= (h->lam_e[k] != h->lam_e[0]) || (h->lam_e[k] != 1.0);
        poly = lam_e_varies && (t.self_aa != 0.0);
        if (poly) {
            if (t.probe_R != h->R) { dfree(t.d_probe); REMD_CHECK(h, hipMalloc(&t.d_probe, sizeof(double) * 3 * h->R)); t.probe_R = h->R; }
            const double probes[3] = { 0.0, 0.5, 1.0 };
            for (int q = 0; q < 3; ++q) {
                t.lam_e_override = probes[q];
                int rc = remd_compute_forces(h, true);
                t.lam_e_override = -1.0;
                if (rc) return rc;
                REMD_CHECK(h, hipMemcpyAsync(t.d_probe + (size_t)q * h->R, h->d_potential, sizeof(double) * h->R, hipMemcpyDeviceToDevice, h->stream));
            }
            h->forces_valid = false;        // the last pass used a probe lambda, not the replicas' own
            hipLaunchKernelGGL(assemble_ukl_poly_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->R, h->K, t.d_probe,
                               h->d_beta, h->d_econst, h->d_lam_e, alch, t.d_own, h->baro_frequency > 0 ? h->d_pressure : (const double*)nullptr,
                               h->d_box, h->econst_vref, d_rows, h->d_potential);
            REMD_CHECK(h, hipGetLastError());
            return 0;
        }
    }
    hipLaunchKernelGGL(assemble_ukl_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->R, h->K,
                       h->d_potential, h->d_beta, h->d_econst, alch, alch ? d_own_states : (const int*)nullptr, h->baro_frequency > 0 ? h->d_pressure : (const double*)nullptr, h->d_box, h->econst_vref, d_rows);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
