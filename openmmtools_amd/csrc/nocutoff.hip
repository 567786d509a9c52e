// NonbondedForce with NoCutoff (gfx950): the vacuum test systems of the reference (testsystems.AlanineDipeptideVacuum, testsystems.py:3352-3388,
// LennardJonesCluster :1700-1780, TolueneVacuum ...; openmm.NonbondedForce.NoCutoff): every pair of atoms, plain Lennard-Jones + Coulomb,
// no periodic box, no switching function, no dispersion correction; exceptions replace their pair.  The f64 restatement is
// oracle/forcefield.py (method 3).
//
// These systems are small (tens to a few thousand atoms), so the sum is the direct one: workgroup = (tile of 64 atoms i, replica), the j
// atoms staged through LDS 1024 at a time and split over 16 wavefronts, every pair from both sides (the force on i needs no reduction: one fixed-point atomic triple per
// atom and launch), excluded pairs from an N x N bit matrix.  Energies: a second variant with per-workgroup f64 partials in a fixed order.
// The rest of the engine sees such a handle as one without a cutoff-based nonbonded force (remd_ctx::nb_method = REMD_NB_NONE,
// remd_ctx::nocutoff = 1): listed terms, integrator chain, Monte Carlo moves are those of the non-periodic path the harmonic oscillator
// has always used.
#include "remd_internal.h"
#include "listed_terms.h"
#include "nocutoff_pair.h"
#include <cmath>
#include <algorithm>

struct nocutoff_tables {
    int N = 0, words = 0, n_exc = 0, n_tile = 0;
    float4* d_param = nullptr;             // [Npad] q sqrt(k_e), sigma / 2, 2 sqrt(eps), 0
    unsigned int* d_excl = nullptr;        // [N][words] excluded partners (all exceptions)
    int* d_exc_atoms = nullptr; float4* d_exc_par = nullptr;       // non-zero exceptions: k_e qq, sigma, 4 eps
    double* d_epart = nullptr; int epart_R = 0;                    // [R][n_tile + 1]
};
static handle_table<nocutoff_tables> g_nc;

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

// workgroup = (tile of 64 atoms i, replica), NC_WAVES wavefronts: lane = atom i, wavefront w takes the partners j = w (mod NC_WAVES) of
// every block of NC_BLOCK atoms staged in LDS; the partial sums of an atom go through LDS and are added in wavefront order (a fixed order:
// the forces do not depend on scheduling).  (One wavefront per tile looping over all N partners is a chain of N dependent pair terms:
// 31 us for the 156 atoms of CB7:B2 in vacuum, rocprofv3, profiles/r06_43.)
#define NC_WAVES 16
#define NC_BLOCK (64 * NC_WAVES)
template <bool ENERGY>
__global__ __launch_bounds__(NC_BLOCK)
void nocutoff_kernel(int N, int Npad, int words, const float4* __restrict__ param, const unsigned int* __restrict__ excl,
                     int n_exc, const int* __restrict__ exc_atoms, const float4* __restrict__ exc_par,
                     const float4* __restrict__ pos, long long* __restrict__ force, double* __restrict__ epart, int n_tile)
{
    __shared__ float4 s_x[NC_BLOCK], s_p[NC_BLOCK];
    __shared__ unsigned int s_m[64][NC_BLOCK / 32 + 1];       // exclusion words of the tile's atoms for the staged block (padded: no bank conflicts)
    __shared__ float4 s_part[NC_WAVES][64];
    __shared__ double s_e[NC_WAVES];
    const int r = blockIdx.y, tile = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float4* P = pos + (size_t)r * Npad;
    long long* F = force + (size_t)r * 3 * Npad;
    double e = 0.0;
    if (tile < n_tile) {
        const int i = tile * 64 + lane;
        const bool live = i < N;
        const float4 xi = live ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 pi = live ? param[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (int j0 = 0; j0 < N; j0 += NC_BLOCK) {
            __syncthreads();
            {
                const int j = j0 + (int)threadIdx.x;
                s_x[threadIdx.x] = j < N ? P[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                s_p[threadIdx.x] = j < N ? param[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                // 64 atoms x NC_BLOCK / 32 words: two per thread
                for (int q = threadIdx.x; q < 64 * (NC_BLOCK / 32); q += NC_BLOCK) {
                    const int a = q / (NC_BLOCK / 32), wd = q % (NC_BLOCK / 32), ai = tile * 64 + a, gw = (j0 >> 5) + wd;
                    s_m[a][wd] = (ai < N && gw < words) ? excl[(size_t)ai * words + gw] : 0u;
                }
            }
            __syncthreads();
            if (!live) continue;
            const int jn = min(NC_BLOCK, N - j0);
            for (int k = w; k < jn; k += NC_WAVES) {
                if (j0 + k == i || ((s_m[lane][k >> 5] >> (k & 31)) & 1u)) continue;
                nocutoff_pair<ENERGY>(xi, pi, s_x[k], s_p[k], fx, fy, fz, e);
            }
        }
        s_part[w][lane] = make_float4(fx, fy, fz, 0.f);
        __syncthreads();
        if (w == 0 && live) {
            fx = 0.f; fy = 0.f; fz = 0.f;
            for (int q = 0; q < NC_WAVES; ++q) { const float4 t = s_part[q][lane]; fx += t.x; fy += t.y; fz += t.z; }
            add_force(F, Npad, i, fx, fy, fz);
        }
    } else {
        // the last workgroup of a replica: the exceptions (plain Coulomb + Lennard-Jones of the exception's own parameters)
        for (int t = threadIdx.x; t < n_exc; t += NC_BLOCK) {
            const int i = exc_atoms[2 * t], j = exc_atoms[2 * t + 1];
            const float4 par = exc_par[t];
            const float3 d = sub3(ld3(P, j), ld3(P, i));
            const float fr = nocutoff_exception<ENERGY>(par, d, e);
            add_force(F, Npad, i, fr * d.x, fr * d.y, fr * d.z);
            add_force(F, Npad, j, -fr * d.x, -fr * d.y, -fr * d.z);
        }
    }
    if (ENERGY) {
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (lane == 0) s_e[w] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int q = 0; q < NC_WAVES; ++q) tot += s_e[q];
            epart[(size_t)r * (n_tile + 1) + tile] = tot;
        }
    }
}

// sums a replica's partials in a fixed order into slot `slot` of the handle's energy partials
__global__ __launch_bounds__(64)
void nocutoff_reduce_kernel(int n, const double* __restrict__ part, double* __restrict__ epart, int n_epart, int slot)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 64) e += part[(size_t)r * n + t];
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if (threadIdx.x == 0) epart[(size_t)r * n_epart + slot] = e;
}

void remd_nocutoff_release(remd_ctx* h)
{
    nocutoff_tables* t = g_nc.find(h);
    if (t) {
        dfree(t->d_param); dfree(t->d_excl); dfree(t->d_exc_atoms); dfree(t->d_exc_par); dfree(t->d_epart);
        g_nc.erase(h);
    }
    h->nocutoff = 0;
}

int remd_nocutoff_build(remd_ctx* h, const remd_system_desc* d)
{
    remd_nocutoff_release(h);
    const int N = d->n_atoms;
    if (!d->charge || !d->sigma || !d->epsilon) return remd_fail(h, -1, "nonbonded parameter arrays missing");
    if (N > 16384) return remd_fail(h, -3, "NoCutoff: more than 16384 atoms (the direct sum is meant for the vacuum test systems)");
    nocutoff_tables& t = g_nc[h];
    t.N = N; t.words = (N + 31) / 32 + 1; t.n_tile = (N + 63) / 64;
    const double sqk = sqrt(REMD_ONE_4PI_EPS0);
    std::vector<float4> prm(h->Npad, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int i = 0; i < N; ++i) prm[i] = make_float4((float)(d->charge[i] * sqk), (float)(0.5 * d->sigma[i]), (float)(2.0 * sqrt(d->epsilon[i])), 0.f);
    std::vector<unsigned int> ex((size_t)N * t.words, 0u);
    std::vector<int> ea; std::vector<float4> ep;
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
        if (i < 0 || j < 0 || i >= N || j >= N || i == j) { remd_nocutoff_release(h); return remd_fail(h, -3, "bad exception pair"); }
        ex[(size_t)i * t.words + (j >> 5)] |= 1u << (j & 31);
        ex[(size_t)j * t.words + (i >> 5)] |= 1u << (i & 31);
        const double qq = d->exception_params[3 * e], sg = d->exception_params[3 * e + 1], eps = d->exception_params[3 * e + 2];
        if (qq != 0.0 || eps != 0.0) { ea.push_back(i); ea.push_back(j); ep.push_back(make_float4((float)(qq * REMD_ONE_4PI_EPS0), (float)sg, (float)(4.0 * eps), 0.f)); }
    }
    t.n_exc = (int)ea.size() / 2;
    int rc;
    if ((rc = upload(h, t.d_param, prm)) || (rc = upload(h, t.d_excl, ex)) || (rc = upload(h, t.d_exc_atoms, ea)) || (rc = upload(h, t.d_exc_par, ep))) { remd_nocutoff_release(h); return rc; }
    h->nocutoff = 1;
    h->n_exceptions = t.n_exc;
    return 0;
}

// the tables, for the resident small-molecule kernel (integrate.hip)
int remd_nocutoff_info(remd_ctx* h, const float4** param, const unsigned int** excl, int* words, int* n_exc, const int** exc_atoms, const float4** exc_par)
{
    nocutoff_tables* t = g_nc.find(h);
    if (!t) return -1;
    *param = t->d_param; *excl = t->d_excl; *words = t->words; *n_exc = t->n_exc; *exc_atoms = t->d_exc_atoms; *exc_par = t->d_exc_par;
    return 0;
}

// at the head of a force evaluation (positions current, accumulators zeroed), like the harmonic external force
int remd_nocutoff_forces(remd_ctx* h, bool with_energy, int ep_slot)
{
    nocutoff_tables* tp = g_nc.find(h);
    if (!tp) return remd_fail(h, -2, "NoCutoff: no tables on this handle");
    nocutoff_tables& t = *tp;
    remd_prof_scope ps(h, "nonbonded");
    const dim3 grid(t.n_tile + 1, h->R);
    if (with_energy) {
        if (t.epart_R != h->R) { dfree(t.d_epart); REMD_CHECK(h, hipMalloc(&t.d_epart, sizeof(double) * (size_t)h->R * (t.n_tile + 1))); t.epart_R = h->R; }
        hipLaunchKernelGGL(nocutoff_kernel<true>, grid, dim3(NC_BLOCK), 0, h->stream, t.N, h->Npad, t.words, t.d_param, t.d_excl, t.n_exc, t.d_exc_atoms, t.d_exc_par,
                           h->d_pos, h->d_force, t.d_epart, t.n_tile);
        hipLaunchKernelGGL(nocutoff_reduce_kernel, dim3(h->R), dim3(64), 0, h->stream, t.n_tile + 1, t.d_epart, h->d_epart, h->n_epart, ep_slot);
    } else {
        hipLaunchKernelGGL(nocutoff_kernel<false>, grid, dim3(NC_BLOCK), 0, h->stream, t.N, h->Npad, t.words, t.d_param, t.d_excl, t.n_exc, t.d_exc_atoms, t.d_exc_par,
                           h->d_pos, h->d_force, (double*)nullptr, t.n_tile);
    }
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
