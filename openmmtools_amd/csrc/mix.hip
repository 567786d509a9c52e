// Gibbs state-label mixing kernels (gfx950).
//
// Reference semantics (restated, not copied):
//   swap-all        openmmtools/multistate/replicaexchange.py:294-349  (R^3 sequential attempts)
//   swap-neighbors  replicaexchange.py:366-380
//   SAMS global     openmmtools/multistate/sams.py:477-501
//
// The swap-all loop is a serial dependency chain.  It is parallelised *exactly*: attempts
// that touch disjoint replica slots commute (label writes are disjoint, the count
// increments are integer adds), so a wavefront draws 64 consecutive attempts from the
// counter-based Philox stream, builds for every lane the bit mask of EARLIER lanes that
// share a replica with it, and then retires lanes in dependency order: a lane fires as
// soon as all its earlier conflicting lanes have fired.  The result is bit-identical to
// the sequential loop (tests/test_mix_parity.py against oracle/mix_oracle.c).
#include "remd_internal.h"
#include "rng.h"

#define MIX_MAX_LDS_UKL (128 * 1024)

__device__ __forceinline__ double mix_logp(const double* __restrict__ U, int ld, int i, int j, int si, int sj)
{
    // replicaexchange.py:332-336, same association: (-(e_ij + e_ji) + e_ii) + e_jj
    double e_ij = U[(size_t)i * ld + sj];
    double e_ji = U[(size_t)j * ld + si];
    double e_ii = U[(size_t)i * ld + si];
    double e_jj = U[(size_t)j * ld + sj];
    double a = __dadd_rn(e_ij, e_ji);
    double b = __dadd_rn(-a, e_ii);
    return __dadd_rn(b, e_jj);
}

__global__ __launch_bounds__(64)
void mix_swap_all_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld,
                         const double* __restrict__ g_ukl, int64_t* __restrict__ g_labels,
                         unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop,
                         int64_t n_attempts, int ukl_in_lds, int stats_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_u = reinterpret_cast<double*>(smem);
    volatile int* s_lab = reinterpret_cast<int*>(smem + (ukl_in_lds ? (size_t)R * K * sizeof(double) : 0));
    unsigned* s_nprop = reinterpret_cast<unsigned*>(const_cast<int*>(s_lab) + ((R + 1) & ~1));   // [K*K] when stats_in_lds
    unsigned* s_nacc = s_nprop + K * K;
    const int lane = threadIdx.x;

    if (ukl_in_lds)
        for (int t = lane; t < R * K; t += 64) s_u[t] = g_ukl[(size_t)(t / K) * ld + (t % K)];
    for (int t = lane; t < R; t += 64) s_lab[t] = (int)g_labels[t];
    if (stats_in_lds) for (int t = lane; t < 2 * K * K; t += 64) s_nprop[t] = 0u;
    __syncthreads();
    const double* U = ukl_in_lds ? s_u : g_ukl;
    const int ldu = ukl_in_lds ? K : ld;

    for (int64_t base = 0; base < n_attempts; base += 64) {
        const int64_t k = base + lane;
        const bool valid = k < n_attempts;
        philox4 w = remd_philox(seed, REMD_STREAM_SWAP_ALL, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration);
        const int i = (int)remd_mulhi32(w.w[0], (uint32_t)R);     // randint(R), replicaexchange.py:324
        const int j = (int)remd_mulhi32(w.w[1], (uint32_t)R);     // :325
        const double u = remd_u53(w.w[2], w.w[3]);

        // dependency mask over earlier lanes of this batch
        unsigned long long dep = 0ull;
#pragma unroll 8
        for (int m = 0; m < 64; ++m) {
            const int im = __builtin_amdgcn_readlane(i, m);
            const int jm = __builtin_amdgcn_readlane(j, m);
            const bool c = (m < lane) && (im == i || im == j || jm == i || jm == j);
            dep |= (unsigned long long)c << m;
        }
        unsigned long long done = ~__ballot(valid);
        while (done != ~0ull) {
            const bool ready = !((done >> lane) & 1ull) && ((dep & ~done) == 0ull);
            if (ready) {
                const int si = s_lab[i], sj = s_lab[j];                                  // :328-329
                const double log_p = mix_logp(U, ldu, i, j, si, sj);                     // :332-336
                const bool acc = log_p >= 0.0 || u < remd_exp_det(log_p);                // :343
                if (acc) { s_lab[i] = sj; s_lab[j] = si; }                               // :345-346
                if (stats_in_lds) {
                    atomicAdd(&s_nprop[si * K + sj], 1u); atomicAdd(&s_nprop[sj * K + si], 1u);          // :339-340
                    if (acc) { atomicAdd(&s_nacc[si * K + sj], 1u); atomicAdd(&s_nacc[sj * K + si], 1u); }   // :348-349
                } else {
                    atomicAdd(&g_nprop[(size_t)si * K + sj], 1ull); atomicAdd(&g_nprop[(size_t)sj * K + si], 1ull);
                    if (acc) { atomicAdd(&g_nacc[(size_t)si * K + sj], 1ull); atomicAdd(&g_nacc[(size_t)sj * K + si], 1ull); }
                }
            }
            // single wavefront: LDS operations retire in program order, so the next round's label reads see this
            // round's writes; only the compiler must not move them (volatile labels + scheduling barrier).  No
            // s_waitcnt vmcnt here: the fire-and-forget global counter atomics must not stall the chain.
            __builtin_amdgcn_wave_barrier();
            done |= __ballot(ready);
        }
    }
    __syncthreads();
    for (int t = lane; t < R; t += 64) g_labels[t] = (int64_t)s_lab[t];
    if (stats_in_lds)
        for (int t = lane; t < K * K; t += 64) { g_nprop[t] = s_nprop[t]; g_nacc[t] = s_nacc[t]; }
}

// replicaexchange.py:366-380 — neighbouring STATE pairs (s, s+1), s = offset, offset+2, ...
__global__ __launch_bounds__(256)
void mix_swap_neighbors_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld,
                               const double* __restrict__ g_ukl, int64_t* __restrict__ g_labels,
                               unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_inv = reinterpret_cast<int*>(smem);        // state -> replica holding it
    int* s_lab = s_inv + K;
    for (int t = threadIdx.x; t < K; t += blockDim.x) s_inv[t] = -1;
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        int s = (int)g_labels[r];
        s_lab[r] = s;
        // np.where(...) in the reference picks the replica holding the state; states are a
        // permutation in replica exchange.  With duplicates the highest replica index wins
        // (same rule as oracle/mix_oracle.c).
        atomicMax(&s_inv[s], r);
    }
    __syncthreads();
    philox4 w0 = remd_philox(seed, REMD_STREAM_NEIGHBOR, 0u, 0u, (uint64_t)iteration);
    const int offset = (int)(w0.w[0] & 1u);                                              // :373
    const int n_pairs = (R - 1 - offset + 1) / 2;                                        // s < R-1
    for (int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
        const int s = offset + 2 * p;
        if (s >= R - 1) continue;
        const int ri = s_inv[s], rj = s_inv[s + 1];
        if (ri < 0 || rj < 0) continue;
        philox4 w = remd_philox(seed, REMD_STREAM_NEIGHBOR, (uint32_t)(1 + s), 0u, (uint64_t)iteration);
        const double u = remd_u53(w.w[2], w.w[3]);
        const int si = s, sj = s + 1;
        const double log_p = mix_logp(g_ukl, ld, ri, rj, si, sj);
        atomicAdd(&g_nprop[(size_t)si * K + sj], 1ull);
        atomicAdd(&g_nprop[(size_t)sj * K + si], 1ull);
        if (log_p >= 0.0 || u < remd_exp_det(log_p)) {
            s_lab[ri] = sj; s_lab[rj] = si;
            atomicAdd(&g_nacc[(size_t)si * K + sj], 1ull);
            atomicAdd(&g_nacc[(size_t)sj * K + si], 1ull);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) g_labels[r] = (int64_t)s_lab[r];
}

// sams.py:477-501 — one wavefront per replica
__global__ __launch_bounds__(64)
void sams_global_jump_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld,
                             const double* __restrict__ g_ukl, const double* __restrict__ g_logw,
                             int64_t* __restrict__ g_labels,
                             unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop,
                             double* __restrict__ g_logP)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_e = reinterpret_cast<double*>(smem);    // [K] then {lse}; all LDS in the dynamic region
    double* s_lse = s_e + K;
    const int r = blockIdx.x, lane = threadIdx.x;
    const int cur = (int)g_labels[r];
    double m = -INFINITY;
    for (int s = lane; s < K; s += 64) {
        double a = -g_ukl[(size_t)r * ld + s] + g_logw[s];                                // :488
        s_e[s] = a;
        m = fmax(m, a);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __syncthreads();
    for (int s = lane; s < K; s += 64) s_e[s] = remd_exp_det(s_e[s] - m);
    __syncthreads();
    if (lane == 0) {
        double tot = 0.0;
        for (int s = 0; s < K; ++s) tot = __dadd_rn(tot, s_e[s]);     // sequential, as the oracle
        philox4 w = remd_philox(seed, REMD_STREAM_SAMS, (uint32_t)r, 0u, (uint64_t)iteration);
        const double target = __dmul_rn(remd_u53(w.w[2], w.w[3]), tot);
        double cum = 0.0; int pick = K - 1;
        for (int s = 0; s < K; ++s) { cum = __dadd_rn(cum, s_e[s]); if (cum > target) { pick = s; break; } }
        *s_lse = m + log(tot);                                                            // :489
        g_labels[r] = pick;                                                              // :494
        atomicAdd(&g_nacc[(size_t)cur * K + pick], 1ull);                                // :500
    }
    __syncthreads();
    const double lse = *s_lse;
    for (int s = lane; s < K; s += 64) {
        g_logP[(size_t)r * K + s] = (-g_ukl[(size_t)r * ld + s] + g_logw[s]) - lse;       // :498
        atomicAdd(&g_nprop[(size_t)cur * K + s], 1ull);                                  // :499
    }
}

int remd_mix_launch(remd_ctx* h, int scheme, int64_t iteration, int R, int K, int ld, const double* d_ukl,
                    int64_t* d_labels, unsigned long long* d_nacc, unsigned long long* d_nprop,
                    const double* d_logw, double* d_logP, int64_t n_attempts)
{
    REMD_CHECK(h, hipMemsetAsync(d_nacc, 0, sizeof(unsigned long long) * K * K, h->stream));   // :261
    REMD_CHECK(h, hipMemsetAsync(d_nprop, 0, sizeof(unsigned long long) * K * K, h->stream));  // :262
    if (scheme == REMD_MIX_NONE) return 0;
    if (ld <= 0) ld = K;
    if (ld < K) return remd_fail(h, -3, "u_kl leading dimension smaller than K");
    if (scheme == REMD_MIX_SWAP_ALL) {
        if (R != K) return remd_fail(h, -3, "swap-all requires n_replicas == n_states");
        if (n_attempts < 0) n_attempts = (int64_t)R * R * R;                                   // :269
        size_t ukl_bytes = (size_t)R * K * sizeof(double);
        const size_t lab_bytes = sizeof(int) * (size_t)((R + 1) & ~1);
        const size_t stat_bytes = sizeof(unsigned) * 2 * (size_t)K * K;
        // counters (32-bit: <= 2 R^3 per entry) live in LDS when everything fits in 160 KB, else global atomics
        int in_lds = ukl_bytes <= MIX_MAX_LDS_UKL;
        int stats_lds = ((in_lds ? ukl_bytes : 0) + lab_bytes + stat_bytes <= 156 * 1024) && (2.0 * (double)n_attempts < 4.0e9);
        size_t lds = (in_lds ? ukl_bytes : 0) + lab_bytes + (stats_lds ? stat_bytes : 0);
        lds = (lds + 15) & ~(size_t)15;
        REMD_CHECK(h, hipFuncSetAttribute((const void*)mix_swap_all_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        remd_prof_scope ps(h, "mix_swap_all");
        hipLaunchKernelGGL(mix_swap_all_kernel, dim3(1), dim3(64), lds, h->stream,
                           h->seed, iteration, R, K, ld, d_ukl, d_labels, d_nacc, d_nprop, n_attempts, in_lds, stats_lds);
    } else if (scheme == REMD_MIX_SWAP_NEIGHBORS) {
        if (R != K) return remd_fail(h, -3, "swap-neighbors requires n_replicas == n_states");
        size_t lds = sizeof(int) * (size_t)(R + K);
        remd_prof_scope ps(h, "mix_swap_neighbors");
        hipLaunchKernelGGL(mix_swap_neighbors_kernel, dim3(1), dim3(256), lds, h->stream,
                           h->seed, iteration, R, K, ld, d_ukl, d_labels, d_nacc, d_nprop);
    } else if (scheme == REMD_MIX_SAMS_GLOBAL) {
        if (!d_logw || !d_logP) return remd_fail(h, -3, "SAMS mixing needs log_weights and log_P buffers");
        size_t lds = sizeof(double) * (size_t)(K + 2);
        remd_prof_scope ps(h, "sams_global_jump");
        hipLaunchKernelGGL(sams_global_jump_kernel, dim3(R), dim3(64), lds, h->stream,
                           h->seed, iteration, R, K, ld, d_ukl, d_logw, d_labels, d_nacc, d_nprop, d_logP);
    } else {
        return remd_fail(h, -3, "unknown mixing scheme");
    }
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
