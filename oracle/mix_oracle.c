/*
 * mix_oracle.c — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into or called by
 * the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 *
 * Plain-C, sequential restatement of the reference's Gibbs state-label mixing:
 *   - swap-all:        openmmtools/multistate/replicaexchange.py:294-349
 *                      (_mix_all_replicas_numba; same arithmetic as _attempt_swap :382-406)
 *   - swap-neighbors:  replicaexchange.py:366-380
 *   - SAMS global jump sams.py:477-501, logZ update sams.py:606-681, weights :683-691
 *
 * PARITY PINNING: the reference's RNG here is numba's private MT19937 (not seedable from
 * Python), and its tests pin this path only distributionally
 * (openmmtools/tests/test_mixing.py:76-92, chi-square uniformity).  So the *stream* is our
 * own spec (Philox4x32-10, published Random123 known-answer vectors checked in
 * tests/test_oracle_mix.py) while the *arithmetic per attempt* follows the reference line
 * by line; tests/golden/mix_reference_arith.json pins that arithmetic against a pure-Python
 * transcription driven by the same (i, j, u) sequence.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared mix_oracle.c -o _build/libmix_oracle.so -lm
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants) ------------ */
static void philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox(ctr, key, out); }

/* stream spec: counter = (a, b, (u32)t, stream ^ ((u32)(t>>32) << 8)), key = seed */
static void draw(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t t, uint32_t out[4])
{
    uint32_t ctr[4] = { a, b, (uint32_t)t, stream ^ ((uint32_t)(t >> 32) << 8) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    philox(ctr, key, out);
}
void oracle_draw(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t t, uint32_t* out)
{ draw(seed, stream, a, b, t, out); }

static double u53(uint32_t hi, uint32_t lo)
{
    uint64_t m = ((uint64_t)hi << 21) | (uint64_t)(lo >> 11);
    return (double)m / 9007199254740992.0;
}

/* exp() of the Metropolis test, fixed operation sequence (IEEE fma only) so that the
 * device kernel can reproduce it bit for bit; |rel err| < 2e-16 vs libm on [-700, 0].   */
double oracle_exp_det(double x)
{
    static const double c[14] = {
        1.0, 1.0, 0.5, 1.6666666666666666e-01, 4.1666666666666664e-02,
        8.333333333333333e-03, 1.388888888888889e-03, 1.984126984126984e-04,
        2.48015873015873e-05, 2.7557319223985893e-06, 2.755731922398589e-07,
        2.505210838544172e-08, 2.08767569878681e-09, 1.6059043836821613e-10 };
    if (!(x > -700.0)) return 0.0;
    if (x > 0.0) x = 0.0;
    double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = c[13];
    for (int n = 12; n >= 0; --n) p = fma(p, r, c[n]);
    return ldexp(p, (int)k);      /* exact scaling: result is normal for x > -700 */
}

/* replicaexchange.py:321-349 — one attempt on (i, j) with uniform u */
static void attempt_swap(int K, const double* u_kl, int64_t* labels, int64_t* n_acc, int64_t* n_prop,
                         int i, int j, double u)
{
    int64_t si = labels[i], sj = labels[j];                           /* :328-329 */
    double e_ij = u_kl[(size_t)i * K + sj];                           /* :332 */
    double e_ji = u_kl[(size_t)j * K + si];                           /* :333 */
    double e_ii = u_kl[(size_t)i * K + si];                           /* :334 */
    double e_jj = u_kl[(size_t)j * K + sj];                           /* :335 */
    double log_p = -(e_ij + e_ji) + e_ii + e_jj;                      /* :336 */
    n_prop[si * K + sj] += 1;                                         /* :339 */
    n_prop[sj * K + si] += 1;                                         /* :340 */
    if (log_p >= 0.0 || u < oracle_exp_det(log_p)) {                  /* :343 */
        labels[i] = sj;                                               /* :345 */
        labels[j] = si;                                               /* :346 */
        n_acc[si * K + sj] += 1;                                      /* :348 */
        n_acc[sj * K + si] += 1;                                      /* :349 */
    }
}

/* replicaexchange.py:261-262 (zero stats), :269 (nswap = R**3 by default), :321-349 */
void oracle_mix_swap_all(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl,
                         int64_t* labels, int64_t* n_acc, int64_t* n_prop, int64_t n_attempts)
{
    memset(n_acc, 0, sizeof(int64_t) * K * K);
    memset(n_prop, 0, sizeof(int64_t) * K * K);
    if (n_attempts < 0) n_attempts = (int64_t)R * R * R;
    for (int64_t k = 0; k < n_attempts; ++k) {
        uint32_t w[4];
        draw(seed, 1u, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration, w);
        int i = (int)(((uint64_t)w[0] * (uint64_t)R) >> 32);          /* :324 randint(R) */
        int j = (int)(((uint64_t)w[1] * (uint64_t)R) >> 32);          /* :325 */
        attempt_swap(K, u_kl, labels, n_acc, n_prop, i, j, u53(w[2], w[3]));
    }
}

/* same loop driven by a caller-supplied (i, j, u) sequence: used to pin the per-attempt
 * arithmetic against the pure-Python transcription of _attempt_swap                      */
void oracle_mix_swap_sequence(int R, int K, const double* u_kl, int64_t* labels, int64_t* n_acc,
                              int64_t* n_prop, int64_t n, const int32_t* ii, const int32_t* jj,
                              const double* uu)
{
    (void)R;
    memset(n_acc, 0, sizeof(int64_t) * K * K);
    memset(n_prop, 0, sizeof(int64_t) * K * K);
    for (int64_t k = 0; k < n; ++k) attempt_swap(K, u_kl, labels, n_acc, n_prop, ii[k], jj[k], uu[k]);
}

/* replicaexchange.py:366-380: offset in {0,1}; pairs of neighbouring STATES (s, s+1);
 * the replicas currently holding them are located (np.where) and _attempt_swap is applied */
void oracle_mix_swap_neighbors(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl,
                               int64_t* labels, int64_t* n_acc, int64_t* n_prop)
{
    memset(n_acc, 0, sizeof(int64_t) * K * K);
    memset(n_prop, 0, sizeof(int64_t) * K * K);
    uint32_t w[4];
    draw(seed, 2u, 0u, 0u, (uint64_t)iteration, w);
    int offset = (int)(w[0] & 1u);                                    /* :373 */
    for (int s = offset; s < R - 1; s += 2) {                         /* :374 */
        int ri = -1, rj = -1;
        for (int r = 0; r < R; ++r) {                                 /* :378-379 */
            if (labels[r] == s) ri = r;
            if (labels[r] == s + 1) rj = r;
        }
        if (ri < 0 || rj < 0) continue;
        draw(seed, 2u, (uint32_t)(1 + s), 0u, (uint64_t)iteration, w);
        attempt_swap(K, u_kl, labels, n_acc, n_prop, ri, rj, u53(w[2], w[3]));
    }
}

/* sams.py:477-501 (global neighbourhood: locality=None is forced, sams.py:338-339).
 * log_P_k = -u_k + log_w, normalised; new state drawn from P_k.  The draw is by inverse
 * CDF on sequentially accumulated exp terms (np.random.choice(p=...) semantics:
 * cdf = cumsum(p), first index with cdf > u*cdf[-1]).                                    */
void oracle_sams_global_jump(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl,
                             const double* log_w, int64_t* labels, int64_t* n_acc, int64_t* n_prop,
                             double* log_P /*[R][K]*/)
{
    memset(n_acc, 0, sizeof(int64_t) * K * K);
    memset(n_prop, 0, sizeof(int64_t) * K * K);
    double* e = (double*)malloc(sizeof(double) * K);
    for (int r = 0; r < R; ++r) {
        int64_t cur = labels[r];
        double m = -INFINITY;
        for (int s = 0; s < K; ++s) {                                 /* :486-488 */
            double a = -u_kl[(size_t)r * K + s] + log_w[s];
            log_P[(size_t)r * K + s] = a;
            if (a > m) m = a;
        }
        double tot = 0.0;
        for (int s = 0; s < K; ++s) { e[s] = oracle_exp_det(log_P[(size_t)r * K + s] - m); tot += e[s]; }
        double lse = m + log(tot);                                    /* :489 logsumexp */
        for (int s = 0; s < K; ++s) log_P[(size_t)r * K + s] -= lse;
        uint32_t w[4];
        draw(seed, 3u, (uint32_t)r, 0u, (uint64_t)iteration, w);
        double target = u53(w[2], w[3]) * tot;                        /* :493 choice(p=P_k) */
        double cum = 0.0; int pick = K - 1;
        for (int s = 0; s < K; ++s) { cum += e[s]; if (cum > target) { pick = s; break; } }
        labels[r] = pick;                                             /* :494 */
        for (int s = 0; s < K; ++s) n_prop[cur * K + s] += 1;         /* :499 */
        n_acc[cur * K + pick] += 1;                                   /* :500 */
    }
    free(e);
}

/* sams.py:606-681 rao-blackwellized / optimal update with global neighbourhoods.
 * stage, t0 handled by the caller (host python mirrors _update_stage :564-604).          */
void oracle_sams_update_logZ(int R, int K, const int64_t* labels, const double* log_P,
                             const double* log_pi, double gamma0, int64_t iteration, int stage,
                             int64_t t0, int optimal, double* logZ, double* gamma_out)
{
    double pi_star = INFINITY;
    for (int s = 0; s < K; ++s) { double p = exp(log_pi[s]); if (p < pi_star) pi_star = p; }
    double t = (double)iteration, beta_factor = 0.8, gamma = 0.0;
    for (int r = 0; r < R; ++r) {
        if (stage == 0) gamma = gamma0 * fmin(pi_star, pow(t, -beta_factor));           /* :637 */
        else            gamma = gamma0 * fmin(pi_star, 1.0 / (t - (double)t0 + pow((double)t0, beta_factor))); /* :639 */
        if (optimal) logZ[labels[r]] += gamma * exp(-log_pi[labels[r]]);                /* :649-652 */
        else for (int s = 0; s < K; ++s)                                                /* :659-664 */
            logZ[s] += gamma * exp(log_P[(size_t)r * K + s] - log_pi[s]);
    }
    if (stage == 1) { double z0 = logZ[0]; for (int s = 0; s < K; ++s) logZ[s] -= z0; } /* :669-670 */
    if (gamma_out) *gamma_out = gamma;
}
