// Gibbs state-label mixing kernels (gfx950).
//
// Reference semantics (restated, not copied):
//   swap-all        openmmtools/multistate/replicaexchange.py:294-349  (R^3 sequential attempts)
//   swap-neighbors  replicaexchange.py:366-380
//   SAMS global     openmmtools/multistate/sams.py:477-501
//
// The swap-all loop is a serial dependency chain.  It is parallelised *exactly* by speculation: a workgroup
// draws a window of blockDim.x consecutive attempts from the counter-based Philox stream; every attempt keeps a
// provisional accept/reject decision.  Per replica slot the window's attempts form an ordered chain (ordinal =
// number of earlier attempts of the window touching the slot, from occupancy bit masks), and one 32-bit word per
// slot holds the provisionally ACCEPTED attempts of its chain.  The label an attempt sees in slot s is found by
// hopping backwards: nearest earlier accepted attempt of the chain (one masked count-leading-zeros) -> its
// partner slot -> ... until no accepted attempt is left, which names the slot whose window-start label arrives.
// All attempts re-evaluate in sweeps until no decision changes; the dependency is triangular (attempt t only
// depends on t' < t), so the fixed point is unique and equals the sequential result, and attempt t is final
// after at most t+1 sweeps (in practice 2-4 sweeps at replica-exchange acceptance rates).  Then every attempt
// adds its counts and every slot resolves its new label.  The result is bit-identical to the sequential loop
// (tests/test_mix_parity.py against oracle/mix_oracle.c).  The window scales with R (conflict density ~4/R per
// attempt pair) and is cut short where a slot would collect more than 32 attempts.
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

#define MIX_CHAIN 32      // attempts of one window per replica slot (bits of the accepted word)

// workgroup barrier ordering LDS traffic only: the fire-and-forget global counter atomics must not be waited for
__device__ __forceinline__ void mix_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// number of window positions < t set in this slot's occupancy mask
__device__ __forceinline__ int mix_ordinal(const unsigned long long* touch_r, int t)
{
    int n = 0;
    const int wi = t >> 6;
    for (int q = 0; q < wi; ++q) n += __popcll(touch_r[q]);
    if (t & 63) n += __popcll(touch_r[wi] & ((1ull << (t & 63)) - 1ull));
    return n;
}

// slot whose window-start label sits in `slot` just before the chain position `ord` of that slot; `m` = the slot's
// accepted word already masked to positions < ord (loaded by the caller so that two walks can overlap their first read)
__device__ __forceinline__ int mix_origin(const unsigned* accb, const unsigned* chain, int slot, unsigned m)
{
    while (m) {
        const unsigned e = chain[slot * MIX_CHAIN + 31 - __clz((int)m)];    // nearest earlier accepted swap: go to its partner
        slot = (int)(e & 0xffffu);
        const int ord = (int)(e >> 16);
        m = accb[slot] & ((1u << ord) - 1u);       // ord < 32 for every chain entry
    }
    return slot;
}
__device__ __forceinline__ unsigned mix_below(int ord) { return ord >= 32 ? 0xffffffffu : ((1u << ord) - 1u); }

// No volatile LDS accesses here (they compile to serialised flat loads).  Every cross-thread hand-over is
// separated by a workgroup barrier, and reads that race with a same-sweep decision update are harmless: the
// update raises the changed flag, so another sweep follows.
template <bool UKL_LDS>
__global__ __launch_bounds__(1024)
void mix_swap_all_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld,
                         const double* __restrict__ g_ukl, int64_t* __restrict__ g_labels,
                         unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop,
                         int64_t n_attempts, int stats_in_lds, long long* __restrict__ g_dbg, unsigned int* __restrict__ g_log)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, W = blockDim.x, nw = W >> 6;
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = clock64(), w0 = wall_clock64();
#define MIX_TICK(slot) do { if (g_dbg) { long long t1 = clock64(); dbg[slot] += t1 - t0; t0 = t1; } } while (0)
    const int Rpad = (R + 1) & ~1;
    double* s_u = reinterpret_cast<double*>(smem);
    unsigned long long* s_touch = reinterpret_cast<unsigned long long*>(smem + (UKL_LDS ? (size_t)R * K * sizeof(double) : 0));   // [R][nw]
    unsigned* s_chain = reinterpret_cast<unsigned*>(s_touch + (size_t)R * nw);   // [R][32]: partner slot | partner ordinal << 16
    unsigned* s_accb = s_chain + (size_t)R * MIX_CHAIN;                          // [R] accepted attempts of the slot's chain
    int* s_lab = reinterpret_cast<int*>(s_accb + Rpad);                          // [2][Rpad]
    int* s_flag = s_lab + 2 * Rpad;          // [0..2] rotating "a decision changed" flags, [3] effective window length
    unsigned* s_nprop = reinterpret_cast<unsigned*>(s_flag + 4);                 // [K*K] when stats_in_lds
    unsigned* s_nacc = s_nprop + K * K;

    if (UKL_LDS)
        for (int t = tid; t < R * K; t += W) s_u[t] = g_ukl[(size_t)(t / K) * ld + (t % K)];
    for (int t = tid; t < R; t += W) s_lab[t] = (int)g_labels[t];
    for (int t = tid; t < R * nw; t += W) s_touch[t] = 0ull;
    if (tid < 4) s_flag[tid] = 0;
    if (stats_in_lds) for (int t = tid; t < 2 * K * K; t += W) s_nprop[t] = 0u;
    __syncthreads();
    const int ldu = UKL_LDS ? K : ld;
    int sweep = 0;      // running sweep counter: selects the rotating flag slot
    int cur_lab = 0;    // which half of s_lab holds the labels at window start

    MIX_TICK(0);
    for (int64_t base = 0; base < n_attempts;) {
        const int64_t k = base + tid;
        const bool active = k < n_attempts;
        philox4 w = remd_philox(seed, REMD_STREAM_SWAP_ALL, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration);
        const int i = (int)remd_mulhi32(w.w[0], (uint32_t)R);     // randint(R), replicaexchange.py:324
        const int j = (int)remd_mulhi32(w.w[1], (uint32_t)R);     // :325
        const double u = remd_u53(w.w[2], w.w[3]);
        const int* lab0 = s_lab + cur_lab * Rpad;
        int* lab1 = s_lab + (cur_lab ^ 1) * Rpad;

        if (active) {
            atomicOr(&s_touch[(size_t)i * nw + (tid >> 6)], 1ull << (tid & 63));
            atomicOr(&s_touch[(size_t)j * nw + (tid >> 6)], 1ull << (tid & 63));
        }
        for (int r = tid; r < R; r += W) s_accb[r] = 0u;
        if (tid == 0) s_flag[3] = W;
        mix_barrier();
        int ord_i = 0, ord_j = 0;
        if (active) {
            const unsigned long long* ti = s_touch + (size_t)i * nw;
            const unsigned long long* tj = s_touch + (size_t)j * nw;
            const int wi = tid >> 6;
            const unsigned long long low = (1ull << (tid & 63)) - 1ull;
            for (int q = 0; q < wi; ++q) { ord_i += __popcll(ti[q]); ord_j += __popcll(tj[q]); }
            ord_i += __popcll(ti[wi] & low); ord_j += __popcll(tj[wi] & low);
            if (ord_i >= MIX_CHAIN || ord_j >= MIX_CHAIN) atomicMin(&s_flag[3], tid);
        }
        mix_barrier();
        const int weff = s_flag[3];          // >= 1: the first attempt of a window always has ordinal 0
        const bool live = active && tid < weff;
        if (live && i != j) {
            s_chain[i * MIX_CHAIN + ord_i] = (unsigned)j | ((unsigned)ord_j << 16);
            s_chain[j * MIX_CHAIN + ord_j] = (unsigned)i | ((unsigned)ord_i << 16);
        }
        mix_barrier();
        MIX_TICK(1);

        bool acc = false;
        int a_seen = -1, b_seen = -1, si = 0, sj = 0;
        for (;; ++sweep) {
            if (live) {
                const unsigned ma = s_accb[i] & mix_below(ord_i), mb = s_accb[j] & mix_below(ord_j);
                const int a = mix_origin(s_accb, s_chain, i, ma);
                const int b = mix_origin(s_accb, s_chain, j, mb);
                if (a != a_seen || b != b_seen) {
                    a_seen = a; b_seen = b;
                    si = lab0[a]; sj = lab0[b];                                          // :328-329
                    const double log_p = UKL_LDS ? mix_logp(s_u, ldu, i, j, si, sj) : mix_logp(g_ukl, ldu, i, j, si, sj);   // :332-336
                    const bool nacc = log_p >= 0.0 || u < remd_exp_det(log_p);           // :343
                    if (nacc != acc) {
                        acc = nacc;
                        if (i != j) {
                            atomicXor(&s_accb[i], 1u << ord_i); atomicXor(&s_accb[j], 1u << ord_j);
                            s_flag[sweep % 3] = 1;
                        }
                    }
                }
            }
            if (tid == 0) s_flag[(sweep + 1) % 3] = 0;
            mix_barrier();
            dbg[5] += 1;
            if (!s_flag[sweep % 3]) { ++sweep; break; }
        }
        MIX_TICK(2);
        // decisions are final: counts (:339-340, :348-349) and the labels after the window (:345-346)
        if (live) {
            if (stats_in_lds) {
                atomicAdd(&s_nprop[si * K + sj], 1u); atomicAdd(&s_nprop[sj * K + si], 1u);
                if (acc) { atomicAdd(&s_nacc[si * K + sj], 1u); atomicAdd(&s_nacc[sj * K + si], 1u); }
            } else if (g_log) {
                // counters too large for LDS (R > ~128): one coalesced store per attempt instead of 2-4 global atomics from
                // this single CU (which also sat in front of the next window's u_kl loads in the in-order wait counter);
                // mix_stats_from_log_kernel builds the K x K matrices afterwards on the whole chip
                g_log[k] = (unsigned)si | ((unsigned)sj << 15) | (acc ? (1u << 30) : 0u) | (1u << 31);
            } else {
                atomicAdd(&g_nprop[(size_t)si * K + sj], 1ull); atomicAdd(&g_nprop[(size_t)sj * K + si], 1ull);
                if (acc) { atomicAdd(&g_nacc[(size_t)si * K + sj], 1ull); atomicAdd(&g_nacc[(size_t)sj * K + si], 1ull); }
            }
        }
        for (int r = tid; r < R; r += W)
            lab1[r] = lab0[mix_origin(s_accb, s_chain, r, s_accb[r] & mix_below(mix_ordinal(s_touch + (size_t)r * nw, weff)))];
        mix_barrier();
        for (int t = tid; t < R * nw; t += W) s_touch[t] = 0ull;
        cur_lab ^= 1;
        base += weff;
        mix_barrier();
        MIX_TICK(3);
    }
    if (g_dbg && tid == 0) { for (int q = 0; q < 8; ++q) g_dbg[q] = dbg[q]; g_dbg[6] = wall_clock64() - w0; }
    for (int t = tid; t < R; t += W) g_labels[t] = (int64_t)s_lab[cur_lab * Rpad + t];
    if (stats_in_lds)
        for (int t = tid; t < K * K; t += W) { g_nprop[t] = s_nprop[t]; g_nacc[t] = s_nacc[t]; }
}

// ---- swap-all as a dataflow ---------------------------------------------------------------------------------------
// The same window of consecutive attempts, but no speculation: an attempt runs when every earlier attempt of the window
// that touches one of its two replica slots has run.  Per slot ONE 64-bit LDS word holds (label, number of the window's
// attempts on this slot already executed); an attempt knows its ordinal in both slots' chains (occupancy masks, as above),
// polls the two words (one ds_read_b64 each: count and label arrive together), and when both counts equal its ordinals the
// labels it read are exactly those the sequential loop would see.  It decides, then publishes (new label, ordinal + 1) for
// both slots with one 64-bit store each.  Wavefronts run asynchronously inside a window (no barrier per dependency level):
// the cost of a level is two LDS round trips and the f64 arithmetic of one decision, whatever the acceptance rate -- the
// speculative kernel above needs one workgroup sweep per level of CHANGED decisions, which is cheap at parallel-tempering
// acceptance (few %) and 5-10x slower at the 40-50 % of an alchemical ladder.  Lower-numbered attempts never wait on
// higher-numbered ones, every wavefront of the workgroup is resident, so the polling cannot deadlock.
__device__ __forceinline__ unsigned long long mix_slot_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mix_slot_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool UKL_LDS>
__global__ __launch_bounds__(1024)
void mix_swap_all_dataflow_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld,
                                  const double* __restrict__ g_ukl, int64_t* __restrict__ g_labels,
                                  unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop,
                                  int64_t n_attempts, int stats_in_lds, long long* __restrict__ g_dbg, unsigned int* __restrict__ g_log)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, W = blockDim.x, nw = W >> 6;
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = clock64(), w0 = wall_clock64();
    double* s_u = reinterpret_cast<double*>(smem);
    unsigned long long* s_touch = reinterpret_cast<unsigned long long*>(smem + (UKL_LDS ? (size_t)R * K * sizeof(double) : 0));   // [R][nw]
    unsigned long long* s_slot = s_touch + (size_t)R * nw;                       // [R] label | executed attempts of the window << 32
    unsigned* s_nprop = reinterpret_cast<unsigned*>(s_slot + R);                 // [K*K] when stats_in_lds
    unsigned* s_nacc = s_nprop + K * K;

    if (UKL_LDS)
        for (int t = tid; t < R * K; t += W) s_u[t] = g_ukl[(size_t)(t / K) * ld + (t % K)];
    for (int t = tid; t < R; t += W) s_slot[t] = (unsigned long long)(unsigned)g_labels[t];
    for (int t = tid; t < R * nw; t += W) s_touch[t] = 0ull;
    if (stats_in_lds) for (int t = tid; t < 2 * K * K; t += W) s_nprop[t] = 0u;
    __syncthreads();
    const int ldu = UKL_LDS ? K : ld;
    MIX_TICK(0);
    for (int64_t base = 0; base < n_attempts; base += W) {
        const int64_t k = base + tid;
        const bool active = k < n_attempts;
        philox4 w = remd_philox(seed, REMD_STREAM_SWAP_ALL, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration);
        const int i = (int)remd_mulhi32(w.w[0], (uint32_t)R);     // randint(R), replicaexchange.py:324
        const int j = (int)remd_mulhi32(w.w[1], (uint32_t)R);     // :325
        const double u = remd_u53(w.w[2], w.w[3]);
        if (active) {
            atomicOr(&s_touch[(size_t)i * nw + (tid >> 6)], 1ull << (tid & 63));
            atomicOr(&s_touch[(size_t)j * nw + (tid >> 6)], 1ull << (tid & 63));
        }
        mix_barrier();
        unsigned ord_i = 0, ord_j = 0;
        if (active) {
            const unsigned long long* ti = s_touch + (size_t)i * nw;
            const unsigned long long* tj = s_touch + (size_t)j * nw;
            const int wi = tid >> 6;
            const unsigned long long low = (1ull << (tid & 63)) - 1ull;
            for (int q = 0; q < wi; ++q) { ord_i += __popcll(ti[q]); ord_j += __popcll(tj[q]); }
            ord_i += __popcll(ti[wi] & low); ord_j += __popcll(tj[wi] & low);
        }
        MIX_TICK(1);
        // u < exp(log_p) is decided by log_p - log(u) wherever that is not a borderline case (|difference| > 1e-9, far above
        // the error of either function); the borderline falls back to the deterministic exp the oracle uses, so the decision
        // is the oracle's bit for bit while the 18 dependent f64 operations of that exp stay off the dependency chain
        const double lu = log(u);
        const double* row_i = (UKL_LDS ? s_u : g_ukl) + (size_t)i * ldu;
        const double* row_j = (UKL_LDS ? s_u : g_ukl) + (size_t)j * ldu;
        const unsigned long long hi_i = (unsigned long long)(ord_i + 1u) << 32, hi_j = (unsigned long long)(ord_j + 1u) << 32;
        unsigned long long* slot_i = s_slot + i;
        unsigned long long* slot_j = s_slot + j;
        bool pending = active, acc = false;
        unsigned si = 0, sj = 0;
        while (__ballot(pending)) {
            if (pending) {
                const unsigned long long vi = mix_slot_load(slot_i), vj = mix_slot_load(slot_j);
                if ((unsigned)(vi >> 32) == ord_i && (unsigned)(vj >> 32) == ord_j) {
                    si = (unsigned)vi; sj = (unsigned)vj;                                              // :328-329
                    // replicaexchange.py:332-336, same association as mix_logp: (-(e_ij + e_ji) + e_ii) + e_jj
                    const double e_ij = row_i[sj], e_ji = row_j[si], e_ii = row_i[si], e_jj = row_j[sj];
                    const double log_p = __dadd_rn(__dadd_rn(-__dadd_rn(e_ij, e_ji), e_ii), e_jj);
                    const double d = log_p - lu;
                    if (log_p >= 0.0) acc = true;                                                      // :343
                    else if (log_p > -690.0 && d > 1e-9 && d < 1e300) acc = true;                      // (d = inf: u = 0, exact path)
                    else if (log_p > -690.0 && d < -1e-9) acc = false;
                    else acc = u < remd_exp_det(log_p);
                    mix_slot_store(slot_i, (unsigned long long)(acc ? sj : si) | hi_i);                // :345-346 (i == j: si == sj)
                    if (i != j) mix_slot_store(slot_j, (unsigned long long)(acc ? si : sj) | hi_j);
                    pending = false;
                }
            }
            dbg[5] += 1;
        }
        if (active) {                                    // counts (:339-340, :348-349), off the dependency chain
            if (stats_in_lds) {
                atomicAdd(&s_nprop[si * K + sj], 1u); atomicAdd(&s_nprop[sj * K + si], 1u);
                if (acc) { atomicAdd(&s_nacc[si * K + sj], 1u); atomicAdd(&s_nacc[sj * K + si], 1u); }
            } else if (g_log) {
                g_log[k] = (unsigned)si | ((unsigned)sj << 15) | (acc ? (1u << 30) : 0u) | (1u << 31);
            } else {
                atomicAdd(&g_nprop[(size_t)si * K + sj], 1ull); atomicAdd(&g_nprop[(size_t)sj * K + si], 1ull);
                if (acc) { atomicAdd(&g_nacc[(size_t)si * K + sj], 1ull); atomicAdd(&g_nacc[(size_t)sj * K + si], 1ull); }
            }
        }
        MIX_TICK(2);
        mix_barrier();                                   // every attempt of the window has run
        for (int t = tid; t < R * nw; t += W) s_touch[t] = 0ull;
        for (int t = tid; t < R; t += W) s_slot[t] &= 0xffffffffull;          // keep the labels, restart the counts
        mix_barrier();
        MIX_TICK(3);
    }
    if (g_dbg && tid == 0) { for (int q = 0; q < 8; ++q) g_dbg[q] = dbg[q]; g_dbg[6] = wall_clock64() - w0; }
    for (int t = tid; t < R; t += W) g_labels[t] = (int64_t)(unsigned)s_slot[t];
    if (stats_in_lds)
        for (int t = tid; t < K * K; t += W) { g_nprop[t] = s_nprop[t]; g_nacc[t] = s_nacc[t]; }
}

// ---- swap-all, label-independent part hoisted (round 4) ---------------------------------------------------------------
// Everything a window of attempts needs except the labels is a function of (seed, iteration, attempt index) alone: the
// Philox draws (i, j, u), the attempts' ordinals in their two slots' chains, where a window has to be cut because a slot
// would collect more than MIX_CHAIN attempts, and how many attempts each slot sees.  mix_prep_kernel computes that for
// ALL windows at once on the whole chip (one workgroup per window of W attempts) and leaves one 16-byte record per
// attempt; the serial workgroup (mix_swap_all_pre_kernel) is left with what really is sequential: building the chain
// table of a window from the records (two LDS stores per attempt), the speculative sweeps, and the label update.
// Acceptance is decided by log_p - log u away from the borderline, exactly as the dataflow kernel does (the uniform is
// regenerated from its Philox counter for the rare borderline attempt), and every attempt's (s_i, s_j, accepted) goes to
// the attempt log from which mix_stats_from_log_kernel builds the count matrices on the whole chip.
// A window of W attempts is processed as 1 ... MIXP pieces (cut where a slot's chain is full; more than one piece per
// window is rare: P(Poisson(12) >= 32) ~ 1e-6 per slot and window); a window that would need more than MIXP pieces raises
// g_err and the host runs the one-kernel path instead.
#define MIXP 4
struct mix_rec { unsigned ij, ords, lu_lo, lu_hi; };       // i | j << 16 ; ord_i | ord_j << 8 ; log(u) as two words

__global__ __launch_bounds__(1024)
void mix_prep_kernel(uint64_t seed, int64_t iteration, int R, int64_t n_attempts, uint4* __restrict__ g_rec,
                     unsigned* __restrict__ g_piece /*[n_win][MIXP]: start | len << 16*/, unsigned* __restrict__ g_npiece,
                     unsigned char* __restrict__ g_cnt /*[n_win][MIXP][Rpad]*/, int Rpad, unsigned* __restrict__ g_err)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, W = blockDim.x, nw = W >> 6;
    unsigned long long* s_touch = reinterpret_cast<unsigned long long*>(smem);      // [R][nw]
    __shared__ int s_cut;
    const int64_t win = blockIdx.x;
    const int64_t k = win * W + tid;
    const bool active = k < n_attempts;
    const int n_in = (int)min((int64_t)W, n_attempts - win * W);
    philox4 w = remd_philox(seed, REMD_STREAM_SWAP_ALL, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration);
    const int i = (int)remd_mulhi32(w.w[0], (uint32_t)R);     // randint(R), replicaexchange.py:324
    const int j = (int)remd_mulhi32(w.w[1], (uint32_t)R);     // :325
    const double lu = log(remd_u53(w.w[2], w.w[3]));           // u = 0: -inf (the serial kernel takes the exact path)
    int start = 0, np = 0;
    while (start < n_in) {
        for (int t = tid; t < R * nw; t += W) s_touch[t] = 0ull;
        if (tid == 0) s_cut = n_in;
        __syncthreads();
        const bool in_piece = active && tid >= start;
        if (in_piece) {
            atomicOr(&s_touch[(size_t)i * nw + (tid >> 6)], 1ull << (tid & 63));
            atomicOr(&s_touch[(size_t)j * nw + (tid >> 6)], 1ull << (tid & 63));
        }
        __syncthreads();
        int ord_i = 0, ord_j = 0;
        if (in_piece) {
            const unsigned long long* ti = s_touch + (size_t)i * nw;
            const unsigned long long* tj = s_touch + (size_t)j * nw;
            const int wi = tid >> 6;
            const unsigned long long low = (1ull << (tid & 63)) - 1ull;
            for (int q = 0; q < wi; ++q) { ord_i += __popcll(ti[q]); ord_j += __popcll(tj[q]); }
            ord_i += __popcll(ti[wi] & low); ord_j += __popcll(tj[wi] & low);
            if (ord_i >= MIX_CHAIN || ord_j >= MIX_CHAIN) atomicMin(&s_cut, tid);
        }
        __syncthreads();
        const int cut = s_cut;                   // > start: the first attempt of a piece has ordinal 0 in both chains
        if (np < MIXP) {
            if (in_piece && tid < cut)
                g_rec[k] = make_uint4((unsigned)i | ((unsigned)j << 16), (unsigned)ord_i | ((unsigned)ord_j << 8),
                                      (unsigned)__double2loint(lu), (unsigned)__double2hiint(lu));
            // attempts of this piece per slot (= the chain position behind its last one)
            for (int r = tid; r < R; r += W) {
                int n = 0;
                const unsigned long long* tr = s_touch + (size_t)r * nw;
                const int wc = cut >> 6;
                for (int q = 0; q < wc; ++q) n += __popcll(tr[q]);
                if (cut & 63) n += __popcll(tr[wc] & ((1ull << (cut & 63)) - 1ull));
                g_cnt[((size_t)win * MIXP + np) * Rpad + r] = (unsigned char)n;
            }
            if (tid == 0) g_piece[win * MIXP + np] = (unsigned)start | ((unsigned)(cut - start) << 16);
        } else if (tid == 0) atomicExch(g_err, 1u);
        ++np;
        start = cut;
        __syncthreads();
    }
    if (tid == 0) g_npiece[win] = (unsigned)min(np, MIXP);
}

template <bool UKL_LDS>
__global__ __launch_bounds__(1024)
void mix_swap_all_pre_kernel(uint64_t seed, int64_t iteration, int R, int K, int ld, const double* __restrict__ g_ukl,
                             int64_t* __restrict__ g_labels, int64_t n_attempts, const uint4* __restrict__ g_rec,
                             const unsigned* __restrict__ g_piece, const unsigned* __restrict__ g_npiece,
                             const unsigned char* __restrict__ g_cnt, int Rpad_cnt, unsigned int* __restrict__ g_log,
                             long long* __restrict__ g_dbg, const unsigned* __restrict__ g_err)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (a window of mix_prep_kernel needed more than MIXP pieces: nothing is touched here, the host runs the one-kernel path when it
    // reads the flag at the end of the call -- no host round trip between the two launches)
    if (*g_err) return;
    const int tid = threadIdx.x, W = blockDim.x;
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = clock64(), w0 = wall_clock64();
    const int Rpad = (R + 1) & ~1;
    double* s_u = reinterpret_cast<double*>(smem);
    unsigned* s_chain = reinterpret_cast<unsigned*>(smem + (UKL_LDS ? (size_t)R * K * sizeof(double) : 0));   // [R][32]
    unsigned* s_accb = s_chain + (size_t)R * MIX_CHAIN;                          // [R]
    int* s_lab = reinterpret_cast<int*>(s_accb + Rpad);                          // [2][Rpad]
    int* s_flag = s_lab + 2 * Rpad;                                              // [0..2] rotating "a decision changed" flags
    if (UKL_LDS)
        for (int t = tid; t < R * K; t += W) s_u[t] = g_ukl[(size_t)(t / K) * ld + (t % K)];
    for (int t = tid; t < R; t += W) s_lab[t] = (int)g_labels[t];
    if (tid < 4) s_flag[tid] = 0;
    __syncthreads();
    const int ldu = UKL_LDS ? K : ld;
    const double* U = UKL_LDS ? s_u : g_ukl;
    const int64_t n_win = (n_attempts + W - 1) / W;
    int sweep = 0, cur_lab = 0;
    MIX_TICK(0);
    // records of the next window's first piece travel while this window is decided
    unsigned pc_next = n_win > 0 ? g_piece[0] : 0u, np_next = n_win > 0 ? g_npiece[0] : 0u;
    uint4 rec_next = make_uint4(0u, 0u, 0u, 0u);
    if (n_win > 0 && tid < (int)(pc_next >> 16)) rec_next = g_rec[tid];
    for (int64_t win = 0; win < n_win; ++win) {
        const unsigned np = np_next;
        unsigned pc = pc_next;
        uint4 rec = rec_next;
        if (win + 1 < n_win) {
            pc_next = g_piece[(win + 1) * MIXP]; np_next = g_npiece[win + 1];
            if (tid < (int)(pc_next >> 16)) rec_next = g_rec[(win + 1) * W + tid];
        }
        for (unsigned p = 0; p < np; ++p) {
            if (p > 0) {
                pc = g_piece[win * MIXP + p];
                if (tid < (int)(pc >> 16)) rec = g_rec[win * W + (pc & 0xffffu) + tid];
            }
            const int len = (int)(pc >> 16);
            const int64_t k = win * W + (int)(pc & 0xffffu) + tid;
            const bool live = tid < len;
            const int i = (int)(rec.x & 0xffffu), j = (int)(rec.x >> 16);
            const int ord_i = (int)(rec.y & 0xffu), ord_j = (int)((rec.y >> 8) & 0xffu);
            const double lu = __hiloint2double((int)rec.w, (int)rec.z);
            const int* lab0 = s_lab + cur_lab * Rpad;
            int* lab1 = s_lab + (cur_lab ^ 1) * Rpad;
            for (int r = tid; r < R; r += W) s_accb[r] = 0u;
            if (live && i != j) {
                s_chain[i * MIX_CHAIN + ord_i] = (unsigned)j | ((unsigned)ord_j << 16);
                s_chain[j * MIX_CHAIN + ord_j] = (unsigned)i | ((unsigned)ord_i << 16);
            }
            mix_barrier();
            MIX_TICK(1);
            bool acc = false;
            int a_seen = -1, b_seen = -1, si = 0, sj = 0;
            for (;; ++sweep) {
                if (live) {
                    const unsigned ma = s_accb[i] & mix_below(ord_i), mb = s_accb[j] & mix_below(ord_j);
                    const int a = mix_origin(s_accb, s_chain, i, ma);
                    const int b = mix_origin(s_accb, s_chain, j, mb);
                    if (a != a_seen || b != b_seen) {
                        a_seen = a; b_seen = b;
                        si = lab0[a]; sj = lab0[b];                                          // :328-329
                        const double log_p = mix_logp(U, ldu, i, j, si, sj);                 // :332-336
                        const double d = log_p - lu;
                        bool nacc;
                        if (log_p >= 0.0) nacc = true;                                       // :343
                        else if (!(log_p > -700.0)) nacc = false;                            // remd_exp_det gives 0: u < 0 never holds
                        else if (d > 1e-9 && d < 1e300) nacc = true;                         // (d = inf: u = 0, exact path)
                        else if (d < -1e-9) nacc = false;
                        else {
                            const philox4 w = remd_philox(seed, REMD_STREAM_SWAP_ALL, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), (uint64_t)iteration);
                            nacc = remd_u53(w.w[2], w.w[3]) < remd_exp_det(log_p);
                        }
                        if (nacc != acc) {
                            acc = nacc;
                            if (i != j) {
                                atomicXor(&s_accb[i], 1u << ord_i); atomicXor(&s_accb[j], 1u << ord_j);
                                s_flag[sweep % 3] = 1;
                            }
                        }
                    }
                }
                if (tid == 0) s_flag[(sweep + 1) % 3] = 0;
                mix_barrier();
                dbg[5] += 1;
                if (!s_flag[sweep % 3]) { ++sweep; break; }
            }
            MIX_TICK(2);
            if (live) g_log[k] = (unsigned)si | ((unsigned)sj << 15) | (acc ? (1u << 30) : 0u) | (1u << 31);
            const unsigned char* cnt = g_cnt + ((size_t)win * MIXP + p) * Rpad_cnt;
            for (int r = tid; r < R; r += W)
                lab1[r] = lab0[mix_origin(s_accb, s_chain, r, s_accb[r] & mix_below((int)cnt[r]))];
            cur_lab ^= 1;
            mix_barrier();
            MIX_TICK(3);
        }
    }
    if (g_dbg && tid == 0) { for (int q = 0; q < 8; ++q) g_dbg[q] = dbg[q]; g_dbg[6] = wall_clock64() - w0; }
    for (int t = tid; t < R; t += W) g_labels[t] = (int64_t)s_lab[cur_lab * Rpad + t];
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

// proposed / accepted state-pair counts (replicaexchange.py:339-340, 348-349) from the attempt log of mix_swap_all_kernel
__global__ __launch_bounds__(256)
void mix_stats_from_log_kernel(int64_t n_attempts, const unsigned int* __restrict__ log, int K,
                               unsigned long long* __restrict__ g_nacc, unsigned long long* __restrict__ g_nprop,
                               const unsigned* __restrict__ g_err = nullptr)
{
    if (g_err && *g_err) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n_attempts; k += (int64_t)gridDim.x * 256) {
        const unsigned int e = log[k];
        if (!(e >> 31)) continue;
        const int si = (int)(e & 0x7fffu), sj = (int)((e >> 15) & 0x7fffu);
        atomicAdd(&g_nprop[(size_t)si * K + sj], 1ull); atomicAdd(&g_nprop[(size_t)sj * K + si], 1ull);
        if ((e >> 30) & 1u) { atomicAdd(&g_nacc[(size_t)si * K + sj], 1ull); atomicAdd(&g_nacc[(size_t)sj * K + si], 1ull); }
    }
}

// buffers of the hoisted swap-all path, per handle (grow-only)
struct mix_pre_buffers {
    uint4* rec = nullptr; size_t rec_n = 0;
    unsigned* piece = nullptr; unsigned* npiece = nullptr; size_t win_n = 0;
    unsigned char* cnt = nullptr; size_t cnt_n = 0;
    unsigned* err = nullptr;
    long long* dbg = nullptr;             // REMD_MIX_DEBUG phase counters (per handle: handles may live on different devices)
    ~mix_pre_buffers() { hipFree(rec); hipFree(piece); hipFree(npiece); hipFree(cnt); hipFree(err); hipFree(dbg); }
};
static handle_table<mix_pre_buffers> g_mix_pre;
void remd_mix_release(remd_ctx* h) { g_mix_pre.erase(h); }
// the overflow flag of the hoisted swap-all path launched last (device pointer; NULL: that path did not run): api.hip reads it with
// the results, at the call's one synchronisation, and repeats the call on the one-kernel path when it is raised
const unsigned* remd_mix_pending_flag(remd_ctx* h)
{
    mix_pre_buffers* B = g_mix_pre.find(h);
    return (B && h->mix_pre_launched) ? B->err : nullptr;
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
        if (R > 32767) return remd_fail(h, -3, "swap-all supports at most 32767 replicas");
        // window = 6R attempts (conflict probability between two attempts ~4/R), whole wavefronts, <= 768 threads (measured optimum)
        const int force_waves = 0;
        int waves = force_waves > 0 ? force_waves : (4 * R + 63) / 64;
        waves = std::max(1, std::min(16, waves));
        // window of 6 R attempts (21.9 ms at R = 128: DESIGN.md 7b; the dataflow kernel: 43.9 ms, 45.5 at 4 R, 52 at 2 R); REMD_MIX_PERR: experiments
        const int per_r = getenv("REMD_MIX_PERR") ? std::max(1, atoi(getenv("REMD_MIX_PERR"))) : 6;
        const int max_waves = 12;
        if (force_waves <= 0) waves = std::max(1, std::min(max_waves, (per_r * R + 63) / 64));
        while (waves > 1 && (size_t)R * waves * 8 > 64 * 1024) --waves;      // occupancy masks [R][waves] u64
        const size_t Rp = (size_t)((R + 1) & ~1);
        auto work_for = [&](int wv) { return 8 * (size_t)R * wv + 4 * (size_t)R * 32 + 4 * Rp + 8 * Rp + 16; };
        // a slightly shorter window (>= 5R attempts) that lets u_kl stay in LDS beats a longer one reading it from L2
        // (R = 128: 10 wavefronts, 21.9 ms instead of 25.0 ms with 12; measured, tools/mix_microbench.py)
        if (force_waves <= 0 && ukl_bytes + work_for(waves) > 156 * 1024)
            for (int wv = waves - 1; wv >= std::max(1, (5 * R + 63) / 64); --wv)
                if (ukl_bytes + work_for(wv) <= 156 * 1024) { waves = wv; break; }
        const size_t work_bytes = work_for(waves);
        const size_t stat_bytes = sizeof(unsigned) * 2 * (size_t)K * K;
        // u_kl, then the counters (32-bit: <= 2 R^3 per entry), live in LDS when they fit in 160 KB; else global memory
        int in_lds = ukl_bytes + work_bytes <= 156 * 1024;
        int stats_lds = ((in_lds ? ukl_bytes : 0) + work_bytes + stat_bytes <= 156 * 1024) && (2.0 * (double)n_attempts < 4.0e9);
        size_t lds = (in_lds ? ukl_bytes : 0) + work_bytes + (stats_lds ? stat_bytes : 0);
        lds = (lds + 15) & ~(size_t)15;
        if (lds > 160 * 1024) return remd_fail(h, -3, "swap-all working set does not fit in LDS");
        // which kernel: both give the sequential loop's result bit for bit.  Speculative windows are 2-6x faster while few
        // attempts are accepted (parallel tempering: 6 % -> 22 ms at R = 128 against 43), the dataflow kernel does not care
        // (43 ms) where speculation degrades (45 % on an alchemical ladder -> 116 ms); the acceptance of the handle's previous
        // swap-all call decides, crossover measured at 15-20 % for R = 16 ... 128 (profiles/r03_g_mix_kernels.txt).
        // REMD_MIX_FLOW = 0 / 1 pins the kernel (parity tests).
        const char* flow_env = getenv("REMD_MIX_FLOW");
        const int flow = flow_env ? atoi(flow_env) : (h->mix_acc_rate > 0.17 ? 1 : 0);
        static const bool debug = getenv("REMD_MIX_DEBUG") != nullptr;
        long long* d_dbg = nullptr;
        if (debug) {
            mix_pre_buffers& Bd = g_mix_pre[h];
            if (!Bd.dbg) REMD_CHECK(h, hipMalloc(&Bd.dbg, 8 * sizeof(long long)));
            d_dbg = Bd.dbg;
        }
        h->mix_pre_launched = false;
        // speculative windows with the label-independent part hoisted into a whole-chip kernel (mix_prep_kernel); REMD_MIX_PRE=0
        // keeps everything in the one serial workgroup (parity tests run both)
        const bool pre_off = getenv("REMD_MIX_PRE") && atoi(getenv("REMD_MIX_PRE")) == 0;
        if (!flow && !pre_off && !h->mix_no_pre && n_attempts > 0 && (size_t)n_attempts * sizeof(unsigned int) <= ((size_t)1 << 30)) {
            const int W = 64 * waves;
            const int64_t n_win = (n_attempts + W - 1) / W;
            const int Rp4 = (R + 3) & ~3;
            mix_pre_buffers& B = g_mix_pre[h];
            if (B.rec_n < (size_t)n_win * W) { hipFree(B.rec); B.rec = nullptr; REMD_CHECK(h, hipMalloc(&B.rec, sizeof(uint4) * (size_t)n_win * W)); B.rec_n = (size_t)n_win * W; }
            if (B.win_n < (size_t)n_win) {
                hipFree(B.piece); hipFree(B.npiece); B.piece = B.npiece = nullptr;
                REMD_CHECK(h, hipMalloc(&B.piece, sizeof(unsigned) * (size_t)n_win * MIXP));
                REMD_CHECK(h, hipMalloc(&B.npiece, sizeof(unsigned) * (size_t)n_win));
                B.win_n = (size_t)n_win;
            }
            if (B.cnt_n < (size_t)n_win * MIXP * Rp4) { hipFree(B.cnt); B.cnt = nullptr; REMD_CHECK(h, hipMalloc(&B.cnt, (size_t)n_win * MIXP * Rp4)); B.cnt_n = (size_t)n_win * MIXP * Rp4; }
            if (!B.err) REMD_CHECK(h, hipMalloc(&B.err, sizeof(unsigned)));
            if (h->mix_log_n < (size_t)n_attempts) {
                if (h->d_mix_log) { hipFree(h->d_mix_log); h->d_mix_log = nullptr; h->mix_log_n = 0; }
                REMD_CHECK(h, hipMalloc(&h->d_mix_log, sizeof(unsigned int) * (size_t)n_attempts));
                h->mix_log_n = (size_t)n_attempts;
            }
            remd_prof_scope ps(h, "mix_swap_all");
            REMD_CHECK(h, hipMemsetAsync(B.err, 0, sizeof(unsigned), h->stream));
            hipLaunchKernelGGL(mix_prep_kernel, dim3((unsigned)n_win), dim3(W), sizeof(unsigned long long) * (size_t)R * waves, h->stream,
                               h->seed, iteration, R, n_attempts, B.rec, B.piece, B.npiece, B.cnt, Rp4, B.err);
            {
                const size_t pre_work = 4 * (size_t)R * MIX_CHAIN + 4 * Rp + 8 * Rp + 16;
                const int pre_in_lds = ukl_bytes + pre_work <= 156 * 1024;
                size_t pre_lds = ((pre_in_lds ? ukl_bytes : 0) + pre_work + 15) & ~(size_t)15;
                auto pk = pre_in_lds ? mix_swap_all_pre_kernel<true> : mix_swap_all_pre_kernel<false>;
                REMD_CHECK(h, hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre_lds));
                hipLaunchKernelGGL(pk, dim3(1), dim3(W), pre_lds, h->stream, h->seed, iteration, R, K, ld, d_ukl, d_labels, n_attempts,
                                   (const uint4*)B.rec, (const unsigned*)B.piece, (const unsigned*)B.npiece, (const unsigned char*)B.cnt, Rp4,
                                   h->d_mix_log, d_dbg, (const unsigned*)B.err);
                hipLaunchKernelGGL(mix_stats_from_log_kernel, dim3((unsigned)std::min<int64_t>(4096, (n_attempts + 255) / 256)), dim3(256), 0, h->stream,
                                   n_attempts, h->d_mix_log, K, d_nacc, d_nprop, (const unsigned*)B.err);
                h->mix_pre_launched = true;
                if (debug) {
                    long long hd[8];
                    REMD_CHECK(h, hipStreamSynchronize(h->stream));
                    REMD_CHECK(h, hipMemcpy(hd, d_dbg, sizeof(hd), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[mix-pre] R=%d W=%d lds=%zu ukl_lds=%d cycles: setup %lld chain-build %lld sweeps %lld finalize %lld n_sweeps %lld "
                            "wall(100MHz ticks) %lld\n", R, W, pre_lds, pre_in_lds, hd[0], hd[1], hd[2], hd[3], hd[5], hd[6]);
                }
                REMD_CHECK(h, hipGetLastError());
                return 0;
            }
        }
        auto kern = flow ? (in_lds ? mix_swap_all_dataflow_kernel<true> : mix_swap_all_dataflow_kernel<false>)
                         : (in_lds ? mix_swap_all_kernel<true> : mix_swap_all_kernel<false>);
        REMD_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        remd_prof_scope ps(h, "mix_swap_all");
        unsigned int* d_log = nullptr;
        const bool use_log = true;              // counters beyond the LDS: attempt log instead of global atomics
        if (!stats_lds && use_log && (size_t)n_attempts * sizeof(unsigned int) <= ((size_t)1 << 30)) {
            if (h->mix_log_n < (size_t)n_attempts) {
                if (h->d_mix_log) { hipFree(h->d_mix_log); h->d_mix_log = nullptr; h->mix_log_n = 0; }
                REMD_CHECK(h, hipMalloc(&h->d_mix_log, sizeof(unsigned int) * (size_t)n_attempts));
                h->mix_log_n = (size_t)n_attempts;
            }
            d_log = h->d_mix_log;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), lds, h->stream,
                           h->seed, iteration, R, K, ld, d_ukl, d_labels, d_nacc, d_nprop, n_attempts, stats_lds, d_dbg, d_log);
        if (d_log)
            hipLaunchKernelGGL(mix_stats_from_log_kernel, dim3((unsigned)std::min<int64_t>(4096, (n_attempts + 255) / 256)), dim3(256), 0, h->stream,
                               n_attempts, d_log, K, d_nacc, d_nprop);
        if (debug) {
            long long hd[8];
            REMD_CHECK(h, hipStreamSynchronize(h->stream));
            REMD_CHECK(h, hipMemcpy(hd, d_dbg, sizeof(hd), hipMemcpyDeviceToHost));
            fprintf(stderr, "[mix] R=%d waves=%d lds=%zu ukl_lds=%d stats_lds=%d cycles: setup %lld window-prep %lld sweeps %lld finalize %lld "
                    "n_sweeps %lld wall(100MHz ticks) %lld\n", R, waves, lds, in_lds, stats_lds, hd[0], hd[1], hd[2], hd[3], hd[5], hd[6]);
        }
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
