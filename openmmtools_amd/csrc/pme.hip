// Smooth particle-mesh Ewald reciprocal space (Essmann et al. 1995), batched over replicas (gfx950).
//
// Pipeline per force evaluation, all replicas at once (3 FFT launches):
//   memset(mesh) -> spread (order-5 B-splines, 32-bit fixed-point atomics => reproducible charge mesh)
//   -> z forward: 8 real lines per workgroup -> half spectrum, stored kz-major as planes spec[kz][x][y]
//   -> xy fused: one (kz, replica) plane resident in LDS (2 x nx x ny x 8 B <= 160 KiB on gfx950):
//      forward y, forward x, influence function (+ energy), inverse x, inverse y — one read + one write per point
//   -> z inverse: Hermitian completion in LDS, real potential mesh -> gather forces (125 points per atom).
// Meshes whose plane exceeds the LDS fall back to separate strided passes.  The 1-D FFTs are in-tree
// mixed-radix (2,3,4,5) Stockham transforms.
//
// Reference semantics: OpenMM NonbondedForce PME as configured by testsystems.py:3504-3517
// (ewaldErrorTolerance 1e-5, cutoff 1 nm).  f64 restatement: oracle/md_oracle.py (pme_reciprocal),
// itself pinned against direct Ewald summation.
#include "remd_internal.h"
#include "listed_terms.h"
#include "pme_pow2.h"
#include <cmath>
#include <vector>
#include <type_traits>

#ifndef PME_PRIO
#define PME_PRIO 3          // wave priority of the mesh kernels (A/B: tools/build_variant.sh -DPME_PRIO=0)
#endif
#define PME_ORDER 5
#define FFT_B 8
#define FFT_T 32
#define PME_FIXED_SCALE 68719476736.0      // 2^36

struct fft_sched { const uint2* tab; int off[8]; };   // see fft_stage_sched

struct pme_state {
    bool prio_hi = true;                  // mesh kernels at raised wave priority (remd_pme_forces copies remd_ctx::mesh_prio_hi)
    int n[4] = {0, 0, 0, 0};           // mesh dimensions; n[3] = nz / 2 (length of the packed real-to-complex z transform)
    int R = 0;
    size_t npts = 0;
    bool ready = false;                // buffers of the force path are allocated (not the FFT test hook's)
    float2* d_grid = nullptr;          // [R][nz/2+1][nx][ny] half spectrum, kz-major (or the full complex grid of the test hook)
    hipStream_t stream = nullptr;      // stream of the current remd_pme_forces call
    int nzc = 0; size_t nspec = 0; size_t xy_lds = 0; bool xy_fused = false; int xy_threads = 512;
    int xy_pow2 = 0;                   // 64 / 128: the plane pass runs on the register transforms of pme_pow2.h
    int* d_col_count = nullptr; int* d_col_start = nullptr; int* d_cursor = nullptr; int* d_atom_col = nullptr; int* d_col_atoms = nullptr;
    float2* d_tw[4] = {nullptr, nullptr, nullptr, nullptr};   // twiddle tables exp(-2 pi i k / n)
    float* d_bmod[3] = {nullptr, nullptr, nullptr};  // |b(m)|^-2 ... stored as B-spline moduli squared inverse
    int nrad[4] = {0, 0, 0, 0}; int radix[4][8];
    int xs_sw = 0;                      // y-slab width of pme_x_fused_kernel (planes that do not fit the LDS)
    int ys_sh = 0;                      // x-slab height of pme_y_slab_kernel (the y passes of those planes)
    float* d_infl = nullptr; int infl_version = -1;   // influence function [R][nz/2+1][nx][ny], rebuilt when a box changes
    int infl_rep = 0;                                 // planes between two replicas' tables (0: all boxes equal, one table)
    bool z_half = false;               // nz even: z transforms run as nz/2-point complex FFTs of packed real pairs
    double* d_energy = nullptr;        // [R][n_eblk]
    int n_eblk = 0;
    fft_sched sch_x, sch_y, sch_z;     // butterfly schedules of the in-place passes (xy planes; z lines for (sch_nl, sch_zt))
    uint2* d_sched[3] = {nullptr, nullptr, nullptr};
    int sch_nl = 0, sch_zt = 0;
    // bins filled by the integrator chain's epilogue (no binning launch on the critical path): count[2][R][nx] double buffered by
    // evaluation parity (the spreading pass zeroes the other one), atoms[R][nx][cbin_cap]; cbin_use: this evaluation reads them
    float* d_cbin_q = nullptr;
    // mesh forces by position in the bins (round 6): the gather's atomics of neighbouring lanes fall on neighbouring addresses, and
    // pme_unbin_forces_kernel hands every atom its total with ONE scattered triple (the gather used to issue five per atom)
    unsigned long long* d_fbin = nullptr; size_t fbin_P = 0;
    int* d_cbin_count = nullptr; float4* d_cbin_atoms = nullptr; int cbin_cap = 0, cbin_parity = 0; bool cbin_use = false;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// mnb[s] / mNs[s]: ceil(2^32 / d) for d = n/radix[s] and d = Ns(s): exact unsigned division of indices < 2^16
struct fft_plan { int n; int nrad; int radix[8]; unsigned mnb[8]; unsigned mNs[8]; int prio; /* raise the wave priority: the mesh chain is the critical path */ };
__host__ __device__ __forceinline__ unsigned fft_magic(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ int fft_div(int x, unsigned magic, int d) { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); }

// butterflies; SIGN = -1 forward, +1 inverse
template <int SIGN>
__device__ __forceinline__ void bfly2(float2* v) { const float2 a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
template <int SIGN>
__device__ __forceinline__ void bfly3(float2* v)
{
    const float c = -0.5f, s = SIGN * 0.86602540378443865f;
    const float2 t1 = cadd(v[1], v[2]);
    const float2 t2 = make_float2(v[0].x + c * t1.x, v[0].y + c * t1.y);
    const float2 d = csub(v[1], v[2]);
    const float2 t3 = make_float2(-s * d.y, s * d.x);     // i*s*d
    v[0] = cadd(v[0], t1); v[1] = cadd(t2, t3); v[2] = csub(t2, t3);
}
template <int SIGN>
__device__ __forceinline__ void bfly4(float2* v)
{
    const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    const float2 id = make_float2(-SIGN * d.y, SIGN * d.x);   // SIGN * i * d
    v[0] = cadd(a, c); v[2] = csub(a, c); v[1] = cadd(b, id); v[3] = csub(b, id);
}
// radix 8 = two radix-4 butterflies on the even / odd inputs + the eighth roots of unity (round 4: power-of-two meshes of the
// rebalanced Ewald split, 64 = 8 x 8 is two LDS stages per line instead of three)
template <int SIGN>
__device__ __forceinline__ void bfly8(float2* v)
{
    float2 e[4] = { v[0], v[2], v[4], v[6] }, o[4] = { v[1], v[3], v[5], v[7] };
    bfly4<SIGN>(e); bfly4<SIGN>(o);
    const float h = 0.70710678118654752f;
    const float2 o1 = make_float2((o[1].x - SIGN * o[1].y) * h, (o[1].y + SIGN * o[1].x) * h);      // * (1 + SIGN i) / sqrt 2
    const float2 o2 = make_float2(-SIGN * o[2].y, SIGN * o[2].x);                                   // * SIGN i
    const float2 o3 = make_float2((-o[3].x - SIGN * o[3].y) * h, (-o[3].y + SIGN * o[3].x) * h);    // * (-1 + SIGN i) / sqrt 2
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
    v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
}
template <int SIGN>
__device__ __forceinline__ void bfly5(float2* v)
{
    const float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f;
    const float s1 = SIGN * 0.95105651629515353f, s2 = SIGN * 0.58778525229247314f;
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 x0 = v[0];
    v[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    const float2 p1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y);
    const float2 p2 = make_float2(x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y);
    const float2 q1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 q2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    // v[k] = p + i*q  (i*q = (-q.y, q.x))
    v[1] = make_float2(p1.x - q1.y, p1.y + q1.x); v[4] = make_float2(p1.x + q1.y, p1.y - q1.x);
    v[2] = make_float2(p2.x - q2.y, p2.y + q2.x); v[3] = make_float2(p2.x + q2.y, p2.y - q2.x);
}

// ---- generic strided-line Stockham FFT in LDS ---------------------------------------------------------
// nlines lines of length n; element e of line l sits at buf[l*ls + e*es].  Consecutive threads take the
// unit-stride index (lines when ls == 1, butterflies when es == 1) so LDS accesses stay conflict-light.
template <int SIGN>
__device__ float2* fft_lines_lds(const fft_plan& pl, float2* src, float2* dst, int nlines, int ls, int es,
                                 const float2* __restrict__ tw, int tid, int nthreads, bool lines_fastest = false)
{
    const int n = pl.n;
    const unsigned mlines = fft_magic((unsigned)nlines);
    int Ns = 1;
    for (int s = 0; s < pl.nrad; ++s) {
        const int Rx = pl.radix[s];
        const int nb = n / Rx;
        const int total = nlines * nb;
        const int tstride = n / (Ns * Rx);
        const unsigned mnb = pl.mnb[s], mNs = pl.mNs[s];
        for (int idx = tid; idx < total; idx += nthreads) {
            int l, j;
            if (ls == 1 || lines_fastest) { j = fft_div(idx, mlines, nlines); l = idx - j * nlines; }
            else { l = fft_div(idx, mnb, nb); j = idx - l * nb; }
            const int jq = fft_div(j, mNs, Ns);
            const int k = j - jq * Ns;
            const int tstep = k * tstride;                  // tstep * r < n for every r < Rx: no modulo needed
            const float2* S = src + l * ls;
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) if (r < Rx) {
                v[r] = S[(j + r * nb) * es];
                if (s > 0 && r > 0) {                       // first stage (Ns = 1) and r = 0: twiddle = 1
                    float2 w = tw[tstep * r];
                    if (SIGN > 0) w.y = -w.y;
                    v[r] = cmul(v[r], w);
                }
            }
            if (Rx == 2) bfly2<SIGN>(v); else if (Rx == 3) bfly3<SIGN>(v); else if (Rx == 4) bfly4<SIGN>(v); else if (Rx == 8) bfly8<SIGN>(v); else bfly5<SIGN>(v);
            const int d0 = jq * Ns * Rx + k;
            float2* D = dst + l * ls;
#pragma unroll
            for (int r = 0; r < 8; ++r) if (r < Rx) D[(d0 + r * Ns) * es] = v[r];
        }
        __syncthreads();
        float2* tmp = src; src = dst; dst = tmp;
        Ns *= Rx;
    }
    return src;
}

// ---- in-place variant: every thread keeps its share of the points in registers across the barrier, so only ONE
// LDS image of the data is needed (half the footprint of the ping-pong version => more workgroups per CU).
template <int SIGN, int RX> __device__ __forceinline__ void bfly(float2* v)
{
    if (RX == 2) bfly2<SIGN>(v); else if (RX == 3) bfly3<SIGN>(v); else if (RX == 4) bfly4<SIGN>(v); else if (RX == 8) bfly8<SIGN>(v); else bfly5<SIGN>(v);
}

// Butterfly schedule of one in-place pass, precomputed on the host (build_sched): the index arithmetic of a
// mixed-radix stage (two divisions by non-constant sizes, five multiplies) is the same for every workgroup and
// every call, and on the VALU it cost more than the floating-point butterflies themselves.  Entry of butterfly
// (stage s, slot b, thread t) at tab[off[s] + b * nthreads + t]: x = first input element | first output element << 16
// (0xffffffff: idle), y = twiddle step.

template <int SIGN, int RX, int PPT>
__device__ __forceinline__ void fft_stage_sched(float2* buf, const uint2* __restrict__ tab, int in_stride, int out_stride,
                                                bool twiddle, const float2* __restrict__ tw, int tid, int nthreads)
{
    constexpr int NB = (PPT + RX - 1) / RX;
    float2 v[NB][RX];
    int dst[NB];
    uint2 e[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) e[b] = tab[b * nthreads + tid];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        dst[b] = -1;
        if (e[b].x != 0xffffffffu) {
            const float2* S = buf + (e[b].x & 0xffffu);
            const int tstep = (int)e[b].y;
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                v[b][r] = S[r * in_stride];
                if (twiddle && r > 0) {
                    float2 w = tw[tstep * r];
                    if (SIGN > 0) w.y = -w.y;
                    v[b][r] = cmul(v[b][r], w);
                }
            }
            bfly<SIGN, RX>(v[b]);
            dst[b] = (int)(e[b].x >> 16);
        }
    }
    __syncthreads();                                    // every input of this stage is in registers
#pragma unroll
    for (int b = 0; b < NB; ++b) if (dst[b] >= 0) {
#pragma unroll
        for (int r = 0; r < RX; ++r) buf[dst[b] + r * out_stride] = v[b][r];
    }
    __syncthreads();
}

// in-place FFT of nlines lines (element e of line l at buf[l*ls + e*es]); every thread keeps its share of the points in
// registers across the barrier, so only ONE LDS image of the data is needed
template <int SIGN, int PPT>
__device__ __forceinline__ void fft_lines_inplace(const fft_plan& pl, const fft_sched& sc, float2* buf, int es,
                                                  const float2* __restrict__ tw, int tid, int nthreads)
{
    int Ns = 1;
    for (int s = 0; s < pl.nrad; ++s) {
        const int Rx = pl.radix[s];
        const int in_stride = (pl.n / Rx) * es, out_stride = Ns * es;
        const uint2* tab = sc.tab + sc.off[s];
        if (Rx == 8) fft_stage_sched<SIGN, 8, PPT>(buf, tab, in_stride, out_stride, s > 0, tw, tid, nthreads);
        else if (Rx == 4) fft_stage_sched<SIGN, 4, PPT>(buf, tab, in_stride, out_stride, s > 0, tw, tid, nthreads);
        else if (Rx == 5) fft_stage_sched<SIGN, 5, PPT>(buf, tab, in_stride, out_stride, s > 0, tw, tid, nthreads);
        else if (Rx == 3) fft_stage_sched<SIGN, 3, PPT>(buf, tab, in_stride, out_stride, s > 0, tw, tid, nthreads);
        else fft_stage_sched<SIGN, 2, PPT>(buf, tab, in_stride, out_stride, s > 0, tw, tid, nthreads);
        Ns *= Rx;
    }
}

#define PME_MESH_SCALE 16777216.0f     // 2^24: 32-bit fixed-point charge mesh, |sum| < 128

// order-5 cardinal B-spline weights and derivatives: w[j] = M5(f + j), d[j] = M5'(f + j), j = 0..4,
// belonging to mesh index k0 - j.
__device__ __forceinline__ void bspline5(float f, float* w, float* d)
{
    float a[5] = { f, 1.f - f, 0.f, 0.f, 0.f };          // M2
#pragma unroll
    for (int m = 3; m <= 4; ++m) {
        const float div = 1.f / (m - 1);
#pragma unroll
        for (int j = 4; j >= 0; --j) if (j < m) {
            const float prev = (j > 0) ? a[j - 1] : 0.f;
            const float cur = (j < m - 1) ? a[j] : 0.f;
            a[j] = div * ((f + j) * cur + (m - f - j) * prev);
        }
    }
    // a = M4; derivative of M5 and M5 itself
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float prev = (j > 0) ? a[j - 1] : 0.f;
        const float cur = (j < 4) ? a[j] : 0.f;
        d[j] = cur - prev;
        w[j] = 0.25f * ((f + j) * cur + (5.f - f - j) * prev);
    }
}

// ---- atom binning by mesh column ------------------------------------------------------------------------
// column of atom i = (kx, ky) = integer part of its scaled fractional x, y coordinate.  The fused spread + z-FFT
// kernel walks, for its 8 lines, the atoms of the 5 x 12 columns whose order-5 stencils can reach them.
__device__ __forceinline__ void pme_scaled(const float4 x, const float* __restrict__ box4, int nx, int ny, int nz,
                                           float& ux, float& uy, float& uz, int& kx, int& ky, int& kz)
{
    remd_pme_scaled1(x.x, box4[0], nx, ux, kx);
    remd_pme_scaled1(x.y, box4[1], ny, uy, ky);
    remd_pme_scaled1(x.z, box4[2], nz, uz, kz);
}

// The atoms a mesh row x can receive charge from / give force to: those of the five bins kx = x .. x+4 (mod nx).  Two
// layouts of the bins: compact (pme_bin_kernel: start[nx+1] + one atom array, the five bins are at most two contiguous
// runs) or capped (binned by the integrator chain's epilogue: count[nx] + atoms[nx][cap], five runs).
struct pme_cand { int n[5]; int base[5]; int ntot; };
__device__ __forceinline__ pme_cand pme_candidates(const int* __restrict__ cs, int x, int nx, int cap)
{
    pme_cand c;
    if (cap > 0) {
        c.ntot = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            int kx = x + b; if (kx >= nx) kx -= nx;
            c.n[b] = min(cs[kx], cap); c.base[b] = kx * cap; c.ntot += c.n[b];
        }
    } else {
        const int xe = x + 5;
        const int beg0 = cs[x], end0 = cs[xe <= nx ? xe : nx];
        const int beg1 = cs[0], end1 = (xe <= nx) ? beg1 : cs[xe - nx];
        c.n[0] = end0 - beg0; c.base[0] = beg0; c.n[1] = end1 - beg1; c.base[1] = beg1;
        c.n[2] = c.n[3] = c.n[4] = 0; c.base[2] = c.base[3] = c.base[4] = 0;
        c.ntot = c.n[0] + c.n[1];
    }
    return c;
}
// atom index, position and effective charge of candidate t: capped bins hold float4(x, y, z, index) + a charge array (both
// loads independent), compact ones indices into pos[] / param[]
__device__ __forceinline__ int pme_cand_atom(const pme_cand& c, const int* __restrict__ ca, int t, bool capped,
                                             const float4* __restrict__ P, float4& xi, const float* __restrict__ cq,
                                             const float4* __restrict__ param, const float* __restrict__ rep_lam, int r, float& q, int* flat = nullptr)
{
    int base = c.base[0];
#pragma unroll
    for (int b = 0; b < 4; ++b) if (t >= c.n[b]) { t -= c.n[b]; base = c.base[b + 1]; } else break;
    if (flat) *flat = base + t;              // position in the bins' storage: consecutive candidates of a bin, consecutive positions
    int i;
    if (capped) {
        xi = reinterpret_cast<const float4*>(ca)[base + t];
        i = __float_as_int(xi.w);
        if (cq) { q = cq[base + t]; return i; }
    } else {
        i = ca[base + t];
        xi = P[i];
    }
    const float4 pr = param[i];
    q = pr.x;
    // the atom's charge scales with lambda_electrostatics: slot 2 of the replica's float4 (one alchemical region, forces.hip), or -- general
    // regions under the exact PME treatment, alch_regions.hip -- the slot the atom's code 8 + region names
    if (rep_lam && pr.w != 0.f) q *= rep_lam[4 * r + (pr.w >= 8.f ? (int)pr.w - 8 : 2)];
    return i;
}

// bin atoms by their mesh column kx (one workgroup per replica, everything in LDS): the fused spread + z-FFT
// workgroup of row x walks the atoms of bins kx = x .. x+4.  The order inside a bin is irrelevant because the
// charges are accumulated as integers.
__global__ __launch_bounds__(1024)
void pme_bin_kernel(int N, int Npad, int nx, int ny, int nz, const float4* __restrict__ pos, const float* __restrict__ box,
                    int* __restrict__ bin_start /*[R][nx+1]*/, int* __restrict__ bin_atoms /*[R][Npad]*/,
                    unsigned int* fork_flag, unsigned int fork_seq)
{
    // this launch sits directly behind the integrator on the main stream: its start publishes "positions are final" to the
    // direct-space kernels polling on the second stream (remd_ctx::d_sync)
    if (fork_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(fork_flag, fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_setprio(PME_PRIO);    // latency-bound pipeline sharing the CUs with the VALU-bound direct-space kernels: win issue arbitration
    __shared__ int s_cnt[257], s_start[257], s_cur[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k <= nx; k += 1024) s_cnt[k] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        float ux, uy, uz; int kx, ky, kz;
        pme_scaled(pos[(size_t)r * Npad + i], box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
        if (kx >= nx) kx -= nx;
        atomicAdd(&s_cnt[kx], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < nx; ++k) { s_start[k] = run; s_cur[k] = run; run += s_cnt[k]; }
        s_start[nx] = run;
    }
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        float ux, uy, uz; int kx, ky, kz;
        pme_scaled(pos[(size_t)r * Npad + i], box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
        if (kx >= nx) kx -= nx;
        const int slot = atomicAdd(&s_cur[kx], 1);
        bin_atoms[(size_t)r * Npad + slot] = i;
    }
    for (int k = tid; k <= nx; k += 1024) bin_start[(size_t)r * (nx + 1) + k] = s_start[k];
}

#define Z_PPT 12             // points per thread in registers: nl * nz <= Z_PPT * ZT (ZT = workgroup size, 256 or 512)

// fused spreading + forward z FFT.  Workgroup = nl lines (x, y0..y0+nl-1), nl a divisor of ny (the whole row when it
// fits).  Charges are accumulated in LDS as 32-bit fixed point (order-independent => bit-reproducible), converted to
// complex f32 and transformed in place; the half spectrum is written kz-major in runs of nl points.
// HALF (nz even): the real lines are packed as nz/2 complex numbers z[n] = x[2n] + i x[2n+1], transformed with an
// nz/2-point FFT (pl, sc describe THAT transform) and untangled into the half spectrum while it is written out:
//   X[k] = (Z[k] + conj Z[M-k]) / 2 - (i/2) W^k (Z[k] - conj Z[M-k]),  W = exp(-2 pi i / nz),  M = nz/2,  Z[M] = Z[0].
// Half the butterflies and half the LDS of the plain complex transform (four workgroups per CU instead of three).
template <int Z_THREADS, bool HALF>
__global__ __launch_bounds__(Z_THREADS)
void pme_spread_zfwd_kernel(fft_plan pl, fft_sched sc, int nl, int nx, int ny, int Npad, const float4* __restrict__ pos,
                            const float4* __restrict__ param, const float* __restrict__ box, const float* __restrict__ rep_lam,
                            const int* __restrict__ col_start, const int* __restrict__ col_atoms,
                            float2* __restrict__ spec, const float2* tw, const float2* tw_half,
                            int bin_cap, int* __restrict__ zero_count, unsigned int* fork_flag, unsigned int fork_seq,
                            const float* __restrict__ bin_q, listed_tables LT, int n_mesh_blocks, long long* __restrict__ force)
{
    if ((int)blockIdx.x >= n_mesh_blocks) {
        // workgroups behind the mesh rows: the listed terms of this force evaluation (listed_terms.h), one term per thread
        listed_forces_body(LT, Npad, pos, box, force, ((int)blockIdx.x - n_mesh_blocks) * Z_THREADS + (int)threadIdx.x, blockIdx.y);
        return;
    }
    if (pl.prio) __builtin_amdgcn_s_setprio(PME_PRIO);    // latency-bound pipeline sharing the CUs with the VALU-bound direct-space kernels: win issue arbitration
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = pl.n;                                     // length of the FFT that is run
    const int nz = HALF ? 2 * M : M, nzc = nz / 2 + 1, PZ = HALF ? ((M + 1) | 1) : (nz | 1);
    float2* buf = reinterpret_cast<float2*>(smem);          // [nl][PZ]
    float2* s_tw = buf + nl * PZ;                           // [nz] twiddles of the full length
    float2* s_twh = s_tw + nz;                              // [M] twiddles of the half length (HALF only)
    int* acc = reinterpret_cast<int*>(buf);                 // [nl][nz] aliases buf: converted through registers below
    const int r = blockIdx.y, tid = threadIdx.x;
    // a mesh row x is covered by ceil(ny / nl) workgroups of nl lines (the last one may hang over the end of the row)
    const int nbpr = (ny + nl - 1) / nl;
    const int x = blockIdx.x / nbpr, y0 = (blockIdx.x - x * nbpr) * nl;
    for (int idx = tid; idx < nl * nz; idx += Z_THREADS) acc[idx] = 0;
    for (int idx = tid; idx < nz; idx += Z_THREADS) s_tw[idx] = tw[idx];
    if (HALF) for (int idx = tid; idx < M; idx += Z_THREADS) s_twh[idx] = tw_half[idx];
    __syncthreads();
    const int* cs = col_start + (size_t)r * (bin_cap > 0 ? nx : nx + 1);
    const int* ca = col_atoms + (size_t)r * (bin_cap > 0 ? (size_t)nx * bin_cap * 4 : (size_t)Npad);     // capped entries are float4
    const float4* P = pos + (size_t)r * Npad;
    if (zero_count && blockIdx.x == 0) for (int k = tid; k < nx; k += Z_THREADS) zero_count[(size_t)r * nx + k] = 0;   // the bins of the NEXT evaluation
    if (fork_flag && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)             // (no binning launch in front: this one publishes the fork)
        __hip_atomic_store(fork_flag, fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // candidate atoms: mesh column kx in {x .. x+4} (stencil index a = kx - x); the y stencil is tested per atom.  All
    // candidates are taken in ONE pass: one dependent chain of global loads (bin -> atom -> position) per workgroup.
    {
        const pme_cand cnd = pme_candidates(cs, x, nx, bin_cap);
        const int ntot = cnd.ntot;
        for (int t = tid; t < ntot; t += Z_THREADS) {
            float4 xi; float q;
            pme_cand_atom(cnd, ca, t, bin_cap > 0, P, xi, bin_q ? bin_q + (size_t)r * nx * bin_cap : (const float*)nullptr, param, rep_lam, r, q);
            if (q == 0.f) continue;
            float ux, uy, uz; int kx, ky, kz;
            pme_scaled(xi, box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
            float wx[5], wy[5], wz[5], dx[5], dy[5], dz[5];
            bspline5(ux - kx, wx, dx); bspline5(uy - ky, wy, dy); bspline5(uz - kz, wz, dz);
            if (kx >= nx) kx -= nx; if (ky >= ny) ky -= ny; if (kz >= nz) kz -= nz;
            int a = kx - x; if (a < 0) a += nx;              // 0..4 by construction of the bins
            float wa = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (k == a) wa = wx[k];
            const float qa = q * wa * PME_MESH_SCALE;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                int iy = ky - b; if (iy < 0) iy += ny;
                const int lb = iy - y0;                      // line inside this workgroup?
                if (lb < 0 || lb >= nl) continue;
                const float qab = qa * wy[b];
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    int iz = kz - c; if (iz < 0) iz += nz;
                    atomicAdd(&acc[lb * nz + iz], __float2int_rn(qab * wz[c]));
                }
            }
        }
    }
    __syncthreads();
    const unsigned mM = fft_magic((unsigned)M);
    {
        // acc and buf share LDS: every thread first pulls its share of the integer mesh into registers, then all
        // threads write the complex image (nl * M <= Z_PPT * Z_THREADS)
        float2 val[Z_PPT];
#pragma unroll
        for (int q = 0; q < Z_PPT; ++q) {
            const int idx = tid + q * Z_THREADS;             // complex point: line l = idx / M, element n = idx % M
            val[q] = make_float2(0.f, 0.f);
            if (idx < nl * M) {
                if (HALF) { const int2 w = reinterpret_cast<const int2*>(acc)[idx]; val[q] = make_float2((float)w.x * (1.0f / PME_MESH_SCALE), (float)w.y * (1.0f / PME_MESH_SCALE)); }
                else val[q].x = (float)acc[idx] * (1.0f / PME_MESH_SCALE);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Z_PPT; ++q) {
            const int idx = tid + q * Z_THREADS;
            if (idx < nl * M) { const int l = fft_div(idx, mM, M); buf[idx + l * (PZ - M)] = val[q]; }   // l*PZ + n
        }
    }
    __syncthreads();
    fft_lines_inplace<-1, Z_PPT>(pl, sc, buf, 1, HALF ? s_twh : s_tw, tid, Z_THREADS);
    float2* S = spec + (size_t)r * nzc * nx * ny;
    const unsigned mnl = fft_magic((unsigned)nl);
    for (int idx = tid; idx < nl * nzc; idx += Z_THREADS) {
        const int kz = fft_div(idx, mnl, nl), b = idx - kz * nl;
        float2 X;
        if (HALF) {
            const float2 Zk = buf[b * PZ + (kz == M ? 0 : kz)], Zm = buf[b * PZ + (kz == 0 ? 0 : M - kz)];
            const float2 E = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));          // (Z[k] + conj Z[M-k]) / 2
            const float2 O = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));         // -(i/2) (Z[k] - conj Z[M-k])
            X = cadd(E, cmul(s_tw[kz], O));
        } else {
            X = buf[b * PZ + kz];
        }
        if (y0 + b < ny) S[((size_t)kz * nx + x) * ny + y0 + b] = X;
    }
}

// inverse z + force gather in one launch.  After the inverse z transforms a workgroup holds the real potential of its mesh
// row(s) x in LDS; the atoms whose 5-point x stencil touches that row are exactly the ones of bins kx = x .. x+4 (the
// candidates of the spreading pass), so each of them takes ITS SHARE of the force from this row here -- 25 LDS reads --
// and adds it to the fixed-point accumulators.  An atom gets its force from five workgroups (15 integer atomics instead
// of 3; sums are order independent), and the real potential mesh (4 B / point written, then 125 reads per atom from L2) and
// one dependent launch on the critical path of a step disappear.
template <int Z_THREADS, bool HALF>
__global__ __launch_bounds__(Z_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void pme_zinv_gather_kernel(fft_plan pl, fft_sched sc, int nl, int nx, int ny, const float2* __restrict__ spec,
                            const float2* tw, const float2* tw_half, int Npad, const float4* __restrict__ pos,
                            const float4* __restrict__ param, const float* __restrict__ box, const float* __restrict__ rep_lam,
                            const int* __restrict__ col_start, const int* __restrict__ col_atoms, long long* __restrict__ force,
                            int bin_cap, const float* __restrict__ bin_q, unsigned long long* __restrict__ fbin, size_t fbin_P)
{
    if (pl.prio) __builtin_amdgcn_s_setprio(PME_PRIO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = pl.n;
    const int nz = HALF ? 2 * M : M, nzc = nz / 2 + 1, PZ = HALF ? ((M + 1) | 1) : (nz | 1);
    float2* buf = reinterpret_cast<float2*>(smem);
    float2* s_tw = buf + nl * PZ;
    float2* s_twh = s_tw + nz;
    const int r = blockIdx.y, tid = threadIdx.x;
    // a mesh row x is covered by ceil(ny / nl) workgroups of nl lines (the last one may hang over the end of the row)
    const int nbpr = (ny + nl - 1) / nl;
    const int x = blockIdx.x / nbpr, y0 = (blockIdx.x - x * nbpr) * nl;
    for (int idx = tid; idx < nz; idx += Z_THREADS) s_tw[idx] = tw[idx];
    if (HALF) for (int idx = tid; idx < M; idx += Z_THREADS) s_twh[idx] = tw_half[idx];
    const float2* S = spec + (size_t)r * nzc * nx * ny;
    const unsigned mnl = fft_magic((unsigned)nl);
    for (int idx = tid; idx < nl * nzc; idx += Z_THREADS) {
        const int kz = fft_div(idx, mnl, nl), b = idx - kz * nl;
        const float2 v = (y0 + b < ny) ? S[((size_t)kz * nx + x) * ny + y0 + b] : make_float2(0.f, 0.f);
        buf[b * PZ + kz] = v;
        if (!HALF && kz > 0 && kz < nz - kz) buf[b * PZ + nz - kz] = make_float2(v.x, -v.y);
    }
    __syncthreads();
    if (HALF) {
        for (int idx = tid; idx < nl * (M / 2 + 1); idx += Z_THREADS) {
            const int k = fft_div(idx, mnl, nl), b = idx - k * nl;
            const float2 Xk = buf[b * PZ + k], Xm = buf[b * PZ + M - k];
            const float2 A = make_float2(Xk.x + Xm.x, Xk.y - Xm.y);
            const float2 B = make_float2(Xk.x - Xm.x, Xk.y + Xm.y);
            const float2 w = s_tw[k];
            const float2 t = cmul(make_float2(w.x, -w.y), B);
            buf[b * PZ + k] = make_float2(A.x - t.y, A.y + t.x);
            if (k != 0 && k != M - k) {
                const float2 u = cmul(w, make_float2(B.x, -B.y));
                buf[b * PZ + M - k] = make_float2(A.x - u.y, -A.y + u.x);
            }
        }
        __syncthreads();
    }
    fft_lines_inplace<+1, Z_PPT>(pl, sc, buf, 1, HALF ? s_twh : s_tw, tid, Z_THREADS);
    // potential of line l at z: HALF keeps the real line as pairs (x[2n], x[2n+1]) = consecutive floats; else the real parts
    const float* phi = reinterpret_cast<const float*>(buf);
    const int zs = HALF ? 1 : 2, ls = 2 * PZ;
    const int* cs = col_start + (size_t)r * (bin_cap > 0 ? nx : nx + 1);
    const int* ca = col_atoms + (size_t)r * (bin_cap > 0 ? (size_t)nx * bin_cap * 4 : (size_t)Npad);
    const float4* P = pos + (size_t)r * Npad;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    unsigned long long* F = reinterpret_cast<unsigned long long*>(force + (size_t)r * 3 * Npad);
    const pme_cand cnd = pme_candidates(cs, x, nx, bin_cap);
    const int ntot = cnd.ntot;
    unsigned long long* FB = fbin ? fbin + (size_t)r * 3 * fbin_P : (unsigned long long*)nullptr;
    for (int t = tid; t < ntot; t += Z_THREADS) {
        float4 xi; float q; int flat;
        const int i = pme_cand_atom(cnd, ca, t, bin_cap > 0, P, xi, bin_q ? bin_q + (size_t)r * nx * bin_cap : (const float*)nullptr, param, rep_lam, r, q, &flat);
        if (q == 0.f) continue;
        float ux, uy, uz; int kx, ky, kz;
        pme_scaled(xi, box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
        float wx[5], wy[5], wz[5], dx[5], dy[5], dz[5];
        bspline5(ux - kx, wx, dx); bspline5(uy - ky, wy, dy); bspline5(uz - kz, wz, dz);
        if (kx >= nx) kx -= nx; if (ky >= ny) ky -= ny; if (kz >= nz) kz -= nz;
        int a = kx - x; if (a < 0) a += nx;              // 0..4 by construction of the bins
        float wxa = 0.f, dxa = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (k == a) { wxa = wx[k]; dxa = dx[k]; }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        bool any = false;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            int iy = ky - b; if (iy < 0) iy += ny;
            const int lb = iy - y0;
            if (lb < 0 || lb >= nl) continue;
            any = true;
            float sx = 0.f, sz = 0.f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                int iz = kz - c; if (iz < 0) iz += nz;
                const float p = phi[lb * ls + iz * zs];
                sx += wz[c] * p; sz += dz[c] * p;
            }
            gx += wy[b] * sx; gy += dy[b] * sx; gz += wy[b] * sz;
        }
        if (!any) continue;
        // dE/dx = q * dtheta/du * du/dx, du/dx = n / L
        const float Fx = -q * dxa * gx * nx / Lx, Fy = -q * wxa * gy * ny / Ly, Fz = -q * wxa * gz * nz / Lz;
        if (FB) {          // by position in the bins: neighbouring lanes, neighbouring addresses (pme_unbin_forces_kernel takes them to the atoms)
            atomicAdd(&FB[flat], remd_f2fix(Fx));
            atomicAdd(&FB[fbin_P + flat], remd_f2fix(Fy));
            atomicAdd(&FB[2 * fbin_P + flat], remd_f2fix(Fz));
            continue;
        }
        atomicAdd(&F[i], remd_f2fix(Fx));
        atomicAdd(&F[Npad + i], remd_f2fix(Fy));
        atomicAdd(&F[2 * Npad + i], remd_f2fix(Fz));
    }
}

// ---- the z passes of power-of-two meshes (nz = 64, 128; nz / 2 = M = 8 * R2 packed points per line) on the register
// transforms of pme_pow2.h.  A workgroup takes 64 lines (x, y0 .. y0 + 63) on 64 * R2 threads: thread (l = tid mod 64,
// j = tid / 64) holds elements j + R2 q, q < 8, of line l -- j is uniform over a wavefront, so every twiddle is a scalar
// load, and every LDS image is [element][line]: consecutive lanes, consecutive addresses, in the accumulators of the
// spreading, the exchange between the two butterflies, the untangling of the packed transform and the potential the
// gather reads; the half spectrum leaves and enters in runs of 64 numbers.  Five workgroup barriers behind the spreading
// (the scheduled pass: two per stage + four).
template <int M, int R2>
__global__ __launch_bounds__(64 * R2)
void pme_spread_zfwd_pow2_kernel(int nx, int ny, int Npad, const float4* __restrict__ pos,
                                 const float4* __restrict__ param, const float* __restrict__ box, const float* __restrict__ rep_lam,
                                 const int* __restrict__ col_start, const int* __restrict__ col_atoms,
                                 float2* __restrict__ spec, const float2* __restrict__ tw, const float2* __restrict__ tw_half,
                                 int bin_cap, int* __restrict__ zero_count, unsigned int* fork_flag, unsigned int fork_seq,
                                 const float* __restrict__ bin_q, listed_tables LT, int n_mesh_blocks, long long* __restrict__ force, int prio)
{
    constexpr int R1 = 8, NL = 64, ZT = NL * R2, MM = R1 / R2, nz = 2 * M;
    static_assert(M == R1 * R2 && (R2 == 4 || R2 == 8), "M = 8 * R2");
    if ((int)blockIdx.x >= n_mesh_blocks) {
        listed_forces_body(LT, Npad, pos, box, force, ((int)blockIdx.x - n_mesh_blocks) * ZT + (int)threadIdx.x, blockIdx.y);
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(PME_PRIO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* B = reinterpret_cast<float2*>(smem);            // [M][NL]
    int* acc = reinterpret_cast<int*>(smem);                // [nz][NL]: the same bytes
    const int r = blockIdx.y, tid = threadIdx.x;
    const int nbpr = ny / NL;
    const int x = blockIdx.x / nbpr, y0 = (blockIdx.x - x * nbpr) * NL;
    for (int idx = tid; idx < nz * NL; idx += ZT) acc[idx] = 0;
    __syncthreads();
    const int* cs = col_start + (size_t)r * (bin_cap > 0 ? nx : nx + 1);
    const int* ca = col_atoms + (size_t)r * (bin_cap > 0 ? (size_t)nx * bin_cap * 4 : (size_t)Npad);
    const float4* P = pos + (size_t)r * Npad;
    if (zero_count && blockIdx.x == 0) for (int k = tid; k < nx; k += ZT) zero_count[(size_t)r * nx + k] = 0;
    if (fork_flag && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        __hip_atomic_store(fork_flag, fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    {
        const pme_cand cnd = pme_candidates(cs, x, nx, bin_cap);
        const int ntot = cnd.ntot;
        for (int t = tid; t < ntot; t += ZT) {
            float4 xi; float q;
            pme_cand_atom(cnd, ca, t, bin_cap > 0, P, xi, bin_q ? bin_q + (size_t)r * nx * bin_cap : (const float*)nullptr, param, rep_lam, r, q);
            if (q == 0.f) continue;
            float ux, uy, uz; int kx, ky, kz;
            pme_scaled(xi, box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
            float wx[5], wy[5], wz[5], dx[5], dy[5], dz[5];
            bspline5(ux - kx, wx, dx); bspline5(uy - ky, wy, dy); bspline5(uz - kz, wz, dz);
            if (kx >= nx) kx -= nx; if (ky >= ny) ky -= ny; if (kz >= nz) kz -= nz;
            int a = kx - x; if (a < 0) a += nx;
            float wa = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (k == a) wa = wx[k];
            const float qa = q * wa * PME_MESH_SCALE;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                int iy = ky - b; if (iy < 0) iy += ny;
                const int lb = iy - y0;
                if (lb < 0 || lb >= NL) continue;
                const float qab = qa * wy[b];
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const int iz = (kz - c) & (nz - 1);
                    atomicAdd(&acc[iz * NL + lb], __float2int_rn(qab * wz[c]));
                }
            }
        }
    }
    __syncthreads();
    const int l = tid & (NL - 1);
    const int j = __builtin_amdgcn_readfirstlane(tid / NL);
    float2 v[R1], w[R1];
#pragma unroll
    for (int q = 0; q < R1; ++q) {
        const int n = j + R2 * q;               // packed point n = (x[2n], x[2n+1])
        v[q] = make_float2((float)acc[(2 * n) * NL + l] * (1.0f / PME_MESH_SCALE), (float)acc[(2 * n + 1) * NL + l] * (1.0f / PME_MESH_SCALE));
    }
    __syncthreads();
    p2_dft8<-1>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[k1] = p2_twid<-1>(v[k1], tw_half[j * k1]);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) B[(k1 * R2 + j) * NL + l] = v[k1];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const int k1 = j + R2 * m;
        float2 u[R2];
#pragma unroll
        for (int jp = 0; jp < R2; ++jp) u[jp] = B[(k1 * R2 + jp) * NL + l];
        p2_dft<-1, R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) w[m * R2 + k2] = u[k2];      // Z[k1 + R1 k2]
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) B[(j + R2 * m + R1 * k2) * NL + l] = w[m * R2 + k2];
    __syncthreads();
    // untangle: X[k] = (Z[k] + conj Z[M-k]) / 2 - (i/2) W^k (Z[k] - conj Z[M-k]), W = exp(-2 pi i / nz), Z[M] = Z[0]
    float2* S = spec + (size_t)r * (M + 1) * nx * ny + (size_t)x * ny + y0;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const int k = j + R2 * m + R1 * k2;
            const float2 Zk = w[m * R2 + k2], Zm = B[((M - k) & (M - 1)) * NL + l];
            const float2 E = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
            const float2 O = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));
            S[(size_t)k * nx * ny + l] = p2_add(E, p2_mul(tw[k], O));
            if (k == 0) S[(size_t)M * nx * ny + l] = make_float2(Zk.x - Zk.y, 0.f);     // W^M = -1
        }
}

template <int M, int R2>
__global__ __launch_bounds__(64 * R2) __attribute__((amdgpu_waves_per_eu(8, 8)))
void pme_zinv_gather_pow2_kernel(int nx, int ny, const float2* __restrict__ spec, const float2* __restrict__ tw,
                                 const float2* __restrict__ tw_half, int Npad, const float4* __restrict__ pos,
                                 const float4* __restrict__ param, const float* __restrict__ box, const float* __restrict__ rep_lam,
                                 const int* __restrict__ col_start, const int* __restrict__ col_atoms, long long* __restrict__ force,
                                 int bin_cap, const float* __restrict__ bin_q, int prio, unsigned long long* __restrict__ fbin, size_t fbin_P)
{
    constexpr int R1 = 8, NL = 64, ZT = NL * R2, MM = R1 / R2, nz = 2 * M;
    if (prio) __builtin_amdgcn_s_setprio(PME_PRIO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* B = reinterpret_cast<float2*>(smem);            // [M + 1][NL]
    float* phi = reinterpret_cast<float*>(smem);            // [nz][NL]: the real potential of the 64 lines
    const int r = blockIdx.y, tid = threadIdx.x;
    const int nbpr = ny / NL;
    const int x = blockIdx.x / nbpr, y0 = (blockIdx.x - x * nbpr) * NL;
    const int l = tid & (NL - 1);
    const int j = __builtin_amdgcn_readfirstlane(tid / NL);
    const float2* S = spec + (size_t)r * (M + 1) * nx * ny + (size_t)x * ny + y0;
    float2 v[R1], w[R1];
#pragma unroll
    for (int q = 0; q < R1; ++q) w[q] = S[(size_t)(j + R2 * q) * nx * ny + l];     // X[k], k = j + R2 q
    float2 XM = make_float2(0.f, 0.f);
    if (j == 0) XM = S[(size_t)M * nx * ny + l];
    // the first candidate atom of this thread for the gather at the end: its chain of dependent loads (bin -> atom -> charge)
    // runs under the transform
    const int* cs = col_start + (size_t)r * (bin_cap > 0 ? nx : nx + 1);
    const int* ca = col_atoms + (size_t)r * (bin_cap > 0 ? (size_t)nx * bin_cap * 4 : (size_t)Npad);
    const float4* P = pos + (size_t)r * Npad;
    const float* bq = bin_q ? bin_q + (size_t)r * nx * bin_cap : (const float*)nullptr;
    const pme_cand cnd = pme_candidates(cs, x, nx, bin_cap);
    const int ntot = cnd.ntot;
    float4 xi0 = make_float4(0.f, 0.f, 0.f, 0.f); float q0 = 0.f; int i0 = 0, flat0 = 0;
    if (tid < ntot) i0 = pme_cand_atom(cnd, ca, tid, bin_cap > 0, P, xi0, bq, param, rep_lam, r, q0, &flat0);
#pragma unroll
    for (int q = 0; q < R1; ++q) B[(j + R2 * q) * NL + l] = w[q];
    if (j == 0) B[M * NL + l] = XM;
    __syncthreads();
    // packed spectrum: Z[k] = A + i conj(W^k) Bv, A = X[k] + conj X[M-k], Bv = X[k] - conj X[M-k]
#pragma unroll
    for (int q = 0; q < R1; ++q) {
        const int k = j + R2 * q;
        const float2 Xk = w[q], Xm = B[(M - k) * NL + l];
        const float2 A = make_float2(Xk.x + Xm.x, Xk.y - Xm.y);
        const float2 Bv = make_float2(Xk.x - Xm.x, Xk.y + Xm.y);
        const float2 t = p2_twid<+1>(Bv, tw[k]);
        v[q] = make_float2(A.x - t.y, A.y + t.x);
    }
    __syncthreads();
    p2_dft8<+1>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[k1] = p2_twid<+1>(v[k1], tw_half[j * k1]);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) B[(k1 * R2 + j) * NL + l] = v[k1];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const int k1 = j + R2 * m;
        float2 u[R2];
#pragma unroll
        for (int jp = 0; jp < R2; ++jp) u[jp] = B[(k1 * R2 + jp) * NL + l];
        p2_dft<+1, R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) w[m * R2 + k2] = u[k2];      // packed point n = k1 + R1 k2: (phi[2n], phi[2n+1])
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const int n = j + R2 * m + R1 * k2;
            phi[(2 * n) * NL + l] = w[m * R2 + k2].x;
            phi[(2 * n + 1) * NL + l] = w[m * R2 + k2].y;
        }
    __syncthreads();
#ifdef EXP_ZI_NOGATHER    // knock-out probe: transform only
    if (phi[tid] != 1.2345e33f) return;
#endif
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    unsigned long long* F = reinterpret_cast<unsigned long long*>(force + (size_t)r * 3 * Npad);
    unsigned long long* FB = fbin ? fbin + (size_t)r * 3 * fbin_P : (unsigned long long*)nullptr;
    for (int t = tid; t < ntot; t += ZT) {
        float4 xi = xi0; float q = q0; int i = i0, flat = flat0;
        if (t != tid) i = pme_cand_atom(cnd, ca, t, bin_cap > 0, P, xi, bq, param, rep_lam, r, q, &flat);
        if (q == 0.f) continue;
        float ux, uy, uz; int kx, ky, kz;
        pme_scaled(xi, box + 4 * r, nx, ny, nz, ux, uy, uz, kx, ky, kz);
        float wx[5], wy[5], wz[5], dx[5], dy[5], dz[5];
        bspline5(ux - kx, wx, dx); bspline5(uy - ky, wy, dy); bspline5(uz - kz, wz, dz);
        if (kx >= nx) kx -= nx; if (ky >= ny) ky -= ny; if (kz >= nz) kz -= nz;
        int a = kx - x; if (a < 0) a += nx;
        float wxa = 0.f, dxa = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (k == a) { wxa = wx[k]; dxa = dx[k]; }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        bool any = false;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            int iy = ky - b; if (iy < 0) iy += ny;
            const int lb = iy - y0;
            if (lb < 0 || lb >= NL) continue;
            any = true;
            float sx = 0.f, sz = 0.f;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int iz = (kz - c) & (nz - 1);
                const float p = phi[iz * NL + lb];
                sx += wz[c] * p; sz += dz[c] * p;
            }
            gx += wy[b] * sx; gy += dy[b] * sx; gz += wy[b] * sz;
        }
        if (!any) continue;
        const float Fx = -q * dxa * gx * nx / Lx, Fy = -q * wxa * gy * ny / Ly, Fz = -q * wxa * gz * nz / Lz;
#ifdef EXP_ZI_NOATOM      // knock-out probe: the gather without its atomics (results wrong on purpose)
        if (Fx != 1.2345e33f) continue;
#endif
        if (FB) {
            atomicAdd(&FB[flat], remd_f2fix(Fx));
            atomicAdd(&FB[fbin_P + flat], remd_f2fix(Fy));
            atomicAdd(&FB[2 * fbin_P + flat], remd_f2fix(Fz));
            continue;
        }
        atomicAdd(&F[i], remd_f2fix(Fx));
        atomicAdd(&F[Npad + i], remd_f2fix(Fy));
        atomicAdd(&F[2 * Npad + i], remd_f2fix(Fz));
    }
}

// The mesh forces from their positions in the bins to the atoms: one thread per position reads (and zeroes, for the next
// evaluation) the three sums the gather left there and adds them to the atom's accumulators -- the one scattered atomic triple
// an atom gets from the mesh.  cap > 0: capped bins (count[nx], float4 entries with the atom index in .w); else the compact
// array of pme_bin_kernel.  Integer sums: the totals are the ones five scattered triples per atom used to give.
__global__ __launch_bounds__(256)
void pme_unbin_forces_kernel(int nx, int cap, int N, int Npad, size_t P, const int* __restrict__ count, const int* __restrict__ atoms,
                             unsigned long long* __restrict__ fbin, long long* __restrict__ force)
{
    const int r = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    int i = -1;
    if (cap > 0) {
        if (p < nx * cap) {
            const int kx = p / cap, sl = p - kx * cap;
            if (sl < min(count[(size_t)r * nx + kx], cap)) i = __float_as_int(reinterpret_cast<const float4*>(atoms)[(size_t)r * nx * cap + p].w);
        }
    } else if (p < N) i = atoms[(size_t)r * Npad + p];
    if (i < 0) return;
    unsigned long long* FB = fbin + (size_t)r * 3 * P;
    const unsigned long long vx = FB[p], vy = FB[P + p], vz = FB[2 * P + p];
    if ((vx | vy | vz) == 0ull) return;
    FB[p] = 0ull; FB[P + p] = 0ull; FB[2 * P + p] = 0ull;
    unsigned long long* F = reinterpret_cast<unsigned long long*>(force + (size_t)r * 3 * Npad);
    atomicAdd(&F[i], vx); atomicAdd(&F[Npad + i], vy); atomicAdd(&F[2 * Npad + i], vz);
}

// Influence function G(kx, ky, kz) = exp(-pi^2 m^2 / alpha^2) / (pi V m^2 |b_x b_y b_z|^2) of every replica's box, laid out
// like the half spectrum.  It only depends on the box, so it is tabulated when a box changes (NVT: once) instead of being
// recomputed (~45 VALU instructions per mesh point incl. an IEEE division and an exp) in every XY pass.
__global__ __launch_bounds__(256)
void pme_influence_table_kernel(int nx, int ny, int nz, const float* __restrict__ bmx, const float* __restrict__ bmy,
                                const float* __restrict__ bmz, const float* __restrict__ box, float alpha, float* __restrict__ infl, int perm_r1)
{
    const int kz = blockIdx.x, r = blockIdx.y, nzc = nz / 2 + 1, np = nx * ny;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const double V = (double)Lx * Ly * Lz;
    const float pref = (float)(1.0 / (M_PI * V));               // charges already carry sqrt(k_e)
    const float fac = (float)(M_PI * M_PI) / (alpha * alpha);
    const float mz = kz / Lz;                                   // kz <= nz/2
    const float bz = bmz[kz];
    float* G = infl + ((size_t)r * nzc + kz) * np;
    for (int idx = threadIdx.x; idx < np; idx += blockDim.x) {
        const int kx = idx / ny, ky = idx - kx * ny;
        const int m0 = (kx <= nx / 2) ? kx : kx - nx, m1 = (ky <= ny / 2) ? ky : ky - ny;
        const float mx = m0 / Lx, my = m1 / Ly;
        const float msq = mx * mx + my * my + mz * mz;
        float g = 0.f;
        if (msq > 0.f) g = pref * __expf(-fac * msq) / (msq * bmx[kx] * bmy[ky] * bz);
        int dst = idx;
        if (perm_r1 > 0) {
            // the order pme_xy_pow2_kernel holds the spectrum in (pme_pow2.h): thread kx * 8 + j, register m * 8 + k2 is
            // ky = j + 8 m + R1 k2
            const int j = ky & 7, hi = ky >> 3, mm = perm_r1 >> 3, m = hi % mm, k2 = hi / mm;
            dst = (m * 8 + k2) * (np / perm_r1) + kx * 8 + j;
        }
        G[dst] = g;
    }
}

// one (kz, replica) plane resident in LDS: forward y, forward x, influence function (+ energy), inverse x, inverse y.
// In-place stages: LDS = nx (ny+1) 8 B (52 KB for 80 x 80) => three workgroups per CU overlap their load / FFT / store.
#define XY_MAX_THREADS 1024
#define XY_PPT 15            // points per thread held in registers: nx * ny <= XY_PPT * XY_THREADS
__global__ __launch_bounds__(XY_MAX_THREADS)
void pme_xy_fused_kernel(fft_plan plx, fft_plan ply, fft_sched scx, fft_sched scy, int nz, float2* __restrict__ spec,
                         const float2* twx, const float2* twy,
                         const float* __restrict__ bmx, const float* __restrict__ bmy, const float* __restrict__ bmz,
                         const float* __restrict__ box, float alpha, int with_energy, double* __restrict__ energy, int n_eblk,
                         const float* __restrict__ infl, int infl_rep)
{
    if (plx.prio) __builtin_amdgcn_s_setprio(PME_PRIO);    // the critical path of the two streams wins issue arbitration (forces.hip: the tuner)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nx = plx.n, ny = ply.n, nzc = nz / 2 + 1;
    const int np = nx * ny;
    const int PS = ny | 1;                                // odd row stride: bank-conflict-free column access
    const int npp = nx * PS;
    float2* buf = reinterpret_cast<float2*>(smem);
    float2* s_twx = buf + npp;                           // twiddle tables staged in LDS
    float2* s_twy = s_twx + nx;
    double* s_e = reinterpret_cast<double*>(s_twy + ny);
    const int kz = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    const int XY_THREADS = blockDim.x;                   // 512 (two workgroups per CU) or 1024 for larger planes
    float2* P = spec + ((size_t)r * nzc + kz) * np;
    const unsigned mny = fft_magic((unsigned)ny);
    const int pad = PS - ny;                               // 0 or 1
    for (int idx = tid; idx < np; idx += XY_THREADS) { const int x = fft_div(idx, mny, ny); buf[idx + x * pad] = P[idx]; }   // x*PS + y
    for (int idx = tid; idx < nx; idx += XY_THREADS) s_twx[idx] = twx[idx];
    for (int idx = tid; idx < ny; idx += XY_THREADS) s_twy[idx] = twy[idx];
    twx = s_twx; twy = s_twy;
    __syncthreads();
    fft_lines_inplace<-1, XY_PPT>(ply, scy, buf, 1, twy, tid, XY_THREADS);      // along y
    fft_lines_inplace<-1, XY_PPT>(plx, scx, buf, PS, twx, tid, XY_THREADS);     // along x
    {
        const float wz = (kz == 0 || 2 * kz == nz) ? 1.f : 2.f;     // Hermitian half: weight of the mirrored plane
        const float* __restrict__ G = infl + ((size_t)r * infl_rep + kz) * np;
        double e_acc = 0.0;
        for (int idx = tid; idx < np; idx += XY_THREADS) {
            const int kx = fft_div(idx, mny, ny);
            const float g = G[idx];
            const float2 sv = buf[idx + kx * pad];              // kx*PS + ky
            if (with_energy) e_acc += 0.5 * (double)(wz * g) * ((double)sv.x * sv.x + (double)sv.y * sv.y);
            buf[idx + kx * pad] = make_float2(sv.x * g, sv.y * g);
        }
        if (with_energy) {
            for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
            if ((tid & 63) == 0) s_e[tid >> 6] = e_acc;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int w = 0; w < XY_THREADS / 64; ++w) tot += s_e[w];
                energy[(size_t)r * n_eblk + kz] = tot;
            }
        }
        __syncthreads();
    }
    fft_lines_inplace<+1, XY_PPT>(plx, scx, buf, PS, twx, tid, XY_THREADS);
    fft_lines_inplace<+1, XY_PPT>(ply, scy, buf, 1, twy, tid, XY_THREADS);
    for (int idx = tid; idx < np; idx += XY_THREADS) { const int x = fft_div(idx, mny, ny); P[idx] = buf[idx + x * pad]; }
}

// Planes too large for the LDS-resident XY pass (144 x 144 complex = 162 KB + tables): the two y passes stay separate
// (contiguous lines), but forward x, influence function (+ energy) and inverse x are fused on a slab of `sw` y columns
// resident in LDS — three sweeps over the half spectrum instead of five.
#define XS_THREADS 256
__global__ __launch_bounds__(XS_THREADS)
void pme_x_fused_kernel(fft_plan plx, fft_sched scx, int ny, int nz, int sw, float2* __restrict__ spec, const float2* twx,
                        int with_energy, double* __restrict__ energy, int n_eblk, const float* __restrict__ infl, int infl_rep)
{
    if (plx.prio) __builtin_amdgcn_s_setprio(PME_PRIO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nx = plx.n, nzc = nz / 2 + 1, nslab = ny / sw, PS = sw | 1;
    float2* buf = reinterpret_cast<float2*>(smem);           // [nx][PS]
    float2* s_twx = buf + nx * PS;
    double* s_e = reinterpret_cast<double*>(s_twx + nx + (nx & 1));
    const int kz = blockIdx.x / nslab, slab = blockIdx.x - kz * nslab, r = blockIdx.y, tid = threadIdx.x;
    const int y0 = slab * sw;
    float2* P = spec + ((size_t)r * nzc + kz) * nx * ny;
    const float* __restrict__ G = infl + ((size_t)r * infl_rep + kz) * nx * ny;
    const unsigned msw = fft_magic((unsigned)sw);
    for (int idx = tid; idx < nx * sw; idx += XS_THREADS) { const int x = fft_div(idx, msw, sw), yy = idx - x * sw; buf[x * PS + yy] = P[x * ny + y0 + yy]; }
    for (int idx = tid; idx < nx; idx += XS_THREADS) s_twx[idx] = twx[idx];
    __syncthreads();
    fft_lines_inplace<-1, XY_PPT>(plx, scx, buf, PS, s_twx, tid, XS_THREADS);
    const float wz = (kz == 0 || 2 * kz == nz) ? 1.f : 2.f;
    double e_acc = 0.0;
    for (int idx = tid; idx < nx * sw; idx += XS_THREADS) {
        const int kx = fft_div(idx, msw, sw), yy = idx - kx * sw;
        const float g = G[kx * ny + y0 + yy];
        const float2 sv = buf[kx * PS + yy];
        if (with_energy) e_acc += 0.5 * (double)(wz * g) * ((double)sv.x * sv.x + (double)sv.y * sv.y);
        buf[kx * PS + yy] = make_float2(sv.x * g, sv.y * g);
    }
    if (with_energy) {
        for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
        if ((tid & 63) == 0) s_e[tid >> 6] = e_acc;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < XS_THREADS / 64; ++w) tot += s_e[w];
            energy[(size_t)r * n_eblk + blockIdx.x] = tot;
        }
    }
    __syncthreads();
    fft_lines_inplace<+1, XY_PPT>(plx, scx, buf, PS, s_twx, tid, XS_THREADS);
    for (int idx = tid; idx < nx * sw; idx += XS_THREADS) { const int x = fft_div(idx, msw, sw), yy = idx - x * sw; P[x * ny + y0 + yy] = buf[x * PS + yy]; }
}

// The y passes of planes too large for the LDS-resident XY pass: a slab of `sh` x rows of one (kz, replica) plane is one
// contiguous block of sh * ny numbers -- loaded once, transformed along y in place on the butterfly schedules, stored once.
// (Round 3: replaces the generic strided-line pass in the force path; DHFR mesh 144^3: 365 / 275 us -> 170 / 159 us per pass.)
template <int SIGN>
__global__ __launch_bounds__(XS_THREADS)
void pme_y_slab_kernel(fft_plan ply, fft_sched scy, int nx, int nz, int sh, float2* __restrict__ spec, const float2* twy)
{
    if (ply.prio) __builtin_amdgcn_s_setprio(PME_PRIO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ny = ply.n, nzc = nz / 2 + 1, nslab = nx / sh, PS = ny | 1;
    float2* buf = reinterpret_cast<float2*>(smem);           // [sh][PS]
    float2* s_twy = buf + sh * PS;
    const int kz = blockIdx.x / nslab, slab = blockIdx.x - kz * nslab, r = blockIdx.y, tid = threadIdx.x;
    float2* P = spec + (((size_t)r * nzc + kz) * nx + (size_t)slab * sh) * ny;
    const unsigned mny = fft_magic((unsigned)ny);
    for (int idx = tid; idx < sh * ny; idx += XS_THREADS) { const int x = fft_div(idx, mny, ny); buf[idx + x * (PS - ny)] = P[idx]; }
    for (int idx = tid; idx < ny; idx += XS_THREADS) s_twy[idx] = twy[idx];
    __syncthreads();
    fft_lines_inplace<SIGN, XY_PPT>(ply, scy, buf, 1, s_twy, tid, XS_THREADS);
    for (int idx = tid; idx < sh * ny; idx += XS_THREADS) { const int x = fft_div(idx, mny, ny); P[idx] = buf[idx + x * (PS - ny)]; }
}

// MODE 0: plain pass.  (kept for the 3-D FFT test hook and as the fall-back for planes larger than the LDS)
template <int SIGN, int MODE>
__global__ __launch_bounds__(FFT_B * FFT_T)
void fft_pass_kernel(fft_plan pl, float2* __restrict__ grid, size_t rep_stride, int es, int lines_per_rep,
                     int line_div, size_t line_hi_stride, int contiguous, const float2* __restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = pl.n;
    float2* bufA = reinterpret_cast<float2*>(smem);
    float2* bufB = bufA + FFT_B * n;
    const int r = blockIdx.y;
    const int l0 = blockIdx.x * FFT_B;
    float2* G = grid + (size_t)r * rep_stride;
    const int tid = threadIdx.x;
    auto base = [&](int l) -> size_t { return (size_t)(l / line_div) * line_hi_stride + (size_t)(l % line_div) * (contiguous ? (size_t)n : 1); };
    if (contiguous) {
        for (int idx = tid; idx < FFT_B * n; idx += FFT_B * FFT_T) {
            const int b = idx / n, e = idx % n;
            const int l = l0 + b;
            bufA[b * n + e] = (l < lines_per_rep) ? G[base(l) + e] : make_float2(0.f, 0.f);
        }
    } else {
        for (int idx = tid; idx < FFT_B * n; idx += FFT_B * FFT_T) {
            const int e = idx / FFT_B, b = idx % FFT_B;
            const int l = l0 + b;
            bufA[b * n + e] = (l < lines_per_rep) ? G[base(l) + (size_t)e * es] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    float2* res = fft_lines_lds<SIGN>(pl, bufA, bufB, FFT_B, n, 1, tw, tid, FFT_B * FFT_T);
    if (contiguous) {
        for (int idx = tid; idx < FFT_B * n; idx += FFT_B * FFT_T) {
            const int b = idx / n, e = idx % n;
            const int l = l0 + b;
            if (l < lines_per_rep) G[base(l) + e] = res[b * n + e];
        }
    } else {
        for (int idx = tid; idx < FFT_B * n; idx += FFT_B * FFT_T) {
            const int e = idx / FFT_B, b = idx % FFT_B;
            const int l = l0 + b;
            if (l < lines_per_rep) G[base(l) + (size_t)e * es] = res[b * n + e];
        }
    }
}

// fall-back influence-function pass for meshes whose (x,y) plane does not fit the LDS
__global__ __launch_bounds__(256)
void pme_influence_kernel(int nx, int ny, int nz, float2* __restrict__ spec, const float* __restrict__ bmx,
                          const float* __restrict__ bmy, const float* __restrict__ bmz, const float* __restrict__ box,
                          float alpha, int with_energy, double* __restrict__ energy, int n_eblk)
{
    __shared__ double s_e[4];
    const int kz = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nzc = nz / 2 + 1, np = nx * ny;
    float2* P = spec + ((size_t)r * nzc + kz) * np;
    const float Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const double V = (double)Lx * Ly * Lz;
    const float pref = (float)(1.0 / (M_PI * V));
    const float fac = (float)(M_PI * M_PI) / (alpha * alpha);
    const float mz = kz / Lz, bz = bmz[kz];
    const float wz = (kz == 0 || 2 * kz == nz) ? 1.f : 2.f;
    double e_acc = 0.0;
    for (int idx = tid; idx < np; idx += 256) {
        const int kx = idx / ny, ky = idx - kx * ny;
        const int m0 = (kx <= nx / 2) ? kx : kx - nx, m1 = (ky <= ny / 2) ? ky : ky - ny;
        const float mx = m0 / Lx, my = m1 / Ly;
        const float msq = mx * mx + my * my + mz * mz;
        float g = 0.f;
        if (msq > 0.f) g = pref * __expf(-fac * msq) / (msq * bmx[kx] * bmy[ky] * bz);
        const float2 sv = P[idx];
        if (with_energy) e_acc += 0.5 * (double)(wz * g) * ((double)sv.x * sv.x + (double)sv.y * sv.y);
        P[idx] = make_float2(sv.x * g, sv.y * g);
    }
    if (with_energy) {
        for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
        if ((tid & 63) == 0) s_e[tid >> 6] = e_acc;
        __syncthreads();
        if (tid == 0) energy[(size_t)r * n_eblk + kz] = s_e[0] + s_e[1] + s_e[2] + s_e[3];
    }
}

__global__ void pme_energy_reduce_kernel(int n_eblk, const double* __restrict__ e, double* __restrict__ epart, int n_epart, int slot)
{
    const int r = blockIdx.x;
    double acc = 0.0;
    for (int t = threadIdx.x; t < n_eblk; t += 64) acc += e[(size_t)r * n_eblk + t];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) epart[(size_t)r * n_epart + slot] = acc;
}

// ---------------------------------------------------------------------------------------------------
static bool factorize(int n, int* radix, int& nrad)
{
    nrad = 0;
    int m = n;
    // the power of two 2^e in ceil(e / 3) stages of radix 8 / 4 / 2, bits spread evenly (16 = 4 x 4, 32 = 8 x 4, 64 = 8 x 8,
    // 128 = 8 x 4 x 4): every stage is a pass over the LDS image and two workgroup barriers.  REMD_PME_RADIX8=0: 4s and 2s only.
    static const bool radix8 = !(getenv("REMD_PME_RADIX8") && atoi(getenv("REMD_PME_RADIX8")) == 0);
    int e = 0;
    while (m % 2 == 0) { ++e; m /= 2; }
    if (radix8) {
        const int st = (e + 2) / 3;
        for (int k = 0, left = e; k < st; ++k) { const int b = (left + (st - k) - 1) / (st - k); radix[nrad++] = 1 << b; left -= b; }
    } else {
        while (e >= 2) { radix[nrad++] = 4; e -= 2; }
        if (e) radix[nrad++] = 2;
    }
    while (m % 3 == 0) { radix[nrad++] = 3; m /= 3; }
    while (m % 5 == 0) { radix[nrad++] = 5; m /= 5; }
    return m == 1 && nrad <= 8;
}

static double bspline_M(int order, double u)
{
    if (order == 2) return (u < 0 || u > 2) ? 0.0 : 1.0 - fabs(u - 1.0);
    return u / (order - 1) * bspline_M(order - 1, u) + (order - u) / (order - 1) * bspline_M(order - 1, u - 1.0);
}

int remd_pme_destroy(remd_ctx* h)
{
    pme_state* s = (pme_state*)h->pme;
    if (!s) return 0;
    if (s->d_grid) hipFree(s->d_grid);
    if (s->d_col_count) hipFree(s->d_col_count); if (s->d_col_start) hipFree(s->d_col_start); if (s->d_cursor) hipFree(s->d_cursor);
    if (s->d_atom_col) hipFree(s->d_atom_col); if (s->d_col_atoms) hipFree(s->d_col_atoms);
    for (int k = 0; k < 4; ++k) if (s->d_tw[k]) hipFree(s->d_tw[k]);
    for (int k = 0; k < 3; ++k) if (s->d_bmod[k]) hipFree(s->d_bmod[k]);
    if (s->d_energy) hipFree(s->d_energy);
    if (s->d_infl) hipFree(s->d_infl);
    if (s->d_fbin) hipFree(s->d_fbin);
    if (s->d_cbin_count) hipFree(s->d_cbin_count); if (s->d_cbin_atoms) hipFree(s->d_cbin_atoms); if (s->d_cbin_q) hipFree(s->d_cbin_q);
    for (int k = 0; k < 3; ++k) if (s->d_sched[k]) hipFree(s->d_sched[k]);
    delete s;
    h->pme = nullptr;
    return 0;
}

static fft_plan make_plan(pme_state* s, int axis)
{
    fft_plan pl; pl.n = s->n[axis]; pl.nrad = s->nrad[axis]; pl.prio = s->prio_hi ? 1 : 0;
    int Ns = 1;
    for (int k = 0; k < 8; ++k) {
        pl.radix[k] = s->radix[axis][k];
        pl.mnb[k] = pl.mNs[k] = 0;
        if (k < pl.nrad) { pl.mnb[k] = fft_magic((unsigned)(pl.n / pl.radix[k])); pl.mNs[k] = fft_magic((unsigned)Ns); Ns *= pl.radix[k]; }
    }
    return pl;
}

// host mirror of the index arithmetic of one in-place pass (element e of line l at l*ls + e*es, consecutive threads
// take consecutive lines): fills the butterfly schedule read by fft_stage_sched
static int build_sched(remd_ctx* h, pme_state* s, int axis, int nlines, int ls, int es, int nthreads, int ppt,
                       fft_sched* out, uint2** d_tab)
{
    const int n = s->n[axis];
    std::vector<uint2> tab;
    int Ns = 1;
    for (int st = 0; st < s->nrad[axis]; ++st) {
        const int Rx = s->radix[axis][st];
        const int NB = (ppt + Rx - 1) / Rx;
        const int nb = n / Rx, total = nlines * nb, tstride = n / (Ns * Rx);
        if ((long long)NB * nthreads < total) return remd_fail(h, -3, "PME FFT pass does not fit the workgroup registers");
        out->off[st] = (int)tab.size();
        const size_t base = tab.size();
        tab.resize(base + (size_t)NB * nthreads, make_uint2(0xffffffffu, 0u));
        if ((nlines - 1) * ls + (n - 1) * es > 0xffff) return remd_fail(h, -3, "PME plane too large for the FFT schedule");
        for (int b = 0; b < NB; ++b)
            for (int t = 0; t < nthreads; ++t) {
                const int idx = t + b * nthreads;
                if (idx >= total) continue;
                const int l = idx % nlines, j = idx / nlines;           // consecutive threads take consecutive lines
                const int jq = j / Ns, k = j % Ns;
                const int src = l * ls + j * es, dst = l * ls + (jq * Ns * Rx + k) * es;
                tab[base + (size_t)b * nthreads + t] = make_uint2((unsigned)src | ((unsigned)dst << 16), (unsigned)(k * tstride));
            }
        Ns *= Rx;
    }
    if (*d_tab) { hipFree(*d_tab); *d_tab = nullptr; }
    REMD_CHECK(h, hipMalloc(d_tab, sizeof(uint2) * tab.size()));
    REMD_CHECK(h, hipMemcpy(*d_tab, tab.data(), sizeof(uint2) * tab.size(), hipMemcpyHostToDevice));
    out->tab = *d_tab;
    return 0;
}

// full_complex: allocate the [R][nx][ny][nz] complex grid of the FFT test hook instead of the PME buffers
static int pme_setup_impl(remd_ctx* h, bool full_complex)
{
    remd_pme_destroy(h);
    pme_state* s = new pme_state();
    h->pme = s;
    for (int k = 0; k < 3; ++k) {
        s->n[k] = h->grid[k];
        if (s->n[k] < 6 || s->n[k] > 256 || !factorize(s->n[k], s->radix[k], s->nrad[k]))
            return remd_fail(h, -3, "PME mesh sizes must be products of 2, 3, 5 between 6 and 256");
    }
    s->n[3] = s->n[2] / 2;
    s->z_half = (s->n[2] % 2 == 0) && s->n[3] >= 3 && factorize(s->n[3], s->radix[3], s->nrad[3]) &&
                true;
    s->R = h->R;
    s->npts = (size_t)s->n[0] * s->n[1] * s->n[2];
    s->nzc = s->n[2] / 2 + 1;
    s->nspec = (size_t)s->nzc * s->n[0] * s->n[1];
    if (full_complex) {
        REMD_CHECK(h, hipMalloc(&s->d_grid, sizeof(float2) * s->npts * s->R));
    } else {
        s->ready = true;
        REMD_CHECK(h, hipMalloc(&s->d_grid, sizeof(float2) * s->nspec * s->R));
        REMD_CHECK(h, hipMalloc(&s->d_col_start, sizeof(int) * (size_t)(s->n[0] + 1) * s->R));
        REMD_CHECK(h, hipMalloc(&s->d_col_atoms, sizeof(int) * (size_t)h->Npad * s->R));
        if (!full_complex && !(getenv("REMD_PME_CHAINBIN") && atoi(getenv("REMD_PME_CHAINBIN")) == 0)) {
            // four times the mean occupancy of a mesh column (x bins are 1 / nx of a homogeneous box): overflow is detected
            s->cbin_cap = std::min(h->Npad, std::max(64, 4 * ((h->N + s->n[0] - 1) / s->n[0])));
            if (getenv("REMD_PME_CBIN_CAP")) s->cbin_cap = std::max(1, atoi(getenv("REMD_PME_CBIN_CAP")));      // test hook: provoke the overflow path
            REMD_CHECK(h, hipMalloc(&s->d_cbin_count, sizeof(int) * 2 * (size_t)s->R * s->n[0]));
            REMD_CHECK(h, hipMemset(s->d_cbin_count, 0, sizeof(int) * 2 * (size_t)s->R * s->n[0]));
            REMD_CHECK(h, hipMalloc(&s->d_cbin_atoms, sizeof(float4) * (size_t)s->R * s->n[0] * s->cbin_cap));
            REMD_CHECK(h, hipMalloc(&s->d_cbin_q, sizeof(float) * (size_t)s->R * s->n[0] * s->cbin_cap));
        }
    }
    if (s->z_half) {
        const int n = s->n[3];
        std::vector<float2> tw(n);
        for (int j = 0; j < n; ++j) tw[j] = make_float2((float)cos(2.0 * M_PI * j / n), (float)(-sin(2.0 * M_PI * j / n)));
        REMD_CHECK(h, hipMalloc(&s->d_tw[3], sizeof(float2) * n));
        REMD_CHECK(h, hipMemcpy(s->d_tw[3], tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    }
    for (int k = 0; k < 3; ++k) {
        const int n = s->n[k];
        std::vector<float2> tw(n);
        for (int j = 0; j < n; ++j) tw[j] = make_float2((float)cos(2.0 * M_PI * j / n), (float)(-sin(2.0 * M_PI * j / n)));
        REMD_CHECK(h, hipMalloc(&s->d_tw[k], sizeof(float2) * n));
        REMD_CHECK(h, hipMemcpy(s->d_tw[k], tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
        // |sum_{k=0}^{order-2} M_n(k+1) exp(2 pi i m k / K)|^2  (Essmann eq. 4.4 denominator)
        std::vector<float> bm(n);
        for (int m = 0; m < n; ++m) {
            double re = 0, im = 0;
            for (int k2 = 0; k2 <= PME_ORDER - 2; ++k2) {
                const double w = bspline_M(PME_ORDER, k2 + 1.0), arg = 2.0 * M_PI * m * k2 / n;
                re += w * cos(arg); im += w * sin(arg);
            }
            bm[m] = (float)(re * re + im * im);
        }
        // odd spline orders have a zero of the modulus at m = K/2: replace it by the mean of its neighbours
        for (int m = 0; m < n; ++m)
            if (bm[m] < 1e-7f) bm[m] = 0.5f * (bm[(m + n - 1) % n] + bm[(m + 1) % n]);
        REMD_CHECK(h, hipMalloc(&s->d_bmod[k], sizeof(float) * n));
        REMD_CHECK(h, hipMemcpy(s->d_bmod[k], bm.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    }
    s->n_eblk = s->nzc;
    REMD_CHECK(h, hipMalloc(&s->d_energy, sizeof(double) * (size_t)s->n_eblk * s->R));
    s->xy_lds = sizeof(float2) * ((size_t)s->n[0] * (s->n[1] | 1) + s->n[0] + s->n[1]) + 128;
    // workgroup size of the XY kernel: every stage issues ceil(butterflies / threads) butterfly slots per thread,
    // masked-off slots still cost issue cycles, so pick the wavefront count that wastes the fewest (75 x 75: 384 threads
    // run 1125 radix-5 and 1875 radix-3 butterflies in 3 and 5 full slots; 512 threads would idle a quarter of them)
    {
        const long long np = (long long)s->n[0] * s->n[1];
        long long best_cost = -1; int best_t = 0;
        // multiples of 256 threads only: the dispatcher reserves ceil(waves / 4) wave slots on EVERY SIMD for a workgroup
        // (tools/probes/occupancy_probe.hip), so a 6-wavefront workgroup occupies the slots of 8; next to the pair kernel's
        // resident workgroups 512 threads beat the 384 that waste the fewest butterfly slots (108.3 vs 110.2 ms per 500 steps)
        // (ties go to 512 threads: 64 x 64 planes measured 92.7 ms per 500 steps with 512 against 94.9 with 256 and the same issued lane-points)
        const int xy_cands[4] = {512, 256, 768, 1024};
        for (int tc = 0; tc < 4; ++tc) {
            const int t = xy_cands[tc];
            // (what bounds a stage is its butterfly slots per thread, ceil(XY_PPT / radix) of them: a 128 x 128 plane is 16 points
            // per thread of a 1024-thread workgroup = 4 radix-4 slots, although 16 > XY_PPT)
            long long cost = 0; bool ok = true;
            for (int ax = 0; ax < 2 && ok; ++ax)
                for (int st = 0; st < s->nrad[ax]; ++st) {
                    const int rx = s->radix[ax][st];
                    const long long slots = (np / rx + t - 1) / t;
                    if (slots > (XY_PPT + rx - 1) / rx) ok = false;
                    cost += slots * t * rx;                    // issued lane-points of this stage
                }
            if (ok && (best_cost < 0 || cost < best_cost)) { best_cost = cost; best_t = t; }
        }
        if (getenv("REMD_PME_XYT")) {            // experiment hook: workgroup size of the plane pass (must leave <= ceil(XY_PPT / radix) slots per thread)
            const int t = atoi(getenv("REMD_PME_XYT"));
            bool ok = t >= 64 && t <= 1024 && t % 64 == 0;
            for (int ax = 0; ax < 2 && ok; ++ax)
                for (int st = 0; st < s->nrad[ax]; ++st) { const int rx = s->radix[ax][st]; if ((np / rx + t - 1) / t > (XY_PPT + rx - 1) / rx) ok = false; }
            if (ok) best_t = t;
        }
        s->xy_threads = best_t > 0 ? best_t : 1024;
        s->xy_fused = best_t > 0 && s->xy_lds <= 160 * 1024;
        h->xy_lds_bytes = s->xy_fused ? s->xy_lds : 0;
    }
    if (!s->xy_fused && !full_complex) {
        // widest divisor of ny whose slab fits the registers of 256 threads and ~40 KB of LDS
        for (int c = 1; c <= s->n[1]; ++c)
            if (s->n[1] % c == 0 && (long long)s->n[0] * c <= (long long)XY_PPT * XS_THREADS && (size_t)s->n[0] * (c | 1) * 8 <= 40 * 1024) s->xs_sw = c;
        if (s->xs_sw > 0) {
            int rc = build_sched(h, s, 0, s->xs_sw, 1, s->xs_sw | 1, XS_THREADS, XY_PPT, &s->sch_x, &s->d_sched[0]);
            if (rc) return rc;
            s->n_eblk = s->nzc * (s->n[1] / s->xs_sw);
            hipFree(s->d_energy); s->d_energy = nullptr;
            REMD_CHECK(h, hipMalloc(&s->d_energy, sizeof(double) * (size_t)s->n_eblk * s->R));
            REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_x_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            // the y passes on slabs of x rows: tallest divisor of nx that fits the same registers and ~40 KB of LDS
            for (int c = 1; c <= s->n[0]; ++c)
                if (s->n[0] % c == 0 && (long long)s->n[1] * c <= (long long)XY_PPT * XS_THREADS && (size_t)c * (s->n[1] | 1) * 8 <= 40 * 1024) s->ys_sh = c;
            if (s->ys_sh > 0) {
                rc = build_sched(h, s, 1, s->ys_sh, s->n[1] | 1, 1, XS_THREADS, XY_PPT, &s->sch_y, &s->d_sched[1]);
                if (rc) return rc;
                REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_y_slab_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_y_slab_kernel<+1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            }
        }
    }
    // square power-of-two planes: the register-resident pass (REMD_PME_POW2=0: the scheduled mixed-radix passes; 1 / 2: plane pass / z passes only)
    if (s->xy_fused && !full_complex && s->n[0] == s->n[1] && (s->n[0] == 64 || s->n[0] == 128) &&
        !(getenv("REMD_PME_POW2") && !(atoi(getenv("REMD_PME_POW2")) & 1))) {                 // bit 0: the plane pass
        s->xy_pow2 = s->n[0];
        s->xy_threads = s->n[0] == 64 ? 512 : 1024;
        s->xy_lds = sizeof(float2) * ((size_t)s->n[0] * (s->n[0] + 8) + s->n[0]) + sizeof(double) * 16;
        h->xy_lds_bytes = s->xy_lds;
        REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_pow2_kernel<64, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_pow2_kernel<64, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_pow2_kernel<128, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_pow2_kernel<128, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (s->xy_fused && !full_complex && !s->xy_pow2) {
        REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->xy_lds));
        const int PS = s->n[1] | 1;
        int rc = build_sched(h, s, 1, s->n[0], PS, 1, s->xy_threads, XY_PPT, &s->sch_y, &s->d_sched[1]);      // along y: lines = x rows
        if (!rc) rc = build_sched(h, s, 0, s->n[1], 1, PS, s->xy_threads, XY_PPT, &s->sch_x, &s->d_sched[0]);  // along x: lines = y columns
        if (rc) return rc;
    }
#define Z_LDS_ATTR(ZT, HF) \
    REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_spread_zfwd_kernel<ZT, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_zinv_gather_kernel<ZT, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Z_LDS_ATTR(512, false) Z_LDS_ATTR(256, false) Z_LDS_ATTR(512, true) Z_LDS_ATTR(256, true)
#undef Z_LDS_ATTR
    // the fills above ran on the null stream; the handle's stream is non-blocking and does not wait for it
    REMD_CHECK(h, hipDeviceSynchronize());
    return 0;
}

int remd_pme_setup(remd_ctx* h) { return pme_setup_impl(h, false); }

template <int SIGN>
static void launch_pass(remd_ctx* h, pme_state* s, float2* data, size_t rep_stride, int axis_n_index, int es, int lines,
                        int line_div, size_t hi, int contiguous)
{
    fft_plan pl = make_plan(s, axis_n_index);
    dim3 grid((lines + FFT_B - 1) / FFT_B, s->R);
    const size_t lds = sizeof(float2) * 2 * FFT_B * pl.n;
    hipLaunchKernelGGL((fft_pass_kernel<SIGN, 0>), grid, dim3(FFT_B * FFT_T), lds, s->stream ? s->stream : h->stream, pl, data, rep_stride, es, lines,
                       line_div, hi, contiguous, s->d_tw[axis_n_index]);
}

const float* remd_nb_rep_lam(remd_ctx* h);
const float4* remd_nb_param(remd_ctx* h);

int remd_pme_forces(remd_ctx* h, bool with_energy, hipStream_t st, int part)
{
    pme_state* s = (pme_state*)h->pme;
    if (!s || s->R != h->R || !s->ready) { int rc = remd_pme_setup(h); if (rc) return rc; s = (pme_state*)h->pme; }
    s->stream = st;
    s->prio_hi = h->mesh_prio_hi;
    const int nx = s->n[0], ny = s->n[1], nz = s->n[2];
    const float* rep_lam = remd_nb_rep_lam(h);
    const float4* param = remd_nb_param(h);
    if (part & 1) {
    // bins from the integrator chain (h->cbins_ready: the chain launched just before this evaluation filled buffer cbin_parity)
    s->cbin_use = h->cbins_ready && s->d_cbin_count;
    h->cbins_ready = false;
    if (!s->cbin_use)
    {
        remd_prof_scope ps(h, "pme_bin", st);
        hipLaunchKernelGGL(pme_bin_kernel, dim3(h->R), dim3(1024), 0, st, h->N, h->Npad, nx, ny, nz, h->d_pos, h->d_box,
                           s->d_col_start, s->d_col_atoms, h->fork_seq_pending ? h->d_sync : (unsigned int*)nullptr, h->fork_seq_pending);
        h->fork_seq_pending = 0;
    }
    {
        remd_prof_scope ps(h, "pme_fft", st);
        // lines per workgroup: a divisor of ny whose points fit the registers of the workgroup
        const bool half = s->z_half;
        const int zaxis = half ? 3 : 2;                  // the transform that is run: nz/2 packed points or nz points
        const int M = s->n[zaxis], PZ = half ? ((M + 1) | 1) : (nz | 1);
        // 256 threads when a whole mesh row still fits their registers (fuller butterfly slots, 8 workgroups per CU;
        // measured 6.75 -> 6.95 it/s on the 75 x 75 x 72 mesh), else 512
        const int ZT = (long long)ny * M <= (long long)Z_PPT * 256 ? 256 : 512;
        int nl = 1;
        for (int c = 1; c <= ny; ++c) if (ny % c == 0 && c * M <= Z_PPT * ZT) nl = c;
        const size_t zlds = sizeof(float2) * ((size_t)nl * PZ + nz + (half ? M : 0));
        const dim3 zgrid(nx * ((ny + nl - 1) / nl), s->R);
        if (s->sch_nl != nl || s->sch_zt != ZT) {
            int rc = build_sched(h, s, zaxis, nl, PZ, 1, ZT, Z_PPT, &s->sch_z, &s->d_sched[2]);
            if (rc) return rc;
            s->sch_nl = nl; s->sch_zt = ZT;
        }
#define LAUNCH_Z(KERN, ZTT, HF, ...) hipLaunchKernelGGL((KERN<ZTT, HF>), zgrid, dim3(ZTT), zlds, st, make_plan(s, zaxis), s->sch_z, nl, nx, ny, __VA_ARGS__)
#define DISPATCH_Z(KERN, ...) do { if (ZT == 256) { if (half) LAUNCH_Z(KERN, 256, true, __VA_ARGS__); else LAUNCH_Z(KERN, 256, false, __VA_ARGS__); } \
                                   else { if (half) LAUNCH_Z(KERN, 512, true, __VA_ARGS__); else LAUNCH_Z(KERN, 512, false, __VA_ARGS__); } } while (0)
        // bins: compact arrays of the binning launch, or the capped ones the integrator chain filled (buffer cbin_parity; this
        // pass zeroes the other buffer for the next evaluation and, with no binning launch in front, publishes the fork)
        const int* bin_cs = s->cbin_use ? s->d_cbin_count + (size_t)s->cbin_parity * s->R * nx : s->d_col_start;
        const int* bin_ca = s->cbin_use ? reinterpret_cast<const int*>(s->d_cbin_atoms) : s->d_col_atoms;
        const int bin_cap = s->cbin_use ? s->cbin_cap : 0;
        int* bin_zero = s->cbin_use ? s->d_cbin_count + (size_t)(1 - s->cbin_parity) * s->R * nx : (int*)nullptr;
        unsigned int* fflag = (s->cbin_use && h->fork_seq_pending) ? h->d_sync : (unsigned int*)nullptr;
        const unsigned int fseq = h->fork_seq_pending;
        if (s->cbin_use) { h->fork_seq_pending = 0; }
        // listed terms riding in this launch (forces.hip decides: remd_ctx::mesh_listed_total)
        const int n_mesh_blocks = (int)zgrid.x;
        const int n_listed_blocks = h->mesh_listed_total > 0 ? (h->mesh_listed_total + ZT - 1) / ZT : 0;
        // power-of-two z: 64 lines per workgroup on the register transforms (REMD_PME_POW2=0: the scheduled passes)
        const bool pow2_env = !(getenv("REMD_PME_POW2") && !(atoi(getenv("REMD_PME_POW2")) & 2));      // bit 1: the z passes
        const int zp2 = (pow2_env && half && (nz == 64 || nz == 128) && ny % 64 == 0 && nl == 64 && ZT == 64 * (nz / 16)) ? nz : 0;
        const size_t zlds2 = sizeof(float2) * (size_t)(nz / 2 + 1) * 64;
        if (zp2) {
            const dim3 zg(zgrid.x + n_listed_blocks, zgrid.y);
#define LAUNCH_ZF2(MMM, RR) hipLaunchKernelGGL((pme_spread_zfwd_pow2_kernel<MMM, RR>), zg, dim3(64 * RR), zlds2, st, nx, ny, h->Npad, h->d_pos, param, h->d_box, rep_lam, \
                       bin_cs, bin_ca, s->d_grid, s->d_tw[2], s->d_tw[3], bin_cap, bin_zero, fflag, fseq, \
                       (s->cbin_use && !rep_lam) ? s->d_cbin_q : (const float*)nullptr, h->mesh_listed, n_mesh_blocks, h->d_force, s->prio_hi ? 1 : 0)
            if (zp2 == 64) LAUNCH_ZF2(32, 4); else LAUNCH_ZF2(64, 8);
#undef LAUNCH_ZF2
        } else
        {
            const dim3 zgrid_keep = zgrid;
            const dim3 zgrid(zgrid_keep.x + n_listed_blocks, zgrid_keep.y);
            DISPATCH_Z(pme_spread_zfwd_kernel, h->Npad, h->d_pos, param, h->d_box, rep_lam, bin_cs, bin_ca, s->d_grid,
                       s->d_tw[2], s->d_tw[3], bin_cap, bin_zero, fflag, fseq, (s->cbin_use && !rep_lam) ? s->d_cbin_q : (const float*)nullptr,
                       h->mesh_listed, n_mesh_blocks, h->d_force);
        }
        h->mesh_listed_total = 0;
        if (s->xy_fused || s->xs_sw > 0) {
            if (!s->d_infl) REMD_CHECK(h, hipMalloc(&s->d_infl, sizeof(float) * s->nspec * s->R));
            if (s->infl_version != h->box_version) {
                // one table serves every replica while all boxes are the same (constant volume): 1 / R of the table traffic
                hipLaunchKernelGGL(pme_influence_table_kernel, dim3(s->nzc, h->box_uniform ? 1 : s->R), dim3(256), 0, st, nx, ny, nz, s->d_bmod[0], s->d_bmod[1],
                                   s->d_bmod[2], h->d_box, (float)h->ewald_alpha, s->d_infl, s->xy_pow2 == 64 ? 8 : s->xy_pow2 == 128 ? 16 : 0);
                s->infl_version = h->box_version;
                s->infl_rep = h->box_uniform ? 0 : s->nzc;
            }
        }
        if (s->xy_pow2) {
            remd_prof_scope pxy(h, "pme_xy", st);
#define LAUNCH_XY_P2(NN, RR, TT, WE) hipLaunchKernelGGL((pme_xy_pow2_kernel<NN, RR, WE>), dim3(s->nzc, s->R), dim3(TT), s->xy_lds, st, nz, s->d_grid, \
                                   s->d_tw[0], s->d_energy, s->n_eblk, s->d_infl, s->infl_rep, s->prio_hi ? 1 : 0)
            if (s->xy_pow2 == 64) { if (with_energy) LAUNCH_XY_P2(64, 8, 512, true); else LAUNCH_XY_P2(64, 8, 512, false); }
            else { if (with_energy) LAUNCH_XY_P2(128, 16, 1024, true); else LAUNCH_XY_P2(128, 16, 1024, false); }
#undef LAUNCH_XY_P2
            if (h->pair_after_xy && h->ev_xy) { hipEventRecord(h->ev_xy, st); h->xy_recorded = true; }
        } else if (s->xy_fused) {
            remd_prof_scope pxy(h, "pme_xy", st);
            hipLaunchKernelGGL(pme_xy_fused_kernel, dim3(s->nzc, s->R), dim3(s->xy_threads), s->xy_lds, st, make_plan(s, 0), make_plan(s, 1),
                               s->sch_x, s->sch_y, nz, s->d_grid, s->d_tw[0], s->d_tw[1], s->d_bmod[0], s->d_bmod[1], s->d_bmod[2], h->d_box,
                               (float)h->ewald_alpha, with_energy ? 1 : 0, s->d_energy, s->n_eblk, s->d_infl, s->infl_rep);
            // the pair kernel of this evaluation is held back until the plane pass has ENDED (remd_compute_forces waits for this event on
            // the direct-space stream): planes that take a CU's whole LDS cannot be placed beside resident pair workgroups
            if (h->pair_after_xy && h->ev_xy) { hipEventRecord(h->ev_xy, st); h->xy_recorded = true; }
        } else if (s->xs_sw > 0) {
            // spec layout [kz][x][y]: y passes on contiguous lines, then the fused x pass on LDS-resident y slabs
            const size_t ylds = sizeof(float2) * ((size_t)s->ys_sh * (ny | 1) + ny + 2);
            if (s->ys_sh > 0)
                hipLaunchKernelGGL(pme_y_slab_kernel<-1>, dim3(s->nzc * (nx / s->ys_sh), s->R), dim3(XS_THREADS), ylds, st, make_plan(s, 1), s->sch_y,
                                   nx, nz, s->ys_sh, s->d_grid, s->d_tw[1]);
            else
                launch_pass<-1>(h, s, s->d_grid, s->nspec, 1, 1, s->nzc * nx, s->nzc * nx, 0, 1);
            const int PS = s->xs_sw | 1;
            const size_t lds = sizeof(float2) * ((size_t)nx * PS + nx + 2) + 64;
            hipLaunchKernelGGL(pme_x_fused_kernel, dim3(s->nzc * (ny / s->xs_sw), s->R), dim3(XS_THREADS), lds, st, make_plan(s, 0), s->sch_x,
                               ny, nz, s->xs_sw, s->d_grid, s->d_tw[0], with_energy ? 1 : 0, s->d_energy, s->n_eblk, s->d_infl, s->infl_rep);
            if (s->ys_sh > 0)
                hipLaunchKernelGGL(pme_y_slab_kernel<+1>, dim3(s->nzc * (nx / s->ys_sh), s->R), dim3(XS_THREADS), ylds, st, make_plan(s, 1), s->sch_y,
                                   nx, nz, s->ys_sh, s->d_grid, s->d_tw[1]);
            else
                launch_pass<+1>(h, s, s->d_grid, s->nspec, 1, 1, s->nzc * nx, s->nzc * nx, 0, 1);
        } else {
            // spec layout [kz][x][y]: y lines contiguous, x lines strided by ny
            launch_pass<-1>(h, s, s->d_grid, s->nspec, 1, 1, s->nzc * nx, s->nzc * nx, 0, 1);
            launch_pass<-1>(h, s, s->d_grid, s->nspec, 0, ny, s->nzc * ny, ny, (size_t)nx * ny, 0);
            hipLaunchKernelGGL(pme_influence_kernel, dim3(s->nzc, s->R), dim3(256), 0, st, nx, ny, nz, s->d_grid, s->d_bmod[0],
                               s->d_bmod[1], s->d_bmod[2], h->d_box, (float)h->ewald_alpha, with_energy ? 1 : 0, s->d_energy, s->n_eblk);
            launch_pass<+1>(h, s, s->d_grid, s->nspec, 0, ny, s->nzc * ny, ny, (size_t)nx * ny, 0);
            launch_pass<+1>(h, s, s->d_grid, s->nspec, 1, 1, s->nzc * nx, s->nzc * nx, 0, 1);
        }
        // mesh forces by bin position + one hand-over to the atoms (REMD_PME_FBIN=0: five scattered atomic triples per atom)
        const bool fbin_env = !(getenv("REMD_PME_FBIN") && atoi(getenv("REMD_PME_FBIN")) == 0);
        if (fbin_env && !s->d_fbin) {
            s->fbin_P = std::max((size_t)nx * (size_t)std::max(s->cbin_cap, 0), (size_t)h->Npad);
            REMD_CHECK(h, hipMalloc(&s->d_fbin, sizeof(unsigned long long) * 3 * s->fbin_P * s->R));
            REMD_CHECK(h, hipMemsetAsync(s->d_fbin, 0, sizeof(unsigned long long) * 3 * s->fbin_P * s->R, st));
        }
        unsigned long long* fbin = fbin_env ? s->d_fbin : (unsigned long long*)nullptr;
        {
        remd_prof_scope pzg(h, "pme_zinv_gather", st);
        if (zp2) {
#define LAUNCH_ZI2(MMM, RR) hipLaunchKernelGGL((pme_zinv_gather_pow2_kernel<MMM, RR>), zgrid, dim3(64 * RR), zlds2, st, nx, ny, s->d_grid, s->d_tw[2], s->d_tw[3], \
                       h->Npad, h->d_pos, param, h->d_box, rep_lam, bin_cs, bin_ca, h->d_force, bin_cap, \
                       (s->cbin_use && !rep_lam) ? s->d_cbin_q : (const float*)nullptr, s->prio_hi ? 1 : 0, fbin, s->fbin_P)
            if (zp2 == 64) LAUNCH_ZI2(32, 4); else LAUNCH_ZI2(64, 8);
#undef LAUNCH_ZI2
        } else
        DISPATCH_Z(pme_zinv_gather_kernel, s->d_grid, s->d_tw[2], s->d_tw[3], h->Npad, h->d_pos, param, h->d_box, rep_lam,
                   bin_cs, bin_ca, h->d_force, bin_cap, (s->cbin_use && !rep_lam) ? s->d_cbin_q : (const float*)nullptr, fbin, s->fbin_P);
        if (fbin) {
            const int npos = bin_cap > 0 ? nx * bin_cap : h->N;
            hipLaunchKernelGGL(pme_unbin_forces_kernel, dim3((npos + 255) / 256, s->R), dim3(256), 0, st, nx, bin_cap, h->N, h->Npad, s->fbin_P,
                               bin_cs, bin_ca, fbin, h->d_force);
        }
        if (s->cbin_use) s->cbin_parity ^= 1;          // the next chain fills the buffer this evaluation has just zeroed
        }
#undef DISPATCH_Z
#undef LAUNCH_Z
    }
    }   // part 1
    if (!(part & 2)) { REMD_CHECK(h, hipGetLastError()); return 0; }
    if (with_energy)
        hipLaunchKernelGGL(pme_energy_reduce_kernel, dim3(h->R), dim3(64), 0, st, s->n_eblk, s->d_energy, h->d_epart,
                           h->n_epart, 6 /*EP_PME*/);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

// test hook: in-place 3-D complex FFT of a host array [nx][ny][nz] (interleaved re, im) on the strided-pass kernels
// bins for the NEXT force evaluation, to be filled by the integrator chain that finalises the positions (integrate.hip)
remd_chain_bins remd_pme_chain_bins(remd_ctx* h)
{
    remd_chain_bins b;
    pme_state* s = (pme_state*)h->pme;
    if (!s || !s->d_cbin_count || s->R != h->R || !h->pme_concurrent || h->no_chain_bins) return b;
    b.nx = s->n[0]; b.cap = s->cbin_cap; b.count = s->d_cbin_count + (size_t)s->cbin_parity * s->R * s->n[0]; b.atoms = s->d_cbin_atoms;
    b.box = h->d_box; b.err = h->d_sync + 2;
    // charges ride in the bins only when they do not depend on the replica's state: the chain runs before the evaluation
    // that refreshes the per-replica lambdas
    b.param = remd_nb_param(h);
    b.q = (b.param && !remd_nb_rep_lam(h)) ? s->d_cbin_q : nullptr;
    return b;
}

int remd_test_fft3d_impl(remd_ctx* h, int nx, int ny, int nz, float* data, int inverse)
{
    pme_state* old = (pme_state*)h->pme;
    const int oldgrid[3] = { h->grid[0], h->grid[1], h->grid[2] };
    const int oldR = h->R;
    h->pme = nullptr; h->grid[0] = nx; h->grid[1] = ny; h->grid[2] = nz; h->R = 1;
    int rc = pme_setup_impl(h, true);
    if (!rc) {
        pme_state* s = (pme_state*)h->pme;
        hipMemcpy(s->d_grid, data, sizeof(float2) * s->npts, hipMemcpyHostToDevice);
        auto all = [&](auto tag) {
            constexpr int SG = decltype(tag)::value;
            launch_pass<SG>(h, s, s->d_grid, s->npts, 2, 1, nx * ny, nx * ny, 0, 1);
            launch_pass<SG>(h, s, s->d_grid, s->npts, 1, nz, nx * nz, nz, (size_t)ny * nz, 0);
            launch_pass<SG>(h, s, s->d_grid, s->npts, 0, ny * nz, ny * nz, ny * nz, 0, 0);
        };
        if (!inverse) all(std::integral_constant<int, -1>()); else all(std::integral_constant<int, +1>());
        hipStreamSynchronize(h->stream);
        hipMemcpy(data, s->d_grid, sizeof(float2) * s->npts, hipMemcpyDeviceToHost);
        if (hipGetLastError() != hipSuccess) rc = remd_fail(h, -2, "fft test launch failed");
    }
    remd_pme_destroy(h);
    h->pme = old; h->grid[0] = oldgrid[0]; h->grid[1] = oldgrid[1]; h->grid[2] = oldgrid[2]; h->R = oldR;
    return rc;
}
