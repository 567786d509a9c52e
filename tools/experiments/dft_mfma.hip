// The XY pass of the PME mesh on the matrix cores (gfx950).
//
// A mesh plane of n x n <= 80 x 80 complex points is transformed as four dense matrix products with the DFT matrix W
// (forward y, forward x, [influence function], inverse y, inverse x) instead of four rounds of 75-point FFTs.  The FFT
// rounds are VALU / LDS-issue bound and share the vector units with the direct-space pair kernel running beside them
// on the second stream; the products run on the otherwise idle MFMA pipes.  f32 accuracy from f16 inputs: every operand
// is split x = hi + lo (two f16, 22 significant bits) and a product is hi*hi + lo*hi + hi*lo with f32 accumulation
// (`v_mfma_f32_16x16x32_f16`); the plane is scaled by a power of two at load (largest element at 2^13) and by the
// a-priori factor 2^-5 after a pass, which keeps every element inside f16's range and the lo halves' subnormal floor
// (2^-25) far below the plane's largest element.  Measured against an f64 FFT the result is closer than an f32 FFT's (tests/test_dft_mfma_gpu.py,
// tools/dft_split_precision.py).
//
// Layout.  An operand is a row-major [80][80] f16 image, 160 B per row (a stride that makes the 16-lane groups of a
// `ds_read_b128` fragment load hit 16 distinct 16-byte slots).  Four images per matrix: re_hi, re_lo, im_hi, im_lo.
// One plane per workgroup of five wavefronts: the plane's images are the only LDS (51 KB => three workgroups per CU by
// LDS), W is the B operand and lives in REGISTERS -- a wavefront owns 16 columns n of the result, i.e. twelve fragments
// of W (48 VGPRs) loaded once per kernel -- and the data streams from LDS as the A operand A[m][k], k contracted.  The
// result tile D[m][n] comes out of the MFMA with four consecutive m per lane, so it is written back as the image [n][m] --
// the transposition every pass needs so that the NEXT pass contracts the other index:
// [x][y] -> (y) -> [ky][x] -> (x) -> [kx][ky] -> (ky) -> [y][kx] -> (kx) -> [x][y].  The five result tiles of a wavefront wait
// in registers until every wavefront has read the image.
// Status (888 planes of 75 x 75, stand-alone): 107 us against the FFT kernel's 67, so the pass is opt-in.  A workgroup lives
// 26 us (7 us loading its plane, 4 x 4.5 us passes) but only ONE is resident per CU: the dispatcher reserves
// ceil(wavefronts / 4) slots on every SIMD for a workgroup, i.e. 2 for these five wavefronts, and at 168 VGPRs a SIMD has 3
// (tools/probes/occupancy_probe.hip; REMD_DFT_TIMES=1 prints the per-phase times and the residency from timestamps taken
// in the kernel).  Two resident workgroups need <= 128 VGPRs (W's lo halves in LDS) or four-wavefront workgroups.
// (The first version kept W in LDS and two planes per workgroup: 154 KB, one workgroup per CU by construction: 89 us.)
// K = 80 is covered by three K = 32 steps; the lanes that would read k >= 80 re-read valid data against a zero W.
#include "remd_internal.h"
#include <vector>
#include <cmath>

typedef _Float16 dm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dm_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 dm_h2 __attribute__((ext_vector_type(2)));
typedef float dm_f4 __attribute__((ext_vector_type(4)));

#define DM_NP 80
#define DM_ROW 160                       // bytes per image row
#define DM_ARR (DM_NP * DM_ROW)          // one f16 image
#define DM_MAT (4 * DM_ARR)              // re_hi, re_lo, im_hi, im_lo
#define DM_THREADS 320                   // five wavefronts: one 16-column strip of the result each
#define DM_LDS (DM_MAT + 128)

__device__ __forceinline__ float dm_wave_max(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// power-of-two scale that puts `mx` into [2^13, 2^14), and its inverse
__device__ __forceinline__ void dm_scale(float mx, float& s, float& inv_s)
{
    int E = (int)((__float_as_uint(mx) >> 23) & 255u);
    E = max(E, 14); E = min(E, 250);
    s = __uint_as_float((unsigned)(267 - E) << 23);
    inv_s = __uint_as_float((unsigned)(E - 13) << 23);
}

struct dm_ctx {
    char* D;                             // the plane's four f16 images in LDS
    int n, lane, wave;
};

// a plane enters a pass with its largest element below 2^14; |sum_k a_k w_k| <= 80 sqrt(2) max|a| < 2^21, and 2^-5 of that is
// still below f16's 65504: the a-priori rescaling after a pass (no reduction over the plane needed)
#define DM_PASS_SCALE (1.f / 32.f)
#define DM_PASS_INV 32.f

// v = hi + lo: hi = v truncated to f16's 11 significant bits (a mask: exactly representable, no conversion back needed
// for the residual), lo = v - hi as an f16 (subnormal below 2^-14, i.e. an absolute error of 2^-25 against a plane whose
// largest element stays above 2^5: tools/dft_split_precision.py models the whole pipeline, 6e-7 of the largest element).
__device__ __forceinline__ void dm_split(float v, _Float16& hi, _Float16& lo)
{
    const float h = __uint_as_float(__float_as_uint(v) & 0xffffe000u);
    hi = (_Float16)h;
    lo = (_Float16)(v - h);
}

// One pass: D[m][n] = sum_k A[m][k] W[k][n] (INV: conj W).  The wavefront owns the 16 columns n of its strip: W's
// fragments for them (`wf`: re_hi, re_lo, im_hi, im_lo x three k steps) live in registers for the whole kernel, the data
// (A operand) streams from LDS tile row by tile row, and the five result tiles are held until every wavefront has read the
// image (barrier) and then written back transposed, [n][m].
// `gs`: factor applied to the accumulators before they are split again (the a-priori bound 2^-7, times 1 / g_bound in the
// influence pass).  EPI: 0 write back, 1 influence function (+ energy) and write back, 2 final: plane to global (x `gs`),
// 3 test hook: forward transform [kx][ky] to global (x `gs`).
template <bool INV, int EPI>
__device__ __forceinline__ void dm_pass(const dm_ctx& c, const dm_h8 (&wf)[4][3], float gs, float e_scale, float2* __restrict__ P,
                                        const float* __restrict__ G, int with_energy, double* s_e)
{
    const int lane = c.lane, kb = lane >> 4, l15 = lane & 15;
    const int col = 16 * c.wave + l15;     // this lane's column n (the new index)
    dm_f4 rr[5], ri[5];
    double e_acc = 0.0;
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        const dm_f4 z = {0.f, 0.f, 0.f, 0.f};
        dm_f4 p1 = z, p2 = z, q1 = z, q2 = z;
        const int arow = (16 * mt + l15) * DM_ROW;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int kbe = (ks == 2) ? (kb & 1) : kb;      // k >= 80: re-read valid data, W's fragment is zero there
            const int off = arow + (ks * 32 + kbe * 8) * 2;
            const dm_h8 arh = *reinterpret_cast<const dm_h8*>(c.D + off);
            const dm_h8 arl = *reinterpret_cast<const dm_h8*>(c.D + DM_ARR + off);
            const dm_h8 aih = *reinterpret_cast<const dm_h8*>(c.D + 2 * DM_ARR + off);
            const dm_h8 ail = *reinterpret_cast<const dm_h8*>(c.D + 3 * DM_ARR + off);
            // hi*hi + lo*hi + hi*lo into one accumulator per real product (a dependent MFMA every fourth issue)
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh, wf[0][ks], p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(aih, wf[2][ks], p2, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh, wf[2][ks], q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(aih, wf[0][ks], q2, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arl, wf[0][ks], p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ail, wf[2][ks], p2, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arl, wf[2][ks], q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ail, wf[0][ks], q2, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh, wf[1][ks], p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(aih, wf[3][ks], p2, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(arh, wf[3][ks], q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(aih, wf[1][ks], q2, 0, 0, 0);
        }
        // W = cos - i sin: forward (a_r + i a_i) W, inverse (a_r + i a_i) conj(W)
        dm_f4 vr = INV ? p1 + p2 : p1 - p2;
        dm_f4 vi = INV ? q2 - q1 : q1 + q2;
        const int m0 = 16 * mt + 4 * kb;   // this lane's four rows m0 .. m0+3 (the index that is NOT contracted)
        if (EPI == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float g = 1.f;
                if (G) g = (col < c.n && m0 + i < c.n) ? G[col * c.n + m0 + i] : 0.f;
                if (with_energy) {
                    const double sr = (double)(vr[i] * e_scale), si = (double)(vi[i] * e_scale);
                    e_acc += (double)g * (sr * sr + si * si);
                }
                vr[i] *= g; vi[i] *= g;
            }
        }
        vr *= gs; vi *= gs;
        if (EPI == 2 || EPI == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (col < c.n && m0 + i < c.n) P[col * c.n + m0 + i] = make_float2(vr[i], vi[i]);
        } else {
            rr[mt] = vr; ri[mt] = vi;
        }
    }
    if (EPI == 2 || EPI == 3) return;
    if (EPI == 1 && with_energy) {
        for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
        if (lane == 0) s_e[c.wave] = e_acc;
    }
    __syncthreads();                       // every wavefront has read the whole image: it may be overwritten
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        const int off = col * DM_ROW + (16 * mt + 4 * kb) * 2;
        dm_h4 rh, rl, ih, il;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            _Float16 h, l;
            dm_split(rr[mt][i], h, l); rh[i] = h; rl[i] = l;
            dm_split(ri[mt][i], h, l); ih[i] = h; il[i] = l;
        }
        *reinterpret_cast<dm_h4*>(c.D + off) = rh;
        *reinterpret_cast<dm_h4*>(c.D + DM_ARR + off) = rl;
        *reinterpret_cast<dm_h4*>(c.D + 2 * DM_ARR + off) = ih;
        *reinterpret_cast<dm_h4*>(c.D + 3 * DM_ARR + off) = il;
    }
    __syncthreads();                       // the transposed image is complete
}

// planes: [nplanes][n][n] float2 (for the mesh: plane = r * nzc + kz of the half spectrum); infl laid out as the planes
// (NULL = 1) with gbound[plane] >= max infl of the plane (NULL = 1).  One plane per workgroup.
// mode 0: forward, influence, inverse (in place); mode 1 (test hook): forward only, output [kx][ky]
__global__ __launch_bounds__(DM_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
void pme_xy_mfma_kernel(int n, int nplanes, int nzc, int nz, float2* __restrict__ spec, const uint4* __restrict__ wtab,
                        const float* __restrict__ infl, const float* __restrict__ gbound, int with_energy,
                        double* __restrict__ energy, int n_eblk, int mode, long long* tdbg)
{
    __builtin_amdgcn_s_setprio(3);
#define TSTAMP(k) do { if (tdbg && threadIdx.x == 0) tdbg[(size_t)blockIdx.x * 12 + (k)] = (long long)wall_clock64(); } while (0)
    TSTAMP(0);
    if (tdbg && threadIdx.x == 0) { tdbg[(size_t)blockIdx.x * 12 + 8] = __builtin_amdgcn_s_getreg(63492); tdbg[(size_t)blockIdx.x * 12 + 9] = __builtin_amdgcn_s_getreg(63508); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    dm_ctx c;
    c.n = n; c.lane = tid & 63; c.wave = tid >> 6; c.D = smem;
    float* s_max = reinterpret_cast<float*>(smem + DM_MAT);
    double* s_e = reinterpret_cast<double*>(smem + DM_MAT + 64);
    const int plane = blockIdx.x;
    float2* P = spec + (size_t)plane * n * n;
    const float* G = infl ? infl + (size_t)plane * n * n : nullptr;
    TSTAMP(1);
    // load: two neighbouring y per thread, ten pairs per thread; rows / columns >= n are the zero padding
    float2 va[10], vb[10];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int idx = tid + 320 * j, x = idx / 40, y = 2 * (idx - 40 * x);
        va[j] = (x < n && y < n) ? P[x * n + y] : make_float2(0.f, 0.f);
        vb[j] = (x < n && y + 1 < n) ? P[x * n + y + 1] : make_float2(0.f, 0.f);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[j].x), fabsf(va[j].y)), fmaxf(fabsf(vb[j].x), fabsf(vb[j].y))));
    }
    mx = dm_wave_max(mx);
    if (c.lane == 0) s_max[c.wave] = mx;
    __syncthreads();
    TSTAMP(2);
    mx = 0.f;
    for (int w = 0; w < 5; ++w) mx = fmaxf(mx, s_max[w]);
    float s, inv_s;
    dm_scale(mx, s, inv_s);               // the plane's largest element at 2^13; every later pass is bounded a priori
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int idx = tid + 320 * j, x = idx / 40, y = 2 * (idx - 40 * x);
        const int off = x * DM_ROW + y * 2;
        dm_h2 rh, rl, ih, il;
        _Float16 h, l;
        dm_split(va[j].x * s, h, l); rh[0] = h; rl[0] = l;
        dm_split(vb[j].x * s, h, l); rh[1] = h; rl[1] = l;
        dm_split(va[j].y * s, h, l); ih[0] = h; il[0] = l;
        dm_split(vb[j].y * s, h, l); ih[1] = h; il[1] = l;
        *reinterpret_cast<dm_h2*>(c.D + off) = rh;
        *reinterpret_cast<dm_h2*>(c.D + DM_ARR + off) = rl;
        *reinterpret_cast<dm_h2*>(c.D + 2 * DM_ARR + off) = ih;
        *reinterpret_cast<dm_h2*>(c.D + 3 * DM_ARR + off) = il;
    }
    __syncthreads();
    TSTAMP(3);
    // W's fragments of this wavefront's strip, from the global image [4][80][80] f16 (L2 resident: 51 KB shared by all)
    dm_h8 wf[4][3];
    {
        // the table is stored in fragment order [strip][image][k step][lane] x 8 f16: one coalesced 1 KB load per fragment
        const dm_h8* wt = reinterpret_cast<const dm_h8*>(wtab) + (size_t)c.wave * 12 * 64 + c.lane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wf[q][ks] = wt[(q * 3 + ks) * 64];
    }
    // scaled image d0 = s true; d1 = acc1 / 32; d2 = acc2 / 32 x g / gb; d3 = acc3 / 32; out = acc4 x 32^3 gb / s
    if (mode == 1) {
        dm_pass<false, 0>(c, wf, DM_PASS_SCALE, 0.f, P, G, 0, s_e);                // along y: [x][y]   -> [ky][x]
        dm_pass<false, 3>(c, wf, inv_s * DM_PASS_INV, 0.f, P, G, 0, s_e);
        return;
    }
    float gb = 1.f;
    if (gbound) {                           // power of two >= the plane's largest influence value
        const float gm = fmaxf(gbound[plane], 1e-30f);
        gb = __uint_as_float((__float_as_uint(gm) + 0x007fffffu) & 0x7f800000u);
    }
    const int kz = plane % nzc;
    const int we = (with_energy && energy) ? 1 : 0;
    dm_pass<false, 0>(c, wf, DM_PASS_SCALE, 0.f, P, G, 0, s_e);                    // along y:  [x][y]   -> [ky][x]
    TSTAMP(4);
    dm_pass<false, 1>(c, wf, DM_PASS_SCALE / gb, inv_s * DM_PASS_INV, P, G, we, s_e);    // along x:  [ky][x]  -> [kx][ky], x G
    TSTAMP(5);
    if (we && tid == 0) {
        const double wz = (kz == 0 || 2 * kz == nz) ? 1.0 : 2.0;      // Hermitian half: weight of the mirrored plane
        double tot = 0.0;
        for (int w = 0; w < 5; ++w) tot += s_e[w];
        energy[(size_t)(plane / nzc) * n_eblk + kz] = 0.5 * wz * tot;
    }
    dm_pass<true, 0>(c, wf, DM_PASS_SCALE, 0.f, P, G, 0, s_e);                     // along ky: [kx][ky] -> [y][kx]
    TSTAMP(6);
    dm_pass<true, 2>(c, wf, inv_s * (DM_PASS_INV * DM_PASS_INV * DM_PASS_INV) * gb, 0.f, P, G, 0, s_e);   // along kx: [y][kx] -> [x][y], to global
    TSTAMP(7);
}

// the LDS image of W_n = exp(-2 pi i a k / n), zero padded to 80 x 80: re_hi, re_lo, im_hi, im_lo
int remd_dftmm_build_table(remd_ctx* h, int n, void** d_table)
{
    if (n > DM_NP) return remd_fail(h, -3, "dft_mfma: plane edge > 80");
    // W_n = exp(-2 pi i a k / n), zero padded, split into f16 hi + lo, in the register layout of the B operand:
    // [strip = column tile][image: re_hi, re_lo, im_hi, im_lo][k step][lane] x 8 consecutive k; lane = (kb = lane / 16, column
    // = lane % 16) holds k = 32 step + 8 kb .. + 7
    std::vector<_Float16> img((size_t)5 * 4 * 3 * 64 * 8, (_Float16)0.f);
    for (int strip = 0; strip < 5; ++strip)
        for (int ks = 0; ks < 3; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int col = 16 * strip + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
                    if (col >= n || k >= n) continue;
                    const double th = 2.0 * M_PI * (double)(((long long)col * k) % n) / n;
                    const float w[2] = { (float)cos(th), (float)(-sin(th)) };
                    for (int q = 0; q < 2; ++q) {
                        const _Float16 hi = (_Float16)w[q];
                        const _Float16 lo = (_Float16)(w[q] - (float)hi);
                        img[((((size_t)strip * 4 + 2 * q) * 3 + ks) * 64 + lane) * 8 + j] = hi;
                        img[((((size_t)strip * 4 + 2 * q + 1) * 3 + ks) * 64 + lane) * 8 + j] = lo;
                    }
                }
    REMD_CHECK(h, hipMalloc(d_table, img.size() * sizeof(_Float16)));
    REMD_CHECK(h, hipMemcpy(*d_table, img.data(), img.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS));
    return 0;
}

void remd_dftmm_launch_xy(hipStream_t st, int n, int nplanes, int nzc, int nz, float2* spec, const void* table, const float* infl,
                          const float* gbound, int with_energy, double* energy, int n_eblk, int mode, long long* tdbg = nullptr);
void remd_dftmm_launch_xy(hipStream_t st, int n, int nplanes, int nzc, int nz, float2* spec, const void* table, const float* infl,
                          const float* gbound, int with_energy, double* energy, int n_eblk, int mode, long long* tdbg)
{
    hipLaunchKernelGGL(pme_xy_mfma_kernel, dim3(nplanes), dim3(DM_THREADS), DM_LDS, st, n, nplanes, nzc, nz, spec,
                       reinterpret_cast<const uint4*>(table), infl, gbound, with_energy, energy, n_eblk, mode, tdbg);
}

// test hook: `nplanes` planes of n x n complex numbers, transformed in place (mode 1: forward 2-D DFT, output [kx][ky];
// mode 0: forward, inverse = n^2 x the input)
int remd_test_xy_mfma_impl(remd_ctx* h, int n, int nplanes, float* data, int mode)
{
    void* d_tab = nullptr; float2* d = nullptr;
    int rc = remd_dftmm_build_table(h, n, &d_tab);
    if (rc) return rc;
    const size_t bytes = sizeof(float2) * (size_t)n * n * nplanes;
    REMD_CHECK(h, hipMalloc(&d, bytes));
    REMD_CHECK(h, hipMemcpy(d, data, bytes, hipMemcpyHostToDevice));
    long long* d_t = nullptr;
    if (getenv("REMD_DFT_TIMES")) { hipMalloc(&d_t, sizeof(long long) * 12 * nplanes); hipMemset(d_t, 0, sizeof(long long) * 12 * nplanes); }
    remd_dftmm_launch_xy(h->stream, n, nplanes, nplanes, 2 * nplanes, d, d_tab, nullptr, nullptr, 0, nullptr, 0, mode, d_t);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    if (d_t) {
        std::vector<long long> t((size_t)12 * nplanes);
        hipMemcpy(t.data(), d_t, sizeof(long long) * t.size(), hipMemcpyDeviceToHost);
        long long t0 = t[0], t1 = 0;
        for (int p = 0; p < nplanes; ++p) { t0 = std::min(t0, t[(size_t)p * 12]); t1 = std::max(t1, t[(size_t)p * 12 + 7]); }
        double ph[8] = {0}; double life = 0; int late = 0;
        for (int p = 0; p < nplanes; ++p) {
            for (int k = 1; k < 8; ++k) ph[k] += (double)(t[(size_t)p * 12 + k] - t[(size_t)p * 12 + k - 1]);
            life += (double)(t[(size_t)p * 12 + 7] - t[(size_t)p * 12]);
            if (t[(size_t)p * 12] - t0 > 500) ++late;
        }
        { int hs[16] = {0}, he[16] = {0};
          for (int p = 0; p < nplanes; ++p) { hs[std::min(15, (int)((t[(size_t)p * 12] - t0) / 1000))]++; he[std::min(15, (int)((t[(size_t)p * 12 + 7] - t0) / 1000))]++; }
          fprintf(stderr, "[dft] starts per 10 us:"); for (int b = 0; b < 16; ++b) fprintf(stderr, " %d", hs[b]);
          fprintf(stderr, "\n[dft] ends   per 10 us:"); for (int b = 0; b < 16; ++b) fprintf(stderr, " %d", he[b]); fprintf(stderr, "\n"); }
        { std::map<long long, int> cnt; int first_round = 0;
          for (int p = 0; p < nplanes; ++p) if (t[(size_t)p * 12] - t0 < 500) { const long long hw = t[(size_t)p * 12 + 8], xcc = t[(size_t)p * 12 + 9] & 15;
              const long long cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; cnt[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; ++first_round; }
          int mxc = 0; for (auto& kv : cnt) mxc = std::max(mxc, kv.second);
          fprintf(stderr, "[dft] first round: %d workgroups on %d distinct (xcc, se, sh, cu), at most %d on one; sample hw_id %llx xcc %llx\n", first_round, (int)cnt.size(), mxc,
                  (unsigned long long)t[8], (unsigned long long)t[9]); }
        fprintf(stderr, "[dft] span %.1f us; per workgroup (10 ns ticks -> us): life %.1f; W %.1f load %.1f image %.1f pass1 %.1f pass2 %.1f pass3 %.1f pass4 %.1f; %d of %d workgroups started > 5 us after the first\n",
                (t1 - t0) * 0.01, life / nplanes * 0.01, ph[1] / nplanes * 0.01, ph[2] / nplanes * 0.01, ph[3] / nplanes * 0.01, ph[4] / nplanes * 0.01,
                ph[5] / nplanes * 0.01, ph[6] / nplanes * 0.01, ph[7] / nplanes * 0.01, late, nplanes);
        hipFree(d_t);
    }
    REMD_CHECK(h, hipGetLastError());
    REMD_CHECK(h, hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost));
    hipFree(d); hipFree(d_tab);
    return 0;
}
