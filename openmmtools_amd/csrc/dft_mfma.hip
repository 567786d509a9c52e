// The XY pass of the PME mesh on the matrix cores (gfx950).
//
// A mesh plane of n x n <= 80 x 80 complex points is transformed as four dense matrix products with the DFT matrix W
// (forward y, forward x, [influence function], inverse y, inverse x) instead of four rounds of 75-point FFTs.  The FFT
// rounds are VALU / LDS-issue bound and share the vector units with the direct-space pair kernel running beside them
// on the second stream; the products run on the otherwise idle MFMA pipes.  f32 accuracy from f16 inputs: every operand
// is split x = hi + lo (two f16, 22 significant bits) and a product is hi*hi + lo*hi + hi*lo with f32 accumulation
// (`v_mfma_f32_16x16x32_f16`); the lo halves are stored x 2^11 and accumulated apart, so they stay normal numbers;
// the plane is scaled by a power of two at load (largest element at 2^13) and by the a-priori bound 2^-7 after a pass.  Measured against an f64 FFT the result is closer than an f32 FFT's (tests/test_dft_mfma_gpu.py,
// tools/dft_split_precision.py).
//
// Layout.  An operand is a row-major [80][80] f16 image, 160 B per row (a stride that makes the 16-lane groups of a
// `ds_read_b128` fragment load hit 16 distinct 16-byte slots).  Four images per matrix: re_hi, re_lo, im_hi, im_lo.
// W (symmetric) sits in LDS once per workgroup (51 KB) and serves as the B operand; a workgroup owns TWO planes
// (2 x 51 KB), five wavefronts each, one 16-row strip of the plane per wavefront.  Data is always the A operand
// A[m][k], k contracted: the result tile D[m][n] comes out of the MFMA with four consecutive m per lane, so it is
// written back as the image [n][m] -- the transposition every pass needs so that the NEXT pass contracts the other
// index.  [x][y] -> (y) -> [ky][x] -> (x) -> [kx][ky] -> (ky) -> [y][kx] -> (kx) -> [x][y].
// K = 80 is covered by three K = 32 steps; the lanes that would read k >= 80 re-read valid data and get a zero A.
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
#define DM_THREADS 640
#define DM_LDS (3 * DM_MAT + 256)

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
    char* W; char* D;
    int n, lane, wave, half, mt;
};

#define DM_LO_SCALE 2048.f               // lo halves are stored x 2^11 (never subnormal before hi is)
#define DM_LO_INV (1.f / 2048.f)
#define DM_PASS_SCALE (1.f / 128.f)      // |sum_k a_k w_k| <= 75 sqrt(2) max|a| < 2^7 max|a|

__device__ __forceinline__ void dm_split(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * DM_LO_SCALE);
}

// One pass: D[m][n] = sum_k A[m][k] W[k][n] (INV: conj W) on this wavefront's 16-row strip, written back transposed.
// `gs`: factor applied to the accumulators before they are split again (the a-priori bound 2^-7, times g / g_bound in
// the influence pass).  EPI: 0 write back, 1 influence function (+ energy) and write back, 2 final: plane to global
// (x `gs`), 3 test hook: forward transform [kx][ky] to global (x `gs`).
template <bool INV, int EPI>
__device__ __forceinline__ void dm_pass(const dm_ctx& c, float gs, float e_scale, float2* __restrict__ P, const float* __restrict__ G,
                                        bool active, int with_energy, double* s_e)
{
    const int lane = c.lane, kb = lane >> 4, l15 = lane & 15;
    dm_h8 a[4][3];                         // the strip's A fragments: re_hi, re_lo, im_hi, im_lo x three k steps
    {
        const int arow = (16 * c.mt + l15) * DM_ROW;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int kbe = (ks == 2) ? (kb & 1) : kb;
            const int off = arow + (ks * 32 + kbe * 8) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q][ks] = *reinterpret_cast<const dm_h8*>(c.D + q * DM_ARR + off);
                if (ks == 2 && kb >= 2) a[q][ks] = (dm_h8)(_Float16)0;
            }
        }
    }
    __syncthreads();                       // every wavefront holds its strip: the plane image may be overwritten
    const int m0 = 16 * c.mt + 4 * kb;     // this lane's four rows m0 .. m0+3 (the index that is NOT contracted)
    double e_acc = 0.0;
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const dm_f4 z = {0.f, 0.f, 0.f, 0.f};
        dm_f4 p1 = z, p2 = z, q1 = z, q2 = z, p1x = z, p2x = z, q1x = z, q2x = z;
        const int nrow = 16 * nt + l15;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int kbe = (ks == 2) ? (kb & 1) : kb;
            const int off = nrow * DM_ROW + (ks * 32 + kbe * 8) * 2;
            const dm_h8 wrh = *reinterpret_cast<const dm_h8*>(c.W + off);
            const dm_h8 wrl = *reinterpret_cast<const dm_h8*>(c.W + DM_ARR + off);
            const dm_h8 wih = *reinterpret_cast<const dm_h8*>(c.W + 2 * DM_ARR + off);
            const dm_h8 wil = *reinterpret_cast<const dm_h8*>(c.W + 3 * DM_ARR + off);
            p1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ks], wrh, p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2][ks], wih, p2, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ks], wih, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2][ks], wrh, q2, 0, 0, 0);
            p1x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][ks], wrh, p1x, 0, 0, 0);
            p2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3][ks], wih, p2x, 0, 0, 0);
            q1x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][ks], wih, q1x, 0, 0, 0);
            q2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3][ks], wrh, q2x, 0, 0, 0);
            p1x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ks], wrl, p1x, 0, 0, 0);
            p2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2][ks], wil, p2x, 0, 0, 0);
            q1x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ks], wil, q1x, 0, 0, 0);
            q2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2][ks], wrl, q2x, 0, 0, 0);
        }
        // hi*hi + 2^-11 (lo*hi + hi*lo); W = cos - i sin: forward (a_r + i a_i) W, inverse (a_r + i a_i) conj(W)
        p1 += p1x * DM_LO_INV; p2 += p2x * DM_LO_INV; q1 += q1x * DM_LO_INV; q2 += q2x * DM_LO_INV;
        dm_f4 vr = INV ? p1 + p2 : p1 - p2;
        dm_f4 vi = INV ? q2 - q1 : q1 + q2;
        const int col = 16 * nt + l15;     // column n: the new index
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
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col < c.n && m0 + i < c.n) P[col * c.n + m0 + i] = make_float2(vr[i], vi[i]);
            }
        } else {
            const int off = col * DM_ROW + m0 * 2;
            dm_h4 rh, rl, ih, il;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 h, l;
                dm_split(vr[i], h, l); rh[i] = h; rl[i] = l;
                dm_split(vi[i], h, l); ih[i] = h; il[i] = l;
            }
            *reinterpret_cast<dm_h4*>(c.D + off) = rh;
            *reinterpret_cast<dm_h4*>(c.D + DM_ARR + off) = rl;
            *reinterpret_cast<dm_h4*>(c.D + 2 * DM_ARR + off) = ih;
            *reinterpret_cast<dm_h4*>(c.D + 3 * DM_ARR + off) = il;
        }
    }
    if (EPI == 1 && with_energy) {
        for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
        if (lane == 0) s_e[c.wave] = e_acc;
    }
    if (EPI == 0 || EPI == 1) __syncthreads();     // the transposed image is complete
}

// planes: [nplanes][n][n] float2 (for the mesh: plane = r * nzc + kz of the half spectrum); infl laid out as the planes
// (NULL = 1) with gbound[plane] >= max infl of the plane (NULL = 1)
// mode 0: forward, influence, inverse (in place); mode 1 (test hook): forward only, output [kx][ky]
__global__ __launch_bounds__(DM_THREADS)
void pme_xy_mfma_kernel(int n, int nplanes, int nzc, int nz, float2* __restrict__ spec, const uint4* __restrict__ wtab,
                        const float* __restrict__ infl, const float* __restrict__ gbound, int with_energy,
                        double* __restrict__ energy, int n_eblk, int mode)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    dm_ctx c;
    c.n = n; c.lane = tid & 63; c.wave = tid >> 6; c.half = c.wave / 5; c.mt = c.wave - 5 * c.half;
    c.W = smem; c.D = smem + DM_MAT + c.half * DM_MAT;
    float* s_max = reinterpret_cast<float*>(smem + 3 * DM_MAT);
    double* s_e = reinterpret_cast<double*>(smem + 3 * DM_MAT + 64);
    int plane = blockIdx.x * 2 + c.half;
    const bool active = plane < nplanes;
    plane = min(plane, nplanes - 1);
    for (int i = tid; i < DM_MAT / 16; i += DM_THREADS) reinterpret_cast<uint4*>(c.W)[i] = wtab[i];
    float2* P = spec + (size_t)plane * n * n;
    const float* G = infl ? infl + (size_t)plane * n * n : nullptr;
    // load: two neighbouring y per thread, ten pairs per thread; rows / columns >= n are the zero padding
    const int t320 = tid - c.half * 320;
    float2 va[10], vb[10];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int idx = t320 + 320 * j, x = idx / 40, y = 2 * (idx - 40 * x);
        va[j] = (x < n && y < n) ? P[x * n + y] : make_float2(0.f, 0.f);
        vb[j] = (x < n && y + 1 < n) ? P[x * n + y + 1] : make_float2(0.f, 0.f);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[j].x), fabsf(va[j].y)), fmaxf(fabsf(vb[j].x), fabsf(vb[j].y))));
    }
    mx = dm_wave_max(mx);
    if (c.lane == 0) s_max[c.wave] = mx;
    __syncthreads();
    mx = 0.f;
    for (int w = 0; w < 5; ++w) mx = fmaxf(mx, s_max[c.half * 5 + w]);
    float s, inv_s;
    dm_scale(mx, s, inv_s);               // the plane's largest element at 2^13; every later pass is bounded a priori
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int idx = t320 + 320 * j, x = idx / 40, y = 2 * (idx - 40 * x);
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
    // scaled image d0 = s true; d1 = acc1 / 128; d2 = acc2 / 128 x g / gb; d3 = acc3 / 128; out = acc4 x 128^3 gb / s
    if (mode == 1) {
        dm_pass<false, 0>(c, DM_PASS_SCALE, 0.f, P, G, active, 0, s_e);               // along y: [x][y]   -> [ky][x]
        dm_pass<false, 3>(c, inv_s * 128.f, 0.f, P, G, active, 0, s_e);
        return;
    }
    float gb = 1.f;
    if (gbound) {                           // power of two >= the plane's largest influence value
        const float gm = fmaxf(gbound[plane], 1e-30f);
        gb = __uint_as_float((__float_as_uint(gm) + 0x007fffffu) & 0x7f800000u);
    }
    const int kz = plane % nzc;
    dm_pass<false, 0>(c, DM_PASS_SCALE, 0.f, P, G, active, 0, s_e);                   // along y:  [x][y]   -> [ky][x]
    dm_pass<false, 1>(c, DM_PASS_SCALE / gb, inv_s * 128.f, P, G, active, (with_energy && energy) ? 1 : 0, s_e);   // along x -> [kx][ky], x G
    if (with_energy && energy && c.mt == 0 && c.lane == 0 && active) {
        const double wz = (kz == 0 || 2 * kz == nz) ? 1.0 : 2.0;      // Hermitian half: weight of the mirrored plane
        double tot = 0.0;
        for (int w = 0; w < 5; ++w) tot += s_e[c.half * 5 + w];
        energy[(size_t)(plane / nzc) * n_eblk + kz] = 0.5 * wz * tot;
    }
    dm_pass<true, 0>(c, DM_PASS_SCALE, 0.f, P, G, active, 0, s_e);                    // along ky: [kx][ky] -> [y][kx]
    dm_pass<true, 2>(c, inv_s * (128.f * 128.f * 128.f) * gb, 0.f, P, G, active, 0, s_e);   // along kx: [y][kx] -> [x][y], to global
}

// the LDS image of W_n = exp(-2 pi i a k / n), zero padded to 80 x 80: re_hi, re_lo, im_hi, im_lo
int remd_dftmm_build_table(remd_ctx* h, int n, void** d_table)
{
    if (n > DM_NP) return remd_fail(h, -3, "dft_mfma: plane edge > 80");
    std::vector<_Float16> img((size_t)4 * DM_NP * DM_NP, (_Float16)0.f);
    for (int a = 0; a < n; ++a)
        for (int k = 0; k < n; ++k) {
            const double th = 2.0 * M_PI * (double)(((long long)a * k) % n) / n;
            const float w[2] = { (float)cos(th), (float)(-sin(th)) };
            for (int q = 0; q < 2; ++q) {
                const _Float16 hi = (_Float16)w[q];
                const _Float16 lo = (_Float16)((w[q] - (float)hi) * 2048.f);        // DM_LO_SCALE
                img[(size_t)(2 * q) * DM_NP * DM_NP + a * DM_NP + k] = hi;
                img[(size_t)(2 * q + 1) * DM_NP * DM_NP + a * DM_NP + k] = lo;
            }
        }
    REMD_CHECK(h, hipMalloc(d_table, DM_MAT));
    REMD_CHECK(h, hipMemcpy(*d_table, img.data(), DM_MAT, hipMemcpyHostToDevice));
    REMD_CHECK(h, hipFuncSetAttribute((const void*)pme_xy_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS));
    return 0;
}

void remd_dftmm_launch_xy(hipStream_t st, int n, int nplanes, int nzc, int nz, float2* spec, const void* table, const float* infl,
                          const float* gbound, int with_energy, double* energy, int n_eblk, int mode)
{
    hipLaunchKernelGGL(pme_xy_mfma_kernel, dim3((nplanes + 1) / 2), dim3(DM_THREADS), DM_LDS, st, n, nplanes, nzc, nz, spec,
                       reinterpret_cast<const uint4*>(table), infl, gbound, with_energy, energy, n_eblk, mode);
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
    remd_dftmm_launch_xy(h->stream, n, nplanes, nplanes, 2 * nplanes, d, d_tab, nullptr, nullptr, 0, nullptr, 0, mode);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    REMD_CHECK(h, hipGetLastError());
    REMD_CHECK(h, hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost));
    hipFree(d); hipFree(d_tab);
    return 0;
}
