// Achievable-roof microbenchmarks ("measure the achievable roofs with a STREAM-triad and an FMA
// microbenchmark on the actual box and use those as denominators").  bench.py reports them next to the spec peaks of
// MI355X_MICROARCH.md; they do not touch any engine state.
#include "remd_internal.h"

typedef float float2v __attribute__((ext_vector_type(2)));

// STREAM triad c = a + s b on float4: 2 reads + 1 write of 16 bytes per element
__global__ __launch_bounds__(256)
void roof_triad_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, float s, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

// 16 independent v_fma_f32 chains per lane (enough to cover the dependent-issue latency at 4 waves per SIMD)
__global__ __launch_bounds__(256)
void roof_fma_kernel(float* __restrict__ out, int iters, float seed, unsigned long long* __restrict__ clk = nullptr)
{
    // clk: shader cycles (s_memtime) and 100 MHz wall-clock ticks spent by wavefront 0 inside the FMA loop: the clock the chip
    // sustains under this load (the 157.3 TFLOP/s vector peak is quoted at 2.4 GHz)
    const bool stamp = clk && blockIdx.x == 0 && threadIdx.x == 0;
    const unsigned long long c0 = stamp ? clock64() : 0ull, w0 = stamp ? wall_clock64() : 0ull;
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = seed + 0.001f * (float)(threadIdx.x + k);
    const float m = 0.999999f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = __builtin_fmaf(a[k], m, c);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the chains alive
    if (stamp) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

// the same with v_pk_fma_f32 (two f32 FMAs per lane per instruction): the form the 157.3 TFLOP/s vector peak counts
__global__ __launch_bounds__(256)
void roof_pkfma_kernel(float* __restrict__ out, int iters, float seed)
{
    float2v a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k].x = seed + 0.001f * (float)(threadIdx.x + k); a[k].y = seed - 0.001f * (float)(threadIdx.x + k); }
    const float2v m = {0.999999f, 0.999998f}, c = {1e-7f, 2e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = __builtin_elementwise_fma(a[k], m, c);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k].x + a[k].y;
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int remd_roof_microbench(remd_handle h, double* stream_gb_per_s, double* fma_tflop_per_s, double* pk_fma_tflop_per_s)
{
    if (!h) return -1;
    hipSetDevice(h->device);
    hipEvent_t e0, e1;
    REMD_CHECK(h, hipEventCreate(&e0)); REMD_CHECK(h, hipEventCreate(&e1));
    float ms = 0.f;
    if (stream_gb_per_s) {
        const size_t n4 = (size_t)1 << 25;                          // 3 arrays x 512 MiB: past the 256 MiB Infinity Cache
        float4 *a = nullptr, *b = nullptr, *c = nullptr;
        REMD_CHECK(h, hipMalloc(&a, n4 * sizeof(float4))); REMD_CHECK(h, hipMalloc(&b, n4 * sizeof(float4))); REMD_CHECK(h, hipMalloc(&c, n4 * sizeof(float4)));
        REMD_CHECK(h, hipMemsetAsync(a, 0, n4 * sizeof(float4), h->stream)); REMD_CHECK(h, hipMemsetAsync(b, 0, n4 * sizeof(float4), h->stream));
        const int reps = 10;
        hipLaunchKernelGGL(roof_triad_kernel, dim3(256 * 16), dim3(256), 0, h->stream, a, b, c, 1.5f, n4);     // warm-up
        hipEventRecord(e0, h->stream);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(roof_triad_kernel, dim3(256 * 16), dim3(256), 0, h->stream, a, b, c, 1.5f, n4);
        hipEventRecord(e1, h->stream);
        REMD_CHECK(h, hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        *stream_gb_per_s = 3.0 * (double)n4 * sizeof(float4) * reps / (ms * 1e-3) / 1e9;
        hipFree(a); hipFree(b); hipFree(c);
    }
    float* out = nullptr;
    const int blocks = 256 * 8, iters = 8192;
    REMD_CHECK(h, hipMalloc(&out, sizeof(float) * blocks * 256));
    if (fma_tflop_per_s) {
        hipLaunchKernelGGL(roof_fma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, 64, 1.0f, (unsigned long long*)nullptr);
        hipEventRecord(e0, h->stream);
        hipLaunchKernelGGL(roof_fma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, iters, 1.0f, (unsigned long long*)nullptr);
        hipEventRecord(e1, h->stream);
        REMD_CHECK(h, hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        *fma_tflop_per_s = 2.0 * 16.0 * (double)iters * blocks * 256 / (ms * 1e-3) / 1e12;
    }
    if (pk_fma_tflop_per_s) {
        hipLaunchKernelGGL(roof_pkfma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, 64, 1.0f);
        hipEventRecord(e0, h->stream);
        hipLaunchKernelGGL(roof_pkfma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, iters, 1.0f);
        hipEventRecord(e1, h->stream);
        REMD_CHECK(h, hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        *pk_fma_tflop_per_s = 4.0 * 16.0 * (double)iters * blocks * 256 / (ms * 1e-3) / 1e12;
    }
    hipFree(out);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

// The shader clock under sustained v_fma_f32 load (one full-occupancy launch of roof_fma_kernel, cycles / wall time of one
// wavefront): explains measured_roofs.fma_f32 against the 157.3 TFLOP/s spec, which assumes 2.4 GHz.
extern "C" int remd_roof_clock_ghz(remd_handle h, double* ghz_under_fma_load)
{
    if (!h || !ghz_under_fma_load) return -1;
    hipSetDevice(h->device);
    float* out = nullptr; unsigned long long* clk = nullptr;
    const int blocks = 256 * 8, iters = 8192;
    REMD_CHECK(h, hipMalloc(&out, sizeof(float) * blocks * 256));
    REMD_CHECK(h, hipMalloc(&clk, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(roof_fma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, 64, 1.0f, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(roof_fma_kernel, dim3(blocks), dim3(256), 0, h->stream, out, iters, 1.0f, clk);
    unsigned long long hc[2] = {0, 0};
    REMD_CHECK(h, hipMemcpyAsync(hc, clk, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    hipFree(out); hipFree(clk);
    *ghz_under_fma_load = hc[1] > 0 ? (double)hc[0] / ((double)hc[1] * 10.0) : 0.0;     // cycles / (ticks * 10 ns) in GHz
    return 0;
}


// ---- issue floor of the pair kernel's cluster-pair step -------------------------------------------------------
// A replay of the Coulomb-only cluster-pair step of nonbonded_sci_body (forces.hip, the EARLY table path) with every operand in
// registers and no global memory: integer separation (3 sub, 3 cvt), scale and r^2 (2 packed multiplies, multiply, add, fma), table
// key and LDS address (max, bit-field extract, shift-add), one 16-byte LDS read of the cubic's coefficients (a 4 KB table, addresses
// spread over its bins as in the kernel), mantissa remainder (and, cvt), cubic and charge product (3 fma, 2 mul), cutoff compare and
// select, the two packed and two scalar accumulator updates -- 27 VALU instructions + 1 LDS read per step, the compare result taken
// through an SGPR pair as the kernel's ballot is.  `chains` independent steps are interleaved per loop iteration (1 = the kernel as it
// is: every step waits for its own table read; 2 = two steps in flight).  Timed at a chosen number of wavefronts per SIMD (the grid
// fills the chip exactly once; residency is capped with dynamic LDS): cycles per step per SIMD = what the step costs when nothing but
// issue and the LDS round trip is in the way.
typedef float roof_v2f __attribute__((ext_vector_type(2)));
typedef float roof_v4f __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256)
void roof_pair_step_kernel(float* __restrict__ out, int iters, unsigned int seed, float sL, float umin, float rcut2)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_tab[];
    for (int k = threadIdx.x; k < 256; k += 256) s_tab[k] = make_float4(1.0f + 1e-3f * k, -2e-7f, 3e-14f, -1e-21f);
    __syncthreads();
    typedef __attribute__((address_space(3))) const roof_v4f lds_v4f;
    const unsigned int lds_base = (unsigned int)(__UINTPTR_TYPE__)(lds_v4f*)s_tab;
    unsigned int xj[3] = { seed * 2654435761u + threadIdx.x * 40503u, seed * 40503u + threadIdx.x * 2246822519u, seed + threadIdx.x * 3266489917u };
    unsigned int xi[CHAINS][3];
    roof_v2f fxy[CHAINS]; float fz[CHAINS], qi[CHAINS];
    roof_v2f fjxy = { 0.f, 0.f }; float fjz = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        xi[c][0] = xj[0] + 0x01000000u * (c + 1); xi[c][1] = xj[1] - 0x00800000u * (c + 1); xi[c][2] = xj[2] + 0x00c00000u * (c + 1);
        fxy[c] = roof_v2f{ 0.f, 0.f }; fz[c] = 0.f; qi[c] = 0.5f + 0.1f * c;
    }
    const float qj = 0.7f;
    for (int it = 0; it < iters; ++it) {
        float dx[CHAINS], dy[CHAINS], dz[CHAINS], r2c[CHAINS], ttf[CHAINS];
        unsigned int addr[CHAINS];
        unsigned long long in[CHAINS];
        roof_v4f c4[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            dx[c] = (float)(int)(xj[0] - xi[c][0]) * sL; dy[c] = (float)(int)(xj[1] - xi[c][1]) * sL; dz[c] = (float)(int)(xj[2] - xi[c][2]) * sL;
            const float r2 = dx[c] * dx[c] + dy[c] * dy[c] + dz[c] * dz[c];
            in[c] = __builtin_amdgcn_ballot_w64(r2 < rcut2);
            // (the kernel's key is bits [18, 32) of r^2; here 8 bits of it index the 4 KB stand-in table)
            asm("v_max_f32_e32 %0, %2, %3\n\tv_bfe_u32 %1, %0, 15, 8\n\tv_lshl_add_u32 %1, %1, 4, %4"
                : "=&v"(r2c[c]), "=&v"(addr[c]) : "s"(umin), "v"(r2), "s"(lds_base));
            c4[c] = *(lds_v4f*)(__UINTPTR_TYPE__)addr[c];
            ttf[c] = (float)(__float_as_uint(r2c[c]) & 0x3ffffu);
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            float fr = (qi[c] * qj) * fmaf(ttf[c], fmaf(ttf[c], fmaf(ttf[c], c4[c].w, c4[c].z), c4[c].y), c4[c].x);
            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(fr) : "v"(fr), "s"(in[c]));
            const roof_v2f dxy = { dx[c], dy[c] }, fr2 = { fr, fr };
            fxy[c] = __builtin_elementwise_fma(dxy, fr2, fxy[c]);
            fjxy = __builtin_elementwise_fma(-dxy, fr2, fjxy);
            fz[c] = fmaf(dz[c], fr, fz[c]);
            fjz = fmaf(-dz[c], fr, fjz);
        }
        xj[0] += 0x9e3779b9u; xj[1] += 0x7f4a7c15u; xj[2] += 0x85ebca6bu;       // the next j cluster
    }
    float sacc = fjxy.x + fjxy.y + fjz;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) sacc += fxy[c].x + fxy[c].y + fz[c];
    if (sacc == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = sacc;      // never true: keeps the chains alive
}

// ns per cluster-pair step per SIMD and the implied cycles (at `ghz`): waves_per_simd in {1 ... 8}, chains in {1, 2}
extern "C" int remd_roof_pair_step(remd_handle h, int waves_per_simd, int chains, double ghz, double* cycles_per_step_per_simd, double* us_total)
{
    if (!h || waves_per_simd < 1 || waves_per_simd > 8 || (chains != 1 && chains != 2)) return -1;
    hipSetDevice(h->device);
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) n_cu = prop.multiProcessorCount; }
    // 256-thread workgroups = one wavefront on each SIMD of a CU; waves_per_simd of them per CU, residency capped by dynamic LDS
    const int wg_per_cu = waves_per_simd;
    const size_t lds = std::max((size_t)4096, (size_t)(160 * 1024 / wg_per_cu - 1024) / 256 * 256);
    const int blocks = n_cu * wg_per_cu, iters = 4096;
    float* out = nullptr;
    REMD_CHECK(h, hipMalloc(&out, sizeof(float) * (size_t)blocks * 256));
    hipEvent_t e0, e1;
    REMD_CHECK(h, hipEventCreate(&e0)); REMD_CHECK(h, hipEventCreate(&e1));
    auto kern = chains == 1 ? roof_pair_step_kernel<1> : roof_pair_step_kernel<2>;
    REMD_CHECK(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float sL = 2.96f / 4294967296.f;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, h->stream, out, 64, 12345u, sL, 0.01f, 1.268f);
    hipEventRecord(e0, h->stream);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, h->stream, out, iters, 12345u, sL, 0.01f, 1.268f);
    hipEventRecord(e1, h->stream);
    REMD_CHECK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
    // every SIMD ran waves_per_simd wavefronts x iters x chains steps
    const double steps_per_simd = (double)waves_per_simd * iters * chains;
    if (us_total) *us_total = 1e3 * ms;
    if (cycles_per_step_per_simd) *cycles_per_step_per_simd = (ms * 1e-3) * (ghz * 1e9) / steps_per_simd;
    return 0;
}
