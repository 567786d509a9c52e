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
