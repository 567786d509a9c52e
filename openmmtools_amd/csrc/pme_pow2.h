// Power-of-two meshes (64^3 of the headline split, 128^3 of DHFR): the mesh transforms held in REGISTERS.
//
// The mixed-radix passes of pme.hip walk a butterfly schedule (a table entry, a twiddle look-up and masked slots per
// butterfly, two workgroup barriers per stage): on 64 x 64 planes a wavefront issued ~2000 VALU + ~1000 SALU + ~380 LDS
// instructions per plane pass (profiles/r03_e_mesh_probe_variants.txt: "bound by the number of instructions of ALL kinds").
// Here a length N = R1 * R2 transform is two register butterflies (radix R1 = 8 or 16, then radix R2 <= 8) with ONE exchange
// through LDS between them, every index a compile-time constant:
//   x = x0 + R2 r,  k = k1 + R1 k2:   X[k1 + R1 k2] = sum_x0 W_R2^(x0 k2) [ W_N^(x0 k1) sum_r v[x0 + R2 r] W_R1^(r k1) ]
// and the output layout of a transform (thread q holds k = q + R2 m + R1 k2) is the input layout (q + R2 r) of the next one,
// so forward -> influence function -> inverse needs no reordering in between.  Thread-level model with the LDS bank
// arithmetic of every access: tools/models/xy_pow2_model.py.
#pragma once
#include <hip/hip_runtime.h>

#define P2_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ float2 p2_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 p2_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 p2_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * w (SIGN < 0, forward) or a * conj(w) (SIGN > 0, inverse); w = exp(-2 pi i p / n) as tabulated
template <int SIGN> __device__ __forceinline__ float2 p2_twid(float2 a, float2 w)
{
    return SIGN < 0 ? make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x) : make_float2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}
// a * (c + SIGN i s): a constant root of unity exp(SIGN i phi), c = cos phi, s = sin phi
template <int SIGN> __device__ __forceinline__ float2 p2_rot(float2 a, float c, float s)
{
    return make_float2(a.x * c - SIGN * s * a.y, a.y * c + SIGN * s * a.x);
}

// natural-order DFTs of 4, 8, 16 points in place; SIGN = -1 forward (exp(-2 pi i jk/n)), +1 inverse, unnormalised
template <int SIGN> __device__ __forceinline__ void p2_dft4(float2& v0, float2& v1, float2& v2, float2& v3)
{
    const float2 a = p2_add(v0, v2), b = p2_sub(v0, v2), c = p2_add(v1, v3), d = p2_sub(v1, v3);
    const float2 id = make_float2(-SIGN * d.y, SIGN * d.x);        // SIGN * i * d
    v0 = p2_add(a, c); v2 = p2_sub(a, c); v1 = p2_add(b, id); v3 = p2_sub(b, id);
}
template <int SIGN> __device__ __forceinline__ void p2_dft8(float2* v)
{
    p2_dft4<SIGN>(v[0], v[2], v[4], v[6]);                         // even inputs -> E[0..3] in v[0], v[2], v[4], v[6]
    p2_dft4<SIGN>(v[1], v[3], v[5], v[7]);                         // odd inputs  -> O[0..3] in v[1], v[3], v[5], v[7]
    const float h = 0.70710678118654752f;
    const float2 o0 = v[1];
    const float2 o1 = p2_rot<SIGN>(v[3], h, h);                    // * exp(SIGN i pi / 4)
    const float2 o2 = make_float2(-SIGN * v[5].y, SIGN * v[5].x);  // * SIGN i
    const float2 o3 = p2_rot<SIGN>(v[7], -h, h);                   // * exp(SIGN i 3 pi / 4)
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = p2_add(e0, o0); v[4] = p2_sub(e0, o0);
    v[1] = p2_add(e1, o1); v[5] = p2_sub(e1, o1);
    v[2] = p2_add(e2, o2); v[6] = p2_sub(e2, o2);
    v[3] = p2_add(e3, o3); v[7] = p2_sub(e3, o3);
}
template <int SIGN> __device__ __forceinline__ void p2_dft16(float2* v)
{
    // n = 4 a + b: X[k' + 4 c] = sum_b W4^(b c) [ W16^(b k') sum_a v[4 a + b] W4^(a k') ]
#pragma unroll
    for (int b = 0; b < 4; ++b) p2_dft4<SIGN>(v[b], v[b + 4], v[b + 8], v[b + 12]);     // v[b + 4 k'] = inner sum for (b, k')
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    // W16^(b k'), b, k' = 1..3: exponents 1 2 3 / 2 4 6 / 3 6 9
    v[1 + 4] = p2_rot<SIGN>(v[1 + 4], c1, s1);   v[1 + 8] = p2_rot<SIGN>(v[1 + 8], h, h);     v[1 + 12] = p2_rot<SIGN>(v[1 + 12], s1, c1);
    v[2 + 4] = p2_rot<SIGN>(v[2 + 4], h, h);     v[2 + 8] = make_float2(-SIGN * v[2 + 8].y, SIGN * v[2 + 8].x);   v[2 + 12] = p2_rot<SIGN>(v[2 + 12], -h, h);
    v[3 + 4] = p2_rot<SIGN>(v[3 + 4], s1, c1);   v[3 + 8] = p2_rot<SIGN>(v[3 + 8], -h, h);    v[3 + 12] = p2_rot<SIGN>(v[3 + 12], -c1, -s1);
    // outer sums over b for every k': inputs v[0 + 4k'], v[1 + 4k'], v[2 + 4k'], v[3 + 4k'] -> X[k' + 4 c] for c = 0..3
    float2 o[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float2 t0 = v[4 * k], t1 = v[4 * k + 1], t2 = v[4 * k + 2], t3 = v[4 * k + 3];
        p2_dft4<SIGN>(t0, t1, t2, t3);
        o[k] = t0; o[k + 4] = t1; o[k + 8] = t2; o[k + 12] = t3;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}
template <int SIGN, int R> __device__ __forceinline__ void p2_dft(float2* v)
{
    if (R == 16) p2_dft16<SIGN>(v);
    else if (R == 8) p2_dft8<SIGN>(v);
    else p2_dft4<SIGN>(v[0], v[1], v[2], v[3]);
}

// One line of N = R1 * R2 points shared by R2 neighbouring lanes of a wavefront (lane j of the line holds v[r] = element
// j + R2 r): the transform of the line, in[] -> out[], with the exchange between its two butterflies in the line's own
// LDS region `line` (>= N numbers, nobody else's): wave-local, no workgroup barrier.  out[m * R2 + k2] = element
// j + R2 m + R1 k2 of the result, m < R1 / R2.  Slot of (k1, j) in the exchange: k1 * R2 + ((j + k1) mod R2) -- both the
// writes (k1 fixed, j across lanes) and the reads (k1 = lane, j fixed) are free of bank conflicts when consecutive lines
// start 16 banks apart.  tw: exp(-2 pi i p / N), p < N, in LDS.
template <int SIGN, int R1, int R2>
__device__ __forceinline__ void p2_line_fft(float2* v, float2* out, float2* line, int j, const float2* tw)
{
    constexpr int MM = R1 / R2;
    p2_dft<SIGN, R1>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[k1] = p2_twid<SIGN>(v[k1], tw[j * k1]);
    P2_WAVE_SYNC();
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) line[k1 * R2 + ((j + k1) & (R2 - 1))] = v[k1];
    P2_WAVE_SYNC();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const int k1 = j + R2 * m;
        float2 u[R2];
#pragma unroll
        for (int jp = 0; jp < R2; ++jp) u[jp] = line[k1 * R2 + ((jp + k1) & (R2 - 1))];
        p2_dft<SIGN, R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) out[m * R2 + k2] = u[k2];
    }
    P2_WAVE_SYNC();
}

// ---- the plane pass: one (kz, replica) plane of N x N numbers, N = R1 * 8, on N * N / R1 threads.
// forward x | transpose | forward y | influence function (+ energy) | inverse y | transpose | inverse x: six trips through
// LDS and six workgroup barriers (the scheduled pass: eight stages, sixteen barriers).  LDS: the plane with rows PS = N + 8
// apart (2 PS mod 64 = 16: four consecutive lines of a 32-lane group sit on four disjoint quarters of the banks).
// infl_perm: the influence function in the order the threads hold the spectrum -- [plane][i * T + tid] for register i
// (pme_influence_table_kernel with perm_r1 = R1).
// 128 x 128 planes: 1024 threads = 4 wavefronts per SIMD; compiled for FIVE (<= 96 registers) so that a plane workgroup's 384 registers
// per SIMD lane and 140 KB of LDS fit beside two resident workgroups of the pair kernel (2 x 64 registers, 2 x 10 KB) -- DHFR's
// plane pass otherwise enters a CU only when the pair kernel's workgroups have left it (A/B: tools/build_variant.sh -DP2_XY128_WPE=4)
#ifndef P2_XY128_WPE
#define P2_XY128_WPE 5
#endif
template <int N, int R1, bool with_energy>
__global__ __launch_bounds__(N * N / R1) __attribute__((amdgpu_waves_per_eu(N == 64 ? 8 : P2_XY128_WPE, N == 64 ? 8 : P2_XY128_WPE)))
void pme_xy_pow2_kernel(int nz, float2* __restrict__ spec, const float2* __restrict__ tw, double* __restrict__ energy,
                        int n_eblk, const float* __restrict__ infl_perm, int infl_rep, int prio)
{
    constexpr int R2 = 8, T = N * N / R1, MM = R1 / R2, PS = N + 8;
    static_assert(N == R1 * R2 && (R1 == 8 || R1 == 16), "N = R1 * 8");
    if (prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* L = reinterpret_cast<float2*>(smem);          // [N][PS]; the x exchanges use it as [R1 * R2][N]
    float2* s_tw = L + N * PS;                            // [N]
    double* s_e = reinterpret_cast<double*>(s_tw + N);    // [T / 64]
    const int kz = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nzc = nz / 2 + 1;
    float2* __restrict__ P = spec + ((size_t)r * nzc + kz) * (N * N);
    const int y = tid & (N - 1), x0 = tid / N;
    const int x0u = __builtin_amdgcn_readfirstlane(x0);  // wave-uniform (N >= 64): the x twiddles are scalar loads
    float2 v[R1], w[R1];
    const unsigned off_in = (unsigned)(x0 * N + y);        // uniform base + 32-bit lane offset: the loads keep no 64-bit addresses
#pragma unroll
    for (int q = 0; q < R1; ++q) v[q] = (P + q * (R2 * N))[off_in];
    if (tid < N) s_tw[tid] = tw[tid];
    float2 twx[R1];
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) twx[k1] = tw[x0u * k1];
    // ---- forward x
    p2_dft<-1, R1>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[k1] = p2_twid<-1>(v[k1], twx[k1]);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) L[(k1 * R2 + x0) * N + y] = v[k1];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const int k1 = x0 + R2 * m;
        float2 u[R2];
#pragma unroll
        for (int xp = 0; xp < R2; ++xp) u[xp] = L[(k1 * R2 + xp) * N + y];
        p2_dft<-1, R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) w[m * R2 + k2] = u[k2];
    }
    __syncthreads();
    // ---- transpose: thread (y, x0) holds kx = x0 + R2 m + R1 k2; thread (line kx, j) takes y = j + R2 q
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) L[(x0 + R2 * m + R1 * k2) * PS + y] = w[m * R2 + k2];
    __syncthreads();
    const int kx = tid / R2, j = tid & (R2 - 1);
    float2* line = L + kx * PS;
#pragma unroll
    for (int q = 0; q < R1; ++q) v[q] = line[j + R2 * q];
    // ---- forward y: w[m * R2 + k2] = spectrum at ky = j + R2 m + R1 k2
    p2_line_fft<-1, R1, R2>(v, w, line, j, s_tw);
    // ---- influence function (+ energy)
    {
        const float* __restrict__ G = infl_perm + ((size_t)r * infl_rep + kz) * (N * N);
        float g[R1];
#pragma unroll
        for (int i = 0; i < R1; ++i) g[i] = (G + i * T)[(unsigned)tid];
        if (with_energy) {
            const float wz = (kz == 0 || 2 * kz == nz) ? 1.f : 2.f;     // Hermitian half: weight of the mirrored plane
            double e_acc = 0.0;
#pragma unroll
            for (int i = 0; i < R1; ++i) e_acc += 0.5 * (double)(wz * g[i]) * ((double)w[i].x * w[i].x + (double)w[i].y * w[i].y);
            for (int off = 32; off > 0; off >>= 1) e_acc += __shfl_xor(e_acc, off);
            if ((tid & 63) == 0) s_e[tid >> 6] = e_acc;
        }
        // register (m, k2) is input q = m + MM k2 of the inverse transform (ky = j + R2 q)
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[m + MM * k2] = make_float2(w[m * R2 + k2].x * g[m * R2 + k2], w[m * R2 + k2].y * g[m * R2 + k2]);
    }
    // ---- inverse y: w[m * R2 + k2] = value at y = j + R2 m + R1 k2   (ji: the lane's j laundered, see below)
    int ji = j;
    asm volatile("" : "+v"(ji));
    p2_line_fft<+1, R1, R2>(v, w, line, ji, s_tw);
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) line[ji + R2 * m + R1 * k2] = w[m * R2 + k2];
    __syncthreads();
    if (with_energy && tid == 0) {
        double tot = 0.0;
        for (int q = 0; q < T / 64; ++q) tot += s_e[q];
        energy[(size_t)r * n_eblk + kz] = tot;
    }
    // (the inverse half computes its LDS addresses afresh from laundered indices: the compiler would otherwise keep the forward half's
    //  addresses beyond the 64 KB an LDS instruction's offset field reaches alive through the whole pass -- 30 spilled registers on
    //  the 128 x 128 plane)
    int x0i = x0, yi = y;
    asm volatile("" : "+v"(x0i), "+v"(yi));
    // ---- transpose back: thread (y, x0) takes kx = x0 + R2 q
#pragma unroll
    for (int q = 0; q < R1; ++q) v[q] = L[(x0i + R2 * q) * PS + yi];
    __syncthreads();
    // ---- inverse x
    p2_dft<+1, R1>(v);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[k1] = p2_twid<+1>(v[k1], twx[k1]);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) L[(k1 * R2 + x0i) * N + yi] = v[k1];
    __syncthreads();
    // (the store addresses are recomputed behind an opaque move: the compiler would otherwise keep the eight 64-bit load
    //  addresses of the first lines alive through the whole pass and spill them)
    float2* Po = P; unsigned off_out = (unsigned)tid;
    asm volatile("" : "+s"(Po), "+v"(off_out));
    off_out = (off_out / N) * N + (off_out & (N - 1));      // x0 * N + y
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const int k1 = x0i + R2 * m;
        float2 u[R2];
#pragma unroll
        for (int xp = 0; xp < R2; ++xp) u[xp] = L[(k1 * R2 + xp) * N + yi];
        p2_dft<+1, R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) (Po + (R2 * m + R1 * k2) * N)[off_out] = u[k2];
    }
}
