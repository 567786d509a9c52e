// GBSA implicit solvent (gfx950): OpenMM's GBSAOBCForce -- OBC2 Born radii + ACE surface term -- in the form the reference's alchemical
// factory gives it (alchemy/alchemy.py:2144-2225, _alchemically_modify_GBSAOBCForce: a CustomGBForce whose expression strings are the
// definition followed here; include/remd_hip.h remd_set_gbsa states them).  NoCutoff systems only (the implicit-solvent test systems of
// the reference: tens to a few thousand atoms).  f64 restatement: oracle/gbsa.py, pinned to the reference's strings.
//
// Three launches per force evaluation, each the direct all-pairs sum of nocutoff.hip (workgroup = 64 atoms i of one replica, 16 wavefronts that
// split the partners j staged through LDS, every pair from both sides so that nothing is reduced across atoms):
//   gb_born_kernel    I_i = sum_j s_j H(r_ij; or_i, sr_j)  ->  B_i and dB_i/dI_i
//   gb_pair_kernel    self + surface + pair energies, the pair term's direct force on i, dE/dB_i  ->  c_i = dE/dB_i dB_i/dI_i
//   gb_chain_kernel   the force through the Born radii: sum_j [c_i s_j H'(r; or_i, sr_j) + c_j s_i H'(r; or_j, sr_i)] (x_j - x_i) / r
// s = lambda_electrostatics of the replica's state on the alchemical particles, 1 elsewhere.
#include "remd_internal.h"
#include "listed_terms.h"
#include <cmath>
#include <algorithm>

#define GB_KE 138.935485f
#define GB_OFFSET 0.009f
#define GB_SA 28.3919551f

struct gbsa_tables {
    int N = 0, n_tile = 0;
    float tau = 0.f; int sasa = 1;
    bool any_alch = false;
    float4* d_par = nullptr;               // [Npad] q, R, scale, alchemical
    float2* d_born = nullptr;              // [R][Npad] B, dB/dI
    float* d_c = nullptr;                  // [R][Npad] dE/dB dB/dI
    float* d_lam = nullptr; std::vector<float> lam_host;      // [R] lambda_electrostatics of each replica's state
    float* d_state_lam = nullptr; std::vector<float> state_lam_host;   // [K] of every state (u_kl columns)
    // a deep copy of the descriptor of remd_set_gbsa: the blocks of a phased propagation (api.hip) are set up from it
    remd_gbsa_desc store{}; std::vector<double> st_charge, st_radius, st_scale; std::vector<int32_t> st_alch;
    double* d_epart = nullptr; double* d_col = nullptr; int buf_R = 0;
};
static handle_table<gbsa_tables> g_gb;

template <typename T> static void dfree(T*& p) { if (p) { hipFree(p); p = nullptr; } }

// H(r; or1, sr2) of the computed value I and its derivative in r (alchemy.py:2195-2201; the step functions are constants under the derivative)
__device__ __forceinline__ void gb_H(float r, float or1, float sr2, float& H, float& dH)
{
    H = 0.f; dH = 0.f;
    if (r + sr2 - or1 < 0.f) return;
    const float U = r + sr2, D = fabsf(r - sr2);
    const bool moving = D > or1;
    const float L = moving ? D : or1, dL = moving ? (r > sr2 ? 1.f : -1.f) : 0.f;
    const bool inside = sr2 - r - or1 >= 0.f;
    const float iL = 1.f / L, iU = 1.f / U, ir = 1.f / r;
    const float C = inside ? 2.f * (1.f / or1 - iL) : 0.f, dC = inside ? 2.f * dL * iL * iL : 0.f;
    const float a = iU * iU - iL * iL, w = r - sr2 * sr2 * ir, lg = logf(L * iU);
    H = 0.5f * (iL - iU + 0.25f * w * a + 0.5f * lg * ir + C);
    dH = 0.5f * (-dL * iL * iL + iU * iU + 0.25f * (1.f + sr2 * sr2 * ir * ir) * a + 0.25f * w * (-2.f * iU * iU * iU + 2.f * dL * iL * iL * iL)
                 + 0.5f * ((dL * iL - iU) * ir - lg * ir * ir) + dC);
}

// The three kernels: workgroup = (64 atoms i, replica) of GB_WAVES wavefronts; lane = atom i, wavefront w takes the partners
// j = w (mod GB_WAVES) of every block of GB_BLOCK atoms staged in LDS, an atom's partial sums are added in wavefront order (fixed order).
// (One wavefront per tile summing N dependent partner terms -- each with a logarithm and a handful of divisions -- was 10 + 7 + 17 us for
// 22 atoms: profiles/r06_43.)
#define GB_WAVES 16
#define GB_BLOCK (64 * GB_WAVES)
__global__ __launch_bounds__(GB_BLOCK)
void gb_born_kernel(int N, int Npad, const float4* __restrict__ par, const float* __restrict__ lam, int lam_stride, const float4* __restrict__ pos, float2* __restrict__ born)
{
    __shared__ float4 s_x[GB_BLOCK];       // x, y, z, sr_j
    __shared__ float s_s[GB_BLOCK];        // s_j
    __shared__ float s_part[GB_WAVES][64];
    const int r = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const float4* P = pos + (size_t)r * Npad;
    const float l = lam[(size_t)r * lam_stride];         // (stride 0: every replica at one state's lambda, a u_kl column)
    const bool live = i < N;
    const float4 xi = live ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pi = live ? par[i] : make_float4(0.f, 1.f, 0.f, 0.f);
    const float or_i = pi.y - GB_OFFSET;
    float I = 0.f;
    for (int j0 = 0; j0 < N; j0 += GB_BLOCK) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < N) {
            const float4 xj = P[j0 + threadIdx.x], pj = par[j0 + threadIdx.x];
            s_x[threadIdx.x] = make_float4(xj.x, xj.y, xj.z, pj.z * (pj.y - GB_OFFSET));
            s_s[threadIdx.x] = pj.w != 0.f ? l : 1.f;
        }
        __syncthreads();
        if (!live) continue;
        const int jn = min(GB_BLOCK, N - j0);
        for (int k = w; k < jn; k += GB_WAVES) {
            if (j0 + k == i) continue;
            const float4 xj = s_x[k];
            const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
            float H, dH;
            gb_H(sqrtf(dx * dx + dy * dy + dz * dz), or_i, xj.w, H, dH);
            I += s_s[k] * H;
        }
    }
    s_part[w][lane] = I;
    __syncthreads();
    if (w == 0 && live) {
        I = 0.f;
        for (int q = 0; q < GB_WAVES; ++q) I += s_part[q][lane];
        const float psi = I * or_i, th = tanhf(psi - 0.8f * psi * psi + 4.85f * psi * psi * psi);
        const float B = 1.f / (1.f / or_i - th / pi.y);
        born[(size_t)r * Npad + i] = make_float2(B, B * B * (1.f - th * th) * (1.f - 1.6f * psi + 14.55f * psi * psi) * or_i / pi.y);
    }
}

template <bool ENERGY, bool FORCE>
__global__ __launch_bounds__(GB_BLOCK)
void gb_pair_kernel(int N, int Npad, float tau, int sasa, const float4* __restrict__ par, const float* __restrict__ lam, int lam_stride, const float4* __restrict__ pos,
                    const float2* __restrict__ born, float* __restrict__ cfac, long long* __restrict__ force, double* __restrict__ epart, int n_tile)
{
    __shared__ float4 s_x[GB_BLOCK];       // x, y, z, B_j
    __shared__ float s_q[GB_BLOCK];        // s_j q_j
    __shared__ float4 s_part[GB_WAVES][64];
    __shared__ double s_e[GB_WAVES];
    const int r = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const float4* P = pos + (size_t)r * Npad;
    const float2* BR = born + (size_t)r * Npad;
    const float l = lam[(size_t)r * lam_stride];         // (stride 0: every replica at one state's lambda, a u_kl column)
    const bool live = i < N;
    const float4 xi = live ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pi = live ? par[i] : make_float4(0.f, 1.f, 0.f, 0.f);
    const float2 bi = live ? BR[i] : make_float2(1.f, 0.f);
    const float si = pi.w != 0.f ? l : 1.f, Qi = si * pi.x;
    float fx = 0.f, fy = 0.f, fz = 0.f, dEdB = 0.f;
    double e = 0.0;
    if (live && w == 0) {
        const float self = 0.5f * GB_KE * tau * si * pi.x * pi.x / bi.x;
        dEdB += self / bi.x;
        if (ENERGY) e -= (double)self;
        if (sasa) {
            const float rb = pi.y / bi.x, rb2 = rb * rb, rb6 = rb2 * rb2 * rb2, pre = si * GB_SA * (pi.y + 0.14f) * (pi.y + 0.14f);
            dEdB -= 6.f * pre * rb6 / bi.x;
            if (ENERGY) e += (double)(pre * rb6);
        }
    }
    for (int j0 = 0; j0 < N; j0 += GB_BLOCK) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < N) {
            const float4 xj = P[j0 + threadIdx.x], pj = par[j0 + threadIdx.x];
            s_x[threadIdx.x] = make_float4(xj.x, xj.y, xj.z, BR[j0 + threadIdx.x].x);
            s_q[threadIdx.x] = (pj.w != 0.f ? l : 1.f) * pj.x;
        }
        __syncthreads();
        if (!live) continue;
        const int jn = min(GB_BLOCK, N - j0);
        for (int k = w; k < jn; k += GB_WAVES) {
            if (j0 + k == i) continue;
            const float4 xj = s_x[k];
            const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
            const float r2 = dx * dx + dy * dy + dz * dz;
            const float D = bi.x * xj.w, ex = __expf(-r2 / (4.f * D)), f2 = r2 + D * ex, inv_f = rsqrtf(f2);
            const float QQ = GB_KE * tau * Qi * s_q[k];
            const float dEdf = QQ / f2;
            dEdB += dEdf * ex * (1.f + r2 / (4.f * D)) * 0.5f * inv_f * xj.w;
            if (FORCE) { const float gr = dEdf * (1.f - 0.25f * ex) * inv_f; fx += gr * dx; fy += gr * dy; fz += gr * dz; }
            if (ENERGY) e -= 0.5 * (double)(QQ * inv_f);
        }
    }
    if (FORCE) {
        s_part[w][lane] = make_float4(fx, fy, fz, dEdB);
        __syncthreads();
        if (w == 0 && live) {
            fx = 0.f; fy = 0.f; fz = 0.f; dEdB = 0.f;
            for (int q = 0; q < GB_WAVES; ++q) { const float4 t = s_part[q][lane]; fx += t.x; fy += t.y; fz += t.z; dEdB += t.w; }
            cfac[(size_t)r * Npad + i] = dEdB * bi.y;
            add_force(force + (size_t)r * 3 * Npad, Npad, i, fx, fy, fz);
        }
    }
    if (ENERGY) {
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (lane == 0) s_e[w] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int q = 0; q < GB_WAVES; ++q) tot += s_e[q];
            epart[(size_t)r * n_tile + blockIdx.x] = tot;
        }
    }
}

__global__ __launch_bounds__(GB_BLOCK)
void gb_chain_kernel(int N, int Npad, const float4* __restrict__ par, const float* __restrict__ lam, int lam_stride, const float4* __restrict__ pos,
                     const float* __restrict__ cfac, long long* __restrict__ force)
{
    __shared__ float4 s_x[GB_BLOCK];       // x, y, z, or_j
    __shared__ float4 s_p[GB_BLOCK];       // sr_j, s_j, c_j, -
    __shared__ float4 s_part[GB_WAVES][64];
    const int r = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const float4* P = pos + (size_t)r * Npad;
    const float* Cf = cfac + (size_t)r * Npad;
    const float l = lam[(size_t)r * lam_stride];         // (stride 0: every replica at one state's lambda, a u_kl column)
    const bool live = i < N;
    const float4 xi = live ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pi = live ? par[i] : make_float4(0.f, 1.f, 0.f, 0.f);
    const float or_i = pi.y - GB_OFFSET, sr_i = pi.z * or_i, si = pi.w != 0.f ? l : 1.f, ci = live ? Cf[i] : 0.f;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int j0 = 0; j0 < N; j0 += GB_BLOCK) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < N) {
            const float4 xj = P[j0 + threadIdx.x], pj = par[j0 + threadIdx.x];
            const float orj = pj.y - GB_OFFSET;
            s_x[threadIdx.x] = make_float4(xj.x, xj.y, xj.z, orj);
            s_p[threadIdx.x] = make_float4(pj.z * orj, pj.w != 0.f ? l : 1.f, Cf[j0 + threadIdx.x], 0.f);
        }
        __syncthreads();
        if (!live) continue;
        const int jn = min(GB_BLOCK, N - j0);
        for (int k = w; k < jn; k += GB_WAVES) {
            if (j0 + k == i) continue;
            const float4 xj = s_x[k], pj = s_p[k];
            const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
            const float rr = sqrtf(dx * dx + dy * dy + dz * dz);
            float H, dH1, dH2;
            gb_H(rr, or_i, pj.x, H, dH1);              // B_i depends on j
            gb_H(rr, xj.w, sr_i, H, dH2);              // B_j depends on i
            const float gr = (ci * pj.y * dH1 + pj.z * si * dH2) / rr;
            fx += gr * dx; fy += gr * dy; fz += gr * dz;
        }
    }
    s_part[w][lane] = make_float4(fx, fy, fz, 0.f);
    __syncthreads();
    if (w == 0 && live) {
        fx = 0.f; fy = 0.f; fz = 0.f;
        for (int q = 0; q < GB_WAVES; ++q) { const float4 t = s_part[q][lane]; fx += t.x; fy += t.y; fz += t.z; }
        add_force(force + (size_t)r * 3 * Npad, Npad, i, fx, fy, fz);
    }
}

__global__ __launch_bounds__(64)
void gb_reduce_kernel(int n, const double* __restrict__ part, double* __restrict__ out, int stride, int offset, int add)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 64) e += part[(size_t)r * n + t];
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if (threadIdx.x == 0) { double* o = out + (size_t)r * stride + offset; *o = add ? *o + e : e; }
}

// Systems of up to 64 atoms (the implicit-solvent test systems the reference's sampler tests run on: AlanineDipeptideImplicit, 22 atoms):
// the three kernels above are one wavefront per replica summing N dependent partner terms each -- 10 + 7 + 17 us for 22 atoms
// (profiles/r06_43).  Here ONE launch, a workgroup of 16 wavefronts per replica: lane = atom i, wavefront w takes the partners
// j = w, w + 16, w + 32, w + 48, the 16 partial sums of an atom go through LDS and are added in wavefront order (a fixed order: the
// result does not depend on scheduling); Born radii, dE/dB and c_i stay in LDS between the three passes.
template <bool ENERGY, bool FORCE>
__global__ __launch_bounds__(1024)
void gb_small_kernel(int N, int Npad, float tau, int sasa, const float4* __restrict__ par, const float* __restrict__ lam, int lam_stride, const float4* __restrict__ pos,
                     long long* __restrict__ force, double* __restrict__ out, int out_stride, int out_offset, int add)
{
    __shared__ float4 s_x[64];             // x, y, z, or_j
    __shared__ float4 s_p[64];             // q_j, R_j, sr_j, s_j
    __shared__ float2 s_born[64];          // B, dB/dI
    __shared__ float s_c[64];              // dE/dB dB/dI
    __shared__ float4 s_part[16][64];
    __shared__ double s_e[16];
    const int r = blockIdx.x, i = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float4* P = pos + (size_t)r * Npad;
    const float l = lam[(size_t)r * lam_stride];         // (stride 0: every replica at one state's lambda, a u_kl column)
    if (w == 0) {
        const float4 x = i < N ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f), p = i < N ? par[i] : make_float4(0.f, 1.f, 0.f, 0.f);
        const float orj = p.y - GB_OFFSET;
        s_x[i] = make_float4(x.x, x.y, x.z, orj);
        s_p[i] = make_float4(p.x, p.y, p.z * orj, p.w != 0.f ? l : 1.f);
    }
    __syncthreads();
    const bool live = i < N;
    const float4 xi = s_x[i], pi = s_p[i];
    const float or_i = xi.w, sr_i = pi.z, si = pi.w;
    // pass 1: I_i -> B_i, dB_i/dI_i
    {
        float I = 0.f;
        if (live)
            for (int j = w; j < N; j += 16) {
                if (j == i) continue;
                const float4 xj = s_x[j], pj = s_p[j];
                const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                float H, dH;
                gb_H(sqrtf(dx * dx + dy * dy + dz * dz), or_i, pj.z, H, dH);
                I += pj.w * H;
            }
        s_part[w][i].x = I;
    }
    __syncthreads();
    if (w == 0 && live) {
        float I = 0.f;
        for (int q = 0; q < 16; ++q) I += s_part[q][i].x;
        const float psi = I * or_i, th = tanhf(psi - 0.8f * psi * psi + 4.85f * psi * psi * psi);
        const float B = 1.f / (1.f / or_i - th / pi.y);
        s_born[i] = make_float2(B, B * B * (1.f - th * th) * (1.f - 1.6f * psi + 14.55f * psi * psi) * or_i / pi.y);
    }
    __syncthreads();
    // pass 2: self + surface + pair energies, the pair term's direct force on i, dE/dB_i
    float fx = 0.f, fy = 0.f, fz = 0.f;
    {
        const float2 bi = live ? s_born[i] : make_float2(1.f, 0.f);
        const float Qi = si * pi.x;
        float dEdB = 0.f;
        double e = 0.0;
        if (live && w == 0) {
            const float self = 0.5f * GB_KE * tau * si * pi.x * pi.x / bi.x;
            dEdB += self / bi.x;
            if (ENERGY) e -= (double)self;
            if (sasa) {
                const float rb = pi.y / bi.x, rb2 = rb * rb, rb6 = rb2 * rb2 * rb2, pre = si * GB_SA * (pi.y + 0.14f) * (pi.y + 0.14f);
                dEdB -= 6.f * pre * rb6 / bi.x;
                if (ENERGY) e += (double)(pre * rb6);
            }
        }
        if (live)
            for (int j = w; j < N; j += 16) {
                if (j == i) continue;
                const float4 xj = s_x[j], pj = s_p[j];
                const float Bj = s_born[j].x;
                const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                const float r2 = dx * dx + dy * dy + dz * dz;
                const float D = bi.x * Bj, ex = __expf(-r2 / (4.f * D)), f2 = r2 + D * ex, inv_f = rsqrtf(f2);
                const float QQ = GB_KE * tau * Qi * (pj.w * pj.x);
                const float dEdf = QQ / f2;
                dEdB += dEdf * ex * (1.f + r2 / (4.f * D)) * 0.5f * inv_f * Bj;
                if (FORCE) { const float gr = dEdf * (1.f - 0.25f * ex) * inv_f; fx += gr * dx; fy += gr * dy; fz += gr * dz; }
                if (ENERGY) e -= 0.5 * (double)(QQ * inv_f);
            }
        s_part[w][i] = make_float4(fx, fy, fz, dEdB);
        if (ENERGY) {
            for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
            if (i == 0) s_e[w] = e;
        }
        __syncthreads();
        if (w == 0) {
            fx = 0.f; fy = 0.f; fz = 0.f; dEdB = 0.f;
            for (int q = 0; q < 16; ++q) { const float4 t = s_part[q][i]; fx += t.x; fy += t.y; fz += t.z; dEdB += t.w; }
            s_c[i] = dEdB * bi.y;
            if (ENERGY && i == 0) {
                double tot = 0.0;
                for (int q = 0; q < 16; ++q) tot += s_e[q];
                double* o = out + (size_t)r * out_stride + out_offset;
                *o = add ? *o + tot : tot;
            }
        }
        __syncthreads();
    }
    if (!FORCE) return;
    // pass 3: the force through the Born radii
    {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (live) {
            const float ci = s_c[i];
            for (int j = w; j < N; j += 16) {
                if (j == i) continue;
                const float4 xj = s_x[j], pj = s_p[j];
                const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                const float rr = sqrtf(dx * dx + dy * dy + dz * dz);
                float H, dH1, dH2;
                gb_H(rr, or_i, pj.z, H, dH1);              // B_i depends on j
                gb_H(rr, xj.w, sr_i, H, dH2);              // B_j depends on i
                const float gr = (ci * pj.w * dH1 + s_c[j] * si * dH2) / rr;
                gx += gr * dx; gy += gr * dy; gz += gr * dz;
            }
        }
        s_part[w][i] = make_float4(gx, gy, gz, 0.f);
        __syncthreads();
        if (w == 0 && live) {
            for (int q = 0; q < 16; ++q) { const float4 t = s_part[q][i]; fx += t.x; fy += t.y; fz += t.z; }
            add_force(force + (size_t)r * 3 * Npad, Npad, i, fx, fy, fz);
        }
    }
}

void remd_gbsa_release(remd_ctx* h)
{
    gbsa_tables* t = g_gb.find(h);
    if (t) {
        dfree(t->d_par); dfree(t->d_born); dfree(t->d_c); dfree(t->d_lam); dfree(t->d_state_lam); dfree(t->d_epart); dfree(t->d_col);
        g_gb.erase(h);
    }
    h->gbsa = 0;
}

int remd_set_gbsa(remd_handle h, const remd_gbsa_desc* d)
{
    if (!h) return remd_fail(h, -1, "remd_set_gbsa: NULL handle");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    remd_gbsa_release(h);
    h->config_version++;
    h->forces_valid = false;
    if (!d) return 0;
    if (!h->has_system || !h->nocutoff) return remd_fail(h, -3, "remd_set_gbsa: GBSA needs a system with a NoCutoff NonbondedForce (call remd_set_system first)");
    if (d->n_atoms != h->N || !d->charge || !d->radius || !d->scale || !(d->solute_dielectric > 0) || !(d->solvent_dielectric > 0)) return remd_fail(h, -1, "remd_set_gbsa: bad arguments");
    gbsa_tables& t = g_gb[h];
    t.N = h->N; t.n_tile = (h->N + 63) / 64;
    t.tau = (float)(1.0 / d->solute_dielectric - 1.0 / d->solvent_dielectric);
    t.sasa = d->surface_area ? 1 : 0;
    std::vector<float4> par(h->Npad, make_float4(0.f, 1.f, 0.f, 0.f));
    for (int i = 0; i < h->N; ++i) {
        if (!(d->radius[i] > 0.009)) { remd_gbsa_release(h); return remd_fail(h, -1, "remd_set_gbsa: radii must exceed the offset 0.009 nm"); }
        const bool a = d->alchemical && d->alchemical[i];
        t.any_alch |= a;
        par[i] = make_float4((float)d->charge[i], (float)d->radius[i], (float)d->scale[i], a ? 1.f : 0.f);
    }
    REMD_CHECK(h, hipMalloc(&t.d_par, sizeof(float4) * par.size()));
    REMD_CHECK(h, hipMemcpy(t.d_par, par.data(), sizeof(float4) * par.size(), hipMemcpyHostToDevice));
    h->gbsa = 1;
    if (!h->parent) {
        t.store = *d;
        t.st_charge.assign(d->charge, d->charge + h->N); t.st_radius.assign(d->radius, d->radius + h->N); t.st_scale.assign(d->scale, d->scale + h->N);
        t.store.charge = t.st_charge.data(); t.store.radius = t.st_radius.data(); t.store.scale = t.st_scale.data();
        if (d->alchemical) { t.st_alch.assign(d->alchemical, d->alchemical + h->N); t.store.alchemical = t.st_alch.data(); } else { t.st_alch.clear(); t.store.alchemical = nullptr; }
    }
    return 0;
}

// the implicit solvent of `parent` on one of its blocks (api.hip phase_children)
int remd_gbsa_clone(remd_ctx* parent, remd_ctx* child)
{
    gbsa_tables* t = g_gb.find(parent);
    if (!t || !parent->gbsa) return 0;
    const int rc = remd_set_gbsa(child, &t->store);
    if (rc) return remd_fail(parent, rc, std::string("phases: ") + child->err);
    return 0;
}

static int gb_buffers(remd_ctx* h, gbsa_tables& t)
{
    if (t.buf_R == h->R && t.d_born) return 0;
    dfree(t.d_born); dfree(t.d_c); dfree(t.d_lam); dfree(t.d_epart); dfree(t.d_col);
    REMD_CHECK(h, hipMalloc(&t.d_born, sizeof(float2) * (size_t)h->R * h->Npad));
    REMD_CHECK(h, hipMalloc(&t.d_c, sizeof(float) * (size_t)h->R * h->Npad));
    REMD_CHECK(h, hipMalloc(&t.d_lam, sizeof(float) * h->R));
    REMD_CHECK(h, hipMalloc(&t.d_epart, sizeof(double) * (size_t)h->R * t.n_tile));
    REMD_CHECK(h, hipMalloc(&t.d_col, sizeof(double) * h->R));
    t.buf_R = h->R; t.lam_host.clear();
    return 0;
}

static int gb_set_lambdas(remd_ctx* h, gbsa_tables& t, const std::vector<float>& lam)
{
    if (lam == t.lam_host) return 0;
    REMD_CHECK(h, hipMemcpyAsync(t.d_lam, lam.data(), sizeof(float) * lam.size(), hipMemcpyHostToDevice, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    t.lam_host = lam;
    return 0;
}

// lambda_electrostatics of state k for the alchemical particles: region 1 of the general regions (the factory's GBSA knows one region), else the state's own
static float gb_state_lambda(remd_ctx* h, int k)
{
    double le = 1.0;
    if (h->n_regions > 0) { if (remd_regions_state_le(h, k, 0, &le)) le = 1.0; }
    else if (!h->lam_e.empty() && k >= 0 && k < (int)h->lam_e.size()) le = h->lam_e[k];
    return (float)le;
}

// the one-launch kernel for up to 64 atoms (REMD_GB_SMALL=0: the three launches, for the A/B and the tests of the large path)
static bool gb_small(const gbsa_tables& t)
{
    return t.N <= 64 && !(getenv("REMD_GB_SMALL") && atoi(getenv("REMD_GB_SMALL")) == 0);
}

int remd_gbsa_forces(remd_ctx* h, bool with_energy, int ep_slot)
{
    gbsa_tables* tp = g_gb.find(h);
    if (!tp) return remd_fail(h, -2, "GBSA: no tables on this handle");
    gbsa_tables& t = *tp;
    int rc = gb_buffers(h, t);
    if (rc) return rc;
    std::vector<float> lam(h->R, 1.f);
    if (t.any_alch) for (int r = 0; r < h->R; ++r) lam[r] = gb_state_lambda(h, h->labels.empty() ? 0 : (int)h->labels[h->r_begin + r]);
    if ((rc = gb_set_lambdas(h, t, lam))) return rc;
    remd_prof_scope ps(h, "gbsa");
    if (gb_small(t)) {
        if (with_energy) hipLaunchKernelGGL((gb_small_kernel<true, true>), dim3(h->R), dim3(1024), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_lam, 1, h->d_pos, h->d_force, h->d_epart, h->n_epart, ep_slot, 0);
        else hipLaunchKernelGGL((gb_small_kernel<false, true>), dim3(h->R), dim3(1024), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_lam, 1, h->d_pos, h->d_force, (double*)nullptr, 0, 0, 0);
        REMD_CHECK(h, hipGetLastError());
        return 0;
    }
    const dim3 grid(t.n_tile, h->R);
    hipLaunchKernelGGL(gb_born_kernel, grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.d_par, t.d_lam, 1, h->d_pos, t.d_born);
    if (with_energy) {
        hipLaunchKernelGGL((gb_pair_kernel<true, true>), grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_lam, 1, h->d_pos, t.d_born, t.d_c, h->d_force, t.d_epart, t.n_tile);
        hipLaunchKernelGGL(gb_reduce_kernel, dim3(h->R), dim3(64), 0, h->stream, t.n_tile, t.d_epart, h->d_epart, h->n_epart, ep_slot, 0);
    } else
        hipLaunchKernelGGL((gb_pair_kernel<false, true>), grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_lam, 1, h->d_pos, t.d_born, t.d_c, h->d_force, (double*)nullptr, t.n_tile);
    hipLaunchKernelGGL(gb_chain_kernel, grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.d_par, t.d_lam, 1, h->d_pos, t.d_c, h->d_force);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

// u_kl: the GB energy of every replica at every state's lambda, ADDED to d_alch[r][k] (alchemical particles only: else it is the same in every column)
int remd_gbsa_ukl(remd_ctx* h, double* d_alch)
{
    gbsa_tables* tp = g_gb.find(h);
    if (!tp || !tp->any_alch) return 0;
    gbsa_tables& t = *tp;
    int rc = gb_buffers(h, t);
    if (rc) return rc;
    const dim3 grid(t.n_tile, h->R);
    // the states' lambdas on the device (uploaded when they change: one synchronisation then, none per column)
    {
        std::vector<float> sl(h->K);
        for (int k = 0; k < h->K; ++k) sl[k] = gb_state_lambda(h, k);
        if (sl != t.state_lam_host) {
            if ((int)t.state_lam_host.size() != h->K) { dfree(t.d_state_lam); REMD_CHECK(h, hipMalloc(&t.d_state_lam, sizeof(float) * h->K)); }
            REMD_CHECK(h, hipMemcpyAsync(t.d_state_lam, sl.data(), sizeof(float) * h->K, hipMemcpyHostToDevice, h->stream));
            REMD_CHECK(h, hipStreamSynchronize(h->stream));
            t.state_lam_host = sl;
        }
    }
    for (int k = 0; k < h->K; ++k) {
        if (gb_small(t)) {
            hipLaunchKernelGGL((gb_small_kernel<true, false>), dim3(h->R), dim3(1024), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_state_lam + k, 0, h->d_pos, (long long*)nullptr, d_alch, h->K, k, 1);
            continue;
        }
        hipLaunchKernelGGL(gb_born_kernel, grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.d_par, t.d_state_lam + k, 0, h->d_pos, t.d_born);
        hipLaunchKernelGGL((gb_pair_kernel<true, false>), grid, dim3(GB_BLOCK), 0, h->stream, t.N, h->Npad, t.tau, t.sasa, t.d_par, t.d_state_lam + k, 0, h->d_pos, t.d_born, t.d_c,
                           (long long*)nullptr, t.d_epart, t.n_tile);
        hipLaunchKernelGGL(gb_reduce_kernel, dim3(h->R), dim3(64), 0, h->stream, t.n_tile, t.d_epart, d_alch, h->K, k, 1);
    }
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
