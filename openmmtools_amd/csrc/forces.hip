// Force / potential-energy kernels (gfx950): harmonic external force, bonded terms,
// nonbonded direct space.  Forces accumulate in 64-bit fixed point (deterministic integer
// atomics, scale 2^32); energies are written as per-block f64 partials that are summed in a
// fixed order by reduce_energy_kernel (bit-reproducible).
//
// Functional forms restated from the reference:
//   testsystems.HarmonicOscillator      testsystems.py:779-786   U = K/2 ((x-x0)^2+y^2+z^2) + U0
//   NonbondedForce / bonded terms       OpenMM semantics as built by testsystems.py:1957-2017,
//                                       3496-3527 (see oracle/md_oracle.py for the f64 restatement)
#include "remd_internal.h"

#define EP_EXT      0
#define EP_BOND     1
#define EP_ANGLE    2
#define EP_TORSION  3
#define EP_EXCEPT   4
#define EP_EXCLCORR 5
#define EP_PME      6
#define EP_CONST    7
#define EP_NB0      8      // first nonbonded block slot

__device__ __forceinline__ void add_force(long long* __restrict__ F, int Npad, int i, float fx, float fy, float fz)
{
    unsigned long long* U = reinterpret_cast<unsigned long long*>(F);
    atomicAdd(&U[i],            (unsigned long long)(long long)((double)fx * REMD_FORCE_SCALE));
    atomicAdd(&U[Npad + i],     (unsigned long long)(long long)((double)fy * REMD_FORCE_SCALE));
    atomicAdd(&U[2 * Npad + i], (unsigned long long)(long long)((double)fz * REMD_FORCE_SCALE));
}

template <bool ENERGY>
__global__ __launch_bounds__(64)
void ext_force_kernel(int n_ext, const int* __restrict__ ext_atoms, float K, float x0, double U0,
                      int Npad, const float4* __restrict__ pos, long long* __restrict__ force,
                      double* __restrict__ epart, int n_epart)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n_ext; t += 64) {
        const int i = ext_atoms[t];
        const float4 p = pos[(size_t)r * Npad + i];
        const float dx = p.x - x0;
        add_force(force + (size_t)r * 3 * Npad, Npad, i, -K * dx, -K * p.y, -K * p.z);
        if (ENERGY) e += 0.5 * (double)K * ((double)dx * dx + (double)p.y * p.y + (double)p.z * p.z) + U0;
    }
    if (ENERGY) {
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (threadIdx.x == 0) epart[(size_t)r * n_epart + EP_EXT] = e;
    }
}

// sums the partial slots of each replica in fixed order: lane-strided, then xor-shuffle tree
__global__ __launch_bounds__(64)
void reduce_energy_kernel(int n_epart, const double* __restrict__ epart, double* __restrict__ potential)
{
    const int r = blockIdx.x;
    double e = 0.0;
    for (int t = threadIdx.x; t < n_epart; t += 64) e += epart[(size_t)r * n_epart + t];
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
    if (threadIdx.x == 0) potential[r] = e;
}

// u_kl rows (states.py:1908-1917 with pressure=None; paralleltempering.py:206-215):
//   u[r][l] = beta_l * (U_r + econst_l)          when all states share the Hamiltonian.
// The lambda-dependent part (alchemical states) is added by the alchemical kernel.
__global__ void assemble_ukl_kernel(int R, int K, const double* __restrict__ potential,
                                    const double* __restrict__ beta, const double* __restrict__ econst,
                                    const double* __restrict__ alch /*[R][K] or null*/,
                                    double* __restrict__ ukl_rows)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * K) return;
    const int r = t / K, l = t % K;
    double U = potential[r] + econst[l];
    if (alch) U += alch[t];
    ukl_rows[t] = beta[l] * U;
}

int remd_build_nonbonded(remd_ctx* h, const remd_system_desc* d)
{
    (void)d;
    h->nb_method = d->nb_method;
    if (d->nb_method != REMD_NB_NONE) return remd_fail(h, -4, "nonbonded forces not built into this libremd_hip.so yet");
    return 0;
}

int remd_compute_forces(remd_ctx* h, bool with_energy)
{
    REMD_CHECK(h, hipMemsetAsync(h->d_force, 0, sizeof(long long) * 3 * (size_t)h->Npad * h->R, h->stream));
    if (with_energy)
        REMD_CHECK(h, hipMemsetAsync(h->d_epart, 0, sizeof(double) * (size_t)h->n_epart * h->R, h->stream));
    if (h->n_ext > 0) {
        remd_prof_scope ps(h, "ext_force");
        if (with_energy)
            hipLaunchKernelGGL(ext_force_kernel<true>, dim3(h->R), dim3(64), 0, h->stream, h->n_ext, h->d_ext_atoms,
                               (float)h->ext_K, (float)h->ext_x0, h->ext_U0, h->Npad, h->d_pos, h->d_force, h->d_epart, h->n_epart);
        else
            hipLaunchKernelGGL(ext_force_kernel<false>, dim3(h->R), dim3(64), 0, h->stream, h->n_ext, h->d_ext_atoms,
                               (float)h->ext_K, (float)h->ext_x0, h->ext_U0, h->Npad, h->d_pos, h->d_force, h->d_epart, h->n_epart);
    }
    if (with_energy) {
        hipLaunchKernelGGL(reduce_energy_kernel, dim3(h->R), dim3(64), 0, h->stream, h->n_epart, h->d_epart, h->d_potential);
    }
    REMD_CHECK(h, hipGetLastError());
    h->forces_valid = true;
    return 0;
}

int remd_assemble_ukl(remd_ctx* h, double* d_rows)
{
    const int n = h->R * h->K;
    hipLaunchKernelGGL(assemble_ukl_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->R, h->K,
                       h->d_potential, h->d_beta, h->d_econst, (const double*)nullptr, d_rows);
    REMD_CHECK(h, hipGetLastError());
    return 0;
}
