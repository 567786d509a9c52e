#include "remd_internal.h"
#include "rng.h"
#include <cmath>

// Monte Carlo barostat: what an NPT ThermodynamicState means in the reference (states.py:1177-1181 adds an
// openmm.MonteCarloBarostat, frequency 25, to the System; it fires inside LangevinIntegrator's addUpdateContextState step,
// integrators.py:1313).  The algorithm is OpenMM's MonteCarloBarostatImpl::updateContextState, restated: every
// `frequency` steps  dV = volumeScale * 2 (u - 1/2);  every molecule's centre (arithmetic mean, wrapped into the box) is
// scaled by s = (V'/V)^(1/3) together with the box;  w = U' - U + p dV - N_mol kT ln(V'/V);  reject (restore) if w > 0 and
// u' > exp(-w / kT);  after >= 10 attempts volumeScale /= 1.1 below 25 % acceptance, *= 1.1 (capped at 0.3 V) above 75 %.
// All local replicas attempt at once, each with its own state's p and kT and its own Philox draws.
__global__ void baro_draw_kernel(int R, int r_begin, uint64_t seed, long long attempt, float* __restrict__ box,
                                 float* __restrict__ box_old, double* __restrict__ st, const unsigned int* __restrict__ noise_id)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double* S = st + (size_t)r * 8;
    const double Lx = box[4 * r], Ly = box[4 * r + 1], Lz = box[4 * r + 2];
    const double V = Lx * Ly * Lz;
    if (S[0] <= 0.0) S[0] = 0.01 * V;                                   // initial volumeScale (MonteCarloBarostatImpl::initialize)
    const philox4 w = remd_philox(seed, REMD_STREAM_BAROSTAT, 0u, noise_id ? noise_id[r] : (uint32_t)(r_begin + r), (uint64_t)attempt);
    const double dV = S[0] * 2.0 * (remd_u53(w.w[2], w.w[3]) - 0.5);
    const double newV = V + dV;
    const double scale = cbrt(newV / V);
    S[5] = dV; S[6] = newV; S[7] = V;
    box_old[4 * r] = box[4 * r]; box_old[4 * r + 1] = box[4 * r + 1]; box_old[4 * r + 2] = box[4 * r + 2];
    box[4 * r] = (float)(Lx * scale); box[4 * r + 1] = (float)(Ly * scale); box[4 * r + 2] = (float)(Lz * scale);
}

// one thread per molecule (contiguous atom range): centre -> wrapped centre -> scaled centre (OpenMM scalePositions)
__global__ __launch_bounds__(256)
void baro_scale_kernel(int n_mol, const int* __restrict__ first, const int* __restrict__ size, int Npad,
                       float4* __restrict__ pos, const float* __restrict__ box_old, const double* __restrict__ st)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (g >= n_mol) return;
    const double* S = st + (size_t)r * 8;
    const float scale = (float)cbrt(S[6] / S[7]);
    float4* P = pos + (size_t)r * Npad;
    const int a0 = first[g], n = size[g];
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int k = 0; k < n; ++k) { const float4 p = P[a0 + k]; cx += p.x; cy += p.y; cz += p.z; }
    const float inv = 1.f / (float)n;
    cx *= inv; cy *= inv; cz *= inv;
    const float Lx = box_old[4 * r], Ly = box_old[4 * r + 1], Lz = box_old[4 * r + 2];
    const float wx = cx - floorf(cx / Lx) * Lx, wy = cy - floorf(cy / Ly) * Ly, wz = cz - floorf(cz / Lz) * Lz;
    const float dx = wx * (scale - 1.f) - (cx - wx), dy = wy * (scale - 1.f) - (cy - wy), dz = wz * (scale - 1.f) - (cz - wz);
    for (int k = 0; k < n; ++k) { float4 p = P[a0 + k]; p.x += dx; p.y += dy; p.z += dz; P[a0 + k] = p; }
}

__global__ void baro_decide_kernel(int R, int r_begin, uint64_t seed, long long attempt, int n_mol, const double* __restrict__ U_old,
                                   const double* __restrict__ U_new, const int64_t* __restrict__ labels,
                                   const double* __restrict__ beta, const double* __restrict__ pressure,
                                   const double* __restrict__ econst, double econst_vref,
                                   float* __restrict__ box, const float* __restrict__ box_old, double* __restrict__ st,
                                   int* __restrict__ accepted, const unsigned int* __restrict__ noise_id,
                                   float min_edge, unsigned int* __restrict__ err)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double* S = st + (size_t)r * 8;
    const int k = (int)labels[r_begin + r];
    const double kT = 1.0 / beta[k];
    // + the current state's long-range constant ~ 1/V (alchemical sterics correction), which d_potential does not carry
    const double dlr = (econst_vref > 0.0) ? econst[k] * econst_vref * (1.0 / S[6] - 1.0 / S[7]) : 0.0;
    const double w = U_new[r] - U_old[r] + dlr + pressure[k] * S[5] - (double)n_mol * kT * log(S[6] / S[7]);
    const philox4 q = remd_philox(seed, REMD_STREAM_BAROSTAT, 1u, noise_id ? noise_id[r] : (uint32_t)(r_begin + r), (uint64_t)attempt);
    bool reject = !(w <= 0.0) && !(remd_u53(q.w[2], q.w[3]) <= exp(-w / kT));           // NaN energies reject
    // a trial box with an edge below twice the longer cutoff (the Coulomb range of the Ewald split may exceed the NonbondedForce
    // cutoff) was evaluated with a broken minimum image: never accept it, and say so -- OpenMM raises "The periodic box size has
    // decreased to less than twice the nonbonded cutoff" here (sticky device flag 6: api.hip turns it into the error)
    if (fminf(box[4 * r], fminf(box[4 * r + 1], box[4 * r + 2])) < min_edge) { reject = true; atomicCAS(err, 0u, 6u); }
    accepted[r] = reject ? 0 : 1;
    if (reject) { box[4 * r] = box_old[4 * r]; box[4 * r + 1] = box_old[4 * r + 1]; box[4 * r + 2] = box_old[4 * r + 2]; }
    else { S[2] += 1.0; S[4] += 1.0; }
    S[1] += 1.0; S[3] += 1.0;
    if (S[1] >= 10.0) {
        const double V = (double)box[4 * r] * (double)box[4 * r + 1] * (double)box[4 * r + 2];
        if (S[2] < 0.25 * S[1]) { S[0] /= 1.1; S[1] = 0.0; S[2] = 0.0; }
        else if (S[2] > 0.75 * S[1]) { S[0] = fmin(S[0] * 1.1, V * 0.3); S[1] = 0.0; S[2] = 0.0; }
    }
}

__global__ __launch_bounds__(256)
void baro_restore_kernel(int N, int Npad, const int* __restrict__ accepted, float4* __restrict__ pos, const float4* __restrict__ x0,
                         long long* __restrict__ force, const long long* __restrict__ f0, double* __restrict__ potential,
                         const double* __restrict__ U0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (accepted[r]) return;
    if (i == 0) potential[r] = U0[r];
    if (i >= N) return;
    const size_t o = (size_t)r * Npad, of = (size_t)r * 3 * Npad;
    pos[o + i] = x0[o + i];
    force[of + i] = f0[of + i]; force[of + Npad + i] = f0[of + Npad + i]; force[of + 2 * Npad + i] = f0[of + 2 * Npad + i];
}

int remd_barostat_buffers(remd_ctx* h)
{
    const int R = h->R, Npad = h->Npad;
    if (!h->d_baro) {
        REMD_CHECK(h, hipMalloc(&h->d_baro, sizeof(double) * 8 * R)); REMD_CHECK(h, hipMemsetAsync(h->d_baro, 0, sizeof(double) * 8 * R, h->stream));
        REMD_CHECK(h, hipMalloc(&h->d_box_old, sizeof(float) * 4 * R));
        REMD_CHECK(h, hipMalloc(&h->d_baro_x0, sizeof(float4) * (size_t)R * Npad));
        REMD_CHECK(h, hipMalloc(&h->d_baro_f0, sizeof(long long) * 3 * (size_t)R * Npad));
        REMD_CHECK(h, hipMalloc(&h->d_baro_U0, sizeof(double) * R));
        REMD_CHECK(h, hipMalloc(&h->d_baro_acc, sizeof(int) * R));
    }
    return 0;
}

int remd_barostat_attempt(remd_ctx* h)
{
    const int* grp_first = nullptr; const int* grp_size = nullptr;
    const int n_groups = remd_nb_molecules(h, &grp_first, &grp_size);
    if (n_groups <= 0) return remd_fail(h, -3, "barostat: the system has no molecule table (needs a NonbondedForce)");
    const int R = h->R, Npad = h->Npad;
    int rc;
    if ((rc = remd_barostat_buffers(h))) return rc;
    h->force_zeroed = false;
    if ((rc = remd_compute_forces(h, true))) return rc;                         // U and forces of the current configuration
    REMD_CHECK(h, hipMemcpyAsync(h->d_baro_U0, h->d_potential, sizeof(double) * R, hipMemcpyDeviceToDevice, h->stream));
    REMD_CHECK(h, hipMemcpyAsync(h->d_baro_f0, h->d_force, sizeof(long long) * 3 * (size_t)R * Npad, hipMemcpyDeviceToDevice, h->stream));
    REMD_CHECK(h, hipMemcpyAsync(h->d_baro_x0, h->d_pos, sizeof(float4) * (size_t)R * Npad, hipMemcpyDeviceToDevice, h->stream));
    const long long attempt = h->baro_attempts++;
    hipLaunchKernelGGL(baro_draw_kernel, dim3((R + 63) / 64), dim3(64), 0, h->stream, R, h->r_begin, h->seed, attempt, h->d_box,
                       h->d_box_old, h->d_baro, h->d_noise_id);
    hipLaunchKernelGGL(baro_scale_kernel, dim3((n_groups + 255) / 256, R), dim3(256), 0, h->stream, n_groups, grp_first,
                       grp_size, Npad, h->d_pos, h->d_box_old, h->d_baro);
    h->box_uniform = false;                  // (every replica draws its own volume)
    h->box_version++;
    h->force_zeroed = false;
    if ((rc = remd_compute_forces(h, true))) return rc;                         // U' and forces of the scaled configuration
    hipLaunchKernelGGL(baro_decide_kernel, dim3((R + 63) / 64), dim3(64), 0, h->stream, R, h->r_begin, h->seed, attempt, n_groups,
                       h->d_baro_U0, h->d_potential, h->d_labels, h->d_beta, h->d_pressure, h->d_econst, h->econst_vref, h->d_box, h->d_box_old, h->d_baro,
                       h->d_baro_acc, h->d_noise_id, (float)(2.0 * std::max(h->cutoff, h->coulomb_cutoff)), h->d_sync + 2);
    hipLaunchKernelGGL(baro_restore_kernel, dim3((h->N + 255) / 256, R), dim3(256), 0, h->stream, h->N, Npad, h->d_baro_acc, h->d_pos,
                       h->d_baro_x0, h->d_force, h->d_baro_f0, h->d_potential, h->d_baro_U0);
    h->box_version++;
    h->forces_valid = true; h->force_zeroed = false;
    REMD_CHECK(h, hipGetLastError());
    return 0;
}

