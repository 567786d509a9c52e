// The pair and exception arithmetic of a NoCutoff NonbondedForce (nocutoff.hip), shared with the resident small-molecule kernel of
// integrate.hip: both sum an atom's pair forces in ascending partner order in fp32 and convert once to fixed point, so the two paths give
// the same bits.
#pragma once
#include "remd_internal.h"

// param: q sqrt(k_e), sigma / 2, 2 sqrt(eps), -.  U = eps4 s6 (s6 - 1) + qq / r;  fr = dU/dr / r, F_i += fr (x_j - x_i)
template <bool ENERGY>
__device__ __forceinline__ void nocutoff_pair(const float4 xi, const float4 pi, const float4 xj, const float4 pj,
                                              float& fx, float& fy, float& fz, double& e)
{
    const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
    const float r2 = dx * dx + dy * dy + dz * dz;
    const float inv_r = rsqrtf(r2), inv_r2 = inv_r * inv_r;
    const float sig = pi.y + pj.y, eps4 = pi.z * pj.z, qq = pi.x * pj.x;
    const float s2 = sig * sig * inv_r2, s6 = s2 * s2 * s2;
    const float fr = eps4 * s6 * (6.f - 12.f * s6) * inv_r2 - qq * inv_r * inv_r2;
    fx += fr * dx; fy += fr * dy; fz += fr * dz;
    if (ENERGY) e += 0.5 * ((double)(eps4 * s6 * (s6 - 1.f)) + (double)(qq * inv_r));
}

// an exception (k_e qq, sigma, 4 eps): plain Coulomb + Lennard-Jones of its own parameters; d = x_j - x_i; returns dU/dr / r
template <bool ENERGY>
__device__ __forceinline__ float nocutoff_exception(const float4 par, const float3 d, double& e)
{
    const float r2 = d.x * d.x + d.y * d.y + d.z * d.z, inv_r = rsqrtf(r2), inv_r2 = inv_r * inv_r;
    const float s2 = par.y * par.y * inv_r2, s6 = s2 * s2 * s2;
    const float fr = par.z * s6 * (6.f - 12.f * s6) * inv_r2 - par.x * inv_r * inv_r2;
    if (ENERGY) e += (double)(par.z * s6 * (s6 - 1.f)) + (double)(par.x * inv_r);
    return fr;
}
