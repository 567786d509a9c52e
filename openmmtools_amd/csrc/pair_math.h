// Pair arithmetic of the direct-space nonbonded kernels (forces.hip) and of the resident small-system kernel (integrate.hip):
// one definition, so that both paths evaluate a pair with the same instructions.
// Reference semantics: OpenMM NonbondedForce as the test systems configure it (testsystems.py:1978-2000, 3504-3517) and the
// alchemical soft-core of alchemy.py:1383-1388; f64 restatement: oracle/forcefield.py.
#pragma once
#include <hip/hip_runtime.h>
#include "coulomb_table.h"

#define NB_LJ_ONLY 0
#define NB_RF      1
#define NB_EWALD   2
#define NB_EWALD_NOLJ 3   // Coulomb only: the LJ-active atoms are handled by a second, much shorter cluster list
#define NB_RF_NOLJ    4


struct nb_params {
    float rc, rc2, rs, inv_sw;        // cutoff, cutoff^2, switching distance (<0: none), 1/(rc-rs)
    float krf, crf;                   // reaction field
    float alpha, two_alpha_sqrtpi;    // Ewald
    int excl_words;                   // 64-bit words of the exclusion window per atom
    int n_jsplit;
    // Ewald split chosen by the host (remd_set_coulomb_cutoff): the erfc tail is summed to rcc >= rc while the Lennard-Jones
    // terms keep the NonbondedForce cutoff rc and its switch; rcc2 = rc2 unless the host asked for a longer Coulomb range
    float rcc2;
    // force-only Coulomb kernel from a table in r^2 (coulomb_table.h): first bin's key, number of bins, clamp of r^2
    int ctab_key0, ctab_n; float ctab_umin;
    int prio;                         // pair kernel at raised wave priority (the direct-space stream is the critical path)
    // reaction field as the reference's alchemical factory re-writes it (remd_set_reaction_field; forcefactories.py:76-84, forces.py:1110-1150):
    // c_rf = 0 and the pair term switched from rs_c to the cutoff; rs_c < 0: OpenMM's shifted reaction field
    float rs_c, inv_sw_c;
};


// ---- nonbonded pair arithmetic -------------------------------------------------------------------
__device__ __forceinline__ void switch_fn(const nb_params& p, float r, float& U, float& dUdr)
{
    if (p.rs >= 0.f && r > p.rs) {
        const float x = (r - p.rs) * p.inv_sw;
        const float S = 1.f + x * x * x * (-10.f + x * (15.f - 6.f * x));
        const float dS = x * x * (-30.f + x * (60.f - 30.f * x)) * p.inv_sw;
        dUdr = S * dUdr + U * dS;
        U *= S;
    }
}

// returns energy, writes dU/dr / r  (so that F_i = fr * (xj - xi))
// FAST_ERFC (force-only evaluations): the Ewald direct-space force comes from the cubic table `ctab` (LDS, coulomb_table.h)
// when one is given (TABLE; `ctab` = the table's LDS address minus its first key, so that the key of r^2 indexes it directly),
// from the Abramowitz & Stegun erfc otherwise; energy evaluations use erfcf.
template <int METHOD, bool ALCH, bool FAST_ERFC, bool TABLE = false>
__device__ __forceinline__ float pair_interaction(const nb_params& p, float r2, float4 pi, float4 pj,
                                                  float lam_a, float sc, float& fr, bool energy_skip_na, float& e_out,
                                                  const float4* ctab = nullptr)
{
    // hardware v_rsq_f32 / v_rcp_f32 (1 ulp) instead of the libm wrappers: r2 is never denormal here and the
    // denormal/IEEE-division guards cost a sixth of this VALU-bound loop
    const float inv_r = __builtin_amdgcn_rsqf(r2);
    const float r = r2 * inv_r;
    const float sig = pi.y + pj.y, eps4 = pi.z * pj.z;
    float U = 0.f, dUdr = 0.f;
    bool na = false;
    // (an unsplit Ewald kernel that sums the Coulomb tail beyond the Lennard-Jones cutoff: LJ stops at rc)
    if (METHOD <= NB_EWALD && eps4 != 0.f && (METHOD != NB_EWALD || r2 < p.rc2)) {
        // (param.w: 0 non-alchemical, 1 alchemical, 2 alchemical with annihilate_sterics -- then alchemical/alchemical pairs are
        // lambda-controlled too, alchemy.py:1767-1779)
        if (ALCH && ((pi.w != pj.w) || pi.w > 1.5f)) {
            // soft-core (alchemy.py:1383-1388 with softcore_c = 6): x = 1/(alpha(1-l)^b + (r/sigma)^6)
            na = true;
            const float is2 = 1.f / (sig * sig);
            const float t = r2 * r2 * r2 * is2 * is2 * is2;
            const float x = 1.f / (sc + t);
            U = lam_a * eps4 * x * (x - 1.f);
            dUdr = lam_a * eps4 * (2.f * x - 1.f) * (-x * x * 6.f * t * inv_r);
        } else {
            const float s2 = sig * sig * inv_r * inv_r;
            const float s6 = s2 * s2 * s2;
            U = eps4 * s6 * (s6 - 1.f);
            dUdr = eps4 * s6 * (6.f - 12.f * s6) * inv_r;
        }
        switch_fn(p, r, U, dUdr);
    }
    float Uc = 0.f, dUc = 0.f, frc = 0.f;
    if (METHOD != NB_LJ_ONLY) {
        const float qq = pi.x * pj.x;
        if ((METHOD == NB_EWALD || METHOD == NB_EWALD_NOLJ) && TABLE) {
            // coulomb_table.h: key = exponent and leading mantissa bits of r^2, cubic in the remaining mantissa bits;
            // the table holds -G, so that frc = dUc/dr / r
            const unsigned int bits = __float_as_uint(r2);
            const float tf = (float)(bits & CTAB_MASK);
            const float4 c = ctab[bits >> CTAB_SHIFT];      // (ctab is biased by the first bin's key: one shift-add per address)
            frc = qq * fmaf(tf, fmaf(tf, fmaf(tf, c.w, c.z), c.y), c.x);
        } else if (METHOD == NB_EWALD || METHOD == NB_EWALD_NOLJ) {
            const float ar = p.alpha * r;
            const float ex = __expf(-ar * ar);
            float erfc_ar;
            if (FAST_ERFC) {
                // Abramowitz & Stegun 7.1.26 (|abs err| < 1.5e-7): force-only evaluations; energies use erfcf
                const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ar);
                erfc_ar = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f)))) * ex;
            } else {
                erfc_ar = erfcf(ar);
            }
            Uc = qq * erfc_ar * inv_r;
            dUc = -qq * (erfc_ar * inv_r + p.two_alpha_sqrtpi * ex) * inv_r;
        } else {
            Uc = qq * (inv_r + p.krf * r2 - p.crf);
            dUc = qq * (2.f * p.krf * r - inv_r * inv_r);
            if (p.rs_c >= 0.f && r > p.rs_c) {
                const float x = (r - p.rs_c) * p.inv_sw_c;
                const float S = 1.f + x * x * x * (-10.f + x * (15.f - 6.f * x));
                dUc = S * dUc + Uc * (x * x * (-30.f + x * (60.f - 30.f * x)) * p.inv_sw_c);
                Uc *= S;
            }
        }
    }
    // (x + 0.f cannot be folded without nsz: name the sum the variant has)
    if (TABLE && (METHOD == NB_EWALD || METHOD == NB_EWALD_NOLJ)) fr = (METHOD == NB_EWALD_NOLJ) ? frc : dUdr * inv_r + frc;
    else fr = (METHOD == NB_LJ_ONLY ? dUdr : METHOD > NB_EWALD ? dUc : dUdr + dUc) * inv_r;
    e_out = ((ALCH && na && energy_skip_na) ? 0.f : U) + Uc;
    return e_out;
}

