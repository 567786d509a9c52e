/*
 * remd_hip.h — C ABI of libremd_hip.so, the MI355X (gfx950) replica-exchange engine.
 *
 * This is the drop-in boundary for ONE hot path of choderalab/openmmtools: the
 * replica-exchange iteration  mix -> propagate -> u_kl
 * (reference: openmmtools/multistate/multistatesampler.py:766-804).
 * The reference has no FFI; its narrowest seam is the three overridable hooks
 *   MultiStateSampler._mix_replicas        (multistatesampler.py:1500, replicaexchange.py:255, sams.py:395)
 *   MultiStateSampler._propagate_replicas  (multistatesampler.py:1287)
 *   MultiStateSampler._compute_energies    (multistatesampler.py:1436)
 * plus the ContextCache attributes (multistatesampler.py:1755-1764).  Each entry point
 * below names the reference interface it replaces.  See INTEGRATION.md for the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; remd_last_error() gives text
 *   - plain pointers and sizes only; host pointers unless the name starts with d_
 *   - units: nm, ps, amu, kJ/mol, elementary charge (OpenMM's md_unit_system)
 *   - one handle per GPU / rank; a handle is not thread-safe, distinct handles are
 *   - all kernels are launched on the stream given at creation (NULL = a private non-blocking stream owned by the
 *     handle; entry points that return results synchronise it before they return)
 *   - random numbers: counter-based Philox4x32-10, key = seed, counters documented
 *     per entry point (DESIGN.md "RNG stream spec"); results do not depend on how
 *     replicas are sharded over ranks
 */
#ifndef REMD_HIP_H
#define REMD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct remd_ctx* remd_handle;

/* nonbonded methods (mirror openmm.NonbondedForce enum subset the configs use) */
#define REMD_NB_NONE            0
#define REMD_NB_CUTOFF_PERIODIC 1   /* reaction field */
#define REMD_NB_PME             2
#define REMD_NB_NOCUTOFF        3   /* every pair, no box: the vacuum test systems (csrc/nocutoff.hip); cutoff, switch, dispersion correction unused */

/* mixing schemes (replicaexchange.py:265-283, sams.py:410-417) */
#define REMD_MIX_NONE           0
#define REMD_MIX_SWAP_ALL       1
#define REMD_MIX_SWAP_NEIGHBORS 2
#define REMD_MIX_SAMS_GLOBAL    3

/*
 * Flat description of an openmm.System restricted to the force classes the five
 * benchmark test systems use (testsystems.py:685-840, 1872-2030, 3465-3527,
 * 3789-3857, 3863-3923).  All arrays are host memory, read during remd_set_system.
 */
typedef struct remd_system_desc {
    int32_t n_atoms;
    const double* mass;             /* [N] amu                                           */

    /* CustomExternalForce of testsystems.HarmonicOscillator (testsystems.py:779-786):
       U = K/2 ((x-x0)^2 + y^2 + z^2) + U0 on the listed atoms                           */
    int32_t n_ext;
    const int32_t* ext_atoms;       /* [n_ext]                                           */
    double ext_K, ext_x0, ext_U0;

    /* HarmonicBondForce / HarmonicAngleForce / PeriodicTorsionForce                     */
    int32_t n_bonds;    const int32_t* bond_atoms;    const double* bond_params;    /* [n][2]; [n][2] = r0, k            */
    int32_t n_angles;   const int32_t* angle_atoms;   const double* angle_params;   /* [n][3]; [n][2] = theta0, k        */
    int32_t n_torsions; const int32_t* torsion_atoms; const double* torsion_params; /* [n][4]; [n][3] = n, phase, k      */

    /* NonbondedForce                                                                     */
    int32_t nb_method;              /* REMD_NB_*                                         */
    double cutoff;                  /* nm                                                */
    double switch_distance;         /* nm, <= 0: no switching function                   */
    double rf_dielectric;           /* reaction-field solvent dielectric (78.3)          */
    double ewald_alpha;             /* 1/nm (PME)                                        */
    int32_t pme_grid[3];            /* PME mesh (each a product of 2,3,5; <= 256)        */
    int32_t use_dispersion_correction;
    const double* charge;           /* [N] e                                             */
    const double* sigma;            /* [N] nm                                            */
    const double* epsilon;          /* [N] kJ/mol                                        */
    int32_t n_exceptions;
    const int32_t* exception_atoms; /* [n][2]                                            */
    const double* exception_params; /* [n][3] = chargeProd, sigma, epsilon               */

    /* constraints: rigid 3-site waters (SETTLE) and X-H clusters (SHAKE)                */
    int32_t n_settle;
    const int32_t* settle_atoms;    /* [n][3] = O, H1, H2                                */
    double settle_dOH, settle_dHH;  /* nm                                                */
    int32_t n_shake;
    const int32_t* shake_atoms;     /* [n][4] = heavy, h1, h2|-1, h3|-1                  */
    const double* shake_dist;       /* [n][3] nm                                         */

    int32_t cmm_frequency;          /* CMMotionRemover frequency in steps, 0 = absent    */

    /* AbsoluteAlchemicalFactory region (alchemy.py:417-427, 1356-1390): soft-core LJ on
       alchemical/non-alchemical pairs; alchemical atoms must be uncharged or use PME
       'exact' treatment (charge scaling by lambda_electrostatics)                       */
    int32_t n_alch;
    const int32_t* alch_atoms;      /* [n_alch]                                          */
    double softcore_alpha, softcore_a, softcore_b, softcore_c;
} remd_system_desc;

/* ---- lifetime -------------------------------------------------------------------- */
/* Replaces cache.ContextCache construction (cache.py:313-346): one batched device-state
   pool per GPU.  stream: a hipStream_t (as void*; must be capturable, i.e. not the legacy default stream) or NULL.                              */
int  remd_create(remd_handle* out, int device, void* stream);
int  remd_destroy(remd_handle h);
const char* remd_last_error(remd_handle h);   /* h may be NULL: last global error         */
int  remd_version(void);

/* ---- set-up (reference: MultiStateSampler._pre_write_create, multistatesampler.py:836-926) */
int  remd_set_system(remd_handle h, const remd_system_desc* desc);

/* Ewald split of a PME system, to be called BEFORE remd_set_system (0 = none, the default): the direct-space erfc sum runs to
   coulomb_cutoff_nm >= desc->cutoff while every Lennard-Jones term keeps desc->cutoff and its switching function, and the
   host passes the ewald_alpha / pme_grid that belong to that range in the descriptor (OpenMM's rule applied to the Coulomb
   range: alpha = sqrt(-ln 2 tol) / r, mesh >= 2 alpha L / (3 tol^(1/5)); reference: the formula quoted at
   alchemy/alchemy.py:1528-1532, openmmtools_amd/system.py:ewald_parameters).  The Ewald sum does not depend on where it is
   split, so potentials and forces agree with the reference split to the Ewald error tolerance on both sides; what changes is
   how much work the pair kernel and the mesh get (a longer range buys a smaller mesh).  Ignored by non-PME methods.        */
int  remd_set_coulomb_cutoff(remd_handle h, double coulomb_cutoff_nm);

/* Reaction field of a CutoffPeriodic system, to be called BEFORE remd_set_system.  unshifted = 0 (default): OpenMM's NonbondedForce,
   qq (1/r + k_rf r^2 - c_rf).  unshifted = 1: what the reference's alchemical factory turns the WHOLE system's reaction field into under
   its default alchemical_rf_treatment='switched' (alchemy.py:744-749 -> forcefactories.replace_reaction_field :76-84 ->
   forces.UnshiftedReactionFieldForce, forces.py:1110-1150): qq (1/r + k_rf r^2), c_rf = 0, times OpenMM's switching function from
   cutoff - switch_width_nm to the cutoff (0: truncated); exceptions keep their plain qq / r.  Ignored by PME systems.                */
int  remd_set_reaction_field(remd_handle h, int unshifted, double switch_width_nm);

/* K thermodynamic states: beta [1/(kJ/mol)], lambda_sterics, lambda_electrostatics, and an
   additive potential-energy constant per state in kJ/mol (e.g. the lambda-dependent
   long-range correction of the alchemical CustomNonbondedForce).  Arrays may be NULL
   (lambda = 1, const = 0).  Reference: states.py:1908-1917 (reduced potential),
   paralleltempering.py:206-215, states.py:911-992.                                      */
int  remd_set_states(remd_handle h, int K, const double* beta,
                     const double* lambda_sterics, const double* lambda_electrostatics,
                     const double* energy_const);

/* LangevinIntegrator (integrators.py:1071-1158) as used by
   mcmc.LangevinSplittingDynamicsMove (mcmc.py:1280-1316): splitting string of V/R/O tokens.
   constraint_tolerance (reference default 1e-8, integrators.py:1073): a relative distance error, as OpenMM's.  Rigid waters are
   solved analytically (SETTLE).  X-H star clusters (<= 3 hydrogens on one heavy atom): positions by Newton iterations on the K x K
   system of Lagrange multipliers until every bond is within max(constraint_tolerance, 2e-7) -- fp32 coordinates relative to the
   cluster's central atom hold no more -- with at most 8 updates (remd_get_constraint_stats reports what was needed; quadratic
   convergence from the unconstrained step: two or three); velocities by one exact K x K linear solve.  A request below 2e-7 is
   served at 2e-7 and the host classes say so once.  tests/test_forcefield_parity.py
   (test_constraint_tolerance_sets_the_newton_iterations_of_the_xh_solve, test_alanine_constraints_and_substeps) hold it.
   libremd_cpu.so (f64) iterates to the tolerance itself.                                                                        */
int  remd_set_integrator(remd_handle h, const char* splitting, double timestep_ps,
                         double collision_rate_invps, int n_steps,
                         int reassign_velocities, double constraint_tolerance);

/* Heat, shadow work and Metropolization of LangevinIntegrator (integrators.py:1077-1125 constructor flags, :1175-1204 globals,
   :1404-1460 substeps, :1539-1557 Metropolization; mcmc.py:1282-1316 passes the flags through the move).
     heat         = sum over the O substeps of the change of the kinetic energy (:1448-1460), kJ/mol, per replica;
     shadow work  = sum over the V substeps of the change of the kinetic energy (:1433-1446) and over the R substeps of the
                    change of kinetic + potential energy (:1404-1423), kJ/mol, per replica;
     a splitting string may hold "{" ... "}" around V / R substeps ("O { V R V } O"): at "}" every replica accepts the substeps
     since "{" with probability min(1, exp(-shadow_work / kT)) (uniform from the Philox stream 7: (index of the "}" in the
     string, global replica, global step)), on rejection x = x_old, v = -v_old; the shadow work restarts from 0 (:1544-1557).
     A Metropolized string measures shadow work whatever the flag says (:1117-1119).
   Measuring shadow work needs the potential energy before and after every R substep: two energy evaluations per "V R O R V"
   step instead of one force evaluation (the reference pays the same through CustomIntegrator's `energy`).  Values accumulate
   over remd_propagate / remd_step calls until remd_reset_work (LangevinIntegrator.reset, :1213-1218).
   remd_get_work: any pointer may be NULL; arrays [R_local]; n_accepted / n_trials count the "}" decisions (:1286-1297).   */
int  remd_set_work_measurement(remd_handle h, int measure_heat, int measure_shadow_work);
int  remd_get_work(remd_handle h, double* heat, double* shadow_work, int64_t* n_accepted, int64_t* n_trials);
int  remd_reset_work(remd_handle h);

/* Multiple-time-step splittings of LangevinIntegrator ("V0 V1 R R O R R V1 R R O R R V1 V0", integrators.py:1036-1053):
   a V<g> substep kicks with the forces of force group g only and dt / (number of V<g> tokens) (:1437-1438); with one
   distinct group, or none, every V uses all forces (:1440).  groups[6] = force group (Force.getForceGroup(); for the
   reciprocal part NonbondedForce.getReciprocalSpaceForceGroup(), -1 there meaning "as the direct part") of, in order:
   the external force, bonds, angles, torsions, NonbondedForce direct space (with its exceptions and the Ewald exclusion
   correction), PME reciprocal space.  Default: all 0.  Groups 0 ... 3 may appear in a multiple-time-step splitting;
   every force class must then sit in a group the splitting names.  libremd_cpu.so stores the groups and refuses
   multiple-time-step splittings (-3).                                                                              */
int  remd_set_force_groups(remd_handle h, const int32_t* groups);

/* BaseIntegratorMove.n_restart_attempts (mcmc.py:668-776, retry loop :706-759): when a replica
   holds a NaN after remd_propagate's MD steps, its pre-propagate positions/velocities are
   restored and the move is repeated (fresh noise: the Philox counters carry the attempt
   number) up to n more times; replicas that were fine keep the result of the attempt in which
   they first succeeded.  nan_flags report the replicas that failed every attempt.  Default 0. */
int  remd_set_restart_attempts(remd_handle h, int n_restart_attempts);

/* NPT states.  The reference's ThermodynamicState with a pressure adds an openmm.MonteCarloBarostat (frequency 25) to the
   System (states.py:1177-1181, 893-898); it fires inside the integrator's updateContextState step
   (integrators.py:1313).  pressure: [K] per state in kJ/mol/nm^3 (bar x 0.06022140857), or NULL / frequency 0 to switch
   the barostat off.  The volume moves follow OpenMM's MonteCarloBarostatImpl (molecule-centre scaling, adaptive volume
   step); u_kl gains beta_l p_l V_r (states.py:1913-1914).  Boxes change: read them back with remd_get_boxes.          */
int  remd_set_barostat(remd_handle h, int K, const double* pressure, int frequency);
int  remd_get_boxes(remd_handle h, double* box /*[R_local][3]*/);
/* The per-state energy constants of remd_set_states are long-range corrections ~ 1/V (alchemy.py:1786-1789); with a
   barostat they must follow the box: pass the volume (nm^3) they were evaluated at, 0 = volume independent (default). */
int  remd_set_energy_const_volume(remd_handle h, double reference_volume);
int  remd_get_barostat_stats(remd_handle h, double* volume_scale /*[R_local]*/, int64_t* n_attempted, int64_t* n_accepted);
/* mcmc.py:1597-1700 MonteCarloBarostatMove: n_attempts volume moves of every local replica outside the integrator (the
   reference runs a DummyIntegrator for n_attempts steps with the barostat's frequency temporarily set to 1).  Same
   move, same random stream and attempt counter as the in-integrator barostat; velocities are not touched.  Needs
   remd_set_barostat (returns -3 otherwise).  Returns after the moves have completed.                                  */
int  remd_barostat_attempts(remd_handle h, int n_attempts);

/* MultiStateSampler.minimize (multistatesampler.py:611-647; _minimize_replica :1351-1434) with the reference's
   FIREMinimizationIntegrator (integrators.py:2290-2469, default parameters: timestep 1 fs, alpha 0.1, dt_max 10 fs,
   f_inc 1.1, f_dec 0.5, f_alpha 0.99, N_min 5): every local replica is minimised at its current state's Hamiltonian.
   tolerance: kJ/mol/nm (converged when |f| / n_dof <= tolerance, as the reference tests it); max_iterations = 0: until
   every replica has converged (polled every 50 steps), else exactly that many FIRE steps.  Velocities are zeroed first.
   converged: [R_local] flags (may be NULL); n_iterations: steps taken (may be NULL).                                  */
int  remd_minimize(remd_handle h, double tolerance_kj_per_mol_nm, int max_iterations,
                   int32_t* converged, int32_t* n_iterations);

/* Replicas r_begin .. r_begin+R_local-1 of R_global live on this handle.
   x, v: [R_local][N][3] (v may be NULL -> zero; x may be NULL when the coordinates follow through
   remd_copy_replicas: the call then only sizes the handle); box: [R_local][3] orthorhombic edge
   lengths; labels: [R_global] state index of every replica.
   Reference: SamplerState.apply_to_context (states.py:2257-2279).                       */
int  remd_set_replicas(remd_handle h, int R_global, int r_begin, int R_local,
                       const double* x, const double* v, const double* box,
                       const int64_t* labels);

/* What keys the random streams of the local replicas (velocity reassignment, Ornstein-Uhlenbeck noise, Metropolis and barostat
   draws): by default the global replica index r_begin + r of remd_set_replicas, which makes a trajectory independent of how the
   ensemble is sharded.  A host that gives one handle a NON-CONTIGUOUS subset of the ensemble (one handle per compatibility group of
   states: states.py:186-217, multistatesampler.py:1296-1320 propagates a replica in the Context of its own state's System) passes the
   subset's global indices here, after remd_set_replicas (which resets them); NULL restores the default.                       */
int  remd_set_replica_ids(remd_handle h, const int64_t* global_replica_index /* [R_local] or NULL */);

/* Options of the alchemical region that the descriptor does not carry; call before remd_set_system.  annihilate_sterics
   (AlchemicalRegion.annihilate_sterics, alchemy.py:421, 1767-1779, 1841-1846): the Lennard-Jones interactions BETWEEN alchemical
   atoms (pairs and exceptions) are soft-core and lambda_sterics-controlled like those with the environment; 0 (default): they
   stay at full strength ("decoupling").                                                                                       */
int  remd_set_alchemical_options(remd_handle h, int annihilate_sterics);

/* General alchemical regions: what AbsoluteAlchemicalFactory._alchemically_modify_NonbondedForce (alchemy.py:1539-2038) builds when
   the descriptor's one-region / exact-PME fast path (n_alch, softcore_*) does not cover the request -- several named regions with
   their own lambda_sterics_<name> / lambda_electrostatics_<name> (alchemy.py:1360-1377, 1412-1421), pairs of regions that interact
   through the PRODUCT of their lambdas (alchemical_regions_interactions, :1684-1690, 1766-1770), soft-core electrostatics of the
   'direct-space' and 'coulomb' PME treatments and of the reaction-field treatments (:1392-1537: softcore_beta, d, e, f), any
   softcore_a / b / c, annihilate_sterics / annihilate_electrostatics per region (:1771-1775, 1800-1803).
   The force split is the reference's: the descriptor of remd_set_system then describes the NonbondedForce the factory leaves behind
   (alchemical atoms with charge 0 and epsilon 0, exceptions that touch them zeroed but kept as exclusions: :1903-1911, 2001-2006;
   n_alch = 0) and THIS call adds the custom forces, evaluated by one launch over the (alchemical atom, atom) pairs per force
   evaluation (csrc/alch_regions.hip):
     pairs inside the cutoff, not excluded, classes (environment, region y), (y, y) and (a, b) for interacting regions:
       U_sterics        = l^a 4 eps x (x - 1),  x = (sigma / reff)^6,  reff = sigma (alpha (1 - l)^b + (r / sigma)^c)^(1/c)        :1383-1388
       U_electrostatics = l^d k_e q1 q2 g(reff_e),  reff_e = sigma (beta (1 - l)^e + (r / sigma)^f)^(1/f)                        :1425-1430
       g(x) = erfc(elec_alpha x) / x + elec_krf x^2 - elec_crf                                                               :1434, 1505-1507, 1534-1536
       sterics switched as the NonbondedForce (desc->switch_distance), electrostatics from elec_switch_distance (< 0: not switched) :1780-1782, 1818-1824
     l = the region's lambda for (environment, y); for (y, y) the region's lambda if it annihilates, else 1; the product for (a, b);
     the soft-core constants of a class are those of region y (of b for an interacting pair, as the factory's loop leaves them, :2009-2017);
     exceptions with an alchemical atom: the same sterics without cutoff or switch, electrostatics l^d k_e qq / reff_e            :1374-1380, 1434, 1456-1461
       (between atoms of two regions: a bond of the lower region's (environment, region) force, as the factory's loop leaves it, :1972-2006)
   charge / sigma / epsilon and the exceptions passed here are the REFERENCE NonbondedForce's (sigma = 0 already replaced by
   0.1 nm, :1638-1661).  electrostatics = 0: no electrostatic custom forces (alchemical atoms without charge).
   exact_pme = 1 (PME systems; alchemy.py:1663-1681, 1893-1899, 1978-1982: the factory's default treatment): no electrostatic custom
   forces -- the charge of an atom of region x enters the WHOLE Ewald sum as lambda_electrostatics_x q (the NonbondedForce's parameter
   offsets), a charged exception that touches region x counts lambda_electrostatics_x qq (the offset of the first region that meets it),
   and every pair of atoms of two regions that do NOT interact is excluded (:1663-1672).  The descriptor of remd_set_system then has the
   alchemical charges at 0 as the factory leaves them, `charge` / the exceptions here carry the offsets.  Device: the pair kernels see the
   environment's charges only; the alchemical atoms' direct-space terms, the Ewald corrections of their excluded pairs and the exceptions
   come from the custom-forces launch (erfc to the Coulomb range of the handle's Ewald split), the mesh takes the scaled charges; u_kl
   from (n + 1)(n + 2) / 2 energy passes: the Coulomb energy is a quadratic form in the regions' lambda_electrostatics.
   Sterics as above (of a pair of interacting regions: none, as the reference's loop leaves them -- tables zeroed, :1886-1911).
   Call AFTER remd_set_system (which forgets the regions) and follow remd_set_states by remd_set_region_lambdas.  n_regions = 0 or
   desc = NULL: none.  NoCutoff systems: no box, no cutoff, no switch (the
   custom forces copy the NonbondedForce's method, :1793-1796).                   */
typedef struct remd_alch_regions_desc {
    int32_t n_atoms;                     /* the system's                                                     */
    int32_t n_regions;
    const int32_t* region_of_atom;       /* [n_atoms] 0 = environment, g = region g (1-based)               */
    const double*  softcore;             /* [n_regions][8] alpha, beta, a, b, c, d, e, f                      */
    const int32_t* annihilate;           /* [n_regions][2] sterics, electrostatics                           */
    int32_t n_interactions;
    const int32_t* interactions;         /* [n_interactions][2] 1-based region indices (a, b)               */
    const double *charge, *sigma, *epsilon;      /* [n_atoms] reference parameters (e, nm, kJ/mol)          */
    int32_t n_exceptions;                /* the reference exceptions that touch an alchemical atom           */
    const int32_t* exception_atoms;      /* [n_exceptions][2]                                                */
    const double*  exception_params;     /* [n_exceptions][3] chargeprod, sigma, epsilon                     */
    int32_t electrostatics;              /* 0 / 1                                                            */
    double elec_alpha, elec_krf, elec_crf, elec_switch_distance;
    int32_t exact_pme;                   /* 1: the exact PME treatment (below); electrostatics / elec_* unused */
    /* alchemically softened bonded terms (alchemy.py:1115-1354): U = lambda_{bonds, angles, torsions}_<region> x the reference's harmonic
       bond (K/2)(r - r0)^2, harmonic angle (K/2)(theta - theta0)^2, periodic torsion k (1 + cos(n phi - phase)); the host takes these terms
       OUT of the descriptor of remd_set_system (and, as the factory does once any term of a class is alchemical, the terms of that class
       that connect two regions which do not interact, :1156-1162).  remd_set_region_bonded_lambdas gives their lambdas (default 1).     */
    int32_t n_bonds;    const int32_t* bond_atoms;    const double* bond_params;    const int32_t* bond_region;      /* [n][2]; [n][2] = r0, K; [n] 1-based */
    int32_t n_angles;   const int32_t* angle_atoms;   const double* angle_params;   const int32_t* angle_region;     /* [n][3]; [n][2] = theta0, K           */
    int32_t n_torsions; const int32_t* torsion_atoms; const double* torsion_params; const int32_t* torsion_region;   /* [n][4]; [n][3] = n, phase, k         */
    int32_t consistent_exceptions;       /* AbsoluteAlchemicalFactory(consistent_exceptions=True), alchemy.py:1456-1461: the electrostatics of the
                                            exceptions use g of the pairs (elec_alpha / elec_krf / elec_crf; no cutoff, no switch) instead of 1 / reff */
} remd_alch_regions_desc;
int  remd_set_alchemical_regions(remd_handle h, const remd_alch_regions_desc* desc);
/* GBSA implicit solvent of a NoCutoff system: OpenMM's GBSAOBCForce (OBC2 Born radii, ACE surface term) in the form the reference's
   alchemical factory gives it (alchemy.py:2144-2225, _alchemically_modify_GBSAOBCForce -- a CustomGBForce whose expression strings are the
   definition followed here; offset 0.009 nm, tanh(psi - 0.8 psi^2 + 4.85 psi^3), k_e = 138.935485, surface term 28.3919551 (R + 0.14)^2 (R / B)^6):
     I_i = sum_{j != i} s_j H(r_ij; R_i - offset, scale_j (R_j - offset)),   B_i = 1 / (1 / (R_i - offset) - tanh(...) / R_i),
     E = sum_i s_i [-k_e tau q_i^2 / (2 B_i) + surface_i] - sum_{i<j} k_e tau s_i q_i s_j q_j / sqrt(r^2 + B_i B_j exp(-r^2 / (4 B_i B_j))),
   tau = 1 / solute_dielectric - 1 / solvent_dielectric, s_i = lambda_electrostatics on the alchemical particles (of the replica's state: region 1
   of remd_set_region_lambdas -- the factory supports one region with GBSA, :2168-2171) and 1 elsewhere.  Call AFTER remd_set_system
   (which forgets it); desc = NULL: none.  u_kl re-evaluates the GB energy at every state's lambda (it is not a polynomial in lambda). */
typedef struct remd_gbsa_desc {
    int32_t n_atoms;
    const double *charge, *radius, *scale;     /* [n_atoms] GBSAOBCForce particle parameters: e, nm, -            */
    const int32_t* alchemical;                 /* [n_atoms] 0 / 1, or NULL (no alchemical particle)                */
    double solute_dielectric, solvent_dielectric;
    int32_t surface_area;                      /* 1: with the ACE surface term (GBSAOBCForce's default)            */
} remd_gbsa_desc;
int  remd_set_gbsa(remd_handle h, const remd_gbsa_desc* desc);

/* lambda_sterics / lambda_electrostatics of every region at every state: [K][n_regions], K as in remd_set_states (call after it).  The
   energy_const of remd_set_states carries the long-range corrections of the sterics custom forces (alchemy.py:1786-1789).      */
int  remd_set_region_lambdas(remd_handle h, int K, int n_regions, const double* lambda_sterics, const double* lambda_electrostatics);
/* lambda_bonds / lambda_angles / lambda_torsions of every region at every state, [K][n_regions] each (NULL: 1), after
   remd_set_region_lambdas.  AlchemicalState.lambda_bonds ... (alchemy.py:196-199).                                               */
int  remd_set_region_bonded_lambdas(remd_handle h, int K, int n_regions, const double* lambda_bonds, const double* lambda_angles,
                                    const double* lambda_torsions);

/* Device-to-device transfer of replicas between two handles on the same device that hold the same particles (one handle per
   compatibility group of states: the reference propagates a replica in the Context of its own state's System,
   multistatesampler.py:1296-1320, and evaluates every configuration in one Context per group, :1470-1490 -- the coordinates it moves
   between Contexts through SamplerState.apply_to_context, states.py:2257-2279, stay on the device here): positions, velocities
   and box edges of src's local replicas src_slot[k] replace those of dst's local replicas dst_slot[k], k < n.  what: bit 0
   positions, bit 1 velocities, bit 2 boxes.  dst is sized by remd_set_replicas first (x = NULL there: "coordinates follow").
   Returns after the copy is complete; dst re-sorts its molecules at the next force evaluation. */
int  remd_copy_replicas(remd_handle dst, const int32_t* dst_slot, remd_handle src, const int32_t* src_slot, int32_t n, int32_t what);

int  remd_set_labels(remd_handle h, const int64_t* labels /*[R_global]*/);
int  remd_seed(remd_handle h, uint64_t seed);

/* ---- the hot path ------------------------------------------------------------------ */
/* Replaces MultiStateSampler._propagate_replicas (multistatesampler.py:1287-1337) and
   BaseIntegratorMove.apply (mcmc.py:668-776) for every local replica at once: optional
   Maxwell-Boltzmann velocity reassignment, n_steps of the splitting, NaN flag per replica.
   nan_flags: host [R_local] or NULL.                                                     */
/* Phases (round 6): remd_propagate of ONE handle as two blocks of its local replicas whose MD steps take turns on the device -- block
   A's step s, block B's step s, A's step s + 1, ... enqueued by the calling thread, block A on the handle's own pair of streams, block B
   on one more pair -- so that the integrator chain of one block (the serial part of an MD step: a few hundred wavefronts waiting for
   one dependent thing after another) runs beside the pair and mesh kernels of the other.  Replicas are independent between two mixes
   (multistatesampler.py:1296-1297; the reference propagates them one after the other or one per MPI rank), and every per-replica result
   is the one-block result bit for bit (fixed-point force sums, Philox streams keyed by the global replica, the same schedule of spatial
   re-sorts).  n: 0 = by rule (two blocks when the environment variable GPU_MAX_HW_QUEUES is 2 or 3, a propagation is 16 MD steps or more and the handle holds 6 replicas or
   more: HIP's default of 4 queues per priority gives the second block's main stream a FIFTH hardware queue, and queues beyond the four
   pipes of the chip are time-sliced -- two blocks then run 55 % slower than one instead of 10 % faster), 1 = one block, 2 = two blocks.
   Only PME systems under a plain V / R / O splitting without barostat, work measurement or Metropolization run as phases; everything
   else, and every other entry point, is unchanged.  remd_get_phases: the number of blocks the last remd_propagate ran as.            */
int  remd_set_phases(remd_handle h, int32_t n);
int  remd_get_phases(remd_handle h, int32_t* n);

/* Position constraints of X-H star clusters are solved by Newton iterations on the cluster's multipliers until every bond is within
   the integrator's constraint_tolerance (a relative distance error, as OpenMM's; integrators.py:1416-1418 addConstrainPositions), but no
   tighter than 2e-7 -- what fp32 coordinates relative to the cluster's central atom hold -- and with at most 8 updates.  Rigid waters
   are analytic (SETTLE), velocity constraints one exact K x K solve.  max_newton_iterations: the most updates any position solve of this
   handle has needed since remd_create; unconverged: 1 when a solve reached the bound without meeting the tolerance.                 */
int  remd_get_constraint_stats(remd_handle h, int32_t* max_newton_iterations, int32_t* unconverged);

int  remd_propagate(remd_handle h, int64_t iteration, int32_t* nan_flags);

/* remd_propagate for SEVERAL handles of one device in one call from one host thread (round 6): the handles' MD steps take turns, each
   handle on its own pair of streams, so that the integrator chain of one group of replicas -- the serial part of an MD step -- runs
   beside the pair and mesh kernels of another group (replicas are independent between two mixes, multistatesampler.py:1296-1297;
   the reference propagates them one after another or one per MPI rank).  Per-replica results are those of remd_propagate on each
   handle, bit for bit.  hs: n handles with the same n_steps; nan_flags: host, the handles' local replicas concatenated, or NULL.
   While the call runs no workgroup of these handles waits on a CU for another stream (the join of a step is a one-wavefront launch,
   the centre-of-mass momentum sum two launches): a polling integrator chain holds 320 registers per lane of every CU it sits on.
   A handle that reports a device-side fault, or a NaN with restart attempts left, is run again alone through remd_propagate.   */
int  remd_propagate_many(remd_handle* hs, int32_t n, int64_t iteration, int32_t* nan_flags);

/* Replaces MultiStateSampler._compute_energies / _compute_replica_energies
   (multistatesampler.py:1436-1494; paralleltempering.py:175-215): rows r_begin.. of the
   reduced-potential matrix.  d_ukl_rows: DEVICE [R_local][K] f64, or NULL to use the
   handle's own full [R_global][K] matrix (rows written in place).  ukl_host: host
   [R_local][K] copy or NULL.  potential_host: host [R_local] kJ/mol or NULL.            */
int  remd_compute_energies(remd_handle h, double* d_ukl_rows, double* ukl_host,
                           double* potential_host);

/* Device address of the handle's own [R_global][K] u_kl matrix (for RCCL all-gather
   through torch.distributed on an aliasing tensor).                                     */
int  remd_ukl_device_ptr(remd_handle h, double** d_ukl);

/* ---- sharding without a Python host: RCCL behind the C ABI --------------------------- */
/* The reference distributes replicas over MPI ranks with mpiplus (multistatesampler.py:1296-1311, 1448-1449: every rank
   propagates / evaluates its replicas, results are gathered) and broadcasts the mixed labels (replicaexchange.py:255).
   Here: one process per GPU, the host gives every rank a contiguous block of replicas (remd_set_replicas), and the ONE
   data-path collective -- every rank's rows of u_kl to every rank -- runs on RCCL over xGMI inside the library, on the
   handle's stream; remd_mix on the handle's own matrix is then the same deterministic computation on every rank, so no
   label broadcast is needed.  A torch.distributed host may instead gather through remd_ukl_device_ptr (multistate/comm.py).
     remd_comm_unique_id   rank 0: 128 opaque bytes (ncclUniqueId) the host passes to the other ranks by its own means
                           (MPI_Bcast, a file, a socket)
     remd_comm_init        collective over all `world` ranks: joins the communicator on the handle's device
     remd_comm_all_gather_energies
                           collective: after remd_compute_energies(h, NULL, ...) wrote the local rows into the handle's
                           own [R_global][K] matrix, fills in every other rank's rows (blocks may differ in size; they
                           must tile 0..R_global-1 in rank order).  Asynchronous on the handle's stream; world 1 or an
                           unsharded handle: nothing to do.
     remd_comm_finalize    leaves the communicator (remd_destroy does it too)
   librccl is opened at run time by the first of these calls (REMD_RCCL_LIB overrides the name), never linked.            */
#define REMD_COMM_ID_BYTES 128
int  remd_comm_unique_id(void* id /*[REMD_COMM_ID_BYTES]*/);
int  remd_comm_init(remd_handle h, int rank, int world, const void* id /*[REMD_COMM_ID_BYTES]*/);
int  remd_comm_all_gather_energies(remd_handle h);
int  remd_comm_finalize(remd_handle h);

/* Replaces ReplicaExchangeSampler._mix_replicas (replicaexchange.py:255-292),
   _mix_all_replicas_numba (:294-349), _mix_neighboring_replicas (:366-380) and
   SAMSSampler._global_jump (sams.py:477-501).
   d_ukl: DEVICE [R][ld] (first K columns are the sampled states; ld >= K, 0 means K) or
   NULL = the handle's own matrix.  labels: host [R] in/out.
   n_accepted / n_proposed: host [K][K], overwritten (the reference zeroes them per call).
   log_weights: host [K] (SAMS) or NULL.  sams_log_P: host [R][K] out (SAMS) or NULL.    */
int  remd_mix(remd_handle h, int scheme, int64_t iteration, int R, int K,
              const double* d_ukl, int ld, int64_t* labels,
              int64_t* n_accepted, int64_t* n_proposed,
              const double* log_weights, double* sams_log_P);

/* Stand-alone mixing on a host u_kl (copied to the device first): what a caller that
   only wants the Gibbs swap kernel binds (the reference's test_mixing.py:11-46 shape).  */
int  remd_mix_host(remd_handle h, int scheme, int64_t iteration, int R, int K,
                   const double* ukl_host, int64_t* labels,
                   int64_t* n_accepted, int64_t* n_proposed,
                   const double* log_weights, double* sams_log_P, int64_t n_attempts);

/* ---- snapshots / test hooks --------------------------------------------------------- */
/* Replaces SamplerState.update_from_context (mcmc.py:731-773, states.py:2431-2490).
   Any pointer may be NULL.  x, v: [R_local][N][3]; potential, kinetic: [R_local].       */
int  remd_get_replicas(remd_handle h, double* x, double* v, double* potential, double* kinetic);
/* forces in kJ/mol/nm, [R_local][N][3] (evaluates them first)                           */
int  remd_get_forces(remd_handle h, double* f);
/* run a splitting string once per call on all local replicas with explicit step counter
   (test hook: single V / R / O substeps).                                               */
int  remd_step(remd_handle h, const char* splitting, int64_t iteration, int64_t first_step, int n_steps);
/* test hook: potential-energy components per local replica, out[R_local][9] =
   {external, bonds, angles, torsions, exceptions, Ewald exclusion correction, PME reciprocal,
    constants (dispersion + Ewald self + background), nonbonded direct}                  */
int  remd_get_energy_components(remd_handle h, double* out);
int  remd_sync(remd_handle h);
/* test hook: in-place unnormalised 3-D complex FFT of a host array [nx][ny][nz][2] on the
   in-tree mixed-radix FFT that the PME reciprocal pass uses                             */
int  remd_test_fft3d(remd_handle h, int nx, int ny, int nz, float* data, int inverse);

/* test hook (host arithmetic only, no device needed): minus_G[k] = -G(u[k]) of the force-only Ewald direct-space kernels,
   F_i = q_i q_j (-G(r^2)) (x_j - x_i) k_e, G(u) = (erfc(alpha r)/r + 2 alpha/sqrt(pi) exp(-alpha^2 u))/u, as libremd_hip.so
   evaluates it: from the cubic table in r^2 the pair kernels stage in LDS, in the same f32 operations (csrc/coulomb_table.h).
   libremd_cpu.so answers with the f64 closed form.  u in [2^-8, coulomb_cutoff^2] nm^2.                                 */
int  remd_test_coulomb_table(double alpha, double coulomb_cutoff_nm, int n, const float* u, float* minus_G);

/* timing of the last remd_propagate / compute_energies / mix on the handle's stream,
   measured with hipEvents (ms)                                                           */
int  remd_last_timing(remd_handle h, double* propagate_ms, double* energies_ms, double* mix_ms);

/* per-kernel-class accounting for bench.py's roofline object: HIP events are recorded on the
   handle's stream around each launch (no synchronisation at launch time) and resolved when
   queried.  on = 1: only the class set with remd_profile_filter (default "nonbonded");
   on = 2: every class ("integrate_chain", "nonbonded", "pme_fft", "pme_spread", "pme_gather",
   "bonded", "exceptions", "mix_swap_all", ...).                                           */
int  remd_profile_enable(remd_handle h, int on);
int  remd_profile_filter(remd_handle h, const char* kernel_class);
int  remd_profile_get(remd_handle h, const char* kernel_class, int64_t* n_launches, double* total_ms);
int  remd_profile_reset(remd_handle h);

/* measurement hook: the achievable roofs of the box the benchmark runs on -- STREAM triad on 3 x 512 MiB
   (GB/s), independent v_fma_f32 chains and v_pk_fma_f32 chains (TFLOP/s).  Any pointer may be NULL.  bench.py reports
   them beside the spec peaks; the reference has no counterpart.                                                          */
int  remd_roof_microbench(remd_handle h, double* stream_gb_per_s, double* fma_tflop_per_s, double* pk_fma_tflop_per_s);
/* the shader clock (GHz) the chip sustains under that FMA load (cycle counter against the constant 100 MHz wall clock inside one
   wavefront of the microbenchmark): the spec peak of 157.3 TFLOP/s is quoted at 2.4 GHz.  libremd_cpu.so: -3.               */
int  remd_roof_clock_ghz(remd_handle h, double* ghz_under_fma_load);
/* Issue floor of the direct-space pair kernel (diagnostic, roofs.hip): a replay of its cluster-pair step -- 27 VALU instructions and
 * one 16-byte LDS read, operands in registers -- at `waves_per_simd` (1 ... 8) resident wavefronts per SIMD with `chains` (1 or 2)
 * independent steps in flight per wavefront; returns shader cycles per step per SIMD at the clock `ghz` and the launch's time. */
int  remd_roof_pair_step(remd_handle h, int waves_per_simd, int chains, double ghz, double* cycles_per_step_per_simd, double* us_total);

#ifdef __cplusplus
}
#endif
#endif /* REMD_HIP_H */
