// Internal declarations of libremd_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <mutex>
#include <map>
#include <vector>
#include <map>
#include "../../include/remd_hip.h"

#define REMD_KB 0.00831446261815324   // kJ/mol/K  (BOLTZMANN_CONSTANT_kB * AVOGADRO_CONSTANT_NA)
#define REMD_ONE_4PI_EPS0 138.93545764438198

struct remd_error { int code; std::string msg; };

// fixed-point scale of the force accumulators (deterministic integer atomics)
#define REMD_FORCE_SCALE 4294967296.0   // 2^32
// what the X-H position solve is held to when the integrator asks for less: relative bond-length error of fp32 coordinates taken
// relative to the cluster's central atom (a few units in the last place of |r|^2)
#define REMD_CONSTRAINT_TOL_FLOOR 2e-7

// scaled fractional mesh coordinate u in [0, n) of a position component and its integer part: ONE definition, because the PME
// spreading pass recomputes the mesh column of an atom that was binned elsewhere (pme_bin_kernel or the integrator chain's
// epilogue) and the two must agree bit for bit
__device__ __forceinline__ void remd_pme_scaled1(float x, float L, int n, float& u, int& k)
{
    float f = x / L;
    f -= floorf(f);
    u = f * n;
    k = (int)u;
}
// what the integrator chain needs to bin the atoms it has just moved by mesh column kx (pme.hip owns the arrays):
// count[r][nx] (zero on entry) and atoms[r][nx][cap]; NULL count = no binning
// (an entry is the atom's position with its index in .w: the spreading pass then needs no second dependent load for it)
// (and its effective charge in a parallel array: no dependent load of the per-atom parameters either)
struct remd_chain_bins { int nx = 0, cap = 0; int* count = nullptr; float4* atoms = nullptr; const float* box = nullptr; unsigned int* err = nullptr;
                         float* q = nullptr; const float4* param = nullptr; const float* rep_lam = nullptr; };

// (unsigned long long)(long long)((double)f * 2^32), i.e. truncation toward zero, bit for bit, without the f64 conversion
// chain the cast expands to (8 double-rate instructions per component): |f| = floor + fraction is exact in f32, the
// fraction times 2^32 is exact, and the two halves are converted separately.  |f| >= 2^31 saturates as before.
__device__ __forceinline__ unsigned long long remd_f2fix(float f)
{
    const float a = fabsf(f);
    const float hi_f = floorf(a);
    const unsigned int lo = (unsigned int)((a - hi_f) * 4294967296.f);
    const unsigned int hi = (unsigned int)(int)hi_f;
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    return f < 0.f ? 0ull - v : v;
}

// The join of the two streams without a launch of its own (round 4): the direct-space stream's last launch (the scatter of the
// pair kernel's sorted accumulators) has every workgroup count itself done with one atomic that nobody waits for, and the
// integrator chain's prologue polls that count instead of a flag published by a one-wavefront launch behind the scatter (5.6 us
// on what has become the critical path of a step).  Measured on the way and rejected: the scatter's LAST workgroup publishing the
// flag (it has to wait for its arrival atomic: 14 - 28 us), and the chain folding the sorted accumulators itself (no scatter
// launch at all, but +9 us in the one launch of a step that nothing overlaps): profiles/r04_h_rejected.txt, r04_p_fold_rejected.txt.
struct remd_fold_args {
    const unsigned int* done = nullptr; unsigned int target = 0;     // scatter workgroups finished (cumulative over launches)
};

struct listed_tables {
    int n_bonds, n_angles, n_torsions, n_exc, n_excl;
    const int *bond_atoms, *angle_atoms, *torsion_atoms, *exc_atoms, *excl_atoms;
    const float *bond_params, *angle_params, *torsion_params, *exc_params, *excl_qq;
    const int *exc_alch, *excl_alch; const float* rep_lam;
    float alpha, two_alpha_sqrtpi;
    // (term, slot) entries in the order of the atoms they act on (listed_terms.h); aterm NULL: one term per thread
    int n_aterm; const unsigned int* aterm;
};

struct remd_profile_entry { int64_t n = 0; double ms = 0.0; };

// a deep copy of the descriptor of the last remd_set_system: the groups of replicas a handle propagates as phases (api.hip) are set up
// from it without the host
struct remd_desc_store {
    remd_system_desc d{}; bool valid = false;
    std::vector<double> mass, bond_params, angle_params, torsion_params, charge, sigma, epsilon, exception_params, shake_dist;
    std::vector<int32_t> ext_atoms, bond_atoms, angle_atoms, torsion_atoms, exception_atoms, settle_atoms, shake_atoms, alch_atoms;
    void assign(const remd_system_desc* s)
    {
        d = *s;
        auto cpd = [](std::vector<double>& v, const double*& p, size_t n) { if (p && n) { v.assign(p, p + n); p = v.data(); } else { v.clear(); if (!n) p = nullptr; } };
        auto cpi = [](std::vector<int32_t>& v, const int32_t*& p, size_t n) { if (p && n) { v.assign(p, p + n); p = v.data(); } else { v.clear(); if (!n) p = nullptr; } };
        const size_t N = (size_t)d.n_atoms;
        cpd(mass, d.mass, N); cpi(ext_atoms, d.ext_atoms, (size_t)d.n_ext);
        cpi(bond_atoms, d.bond_atoms, 2 * (size_t)d.n_bonds); cpd(bond_params, d.bond_params, 2 * (size_t)d.n_bonds);
        cpi(angle_atoms, d.angle_atoms, 3 * (size_t)d.n_angles); cpd(angle_params, d.angle_params, 2 * (size_t)d.n_angles);
        cpi(torsion_atoms, d.torsion_atoms, 4 * (size_t)d.n_torsions); cpd(torsion_params, d.torsion_params, 3 * (size_t)d.n_torsions);
        cpd(charge, d.charge, N); cpd(sigma, d.sigma, N); cpd(epsilon, d.epsilon, N);
        cpi(exception_atoms, d.exception_atoms, 2 * (size_t)d.n_exceptions); cpd(exception_params, d.exception_params, 3 * (size_t)d.n_exceptions);
        cpi(settle_atoms, d.settle_atoms, 3 * (size_t)d.n_settle);
        cpi(shake_atoms, d.shake_atoms, 4 * (size_t)d.n_shake); cpd(shake_dist, d.shake_dist, 3 * (size_t)d.n_shake);
        cpi(alch_atoms, d.alch_atoms, (size_t)d.n_alch);
        valid = true;
    }
};

struct remd_ctx {
    int device = 0;
    hipStream_t stream = nullptr; bool owns_stream = false;
    std::string err;
    uint64_t seed = 0;

    // ---- system -------------------------------------------------------------------
    int N = 0;            // atoms
    int Npad = 0;         // atoms padded to a multiple of 64
    bool has_system = false;
    float* d_invmass = nullptr;        // [Npad]  (0 for padding)
    float* d_mass = nullptr;           // [Npad]
    double total_mass = 0.0;
    // harmonic external force
    int n_ext = 0; int* d_ext_atoms = nullptr; double ext_K = 0, ext_x0 = 0, ext_U0 = 0;
    // bonded
    int n_bonds = 0, n_angles = 0, n_torsions = 0;
    int* d_bond_atoms = nullptr; float* d_bond_params = nullptr;
    unsigned int* d_aterm = nullptr; int n_aterm = 0;      // (term, slot) entries of the listed terms by atom (forces.hip: build_atom_terms)
    int* d_angle_atoms = nullptr; float* d_angle_params = nullptr;
    int* d_torsion_atoms = nullptr; float* d_torsion_params = nullptr;
    // nonbonded
    int nb_method = REMD_NB_NONE;
    int gbsa = 0;                      // remd_set_gbsa: GBSA (OBC2 + ACE) of a NoCutoff system (gbsa.hip)
    int nocutoff = 0;                  // the descriptor asked for REMD_NB_NOCUTOFF: nb_method stays REMD_NB_NONE for the rest of the engine, nocutoff.hip adds the direct sum
    double cutoff = 0, switch_dist = -1, rf_dielectric = 78.3, ewald_alpha = 0;
    int annihilate_sterics = 0;        // remd_set_alchemical_options: alchemical/alchemical sterics are lambda-controlled too
    int rf_unshifted = 0; double rf_switch_width = 0;   // remd_set_reaction_field: c_rf = 0, pair term switched over the last rf_switch_width nm
    double coulomb_cutoff = 0;         // remd_set_coulomb_cutoff: range of the Ewald direct-space sum (0: the NonbondedForce cutoff)
    int grid[3] = {0, 0, 0};
    int use_disp = 0;
    double disp_coeff = 0;             // E_lrc = disp_coeff / V for the non-alchemical system
    float4* d_nbparam = nullptr;       // [Npad] (q, sigma/2, 2*sqrt(eps), alch flag)
    unsigned long long* d_exclmask = nullptr; // [Npad][EXCL_WORDS] window exclusion bit masks
    int excl_window = 0;               // atoms j in [i-excl_window, i+excl_window] are mask-addressable
    int n_exceptions = 0;              // 1-4 style exceptions with non-zero params
    int* d_exc_atoms = nullptr; float* d_exc_params = nullptr;   // [n][2], [n][3]
    int n_excl_pairs = 0;              // all excluded pairs (for the PME exclusion correction)
    int* d_excl_pairs = nullptr;       // [n][2]
    double self_energy = 0;            // PME self term (kJ/mol), lambda_elec = 1
    // constraints
    int n_settle = 0; int* d_settle_atoms = nullptr; double settle_dOH = 0, settle_dHH = 0;
    int n_shake = 0;  int* d_shake_atoms = nullptr; float* d_shake_dist = nullptr;
    int n_groups = 0; int* d_group_first = nullptr;   // constraint groups (molecule-like units)
    int* d_free_atoms = nullptr; int n_free = 0;      // atoms in no constraint
    int cmm_frequency = 0;
    int n_dof = 0;
    // alchemy
    int n_alch = 0; int* d_alch_atoms = nullptr;
    double sc_alpha = 0.5, sc_a = 1, sc_b = 1, sc_c = 6;
    int n_regions = 0;                 // remd_set_alchemical_regions: general regions (alch_regions.hip holds the tables)
    int regions_exact = 0;             // ... under the exact PME treatment (the regions' scaled charges inside the Ewald sum)

    // ---- states ---------------------------------------------------------------------
    int K = 0;
    std::vector<double> beta, lam_s, lam_e, econst;
    double* d_beta = nullptr; double* d_lam_s = nullptr; double* d_lam_e = nullptr; double* d_econst = nullptr;

    // ---- integrator -----------------------------------------------------------------
    std::string splitting = "V R O R V";
    std::vector<char> tokens;          // parsed 'V','R','O'
    int nV = 0, nR = 0, nO = 0;
    // multiple-time-step splittings (V0 V1 ..., integrators.py:1425-1442): tokens '0'..'3' = V of that force group;
    // nVg[g] = occurrences per step, fgroup[c] = force group of class c (REMD_FG_*), d_force_g[g] = that group's forces
    int nVg[4] = {0, 0, 0, 0};
    int fgroup[6] = {0, 0, 0, 0, 0, 0};
    long long* d_force_g[4] = {nullptr, nullptr, nullptr, nullptr}; size_t force_g_n = 0;
    double dt = 0.001, gamma = 1.0, constraint_tol = 1e-8;
    int n_steps = 1; int reassign = 1;
    bool has_integrator = false;
    // heat / shadow work / Metropolization (integrators.py:1175-1204, 1404-1460, 1539-1557)
    int measure_heat = 0, measure_shadow = 0;                   // (a splitting with '{' '}' measures shadow work whatever the flag says)
    long long* d_snap_work = nullptr;  // [2][R][4] remd_propagate: d_work at the start of the call / of each replica's successful attempt
    long long* d_work = nullptr;       // [R][4] fixed point 2^-24 kJ/mol: heat, shadow work; integers: Metropolis trials, rejections
    double* d_pe_prev = nullptr;       // [R] potential energy at the positions the last energy evaluation saw
    float4* d_xold = nullptr; float4* d_vold = nullptr; int* d_accept = nullptr;   // '{' snapshot, per-replica decision of '}'
    int work_R = 0;

    // ---- replicas -------------------------------------------------------------------
    int R_global = 0, r_begin = 0, R = 0;     // R = local replicas
    // remd_set_replica_ids: what keys the local replicas' random streams instead of r_begin + r (a handle that holds a
    // non-contiguous subset of an ensemble, multistate/_engine_pool.py); NULL: the block's own global indices
    unsigned int* d_noise_id = nullptr;
    float4* d_pos = nullptr;           // [R][Npad] xyz + pad
    float4* d_vel = nullptr;           // [R][Npad] xyz + pad
    // Monte Carlo barostat (OpenMM MonteCarloBarostat as the reference's NPT ThermodynamicState adds it, states.py:1177-1181)
    int baro_frequency = 0; long long baro_steps = 0, baro_attempts = 0; std::vector<double> pressure_host;   // (pressure_host: for the blocks of a phased propagation)
    double econst_vref = 0.0;          // volume at which the per-state energy constants were evaluated (they scale as 1/V); 0: constant
    double* d_pressure = nullptr;      // [K] kJ/mol/nm^3 (bar * N_A * 1e-25)
    double* d_baro = nullptr;          // [R][8]: volumeScale, attempted, accepted (adaptation window), total attempted, total accepted, dV, newV, oldV
    float* d_box_old = nullptr; float4* d_baro_x0 = nullptr; long long* d_baro_f0 = nullptr; double* d_baro_U0 = nullptr; int* d_baro_acc = nullptr;
    bool box_uniform = false;          // every local replica has the same box (set_replicas; a barostat move clears it)
    int box_version = 0;               // bumped whenever the box edges on the device change (PME influence table)
    int n_restart_attempts = 0;        // mcmc.py:706-759
    unsigned int* d_mix_log = nullptr; size_t mix_log_n = 0;      // swap-all attempt log (si, sj, accepted) when the counters do not fit
    bool mix_pre_launched = false, mix_no_pre = false;   // swap-all: the hoisted path ran last (its overflow flag is pending) / is off for the repeat
    double mix_acc_rate = -1.0;        // accepted / proposed of the previous swap-all call (picks the kernel of the next: mix.hip) in LDS
    float4* d_snap_pos = nullptr; float4* d_snap_vel = nullptr;   // pre-propagate state (restart attempts)
    float4* d_fin_pos = nullptr; float4* d_fin_vel = nullptr;     // first successful result of every replica
    float* d_snap_box = nullptr; float* d_fin_box = nullptr;      // boxes move under the barostat: same treatment
    float4* d_pos_ref = nullptr;       // [R][Npad] reference (constrained) positions for SHAKE/SETTLE
    long long* d_force = nullptr;      // [R][3][Npad] fixed point
    float* d_box = nullptr;            // [R][4] lx, ly, lz, pad
    std::vector<double> box_host;      // [R][3]
    int64_t* d_labels = nullptr;       // [R_global]
    std::vector<int64_t> labels;
    double* d_ukl = nullptr;           // [R_global][K]
    double* d_potential = nullptr;     // [R]
    double* d_epart = nullptr; int n_epart = 0;   // per-replica per-block energy partials
    double* d_kinetic = nullptr;       // [R]
    int* d_nan = nullptr;              // [R]
    long long* d_cmm = nullptr;        // [2][R][4] double-buffered fixed-point momentum accumulators
    bool forces_valid = false;
    bool force_zeroed = false;         // the last integrator chain already cleared d_force (skip the memset)

    // ---- PME ------------------------------------------------------------------------
    void* pme = nullptr;               // opaque (pme.hip)

    // ---- mixing scratch ----------------------------------------------------------------
    unsigned long long* d_nacc = nullptr; unsigned long long* d_nprop = nullptr; int stats_K = 0;
    double* d_logw = nullptr; double* d_logP = nullptr; double* d_ukl_tmp = nullptr; size_t ukl_tmp_n = 0;

    // ---- timing / profiling -------------------------------------------------------------
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream: the PME reciprocal pipeline (LDS / latency bound) overlaps the direct-space kernels (VALU bound)
    // fork / join of the two streams by flags in device memory ([0] fork, [1] join, [2] a spin ran out): the first mesh kernel
    // publishes the fork, a one-wavefront kernel at the head of the second stream waits for it, and the mirror image at the
    // join -- an event record / wait costs ~6 us of command-processor latency on the critical path, twice per step.
    // REMD_SYNC_EVENTS=1 keeps the events (also the fall-back of a handle whose polled wait ran out, api.hip).
    unsigned int* d_sync = nullptr; unsigned int sync_seq = 0; unsigned int fork_seq_pending = 0; bool sync_events = false;
    // remd_run_steps: the launch that follows a force evaluation on the main stream is always an integrator chain, so the
    // join is polled in that kernel's prologue (join_deferred = sequence number to wait for) instead of a kernel of its own
    bool defer_join_ok = false; unsigned int join_deferred = 0;
    listed_tables mesh_listed{}; int mesh_listed_total = 0;   // listed terms to ride in the next spreading launch (0: none)
    remd_fold_args fold; bool fold_pending = false;      // the next chain launch polls the scatter's done counter instead of a join flag (forces.hip)
    bool mesh_prio_hi = true;          // which of the two streams' kernels run at raised wave priority (forces.hip: chosen with the pair-kernel residency)
    // set when a wait polled on the device ran out / a capped PME bin overflowed: the handle falls back to events, two chain
    // launches and the binning launch (api.hip: remd_recover_device_flag) instead of staying dead behind a sticky flag
    bool no_device_waits = false, no_chain_bins = false, no_resident = false, no_chain_merge = false;
    // round 6, several handles propagated step by step from one host thread (remd_propagate_many): no workgroup of this handle may sit
    // on a CU polling for another stream while it holds registers another handle's kernels need -- the join is a one-wavefront launch
    // in front of the chain instead of a poll in the chain's prologue (320 registers per lane on every CU it occupies), the momentum
    // sum two launches instead of a barrier over resident workgroups
    bool lean_waits = false, no_chain_barrier = false;
    // round 6 (DHFR-size meshes): hold the pair kernel of a forked evaluation until the LDS-resident plane pass has ended
    bool pair_after_xy = false, xy_recorded = false; hipEvent_t ev_xy = nullptr; size_t xy_lds_bytes = 0;
    // ---- phases (round 6): remd_propagate of ONE handle as two groups of replicas whose MD steps take turns ------------------------
    // The local replicas are split into contiguous blocks, each block propagated by a child context of its own (full tables for its
    // share of the replicas; block 0 launches on THIS handle's two streams, block 1 on one more pair), the blocks' steps enqueued in
    // turn from the calling thread (remd_propagate_many): the integrator chain of one block runs beside the pair and mesh kernels
    // of the other.  Coordinates, velocities and the forces the last evaluation left go to the children device to device in front and
    // come back behind; everything else (energies, mixing, get / set) stays with this handle.  phases_req: remd_set_phases (0 = by rule).
    int phases_req = 0; int phases_last = 1;
    std::vector<remd_ctx*> phase; remd_ctx* parent = nullptr; int seen_parent_box = -1; long long ids_version = 0, seen_parent_ids = -1;     // (child: the parent's box_version its boxes were taken at)
    long long config_version = 0, phase_config = -1;      // children are rebuilt when a setter has run since they were made
    remd_desc_store* sysdesc = nullptr;
    std::vector<int64_t> noise_id_host;                   // remd_set_replica_ids, for the children's slices
    bool borrowed_stream2 = false;     // stream2 belongs to another handle (remd_adopt_streams): not destroyed with this one
    unsigned long long* d_chain_own = nullptr;   // [2] profiling: sum of (end - flag seen) wall-clock ticks of workgroup (0, 0), launches
    unsigned long long* d_chain_sync = nullptr; unsigned int chain_sync_epoch = 0; long long chain_sync_key = -1;    // [2][R][workgroups][3] epoch-tagged partial momentum sums of the 'M' token (integrate.hip)
    bool cbins_ready = false;          // the chain launched last binned the atoms for the PME pass of the evaluation that follows
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool overlap = true; bool pme_concurrent = false;
    // sharding without a Python host (comm.hip): an RCCL communicator over the ranks of one replica-exchange run
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    std::vector<long long> comm_begin, comm_count;   // every rank's block of replicas, exchanged when the local block changes
    long long* d_comm_part = nullptr; bool comm_part_current = false;
    double t_prop = 0, t_energy = 0, t_mix = 0;
    int profiling = 0;                 // 0 off, 1 filtered class only, 2 all classes
    std::string prof_filter = "nonbonded";
    int prof_every = getenv("REMD_PROF_EVERY") ? std::max(1, atoi(getenv("REMD_PROF_EVERY"))) : 16;   // level 1: sample every n-th launch of a class
    std::map<const char*, long> prof_seen;
    struct pending_t { std::string name; hipEvent_t a, b; };
    std::vector<pending_t> prof_pending;
    std::map<std::string, remd_profile_entry> prof;
};

#define REMD_CHECK(h, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e); remd_set_global_error((h)->err); return -2; } } while (0)

void remd_set_global_error(const std::string& s);
int remd_fail(remd_ctx* h, int code, const std::string& msg);
void remd_comm_release(remd_ctx* h);      // comm.hip

// profiling wrapper: brackets a launch with HIP events recorded on the handle's stream.  Nothing is
// synchronised at launch time; the pairs are resolved in remd_profile_get().  Level 1 records only
// the class named by prof_filter (bench.py: the dominant kernel), level 2 records every class.
// Per-handle side tables (defined in the .hip files that own them).  Distinct handles may be used from distinct threads
// (include/remd_hip.h): look-ups and insertions are serialised, and std::map never moves its elements, so a reference
// obtained here stays valid until the handle itself is destroyed.
template <typename T>
struct handle_table {
    std::mutex m;
    std::map<remd_ctx*, T> map;
    T& operator[](remd_ctx* h) { std::lock_guard<std::mutex> l(m); return map[h]; }
    T* find(remd_ctx* h) { std::lock_guard<std::mutex> l(m); auto it = map.find(h); return it == map.end() ? nullptr : &it->second; }
    void erase(remd_ctx* h) { std::lock_guard<std::mutex> l(m); map.erase(h); }
};

struct remd_prof_scope {
    remd_ctx* h; const char* name; hipEvent_t a = nullptr; bool on = false; hipStream_t st;
    remd_prof_scope(remd_ctx* h_, const char* n, hipStream_t stream = (hipStream_t)-1) : h(h_), name(n) {
        st = (stream == (hipStream_t)-1) ? h->stream : stream;     // events go on the stream the kernel is launched on
        on = h->profiling == 2;
        if (h->profiling == 1) {                       // filter: '|'-separated class-name prefixes
            const std::string nm(n), &f = h->prof_filter;
            for (size_t b = 0; b <= f.size() && !on; ) {
                size_t e = f.find('|', b); if (e == std::string::npos) e = f.size();
                on = e > b && nm.compare(0, e - b, f, b, e - b) == 0;
                b = e + 1;
            }
            // sampled: every prof_every-th launch of a class (an event pair costs host time and, on the main stream, ~12 us of
            // command-processor latency on the critical path of a step: timing every launch slows what it measures)
            if (on) on = (h->prof_seen[n]++ % h->prof_every) == 0;
        }
        if (on) { hipEventCreate(&a); hipEventRecord(a, st); }
    }
    ~remd_prof_scope() {
        if (on) {
            hipEvent_t b; hipEventCreate(&b); hipEventRecord(b, st);
            h->prof_pending.push_back({name, a, b});
        }
    }
};

// ---- mix.hip --------------------------------------------------------------------------
int remd_mix_launch(remd_ctx* h, int scheme, int64_t iteration, int R, int K, int ld, const double* d_ukl,
                    int64_t* d_labels, unsigned long long* d_nacc, unsigned long long* d_nprop,
                    const double* d_logw, double* d_logP, int64_t n_attempts);

// ---- integrate.hip ----------------------------------------------------------------------
int remd_parse_splitting(remd_ctx* h, const char* splitting, std::vector<char>& tokens, int& nV, int& nR, int& nO, int* nVg = nullptr);
int remd_run_steps(remd_ctx* h, const std::vector<char>& tokens, int nV, int nR, int nO,
                   int64_t iteration, int64_t first_step, int n_steps);
int remd_run_steps_many(remd_ctx** hs, int n, int64_t iteration, int64_t first_step, int n_steps);
long long remd_chain_blocks(remd_ctx* h);            // workgroups of one integrator-chain launch   // the handles' steps taking turns
int remd_assign_velocities(remd_ctx* h, int64_t iteration);
int remd_kinetic_energy(remd_ctx* h);
int remd_work_buffers(remd_ctx* h);                    // heat / shadow-work accumulators and the '{' snapshot of the local replicas

// ---- forces.hip -------------------------------------------------------------------------
int remd_barostat_attempt(remd_ctx* h);                      // barostat.hip
int remd_barostat_buffers(remd_ctx* h);                      // (its per-replica state and scratch, allocated on first use)
int remd_nb_molecules(remd_ctx* h, const int** first, const int** size);   // molecule table of the nonbonded setup (device); 0: none
int remd_minimize_impl(remd_ctx* h, double tolerance, int max_iterations, int32_t* converged, int32_t* n_iterations);
void remd_nb_invalidate_sort(remd_ctx* h);            // the next force evaluation re-sorts the molecules
// force classes (bits of the mask of remd_compute_forces, indices of remd_ctx::fgroup)
#define REMD_FG_EXTERNAL 0
#define REMD_FG_BOND 1
#define REMD_FG_ANGLE 2
#define REMD_FG_TORSION 3
#define REMD_FG_NONBONDED 4      /* direct space, exceptions, Ewald exclusion correction */
#define REMD_FG_RECIPROCAL 5
// nocutoff.hip: NonbondedForce with NoCutoff (vacuum systems)
void remd_nocutoff_release(remd_ctx* h);
int remd_nocutoff_build(remd_ctx* h, const remd_system_desc* d);
int remd_nocutoff_forces(remd_ctx* h, bool with_energy, int ep_slot);
int remd_nocutoff_info(remd_ctx* h, const float4** param, const unsigned int** excl, int* words, int* n_exc, const int** exc_atoms, const float4** exc_par);
// gbsa.hip: implicit solvent of a NoCutoff system
void remd_gbsa_release(remd_ctx* h);
int remd_gbsa_forces(remd_ctx* h, bool with_energy, int ep_slot);
int remd_gbsa_ukl(remd_ctx* h, double* d_alch /*[R][K], added to*/);
int remd_regions_state_le(remd_ctx* h, int k, int g, double* le);
// alch_regions.hip: custom forces of general alchemical regions
void remd_regions_release(remd_ctx* h);
int remd_regions_clone(remd_ctx* parent, remd_ctx* child);
int remd_gbsa_clone(remd_ctx* parent, remd_ctx* child);
int remd_regions_forces(remd_ctx* h, bool with_energy, int ep_slot);
int remd_regions_ukl(remd_ctx* h, double* d_out /*[R][K]*/, const int** d_own);
int remd_regions_pme_tables(remd_ctx* h, const float4** param, const float** rep_le);
int remd_regions_le_override(remd_ctx* h, const float* le, int* n, const float** d_state_le);
int remd_compute_forces(remd_ctx* h, bool with_energy, unsigned class_mask = ~0u);   // fills d_force (and d_potential when with_energy)
int remd_build_nonbonded(remd_ctx* h, const remd_system_desc* d);
int remd_build_constraints(remd_ctx* h, const remd_system_desc* d);

// ---- pme.hip ----------------------------------------------------------------------------
int remd_pme_setup(remd_ctx* h);
int remd_pme_destroy(remd_ctx* h);
int remd_pme_forces(remd_ctx* h, bool with_energy, hipStream_t st, int part = 3);   // part 1: bin .. inverse z, part 2: gather (+ energy)
