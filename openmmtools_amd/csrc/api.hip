// C ABI of libremd_hip.so (see include/remd_hip.h for the contract and reference citations).
#include <chrono>
#include "remd_internal.h"
#include "coulomb_table.h"
#include <cstring>
#include <cmath>
#include <mutex>
#include <cstdlib>

int remd_check_finite(remd_ctx* h);
int remd_assemble_ukl(remd_ctx* h, double* d_rows);
int remd_nb_required_epart(remd_ctx* h);
void remd_free_nonbonded(remd_ctx* h);
void remd_mix_release(remd_ctx* h);          // mix.hip
void remd_nb_reset_accumulators(remd_ctx* h); // forces.hip
void remd_nb_invalidate_sort(remd_ctx* h);    // forces.hip
int remd_test_fft3d_impl(remd_ctx* h, int nx, int ny, int nz, float* data, int inverse);
void remd_nb_tune_resolve(remd_ctx* h);
static int remd_check_device_flags(remd_ctx* h, const char* where, bool may_retry = false);
const unsigned* remd_mix_pending_flag(remd_ctx* h);      // mix.hip
void remd_free_constraints(remd_ctx* h);

static std::mutex g_err_mutex;
static std::string g_last_error;

void remd_set_global_error(const std::string& s) { std::lock_guard<std::mutex> l(g_err_mutex); g_last_error = s; }
int remd_fail(remd_ctx* h, int code, const std::string& msg) { if (h) h->err = msg; remd_set_global_error(msg); return code; }

template <typename T> static void dfree(T*& p) { if (p) { hipFree(p); p = nullptr; } }

template <typename T>
static int upload(remd_ctx* h, T*& dptr, const std::vector<T>& host)
{
    dfree(dptr);
    if (host.empty()) return 0;
    REMD_CHECK(h, hipMalloc(&dptr, sizeof(T) * host.size()));
    REMD_CHECK(h, hipMemcpy(dptr, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    return 0;
}

extern "C" {

int remd_version(void) { return 1; }

const char* remd_last_error(remd_handle h)
{
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> l(g_err_mutex);
    static thread_local std::string copy;
    copy = g_last_error;
    return copy.c_str();
}

int remd_create(remd_handle* out, int device, void* stream)
{
    if (!out) return remd_fail(nullptr, -1, "remd_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return remd_fail(nullptr, -2, std::string("remd_create: no HIP device available (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= n) return remd_fail(nullptr, -1, "remd_create: bad device index");
    e = hipSetDevice(device);
    if (e != hipSuccess) return remd_fail(nullptr, -2, std::string("hipSetDevice: ") + hipGetErrorString(e));
    remd_ctx* h = new remd_ctx();
    h->device = device;
    h->stream = (hipStream_t)stream;
    // Spatial partition of the chip between the two streams of a force evaluation (experiment, profiles/r04_cumask_sweep.txt):
    // REMD_CU_PAIR = n restricts the direct-space stream to n of the 256 CUs, REMD_CU_MESH = m the handle's own main stream
    // (only when the caller passed none) to m CUs taken from the other end.  REMD_CU_LAYOUT: 0 = mask bit b is CU b as the
    // runtime numbers them (KFD deals consecutive bits round-robin to the 8 XCDs), 1 = bit b is CU (b % 32) of XCD (b / 32).
    auto cu_mask = [](int n, bool from_top, uint32_t* m) {
        const int layout = getenv("REMD_CU_LAYOUT") ? atoi(getenv("REMD_CU_LAYOUT")) : 0;
        for (int w = 0; w < 8; ++w) m[w] = 0u;
        n = std::max(8, std::min(256, n));
        for (int k = 0; k < n; ++k) {
            int b;
            if (layout == 0) b = from_top ? 255 - k : k;
            else { const int xcd = k % 8, cu = k / 8; b = xcd * 32 + (from_top ? 31 - cu : cu); }
            m[b >> 5] |= 1u << (b & 31);
        }
    };
    const int cu_pair = getenv("REMD_CU_PAIR") ? atoi(getenv("REMD_CU_PAIR")) : 0;
    const int cu_mesh = getenv("REMD_CU_MESH") ? atoi(getenv("REMD_CU_MESH")) : 0;
    if (!h->stream && cu_mesh > 0) {
        uint32_t m[8]; cu_mask(cu_mesh, false, m);
        if (hipExtStreamCreateWithCUMask(&h->stream, 8, m) == hipSuccess) h->owns_stream = true; else { h->stream = nullptr; (void)hipGetLastError(); }
    }
    if (!h->stream) {
        // NULL: a private non-blocking stream (the legacy default stream cannot be captured into a graph, and every entry point
        // that hands results to the caller synchronises before it returns, so nothing relies on default-stream ordering)
        // (REMD_MAIN_PRIO=1: the private main stream at raised queue priority -- experiment hook, profiles/r06_13_stream_priorities.txt)
        int lo0 = 0, hi0 = 0;
        hipDeviceGetStreamPriorityRange(&lo0, &hi0);
        const bool main_hi = getenv("REMD_MAIN_PRIO") && atoi(getenv("REMD_MAIN_PRIO")) != 0;
        if ((main_hi ? hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, hi0) : hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete h; return remd_fail(nullptr, -2, "hipStreamCreate failed");
        }
        h->owns_stream = true;
    }
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1);
    {
        // second stream: carries the direct-space kernels while the (longer) reciprocal-space chain stays on the main one
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (cu_pair > 0) {
            uint32_t m[8]; cu_mask(cu_pair, true, m);
            if (hipExtStreamCreateWithCUMask(&h->stream2, 8, m) != hipSuccess) { h->stream2 = nullptr; (void)hipGetLastError(); }
        }
        const bool direct_lo = getenv("REMD_DIRECT_PRIO") && atoi(getenv("REMD_DIRECT_PRIO")) == 0;      // (experiment hook: normal priority)
        if (!h->stream2 && (direct_lo || hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi) != hipSuccess))
            hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
    }
    { const unsigned evf = hipEventReleaseToDevice | hipEventDisableTiming;   // device-scope release: no system-scope cache write-back per fork / join
      hipEventCreateWithFlags(&h->ev_fork, evf); hipEventCreateWithFlags(&h->ev_join, evf); }
    { const char* env = getenv("REMD_OVERLAP"); h->overlap = !(env && atoi(env) == 0); }
    h->sync_events = getenv("REMD_SYNC_EVENTS") && atoi(getenv("REMD_SYNC_EVENTS")) != 0;
    if (hipMalloc(&h->d_chain_own, 40 * sizeof(unsigned long long)) == hipSuccess) hipMemset(h->d_chain_own, 0, 40 * sizeof(unsigned long long));
    if (hipMalloc(&h->d_sync, 4 * sizeof(unsigned int)) != hipSuccess || hipMemset(h->d_sync, 0, 4 * sizeof(unsigned int)) != hipSuccess) {
        delete h; return remd_fail(nullptr, -2, "remd_create: hipMalloc failed");
    }
    *out = h;
    return 0;
}

int remd_destroy(remd_handle h)
{
    if (!h) return 0;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (remd_ctx* c : h->phase) remd_destroy(c);         // (block 0 borrows this handle's streams: before they go)
    h->phase.clear();
    delete h->sysdesc; h->sysdesc = nullptr;
    remd_comm_release(h);
    remd_pme_destroy(h);
    remd_free_constraints(h);
    remd_free_nonbonded(h);
    remd_nocutoff_release(h);
    remd_gbsa_release(h);
    remd_regions_release(h);
    remd_mix_release(h);
    dfree(h->d_invmass); dfree(h->d_mass); dfree(h->d_ext_atoms);
    dfree(h->d_aterm); h->n_aterm = 0;
    dfree(h->d_bond_atoms); dfree(h->d_bond_params); dfree(h->d_angle_atoms); dfree(h->d_angle_params);
    dfree(h->d_torsion_atoms); dfree(h->d_torsion_params);
    dfree(h->d_nbparam); dfree(h->d_exclmask); dfree(h->d_exc_atoms); dfree(h->d_exc_params); dfree(h->d_excl_pairs);
    dfree(h->d_alch_atoms);
    dfree(h->d_beta); dfree(h->d_lam_s); dfree(h->d_lam_e); dfree(h->d_econst);
    dfree(h->d_pos); dfree(h->d_vel); dfree(h->d_pos_ref); dfree(h->d_force); dfree(h->d_box); dfree(h->d_labels);
    for (int g = 0; g < 4; ++g) dfree(h->d_force_g[g]);
    dfree(h->d_ukl); dfree(h->d_potential); dfree(h->d_epart); dfree(h->d_kinetic); dfree(h->d_nan); dfree(h->d_cmm);
    dfree(h->d_pressure); dfree(h->d_baro); dfree(h->d_box_old); dfree(h->d_baro_x0); dfree(h->d_baro_f0); dfree(h->d_baro_U0); dfree(h->d_baro_acc);
    dfree(h->d_mix_log); dfree(h->d_noise_id);
    dfree(h->d_snap_pos); dfree(h->d_snap_vel); dfree(h->d_fin_pos); dfree(h->d_fin_vel); dfree(h->d_snap_box); dfree(h->d_fin_box);
    dfree(h->d_snap_work);
    dfree(h->d_nacc); dfree(h->d_nprop); dfree(h->d_logw); dfree(h->d_logP); dfree(h->d_ukl_tmp);
    if (h->stream2 && !h->borrowed_stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    if (h->owns_stream && h->stream) hipStreamDestroy(h->stream);
    if (h->d_sync) hipFree(h->d_sync);
    if (h->ev_xy) hipEventDestroy(h->ev_xy);
    if (h->d_chain_own) hipFree(h->d_chain_own);
    if (h->d_work) { hipFree(h->d_work); hipFree(h->d_pe_prev); hipFree(h->d_xold); hipFree(h->d_vold); hipFree(h->d_accept); }
    if (h->d_chain_sync) hipFree(h->d_chain_sync);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    delete h;
    return 0;
}

int remd_seed(remd_handle h, uint64_t seed) { if (!h) return -1; h->seed = seed; h->config_version++; return 0; }

int remd_test_coulomb_table(double alpha, double coulomb_cutoff_nm, int n, const float* u, float* minus_G)
{
    if (!(alpha > 0) || !(coulomb_cutoff_nm > 0) || n < 0 || !u || !minus_G) return remd_fail(nullptr, -1, "remd_test_coulomb_table: bad arguments");
    const coulomb_table_host T = ctab_build(alpha, coulomb_cutoff_nm * coulomb_cutoff_nm);
    for (int k = 0; k < n; ++k) {
        const float uc = std::min(std::max(u[k], T.umin), (float)(coulomb_cutoff_nm * coulomb_cutoff_nm));   // the kernel's v_med3_f32
        minus_G[k] = ctab_eval_host(T, uc);
    }
    return 0;
}

int remd_set_alchemical_options(remd_handle h, int annihilate_sterics)
{
    if (!h || (annihilate_sterics != 0 && annihilate_sterics != 1)) return remd_fail(h, -1, "remd_set_alchemical_options: bad arguments");
    h->annihilate_sterics = annihilate_sterics;      // consumed by the next remd_set_system
    return 0;
}

int remd_set_reaction_field(remd_handle h, int unshifted, double switch_width_nm)
{
    if (!h || (unshifted != 0 && unshifted != 1) || !(switch_width_nm >= 0.0)) return remd_fail(h, -1, "remd_set_reaction_field: bad arguments");
    h->rf_unshifted = unshifted; h->rf_switch_width = switch_width_nm;      // consumed by the next remd_set_system
    return 0;
}

int remd_set_coulomb_cutoff(remd_handle h, double coulomb_cutoff_nm)
{
    if (!h || !(coulomb_cutoff_nm >= 0.0)) return remd_fail(h, -1, "remd_set_coulomb_cutoff: bad arguments");
    h->coulomb_cutoff = coulomb_cutoff_nm;      // consumed by the next remd_set_system
    return 0;
}

int remd_set_system(remd_handle h, const remd_system_desc* d)
{
    if (h) for (int c = 0; c < 6; ++c) h->fgroup[c] = 0;           // until remd_set_force_groups says otherwise
    if (!h || !d) return remd_fail(h, -1, "remd_set_system: NULL argument");
    if (d->n_atoms <= 0 || !d->mass) return remd_fail(h, -1, "remd_set_system: n_atoms/mass missing");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    remd_regions_release(h);                    // remd_set_alchemical_regions follows the system it belongs to
    h->N = d->n_atoms;
    h->Npad = (d->n_atoms + 63) / 64 * 64;
    std::vector<float> im(h->Npad, 0.f), m(h->Npad, 0.f);
    h->total_mass = 0;
    for (int i = 0; i < h->N; ++i) {
        if (!(d->mass[i] > 0)) return remd_fail(h, -3, "massless particles are not supported");
        m[i] = (float)d->mass[i]; im[i] = (float)(1.0 / d->mass[i]); h->total_mass += d->mass[i];
    }
    int rc;
    if ((rc = upload(h, h->d_invmass, im))) return rc;
    if ((rc = upload(h, h->d_mass, m))) return rc;
    h->n_ext = d->n_ext; h->ext_K = d->ext_K; h->ext_x0 = d->ext_x0; h->ext_U0 = d->ext_U0;
    {
        std::vector<int> ea(d->ext_atoms, d->ext_atoms + d->n_ext);
        if ((rc = upload(h, h->d_ext_atoms, ea))) return rc;
    }
    h->cmm_frequency = d->cmm_frequency;
    if (!h->parent) { if (!h->sysdesc) h->sysdesc = new remd_desc_store(); h->sysdesc->assign(d); h->config_version++; }
    if ((rc = remd_build_constraints(h, d))) return rc;
    if ((rc = remd_build_nonbonded(h, d))) return rc;
    h->has_system = true;
    h->forces_valid = false;
    return 0;
}

int remd_set_states(remd_handle h, int K, const double* beta, const double* lam_s, const double* lam_e, const double* econst)
{
    if (!h || K <= 0 || !beta) return remd_fail(h, -1, "remd_set_states: bad arguments");
    hipSetDevice(h->device);
    h->K = K; h->config_version++;
    h->beta.assign(beta, beta + K);
    h->lam_s.assign(K, 1.0); h->lam_e.assign(K, 1.0); h->econst.assign(K, 0.0);
    if (lam_s) h->lam_s.assign(lam_s, lam_s + K);
    if (lam_e) h->lam_e.assign(lam_e, lam_e + K);
    if (econst) h->econst.assign(econst, econst + K);
    for (int k = 0; k < K; ++k) if (!(h->beta[k] > 0)) return remd_fail(h, -1, "remd_set_states: beta must be > 0");
    int rc;
    if ((rc = upload(h, h->d_beta, h->beta))) return rc;
    if ((rc = upload(h, h->d_lam_s, h->lam_s))) return rc;
    if ((rc = upload(h, h->d_lam_e, h->lam_e))) return rc;
    if ((rc = upload(h, h->d_econst, h->econst))) return rc;
    dfree(h->d_ukl);
    if (h->R_global > 0) {
        REMD_CHECK(h, hipMalloc(&h->d_ukl, sizeof(double) * (size_t)h->R_global * K));
        REMD_CHECK(h, hipMemsetAsync(h->d_ukl, 0, sizeof(double) * (size_t)h->R_global * K, h->stream));
    }
    return 0;
}

int remd_set_integrator(remd_handle h, const char* splitting, double dt, double gamma, int n_steps,
                        int reassign, double tol)
{
    if (!h) return -1;
    if (!(dt > 0) || n_steps < 0 || gamma < 0) return remd_fail(h, -1, "remd_set_integrator: bad parameters");
    int rc = remd_parse_splitting(h, splitting, h->tokens, h->nV, h->nR, h->nO, h->nVg);
    if (rc) return rc;
    h->config_version++;
    h->splitting = splitting; h->dt = dt; h->gamma = gamma; h->n_steps = n_steps; h->reassign = reassign;
    h->constraint_tol = tol > 0 ? tol : 1e-8;
    h->has_integrator = true;
    return 0;
}

int remd_set_force_groups(remd_handle h, const int32_t* groups)
{
    if (!h || !groups) return remd_fail(h, -1, "remd_set_force_groups: bad arguments");
    for (int c = 0; c < 6; ++c) {
        if (groups[c] < 0 || groups[c] > 31) return remd_fail(h, -1, "remd_set_force_groups: force groups are 0 ... 31");
        h->fgroup[c] = groups[c];
    }
    h->config_version++;
    return 0;
}

int remd_set_replica_ids(remd_handle h, const int64_t* ids)
{
    if (!h || h->R <= 0) return remd_fail(h, -1, "remd_set_replica_ids: call remd_set_replicas first");
    hipSetDevice(h->device);
    if (h->d_noise_id) { REMD_CHECK(h, hipStreamSynchronize(h->stream)); hipFree(h->d_noise_id); h->d_noise_id = nullptr; }
    h->noise_id_host.clear(); h->ids_version++;
    if (!ids) return 0;                                   // back to the block's own global indices
    h->noise_id_host.assign(ids, ids + h->R);
    std::vector<unsigned int> v(h->R);
    for (int r = 0; r < h->R; ++r) {
        if (ids[r] < 0 || ids[r] > 0xffffffffll) return remd_fail(h, -1, "remd_set_replica_ids: ids must fit 32 bits");
        v[r] = (unsigned int)ids[r];
    }
    REMD_CHECK(h, hipMalloc(&h->d_noise_id, sizeof(unsigned int) * h->R));
    REMD_CHECK(h, hipMemcpy(h->d_noise_id, v.data(), sizeof(unsigned int) * h->R, hipMemcpyHostToDevice));
    return 0;
}

int remd_set_labels(remd_handle h, const int64_t* labels)
{
    if (!h || !labels || h->R_global <= 0) return remd_fail(h, -1, "remd_set_labels: replicas not set");
    for (int r = 0; r < h->R_global; ++r)
        if (labels[r] < 0 || (h->K > 0 && labels[r] >= h->K)) return remd_fail(h, -1, "remd_set_labels: label out of range");
    h->labels.assign(labels, labels + h->R_global);
    REMD_CHECK(h, hipMemcpyAsync(h->d_labels, h->labels.data(), sizeof(int64_t) * h->R_global, hipMemcpyHostToDevice, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int remd_set_replicas(remd_handle h, int R_global, int r_begin, int R_local, const double* x, const double* v,
                      const double* box, const int64_t* labels)
{
    if (!h || !h->has_system) return remd_fail(h, -1, "remd_set_replicas: call remd_set_system first");
    if (R_global <= 0 || R_local <= 0 || r_begin < 0 || r_begin + R_local > R_global || !labels)
        return remd_fail(h, -1, "remd_set_replicas: bad arguments");
    // (validated before anything of the handle changes: a refused call leaves it as it was)
    if (h->nb_method != REMD_NB_NONE && box)
        for (int r = 0; r < R_local; ++r) for (int k = 0; k < 3; ++k)
            if (!(box[3 * r + k] >= 2.0 * h->cutoff)) return remd_fail(h, -1, "remd_set_replicas: box smaller than twice the cutoff");
    hipSetDevice(h->device);
    const bool realloc = (R_local != h->R) || (R_global != h->R_global) || !h->d_pos;
    if (h->d_noise_id) { hipStreamSynchronize(h->stream); hipFree(h->d_noise_id); h->d_noise_id = nullptr; }     // ids belong to one set of replicas
    h->noise_id_host.clear(); h->ids_version++;
    h->R_global = R_global; h->r_begin = r_begin; h->R = R_local;
    const size_t n = (size_t)R_local * h->Npad;
    if (realloc) {
        dfree(h->d_pos); dfree(h->d_vel); dfree(h->d_force); dfree(h->d_box); dfree(h->d_labels); dfree(h->d_ukl);
        for (int g = 0; g < 4; ++g) dfree(h->d_force_g[g]);
        h->force_g_n = 0;
        // per-replica scratch of the barostat and of the restart attempts is sized by R_local as well
        dfree(h->d_baro); dfree(h->d_box_old); dfree(h->d_baro_x0); dfree(h->d_baro_f0); dfree(h->d_baro_U0); dfree(h->d_baro_acc);
        dfree(h->d_snap_pos); dfree(h->d_snap_vel); dfree(h->d_fin_pos); dfree(h->d_fin_vel); dfree(h->d_snap_box); dfree(h->d_fin_box);
        dfree(h->d_snap_work);
        dfree(h->d_potential); dfree(h->d_epart); dfree(h->d_kinetic); dfree(h->d_nan); dfree(h->d_cmm);
        REMD_CHECK(h, hipMalloc(&h->d_pos, sizeof(float4) * n));
        REMD_CHECK(h, hipMalloc(&h->d_vel, sizeof(float4) * n));
        REMD_CHECK(h, hipMalloc(&h->d_force, sizeof(long long) * 3 * n));
        REMD_CHECK(h, hipMalloc(&h->d_box, sizeof(float) * 4 * R_local));
        REMD_CHECK(h, hipMalloc(&h->d_labels, sizeof(int64_t) * R_global));
        REMD_CHECK(h, hipMalloc(&h->d_potential, sizeof(double) * R_local));
        REMD_CHECK(h, hipMalloc(&h->d_kinetic, sizeof(double) * R_local));
        REMD_CHECK(h, hipMalloc(&h->d_nan, sizeof(int) * R_local));
        REMD_CHECK(h, hipMalloc(&h->d_cmm, sizeof(long long) * 4 * 2 * R_local));
        h->n_epart = remd_nb_required_epart(h);
        REMD_CHECK(h, hipMalloc(&h->d_epart, sizeof(double) * (size_t)h->n_epart * R_local));
        if (h->K > 0) {
            REMD_CHECK(h, hipMalloc(&h->d_ukl, sizeof(double) * (size_t)R_global * h->K));
            REMD_CHECK(h, hipMemsetAsync(h->d_ukl, 0, sizeof(double) * (size_t)R_global * h->K, h->stream));
        }
    }
    std::vector<float4> hp(n, make_float4(0, 0, 0, 0)), hv(n, make_float4(0, 0, 0, 0));
    // (x = NULL: the coordinates follow through remd_copy_replicas; until then every replica holds atoms on a coarse lattice)
    for (int r = 0; r < R_local; ++r)
        for (int i = 0; i < h->N; ++i) {
            const double lattice[3] = { 0.3 * (i % 64), 0.3 * ((i / 64) % 64), 0.3 * (i / 4096) };
            const double* p = x ? x + ((size_t)r * h->N + i) * 3 : lattice;
            hp[(size_t)r * h->Npad + i] = make_float4((float)p[0], (float)p[1], (float)p[2], 0.f);
            if (v) {
                const double* w = v + ((size_t)r * h->N + i) * 3;
                hv[(size_t)r * h->Npad + i] = make_float4((float)w[0], (float)w[1], (float)w[2], 0.f);
            }
        }
    // padding atoms are parked far apart so that they never interact
    for (int r = 0; r < R_local; ++r)
        for (int i = h->N; i < h->Npad; ++i) hp[(size_t)r * h->Npad + i] = make_float4(1e6f + 10.f * i, 1e6f, 1e6f, 0.f);
    std::vector<float> hb(4 * (size_t)R_local, 0.f);
    h->box_host.assign(3 * (size_t)R_local, 0.0);
    for (int r = 0; r < R_local; ++r) for (int k = 0; k < 3; ++k) {
        const double L = box ? box[3 * r + k] : 0.0;
        hb[4 * r + k] = (float)L; h->box_host[3 * r + k] = L;
    }
    REMD_CHECK(h, hipMemcpy(h->d_pos, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    REMD_CHECK(h, hipMemcpy(h->d_vel, hv.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    REMD_CHECK(h, hipMemcpy(h->d_box, hb.data(), sizeof(float) * 4 * R_local, hipMemcpyHostToDevice));
    h->box_uniform = true;
    for (int r = 1; r < R_local; ++r) for (int k = 0; k < 3; ++k) if (hb[4 * r + k] != hb[k]) h->box_uniform = false;
    h->box_version++;
    h->forces_valid = false; h->force_zeroed = false;
    h->cbins_ready = false;              // (bins a chain filled for positions that are gone)
    h->comm_part_current = false;       // (the ranks exchange their blocks again at the next all-gather)
    remd_nb_invalidate_sort(h);         // (a molecule order made for other coordinates overflows the cluster lists)
    // (the mesh buffers follow the replica count; a call that only replaces coordinates keeps them -- one handle per compatibility
    // group re-enters here every iteration, multistate/_engine_pool.py)
    if (h->nb_method == REMD_NB_PME && (realloc || !h->pme)) { int rc = remd_pme_setup(h); if (rc) return rc; }
    return remd_set_labels(h, labels);
}

// rows of [R][Npad] float4 arrays from one handle's slots to another's
__global__ __launch_bounds__(256)
void copy_replica_rows_kernel(int Npad, const int* __restrict__ slots /*[2][n]: dst, src*/, int n, float4* __restrict__ dst_pos,
                              const float4* __restrict__ src_pos, float4* __restrict__ dst_vel, const float4* __restrict__ src_vel,
                              float* __restrict__ dst_box, const float* __restrict__ src_box)
{
    const int k = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = slots[k], s = slots[n + k];
    if (i < Npad) {
        if (dst_pos) dst_pos[(size_t)d * Npad + i] = src_pos[(size_t)s * Npad + i];
        if (dst_vel) dst_vel[(size_t)d * Npad + i] = src_vel[(size_t)s * Npad + i];
    }
    if (dst_box && i < 4) dst_box[4 * d + i] = src_box[4 * s + i];
}

int remd_copy_replicas(remd_handle dst, const int32_t* dst_slot, remd_handle src, const int32_t* src_slot, int32_t n, int32_t what)
{
    if (!dst || !src || !dst->has_system || !src->has_system || !dst->d_pos || !src->d_pos)
        return remd_fail(dst, -1, "remd_copy_replicas: both handles need a system and replicas (remd_set_replicas)");
    if (n < 0 || (n > 0 && (!dst_slot || !src_slot)) || (what & ~7) || !(what & 7)) return remd_fail(dst, -1, "remd_copy_replicas: bad arguments");
    if (dst->device != src->device) return remd_fail(dst, -3, "remd_copy_replicas: the handles live on different devices");
    if (dst->N != src->N || dst->Npad != src->Npad) return remd_fail(dst, -1, "remd_copy_replicas: the handles hold different particle counts");
    if (dst == src) return remd_fail(dst, -1, "remd_copy_replicas: source and destination are the same handle");
    if (n == 0) return 0;
    std::vector<int> slots(2 * (size_t)n);
    std::vector<char> seen(dst->R, 0);
    for (int k = 0; k < n; ++k) {
        if (dst_slot[k] < 0 || dst_slot[k] >= dst->R || src_slot[k] < 0 || src_slot[k] >= src->R) return remd_fail(dst, -1, "remd_copy_replicas: slot out of range");
        if (seen[dst_slot[k]]++) return remd_fail(dst, -1, "remd_copy_replicas: a destination slot is named twice");
        slots[k] = dst_slot[k]; slots[n + k] = src_slot[k];
    }
    hipSetDevice(dst->device);
    REMD_CHECK(dst, hipStreamSynchronize(src->stream));              // what the source computed last is complete
    struct scratch { int* p = nullptr; ~scratch() { if (p) hipFree(p); } } slots_dev;      // (freed on every return path)
    REMD_CHECK(dst, hipMalloc(&slots_dev.p, sizeof(int) * slots.size()));
    int* d_slots = slots_dev.p;
    REMD_CHECK(dst, hipMemcpyAsync(d_slots, slots.data(), sizeof(int) * slots.size(), hipMemcpyHostToDevice, dst->stream));
    const bool pos = what & 1, vel = what & 2, box = what & 4;
    hipLaunchKernelGGL(copy_replica_rows_kernel, dim3((dst->Npad + 255) / 256, n), dim3(256), 0, dst->stream, dst->Npad, d_slots, n,
                       pos ? dst->d_pos : (float4*)nullptr, src->d_pos, vel ? dst->d_vel : (float4*)nullptr, src->d_vel,
                       box ? dst->d_box : (float*)nullptr, src->d_box);
    REMD_CHECK(dst, hipGetLastError());
    if (box) {
        std::vector<float> hb(4 * (size_t)dst->R);
        REMD_CHECK(dst, hipMemcpyAsync(hb.data(), dst->d_box, sizeof(float) * hb.size(), hipMemcpyDeviceToHost, dst->stream));
        REMD_CHECK(dst, hipStreamSynchronize(dst->stream));
        bool changed = dst->box_host.size() != 3 * (size_t)dst->R;
        dst->box_host.resize(3 * (size_t)dst->R, 0.0);
        for (int r = 0; r < dst->R; ++r) for (int k = 0; k < 3; ++k) {
            // (the mirror keeps the doubles a remd_set_replicas gave where the device value still is their rounding)
            if ((float)dst->box_host[3 * r + k] != hb[4 * r + k]) { dst->box_host[3 * r + k] = hb[4 * r + k]; changed = true; }
        }
        if (dst->nb_method != REMD_NB_NONE)
            for (int r = 0; r < dst->R; ++r) for (int k = 0; k < 3; ++k)
                if (dst->box_host[3 * r + k] < 2.0 * dst->cutoff) return remd_fail(dst, -1, "remd_copy_replicas: box smaller than twice the cutoff");
        if (changed) {
            dst->box_uniform = true;
            for (int r = 1; r < dst->R; ++r) for (int k = 0; k < 3; ++k) if (hb[4 * r + k] != hb[k]) dst->box_uniform = false;
            dst->box_version++;
        }
    }
    REMD_CHECK(dst, hipStreamSynchronize(dst->stream));
    if (pos || box) {
        dst->forces_valid = false; dst->force_zeroed = false;
        dst->cbins_ready = false;
        dst->comm_part_current = false;
        remd_nb_invalidate_sort(dst);
    }
    return 0;
}

int remd_set_work_measurement(remd_handle h, int measure_heat, int measure_shadow_work)
{
    if (!h) return -1;
    h->measure_heat = measure_heat ? 1 : 0; h->measure_shadow = measure_shadow_work ? 1 : 0;
    return 0;
}

int remd_reset_work(remd_handle h)
{
    if (!h) return -1;
    hipSetDevice(h->device);
    if (h->R <= 0) return 0;
    int rc = remd_work_buffers(h); if (rc) return rc;
    REMD_CHECK(h, hipMemsetAsync(h->d_work, 0, sizeof(long long) * 4 * h->R, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int remd_get_work(remd_handle h, double* heat, double* shadow_work, int64_t* n_accepted, int64_t* n_trials)
{
    if (!h || h->R <= 0) return remd_fail(h, -1, "remd_get_work: no replicas");
    hipSetDevice(h->device);
    int rc = remd_work_buffers(h); if (rc) return rc;
    std::vector<long long> w(4 * (size_t)h->R);
    REMD_CHECK(h, hipMemcpyAsync(w.data(), h->d_work, sizeof(long long) * w.size(), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->R; ++r) {
        if (heat) heat[r] = (double)w[4 * r] / 16777216.0;
        if (shadow_work) shadow_work[r] = (double)w[4 * r + 1] / 16777216.0;
        if (n_trials) n_trials[r] = w[4 * r + 2];
        if (n_accepted) n_accepted[r] = w[4 * r + 2] - w[4 * r + 3];
    }
    return 0;
}

int remd_get_constraint_stats(remd_handle h, int32_t* max_newton_iterations, int32_t* unconverged)
{
    if (!h || !h->d_sync) return remd_fail(h, -1, "remd_get_constraint_stats: no handle");
    hipSetDevice(h->device);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned int w = 0;
    REMD_CHECK(h, hipMemcpy(&w, h->d_sync + 3, sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (max_newton_iterations) *max_newton_iterations = (int32_t)std::min(w, 8u);
    if (unconverged) *unconverged = w > 8u ? 1 : 0;
    return 0;
}

int remd_set_restart_attempts(remd_handle h, int n)
{
    if (!h || n < 0) return remd_fail(h, -1, "remd_set_restart_attempts: bad arguments");
    h->n_restart_attempts = n; h->config_version++;
    return 0;
}

// The device reports what it cannot raise through a sticky word (d_sync[2]): a wait polled on the device that ran out (1), an
// overfull capped PME bin (2), the integrator chain's momentum barrier (3).  The FIRST fault of a call stays in the word (compare-and-swap
// from 0: what follows a fault is garbage that raises others, e.g. exploded positions overfill a bin after a wait ran out, and the remedy
// must be the first one's); once it is raised the device-side polls of the same call stop waiting.  A raised flag is cleared and the mechanism behind it
// switched off for this handle (events instead of polled flags, two chain launches instead of the barrier, the binning launch
// instead of capped bins), so that the handle keeps working; `retry` says whether the caller runs the work again itself (then
// this is not an error yet).
static int remd_recover_device_flag(remd_ctx* h, unsigned int f, const char* where, bool retry)
{
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->stream2) REMD_CHECK(h, hipStreamSynchronize(h->stream2));
    REMD_CHECK(h, hipMemset(h->d_sync + 2, 0, sizeof(unsigned int)));
    if (f == 6)        // (nothing to switch off: the run cannot continue at this cutoff; the move itself was rejected and restored)
        return remd_fail(h, -2, std::string(where) + ": the Monte Carlo barostat proposed a periodic box smaller than twice the nonbonded cutoff "
                                "(for a PME System: the Coulomb range of the Ewald split, remd_set_coulomb_cutoff) -- the box has shrunk too far "
                                "for this cutoff (OpenMM: \"The periodic box size has decreased to less than twice the nonbonded cutoff\")");
    h->join_deferred = 0; h->cbins_ready = false;
    remd_nb_invalidate_sort(h);
    remd_nb_reset_accumulators(h);
    std::string what;
    if (f == 2) { h->no_chain_bins = true; what = "more atoms in one PME mesh column than the chain-binned layout holds; using the binning launch from now on"; }
    else if (f == 3) { h->no_device_waits = true; what = "the integrator chain's momentum barrier ran out; using two chain launches from now on"; }
    else if (f == 7) { h->no_chain_merge = true; what = "a workgroup's partial momentum sum does not fit the 48-bit payload of the chain's exchange words; summing with two chain launches from now on"; }
    else if (f == 4) { h->no_resident = true; what = "more neighbours per atom than the resident small-system kernel's list holds; using the regular launches from now on"; }
    else { h->no_device_waits = true; h->sync_events = true; what = "a wait polled on the device ran out (fork / join flag never arrived); using events from now on"; }
    if (retry) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "[remd] %s: %s (the work is run again)\n", where, what.c_str()); warned = true; }
        return 0;
    }
    return remd_fail(h, -2, std::string(where) + ": " + what);
}

// ---- phases: one handle's replicas as blocks whose MD steps take turns (remd_internal.h: remd_ctx::phase) -------------------------------
int remd_propagate_many(remd_handle* hs, int32_t n, int64_t iteration, int32_t* nan_flags);

int remd_set_phases(remd_handle h, int32_t n)
{
    if (!h || n < 0 || n > 2) return remd_fail(h, -1, "remd_set_phases: 0 (by rule), 1 (off) or 2");
    h->phases_req = n;
    return 0;
}

int remd_get_phases(remd_handle h, int32_t* n)
{
    if (!h || !n) return -1;
    *n = h->phases_last;
    return 0;
}

// how many blocks the next remd_propagate of this handle runs as
static int phases_for(remd_ctx* h)
{
    if (h->parent) return 1;
    int want = h->phases_req;
    if (const char* e = getenv("REMD_PHASES")) want = atoi(e);
    if (want == 1) return 1;
    // what the blocks' interleaved steps need (everything else takes the one-block path):
    //  * a force evaluation that forks into the mesh and the direct-space stream (PME with overlap), a plain single-group V / R / O program
    //    without work measurement or Metropolization (their launches in between are not worth taking turns with); a Monte Carlo barostat is
    //    each block's own affair (per-replica state and counters travel with the replicas);
    //  * no communicator, no profiling of every class;
    //  * two blocks that are each worth a launch: 3 replicas or more per block unless asked for explicitly.
    if (!h->has_system || !h->has_integrator || !h->sysdesc || !h->sysdesc->valid) return 1;
    // (NoCutoff systems -- a step of three or four dependent small launches on ONE stream -- run as two blocks only when asked to: built and
    // measured at the end of round 6, bit-identical, and no faster (24 x CB7:B2 in vacuum 52.7 / 53.2 it/s, the implicit-solvent dipeptide
    // 45.7 / 45.4): with twice the launches per unit of time the HOST's enqueue rate is the limit, ~4.7 us per launch; profiles/r06_45)
    const bool pme_fork = h->nb_method == REMD_NB_PME && h->overlap && h->stream2;
    const bool small_launches = h->nocutoff && h->nb_method == REMD_NB_NONE && (h->gbsa || h->n_regions > 0 || h->N > 64);
    if (!pme_fork && !(small_launches && want == 2)) return 1;
    if (h->measure_heat || h->measure_shadow || h->profiling == 2 || h->comm) return 1;
    if (h->baro_frequency > 0 && (int)h->pressure_host.size() != h->K) return 1;
    for (char c : h->tokens) if (c != 'V' && c != 'R' && c != 'O') return 1;
    if (h->R < 2) return 1;
    if (want == 2) return 2;
    // by rule: only when the process keeps its streams on few hardware queues.  Measured (profiles/r06_phases_*): two blocks on four
    // hardware queues (two per priority) run the headline step 10 % faster than one block; with a FIFTH queue in the process (the HIP
    // default GPU_MAX_HW_QUEUES = 4 gives the second block's main stream one) they run 55 % slower -- queues beyond the four pipes are
    // time-sliced.  The Python package sets GPU_MAX_HW_QUEUES=2 before the runtime starts; a host that does not gets one block.
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    // (1: the blocks' main streams would share one queue and their direct-space streams the other -- 12.2 it/s on the headline against
    // 18.7; 3: the four streams still get a queue each, 18.7; profiles/r06_45 call 49)
    if (!q || atoi(q) < 2 || atoi(q) > 3) return 1;
    // (two blocks from 6 replicas on: alanine dipeptide R = 4 / 6 / 8 / 12 -> +2 / +7 / +11 / +15 % against one block, R = 2 -> -18 %; 8 x CB7:B2
    // +2 %; profiles/r06_45.  Until the blocks got their own rules for the mesh-column bins and the work-item order the bound was 16.)
    // (setting the blocks up for a call -- two stream synchronisations, the replicas copied in and out -- costs ~0.2 ms: short propagations
    // stay in one block; 24 x alanine dipeptide, ms per call one block / two blocks: 1 step 0.32 / 0.54, 5: 0.87 / 1.06, 20: 3.00 / 2.79,
    // 50: 6.75 / 5.49; profiles/r06_45 call 59)
    if (h->n_steps < 16) return 1;
    return h->R >= 6 ? 2 : 1;
}

// ---- do two streams sit on one hardware queue?  HIP deals streams onto GPU_MAX_HW_QUEUES queues per priority by least use, and a process
// that holds other streams of that priority (torch's NCCL stream is a raised-priority one) can get block B's stream on the queue block A's
// is on: the two blocks would then serialise (107 against 73 ms, profiles/r06_phases_hw_queues.txt "shared pair of streams").  Asked of the
// device: a wavefront that waits 30 us of wall clock on each stream at once -- together they take 30 us on two queues and 60 on one.
__global__ void remd_wait_wallclock_kernel(unsigned long long ticks_100mhz)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(8);
}
static bool streams_share_a_queue(hipStream_t a, hipStream_t b)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) hipEventDestroy(e0); return false; }
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    hipLaunchKernelGGL(remd_wait_wallclock_kernel, dim3(1), dim3(64), 0, a, 100ull);       // (first launch of the kernel: code object load)
    hipStreamSynchronize(a);
    hipEventRecord(e0, a);
    hipLaunchKernelGGL(remd_wait_wallclock_kernel, dim3(1), dim3(64), 0, a, 3000ull);      // 30 us
    hipLaunchKernelGGL(remd_wait_wallclock_kernel, dim3(1), dim3(64), 0, b, 3000ull);
    hipEventRecord(e1, b);
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    float ms = 0.f;
    const bool ok = hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipGetLastError();
    return ok && ms > 0.048f;
}
// a stream like `like_priority_of` (raised priority or not) that does not share its hardware queue with `other`: a stream made while the
// unwanted one is still alive lands on the less used queue; three tries, then whatever came last
static hipStream_t stream_beside(hipStream_t other, bool raised)
{
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::vector<hipStream_t> tried;
    hipStream_t s = nullptr;
    for (int k = 0; k < 3; ++k) {
        s = nullptr;
        if ((raised ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) { s = nullptr; break; }
        if (!streams_share_a_queue(other, s)) break;
        tried.push_back(s);
    }
    for (hipStream_t t : tried) if (t != s) hipStreamDestroy(t);
    return s;
}

static int phase_children(remd_ctx* h, int P)
{
    if ((int)h->phase.size() == P && h->phase_config == h->config_version) return 0;
    for (remd_ctx* c : h->phase) remd_destroy(c);
    h->phase.clear();
    for (int p = 0; p < P; ++p) {
        remd_ctx* c = nullptr;
        int rc = remd_create(&c, h->device, p == 0 ? (void*)h->stream : nullptr);
        if (rc) return remd_fail(h, rc, "phases: remd_create of a block failed");
        h->phase.push_back(c);
        c->parent = h;
        if (p == 0) {            // block 0 launches on this handle's own pair of streams (no hardware queue of its own)
            if (c->stream2) hipStreamDestroy(c->stream2);
            c->stream2 = h->stream2; c->borrowed_stream2 = true;
        } else if (!getenv("REMD_CU_PAIR") && !getenv("REMD_CU_MESH")) {
            // block B's streams must not sit on the hardware queues block A's are on (asked of the device, see above)
            if (c->stream2 && h->stream2 && streams_share_a_queue(h->stream2, c->stream2)) {
                hipStream_t s2 = stream_beside(h->stream2, true);
                if (s2) { hipStreamDestroy(c->stream2); c->stream2 = s2; }
                if (getenv("REMD_MANY_VERBOSE")) fprintf(stderr, "[remd] phases: block B's direct-space stream shared a hardware queue with block A's; re-made (%s)\n", s2 && !streams_share_a_queue(h->stream2, s2) ? "apart now" : "still shared");
            }
            if (c->owns_stream && streams_share_a_queue(h->stream, c->stream)) {
                hipStream_t s1 = stream_beside(h->stream, getenv("REMD_MAIN_PRIO") && atoi(getenv("REMD_MAIN_PRIO")) != 0);
                if (s1) { hipStreamDestroy(c->stream); c->stream = s1; }
                if (getenv("REMD_MANY_VERBOSE")) fprintf(stderr, "[remd] phases: block B's main stream shared a hardware queue with block A's; re-made (%s)\n", s1 && !streams_share_a_queue(h->stream, s1) ? "apart now" : "still shared");
            }
        }
        c->sync_events = h->sync_events; c->overlap = h->overlap;
        c->annihilate_sterics = h->annihilate_sterics; c->coulomb_cutoff = h->coulomb_cutoff; c->rf_unshifted = h->rf_unshifted; c->rf_switch_width = h->rf_switch_width;
        if ((rc = remd_set_system(c, &h->sysdesc->d))) return remd_fail(h, rc, std::string("phases: ") + c->err);
        if ((rc = remd_set_states(c, h->K, h->beta.data(), h->lam_s.data(), h->lam_e.data(), h->econst.data()))) return remd_fail(h, rc, std::string("phases: ") + c->err);
        if ((rc = remd_set_integrator(c, h->splitting.c_str(), h->dt, h->gamma, h->n_steps, h->reassign, h->constraint_tol))) return remd_fail(h, rc, std::string("phases: ") + c->err);
        for (int k = 0; k < 6; ++k) c->fgroup[k] = h->fgroup[k];
        c->seed = h->seed; c->n_restart_attempts = h->n_restart_attempts;
        if (h->gbsa && (rc = remd_gbsa_clone(h, c))) return rc;                    // implicit solvent (gbsa.hip)
        if (h->n_regions > 0 && (rc = remd_regions_clone(h, c))) return rc;       // general alchemical regions (alch_regions.hip)
        if (h->baro_frequency > 0) {
            if ((rc = remd_set_barostat(c, h->K, h->pressure_host.data(), h->baro_frequency))) return remd_fail(h, rc, std::string("phases: ") + c->err);
            c->econst_vref = h->econst_vref;
        }
    }
    h->phase_config = h->config_version;
    return 0;
}

static int remd_propagate_phased(remd_ctx* h, int P, int64_t iteration, int32_t* nan_flags)
{
    int rc = phase_children(h, P);
    if (rc) return rc;
    hipEventRecord(h->ev0, h->stream);
    // what this handle's streams still hold (the energy pass and the mix of the iteration before synchronise before they return; a
    // deferred join of a remd_step does not)
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->stream2) REMD_CHECK(h, hipStreamSynchronize(h->stream2));
    const size_t row = (size_t)h->Npad;
    std::vector<int> r0((size_t)P + 1, 0);
    for (int p = 0; p <= P; ++p) r0[p] = (int)((long long)h->R * p / P);
    for (int p = 0; p < P; ++p) {
        remd_ctx* c = h->phase[p];
        const int cnt = r0[p + 1] - r0[p];
        if (c->R != cnt || c->R_global != h->R_global || c->r_begin != h->r_begin + r0[p] || !c->d_pos) {
            if ((rc = remd_set_replicas(c, h->R_global, h->r_begin + r0[p], cnt, nullptr, nullptr, h->box_host.data() + 3 * (size_t)r0[p], h->labels.data())))
                return remd_fail(h, rc, std::string("phases: ") + c->err);
            c->seen_parent_box = h->box_version; c->seen_parent_ids = -1;
        } else {
            if ((rc = remd_set_labels(c, h->labels.data()))) return remd_fail(h, rc, std::string("phases: ") + c->err);
            if (c->seen_parent_box != h->box_version) {
                // the parent's boxes changed since this block took them (remd_set_replicas / remd_copy_replicas of new boxes): the
                // block's host mirror and everything keyed by its box_version (the PME influence table) follow
                for (int r = 0; r < cnt; ++r) for (int k = 0; k < 3; ++k) c->box_host[3 * (size_t)r + k] = h->box_host[3 * (size_t)(r0[p] + r) + k];
                c->box_version++;
                c->seen_parent_box = h->box_version;
            }
        }
        if (c->seen_parent_ids != h->ids_version) {      // remd_set_replica_ids of the parent (or none: the block's own global indices)
            if ((rc = remd_set_replica_ids(c, h->noise_id_host.empty() ? nullptr : h->noise_id_host.data() + r0[p]))) return remd_fail(h, rc, std::string("phases: ") + c->err);
            c->seen_parent_ids = h->ids_version;
        }
        REMD_CHECK(h, hipMemcpyAsync(c->d_pos, h->d_pos + r0[p] * row, sizeof(float4) * row * cnt, hipMemcpyDeviceToDevice, c->stream));
        REMD_CHECK(h, hipMemcpyAsync(c->d_vel, h->d_vel + r0[p] * row, sizeof(float4) * row * cnt, hipMemcpyDeviceToDevice, c->stream));
        REMD_CHECK(h, hipMemcpyAsync(c->d_box, h->d_box + 4 * (size_t)r0[p], sizeof(float) * 4 * cnt, hipMemcpyDeviceToDevice, c->stream));
        // the forces the last evaluation of this handle left (the energy pass of the iteration before) serve the first kick, as they
        // do without phases: the blocks' trajectories are those of the one-block path bit for bit
        REMD_CHECK(h, hipMemcpyAsync(c->d_force, h->d_force + 3 * (size_t)r0[p] * row, sizeof(long long) * 3 * row * cnt, hipMemcpyDeviceToDevice, c->stream));
        c->forces_valid = h->forces_valid; c->force_zeroed = h->force_zeroed;
        c->cbins_ready = false; c->join_deferred = 0; c->fold_pending = false;
        if (h->baro_frequency > 0) {
            // the barostat's per-replica state (volume step, adaptation window, totals) and the handle's step / attempt counters travel
            // with the replicas: the blocks draw and decide what the one-block path draws and decides (Philox by global replica and attempt)
            if ((rc = remd_barostat_buffers(h)) || (rc = remd_barostat_buffers(c))) return rc;
            REMD_CHECK(h, hipMemcpyAsync(c->d_baro, h->d_baro + 8 * (size_t)r0[p], sizeof(double) * 8 * cnt, hipMemcpyDeviceToDevice, c->stream));
            c->baro_steps = h->baro_steps; c->baro_attempts = h->baro_attempts;
            c->econst_vref = h->econst_vref;
        }
        c->box_uniform = h->box_uniform;
        c->profiling = h->profiling; c->prof_filter = h->prof_filter; c->prof_every = h->prof_every;
        remd_nb_invalidate_sort(c);
    }
    std::vector<int32_t> flags((size_t)h->R, 0);
    rc = remd_propagate_many(h->phase.data(), P, iteration, flags.data());
    if (rc) return remd_fail(h, rc, std::string("phases: ") + (h->phase[0]->err.empty() ? h->phase[P - 1]->err : h->phase[0]->err));
    for (int p = 0; p < P; ++p) {
        remd_ctx* c = h->phase[p];
        const int cnt = r0[p + 1] - r0[p];
        REMD_CHECK(h, hipMemcpyAsync(h->d_pos + r0[p] * row, c->d_pos, sizeof(float4) * row * cnt, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_vel + r0[p] * row, c->d_vel, sizeof(float4) * row * cnt, hipMemcpyDeviceToDevice, h->stream));
        if (h->baro_frequency > 0) {
            REMD_CHECK(h, hipMemcpyAsync(h->d_box + 4 * (size_t)r0[p], c->d_box, sizeof(float) * 4 * cnt, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_baro + 8 * (size_t)r0[p], c->d_baro, sizeof(double) * 8 * cnt, hipMemcpyDeviceToDevice, h->stream));
        }
    }
    if (h->baro_frequency > 0) {
        if (h->phase[0]->baro_attempts != h->baro_attempts) { h->box_uniform = false; h->box_version++; }
        h->baro_steps = h->phase[0]->baro_steps; h->baro_attempts = h->phase[0]->baro_attempts;
    }
    h->forces_valid = false; h->force_zeroed = false; h->cbins_ready = false; h->join_deferred = 0; h->fold_pending = false;
    remd_nb_invalidate_sort(h);
    hipEventRecord(h->ev1, h->stream);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->t_prop = ms;
    if (nan_flags) for (int r = 0; r < h->R; ++r) nan_flags[r] = flags[r];
    return 0;
}

int remd_propagate(remd_handle h, int64_t iteration, int32_t* nan_flags)
{
    if (!h || !h->has_system || !h->has_integrator || h->R <= 0 || h->K <= 0)
        return remd_fail(h, -1, "remd_propagate: system/states/integrator/replicas not all set");
    hipSetDevice(h->device);
    {
        const int P = phases_for(h);
        h->phases_last = P;
        if (P > 1) return remd_propagate_phased(h, P, iteration, nan_flags);
    }
    // every propagation starts its spatial order afresh (the first force evaluation re-sorts, then every resort_interval-th): the
    // schedule of re-sorts, and with it the fp32 order of summation inside the pair kernel, depends on the step index only -- not on
    // how many evaluations other calls made in between, nor on whether the replicas run as one block or as phases
    remd_nb_invalidate_sort(h);
    hipEventRecord(h->ev0, h->stream);
    int rc;
    const int attempts = h->n_restart_attempts;
    const size_t seg = (size_t)h->Npad, bytes = sizeof(float4) * seg * h->R;
    int device_retry = 0;
    {
        // mcmc.py:700-703: the state the move starts from is what a failed attempt is reset to (and what a propagation whose
        // device-side waits failed is run again from)
        if (!h->d_snap_pos) {
            REMD_CHECK(h, hipMalloc(&h->d_snap_pos, bytes)); REMD_CHECK(h, hipMalloc(&h->d_snap_vel, bytes));
            REMD_CHECK(h, hipMalloc(&h->d_fin_pos, bytes)); REMD_CHECK(h, hipMalloc(&h->d_fin_vel, bytes));
            REMD_CHECK(h, hipMalloc(&h->d_snap_box, sizeof(float) * 4 * h->R)); REMD_CHECK(h, hipMalloc(&h->d_fin_box, sizeof(float) * 4 * h->R));
        }
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_box, h->d_box, sizeof(float) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_pos, h->d_pos, bytes, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_vel, h->d_vel, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    // heat / shadow work / Metropolis counters (remd_get_work) of an attempt that is thrown away must not stay accumulated
    // (ADVICE r3): they are snapshotted and restored like the coordinates.  [0 .. 4R): at the start, [4R .. 8R): per replica at success
    const size_t wbytes = sizeof(long long) * 4 * h->R;
    const bool had_work = h->d_work != nullptr && h->work_R == h->R;
    if (!h->d_snap_work) REMD_CHECK(h, hipMalloc(&h->d_snap_work, 2 * wbytes));
    if (had_work) REMD_CHECK(h, hipMemcpyAsync(h->d_snap_work, h->d_work, wbytes, hipMemcpyDeviceToDevice, h->stream));
    else REMD_CHECK(h, hipMemsetAsync(h->d_snap_work, 0, wbytes, h->stream));
    std::vector<int> flags(h->R, 0);
    std::vector<char> pending(h->R, 1);
    for (int a = 0;; ++a) {
        // the attempt number rides in the high bits of the iteration counter => fresh velocities and OU noise
        const int64_t it = iteration + ((int64_t)a << 40);
        if (h->reassign) { if ((rc = remd_assign_velocities(h, it))) return rc; }
        if ((rc = remd_run_steps(h, h->tokens, h->nV, h->nR, h->nO, it, 0, h->n_steps))) return rc;
        if ((rc = remd_check_finite(h))) return rc;
        REMD_CHECK(h, hipMemcpyAsync(flags.data(), h->d_nan, sizeof(int) * h->R, hipMemcpyDeviceToHost, h->stream));
        unsigned int spin_out = 0;
        REMD_CHECK(h, hipMemcpyAsync(&spin_out, h->d_sync + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        remd_nb_tune_resolve(h);
        if (spin_out) {
            // a poll on the device ran out or a capped PME bin overflowed: this propagation's forces cannot be trusted.  The
            // state it started from is still there (snapshot above), so the handle drops the mechanism that failed and the
            // attempt is run again with the same random streams -- once.
            const int rc2 = remd_recover_device_flag(h, spin_out, "remd_propagate", device_retry == 0);
            if (rc2) return rc2;
            ++device_retry;
            REMD_CHECK(h, hipMemcpyAsync(h->d_pos, h->d_snap_pos, bytes, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_vel, h->d_snap_vel, bytes, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_box, h->d_snap_box, sizeof(float) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
            if (h->d_work) REMD_CHECK(h, hipMemcpyAsync(h->d_work, h->d_snap_work, wbytes, hipMemcpyDeviceToDevice, h->stream));
            h->box_version++;
            h->forces_valid = false; h->force_zeroed = false;
            --a;
            continue;
        }
        int n_bad = 0;
        for (int r = 0; r < h->R; ++r) if (pending[r] && flags[r]) ++n_bad;
        if (a == 0 && (n_bad == 0 || attempts == 0)) break;              // the normal case: nothing to restore
        const bool last = (n_bad == 0) || (a == attempts);
        for (int r = 0; r < h->R; ++r) {
            if (!pending[r] || (flags[r] && !last)) continue;             // (a replica that failed for good keeps its NaN state)
            REMD_CHECK(h, hipMemcpyAsync(h->d_fin_pos + r * seg, h->d_pos + r * seg, sizeof(float4) * seg, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_fin_vel + r * seg, h->d_vel + r * seg, sizeof(float4) * seg, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_fin_box + 4 * r, h->d_box + 4 * r, sizeof(float) * 4, hipMemcpyDeviceToDevice, h->stream));
            if (h->d_work) REMD_CHECK(h, hipMemcpyAsync(h->d_snap_work + 4 * (h->R + r), h->d_work + 4 * r, sizeof(long long) * 4, hipMemcpyDeviceToDevice, h->stream));
            if (!flags[r]) pending[r] = 0;
        }
        const float4* src_p = last ? h->d_fin_pos : h->d_snap_pos;
        const float4* src_v = last ? h->d_fin_vel : h->d_snap_vel;
        REMD_CHECK(h, hipMemcpyAsync(h->d_pos, src_p, bytes, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_vel, src_v, bytes, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_box, last ? h->d_fin_box : h->d_snap_box, sizeof(float) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
        if (h->d_work) REMD_CHECK(h, hipMemcpyAsync(h->d_work, h->d_snap_work + (last ? 4 * h->R : 0), wbytes, hipMemcpyDeviceToDevice, h->stream));
        h->box_version++;
        h->forces_valid = false; h->force_zeroed = false;
        if (last) { for (int r = 0; r < h->R; ++r) flags[r] = pending[r] ? 1 : 0; break; }
    }
    hipEventRecord(h->ev1, h->stream);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->t_prop = ms;
    if (nan_flags) for (int r = 0; r < h->R; ++r) nan_flags[r] = flags[r];
    return 0;
}

// Several handles of one device propagated in ONE call from one host thread, their MD steps taking turns (integrate.hip:
// remd_run_steps_many): replicas are independent between two mixes (multistatesampler.py:1296-1297), so while one group's integrator
// chain -- the serial part of a step -- runs, another group's pair and mesh kernels have the chip.  Per-replica results are those of
// remd_propagate on each handle, bit for bit (fixed-point forces, Philox streams keyed by the global replica).  The normal case is handled
// here; a handle that reports a device-side fault or a NaN with restart attempts left is put back to its snapshot and run again alone
// through remd_propagate (whose recovery and restart logic then applies).  nan_flags: the handles' replicas, concatenated.
int remd_propagate_many(remd_handle* hs, int32_t n, int64_t iteration, int32_t* nan_flags)
{
    if (!hs || n < 1) return remd_fail(nullptr, -1, "remd_propagate_many: no handles");
    for (int i = 0; i < n; ++i) {
        remd_ctx* h = hs[i];
        if (!h || !h->has_system || !h->has_integrator || h->R <= 0 || h->K <= 0)
            return remd_fail(h, -1, "remd_propagate_many: system/states/integrator/replicas not all set");
        if (h->device != hs[0]->device) return remd_fail(h, -1, "remd_propagate_many: the handles must live on one device");
        if (h->n_steps != hs[0]->n_steps) return remd_fail(h, -1, "remd_propagate_many: the handles must run the same number of steps");
    }
    if (n == 1) return remd_propagate(hs[0], iteration, nan_flags);
    hipSetDevice(hs[0]->device);
    // Two kernels that each hold workgroups at a barrier over sibling workgroups can deadlock each other when they do not both fit the
    // chip (A's waiting workgroups fill one XCD, B's another, each waiting for siblings the other keeps out: seen as the momentum
    // barrier running out with 2 x 64 alanine replicas = 2 x 576 chain workgroups).  The integrator chains hold one workgroup per CU,
    // so the momentum sum is an in-kernel barrier only while ALL the handles' chain workgroups are resident at once; beyond that it is
    // two launches.  The join polled in the chain's prologue cannot deadlock (the kernels it waits for fit beside the chains) but parks
    // CUs: kept up to four rounds of the chip in total, as for one handle.
    long long chain_wgs = 0;
    for (int i = 0; i < n; ++i) chain_wgs += remd_chain_blocks(hs[i]);
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, hs[0]->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
    const bool barrier_ok = chain_wgs <= n_cu, polls_ok = chain_wgs <= 4ll * n_cu;
    for (int i = 0; i < n; ++i) {
        remd_ctx* h = hs[i];
        hipEventRecord(h->ev0, h->stream);
        const size_t bytes = sizeof(float4) * (size_t)h->Npad * h->R;
        if (!h->d_snap_pos) {
            REMD_CHECK(h, hipMalloc(&h->d_snap_pos, bytes)); REMD_CHECK(h, hipMalloc(&h->d_snap_vel, bytes));
            REMD_CHECK(h, hipMalloc(&h->d_fin_pos, bytes)); REMD_CHECK(h, hipMalloc(&h->d_fin_vel, bytes));
            REMD_CHECK(h, hipMalloc(&h->d_snap_box, sizeof(float) * 4 * h->R)); REMD_CHECK(h, hipMalloc(&h->d_fin_box, sizeof(float) * 4 * h->R));
        }
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_box, h->d_box, sizeof(float) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_pos, h->d_pos, bytes, hipMemcpyDeviceToDevice, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(h->d_snap_vel, h->d_vel, bytes, hipMemcpyDeviceToDevice, h->stream));
        const size_t wbytes = sizeof(long long) * 4 * h->R;
        if (!h->d_snap_work) REMD_CHECK(h, hipMalloc(&h->d_snap_work, 2 * wbytes));
        if (h->d_work != nullptr && h->work_R == h->R) REMD_CHECK(h, hipMemcpyAsync(h->d_snap_work, h->d_work, wbytes, hipMemcpyDeviceToDevice, h->stream));
        else REMD_CHECK(h, hipMemsetAsync(h->d_snap_work, 0, wbytes, h->stream));
        // (REMD_MANY_LEAN=1: no workgroup waits on a CU for another stream -- join by a one-wavefront launch, momentum sum as two
        // launches; measured slower than the polling chain once the handles' streams sit on four hardware queues: 76.1 against 72.9 ms)
        static const bool lean_env = getenv("REMD_MANY_LEAN") && atoi(getenv("REMD_MANY_LEAN")) != 0;
        h->lean_waits = lean_env || !polls_ok;
        h->no_chain_barrier = !barrier_ok;
        remd_nb_invalidate_sort(h);          // as remd_propagate: the schedule of spatial re-sorts restarts with every propagation
    }
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) if (hs[i]->reassign) rc = remd_assign_velocities(hs[i], iteration);
    const auto t_enq0 = std::chrono::steady_clock::now();
    if (!rc) rc = remd_run_steps_many(hs, n, iteration, 0, hs[0]->n_steps);
    const auto t_enq1 = std::chrono::steady_clock::now();
    std::vector<std::vector<int>> flags((size_t)n);
    std::vector<unsigned int> spin((size_t)n, 0u);
    for (int i = 0; i < n && !rc; ++i) {
        remd_ctx* h = hs[i];
        if ((rc = remd_check_finite(h))) break;
        flags[i].assign((size_t)h->R, 0);
        REMD_CHECK(h, hipMemcpyAsync(flags[i].data(), h->d_nan, sizeof(int) * h->R, hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipMemcpyAsync(&spin[i], h->d_sync + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
        hipEventRecord(h->ev1, h->stream);
    }
    for (int i = 0; i < n; ++i) {           // (also after an error: no launch of this call may outlive it)
        remd_ctx* h = hs[i];
        h->lean_waits = false; h->no_chain_barrier = false;
        hipStreamSynchronize(h->stream);
        if (h->stream2) hipStreamSynchronize(h->stream2);
    }
    if (rc) return rc;
    if (getenv("REMD_MANY_VERBOSE"))
        fprintf(stderr, "[remd] remd_propagate_many: %d handles, host enqueue of the steps %.2f ms, until the device was done %.2f ms\n", n,
                1e3 * std::chrono::duration<double>(t_enq1 - t_enq0).count(),
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enq0).count());
    int off = 0;
    for (int i = 0; i < n; ++i) {
        remd_ctx* h = hs[i];
        remd_nb_tune_resolve(h);
        int n_bad = 0;
        for (int r = 0; r < h->R; ++r) n_bad += flags[i][r] ? 1 : 0;
        if (spin[i] || (n_bad > 0 && h->n_restart_attempts > 0)) {
            // the rare cases go through the single-handle path from the state this call started from
            const size_t bytes = sizeof(float4) * (size_t)h->Npad * h->R;
            REMD_CHECK(h, hipMemcpyAsync(h->d_pos, h->d_snap_pos, bytes, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_vel, h->d_snap_vel, bytes, hipMemcpyDeviceToDevice, h->stream));
            REMD_CHECK(h, hipMemcpyAsync(h->d_box, h->d_snap_box, sizeof(float) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
            if (h->d_work) REMD_CHECK(h, hipMemcpyAsync(h->d_work, h->d_snap_work, sizeof(long long) * 4 * h->R, hipMemcpyDeviceToDevice, h->stream));
            h->box_version++;
            h->forces_valid = false; h->force_zeroed = false;
            if (!spin[i]) { remd_nb_invalidate_sort(h); }
            int rc1 = remd_propagate(h, iteration, nan_flags ? nan_flags + off : nullptr);
            if (rc1) return rc1;
        } else {
            float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->t_prop = ms;
            if (nan_flags) for (int r = 0; r < h->R; ++r) nan_flags[off + r] = flags[i][r];
        }
        off += h->R;
    }
    return 0;
}

int remd_set_barostat(remd_handle h, int K, const double* pressure, int frequency)
{
    if (!h) return -1;
    hipSetDevice(h->device);
    if (!pressure || frequency <= 0) { if (h->baro_frequency != 0) h->config_version++; h->baro_frequency = 0; return 0; }
    if (K != h->K) return remd_fail(h, -1, "remd_set_barostat: K differs from remd_set_states");
    std::vector<double> p(pressure, pressure + K);
    for (double v : p) if (!(v == v)) return remd_fail(h, -1, "remd_set_barostat: NaN pressure");
    int rc = upload(h, h->d_pressure, p);
    if (rc) return rc;
    if (h->baro_frequency != frequency || h->pressure_host != p) h->config_version++;     // (the blocks of a phased propagation follow)
    h->baro_frequency = frequency;
    h->pressure_host = p;
    return 0;
}

int remd_set_energy_const_volume(remd_handle h, double reference_volume)
{
    if (!h || !(reference_volume >= 0.0)) return remd_fail(h, -1, "remd_set_energy_const_volume: bad arguments");
    h->econst_vref = reference_volume;
    return 0;
}

int remd_get_boxes(remd_handle h, double* box)
{
    if (!h || !box || h->R <= 0 || !h->d_box) return remd_fail(h, -1, "remd_get_boxes: replicas not set");
    hipSetDevice(h->device);
    std::vector<float> hb(4 * (size_t)h->R);
    REMD_CHECK(h, hipMemcpyAsync(hb.data(), h->d_box, sizeof(float) * hb.size(), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->R; ++r) for (int k = 0; k < 3; ++k) box[3 * r + k] = hb[4 * r + k];
    return 0;
}

int remd_barostat_attempts(remd_handle h, int n_attempts)
{
    if (!h) return -1;
    if (n_attempts < 0) return remd_fail(h, -1, "remd_barostat_attempts: negative n_attempts");
    if (h->R <= 0 || !h->d_pos) return remd_fail(h, -1, "remd_barostat_attempts: replicas not set");
    if (h->baro_frequency <= 0 || !h->d_pressure) return remd_fail(h, -3, "remd_barostat_attempts: no barostat (remd_set_barostat)");
    hipSetDevice(h->device);
    for (int a = 0; a < n_attempts; ++a) {
        int rc = remd_barostat_attempt(h);
        if (rc) return rc;
    }
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    return remd_check_device_flags(h, "remd_barostat_attempts");
}

int remd_get_barostat_stats(remd_handle h, double* volume_scale, int64_t* n_attempted, int64_t* n_accepted)
{
    if (!h || h->R <= 0) return remd_fail(h, -1, "remd_get_barostat_stats: replicas not set");
    hipSetDevice(h->device);
    std::vector<double> st(8 * (size_t)h->R, 0.0);
    if (h->d_baro) {
        REMD_CHECK(h, hipMemcpyAsync(st.data(), h->d_baro, sizeof(double) * st.size(), hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
    }
    for (int r = 0; r < h->R; ++r) {
        if (volume_scale) volume_scale[r] = st[8 * r];
        if (n_attempted) n_attempted[r] = (int64_t)st[8 * r + 3];
        if (n_accepted) n_accepted[r] = (int64_t)st[8 * r + 4];
    }
    return 0;
}

int remd_minimize(remd_handle h, double tolerance, int max_iterations, int32_t* converged, int32_t* n_iterations)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return remd_fail(h, -1, "remd_minimize: system/states/replicas not all set");
    if (!(tolerance >= 0) || max_iterations < 0) return remd_fail(h, -1, "remd_minimize: bad arguments");
    hipSetDevice(h->device);
    return remd_minimize_impl(h, tolerance, max_iterations, converged, n_iterations);
}

int remd_step(remd_handle h, const char* splitting, int64_t iteration, int64_t first_step, int n_steps)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return remd_fail(h, -1, "remd_step: not set up");
    hipSetDevice(h->device);
    std::vector<char> tokens; int nV, nR, nO;
    // a test-hook splitting may consist of a single substep: count with the configured integrator
    std::string s(splitting ? splitting : "");
    tokens.clear();
    for (char c : s) { if (c == ' ') continue; c = (char)toupper(c); if (c != 'V' && c != 'R' && c != 'O' && c != '{' && c != '}') return remd_fail(h, -3, "remd_step: token must be V, R, O, { or }"); tokens.push_back(c); }
    nV = h->nV; nR = h->nR; nO = h->nO;
    int rc = remd_run_steps(h, tokens, nV, nR, nO, iteration, first_step, n_steps);
    if (rc) return rc;
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int remd_ukl_device_ptr(remd_handle h, double** d_ukl)
{
    if (!h || !d_ukl || !h->d_ukl) return remd_fail(h, -1, "remd_ukl_device_ptr: states/replicas not set");
    *d_ukl = h->d_ukl;
    return 0;
}

int remd_compute_energies(remd_handle h, double* d_ukl_rows, double* ukl_host, double* potential_host)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return remd_fail(h, -1, "remd_compute_energies: not set up");
    hipSetDevice(h->device);
    for (int attempt = 0;; ++attempt) {
        hipEventRecord(h->ev0, h->stream);
        int rc;
        h->forces_valid = false;
        // (the spatial order of the energy pass is made for ITS positions: the forces it leaves serve the next propagation's first kick,
        // and their fp32 order of summation must not depend on what ran on this handle before -- one block or phases, remd_set_phases)
        remd_nb_invalidate_sort(h);
        if ((rc = remd_compute_forces(h, true))) return rc;
        double* rows = d_ukl_rows ? d_ukl_rows : h->d_ukl + (size_t)h->r_begin * h->K;
        if ((rc = remd_assemble_ukl(h, rows))) return rc;
        hipEventRecord(h->ev1, h->stream);
        if (ukl_host) REMD_CHECK(h, hipMemcpyAsync(ukl_host, rows, sizeof(double) * (size_t)h->R * h->K, hipMemcpyDeviceToHost, h->stream));
        if (potential_host) REMD_CHECK(h, hipMemcpyAsync(potential_host, h->d_potential, sizeof(double) * h->R, hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->t_energy = ms;
        rc = remd_check_device_flags(h, "remd_compute_energies", attempt == 0);
        if (rc != 1) return rc;                          // 1: the failed mechanism is off now, evaluate again
    }
}

static int ensure_mix_buffers(remd_ctx* h, int R, int K)
{
    if (h->stats_K != K || !h->d_nacc) {
        dfree(h->d_nacc); dfree(h->d_nprop); dfree(h->d_logw);
        REMD_CHECK(h, hipMalloc(&h->d_nacc, sizeof(unsigned long long) * (size_t)K * K));
        REMD_CHECK(h, hipMalloc(&h->d_nprop, sizeof(unsigned long long) * (size_t)K * K));
        REMD_CHECK(h, hipMalloc(&h->d_logw, sizeof(double) * K));
        h->stats_K = K;
    }
    if (h->ukl_tmp_n < (size_t)R * K) {
        dfree(h->d_ukl_tmp); dfree(h->d_logP);
        REMD_CHECK(h, hipMalloc(&h->d_ukl_tmp, sizeof(double) * (size_t)R * K));
        REMD_CHECK(h, hipMalloc(&h->d_logP, sizeof(double) * (size_t)R * K));
        h->ukl_tmp_n = (size_t)R * K;
    }
    return 0;
}

static int mix_common(remd_ctx* h, int scheme, int64_t iteration, int R, int K, int ld, const double* d_ukl, int64_t* labels,
                      int64_t* n_accepted, int64_t* n_proposed, const double* log_weights, double* sams_log_P,
                      int64_t n_attempts, int64_t* d_labels)
{
    int rc;
    hipEventRecord(h->ev0, h->stream);
    const std::vector<int64_t> labels_in(labels, labels + R);
  again:
    REMD_CHECK(h, hipMemcpyAsync(d_labels, labels_in.data(), sizeof(int64_t) * R, hipMemcpyHostToDevice, h->stream));
    if (log_weights) REMD_CHECK(h, hipMemcpyAsync(h->d_logw, log_weights, sizeof(double) * K, hipMemcpyHostToDevice, h->stream));
    if ((rc = remd_mix_launch(h, scheme, iteration, R, K, ld, d_ukl, d_labels, h->d_nacc, h->d_nprop,
                              log_weights ? h->d_logw : nullptr, h->d_logP, n_attempts))) { h->mix_no_pre = false; return rc; }
    // the hoisted swap-all path leaves its overflow flag on the device: read here, with the results (one synchronisation per call)
    unsigned pre_overflow = 0;
    if (const unsigned* flag = remd_mix_pending_flag(h))
        REMD_CHECK(h, hipMemcpyAsync(&pre_overflow, flag, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipMemcpyAsync(labels, d_labels, sizeof(int64_t) * R, hipMemcpyDeviceToHost, h->stream));
    if (n_accepted) REMD_CHECK(h, hipMemcpyAsync(n_accepted, h->d_nacc, sizeof(int64_t) * (size_t)K * K, hipMemcpyDeviceToHost, h->stream));
    if (n_proposed) REMD_CHECK(h, hipMemcpyAsync(n_proposed, h->d_nprop, sizeof(int64_t) * (size_t)K * K, hipMemcpyDeviceToHost, h->stream));
    if (sams_log_P && scheme == REMD_MIX_SAMS_GLOBAL)
        REMD_CHECK(h, hipMemcpyAsync(sams_log_P, h->d_logP, sizeof(double) * (size_t)R * K, hipMemcpyDeviceToHost, h->stream));
    hipEventRecord(h->ev1, h->stream);
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    if (pre_overflow && !h->mix_no_pre) {        // (nothing was touched: the serial kernel and the statistics return at once on the flag)
        h->mix_no_pre = true;
        goto again;
    }
    h->mix_no_pre = false;
    float ms = 0; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->t_mix = ms;
    if (scheme == REMD_MIX_SWAP_ALL && n_accepted && n_proposed) {
        double a = 0.0, p = 0.0;
        for (size_t q = 0; q < (size_t)K * K; ++q) { a += (double)n_accepted[q]; p += (double)n_proposed[q]; }
        if (p > 0.0) h->mix_acc_rate = a / p;
    }
    return 0;
}

int remd_mix(remd_handle h, int scheme, int64_t iteration, int R, int K, const double* d_ukl, int ld, int64_t* labels,
             int64_t* n_accepted, int64_t* n_proposed, const double* log_weights, double* sams_log_P)
{
    if (!h || !labels || R <= 0 || K <= 0) return remd_fail(h, -1, "remd_mix: bad arguments");
    hipSetDevice(h->device);
    if (!d_ukl) {
        if (!h->d_ukl || R != h->R_global || K > h->K) return remd_fail(h, -1, "remd_mix: handle has no matching u_kl matrix");
        d_ukl = h->d_ukl; ld = h->K;
    }
    int rc = ensure_mix_buffers(h, R, K);
    if (rc) return rc;
    int64_t* d_labels = nullptr;
    bool own = false;
    if (h->d_labels && R == h->R_global) d_labels = h->d_labels;
    else { REMD_CHECK(h, hipMalloc(&d_labels, sizeof(int64_t) * R)); own = true; }
    rc = mix_common(h, scheme, iteration, R, K, ld, d_ukl, labels, n_accepted, n_proposed, log_weights, sams_log_P, -1, d_labels);
    if (own) hipFree(d_labels);
    if (!rc && !own) h->labels.assign(labels, labels + R);
    return rc;
}

int remd_mix_host(remd_handle h, int scheme, int64_t iteration, int R, int K, const double* ukl_host, int64_t* labels,
                  int64_t* n_accepted, int64_t* n_proposed, const double* log_weights, double* sams_log_P,
                  int64_t n_attempts)
{
    if (!h || !labels || !ukl_host || R <= 0 || K <= 0) return remd_fail(h, -1, "remd_mix_host: bad arguments");
    hipSetDevice(h->device);
    int rc = ensure_mix_buffers(h, R, K);
    if (rc) return rc;
    REMD_CHECK(h, hipMemcpyAsync(h->d_ukl_tmp, ukl_host, sizeof(double) * (size_t)R * K, hipMemcpyHostToDevice, h->stream));
    int64_t* d_labels = nullptr;
    REMD_CHECK(h, hipMalloc(&d_labels, sizeof(int64_t) * R));
    rc = mix_common(h, scheme, iteration, R, K, K, h->d_ukl_tmp, labels, n_accepted, n_proposed, log_weights, sams_log_P,
                    n_attempts, d_labels);
    hipFree(d_labels);
    return rc;
}

int remd_get_replicas(remd_handle h, double* x, double* v, double* potential, double* kinetic)
{
    if (!h || h->R <= 0) return remd_fail(h, -1, "remd_get_replicas: no replicas");
    hipSetDevice(h->device);
    const size_t n = (size_t)h->R * h->Npad;
    std::vector<float4> buf(n);
    if (x) {
        REMD_CHECK(h, hipMemcpyAsync(buf.data(), h->d_pos, sizeof(float4) * n, hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        for (int r = 0; r < h->R; ++r) for (int i = 0; i < h->N; ++i) {
            const float4 p = buf[(size_t)r * h->Npad + i]; double* o = x + ((size_t)r * h->N + i) * 3;
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
        }
    }
    if (v) {
        REMD_CHECK(h, hipMemcpyAsync(buf.data(), h->d_vel, sizeof(float4) * n, hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        for (int r = 0; r < h->R; ++r) for (int i = 0; i < h->N; ++i) {
            const float4 p = buf[(size_t)r * h->Npad + i]; double* o = v + ((size_t)r * h->N + i) * 3;
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
        }
    }
    if (potential) {
        int rc = remd_compute_forces(h, true); if (rc) return rc;
        REMD_CHECK(h, hipMemcpyAsync(potential, h->d_potential, sizeof(double) * h->R, hipMemcpyDeviceToHost, h->stream));
    }
    if (kinetic) {
        int rc = remd_kinetic_energy(h); if (rc) return rc;
        REMD_CHECK(h, hipMemcpyAsync(kinetic, h->d_kinetic, sizeof(double) * h->R, hipMemcpyDeviceToHost, h->stream));
    }
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int remd_get_forces(remd_handle h, double* f)
{
    if (!h || h->R <= 0 || !f) return remd_fail(h, -1, "remd_get_forces: bad arguments");
    hipSetDevice(h->device);
    for (int attempt = 0;; ++attempt) {
        h->forces_valid = false;
        int rc = remd_compute_forces(h, false); if (rc) return rc;
        const size_t n = (size_t)h->R * 3 * h->Npad;
        std::vector<long long> buf(n);
        REMD_CHECK(h, hipMemcpyAsync(buf.data(), h->d_force, sizeof(long long) * n, hipMemcpyDeviceToHost, h->stream));
        REMD_CHECK(h, hipStreamSynchronize(h->stream));
        for (int r = 0; r < h->R; ++r) for (int i = 0; i < h->N; ++i) for (int k = 0; k < 3; ++k)
            f[((size_t)r * h->N + i) * 3 + k] = (double)buf[((size_t)r * 3 + k) * h->Npad + i] / REMD_FORCE_SCALE;
        rc = remd_check_device_flags(h, "remd_get_forces", attempt == 0);
        if (rc != 1) return rc;
    }
}

// the device reports what it cannot raise: a cross-stream poll or the chain's barrier that ran out, an overfull PME bin
// (d_sync[2], sticky).  Called behind the stream synchronisation of every entry point that evaluates forces.
static int remd_check_device_flags(remd_ctx* h, const char* where, bool may_retry)
{
    // may_retry: the caller's work changed nothing but its outputs (an energy / force evaluation): after the failed mechanism is
    // switched off it is simply run again (return 1); otherwise the flag is an error
    unsigned int f = 0;
    REMD_CHECK(h, hipMemcpy(&f, h->d_sync + 2, sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (!f) return 0;
    const int rc = remd_recover_device_flag(h, f, where, may_retry);
    return rc ? rc : 1;
}

int remd_test_fft3d(remd_handle h, int nx, int ny, int nz, float* data, int inverse)
{
    if (!h || !data) return remd_fail(h, -1, "remd_test_fft3d: bad arguments");
    hipSetDevice(h->device);
    return remd_test_fft3d_impl(h, nx, ny, nz, data, inverse);
}


int remd_get_energy_components(remd_handle h, double* out)
{
    if (!h || !out || h->R <= 0) return remd_fail(h, -1, "remd_get_energy_components: bad arguments");
    hipSetDevice(h->device);
    int rc = remd_compute_forces(h, true); if (rc) return rc;
    std::vector<double> ep((size_t)h->n_epart * h->R);
    REMD_CHECK(h, hipMemcpyAsync(ep.data(), h->d_epart, sizeof(double) * ep.size(), hipMemcpyDeviceToHost, h->stream));
    REMD_CHECK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < h->R; ++r) {
        for (int k = 0; k < 8; ++k) out[9 * r + k] = ep[(size_t)r * h->n_epart + k];
        double nb = 0; for (int k = 8; k < h->n_epart; ++k) nb += ep[(size_t)r * h->n_epart + k];
        out[9 * r + 8] = nb;
    }
    return 0;
}

int remd_sync(remd_handle h) { if (!h) return -1; hipSetDevice(h->device); REMD_CHECK(h, hipStreamSynchronize(h->stream)); return 0; }

int remd_last_timing(remd_handle h, double* p, double* e, double* m)
{
    if (!h) return -1;
    if (p) *p = h->t_prop; if (e) *e = h->t_energy; if (m) *m = h->t_mix;
    return 0;
}

static void resolve_profile(remd_ctx* h)
{
    for (auto& p : h->prof_pending) {
        hipEventSynchronize(p.b);
        float ms = 0; hipEventElapsedTime(&ms, p.a, p.b);
        auto& e = h->prof[p.name]; e.n += 1; e.ms += ms;
        hipEventDestroy(p.a); hipEventDestroy(p.b);
    }
    h->prof_pending.clear();
}

int remd_profile_enable(remd_handle h, int on) { if (!h) return -1; h->profiling = on < 0 ? 0 : (on > 2 ? 2 : on); return 0; }
int remd_profile_filter(remd_handle h, const char* kernel_class) { if (!h || !kernel_class) return -1; h->prof_filter = kernel_class; return 0; }
int remd_profile_reset(remd_handle h)
{
    if (!h) return -1;
    for (remd_ctx* c : h->phase) remd_profile_reset(c);
    resolve_profile(h); h->prof.clear();
    if (h->d_chain_own) { hipStreamSynchronize(h->stream); hipMemset(h->d_chain_own, 0, 40 * sizeof(unsigned long long)); }
    return 0;
}
int remd_profile_get(remd_handle h, const char* name, int64_t* n, double* ms)
{
    if (!h || !name) return -1;
    if (!h->phase.empty() && h->phases_last > 1) {
        // the launches of a phased propagation are the blocks': their sums (a launch covers one block's replicas)
        int64_t nn = 0; double mm = 0.0;
        for (remd_ctx* c : h->phase) { int64_t a = 0; double b = 0.0; int rc = remd_profile_get(c, name, &a, &b); if (rc) return rc; nn += a; mm += b; }
        { remd_ctx* self = h; std::vector<remd_ctx*> none; none.swap(self->phase); int64_t a = 0; double b = 0.0; remd_profile_get(self, name, &a, &b); none.swap(self->phase); nn += a; mm += b; }
        if (n) *n = nn; if (ms) *ms = mm;
        return 0;
    }
    resolve_profile(h);
    if (std::string(name) == "integrate_chain_own") {
        // the integrator chain's own time (flag seen -> end, workgroup (0, 0)), from wall-clock stamps taken inside the kernel (100 MHz)
        unsigned long long v[2] = {0ull, 0ull};
        if (h->d_chain_own) { hipStreamSynchronize(h->stream); hipMemcpy(v, h->d_chain_own, sizeof(v), hipMemcpyDeviceToHost); }
        if (n) *n = (int64_t)v[1];
        if (ms) *ms = (double)v[0] * 1e-5;
        return 0;
    }
    if (std::string(name).rfind("integrate_chain_seg", 0) == 0) {
        // per-segment stamps of workgroup (0, 0) (a library built with -DCHAIN_STAMPS, tools/chain_segments.py): slot k of
        // [prologue, token 0 .. token 31 (an 'M' barrier counts as its own token), epilogue stores + binning]
        const int k = atoi(name + 19);
        unsigned long long v[40] = {};
        if (h->d_chain_own && k >= 0 && k < 38) { hipStreamSynchronize(h->stream); hipMemcpy(v, h->d_chain_own, sizeof(v), hipMemcpyDeviceToHost); }
        if (n) *n = (int64_t)v[1];
        if (ms) *ms = (double)v[2 + (k < 0 || k >= 38 ? 0 : k)] * 1e-5;
        return 0;
    }
    auto it = h->prof.find(name);
    if (n) *n = it == h->prof.end() ? 0 : it->second.n;
    if (ms) *ms = it == h->prof.end() ? 0.0 : it->second.ms;
    return 0;
}

} // extern "C"
