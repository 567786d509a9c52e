/*
 * remd_cpu.cpp — libremd_cpu.so: the CPU implementation of include/remd_hip.h.
 *
 * TEST INFRASTRUCTURE AND CPU BASELINE ONLY.  It lives under oracle/, is never imported by the product package
 * (openmmtools_amd loads libremd_hip.so or raises), and exists for two purposes (SURVEY.md 8(b) last line, 8(d);
 * BASELINE.md section 3):
 *   1. the CPU baseline timed beside the MI355X numbers in bench.py (`cpu_baseline`): the same replica-exchange
 *      iteration  mix -> propagate -> u_kl  (openmmtools/multistate/multistatesampler.py:766-804) through the same
 *      C ABI, f64, OpenMP over replicas — the shape of the reference's own CPU path, which distributes replicas over
 *      mpiplus ranks one replica at a time (multistatesampler.py:1296-1297, 1448-1449);
 *   2. a second, compiled checker for the ABI: the -m "not gpu" tests drive it with the very ctypes class the GPU tests
 *      use and compare it with the f64 Python oracle.
 *
 * Algorithms (each restated from the same reference lines as the HIP kernels; the Python oracle oracle/md_oracle.py /
 * oracle/forcefield.py is the independent statement they are both tested against):
 *   Langevin splitting V/R/O           openmmtools/integrators.py:1404-1460, constants :1139-1149
 *   velocity reassignment              openmmtools/mcmc.py:710-711
 *   NaN restart attempts               openmmtools/mcmc.py:706-759
 *   reduced potential / u_kl           openmmtools/states.py:1908-1917, 911-992; paralleltempering.py:206-215
 *   mixing                             oracle/mix_oracle.c (replicaexchange.py:294-406, sams.py:477-501), linked in
 *   forces: harmonic bond / angle, periodic torsion, LJ + switch, reaction field or Ewald direct + smooth PME
 *   (order 5) + exclusion correction + self/background terms, 1-4 exceptions, dispersion correction, soft-core
 *   alchemical sterics and lambda-scaled alchemical charges (alchemy.py:1379-1388, 1675-1680)
 *   constraints: iterative SHAKE / RATTLE per rigid cluster to 1e-12 (the HIP engine uses analytic SETTLE)
 * Exact Verlet lists (cutoff + skin, rebuilt when an atom has moved skin/2), cell-list construction.
 *
 * The Monte Carlo barostat (remd_set_barostat / remd_barostat_attempts) and FIRE minimisation (remd_minimize) follow the same
 * restatements as the Python oracle (OracleBarostat, OracleFIRE); only remd_roof_microbench (a GPU measurement) returns -3.
 */
#include "../../include/remd_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {
void oracle_draw(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t t, uint32_t* out);
void oracle_mix_swap_all(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl, int64_t* labels, int64_t* n_acc,
                         int64_t* n_prop, int64_t n_attempts);
void oracle_mix_swap_neighbors(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl, int64_t* labels,
                               int64_t* n_acc, int64_t* n_prop);
void oracle_sams_global_jump(uint64_t seed, int64_t iteration, int R, int K, const double* u_kl, const double* log_w,
                             int64_t* labels, int64_t* n_acc, int64_t* n_prop, double* log_P);
}

namespace {

constexpr double ONE_4PI_EPS0 = 138.93545764438198;     // openmmtools/constants.py:12-14
constexpr double PI = 3.14159265358979323846;
constexpr int PME_ORDER = 5;
constexpr double SKIN = 0.12;                            // nm, Verlet buffer
constexpr double CONSTRAINT_TOL = 1e-12;
typedef std::complex<double> cplx;

// ------------------------------------------------------------------------------------------------------------------
// mixed-radix complex FFT (2, 3, 4, 5 and a generic butterfly), decimation in time, out of place
// ------------------------------------------------------------------------------------------------------------------
struct FFT1D {
    int n = 0;
    std::vector<int> factors;      // (radix, remaining) pairs
    std::vector<cplx> tw;          // exp(-2 pi i k / n)
    std::vector<cplx> scratch;
    explicit FFT1D(int n_) : n(n_) {
        int m = n, p = 4;
        while (m > 1) {
            while (m % p) { if (p == 4) p = 2; else if (p == 2) p = 3; else p += 2; if (p * p > m) p = m; }
            m /= p; factors.push_back(p); factors.push_back(m);
        }
        tw.resize(n);
        for (int k = 0; k < n; ++k) tw[k] = cplx(cos(-2.0 * PI * k / n), sin(-2.0 * PI * k / n));
        scratch.resize(n);
    }
    void bfly_generic(cplx* out, size_t fstride, int m, int p) {
        std::vector<cplx> s(p);
        for (int u = 0; u < m; ++u) {
            int k = u;
            for (int q = 0; q < p; ++q) { s[q] = out[k]; k += m; }
            k = u;
            for (int q1 = 0; q1 < p; ++q1) {
                size_t twidx = 0;
                cplx acc = s[0];
                for (int q = 1; q < p; ++q) { twidx += fstride * k; if (twidx >= (size_t)n) twidx %= n; acc += s[q] * tw[twidx]; }
                out[k] = acc; k += m;
            }
        }
    }
    void bfly2(cplx* out, size_t fstride, int m) {
        for (int k = 0; k < m; ++k) { const cplx t = out[k + m] * tw[k * fstride]; out[k + m] = out[k] - t; out[k] += t; }
    }
    void bfly3(cplx* out, size_t fstride, int m) {
        const double s3 = tw[fstride * m].imag();          // -sin(2 pi / 3)
        for (int k = 0; k < m; ++k) {
            const cplx a = out[k + m] * tw[k * fstride], b = out[k + 2 * m] * tw[2 * k * fstride];
            const cplx sum = a + b, dif = (a - b) * s3;
            const cplx h = out[k] - 0.5 * sum;
            out[k] += sum;
            out[k + m] = cplx(h.real() - dif.imag(), h.imag() + dif.real());
            out[k + 2 * m] = cplx(h.real() + dif.imag(), h.imag() - dif.real());
        }
    }
    void bfly4(cplx* out, size_t fstride, int m) {
        for (int k = 0; k < m; ++k) {
            const cplx a = out[k], b = out[k + m] * tw[k * fstride], c = out[k + 2 * m] * tw[2 * k * fstride],
                       d = out[k + 3 * m] * tw[3 * k * fstride];
            const cplx t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            const cplx jt3(t3.imag(), -t3.real());          // -i * t3
            out[k] = t0 + t2; out[k + 2 * m] = t0 - t2;
            out[k + m] = t1 + jt3; out[k + 3 * m] = t1 - jt3;
        }
    }
    void bfly5(cplx* out, size_t fstride, int m) {
        const cplx ya = tw[fstride * m], yb = tw[2 * fstride * m];
        for (int k = 0; k < m; ++k) {
            const cplx s0 = out[k], s1 = out[k + m] * tw[k * fstride], s2 = out[k + 2 * m] * tw[2 * k * fstride],
                       s3 = out[k + 3 * m] * tw[3 * k * fstride], s4 = out[k + 4 * m] * tw[4 * k * fstride];
            const cplx s7 = s1 + s4, s10 = s1 - s4, s8 = s2 + s3, s9 = s2 - s3;
            out[k] = s0 + s7 + s8;
            const cplx s5(s0.real() + s7.real() * ya.real() + s8.real() * yb.real(), s0.imag() + s7.imag() * ya.real() + s8.imag() * yb.real());
            const cplx s6(s10.imag() * ya.imag() + s9.imag() * yb.imag(), -s10.real() * ya.imag() - s9.real() * yb.imag());
            out[k + m] = s5 - s6; out[k + 4 * m] = s5 + s6;
            const cplx s11(s0.real() + s7.real() * yb.real() + s8.real() * ya.real(), s0.imag() + s7.imag() * yb.real() + s8.imag() * ya.real());
            const cplx s12(-s10.imag() * yb.imag() + s9.imag() * ya.imag(), s10.real() * yb.imag() - s9.real() * ya.imag());
            out[k + 2 * m] = s11 + s12; out[k + 3 * m] = s11 - s12;
        }
    }
    void work(cplx* out, const cplx* in, size_t fstride, const int* f) {
        const int p = f[0], m = f[1];
        if (m == 1) for (int k = 0; k < p; ++k) out[k] = in[k * fstride];
        else for (int k = 0; k < p; ++k) work(out + (size_t)k * m, in + k * fstride, fstride * p, f + 2);
        switch (p) {
            case 2: bfly2(out, fstride, m); break;
            case 3: bfly3(out, fstride, m); break;
            case 4: bfly4(out, fstride, m); break;
            case 5: bfly5(out, fstride, m); break;
            default: bfly_generic(out, fstride, m, p);
        }
    }
    // forward: sum x e^{-2 pi i jk/n}; inverse: e^{+...}, both unnormalised.  in and out must not alias.
    void transform(const cplx* in, cplx* out, bool inverse) {
        if (n == 1) { out[0] = in[0]; return; }
        if (!inverse) { work(out, in, 1, factors.data()); return; }
        for (int k = 0; k < n; ++k) scratch[k] = std::conj(in[k]);
        work(out, scratch.data(), 1, factors.data());
        for (int k = 0; k < n; ++k) out[k] = std::conj(out[k]);
    }
};

// ------------------------------------------------------------------------------------------------------------------
// random numbers: stream layout of openmmtools_amd/csrc/rng.h (DESIGN.md "RNG stream spec")
// ------------------------------------------------------------------------------------------------------------------
inline double u23(uint32_t w) { return ((double)(w >> 9) + 0.5) / 8388608.0; }
inline void gaussians3(uint64_t seed, uint32_t stream, int atom, int replica, uint64_t t, double g[3])
{
    uint32_t w[4];
    oracle_draw(seed, stream, (uint32_t)atom, (uint32_t)replica, t, w);
    const double r1 = sqrt(-2.0 * log(u23(w[0]))), r2 = sqrt(-2.0 * log(u23(w[2])));
    const double a1 = 2.0 * PI * u23(w[1]), a2 = 2.0 * PI * u23(w[3]);
    g[0] = r1 * cos(a1); g[1] = r1 * sin(a1); g[2] = r2 * cos(a2);
}

struct Cluster { int n_atoms; int atoms[4]; int nc; int ci[3], cj[3]; double d[3]; int settle; };   // local indices in ci/cj; settle: rigid 3-site water

// ------------------------------------------------------------------------------------------------------------------
// system description (copied from remd_system_desc)
// ------------------------------------------------------------------------------------------------------------------
struct System {
    int N = 0;
    std::vector<double> mass, invm;
    std::vector<int> ext_atoms; double ext_K = 0, ext_x0 = 0, ext_U0 = 0;
    std::vector<int> bond_atoms, angle_atoms, torsion_atoms, exc_atoms;
    std::vector<double> bond_params, angle_params, torsion_params, exc_params;
    int method = 0; double rc = 0, rs = -1, rf_eps = 78.3, alpha = 0; int grid[3] = {0, 0, 0}; int use_disp = 0;
    bool annihilate = false;  // AlchemicalRegion.annihilate_sterics (remd_set_alchemical_options)
    // GBSA (remd_set_gbsa): OBC2 + ACE as the reference's alchemical factory writes it (alchemy.py:2144-2225)
    struct GB { int n = 0; std::vector<double> q, R, sc; std::vector<char> alch; double tau = 0; bool sasa = true; } gb;
    bool nocut = false;       // NonbondedForce.NoCutoff (REMD_NB_NOCUTOFF): every pair, no box; `method` stays 0 (nothing periodic)
    bool rf_unshifted = false; double rf_switch_width = 0;   // remd_set_reaction_field: c_rf = 0, pair term switched (forces.py:1110-1150)
    double rcc = 0;       // range of the Ewald direct-space sum (remd_set_coulomb_cutoff); = rc unless the host split the sum elsewhere
    std::vector<double> q, sig, eps;
    std::vector<char> alch;
    bool has_charge = false, has_alch = false;
    double sc_alpha = 0.5, sc_a = 1, sc_b = 1, sc_c = 6;
    std::vector<Cluster> clusters;
    int cmm = 0;
    double total_mass = 0, disp_coeff = 0;
    std::vector<std::vector<int>> excl;           // per atom, sorted partner list (exceptions are excluded from the pair loop)
    std::vector<double> bmod[3];                  // B-spline moduli of the PME mesh
    int n_dof = 0;
    std::vector<std::vector<int>> molecules;      // connected components over exceptions, bonds, constraints (barostat scaling units)
    double settle_ra = 0, settle_rb = 0, settle_rc = 0, settle_dHH = 0, settle_mO = 0, settle_mH = 0;
    // general alchemical regions (remd_set_alchemical_regions): the custom forces of alchemy.py:1539-2038
    struct RegionClass { int kind, a, b, P; };   // (environment, a) / (a, a) / (a, b) interacting; P: region of the soft-core constants
    struct Regions {
        int n = 0, K = 0;
        std::vector<int> region_of, alch, annihilate, cls_of, exc_atoms, exc_cls;
        std::vector<double> softcore, q, sig, eps, exc_params, ls, le;
        std::vector<RegionClass> classes;
        std::vector<std::vector<char>> skip;      // per alchemical atom: candidates that are never evaluated
        bool elec = false; double alpha = 0, krf = 0, crf = 0, rs_e = -1;
        bool consistent_exc = false;              // the exceptions' electrostatics with the pairs' expression (alchemy.py:1456-1461)
        // exact PME treatment (remd_alch_regions_desc.exact_pme): the alchemical atoms' charges (restored in System::q) and the charge
        // products of the exceptions that touch a region count times the region's lambda_electrostatics inside the whole Ewald sum
        // softened bonded terms (lambda_bonds / lambda_angles / lambda_torsions of their region): atoms, parameters, region (1-based)
        std::vector<int> bond_atoms, angle_atoms, torsion_atoms, bond_region, angle_region, torsion_region;
        std::vector<double> bond_params, angle_params, torsion_params, bl;      // bl: [K][3][n]
        bool exact = false;
        std::vector<int> exc_region;              // per exception of the System: region whose lambda scales its charge product (0: none)
    } reg;
};

double M_spline(int o, double u) {                // cardinal B-spline of order o at u
    if (o == 2) return (u < 0 || u > 2) ? 0.0 : 1.0 - fabs(u - 1.0);
    return u / (o - 1) * M_spline(o - 1, u) + (o - u) / (o - 1) * M_spline(o - 1, u - 1.0);
}

std::vector<double> bspline_moduli(int n) {
    std::vector<double> w(PME_ORDER - 1), bm(n);
    for (int k = 0; k < PME_ORDER - 1; ++k) w[k] = M_spline(PME_ORDER, k + 1.0);
    for (int m = 0; m < n; ++m) {
        double c = 0, s = 0;
        for (int k = 0; k < PME_ORDER - 1; ++k) { const double a = 2.0 * PI * m * k / n; c += w[k] * cos(a); s += w[k] * sin(a); }
        bm[m] = c * c + s * s;
    }
    for (int i = 0; i < n; ++i) if (bm[i] < 1e-7) bm[i] = 0.5 * (bm[(i - 1 + n) % n] + bm[(i + 1) % n]);
    return bm;
}

double disp_switch_integral(double sig, double rs, double rc) {
    // int_rs^rc (1 - S(r)) ((sig/r)^12 - (sig/r)^6) r^2 dr   with composite Simpson, 4096 intervals (error ~ h^4 f'''' << 1e-13)
    const int n = 4096; const double h = (rc - rs) / n;
    auto f = [&](double r) {
        const double x = (r - rs) / (rc - rs);
        const double S = 1.0 - 10.0 * x * x * x + 15.0 * x * x * x * x - 6.0 * x * x * x * x * x;
        const double s6 = pow(sig / r, 6);
        return (1.0 - S) * (s6 * s6 - s6) * r * r;
    };
    double acc = f(rs) + f(rc);
    for (int i = 1; i < n; ++i) acc += f(rs + i * h) * ((i & 1) ? 4.0 : 2.0);
    return acc * h / 3.0;
}

double dispersion_coefficient(const System& s) {
    // E_disp = coeff / V, OpenMM NonbondedForce convention (average over the N(N+1)/2 pair multiset); alchemical atoms
    // carry no dispersion correction in the NonbondedForce (their sterics live in the custom forces, alchemy.py:1786-1789)
    std::map<std::pair<double, double>, double> classes;
    for (int i = 0; i < s.N; ++i) classes[{s.sig[i], s.alch[i] ? 0.0 : s.eps[i]}] += 1.0;
    std::vector<std::pair<std::pair<double, double>, double>> cl(classes.begin(), classes.end());
    double s1 = 0, s2 = 0, s3 = 0;
    for (size_t a = 0; a < cl.size(); ++a) for (size_t b = a; b < cl.size(); ++b) {
        const double na = cl[a].second, nb = cl[b].second;
        const double count = (a == b) ? na * (na + 1) / 2.0 : na * nb;
        const double sg = 0.5 * (cl[a].first.first + cl[b].first.first), ep = sqrt(cl[a].first.second * cl[b].first.second);
        if (ep == 0.0) continue;
        s1 += count * ep * pow(sg, 12); s2 += count * ep * pow(sg, 6);
        if (s.rs >= 0 && s.rs < s.rc) s3 += count * ep * disp_switch_integral(sg, s.rs, s.rc);
    }
    const double npairs = (double)s.N * (s.N + 1) / 2.0, N = s.N;
    return 8.0 * N * N * PI * (s1 / npairs / (9.0 * pow(s.rc, 9)) - s2 / npairs / (3.0 * pow(s.rc, 3)) + s3 / npairs);
}

// ------------------------------------------------------------------------------------------------------------------
// one replica: state, neighbour list, PME work space
// ------------------------------------------------------------------------------------------------------------------
struct Replica {
    std::vector<double> x, v, f;                  // [N][3]
    double box[3] = {0, 0, 0};
    bool f_valid = false;
    double baro[5] = {0, 0, 0, 0, 0};             // volume scale, attempted, accepted (adaptation window), total attempted, total accepted
    double heat = 0, shadow = 0; long long n_trials = 0, n_rejected = 0;   // remd_get_work (integrators.py:1175-1204, 1539-1557)
    // Verlet list
    std::vector<double> x_list; double box_list[3] = {0, 0, 0};
    std::vector<int> pair_i, pair_j;              // i < j, not excluded, within rc + skin at build time
    bool list_valid = false;
    // PME
    std::vector<double> Q, phi;
    std::vector<cplx> H;
    int pme_nthreads = 1;
};

struct Timers { double list = 0, pairs = 0, bonded = 0, exc = 0, pme = 0; };
static Timers g_timers;
static const bool g_time = getenv("REMD_CPU_TIMERS") != nullptr;
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Energy { double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; double total() const { double t = 0; for (double v : c) t += v; return t; } };

inline double min_image(double d, double L) { return d - L * nearbyint(d / L); }

void build_list(const System& s, Replica& r)
{
    const int N = s.N;
    const double rl = std::max(s.rc, s.rcc) + SKIN, rl2 = rl * rl;
    r.pair_i.clear(); r.pair_j.clear();
    r.x_list = r.x; for (int k = 0; k < 3; ++k) r.box_list[k] = r.box[k];
    int nc[3]; double cs[3];
    for (int k = 0; k < 3; ++k) { nc[k] = std::max(1, (int)floor(r.box[k] / rl)); cs[k] = r.box[k] / nc[k]; }
    const int ncell = nc[0] * nc[1] * nc[2];
    std::vector<int> head(ncell + 1, 0), cell(N), order(N);
    for (int i = 0; i < N; ++i) {
        int c[3];
        for (int k = 0; k < 3; ++k) {
            double u = r.x[3 * i + k] / r.box[k]; u -= floor(u);
            c[k] = std::min(nc[k] - 1, (int)(u * nc[k]));
        }
        cell[i] = (c[0] * nc[1] + c[1]) * nc[2] + c[2];
        head[cell[i] + 1]++;
    }
    for (int c = 0; c < ncell; ++c) head[c + 1] += head[c];
    { std::vector<int> pos(head.begin(), head.end() - 1); for (int i = 0; i < N; ++i) order[pos[cell[i]]++] = i; }
    std::vector<int> nbr;
    for (int cx = 0; cx < nc[0]; ++cx) for (int cy = 0; cy < nc[1]; ++cy) for (int cz = 0; cz < nc[2]; ++cz) {
        const int c0 = (cx * nc[1] + cy) * nc[2] + cz;
        // each unordered pair of neighbouring cells once; with fewer than 3 cells along a dimension the -1 / +1 images
        // coincide, so the neighbour set is made unique first
        nbr.clear();
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
            const int c1 = (((cx + dx + nc[0]) % nc[0]) * nc[1] + (cy + dy + nc[1]) % nc[1]) * nc[2] + (cz + dz + nc[2]) % nc[2];
            if (c1 >= c0) nbr.push_back(c1);
        }
        std::sort(nbr.begin(), nbr.end());
        nbr.erase(std::unique(nbr.begin(), nbr.end()), nbr.end());
        for (int c1 : nbr) {
            for (int a = head[c0]; a < head[c0 + 1]; ++a) {
                const int i = order[a];
                for (int b = (c1 == c0 ? a + 1 : head[c1]); b < head[c1 + 1]; ++b) {
                    const int j = order[b];
                    double d2 = 0; for (int k = 0; k < 3; ++k) { const double d = min_image(r.x[3 * j + k] - r.x[3 * i + k], r.box[k]); d2 += d * d; }
                    if (d2 >= rl2) continue;
                    const int lo = std::min(i, j), hi = std::max(i, j);
                    if (std::binary_search(s.excl[lo].begin(), s.excl[lo].end(), hi)) continue;
                    r.pair_i.push_back(lo); r.pair_j.push_back(hi);
                }
            }
        }
    }
    r.list_valid = true;
}

void ensure_list(const System& s, Replica& r)
{
    bool ok = r.list_valid && (int)r.x_list.size() == 3 * s.N;
    if (ok) for (int k = 0; k < 3; ++k) if (r.box_list[k] != r.box[k]) ok = false;
    if (ok) {
        const double lim = 0.25 * SKIN * SKIN;
        for (int i = 0; i < s.N && ok; ++i) {
            double d2 = 0; for (int k = 0; k < 3; ++k) { const double d = r.x[3 * i + k] - r.x_list[3 * i + k]; d2 += d * d; }
            if (!(d2 < lim)) ok = false;             // also catches NaN
        }
    }
    if (!ok) build_list(s, r);
}

inline void bspline(double f, double w[PME_ORDER], double dw[PME_ORDER])
{
    // w[j] = M_5(f + j), dw[j] = M_4(f + j) - M_4(f + j - 1); recursion of oracle/forcefield.py:_bspline_weights
    double a[PME_ORDER] = {f, 1.0 - f, 0, 0, 0}, b[PME_ORDER];
    for (int m = 3; m <= PME_ORDER; ++m) {
        if (m == PME_ORDER) for (int j = 0; j < PME_ORDER; ++j) dw[j] = a[j] - (j > 0 ? a[j - 1] : 0.0);
        for (int j = 0; j < PME_ORDER; ++j) {
            if (j < m) {
                const double cur = (j < m - 1) ? a[j] : 0.0, prev = (j > 0) ? a[j - 1] : 0.0;
                b[j] = ((f + j) * cur + (m - f - j) * prev) / (m - 1);
            } else b[j] = 0.0;
        }
        for (int j = 0; j < PME_ORDER; ++j) a[j] = b[j];
    }
    for (int j = 0; j < PME_ORDER; ++j) w[j] = a[j];
}

struct FFTSet { std::unique_ptr<FFT1D> f[3]; };

// smooth PME reciprocal energy (and forces when f != nullptr) for charges qv
double pme_reciprocal(const System& s, Replica& r, const std::vector<double>& qv, double* f, FFTSet& fft)
{
    const int nx = s.grid[0], ny = s.grid[1], nz = s.grid[2], nzh = nz / 2 + 1;
    const size_t G = (size_t)nx * ny * nz, GH = (size_t)nx * ny * nzh;
    r.Q.assign(G, 0.0); r.H.resize(GH);
    const int N = s.N;
    const int n[3] = {nx, ny, nz};
    // spread
    for (int i = 0; i < N; ++i) {
        if (qv[i] == 0.0) continue;
        double w[3][PME_ORDER], dw[3][PME_ORDER]; int k0[3];
        for (int k = 0; k < 3; ++k) {
            double u = r.x[3 * i + k] / r.box[k]; u = (u - floor(u)) * n[k];
            k0[k] = (int)floor(u);
            bspline(u - k0[k], w[k], dw[k]);
        }
        for (int a = 0; a < PME_ORDER; ++a) {
            const int ix = ((k0[0] - a) % nx + nx) % nx;
            for (int b = 0; b < PME_ORDER; ++b) {
                const int iy = ((k0[1] - b) % ny + ny) % ny;
                const double wab = qv[i] * w[0][a] * w[1][b];
                double* row = &r.Q[((size_t)ix * ny + iy) * nz];
                for (int c = 0; c < PME_ORDER; ++c) row[((k0[2] - c) % nz + nz) % nz] += wab * w[2][c];
            }
        }
    }
    // forward transform: z (real input, keep the half spectrum), then y, then x
    std::vector<cplx> lin(std::max(std::max(nx, ny), nz)), lout(lin.size());
    for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy) {
        const double* row = &r.Q[((size_t)ix * ny + iy) * nz];
        for (int k = 0; k < nz; ++k) lin[k] = cplx(row[k], 0.0);
        fft.f[2]->transform(lin.data(), lout.data(), false);
        cplx* h = &r.H[((size_t)ix * ny + iy) * nzh];
        for (int k = 0; k < nzh; ++k) h[k] = lout[k];
    }
    for (int ix = 0; ix < nx; ++ix) for (int kz = 0; kz < nzh; ++kz) {
        for (int iy = 0; iy < ny; ++iy) lin[iy] = r.H[((size_t)ix * ny + iy) * nzh + kz];
        fft.f[1]->transform(lin.data(), lout.data(), false);
        for (int iy = 0; iy < ny; ++iy) r.H[((size_t)ix * ny + iy) * nzh + kz] = lout[iy];
    }
    const double V = r.box[0] * r.box[1] * r.box[2];
    double energy = 0.0;
    for (int iy = 0; iy < ny; ++iy) for (int kz = 0; kz < nzh; ++kz) {
        for (int ix = 0; ix < nx; ++ix) lin[ix] = r.H[((size_t)ix * ny + iy) * nzh + kz];
        fft.f[0]->transform(lin.data(), lout.data(), false);
        // influence function G(m) = k_e exp(-pi^2 m^2 / alpha^2) / (pi V m^2 b(m)), applied in place (x is the last pass)
        const double my = (iy <= ny / 2 ? iy : iy - ny) / r.box[1], mz = kz / r.box[2];
        const double wz = (kz == 0 || (nz % 2 == 0 && kz == nz / 2)) ? 1.0 : 2.0;
        for (int ix = 0; ix < nx; ++ix) {
            const double mx = (ix <= nx / 2 ? ix : ix - nx) / r.box[0];
            const double m2 = mx * mx + my * my + mz * mz;
            double g = 0.0;
            if (m2 > 0) g = ONE_4PI_EPS0 * exp(-PI * PI * m2 / (s.alpha * s.alpha)) / (s.bmod[0][ix] * s.bmod[1][iy] * s.bmod[2][kz] * PI * V * m2);
            energy += 0.5 * wz * g * std::norm(lout[ix]);
            lout[ix] *= g;
        }
        if (f) {
            fft.f[0]->transform(lout.data(), lin.data(), true);
            for (int ix = 0; ix < nx; ++ix) r.H[((size_t)ix * ny + iy) * nzh + kz] = lin[ix];
        }
    }
    if (!f) return energy;
    // inverse y, then z with Hermitian completion -> real potential mesh phi = sum_m G S e^{+2 pi i m r / n}
    for (int ix = 0; ix < nx; ++ix) for (int kz = 0; kz < nzh; ++kz) {
        for (int iy = 0; iy < ny; ++iy) lin[iy] = r.H[((size_t)ix * ny + iy) * nzh + kz];
        fft.f[1]->transform(lin.data(), lout.data(), true);
        for (int iy = 0; iy < ny; ++iy) r.H[((size_t)ix * ny + iy) * nzh + kz] = lout[iy];
    }
    r.phi.resize(G);
    for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy) {
        const cplx* h = &r.H[((size_t)ix * ny + iy) * nzh];
        for (int k = 0; k < nzh; ++k) lin[k] = h[k];
        for (int k = nzh; k < nz; ++k) lin[k] = std::conj(h[nz - k]);
        fft.f[2]->transform(lin.data(), lout.data(), true);
        double* row = &r.phi[((size_t)ix * ny + iy) * nz];
        for (int k = 0; k < nz; ++k) row[k] = lout[k].real();
    }
    // gather: F_i = -q_i sum dtheta/dx phi
    for (int i = 0; i < N; ++i) {
        if (qv[i] == 0.0) continue;
        double w[3][PME_ORDER], dw[3][PME_ORDER]; int k0[3];
        for (int k = 0; k < 3; ++k) {
            double u = r.x[3 * i + k] / r.box[k]; u = (u - floor(u)) * n[k];
            k0[k] = (int)floor(u);
            bspline(u - k0[k], w[k], dw[k]);
        }
        double g[3] = {0, 0, 0};
        for (int a = 0; a < PME_ORDER; ++a) {
            const int ix = ((k0[0] - a) % nx + nx) % nx;
            for (int b = 0; b < PME_ORDER; ++b) {
                const int iy = ((k0[1] - b) % ny + ny) % ny;
                const double* row = &r.phi[((size_t)ix * ny + iy) * nz];
                double s0 = 0, s1 = 0;
                for (int c = 0; c < PME_ORDER; ++c) { const double p = row[((k0[2] - c) % nz + nz) % nz]; s0 += w[2][c] * p; s1 += dw[2][c] * p; }
                g[0] += dw[0][a] * w[1][b] * s0; g[1] += w[0][a] * dw[1][b] * s0; g[2] += w[0][a] * w[1][b] * s1;
            }
        }
        for (int k = 0; k < 3; ++k) f[3 * i + k] -= qv[i] * g[k] * n[k] / r.box[k];
    }
    return energy;
}

// ------------------------------------------------------------------------------------------------------------------
// custom forces of general alchemical regions at the lambdas of state `state` (include/remd_hip.h: remd_set_alchemical_regions;
// alchemy.py:1356-1537 expressions, :1539-2038 force split): all (alchemical atom, atom) pairs inside the cutoff + the exceptions
// ------------------------------------------------------------------------------------------------------------------
static inline void region_switch(double rs, double rc, double rr, double& e, double& dedr)
{
    if (rs >= 0 && rs < rc && rr > rs) {
        const double t = (rr - rs) / (rc - rs);
        const double S = 1.0 - 10.0 * t * t * t + 15.0 * t * t * t * t - 6.0 * t * t * t * t * t;
        const double dS = (-30.0 * t * t + 60.0 * t * t * t - 30.0 * t * t * t * t) / (rc - rs);
        dedr = dedr * S + e * dS; e *= S;
    }
}
static inline void region_class_lambdas(const System::Regions& g, const System::RegionClass& c, int state, double& l_s, double& l_e)
{
    const double* ls = &g.ls[(size_t)state * g.n]; const double* le = &g.le[(size_t)state * g.n];
    if (c.kind == 0) { l_s = ls[c.a - 1]; l_e = le[c.a - 1]; }
    else if (c.kind == 1) { l_s = g.annihilate[2 * (c.a - 1)] ? ls[c.a - 1] : 1.0; l_e = g.annihilate[2 * (c.a - 1) + 1] ? le[c.a - 1] : 1.0; }
    else { l_s = ls[c.a - 1] * ls[c.b - 1]; l_e = le[c.a - 1] * le[c.b - 1]; }
}
static inline void region_sterics(const double* sc, double l, double sg, double ep, double rr, double& e, double& dedr)
{
    const double a = sc[2], b = sc[3], c = sc[4];
    const double t = pow(rr / sg, c), base = sc[0] * pow(1.0 - l, b) + t, x = pow(base, -6.0 / c), la = pow(l, a);
    e = la * 4.0 * ep * x * (x - 1.0);
    dedr = la * 4.0 * ep * (2.0 * x - 1.0) * (-6.0 * x / base * t / rr);
}
static inline void region_elec(const double* sc, double l, double alpha, double krf, double crf, double sg, double qq, double rr, double& e, double& dedr)
{
    const double d = sc[5], ee = sc[6], f = sc[7];
    const double t = pow(rr / sg, f), base = sc[1] * pow(1.0 - l, ee) + t, reff = sg * pow(base, 1.0 / f), ld = pow(l, d);
    double g, dg;
    if (alpha > 0) { const double ar = alpha * reff, ec = erfc(ar); g = ec / reff; dg = -(ec / reff + 2.0 * alpha / sqrt(PI) * exp(-ar * ar)) / reff; }
    else { g = 1.0 / reff; dg = -1.0 / (reff * reff); }
    g += krf * reff * reff - crf; dg += 2.0 * krf * reff;
    e = ld * ONE_4PI_EPS0 * qq * g;
    dedr = ld * ONE_4PI_EPS0 * qq * dg * (reff / base * t / rr);
}
double region_energy(const System& s, const Replica& r, int state, double* f)
{
    const System::Regions& g = s.reg;
    if (g.n == 0) return 0.0;
    const double* x = r.x.data();
    const double rc2 = s.rc * s.rc;
    double E = 0.0;
    for (size_t ia = 0; ia < g.alch.size(); ++ia) {
        const int a = g.alch[ia];
        const std::vector<char>& skip = g.skip[ia];
        for (int j = 0; j < s.N; ++j) {
            if (skip[j]) continue;
            double d[3];
            for (int k = 0; k < 3; ++k) { d[k] = x[3 * j + k] - x[3 * a + k]; if (s.method != 0) d[k] = min_image(d[k], r.box[k]); }
            const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            if (r2 >= rc2) continue;
            const System::RegionClass& c = g.classes[g.cls_of[(size_t)g.region_of[a] * (g.n + 1) + g.region_of[j]]];
            const double* sc = &g.softcore[8 * (size_t)(c.P - 1)];
            double l_s, l_e; region_class_lambdas(g, c, state, l_s, l_e);
            const double rr = sqrt(r2), sg = 0.5 * (g.sig[a] + g.sig[j]), ep = sqrt(g.eps[a] * g.eps[j]), qq = g.q[a] * g.q[j];
            double dedr = 0.0;
            if (ep != 0.0) { double e, de; region_sterics(sc, l_s, sg, ep, rr, e, de); region_switch(s.rs, s.rc, rr, e, de); E += e; dedr += de; }
            if (g.elec && qq != 0.0) { double e, de; region_elec(sc, l_e, g.alpha, g.krf, g.crf, sg, qq, rr, e, de); region_switch(g.rs_e, s.rc, rr, e, de); E += e; dedr += de; }
            if (f && dedr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * a + k] += dedr / rr * d[k]; f[3 * j + k] -= dedr / rr * d[k]; }
        }
    }
    for (size_t e2 = 0; e2 < g.exc_cls.size(); ++e2) {
        const int i = g.exc_atoms[2 * e2], j = g.exc_atoms[2 * e2 + 1];
        double d[3];
        for (int k = 0; k < 3; ++k) { d[k] = x[3 * j + k] - x[3 * i + k]; if (s.method != 0) d[k] = min_image(d[k], r.box[k]); }
        const double rr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const System::RegionClass& c = g.classes[g.exc_cls[e2]];
        const double* sc = &g.softcore[8 * (size_t)(c.P - 1)];
        double l_s, l_e; region_class_lambdas(g, c, state, l_s, l_e);
        const double qq = g.exc_params[3 * e2], sg = g.exc_params[3 * e2 + 1], ep = g.exc_params[3 * e2 + 2];
        double dedr = 0.0;
        if (ep != 0.0) { double e, de; region_sterics(sc, l_s, sg, ep, rr, e, de); E += e; dedr += de; }
        if (g.elec && qq != 0.0) {
            double e, de;
            if (g.consistent_exc) region_elec(sc, l_e, g.alpha, g.krf, g.crf, sg, qq, rr, e, de); else region_elec(sc, l_e, 0.0, 0.0, 0.0, sg, qq, rr, e, de);
            E += e; dedr += de;
        }
        if (f && dedr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * i + k] += dedr / rr * d[k]; f[3 * j + k] -= dedr / rr * d[k]; }
    }
    // softened bonded terms: lambda x harmonic bond / harmonic angle / periodic torsion (alchemy.py:1180, 1261, 1341); central differences
    // of the energy would do for a checker, but the analytic forces are short
    const double* bl = g.bl.empty() ? nullptr : &g.bl[(size_t)state * 3 * g.n];
    auto lam_of = [&](int kind, int region) { return bl ? bl[(size_t)kind * g.n + region - 1] : 1.0; };
    for (size_t b = 0; b < g.bond_region.size(); ++b) {
        const int i = g.bond_atoms[2 * b], j = g.bond_atoms[2 * b + 1];
        const double r0 = g.bond_params[2 * b], k = g.bond_params[2 * b + 1], lam = lam_of(0, g.bond_region[b]);
        double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
        const double rr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        E += lam * 0.5 * k * (rr - r0) * (rr - r0);
        if (f) { const double gg = lam * k * (rr - r0) / rr; for (int c = 0; c < 3; ++c) { f[3 * i + c] += gg * d[c]; f[3 * j + c] -= gg * d[c]; } }
    }
    for (size_t a = 0; a < g.angle_region.size(); ++a) {
        const int i = g.angle_atoms[3 * a], j = g.angle_atoms[3 * a + 1], k = g.angle_atoms[3 * a + 2];
        const double th0 = g.angle_params[2 * a], ka = g.angle_params[2 * a + 1], lam = lam_of(1, g.angle_region[a]);
        double v0[3], v1[3];
        for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
        const double n0 = sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]), n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
        double cs = (v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2]) / (n0 * n1);
        cs = std::max(-1.0, std::min(1.0, cs));
        const double th = acos(cs);
        E += lam * 0.5 * ka * (th - th0) * (th - th0);
        if (f) {
            const double sn = sqrt(std::max(1e-30, 1.0 - cs * cs)), dEdth = lam * ka * (th - th0);
            for (int c = 0; c < 3; ++c) {
                const double gi = -(v1[c] / n1 - cs * v0[c] / n0) / (n0 * sn), gk = -(v0[c] / n0 - cs * v1[c] / n1) / (n1 * sn);
                f[3 * i + c] -= dEdth * gi; f[3 * k + c] -= dEdth * gk; f[3 * j + c] += dEdth * (gi + gk);
            }
        }
    }
    for (size_t t = 0; t < g.torsion_region.size(); ++t) {
        const int a0 = g.torsion_atoms[4 * t], a1 = g.torsion_atoms[4 * t + 1], a2 = g.torsion_atoms[4 * t + 2], a3 = g.torsion_atoms[4 * t + 3];
        const double per = g.torsion_params[3 * t], phase = g.torsion_params[3 * t + 1], kt = g.torsion_params[3 * t + 2], lam = lam_of(2, g.torsion_region[t]);
        double b1[3], b2[3], b3[3], m[3], nn[3];
        for (int c = 0; c < 3; ++c) { b1[c] = x[3 * a1 + c] - x[3 * a0 + c]; b2[c] = x[3 * a2 + c] - x[3 * a1 + c]; b3[c] = x[3 * a3 + c] - x[3 * a2 + c]; }
        m[0] = b1[1] * b2[2] - b1[2] * b2[1]; m[1] = b1[2] * b2[0] - b1[0] * b2[2]; m[2] = b1[0] * b2[1] - b1[1] * b2[0];
        nn[0] = b2[1] * b3[2] - b2[2] * b3[1]; nn[1] = b2[2] * b3[0] - b2[0] * b3[2]; nn[2] = b2[0] * b3[1] - b2[1] * b3[0];
        const double b2n = sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
        const double phi = atan2(b2n * (b1[0] * nn[0] + b1[1] * nn[1] + b1[2] * nn[2]), m[0] * nn[0] + m[1] * nn[1] + m[2] * nn[2]);
        E += lam * kt * (1.0 + cos(per * phi - phase));
        if (f) {
            const double dEdphi = -lam * kt * per * sin(per * phi - phase);
            const double m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2], n2 = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
            const double b1b2 = b1[0] * b2[0] + b1[1] * b2[1] + b1[2] * b2[2], b3b2 = b3[0] * b2[0] + b3[1] * b2[1] + b3[2] * b2[2];
            for (int c = 0; c < 3; ++c) {
                const double g0 = -b2n / m2 * m[c], g3 = b2n / n2 * nn[c];
                const double g1 = (-1.0 - b1b2 / (b2n * b2n)) * g0 + (b3b2 / (b2n * b2n)) * g3, g2 = -(g0 + g1 + g3);
                f[3 * a0 + c] -= dEdphi * g0; f[3 * a1 + c] -= dEdphi * g1; f[3 * a2 + c] -= dEdphi * g2; f[3 * a3 + c] -= dEdphi * g3;
            }
        }
    }
    return E;
}

// ------------------------------------------------------------------------------------------------------------------
// GBSA: OBC2 Born radii + ACE surface term, the expressions of alchemy.py:2144-2225; lam = lambda_electrostatics of the alchemical particles
// ------------------------------------------------------------------------------------------------------------------
static inline void gb_H(double r, double or1, double sr2, double& H, double& dH)
{
    H = dH = 0.0;
    if (r + sr2 - or1 < 0) return;
    const double U = r + sr2, D = fabs(r - sr2);
    const bool moving = D > or1;
    const double L = moving ? D : or1, dL = moving ? (r > sr2 ? 1.0 : -1.0) : 0.0;
    const bool inside = sr2 - r - or1 >= 0;
    const double C = inside ? 2.0 * (1.0 / or1 - 1.0 / L) : 0.0, dC = inside ? 2.0 * dL / (L * L) : 0.0;
    const double a = 1.0 / (U * U) - 1.0 / (L * L), w = r - sr2 * sr2 / r, lg = log(L / U);
    H = 0.5 * (1.0 / L - 1.0 / U + 0.25 * w * a + 0.5 * lg / r + C);
    dH = 0.5 * (-dL / (L * L) + 1.0 / (U * U) + 0.25 * (1.0 + sr2 * sr2 / (r * r)) * a + 0.25 * w * (-2.0 / (U * U * U) + 2.0 * dL / (L * L * L))
                + 0.5 * ((dL / L - 1.0 / U) / r - lg / (r * r)) + dC);
}
double gb_energy(const System& s, const double* x, double lam, double* f)
{
    const System::GB& g = s.gb;
    const int N = g.n;
    const double KE = 138.935485, OFF = 0.009, SA = 28.3919551;
    std::vector<double> sf(N), orr(N), sr(N), B(N), dBdI(N), dEdB(N, 0.0);
    for (int i = 0; i < N; ++i) { sf[i] = g.alch[i] ? lam : 1.0; orr[i] = g.R[i] - OFF; sr[i] = g.sc[i] * orr[i]; }
    auto dist = [&](int i, int j, double d[3]) { for (int k = 0; k < 3; ++k) d[k] = x[3 * j + k] - x[3 * i + k]; return sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); };
    for (int i = 0; i < N; ++i) {
        double I = 0;
        for (int j = 0; j < N; ++j) if (j != i) { double d[3], H, dH; gb_H(dist(i, j, d), orr[i], sr[j], H, dH); I += sf[j] * H; }
        const double psi = I * orr[i], P = psi - 0.8 * psi * psi + 4.85 * psi * psi * psi, th = tanh(P);
        B[i] = 1.0 / (1.0 / orr[i] - th / g.R[i]);
        dBdI[i] = B[i] * B[i] * (1.0 - th * th) * (1.0 - 1.6 * psi + 14.55 * psi * psi) * orr[i] / g.R[i];
    }
    double E = 0;
    for (int i = 0; i < N; ++i) {
        E += -0.5 * KE * g.tau * sf[i] * g.q[i] * g.q[i] / B[i];
        dEdB[i] += 0.5 * KE * g.tau * sf[i] * g.q[i] * g.q[i] / (B[i] * B[i]);
        if (g.sasa) {
            const double rb = g.R[i] / B[i], rb6 = rb * rb * rb * rb * rb * rb, pre = sf[i] * SA * (g.R[i] + 0.14) * (g.R[i] + 0.14);
            E += pre * rb6; dEdB[i] += -6.0 * pre * rb6 / B[i];
        }
    }
    for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j) {
        double d[3]; const double r = dist(i, j, d);
        const double D = B[i] * B[j], ex = exp(-r * r / (4.0 * D)), f2 = r * r + D * ex, ff = sqrt(f2);
        const double QQ = KE * g.tau * sf[i] * g.q[i] * sf[j] * g.q[j];
        E += -QQ / ff;
        // d(-QQ/f) = QQ / f^2 df;  df/dr = (r - r ex / 4) / f;  df/dD = ex (1 + r^2 / (4 D)) / (2 f)
        const double dEdf = QQ / f2;
        const double dfdD = ex * (1.0 + r * r / (4.0 * D)) / (2.0 * ff);
        dEdB[i] += dEdf * dfdD * B[j]; dEdB[j] += dEdf * dfdD * B[i];
        if (f) { const double gr = dEdf * (r - 0.25 * r * ex) / ff / r; for (int k = 0; k < 3; ++k) { f[3 * i + k] += gr * d[k]; f[3 * j + k] -= gr * d[k]; } }
    }
    if (f) {
        // the chain through the Born radii: dE/dB_i dB_i/dI_i s_j H'(r; or_i, sr_j) on the pair (i, j), from both sides
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) if (j != i) {
            double d[3], H, dH; const double r = dist(i, j, d);
            gb_H(r, orr[i], sr[j], H, dH);
            const double gr = dEdB[i] * dBdI[i] * sf[j] * dH / r;          // dE/dr of this contribution, over r
            for (int k = 0; k < 3; ++k) { f[3 * i + k] += gr * d[k]; f[3 * j + k] -= gr * d[k]; }
        }
    }
    return E;
}

// ------------------------------------------------------------------------------------------------------------------
// potential energy and forces of one replica at (lambda_sterics, lambda_electrostatics)
//   parts: which contributions to evaluate (the u_kl assembly re-evaluates only what a lambda changes)
// ------------------------------------------------------------------------------------------------------------------
enum { PART_BONDED = 1, PART_STERICS = 2, PART_SOFTCORE = 4, PART_ELEC = 8, PART_ALL = 15 };

// classes: bit c set = force class c of remd_set_force_groups is evaluated (0 external, 1 bonds, 2 angles, 3 torsions, 4 nonbonded direct space +
// exceptions + exclusion correction + dispersion constant, 5 PME reciprocal space + self terms): the forces of one force group of a
// multiple-time-step splitting (integrators.py:1425-1442; same convention as oracle/forcefield.py energy_torch)
Energy evaluate(const System& s, Replica& r, double lam_s, double lam_e, double* f, FFTSet& fft, int parts = PART_ALL, int classes = 63, int region_state = -1)
{
    Energy E;
    const int N = s.N;
    const double* x = r.x.data();
    if (f) std::fill(f, f + 3 * N, 0.0);
    const bool periodic = s.method != 0;
    auto delta = [&](int i, int j, double d[3]) {
        for (int k = 0; k < 3; ++k) { d[k] = x[3 * j + k] - x[3 * i + k]; if (periodic) d[k] = min_image(d[k], r.box[k]); }
    };
    if (parts & PART_BONDED) {
        if (!s.ext_atoms.empty() && (classes & 1)) {                  // testsystems.py:779-786
            for (int i : s.ext_atoms) {
                const double dx[3] = {x[3 * i] - s.ext_x0, x[3 * i + 1], x[3 * i + 2]};
                E.c[0] += 0.5 * s.ext_K * (dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]) + s.ext_U0;
                if (f) for (int k = 0; k < 3; ++k) f[3 * i + k] -= s.ext_K * dx[k];
            }
        }
        for (size_t b = 0; (classes & 2) && b < s.bond_atoms.size() / 2; ++b) {
            const int i = s.bond_atoms[2 * b], j = s.bond_atoms[2 * b + 1];
            const double r0 = s.bond_params[2 * b], k = s.bond_params[2 * b + 1];
            double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
            const double rr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            E.c[1] += 0.5 * k * (rr - r0) * (rr - r0);
            if (f) { const double g = k * (rr - r0) / rr; for (int c = 0; c < 3; ++c) { f[3 * i + c] += g * d[c]; f[3 * j + c] -= g * d[c]; } }
        }
        for (size_t a = 0; (classes & 4) && a < s.angle_atoms.size() / 3; ++a) {
            const int i = s.angle_atoms[3 * a], j = s.angle_atoms[3 * a + 1], k = s.angle_atoms[3 * a + 2];
            const double th0 = s.angle_params[2 * a], ka = s.angle_params[2 * a + 1];
            double v0[3], v1[3];
            for (int c = 0; c < 3; ++c) { v0[c] = x[3 * i + c] - x[3 * j + c]; v1[c] = x[3 * k + c] - x[3 * j + c]; }
            const double n0 = sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]), n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
            double cs = (v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2]) / (n0 * n1);
            cs = std::max(-1.0, std::min(1.0, cs));
            const double th = acos(cs);
            E.c[2] += 0.5 * ka * (th - th0) * (th - th0);
            if (f) {
                const double sn = sqrt(std::max(1e-30, 1.0 - cs * cs));
                const double dEdth = ka * (th - th0);
                // d theta / d r_i = -(v1/n1 - cs v0/n0) / (n0 sin)
                for (int c = 0; c < 3; ++c) {
                    const double gi = -(v1[c] / n1 - cs * v0[c] / n0) / (n0 * sn), gk = -(v0[c] / n0 - cs * v1[c] / n1) / (n1 * sn);
                    f[3 * i + c] -= dEdth * gi; f[3 * k + c] -= dEdth * gk; f[3 * j + c] += dEdth * (gi + gk);
                }
            }
        }
        for (size_t t = 0; (classes & 8) && t < s.torsion_atoms.size() / 4; ++t) {
            const int a0 = s.torsion_atoms[4 * t], a1 = s.torsion_atoms[4 * t + 1], a2 = s.torsion_atoms[4 * t + 2], a3 = s.torsion_atoms[4 * t + 3];
            const double per = s.torsion_params[3 * t], phase = s.torsion_params[3 * t + 1], kt = s.torsion_params[3 * t + 2];
            double b1[3], b2[3], b3[3], m[3], n[3];
            for (int c = 0; c < 3; ++c) { b1[c] = x[3 * a1 + c] - x[3 * a0 + c]; b2[c] = x[3 * a2 + c] - x[3 * a1 + c]; b3[c] = x[3 * a3 + c] - x[3 * a2 + c]; }
            m[0] = b1[1] * b2[2] - b1[2] * b2[1]; m[1] = b1[2] * b2[0] - b1[0] * b2[2]; m[2] = b1[0] * b2[1] - b1[1] * b2[0];
            n[0] = b2[1] * b3[2] - b2[2] * b3[1]; n[1] = b2[2] * b3[0] - b2[0] * b3[2]; n[2] = b2[0] * b3[1] - b2[1] * b3[0];
            const double b2n = sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
            const double phi = atan2(b2n * (b1[0] * n[0] + b1[1] * n[1] + b1[2] * n[2]), m[0] * n[0] + m[1] * n[1] + m[2] * n[2]);
            E.c[3] += kt * (1.0 + cos(per * phi - phase));
            if (f) {
                const double dEdphi = -kt * per * sin(per * phi - phase);
                const double m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2], n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                const double b1b2 = b1[0] * b2[0] + b1[1] * b2[1] + b1[2] * b2[2], b3b2 = b3[0] * b2[0] + b3[1] * b2[1] + b3[2] * b2[2];
                // Blondel & Karplus: dphi/dr0 = -|b2| m / m^2 ; dphi/dr3 = |b2| n / n^2
                for (int c = 0; c < 3; ++c) {
                    const double g0 = -b2n / m2 * m[c], g3 = b2n / n2 * n[c];
                    const double g1 = (-1.0 - b1b2 / (b2n * b2n)) * g0 + (b3b2 / (b2n * b2n)) * g3;      // d phi / d r1
                    const double g2 = -(g0 + g1 + g3);
                    f[3 * a0 + c] -= dEdphi * g0; f[3 * a1 + c] -= dEdphi * g1; f[3 * a2 + c] -= dEdphi * g2; f[3 * a3 + c] -= dEdphi * g3;
                }
            }
        }
    }
    if (s.nocut) {
        // NoCutoff: every pair that is not an exception, plain Lennard-Jones + Coulomb; the exceptions with their own parameters
        if ((classes & 16) && (parts & (PART_STERICS | PART_ELEC))) {
            double e_nb = 0;
            for (int i = 0; i < N; ++i) {
                const std::vector<int>& ex = s.excl[i];
                for (int j = i + 1; j < N; ++j) {
                    if (std::binary_search(ex.begin(), ex.end(), j)) continue;
                    double d[3]; delta(i, j, d);
                    const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], rr = sqrt(r2);
                    double fr = 0.0;
                    const double ep = sqrt(s.eps[i] * s.eps[j]);
                    if (ep != 0.0 && (parts & PART_STERICS)) {
                        const double sg = 0.5 * (s.sig[i] + s.sig[j]), s2 = sg * sg / r2, s6 = s2 * s2 * s2;
                        e_nb += 4.0 * ep * (s6 * s6 - s6); fr += 4.0 * ep * (12.0 * s6 * s6 - 6.0 * s6) / r2;
                    }
                    const double qq = ONE_4PI_EPS0 * s.q[i] * s.q[j];
                    if (qq != 0.0 && (parts & PART_ELEC)) { e_nb += qq / rr; fr += qq / (rr * r2); }
                    if (f && fr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * j + k] += fr * d[k]; f[3 * i + k] -= fr * d[k]; }
                }
            }
            E.c[8] += e_nb;
            for (size_t e = 0; e < s.exc_atoms.size() / 2; ++e) {
                const int i = s.exc_atoms[2 * e], j = s.exc_atoms[2 * e + 1];
                const double qq = ONE_4PI_EPS0 * s.exc_params[3 * e], sg = s.exc_params[3 * e + 1], ep = s.exc_params[3 * e + 2];
                if (qq == 0.0 && ep == 0.0) continue;
                double d[3]; delta(i, j, d);
                const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], rr = sqrt(r2);
                double fr = 0.0;
                if (ep != 0.0 && (parts & PART_STERICS)) { const double s2 = sg * sg / r2, s6 = s2 * s2 * s2; E.c[4] += 4.0 * ep * s6 * (s6 - 1.0); fr += 4.0 * ep * (12.0 * s6 * s6 - 6.0 * s6) / r2; }
                if (qq != 0.0 && (parts & PART_ELEC)) { E.c[4] += qq / rr; fr += qq / (rr * r2); }
                if (f && fr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * j + k] += fr * d[k]; f[3 * i + k] -= fr * d[k]; }
            }
        }
        if (s.reg.n > 0 && region_state >= 0 && region_state < s.reg.K && (classes & 16) && (parts & PART_SOFTCORE)) E.c[8] += region_energy(s, r, region_state, f);
        if (s.gb.n > 0 && (classes & 16) && (parts & PART_ELEC)) {
            // lambda_electrostatics of the alchemical particles: region 1 of the state (the factory's GBSA knows one region, alchemy.py:2168-2171)
            const double lam_gb = (s.reg.n > 0 && region_state >= 0 && region_state < s.reg.K) ? s.reg.le[(size_t)region_state * s.reg.n] : lam_e;
            E.c[8] += gb_energy(s, x, lam_gb, f);
        }
        return E;
    }
    if (!periodic) return E;
    const double V = r.box[0] * r.box[1] * r.box[2];
    // charges at this lambda_electrostatics (exact PME treatment: alchemical charges scale, alchemy.py:1675-1680)
    std::vector<double> qv;
    const bool elec = (parts & PART_ELEC) && s.has_charge;
    const bool exact_regions = s.reg.n > 0 && s.reg.exact && region_state >= 0 && region_state < s.reg.K;
    const double* reg_le = exact_regions ? &s.reg.le[(size_t)region_state * s.reg.n] : nullptr;
    if (elec) {
        qv.resize(N);
        for (int i = 0; i < N; ++i) qv[i] = s.alch[i] ? s.q[i] * lam_e : s.q[i];
        if (exact_regions) for (int i = 0; i < N; ++i) if (s.reg.region_of[i] > 0) qv[i] = s.q[i] * reg_le[s.reg.region_of[i] - 1];
    }
    // ---- pair loop -------------------------------------------------------------------------------------------
    if ((classes & 16) && (parts & (PART_STERICS | PART_SOFTCORE | PART_ELEC))) {
        double tt0 = g_time ? now_ms() : 0;
        ensure_list(s, r);
        if (g_time) { const double t1 = now_ms(); g_timers.list += t1 - tt0; tt0 = t1; }
        const double rc2 = s.rc * s.rc, rcc2 = std::max(s.rc, s.rcc) * std::max(s.rc, s.rcc);
        const double krf = (s.rf_eps - 1.0) / (2.0 * s.rf_eps + 1.0) / (s.rc * s.rc * s.rc), crf = s.rf_unshifted ? 0.0 : 3.0 * s.rf_eps / (2.0 * s.rf_eps + 1.0) / s.rc;
        const double rs_c = (s.rf_unshifted && s.rf_switch_width > 0 && s.rf_switch_width < s.rc) ? s.rc - s.rf_switch_width : -1.0;
        const double two_a_sqrtpi = 2.0 * s.alpha / sqrt(PI);
        const bool sw = s.rs >= 0 && s.rs < s.rc;
        const double one_m_l = 1.0 - lam_s, la = pow(lam_s, s.sc_a), lb = s.sc_alpha * pow(one_m_l, s.sc_b);
        double e_lj = 0, e_el = 0;
        const size_t np = r.pair_i.size();
        for (size_t p = 0; p < np; ++p) {
            const int i = r.pair_i[p], j = r.pair_j[p];
            double d[3]; delta(i, j, d);
            const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            if (r2 >= rcc2) continue;
            const double rr = sqrt(r2);
            double fr = 0.0;                                       // -dE/dr / r  (force on j = fr * d)
            const double ep = (r2 < rc2) ? sqrt(s.eps[i] * s.eps[j]) : 0.0;      // Lennard-Jones stops at the NonbondedForce cutoff
            // (soft-core, lambda_sterics-controlled: alchemical with non-alchemical; under annihilate_sterics also alchemical with alchemical)
            const bool na = s.has_alch && ((s.alch[i] != s.alch[j]) || (s.annihilate && s.alch[i] && s.alch[j]));
            if (ep != 0.0 && ((na && (parts & PART_SOFTCORE)) || (!na && (parts & PART_STERICS)))) {
                const double sg = 0.5 * (s.sig[i] + s.sig[j]);
                double e, dedr;
                if (!na) {
                    const double s2 = sg * sg / r2, s6 = s2 * s2 * s2;
                    e = 4.0 * ep * (s6 * s6 - s6); dedr = 4.0 * ep * (-12.0 * s6 * s6 + 6.0 * s6) / rr;
                } else {
                    // alchemy.py:1383-1388: U = l^a 4 eps x (x - 1), x = (sigma / r_eff)^6, r_eff = sigma (alpha (1-l)^b + (r/sigma)^c)^(1/c)
                    const double rs_c = pow(rr / sg, s.sc_c);
                    const double base = lb + rs_c;
                    const double xs = pow(base, -6.0 / s.sc_c);
                    e = la * 4.0 * ep * xs * (xs - 1.0);
                    const double dxdr = (-6.0 / s.sc_c) * pow(base, -6.0 / s.sc_c - 1.0) * s.sc_c * rs_c / rr;
                    dedr = la * 4.0 * ep * (2.0 * xs - 1.0) * dxdr;
                }
                if (sw && rr > s.rs) {
                    const double t = (rr - s.rs) / (s.rc - s.rs);
                    const double S = 1.0 - 10.0 * t * t * t + 15.0 * t * t * t * t - 6.0 * t * t * t * t * t;
                    const double dS = (-30.0 * t * t + 60.0 * t * t * t - 30.0 * t * t * t * t) / (s.rc - s.rs);
                    dedr = dedr * S + e * dS; e *= S;
                }
                e_lj += e; fr -= dedr / rr;
            }
            if (elec) {
                const double qq = ONE_4PI_EPS0 * qv[i] * qv[j];
                if (qq != 0.0) {
                    if (s.method == REMD_NB_PME) {
                        const double ar = s.alpha * rr, ec = erfc(ar);
                        e_el += qq * ec / rr;
                        fr += qq * (ec / rr + two_a_sqrtpi * exp(-ar * ar)) / r2;
                    } else {
                        double e = qq * (1.0 / rr + krf * r2 - crf), dedr = qq * (2.0 * krf * rr - 1.0 / r2);
                        region_switch(rs_c, s.rc, rr, e, dedr);
                        e_el += e; fr -= dedr / rr;
                    }
                }
            }
            if (f && fr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * j + k] += fr * d[k]; f[3 * i + k] -= fr * d[k]; }
        }
        E.c[8] += e_lj + e_el;
        if (g_time) g_timers.pairs += now_ms() - tt0;
    }
    // ---- exceptions (no cutoff) and the Ewald correction of every excluded pair ------------------------------------------
    if ((classes & 16) && (parts & (PART_STERICS | PART_SOFTCORE | PART_ELEC))) {
        const size_t ne = s.exc_atoms.size() / 2;
        const double two_a_sqrtpi = 2.0 * s.alpha / sqrt(PI);
        const double la = pow(lam_s, s.sc_a), lb = s.sc_alpha * pow(1.0 - lam_s, s.sc_b);
        for (size_t e = 0; e < ne; ++e) {
            const int i = s.exc_atoms[2 * e], j = s.exc_atoms[2 * e + 1];
            const double qq0 = s.exc_params[3 * e], sg = s.exc_params[3 * e + 1], ep = s.exc_params[3 * e + 2];
            double d[3]; delta(i, j, d);
            const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], rr = sqrt(r2);
            double fr = 0.0;
            // the Lennard-Jones part of an exception between an alchemical and a non-alchemical atom is soft-core and
            // lambda_sterics-controlled like the pair interaction, without cutoff or switch (CustomBondForce of alchemy.py:1836-1851,
            // 1985-1998); alchemical/alchemical exceptions keep lambda = 1 (annihilate_sterics = False)
            // (soft-core, lambda_sterics-controlled: alchemical with non-alchemical; under annihilate_sterics also alchemical with alchemical)
            const bool na = s.has_alch && ((s.alch[i] != s.alch[j]) || (s.annihilate && s.alch[i] && s.alch[j]));
            if (ep != 0.0 && na && (parts & PART_SOFTCORE)) {
                const double rs_c = pow(rr / sg, s.sc_c);
                const double base = lb + rs_c;
                const double xs = pow(base, -6.0 / s.sc_c);
                E.c[4] += la * 4.0 * ep * xs * (xs - 1.0);
                const double dxdr = (-6.0 / s.sc_c) * pow(base, -6.0 / s.sc_c - 1.0) * s.sc_c * rs_c / rr;
                fr -= la * 4.0 * ep * (2.0 * xs - 1.0) * dxdr / rr;
            }
            if ((parts & PART_STERICS) && ep != 0.0 && !na) {
                const double s2 = sg * sg / r2, s6 = s2 * s2 * s2;
                E.c[4] += 4.0 * ep * s6 * (s6 - 1.0);
                fr += 4.0 * ep * (12.0 * s6 * s6 - 6.0 * s6) / r2;
            }
            if (parts & PART_ELEC) {
                if (qq0 != 0.0) {
                    // exception charge products that touch the alchemical region scale with lambda_e (alchemy.py:1964-1966)
                    double qq = ONE_4PI_EPS0 * ((s.has_alch && (s.alch[i] || s.alch[j])) ? qq0 * lam_e : qq0);
                    if (exact_regions && s.reg.exc_region[e] > 0) qq = ONE_4PI_EPS0 * qq0 * reg_le[s.reg.exc_region[e] - 1];      // alchemy.py:1978-1982
                    E.c[4] += qq / rr; fr += qq / (rr * r2);
                }
                if (elec && s.method == REMD_NB_PME) {
                    const double qq = ONE_4PI_EPS0 * qv[i] * qv[j];
                    if (qq != 0.0) {
                        const double ar = s.alpha * rr, ef = erf(ar);
                        E.c[5] -= qq * ef / rr;
                        fr -= qq * (ef / rr - two_a_sqrtpi * exp(-ar * ar)) / r2;
                    }
                }
            }
            if (f && fr != 0.0) for (int k = 0; k < 3; ++k) { f[3 * j + k] += fr * d[k]; f[3 * i + k] -= fr * d[k]; }
        }
    }
    if ((classes & 16) && (parts & PART_STERICS)) E.c[7] += s.disp_coeff / V;
    // custom forces of general alchemical regions, at the lambdas of state region_state (part of the direct-space nonbonded class)
    if (s.reg.n > 0 && region_state >= 0 && region_state < s.reg.K && (classes & 16) && (parts & PART_SOFTCORE)) E.c[8] += region_energy(s, r, region_state, f);
    if ((classes & 32) && elec && s.method == REMD_NB_PME) {
        const double tp0 = g_time ? now_ms() : 0;
        E.c[6] += pme_reciprocal(s, r, qv, f, fft);
        if (g_time) g_timers.pme += now_ms() - tp0;
        double q2 = 0, qs = 0; for (int i = 0; i < N; ++i) { q2 += qv[i] * qv[i]; qs += qv[i]; }
        E.c[7] -= ONE_4PI_EPS0 * s.alpha / sqrt(PI) * q2;
        E.c[7] -= ONE_4PI_EPS0 * PI * qs * qs / (2.0 * s.alpha * s.alpha * V);
    }
    return E;
}

// ------------------------------------------------------------------------------------------------------------------
// constraints: iterative SHAKE (positions along the old bond vectors) and RATTLE (velocities), per rigid cluster
// ------------------------------------------------------------------------------------------------------------------
// analytic SETTLE for a rigid three-site water (Miyamoto & Kollman, J. Comput. Chem. 13, 952 (1992)): the constrained
// positions that iterated SHAKE converges to, in closed form.  a = O, b = H1, c = H2.
inline void settle_water(const System& s, const int* at, const double* x_old, double* x_new)
{
    const double mO = s.settle_mO, mH = s.settle_mH, M = mO + 2.0 * mH;
    const double ra = s.settle_ra, rb = s.settle_rb, rc = s.settle_rc;
    const double *a0 = x_old + 3 * at[0], *b0 = x_old + 3 * at[1], *c0 = x_old + 3 * at[2];
    double *a1 = x_new + 3 * at[0], *b1 = x_new + 3 * at[1], *c1 = x_new + 3 * at[2];
    double xb0[3], xc0[3], com[3], xa1[3], xb1[3], xc1[3];
    for (int k = 0; k < 3; ++k) {
        xb0[k] = b0[k] - a0[k]; xc0[k] = c0[k] - a0[k];
        com[k] = (mO * a1[k] + mH * (b1[k] + c1[k])) / M;
        xa1[k] = a1[k] - com[k]; xb1[k] = b1[k] - com[k]; xc1[k] = c1[k] - com[k];
    }
    auto cross = [](const double* u, const double* v, double* w) { w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0]; };
    auto dot = [](const double* u, const double* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
    double Z[3], X[3], Y[3];
    cross(xb0, xc0, Z); cross(xa1, Z, X); cross(Z, X, Y);
    const double nz = 1.0 / sqrt(dot(Z, Z)), nx = 1.0 / sqrt(dot(X, X)), ny = 1.0 / sqrt(dot(Y, Y));
    for (int k = 0; k < 3; ++k) { Z[k] *= nz; X[k] *= nx; Y[k] *= ny; }
    const double xb0d = dot(X, xb0), yb0d = dot(Y, xb0), xc0d = dot(X, xc0), yc0d = dot(Y, xc0);
    const double za1d = dot(Z, xa1), xb1d = dot(X, xb1), yb1d = dot(Y, xb1), zb1d = dot(Z, xb1), xc1d = dot(X, xc1), yc1d = dot(Y, xc1), zc1d = dot(Z, xc1);
    const double sinphi = za1d / ra, cosphi = sqrt(1.0 - sinphi * sinphi);
    const double sinpsi = (zb1d - zc1d) / (2.0 * rc * cosphi), cospsi = sqrt(1.0 - sinpsi * sinpsi);
    const double ya2d = ra * cosphi, xb2d = -rc * cospsi;
    const double yb2d = -rb * cosphi - rc * sinpsi * sinphi, yc2d = -rb * cosphi + rc * sinpsi * sinphi;
    const double alpha = xb2d * (xb0d - xc0d) + yb0d * yb2d + yc0d * yc2d;
    const double beta = xb2d * (yc0d - yb0d) + xb0d * yb2d + xc0d * yc2d;
    const double gamma = xb0d * yb1d - xb1d * yb0d + xc0d * yc1d - xc1d * yc0d;
    const double al2be2 = alpha * alpha + beta * beta;
    const double sintheta = (alpha * gamma - beta * sqrt(al2be2 - gamma * gamma)) / al2be2, costheta = sqrt(1.0 - sintheta * sintheta);
    const double xa3d = -ya2d * sintheta, ya3d = ya2d * costheta, za3d = za1d;
    const double xb3d = xb2d * costheta - yb2d * sintheta, yb3d = xb2d * sintheta + yb2d * costheta, zb3d = zb1d;
    const double xc3d = -xb2d * costheta - yc2d * sintheta, yc3d = -xb2d * sintheta + yc2d * costheta, zc3d = zc1d;
    for (int k = 0; k < 3; ++k) {
        a1[k] = com[k] + X[k] * xa3d + Y[k] * ya3d + Z[k] * za3d;
        b1[k] = com[k] + X[k] * xb3d + Y[k] * yb3d + Z[k] * zb3d;
        c1[k] = com[k] + X[k] * xc3d + Y[k] * yc3d + Z[k] * zc3d;
    }
}

// positions: SETTLE for the waters, iterated SHAKE (along the old bond vectors) for the X-H clusters
void shake(const System& s, const double* x_old, double* x_new)
{
    for (const Cluster& c : s.clusters) {
        if (c.settle) { settle_water(s, c.atoms, x_old, x_new); continue; }
        for (int iter = 0; iter < 500; ++iter) {
            double worst = 0;
            for (int q = 0; q < c.nc; ++q) {
                const int i = c.atoms[c.ci[q]], j = c.atoms[c.cj[q]];
                double rn[3], ro[3], r2 = 0, dot = 0;
                for (int k = 0; k < 3; ++k) { rn[k] = x_new[3 * j + k] - x_new[3 * i + k]; ro[k] = x_old[3 * j + k] - x_old[3 * i + k]; r2 += rn[k] * rn[k]; dot += rn[k] * ro[k]; }
                const double diff = c.d[q] * c.d[q] - r2;
                worst = std::max(worst, fabs(diff) / (c.d[q] * c.d[q]));
                const double lam = diff / (2.0 * (s.invm[i] + s.invm[j]) * dot);
                for (int k = 0; k < 3; ++k) { x_new[3 * i + k] -= lam * s.invm[i] * ro[k]; x_new[3 * j + k] += lam * s.invm[j] * ro[k]; }
            }
            if (!(worst >= CONSTRAINT_TOL)) break;
        }
    }
}

// velocities (RATTLE): the Lagrange multipliers of a cluster's <= 3 constraints solve a small LINEAR system -- exact, no
// iteration.  Constraint q = (i, j): r_q . (v_j - v_i) = 0 after  v_i += invm_i sum_q' (+-) lam_q' r_q'.
void rattle(const System& s, const double* x, double* v)
{
    for (const Cluster& c : s.clusters) {
        const int n = c.nc;
        double r[3][3], A[3][3], b[3], lam[3];
        for (int q = 0; q < n; ++q) {
            const int i = c.atoms[c.ci[q]], j = c.atoms[c.cj[q]];
            b[q] = 0;
            for (int k = 0; k < 3; ++k) { r[q][k] = x[3 * j + k] - x[3 * i + k]; b[q] += r[q][k] * (v[3 * j + k] - v[3 * i + k]); }
        }
        for (int q = 0; q < n; ++q) for (int p = 0; p < n; ++p) {
            // effect of lam_p (v_ip += lam invm r_p, v_jp -= lam invm r_p) on r_q . (v_jq - v_iq)
            const int iq = c.ci[q], jq = c.cj[q], ip = c.ci[p], jp = c.cj[p];
            double coef = 0;
            if (jq == jp) coef -= s.invm[c.atoms[jp]];
            if (jq == ip) coef += s.invm[c.atoms[ip]];
            if (iq == jp) coef += s.invm[c.atoms[jp]];
            if (iq == ip) coef -= s.invm[c.atoms[ip]];
            A[q][p] = coef * (r[q][0] * r[p][0] + r[q][1] * r[p][1] + r[q][2] * r[p][2]);
        }
        // solve A lam = -b by Gaussian elimination with partial pivoting (n <= 3)
        double Mx[3][4];
        for (int q = 0; q < n; ++q) { for (int p = 0; p < n; ++p) Mx[q][p] = A[q][p]; Mx[q][n] = -b[q]; }
        for (int col = 0; col < n; ++col) {
            int piv = col;
            for (int q = col + 1; q < n; ++q) if (fabs(Mx[q][col]) > fabs(Mx[piv][col])) piv = q;
            if (piv != col) for (int p = 0; p <= n; ++p) std::swap(Mx[col][p], Mx[piv][p]);
            for (int q = col + 1; q < n; ++q) { const double fct = Mx[q][col] / Mx[col][col]; for (int p = col; p <= n; ++p) Mx[q][p] -= fct * Mx[col][p]; }
        }
        for (int q = n - 1; q >= 0; --q) { double acc = Mx[q][n]; for (int p = q + 1; p < n; ++p) acc -= Mx[q][p] * lam[p]; lam[q] = acc / Mx[q][q]; }
        for (int q = 0; q < n; ++q) {
            const int i = c.atoms[c.ci[q]], j = c.atoms[c.cj[q]];
            for (int k = 0; k < 3; ++k) { v[3 * i + k] += lam[q] * s.invm[i] * r[q][k]; v[3 * j + k] -= lam[q] * s.invm[j] * r[q][k]; }
        }
    }
}

} // namespace

// ----------------------------------------------------------------------------------------------------------------------
// the handle
// ----------------------------------------------------------------------------------------------------------------------
struct remd_ctx {
    std::string err;
    System sys;
    bool has_system = false, has_integrator = false;
    int K = 0;
    std::vector<double> beta, lam_s, lam_e, econst;
    double econst_vref = 0.0;
    std::vector<double> pressure; int baro_frequency = 0; long long baro_steps = 0, baro_attempts = 0;
    std::vector<char> tokens; int nV = 0, nR = 0, nO = 0;
    int nVg[4] = {0, 0, 0, 0};          // multiple-time-step splittings: V tokens per force group ('0' ... '3' in tokens)
    int force_groups[6] = {0, 0, 0, 0, 0, 0};   // remd_set_force_groups: group of (external, bonds, angles, torsions, direct, reciprocal)
    double dt = 0, gamma = 0; int n_steps = 0, reassign = 0, n_restart_attempts = 0;
    double coulomb_cutoff = 0;
    int annihilate_sterics = 0;
    int rf_unshifted = 0; double rf_switch_width = 0;
    std::vector<uint32_t> noise_ids;   // remd_set_replica_ids: keys of the local replicas' random streams (empty: r_begin + r)
    int measure_heat = 0, measure_shadow = 0;
    int R = 0, R_global = 0, r_begin = 0;
    std::vector<Replica> reps;
    std::vector<int64_t> labels;
    std::vector<double> ukl;                   // [R_global][K]
    std::vector<double> potential;
    uint64_t seed = 0;
    std::vector<FFTSet> fft;                   // one per OpenMP thread
    double t_prop = 0, t_energy = 0, t_mix = 0;
    int n_threads = 1;
};

static std::mutex g_err_mutex;
static std::string g_last_error;
static int fail(remd_ctx* h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    std::lock_guard<std::mutex> l(g_err_mutex); g_last_error = msg;
    return code;
}

static int parse_splitting(remd_ctx* h, const char* splitting, std::vector<char>& tokens, int& nV, int& nR, int& nO, int* nVg = nullptr)
{
    // integrators.py:1474-1537: V / R / O tokens, Metropolization braces, force-group suffixes V0, V1, ...: with more than one distinct
    // group the splitting is a multiple-time-step one -- every V must name its group (:1527-1529) and kicks with that group's forces
    // and dt / (occurrences of that group) (:1437-1438); with one group (or none) every V uses all forces and dt / (number of V)
    tokens.clear(); nV = nR = nO = 0;
    std::vector<int> vgroup;
    std::string s(splitting ? splitting : "");
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && s[i] == ' ') ++i;
        if (i >= s.size()) break;
        size_t j = i; while (j < s.size() && s[j] != ' ') ++j;
        std::string tok = s.substr(i, j - i);
        for (auto& c : tok) c = (char)toupper(c);
        if (tok[0] == 'V' && tok.find_first_not_of("0123456789", 1) == std::string::npos) {
            int g = -1;
            if (tok.size() > 1) {
                if (tok.size() > 3) return fail(h, -3, "force group of '" + tok + "' out of range");
                g = atoi(tok.c_str() + 1);
                if (g > 31) return fail(h, -3, "OpenMM only allows up to 32 force groups (integrators.py:1346-1347)");
            }
            tokens.push_back('V'); vgroup.push_back(g); nV++;
        }
        else if (tok == "R") { tokens.push_back('R'); nR++; }
        else if (tok == "O") { tokens.push_back('O'); nO++; }
        else if (tok == "{" || tok == "}") tokens.push_back(tok[0]);
        else return fail(h, -3, "unsupported splitting token '" + tok + "' (supported: V V<group> R O { })");
        i = j;
    }
    {
        std::vector<int> distinct;
        for (int g : vgroup) if (g >= 0 && std::find(distinct.begin(), distinct.end(), g) == distinct.end()) distinct.push_back(g);
        int counts[4] = {0, 0, 0, 0};
        if (distinct.size() > 1) {
            if (!nVg) return fail(h, -3, "multiple-time-step splittings are set with remd_set_integrator");
            size_t v = 0;
            for (auto& c : tokens) if (c == 'V') {
                const int g = vgroup[v++];
                if (g < 0) return fail(h, -3, "a multiple-time-step splitting must name the force group of every V (integrators.py:1527-1529)");
                if (g > 3) return fail(h, -3, "force groups above 3 are not supported in multiple-time-step splittings");
                c = (char)('0' + g); counts[g]++;
            }
        }
        if (nVg) for (int g = 0; g < 4; ++g) nVg[g] = counts[g];
    }
    if (tokens.empty()) return fail(h, -3, "empty splitting string");
    if (nR == 0 || nV == 0) return fail(h, -3, "splitting needs at least one R and one V (integrators.py:1376-1385)");
    int depth = 0;
    for (char c : tokens) {
        if (c == '{') { if (++depth > 1) return fail(h, -3, "nested '{' in the splitting string"); }
        else if (c == '}') { if (--depth < 0) return fail(h, -3, "'}' without '{' in the splitting string"); }
        else if (c == 'O' && depth > 0) return fail(h, -3, "O substeps cannot be Metropolized (integrators.py:1387-1401)");
    }
    if (depth != 0) return fail(h, -3, "'{' without '}' in the splitting string");
    return 0;
}

static FFTSet& thread_fft(remd_ctx* h)
{
    int t = 0;
#ifdef _OPENMP
    t = omp_get_thread_num();
#endif
    FFTSet& f = h->fft[t];
    if (h->sys.method == REMD_NB_PME && !f.f[0]) for (int k = 0; k < 3; ++k) f.f[k].reset(new FFT1D(h->sys.grid[k]));
    return f;
}

static void ensure_forces(remd_ctx* h, int r)
{
    Replica& rep = h->reps[r];
    if (rep.f_valid) return;
    const int64_t k = h->labels[h->r_begin + r];
    evaluate(h->sys, rep, h->lam_s[k], h->lam_e[k], rep.f.data(), thread_fft(h), PART_ALL, 63, (int)k);
    rep.f_valid = true;
}

static void barostat_attempt(remd_ctx* h, int r, long long attempt);

static inline uint32_t noise_key(const remd_ctx* h, int r) { return h->noise_ids.empty() ? (uint32_t)(h->r_begin + r) : h->noise_ids[r]; }

// integrators.py:1309-1317, 1404-1460 for one replica (cf. oracle/md_oracle.py:OracleLangevin.run)
static void run_steps(remd_ctx* h, int r, const std::vector<char>& tokens, int nV, int nR, int nO, int64_t iteration,
                      int64_t first_step, int n_steps, bool with_barostat = false)
{
    const System& s = h->sys;
    Replica& rep = h->reps[r];
    const int N = s.N, rg = h->r_begin + r;
    const double kT = 1.0 / h->beta[h->labels[rg]];
    const double hV = h->dt / std::max(1, nV), hR = h->dt / std::max(1, nR), hO = h->dt / std::max(1, nO);
    const double a = exp(-h->gamma * hO), b = sqrt(1.0 - exp(-2.0 * h->gamma * hO));      // integrators.py:1143, 1146
    const bool cons = !s.clusters.empty();
    std::vector<double> x1;
    if (cons) x1.resize(3 * N);
    double* x = rep.x.data(); double* v = rep.v.data();
    bool braces = false;
    for (char c : tokens) braces |= (c == '}');
    const bool m_heat = h->measure_heat != 0, m_shadow = h->measure_shadow != 0 || braces;
    auto ke = [&]() { double e = 0; for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) e += 0.5 * s.mass[i] * v[3 * i + k] * v[3 * i + k]; return e; };
    auto pe = [&]() { const int64_t kk = h->labels[rg]; return evaluate(s, rep, h->lam_s[kk], h->lam_e[kk], nullptr, thread_fft(h), PART_ALL, 63, (int)kk).total(); };
    std::vector<double> xold, vold;
    std::vector<double> fgroup[4]; bool fgroup_valid[4] = {false, false, false, false};      // forces per force group of a multiple-time-step program
    for (int st = 0; st < n_steps; ++st) {
        const int64_t gstep = iteration * (int64_t)h->n_steps + first_step + st;
        if (s.cmm > 0 && ((first_step + st) % s.cmm) == 0) {              // CMMotionRemover at the top of a step (:1313)
            double p[3] = {0, 0, 0};
            for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) p[k] += s.mass[i] * v[3 * i + k];
            for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) v[3 * i + k] -= p[k] / s.total_mass;
        }
        if (with_barostat && h->baro_frequency > 0 && ((h->baro_steps + st + 1) % h->baro_frequency) == 0) {
            // MonteCarloBarostat: every frequency-th step, in the same updateContextState slot (integrators.py:1313); the attempt
            // counter is the handle's (shared by all replicas: they attempt in lock step)
            const long long attempt = h->baro_attempts + (h->baro_steps + st + 1) / h->baro_frequency - h->baro_steps / h->baro_frequency - 1;
            barostat_attempt(h, r, attempt);
            x = rep.x.data();
            for (bool& gv : fgroup_valid) gv = false;
        }
        int oidx = 0, brace = 0;
        for (char tok : tokens) {
            if (tok == '{') {                                                                                            // :1539-1542
                xold = rep.x; vold = rep.v;
            } else if (tok == '}') {                                                                                     // :1544-1557
                uint32_t w[4];
                oracle_draw(h->seed, 7u, (uint32_t)brace, noise_key(h, r), (uint64_t)gstep, w);
                const double u = (double)(((uint64_t)w[2] << 21) | (uint64_t)(w[3] >> 11)) / 9007199254740992.0;
                rep.n_trials++;
                if (!(exp(-rep.shadow / kT) - u >= 0.0)) {
                    rep.n_rejected++;
                    for (int i = 0; i < 3 * N; ++i) { x[i] = xold[i]; v[i] = -vold[i]; }
                    rep.f_valid = false; rep.list_valid = false;
                    for (bool& gv : fgroup_valid) gv = false;
                }
                rep.shadow = 0.0;
                brace++;
            } else if (tok >= '0' && tok <= '3') {
                // kick with the forces of one force group and that group's share of the time step (integrators.py:1437-1438)
                const int g = tok - '0';
                int mask = 0;
                for (int c = 0; c < 6; ++c) if (h->force_groups[c] == g) mask |= 1 << c;
                const int64_t kk = h->labels[rg];
                // (a group's forces are evaluated when one of its kicks comes up and the positions have moved since its last
                // evaluation -- as on the device, integrate.hip: group_valid -- not once per token)
                std::vector<double>& fg = fgroup[g];
                if (!fgroup_valid[g]) {
                    fg.assign(3 * (size_t)N, 0.0);
                    evaluate(s, rep, h->lam_s[kk], h->lam_e[kk], fg.data(), thread_fft(h), PART_ALL, mask, (int)kk);
                    fgroup_valid[g] = true;
                }
                const double hg = h->dt / std::max(1, h->nVg[g]);
                const double ke0 = m_shadow ? ke() : 0.0;
                for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) v[3 * i + k] += hg * fg[3 * i + k] * s.invm[i];
                if (cons) rattle(s, x, v);
                if (m_shadow) rep.shadow += ke() - ke0;
            } else if (tok == 'V') {
                ensure_forces(h, r);
                const double* f = rep.f.data();
                const double ke0 = m_shadow ? ke() : 0.0;
                for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) v[3 * i + k] += hV * f[3 * i + k] * s.invm[i];   // :1440-1442
                if (cons) rattle(s, x, v);
                if (m_shadow) rep.shadow += ke() - ke0;                                                                  // :1444-1446
            } else if (tok == 'R') {
                const double e0 = m_shadow ? ke() + pe() : 0.0;                                                          // :1407-1409
                if (cons) {
                    for (int i = 0; i < 3 * N; ++i) x1[i] = x[i] + hR * v[i];                                            // :1414
                    std::vector<double> xc(x1);
                    shake(s, x, xc.data());                                                                              // :1416
                    for (int i = 0; i < 3 * N; ++i) { v[i] += (xc[i] - x1[i]) / hR; x[i] = xc[i]; }                        // :1417
                    rattle(s, x, v);                                                                                     // :1418
                } else for (int i = 0; i < 3 * N; ++i) x[i] += hR * v[i];
                rep.f_valid = false;
                for (bool& gv : fgroup_valid) gv = false;
                if (m_shadow) rep.shadow += ke() + pe() - e0;                                                            // :1420-1423
            } else {
                const double ke0 = m_heat ? ke() : 0.0;
                const uint64_t cnt = (uint64_t)gstep * (uint64_t)std::max(1, nO) + (uint64_t)oidx;
                for (int i = 0; i < N; ++i) {
                    double g[3]; gaussians3(h->seed, 5u, i, (int)noise_key(h, r), cnt, g);
                    const double sg = b * sqrt(kT * s.invm[i]);
                    for (int k = 0; k < 3; ++k) v[3 * i + k] = a * v[3 * i + k] + sg * g[k];                             // :1455
                }
                if (cons) rattle(s, x, v);
                oidx++;
                if (m_heat) rep.heat += ke() - ke0;                                                                      // :1457-1460
            }
        }
    }
}

static void assign_velocities(remd_ctx* h, int r, int64_t iteration)
{
    const System& s = h->sys;
    Replica& rep = h->reps[r];
    const int rg = h->r_begin + r;
    const double kT = 1.0 / h->beta[h->labels[rg]];
    for (int i = 0; i < s.N; ++i) {                                       // mcmc.py:710-711
        double g[3]; gaussians3(h->seed, 4u, i, (int)noise_key(h, r), (uint64_t)iteration, g);
        const double sg = sqrt(kT * s.invm[i]);
        for (int k = 0; k < 3; ++k) rep.v[3 * i + k] = sg * g[k];
    }
    if (!s.clusters.empty()) rattle(s, rep.x.data(), rep.v.data());
}

static bool finite_state(const Replica& rep)
{
    for (double a : rep.x) if (!std::isfinite(a)) return false;
    for (double a : rep.v) if (!std::isfinite(a)) return false;
    return true;
}

// u_kl row of one replica (states.py:911-992, 1908-1917); returns the potential at the replica's own state
static double ukl_row(remd_ctx* h, int r, double* row)
{
    const System& s = h->sys;
    Replica& rep = h->reps[r];
    FFTSet& fft = thread_fft(h);
    const int K = h->K;
    const int64_t own = h->labels[h->r_begin + r];
    const double V = rep.box[0] * rep.box[1] * rep.box[2];
    const double cscale = (h->econst_vref > 0 && V > 0) ? h->econst_vref / V : 1.0;
    bool lam_varies = false;
    for (int k = 0; k < K; ++k) if (h->lam_s[k] != h->lam_s[0] || h->lam_e[k] != h->lam_e[0]) lam_varies = true;
    if (s.reg.n > 0) {
        // general alchemical regions: everything but the custom forces once, the custom forces at every state's lambdas
        const bool whole = s.reg.exact || s.gb.n > 0;          // (GBSA: not a polynomial in lambda -- one evaluation per state too)
        const double base = whole ? 0.0 : evaluate(s, rep, 1.0, 1.0, nullptr, fft).total();
        double U_own = 0;
        for (int k = 0; k < K; ++k) {
            // (exact PME treatment: the whole Ewald sum depends on the state's lambda_electrostatics -- one evaluation per state)
            const double U = whole ? evaluate(s, rep, 1.0, 1.0, nullptr, fft, PART_ALL, 63, k).total() : base + region_energy(s, rep, k, nullptr);
            row[k] = h->beta[k] * (U + h->econst[k] * cscale + (h->pressure.empty() ? 0.0 : h->pressure[k] * V));
            if (k == own) U_own = U;
        }
        return U_own;
    }
    if (!s.has_alch || !lam_varies) {
        // one energy per replica serves every state (paralleltempering.py:206-215)
        const double U = evaluate(s, rep, h->lam_s[own], h->lam_e[own], nullptr, fft).total();
        for (int k = 0; k < K; ++k) row[k] = h->beta[k] * (U + h->econst[k] * cscale + (h->pressure.empty() ? 0.0 : h->pressure[k] * V));   // states.py:1913-1914
        return U;
    }
    // alchemical states: only the soft-core pairs depend on lambda_sterics; the electrostatic energy is an exact quadratic
    // in lambda_electrostatics (charges and exception charge products are linear in it) -> three evaluations
    const double base = evaluate(s, rep, 1.0, 1.0, nullptr, fft, PART_BONDED | PART_STERICS).total();
    double el[3] = {0, 0, 0};
    if (s.has_charge) for (int q = 0; q < 3; ++q) el[q] = evaluate(s, rep, 1.0, 0.5 * q, nullptr, fft, PART_ELEC).total();
    const double c0 = el[0], c2 = 2.0 * (el[2] - 2.0 * el[1] + el[0]), c1 = el[2] - el[0] - c2;
    std::map<double, double> sc;
    double U_own = 0;
    for (int k = 0; k < K; ++k) {
        auto it = sc.find(h->lam_s[k]);
        if (it == sc.end()) it = sc.emplace(h->lam_s[k], evaluate(s, rep, h->lam_s[k], 1.0, nullptr, fft, PART_SOFTCORE).total()).first;
        const double le = h->lam_e[k];
        const double U = base + it->second + c0 + c1 * le + c2 * le * le;
        row[k] = h->beta[k] * (U + h->econst[k] * cscale + (h->pressure.empty() ? 0.0 : h->pressure[k] * V));
        if (k == own) U_own = U;
    }
    return U_own;
}


// One Monte Carlo volume move of one replica: OpenMM's MonteCarloBarostatImpl::updateContextState restated (cf.
// oracle/md_oracle.py:OracleBarostat): dV = scale * 2 (u - 1/2); every molecule's centre (arithmetic mean, wrapped into the
// box) is scaled by s = (V'/V)^(1/3) together with the box; w = U' - U + c (1/V' - 1/V) + p dV - N_mol kT ln(V'/V); reject if
// w > 0 and u' > exp(-w / kT); after >= 10 attempts the volume step adapts (/ 1.1 below 25 %, x 1.1 capped at 0.3 V above 75 %).
static void barostat_attempt(remd_ctx* h, int r, long long attempt)
{
    const System& s = h->sys;
    Replica& rep = h->reps[r];
    FFTSet& fft = thread_fft(h);
    const int rg = h->r_begin + r, N = s.N;
    const int64_t k = h->labels[rg];
    const double kT = 1.0 / h->beta[k], p = h->pressure[k];
    const double c_lr = h->econst_vref > 0 ? h->econst[k] * h->econst_vref : 0.0;
    const double U0 = evaluate(s, rep, h->lam_s[k], h->lam_e[k], nullptr, fft, PART_ALL, 63, (int)k).total();
    const double V = rep.box[0] * rep.box[1] * rep.box[2];
    if (rep.baro[0] <= 0.0) rep.baro[0] = 0.01 * V;
    uint32_t w[4];
    oracle_draw(h->seed, 6u, 0u, noise_key(h, r), (uint64_t)attempt, w);
    auto u53 = [](uint32_t hi, uint32_t lo) { return (double)(((uint64_t)hi << 21) | (uint64_t)(lo >> 11)) / 9007199254740992.0; };
    const double dV = rep.baro[0] * 2.0 * (u53(w[2], w[3]) - 0.5);
    const double newV = V + dV, scale = cbrt(newV / V);
    const std::vector<double> x0 = rep.x;
    const double box0[3] = {rep.box[0], rep.box[1], rep.box[2]};
    for (const auto& m : s.molecules) {
        double c[3] = {0, 0, 0};
        for (int a : m) for (int q = 0; q < 3; ++q) c[q] += x0[3 * a + q];
        for (int q = 0; q < 3; ++q) {
            c[q] /= (double)m.size();
            const double cw = c[q] - floor(c[q] / box0[q]) * box0[q];
            const double shift = cw * (scale - 1.0) - (c[q] - cw);
            for (int a : m) rep.x[3 * a + q] = x0[3 * a + q] + shift;
        }
    }
    for (int q = 0; q < 3; ++q) rep.box[q] = box0[q] * scale;
    rep.list_valid = false;
    const double U1 = evaluate(s, rep, h->lam_s[k], h->lam_e[k], nullptr, fft, PART_ALL, 63, (int)k).total();
    const double wgt = U1 - U0 + c_lr * (1.0 / newV - 1.0 / V) + p * dV - (double)s.molecules.size() * kT * log(newV / V);
    oracle_draw(h->seed, 6u, 1u, noise_key(h, r), (uint64_t)attempt, w);
    const bool reject = !(wgt <= 0.0) && !(u53(w[2], w[3]) <= exp(-wgt / kT));
    if (reject) { rep.x = x0; for (int q = 0; q < 3; ++q) rep.box[q] = box0[q]; rep.list_valid = false; }
    else { rep.baro[2] += 1; rep.baro[4] += 1; }
    rep.baro[1] += 1; rep.baro[3] += 1;
    rep.f_valid = false;
    (void)N;
    if (rep.baro[1] >= 10) {
        const double Vc = rep.box[0] * rep.box[1] * rep.box[2];
        if (rep.baro[2] < 0.25 * rep.baro[1]) { rep.baro[0] /= 1.1; rep.baro[1] = 0; rep.baro[2] = 0; }
        else if (rep.baro[2] > 0.75 * rep.baro[1]) { rep.baro[0] = std::min(rep.baro[0] * 1.1, Vc * 0.3); rep.baro[1] = 0; rep.baro[2] = 0; }
    }
}

// FIREMinimizationIntegrator (openmmtools/integrators.py:2290-2469) for one replica, cf. oracle/md_oracle.py:OracleFIRE
static void fire_minimize(remd_ctx* h, int r, double ftol, int max_iterations, int* converged_out, int* iters_out)
{
    const System& s = h->sys;
    Replica& rep = h->reps[r];
    FFTSet& fft = thread_fft(h);
    const int N = s.N;
    const int64_t k = h->labels[h->r_begin + r];
    const double timestep = 0.001, alpha0 = 0.1, dt_max = 0.010, f_inc = 1.1, f_dec = 0.5, f_alpha = 0.99; const int n_min = 5;
    std::vector<double> x = rep.x, v(3 * (size_t)N, 0.0), f(3 * (size_t)N), x0, v0, f0, x1(3 * (size_t)N), fn(3 * (size_t)N);   // :2341
    double dt = timestep, alpha = alpha0; int n_neg = 0; bool converged = false;
    const double ndof = 3.0 * N;
    const bool cons = !s.clusters.empty();
    auto ef = [&](const std::vector<double>& y, double* fo) { rep.x = y; rep.list_valid = rep.list_valid; return evaluate(s, rep, h->lam_s[k], h->lam_e[k], fo, fft, PART_ALL, 63, (int)k).total(); };
    double E = ef(x, f.data());
    int it = 0;
    const int limit = max_iterations > 0 ? max_iterations : 200000;
    while (it < limit) {
        double f2 = 0; for (double a : f) f2 += a * a;
        if (sqrt(f2) / ndof <= ftol) converged = true;                                   // :2377-2386
        if (converged) { if (max_iterations == 0) break; ++it; continue; }
        x0 = x; v0 = v; f0 = f; const double E0 = E;                                     // :2392-2394
        for (int i = 0; i < N; ++i) for (int q = 0; q < 3; ++q) v[3 * i + q] += 0.5 * dt * f[3 * i + q] * s.invm[i];   // :2397
        for (int i = 0; i < 3 * N; ++i) x1[i] = x[i] + dt * v[i];                        // :2398-2399
        std::vector<double> xn = x1;
        if (cons) shake(s, x.data(), xn.data());                                         // :2400
        const double En = ef(xn, fn.data());
        for (int i = 0; i < N; ++i) for (int q = 0; q < 3; ++q)
            v[3 * i + q] += 0.5 * dt * fn[3 * i + q] * s.invm[i] + (xn[3 * i + q] - x1[3 * i + q]) / dt;   // :2401
        if (cons) rattle(s, xn.data(), v.data());                                        // :2402
        const double dE = En - E0;                                                       // :2404
        double fmag = 0, vmag = 0, P = 0;
        for (int i = 0; i < 3 * N; ++i) { fmag += fn[i] * fn[i]; vmag += v[i] * v[i]; P += fn[i] * v[i]; }   // :2408-2416
        fmag = sqrt(fmag); vmag = sqrt(vmag);
        if (fmag > 0) for (int i = 0; i < 3 * N; ++i) v[i] = (1.0 - alpha) * v[i] + alpha * (fn[i] / fmag) * vmag;   // :2421
        x = xn; E = En; f = fn;
        if (!(dE < 0)) { x = x0; v = v0; E = E0; f = f0; P = -1.0; }                     // :2423-2431
        if (dt <= 1.0e-5 * timestep) converged = true;                                   // :2433-2437
        if (P > 0) { n_neg += 1; if (n_neg > n_min) { dt = std::min(dt * f_inc, dt_max); alpha *= f_alpha; } }   // :2439-2449
        if (P < 0) { n_neg = 0; dt *= f_dec; std::fill(v.begin(), v.end(), 0.0); alpha = alpha0; }               // :2451-2458
        ++it;
    }
    rep.x = x; rep.v = v; rep.f_valid = false; rep.list_valid = false;
    *converged_out = converged ? 1 : 0; *iters_out = it;
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
    (void)device; (void)stream;
    if (!out) return fail(nullptr, -1, "remd_create: out is NULL");
    remd_ctx* h = new remd_ctx();
#ifdef _OPENMP
    h->n_threads = omp_get_max_threads();
#endif
    h->fft.resize(std::max(256, h->n_threads));
    *out = h;
    return 0;
}

int remd_destroy(remd_handle h)
{
    if (g_time) fprintf(stderr, "[remd_cpu] ms: list %.1f pairs %.1f pme %.1f\n", g_timers.list, g_timers.pairs, g_timers.pme);
    delete h; return 0;
}
int remd_seed(remd_handle h, uint64_t seed) { if (!h) return -1; h->seed = seed; return 0; }

int remd_set_system(remd_handle h, const remd_system_desc* d)
{
    if (h) for (int c = 0; c < 6; ++c) h->force_groups[c] = 0;           // until remd_set_force_groups says otherwise (as libremd_hip.so)
    if (!h || !d) return fail(h, -1, "remd_set_system: NULL argument");
    if (d->n_atoms <= 0 || !d->mass) return fail(h, -1, "remd_set_system: n_atoms/mass missing");
    System s;
    const int N = s.N = d->n_atoms;
    s.mass.assign(d->mass, d->mass + N); s.invm.resize(N);
    for (int i = 0; i < N; ++i) {
        if (!(s.mass[i] > 0)) return fail(h, -3, "massless particles are not supported");
        s.invm[i] = 1.0 / s.mass[i]; s.total_mass += s.mass[i];
    }
    s.ext_atoms.assign(d->ext_atoms, d->ext_atoms + d->n_ext); s.ext_K = d->ext_K; s.ext_x0 = d->ext_x0; s.ext_U0 = d->ext_U0;
    s.bond_atoms.assign(d->bond_atoms, d->bond_atoms + 2 * (size_t)d->n_bonds); s.bond_params.assign(d->bond_params, d->bond_params + 2 * (size_t)d->n_bonds);
    s.angle_atoms.assign(d->angle_atoms, d->angle_atoms + 3 * (size_t)d->n_angles); s.angle_params.assign(d->angle_params, d->angle_params + 2 * (size_t)d->n_angles);
    s.torsion_atoms.assign(d->torsion_atoms, d->torsion_atoms + 4 * (size_t)d->n_torsions); s.torsion_params.assign(d->torsion_params, d->torsion_params + 3 * (size_t)d->n_torsions);
    s.method = d->nb_method; s.rc = d->cutoff; s.rs = d->switch_distance > 0 ? d->switch_distance : -1.0;
    if (d->nb_method == REMD_NB_NOCUTOFF) { s.nocut = true; s.method = 0; s.rc = 1e18; s.rs = -1.0; }
    s.rcc = s.rc;
    if (s.method == REMD_NB_PME && h->coulomb_cutoff > 0.0) {
        if (h->coulomb_cutoff < s.rc) return fail(h, -1, "the Coulomb cutoff of remd_set_coulomb_cutoff is shorter than the NonbondedForce cutoff");
        s.rcc = h->coulomb_cutoff;
    }
    s.annihilate = h->annihilate_sterics != 0;
    s.rf_unshifted = h->rf_unshifted != 0; s.rf_switch_width = h->rf_switch_width;
    s.rf_eps = d->rf_dielectric; s.alpha = d->ewald_alpha; s.use_disp = d->use_dispersion_correction;
    for (int k = 0; k < 3; ++k) s.grid[k] = d->pme_grid[k];
    s.q.assign(N, 0.0); s.sig.assign(N, 1.0); s.eps.assign(N, 0.0); s.alch.assign(N, 0);
    if (s.method || s.nocut) {
        if (!d->charge || !d->sigma || !d->epsilon) return fail(h, -1, "remd_set_system: nonbonded parameters missing");
        s.q.assign(d->charge, d->charge + N); s.sig.assign(d->sigma, d->sigma + N); s.eps.assign(d->epsilon, d->epsilon + N);
    }
    for (double q : s.q) if (q != 0.0) s.has_charge = true;
    if (s.nocut && d->n_alch > 0) return fail(h, -3, "NoCutoff: alchemical atoms go through remd_set_alchemical_regions (the descriptor's one-region path needs a cutoff method)");
    for (int a = 0; a < d->n_alch; ++a) { s.alch[d->alch_atoms[a]] = 1; s.has_alch = true; }
    s.sc_alpha = d->softcore_alpha; s.sc_a = d->softcore_a; s.sc_b = d->softcore_b; s.sc_c = d->softcore_c;
    s.exc_atoms.assign(d->exception_atoms, d->exception_atoms + 2 * (size_t)d->n_exceptions);
    s.exc_params.assign(d->exception_params, d->exception_params + 3 * (size_t)d->n_exceptions);
    s.excl.assign(N, {});
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = std::min(s.exc_atoms[2 * e], s.exc_atoms[2 * e + 1]), j = std::max(s.exc_atoms[2 * e], s.exc_atoms[2 * e + 1]);
        s.excl[i].push_back(j);
    }
    for (auto& v : s.excl) std::sort(v.begin(), v.end());
    int n_con = 0;
    for (int w = 0; w < d->n_settle; ++w) {
        Cluster c{}; c.n_atoms = 3; for (int k = 0; k < 3; ++k) c.atoms[k] = d->settle_atoms[3 * w + k];
        c.nc = 3; c.ci[0] = 0; c.cj[0] = 1; c.d[0] = d->settle_dOH; c.ci[1] = 0; c.cj[1] = 2; c.d[1] = d->settle_dOH; c.ci[2] = 1; c.cj[2] = 2; c.d[2] = d->settle_dHH;
        c.settle = 1;
        s.clusters.push_back(c); n_con += 3;
    }
    for (int w = 0; w < d->n_shake; ++w) {
        Cluster c{}; c.n_atoms = 1; c.atoms[0] = d->shake_atoms[4 * w]; c.nc = 0;
        for (int k = 1; k < 4; ++k) if (d->shake_atoms[4 * w + k] >= 0) {
            c.atoms[c.n_atoms] = d->shake_atoms[4 * w + k]; c.ci[c.nc] = 0; c.cj[c.nc] = c.n_atoms; c.d[c.nc] = d->shake_dist[3 * w + k - 1];
            c.n_atoms++; c.nc++; n_con++;
        }
        s.clusters.push_back(c);
    }
    if (d->n_settle > 0) {
        // Miyamoto & Kollman geometry constants of the rigid water (all waters share masses and distances)
        const int* a = d->settle_atoms;
        s.settle_mO = d->mass[a[0]]; s.settle_mH = d->mass[a[1]];
        s.settle_rc = 0.5 * d->settle_dHH;
        const double t = sqrt(d->settle_dOH * d->settle_dOH - s.settle_rc * s.settle_rc);
        s.settle_ra = 2.0 * s.settle_mH * t / (s.settle_mO + 2.0 * s.settle_mH); s.settle_rb = t - s.settle_ra; s.settle_dHH = d->settle_dHH;
        for (int w = 0; w < d->n_settle; ++w)
            if (d->mass[a[3 * w]] != s.settle_mO || d->mass[a[3 * w + 1]] != s.settle_mH || d->mass[a[3 * w + 2]] != s.settle_mH)
                return fail(h, -3, "remd_set_system: SETTLE waters must share one set of masses");
    }
    s.cmm = d->cmm_frequency;
    s.n_dof = 3 * N - n_con - (s.cmm > 0 ? 3 : 0);
    if (s.method == REMD_NB_PME) {
        for (int k = 0; k < 3; ++k) { if (s.grid[k] < PME_ORDER) return fail(h, -1, "remd_set_system: PME mesh too small"); s.bmod[k] = bspline_moduli(s.grid[k]); }
    }
    s.disp_coeff = (s.method && s.use_disp) ? dispersion_coefficient(s) : 0.0;
    {   // molecules = connected components over exceptions, bonds and constraints (what the barostat scales as units)
        std::vector<int> parent(N);
        for (int i = 0; i < N; ++i) parent[i] = i;
        auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
        auto unite = [&](int a, int b) { const int ra = find(a), rb = find(b); if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); };
        for (int e = 0; e < d->n_exceptions; ++e) unite(d->exception_atoms[2 * e], d->exception_atoms[2 * e + 1]);
        for (int b2 = 0; b2 < d->n_bonds; ++b2) unite(d->bond_atoms[2 * b2], d->bond_atoms[2 * b2 + 1]);
        for (const Cluster& c : s.clusters) for (int q = 1; q < c.n_atoms; ++q) unite(c.atoms[0], c.atoms[q]);
        std::map<int, std::vector<int>> groups;
        for (int i = 0; i < N; ++i) groups[find(i)].push_back(i);
        for (auto& g : groups) s.molecules.push_back(g.second);
    }
    h->sys = std::move(s);
    for (auto& f : h->fft) for (int k = 0; k < 3; ++k) f.f[k].reset();
    h->has_system = true;
    for (auto& r : h->reps) { r.f_valid = false; r.list_valid = false; }
    return 0;
}

int remd_set_states(remd_handle h, int K, const double* beta, const double* lam_s, const double* lam_e, const double* econst)
{
    if (!h || K <= 0 || !beta) return fail(h, -1, "remd_set_states: bad arguments");
    for (int k = 0; k < K; ++k) if (!(beta[k] > 0)) return fail(h, -1, "remd_set_states: beta must be > 0");
    h->K = K;
    h->beta.assign(beta, beta + K);
    h->lam_s.assign(K, 1.0); h->lam_e.assign(K, 1.0); h->econst.assign(K, 0.0);
    if (lam_s) h->lam_s.assign(lam_s, lam_s + K);
    if (lam_e) h->lam_e.assign(lam_e, lam_e + K);
    if (econst) h->econst.assign(econst, econst + K);
    h->ukl.assign((size_t)std::max(0, h->R_global) * K, 0.0);
    for (auto& r : h->reps) r.f_valid = false;
    return 0;
}

int remd_set_integrator(remd_handle h, const char* splitting, double dt, double gamma, int n_steps, int reassign, double tol)
{
    (void)tol;                                     // the f64 solver always iterates to 1e-12
    if (!h) return -1;
    if (!(dt > 0) || n_steps < 0 || gamma < 0) return fail(h, -1, "remd_set_integrator: bad parameters");
    int rc = parse_splitting(h, splitting, h->tokens, h->nV, h->nR, h->nO, h->nVg);
    if (rc) return rc;
    h->dt = dt; h->gamma = gamma; h->n_steps = n_steps; h->reassign = reassign; h->has_integrator = true;
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
    for (auto& rep : h->reps) { rep.heat = 0; rep.shadow = 0; rep.n_trials = 0; rep.n_rejected = 0; }
    return 0;
}
int remd_get_work(remd_handle h, double* heat, double* shadow_work, int64_t* n_accepted, int64_t* n_trials)
{
    if (!h || h->R <= 0) return fail(h, -1, "remd_get_work: no replicas");
    for (int r = 0; r < h->R; ++r) {
        const Replica& rep = h->reps[r];
        if (heat) heat[r] = rep.heat;
        if (shadow_work) shadow_work[r] = rep.shadow;
        if (n_trials) n_trials[r] = rep.n_trials;
        if (n_accepted) n_accepted[r] = rep.n_trials - rep.n_rejected;
    }
    return 0;
}
int remd_set_force_groups(remd_handle h, const int32_t* groups)
{
    if (!h || !groups) return fail(h, -1, "remd_set_force_groups: bad arguments");
    for (int c = 0; c < 6; ++c) if (groups[c] < 0 || groups[c] > 31) return fail(h, -1, "remd_set_force_groups: force groups are 0 ... 31");
    for (int c = 0; c < 6; ++c) h->force_groups[c] = groups[c];
    return 0;
}

// Sharding collectives (include/remd_hip.h): the CPU library is the single-process checker -- it joins a world of one.
int remd_comm_unique_id(void* id)
{
    if (!id) return fail(nullptr, -1, "remd_comm_unique_id: null pointer");
    memset(id, 0, REMD_COMM_ID_BYTES);
    return 0;
}
int remd_comm_init(remd_handle h, int rank, int world, const void* id)
{
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return fail(h, -1, "remd_comm_init: bad arguments");
    if (world != 1) return fail(h, -1, "remd_comm_init: several ranks are not implemented in the CPU library");
    return 0;
}
int remd_comm_all_gather_energies(remd_handle h)
{
    if (!h) return fail(h, -1, "remd_comm_all_gather_energies: null handle");
    if (h->R != h->R_global) return fail(h, -1, "remd_comm_all_gather_energies: replicas are sharded but the CPU library has no communicator");
    return 0;
}
int remd_comm_finalize(remd_handle) { return 0; }

int remd_test_coulomb_table(double alpha, double coulomb_cutoff_nm, int n, const float* u, float* minus_G)
{
    // the closed form the device's table interpolates (include/remd_hip.h), f64
    if (!(alpha > 0) || !(coulomb_cutoff_nm > 0) || n < 0 || !u || !minus_G) return fail(nullptr, -1, "remd_test_coulomb_table: bad arguments");
    for (int k = 0; k < n; ++k) {
        const double uu = u[k], r = sqrt(uu);
        minus_G[k] = (float)(-(erfc(alpha * r) / r + 2.0 * alpha / sqrt(PI) * exp(-alpha * alpha * uu)) / uu);
    }
    return 0;
}

int remd_set_replica_ids(remd_handle h, const int64_t* ids)
{
    if (!h || h->reps.empty()) return fail(h, -1, "remd_set_replica_ids: call remd_set_replicas first");
    h->noise_ids.clear();
    if (!ids) return 0;
    for (size_t r = 0; r < h->reps.size(); ++r) {
        if (ids[r] < 0 || ids[r] > 0xffffffffll) { h->noise_ids.clear(); return fail(h, -1, "remd_set_replica_ids: ids must fit 32 bits"); }
        h->noise_ids.push_back((uint32_t)ids[r]);
    }
    return 0;
}

int remd_set_alchemical_options(remd_handle h, int annihilate_sterics)
{
    if (!h || (annihilate_sterics != 0 && annihilate_sterics != 1)) return fail(h, -1, "remd_set_alchemical_options: bad arguments");
    h->annihilate_sterics = annihilate_sterics;   // consumed by the next remd_set_system (include/remd_hip.h)
    return 0;
}

// general alchemical regions (include/remd_hip.h): after remd_set_system, which forgets them
int remd_set_alchemical_regions(remd_handle h, const remd_alch_regions_desc* d)
{
    if (!h) return fail(h, -1, "remd_set_alchemical_regions: NULL handle");
    if (h->sys.reg.exact) return fail(h, -2, "remd_set_alchemical_regions: call remd_set_system again first (the exact PME treatment changed the system's tables)");
    h->sys.reg = System::Regions();
    for (auto& r : h->reps) r.f_valid = false;
    if (!d || d->n_regions == 0) return 0;
    System& s = h->sys;
    if (!h->has_system) return fail(h, -2, "remd_set_alchemical_regions: call remd_set_system first");
    const int N = s.N, n = d->n_regions;
    if (d->n_atoms != N) return fail(h, -1, "remd_set_alchemical_regions: n_atoms differs from the system's");
    if (n < 0 || n > 64 || !d->region_of_atom || !d->softcore || !d->annihilate || !d->charge || !d->sigma || !d->epsilon ||
        d->n_interactions < 0 || (d->n_interactions > 0 && !d->interactions) || d->n_exceptions < 0 || (d->n_exceptions > 0 && (!d->exception_atoms || !d->exception_params)))
        return fail(h, -1, "remd_set_alchemical_regions: bad arguments");
    if (s.method == 0 && !s.nocut) return fail(h, -3, "alchemical regions need a NonbondedForce");
    if (s.has_alch) return fail(h, -3, "alchemical regions: the descriptor of remd_set_system must be the factory's NonbondedForce (n_alch = 0)");
    System::Regions g;
    g.n = n;
    g.softcore.assign(d->softcore, d->softcore + 8 * (size_t)n);
    g.annihilate.assign(d->annihilate, d->annihilate + 2 * (size_t)n);
    for (int q = 0; q < n; ++q) if (!(g.softcore[8 * (size_t)q + 4] > 0) || !(g.softcore[8 * (size_t)q + 7] > 0)) return fail(h, -1, "alchemical regions: softcore_c and softcore_f must be positive");
    g.cls_of.assign((size_t)(n + 1) * (n + 1), -1);
    for (int q = 1; q <= n; ++q) {
        g.cls_of[q] = g.cls_of[(size_t)q * (n + 1)] = (int)g.classes.size(); g.classes.push_back({0, q, q, q});
        g.cls_of[(size_t)q * (n + 1) + q] = (int)g.classes.size(); g.classes.push_back({1, q, q, q});
    }
    std::vector<char> interacting((size_t)(n + 1) * (n + 1), 0);
    for (int k = 0; k < d->n_interactions; ++k) {
        const int a = d->interactions[2 * k], b = d->interactions[2 * k + 1];
        if (a < 1 || b < 1 || a > n || b > n || a == b) return fail(h, -1, "alchemical regions: bad pair of interacting regions");
        interacting[(size_t)a * (n + 1) + b] = interacting[(size_t)b * (n + 1) + a] = 1;
        if (d->exact_pme) continue;                // exact PME: the pair sees each other's scaled charges; no sterics (tables zeroed, alchemy.py:1886-1911)
        if (g.cls_of[(size_t)a * (n + 1) + b] >= 0) continue;
        g.cls_of[(size_t)a * (n + 1) + b] = g.cls_of[(size_t)b * (n + 1) + a] = (int)g.classes.size(); g.classes.push_back({2, a, b, b});
    }
    if (d->exact_pme && s.method != REMD_NB_PME) return fail(h, -1, "alchemical regions: exact_pme needs a PME system");
    g.region_of.assign(d->region_of_atom, d->region_of_atom + N);
    g.q.assign(d->charge, d->charge + N); g.sig.assign(d->sigma, d->sigma + N); g.eps.assign(d->epsilon, d->epsilon + N);
    for (int i = 0; i < N; ++i) {
        if (g.region_of[i] < 0 || g.region_of[i] > n) return fail(h, -1, "alchemical regions: region index out of range");
        if (g.region_of[i] > 0) { g.alch.push_back(i); if (!(g.sig[i] > 0)) return fail(h, -1, "alchemical regions: sigma must be positive (the factory sets 0 to 0.1 nm, alchemy.py:1638-1648)"); }
    }
    if (g.alch.empty()) return fail(h, -1, "alchemical regions: no alchemical atom");
    std::vector<int> ord(N, -1);
    for (size_t k = 0; k < g.alch.size(); ++k) ord[g.alch[k]] = (int)k;
    g.skip.assign(g.alch.size(), std::vector<char>(N, 0));
    for (size_t ia = 0; ia < g.alch.size(); ++ia) {
        const int a = g.alch[ia], ga = g.region_of[a];
        for (int j = 0; j < N; ++j) {
            const int gj = g.region_of[j];
            if (j == a || (gj > 0 && j < a) || g.cls_of[(size_t)ga * (n + 1) + gj] < 0) g.skip[ia][j] = 1;
            else if (gj == 0 && !(g.sig[j] > 0) && (g.eps[j] != 0.0 || (d->electrostatics && g.q[j] != 0.0)))
                return fail(h, -1, "alchemical regions: sigma must be positive (the factory sets 0 to 0.1 nm, alchemy.py:1638-1648)");
        }
    }
    for (size_t e = 0; e < s.exc_atoms.size() / 2; ++e) {          // every exception of the system is an exclusion of the custom forces (alchemy.py:1944-1947)
        const int i = s.exc_atoms[2 * e], j = s.exc_atoms[2 * e + 1];
        if (ord[i] >= 0) g.skip[ord[i]][j] = 1;
        if (ord[j] >= 0) g.skip[ord[j]][i] = 1;
    }
    g.elec = d->electrostatics != 0 && !d->exact_pme;
    g.exact = d->exact_pme != 0;
    g.consistent_exc = d->consistent_exceptions != 0 && g.elec;
    if (g.exact) {
        for (int i = 0; i < N; ++i) if (g.region_of[i] > 0) { s.q[i] = g.q[i]; if (g.q[i] != 0.0) s.has_charge = true; }
        g.exc_region.assign(s.exc_atoms.size() / 2, 0);
        std::map<std::pair<int, int>, size_t> where;
        for (size_t e = 0; e < s.exc_atoms.size() / 2; ++e) where[{std::min(s.exc_atoms[2 * e], s.exc_atoms[2 * e + 1]), std::max(s.exc_atoms[2 * e], s.exc_atoms[2 * e + 1])}] = e;
        for (int e = 0; e < d->n_exceptions; ++e) {
            const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
            auto it = where.find({std::min(i, j), std::max(i, j)});
            if (it == where.end()) return fail(h, -1, "alchemical regions: an exception that the system does not have");
            const int gi = g.region_of[i], gj = g.region_of[j];
            if (gi == 0 && gj == 0) continue;
            s.exc_params[3 * it->second] = d->exception_params[3 * e];            // the charge product comes back as an offset (LJ: custom bonds)
            g.exc_region[it->second] = (gi > 0 && gj > 0) ? std::min(gi, gj) : std::max(gi, gj);
        }
        // regions that do not interact exclude each other (alchemy.py:1663-1672)
        for (size_t ia = 0; ia < g.alch.size(); ++ia) for (size_t ib = ia + 1; ib < g.alch.size(); ++ib) {
            const int a = g.alch[ia], b = g.alch[ib], ga = g.region_of[a], gb = g.region_of[b];
            if (ga == gb || interacting[(size_t)ga * (n + 1) + gb] || where.count({std::min(a, b), std::max(a, b)})) continue;
            s.exc_atoms.push_back(a); s.exc_atoms.push_back(b);
            s.exc_params.push_back(0.0); s.exc_params.push_back(1.0); s.exc_params.push_back(0.0);
            g.exc_region.push_back(0);
            s.excl[std::min(a, b)].push_back(std::max(a, b));
        }
        for (auto& v : s.excl) std::sort(v.begin(), v.end());
        for (auto& r : h->reps) r.list_valid = false;
    }
    for (int e = 0; e < d->n_exceptions; ++e) {
        const int i = d->exception_atoms[2 * e], j = d->exception_atoms[2 * e + 1];
        if (i < 0 || j < 0 || i >= N || j >= N || i == j) return fail(h, -1, "alchemical regions: bad exception pair");
        const int gi = g.region_of[i], gj = g.region_of[j];
        const double qq = d->exception_params[3 * e], sg = d->exception_params[3 * e + 1], ep = d->exception_params[3 * e + 2];
        if ((gi == 0 && gj == 0) || (ep == 0.0 && (qq == 0.0 || !g.elec))) continue;
        if (!(sg > 0)) return fail(h, -1, "alchemical regions: exception sigma must be positive");
        g.exc_atoms.push_back(i); g.exc_atoms.push_back(j);
        g.exc_params.push_back(qq); g.exc_params.push_back(sg); g.exc_params.push_back(ep);
        // (an exception between two regions: the first region's (environment, region) bond force, alchemy.py:1972-1976, 1992-2006)
        g.exc_cls.push_back((gi > 0 && gj > 0 && gi != gj) ? g.cls_of[std::min(gi, gj)] : g.cls_of[(size_t)gi * (n + 1) + gj]);
    }
    {
        auto bad = [&](int cnt, int width, const int32_t* atoms, const int32_t* reg) {
            for (int k = 0; k < cnt; ++k) { if (reg[k] < 1 || reg[k] > n) return true; for (int q = 0; q < width; ++q) if (atoms[width * k + q] < 0 || atoms[width * k + q] >= N) return true; }
            return false;
        };
        if (d->n_bonds < 0 || d->n_angles < 0 || d->n_torsions < 0 || (d->n_bonds > 0 && (!d->bond_atoms || !d->bond_params || !d->bond_region)) ||
            (d->n_angles > 0 && (!d->angle_atoms || !d->angle_params || !d->angle_region)) || (d->n_torsions > 0 && (!d->torsion_atoms || !d->torsion_params || !d->torsion_region)))
            return fail(h, -1, "alchemical regions: bad softened bonded terms");
        if (bad(d->n_bonds, 2, d->bond_atoms, d->bond_region) || bad(d->n_angles, 3, d->angle_atoms, d->angle_region) || bad(d->n_torsions, 4, d->torsion_atoms, d->torsion_region))
            return fail(h, -1, "alchemical regions: softened bonded term with a bad atom or region");
        g.bond_atoms.assign(d->bond_atoms, d->bond_atoms + 2 * (size_t)d->n_bonds); g.bond_params.assign(d->bond_params, d->bond_params + 2 * (size_t)d->n_bonds);
        g.bond_region.assign(d->bond_region, d->bond_region + d->n_bonds);
        g.angle_atoms.assign(d->angle_atoms, d->angle_atoms + 3 * (size_t)d->n_angles); g.angle_params.assign(d->angle_params, d->angle_params + 2 * (size_t)d->n_angles);
        g.angle_region.assign(d->angle_region, d->angle_region + d->n_angles);
        g.torsion_atoms.assign(d->torsion_atoms, d->torsion_atoms + 4 * (size_t)d->n_torsions); g.torsion_params.assign(d->torsion_params, d->torsion_params + 3 * (size_t)d->n_torsions);
        g.torsion_region.assign(d->torsion_region, d->torsion_region + d->n_torsions);
    }
    g.alpha = d->elec_alpha; g.krf = d->elec_krf; g.crf = d->elec_crf;
    g.rs_e = (g.elec && d->elec_switch_distance >= 0 && d->elec_switch_distance < s.rc) ? d->elec_switch_distance : -1.0;
    s.reg = std::move(g);
    return 0;
}

int remd_set_region_lambdas(remd_handle h, int K, int n_regions, const double* ls, const double* le)
{
    if (!h) return fail(h, -1, "remd_set_region_lambdas: NULL handle");
    System::Regions& g = h->sys.reg;
    if (g.n == 0) return fail(h, -2, "remd_set_region_lambdas: no alchemical regions on this handle");
    if (K != h->K || n_regions != g.n || !ls || !le) return fail(h, -1, "remd_set_region_lambdas: K / n_regions differ from remd_set_states / remd_set_alchemical_regions");
    for (size_t k = 0; k < (size_t)K * g.n; ++k)
        if (!(ls[k] >= 0.0 && ls[k] <= 1.0 && le[k] >= 0.0 && le[k] <= 1.0)) return fail(h, -1, "remd_set_region_lambdas: lambdas must be in [0, 1]");
    g.ls.assign(ls, ls + (size_t)K * g.n); g.le.assign(le, le + (size_t)K * g.n); g.K = K;
    g.bl.clear();                                             // until remd_set_region_bonded_lambdas says otherwise: 1
    for (auto& r : h->reps) r.f_valid = false;
    return 0;
}

int remd_set_region_bonded_lambdas(remd_handle h, int K, int n_regions, const double* lb, const double* la, const double* lt)
{
    if (!h) return fail(h, -1, "remd_set_region_bonded_lambdas: NULL handle");
    System::Regions& g = h->sys.reg;
    if (g.n == 0) return fail(h, -2, "remd_set_region_bonded_lambdas: no alchemical regions on this handle");
    if (K != h->K || K != g.K || n_regions != g.n) return fail(h, -1, "remd_set_region_bonded_lambdas: call remd_set_region_lambdas first (same K, n_regions)");
    const double* src[3] = {lb, la, lt};
    g.bl.assign(3 * (size_t)K * g.n, 1.0);
    for (int q = 0; q < 3; ++q) for (int k = 0; k < K; ++k) for (int r = 0; r < g.n; ++r) {
        const double v = src[q] ? src[q][(size_t)k * g.n + r] : 1.0;
        if (!(v >= 0.0 && v <= 1.0)) return fail(h, -1, "remd_set_region_bonded_lambdas: lambdas must be in [0, 1]");
        g.bl[((size_t)k * 3 + q) * g.n + r] = v;
    }
    for (auto& r : h->reps) r.f_valid = false;
    return 0;
}

int remd_set_gbsa(remd_handle h, const remd_gbsa_desc* d)
{
    if (!h) return fail(h, -1, "remd_set_gbsa: NULL handle");
    h->sys.gb = System::GB();
    for (auto& r : h->reps) r.f_valid = false;
    if (!d) return 0;
    if (!h->has_system || !h->sys.nocut) return fail(h, -3, "remd_set_gbsa: GBSA needs a system with a NoCutoff NonbondedForce (call remd_set_system first)");
    if (d->n_atoms != h->sys.N || !d->charge || !d->radius || !d->scale || !(d->solute_dielectric > 0) || !(d->solvent_dielectric > 0)) return fail(h, -1, "remd_set_gbsa: bad arguments");
    System::GB g;
    g.n = d->n_atoms;
    g.q.assign(d->charge, d->charge + g.n); g.R.assign(d->radius, d->radius + g.n); g.sc.assign(d->scale, d->scale + g.n);
    g.alch.assign(g.n, 0);
    for (int i = 0; i < g.n; ++i) { if (!(g.R[i] > 0.009)) return fail(h, -1, "remd_set_gbsa: radii must exceed the offset 0.009 nm"); if (d->alchemical && d->alchemical[i]) g.alch[i] = 1; }
    g.tau = 1.0 / d->solute_dielectric - 1.0 / d->solvent_dielectric;
    g.sasa = d->surface_area != 0;
    h->sys.gb = std::move(g);
    return 0;
}

int remd_set_reaction_field(remd_handle h, int unshifted, double switch_width_nm)
{
    if (!h || (unshifted != 0 && unshifted != 1) || !(switch_width_nm >= 0.0)) return fail(h, -1, "remd_set_reaction_field: bad arguments");
    h->rf_unshifted = unshifted; h->rf_switch_width = switch_width_nm;      // consumed by the next remd_set_system (include/remd_hip.h)
    return 0;
}

int remd_set_coulomb_cutoff(remd_handle h, double coulomb_cutoff_nm)
{
    if (!h || !(coulomb_cutoff_nm >= 0.0)) return fail(h, -1, "remd_set_coulomb_cutoff: bad arguments");
    h->coulomb_cutoff = coulomb_cutoff_nm;        // consumed by the next remd_set_system (include/remd_hip.h)
    return 0;
}

int remd_set_restart_attempts(remd_handle h, int n)
{
    if (!h || n < 0) return fail(h, -1, "remd_set_restart_attempts: bad arguments");
    h->n_restart_attempts = n;
    return 0;
}

int remd_set_barostat(remd_handle h, int K, const double* pressure, int frequency)
{
    if (!h) return -1;
    if (!pressure || frequency <= 0) { h->baro_frequency = 0; h->pressure.clear(); return 0; }
    if (K != h->K) return fail(h, -1, "remd_set_barostat: K differs from remd_set_states");
    for (int k = 0; k < K; ++k) if (!(pressure[k] == pressure[k])) return fail(h, -1, "remd_set_barostat: NaN pressure");
    h->pressure.assign(pressure, pressure + K);
    h->baro_frequency = frequency;
    return 0;
}
int remd_get_boxes(remd_handle h, double* box)
{
    if (!h || !box || h->R <= 0) return fail(h, -1, "remd_get_boxes: replicas not set");
    for (int r = 0; r < h->R; ++r) for (int k = 0; k < 3; ++k) box[3 * r + k] = h->reps[r].box[k];
    return 0;
}
int remd_set_energy_const_volume(remd_handle h, double v) { if (!h || !(v >= 0)) return fail(h, -1, "remd_set_energy_const_volume: bad arguments"); h->econst_vref = v; return 0; }
int remd_get_barostat_stats(remd_handle h, double* vs, int64_t* na, int64_t* nc)
{
    if (!h || h->R <= 0) return fail(h, -1, "remd_get_barostat_stats: replicas not set");
    for (int r = 0; r < h->R; ++r) { if (vs) vs[r] = h->reps[r].baro[0]; if (na) na[r] = (int64_t)h->reps[r].baro[3]; if (nc) nc[r] = (int64_t)h->reps[r].baro[4]; }
    return 0;
}
int remd_barostat_attempts(remd_handle h, int n)
{
    if (!h) return -1;
    if (n < 0) return fail(h, -1, "remd_barostat_attempts: negative n_attempts");
    if (h->R <= 0) return fail(h, -1, "remd_barostat_attempts: replicas not set");
    if (h->baro_frequency <= 0 || h->pressure.empty()) return fail(h, -3, "remd_barostat_attempts: no barostat (remd_set_barostat)");
    for (int a = 0; a < n; ++a) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int r = 0; r < h->R; ++r) barostat_attempt(h, r, h->baro_attempts);
        h->baro_attempts += 1;
    }
    return 0;
}
int remd_minimize(remd_handle h, double tol, int maxit, int32_t* conv, int32_t* nit)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return fail(h, -1, "remd_minimize: system/states/replicas not all set");
    if (!(tol >= 0) || maxit < 0) return fail(h, -1, "remd_minimize: bad arguments");
    std::vector<int> c(h->R, 0), n(h->R, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < h->R; ++r) fire_minimize(h, r, tol, maxit, &c[r], &n[r]);
    int mx = 0;
    for (int r = 0; r < h->R; ++r) { if (conv) conv[r] = c[r]; mx = std::max(mx, n[r]); }
    if (nit) *nit = mx;
    return 0;
}

int remd_set_labels(remd_handle h, const int64_t* labels)
{
    if (!h || !labels || h->R_global <= 0) return fail(h, -1, "remd_set_labels: replicas not set");
    for (int r = 0; r < h->R_global; ++r)
        if (labels[r] < 0 || (h->K > 0 && labels[r] >= h->K)) return fail(h, -1, "remd_set_labels: label out of range");
    bool changed = h->labels.size() != (size_t)h->R_global;
    for (int r = 0; r < h->R_global && !changed; ++r) if (h->labels[r] != labels[r]) changed = true;
    h->labels.assign(labels, labels + h->R_global);
    if (changed && h->sys.has_alch) for (auto& r : h->reps) r.f_valid = false;     // forces depend on the state's lambdas
    return 0;
}

int remd_set_replicas(remd_handle h, int R_global, int r_begin, int R_local, const double* x, const double* v, const double* box,
                      const int64_t* labels)
{
    if (!h || !h->has_system) return fail(h, -1, "remd_set_replicas: call remd_set_system first");
    if (R_global <= 0 || R_local <= 0 || r_begin < 0 || r_begin + R_local > R_global || !labels)
        return fail(h, -1, "remd_set_replicas: bad arguments");
    const int N = h->sys.N;
    h->R_global = R_global; h->r_begin = r_begin; h->R = R_local;
    h->noise_ids.clear();                   // (ids belong to one set of replicas)
    h->reps.assign(R_local, Replica());
    for (int r = 0; r < R_local; ++r) {
        Replica& rep = h->reps[r];
        if (x) rep.x.assign(x + (size_t)r * 3 * N, x + (size_t)(r + 1) * 3 * N);
        else { rep.x.resize(3 * (size_t)N); for (int i = 0; i < N; ++i) { rep.x[3 * i] = 0.3 * (i % 64); rep.x[3 * i + 1] = 0.3 * ((i / 64) % 64); rep.x[3 * i + 2] = 0.3 * (i / 4096); } }   // coordinates follow: remd_copy_replicas
        if (v) rep.v.assign(v + (size_t)r * 3 * N, v + (size_t)(r + 1) * 3 * N); else rep.v.assign(3 * (size_t)N, 0.0);
        rep.f.assign(3 * (size_t)N, 0.0);
        for (int k = 0; k < 3; ++k) rep.box[k] = box ? box[3 * r + k] : 0.0;
        if (h->sys.method && !(rep.box[0] > 0 && rep.box[1] > 0 && rep.box[2] > 0)) return fail(h, -1, "remd_set_replicas: periodic system needs a box");
        if (h->sys.method) for (int k = 0; k < 3; ++k) if (rep.box[k] < 2.0 * std::max(h->sys.rc, h->sys.rcc)) return fail(h, -1, "remd_set_replicas: box smaller than twice the cutoff");
    }
    h->ukl.assign((size_t)R_global * std::max(0, h->K), 0.0);
    h->potential.assign(R_local, 0.0);
    h->labels.clear();
    return remd_set_labels(h, labels);
}

int remd_copy_replicas(remd_handle dst, const int32_t* dst_slot, remd_handle src, const int32_t* src_slot, int32_t n, int32_t what)
{
    if (!dst || !src || !dst->has_system || !src->has_system || dst->R <= 0 || src->R <= 0)
        return fail(dst, -1, "remd_copy_replicas: both handles need a system and replicas (remd_set_replicas)");
    if (n < 0 || (n > 0 && (!dst_slot || !src_slot)) || (what & ~7) || !(what & 7)) return fail(dst, -1, "remd_copy_replicas: bad arguments");
    if (dst->sys.N != src->sys.N) return fail(dst, -1, "remd_copy_replicas: the handles hold different particle counts");
    if (dst == src) return fail(dst, -1, "remd_copy_replicas: source and destination are the same handle");
    std::vector<char> seen(dst->R, 0);
    for (int k = 0; k < n; ++k) {
        if (dst_slot[k] < 0 || dst_slot[k] >= dst->R || src_slot[k] < 0 || src_slot[k] >= src->R) return fail(dst, -1, "remd_copy_replicas: slot out of range");
        if (seen[dst_slot[k]]++) return fail(dst, -1, "remd_copy_replicas: a destination slot is named twice");
    }
    for (int k = 0; k < n; ++k) {
        Replica& d = dst->reps[dst_slot[k]];
        const Replica& s = src->reps[src_slot[k]];
        if (what & 1) d.x = s.x;
        if (what & 2) d.v = s.v;
        if (what & 4) {
            for (int q = 0; q < 3; ++q) d.box[q] = s.box[q];
            if (dst->sys.method) for (int q = 0; q < 3; ++q) if (d.box[q] < 2.0 * std::max(dst->sys.rc, dst->sys.rcc)) return fail(dst, -1, "remd_copy_replicas: box smaller than twice the cutoff");
        }
        if (what & 5) { d.f_valid = false; d.list_valid = false; }
    }
    return 0;
}

int remd_propagate(remd_handle h, int64_t iteration, int32_t* nan_flags)
{
    if (!h || !h->has_system || !h->has_integrator || h->R <= 0 || h->K <= 0)
        return fail(h, -1, "remd_propagate: system/states/integrator/replicas not all set");
    {
        // a multiple-time-step program must name the force group of every force class (libremd_hip.so returns the same -3: a class
        // in a group that no V of the splitting names would silently never act)
        bool mts = false; unsigned named = 0u;
        for (int g = 0; g < 4; ++g) if (h->nVg[g] > 0) { mts = true; for (int c = 0; c < 6; ++c) if (h->force_groups[c] == g) named |= 1u << c; }
        if (mts)
            for (int c = 0; c < 6; ++c)
                if (h->force_groups[c] > 3 || !(named & (1u << c)))
                    return fail(h, -3, "multiple-time-step splitting: a force class sits in a force group that no V of the splitting names "
                                       "(its forces would never act); groups 0-3 are supported");
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int R = h->R;
    std::vector<int> flags(R, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < R; ++r) {
        Replica& rep = h->reps[r];
        const std::vector<double> x0 = rep.x, v0 = rep.v;
        const double box0[3] = {rep.box[0], rep.box[1], rep.box[2]};
        const double heat0 = rep.heat, shadow0 = rep.shadow; const long long trials0 = rep.n_trials, rejected0 = rep.n_rejected;
        for (int a = 0; a <= h->n_restart_attempts; ++a) {             // mcmc.py:706-759
            const int64_t it = iteration + ((int64_t)a << 40);
            if (a > 0) {
                rep.x = x0; rep.v = v0; for (int q = 0; q < 3; ++q) rep.box[q] = box0[q]; rep.f_valid = false; rep.list_valid = false;
                // (what a discarded attempt accumulated for remd_get_work goes with it)
                rep.heat = heat0; rep.shadow = shadow0; rep.n_trials = trials0; rep.n_rejected = rejected0;
            }
            if (h->reassign) assign_velocities(h, r, it);
            run_steps(h, r, h->tokens, h->nV, h->nR, h->nO, it, 0, h->n_steps, true);
            flags[r] = finite_state(rep) ? 0 : 1;
            if (!flags[r]) break;
        }
    }
    if (h->baro_frequency > 0) {
        h->baro_attempts += (h->baro_steps + h->n_steps) / h->baro_frequency - h->baro_steps / h->baro_frequency;
        h->baro_steps += h->n_steps;
    }
    h->t_prop = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (nan_flags) for (int r = 0; r < R; ++r) nan_flags[r] = flags[r];
    return 0;
}

// phases are a device matter (which kernels share the chip); the CPU port takes the request and runs one block
int remd_set_phases(remd_handle h, int32_t n) { return (!h || n < 0 || n > 2) ? -1 : 0; }
int remd_get_phases(remd_handle h, int32_t* n) { if (!h || !n) return -1; *n = 1; return 0; }

// the CPU port's SHAKE / RATTLE iterate in f64 to the tolerance itself; it keeps no statistics
int remd_get_constraint_stats(remd_handle h, int32_t* max_newton_iterations, int32_t* unconverged)
{
    if (!h) return -1;
    if (max_newton_iterations) *max_newton_iterations = 0;
    if (unconverged) *unconverged = 0;
    return 0;
}

// the device library interleaves the handles' MD steps on one GPU; here the handles are simply propagated one after the other (same results)
int remd_propagate_many(remd_handle* hs, int32_t n, int64_t iteration, int32_t* nan_flags)
{
    if (!hs || n < 1) return -1;
    int off = 0;
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return -1;
        const int rc = remd_propagate(hs[i], iteration, nan_flags ? nan_flags + off : nullptr);
        if (rc) return rc;
        off += hs[i]->R;
    }
    return 0;
}

int remd_step(remd_handle h, const char* splitting, int64_t iteration, int64_t first_step, int n_steps)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return fail(h, -1, "remd_step: not set up");
    std::vector<char> tokens;
    for (const char* c = splitting ? splitting : ""; *c; ++c) {
        if (*c == ' ') continue;
        const char t = (char)toupper(*c);
        if (t != 'V' && t != 'R' && t != 'O' && t != '{' && t != '}') return fail(h, -3, "remd_step: token must be V, R, O, { or }");
        tokens.push_back(t);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < h->R; ++r) run_steps(h, r, tokens, h->nV, h->nR, h->nO, iteration, first_step, n_steps);
    return 0;
}

int remd_ukl_device_ptr(remd_handle h, double** d_ukl)
{
    if (!h || !d_ukl || h->ukl.empty()) return fail(h, -1, "remd_ukl_device_ptr: states/replicas not set");
    *d_ukl = h->ukl.data();
    return 0;
}

int remd_compute_energies(remd_handle h, double* d_rows, double* ukl_host, double* potential_host)
{
    if (!h || !h->has_system || h->R <= 0 || h->K <= 0) return fail(h, -1, "remd_compute_energies: not set up");
    const auto t0 = std::chrono::steady_clock::now();
    if (h->ukl.size() != (size_t)h->R_global * h->K) h->ukl.assign((size_t)h->R_global * h->K, 0.0);
    double* rows = d_rows ? d_rows : h->ukl.data() + (size_t)h->r_begin * h->K;
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < h->R; ++r) h->potential[r] = ukl_row(h, r, rows + (size_t)r * h->K);
    h->t_energy = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ukl_host) memcpy(ukl_host, rows, sizeof(double) * (size_t)h->R * h->K);
    if (potential_host) memcpy(potential_host, h->potential.data(), sizeof(double) * h->R);
    return 0;
}

static int mix_impl(remd_ctx* h, int scheme, int64_t iteration, int R, int K, const double* u, int ld, int64_t* labels,
                    int64_t* nacc, int64_t* nprop, const double* logw, double* logP, int64_t n_attempts)
{
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> uc((size_t)R * K);
    for (int r = 0; r < R; ++r) memcpy(&uc[(size_t)r * K], u + (size_t)r * ld, sizeof(double) * K);
    std::vector<int64_t> a((size_t)K * K, 0), p((size_t)K * K, 0);
    std::vector<double> lp;
    switch (scheme) {
        case REMD_MIX_NONE: break;
        case REMD_MIX_SWAP_ALL: oracle_mix_swap_all(h->seed, iteration, R, K, uc.data(), labels, a.data(), p.data(), n_attempts); break;
        case REMD_MIX_SWAP_NEIGHBORS: oracle_mix_swap_neighbors(h->seed, iteration, R, K, uc.data(), labels, a.data(), p.data()); break;
        case REMD_MIX_SAMS_GLOBAL:
            if (!logw) return fail(h, -1, "remd_mix: SAMS needs log_weights");
            lp.resize((size_t)R * K);
            oracle_sams_global_jump(h->seed, iteration, R, K, uc.data(), logw, labels, a.data(), p.data(), lp.data());
            if (logP) memcpy(logP, lp.data(), sizeof(double) * lp.size());
            break;
        default: return fail(h, -1, "remd_mix: unknown scheme");
    }
    if (nacc) memcpy(nacc, a.data(), sizeof(int64_t) * a.size());
    if (nprop) memcpy(nprop, p.data(), sizeof(int64_t) * p.size());
    h->t_mix = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int remd_mix(remd_handle h, int scheme, int64_t iteration, int R, int K, const double* d_ukl, int ld, int64_t* labels,
             int64_t* nacc, int64_t* nprop, const double* logw, double* logP)
{
    if (!h || !labels || R <= 0 || K <= 0) return fail(h, -1, "remd_mix: bad arguments");
    if (!d_ukl) {
        if (h->ukl.empty() || R != h->R_global || K > h->K) return fail(h, -1, "remd_mix: handle has no matching u_kl matrix");
        d_ukl = h->ukl.data(); ld = h->K;
    }
    if (ld <= 0) ld = K;
    int rc = mix_impl(h, scheme, iteration, R, K, d_ukl, ld, labels, nacc, nprop, logw, logP, -1);
    if (!rc && R == h->R_global && !h->labels.empty()) rc = remd_set_labels(h, labels);
    return rc;
}

int remd_mix_host(remd_handle h, int scheme, int64_t iteration, int R, int K, const double* ukl, int64_t* labels, int64_t* nacc,
                  int64_t* nprop, const double* logw, double* logP, int64_t n_attempts)
{
    if (!h || !labels || !ukl || R <= 0 || K <= 0) return fail(h, -1, "remd_mix_host: bad arguments");
    return mix_impl(h, scheme, iteration, R, K, ukl, K, labels, nacc, nprop, logw, logP, n_attempts);
}

int remd_get_replicas(remd_handle h, double* x, double* v, double* potential, double* kinetic)
{
    if (!h || h->R <= 0) return fail(h, -1, "remd_get_replicas: no replicas");
    const int N = h->sys.N;
    for (int r = 0; r < h->R; ++r) {
        if (x) memcpy(x + (size_t)r * 3 * N, h->reps[r].x.data(), sizeof(double) * 3 * N);
        if (v) memcpy(v + (size_t)r * 3 * N, h->reps[r].v.data(), sizeof(double) * 3 * N);
        if (kinetic) { double ke = 0; for (int i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) ke += 0.5 * h->sys.mass[i] * h->reps[r].v[3 * i + k] * h->reps[r].v[3 * i + k]; kinetic[r] = ke; }
    }
    if (potential) {
        if (h->K <= 0) return fail(h, -1, "remd_get_replicas: states not set");
#pragma omp parallel for schedule(dynamic, 1)
        for (int r = 0; r < h->R; ++r) {
            const int64_t k = h->labels[h->r_begin + r];
            potential[r] = evaluate(h->sys, h->reps[r], h->lam_s[k], h->lam_e[k], nullptr, thread_fft(h), PART_ALL, 63, (int)k).total();
        }
    }
    return 0;
}

int remd_get_forces(remd_handle h, double* f)
{
    if (!h || h->R <= 0 || !f || h->K <= 0) return fail(h, -1, "remd_get_forces: bad arguments");
    const int N = h->sys.N;
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < h->R; ++r) { h->reps[r].f_valid = false; ensure_forces(h, r); memcpy(f + (size_t)r * 3 * N, h->reps[r].f.data(), sizeof(double) * 3 * N); }
    return 0;
}

int remd_get_energy_components(remd_handle h, double* out)
{
    if (!h || !out || h->R <= 0 || h->K <= 0) return fail(h, -1, "remd_get_energy_components: bad arguments");
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < h->R; ++r) {
        const int64_t k = h->labels[h->r_begin + r];
        const Energy E = evaluate(h->sys, h->reps[r], h->lam_s[k], h->lam_e[k], nullptr, thread_fft(h), PART_ALL, 63, (int)k);
        for (int c = 0; c < 9; ++c) out[9 * r + c] = E.c[c];
    }
    return 0;
}

int remd_sync(remd_handle h) { return h ? 0 : -1; }

int remd_test_fft3d(remd_handle h, int nx, int ny, int nz, float* data, int inverse)
{
    if (!h || !data || nx <= 0 || ny <= 0 || nz <= 0) return fail(h, -1, "remd_test_fft3d: bad arguments");
    const int n[3] = {nx, ny, nz};
    std::vector<cplx> a((size_t)nx * ny * nz);
    for (size_t i = 0; i < a.size(); ++i) a[i] = cplx(data[2 * i], data[2 * i + 1]);
    const size_t stride[3] = {(size_t)ny * nz, (size_t)nz, 1};
    for (int dim = 0; dim < 3; ++dim) {
        FFT1D fft(n[dim]);
        std::vector<cplx> lin(n[dim]), lout(n[dim]);
        const int d1 = (dim + 1) % 3, d2 = (dim + 2) % 3;
        for (int p = 0; p < n[d1]; ++p) for (int q = 0; q < n[d2]; ++q) {
            const size_t base = p * stride[d1] + q * stride[d2];
            for (int k = 0; k < n[dim]; ++k) lin[k] = a[base + k * stride[dim]];
            fft.transform(lin.data(), lout.data(), inverse != 0);
            for (int k = 0; k < n[dim]; ++k) a[base + k * stride[dim]] = lout[k];
        }
    }
    for (size_t i = 0; i < a.size(); ++i) { data[2 * i] = (float)a[i].real(); data[2 * i + 1] = (float)a[i].imag(); }
    return 0;
}

/* same contract as the matrix-core XY pass of the device library (csrc/dft_mfma.hip), as plain f64 FFTs */

int remd_last_timing(remd_handle h, double* p, double* e, double* m)
{
    if (!h) return -1;
    if (p) *p = h->t_prop; if (e) *e = h->t_energy; if (m) *m = h->t_mix;
    return 0;
}

int remd_profile_enable(remd_handle h, int on) { (void)on; return h ? 0 : -1; }
int remd_profile_filter(remd_handle h, const char* c) { (void)c; return h ? 0 : -1; }
int remd_profile_reset(remd_handle h) { return h ? 0 : -1; }
int remd_profile_get(remd_handle h, const char* name, int64_t* n, double* ms) { (void)name; if (!h) return -1; if (n) *n = 0; if (ms) *ms = 0; return 0; }

int remd_roof_microbench(remd_handle h, double* a, double* b, double* c)
{
    (void)a; (void)b; (void)c;
    return fail(h, -3, "libremd_cpu: the roof microbenchmarks are GPU measurements (not implemented on the CPU)");
}
int remd_roof_clock_ghz(remd_handle h, double*) { return fail(h, -3, "remd_roof_clock_ghz: GPU-only microbenchmark"); }
int remd_roof_pair_step(remd_handle h, int, int, double, double*, double*) { return fail(h, -3, "remd_roof_pair_step: GPU-only microbenchmark"); }

/* CPU-library extension used by bench.py's cpu_baseline leg: threads OpenMP will use over replicas */
int remd_cpu_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"
