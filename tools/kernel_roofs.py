"""Per-kernel roofline table ("report each kernel's own HBM fraction"): joins the rocprofv3 kernel-trace
summary (average duration per launch) with the PMC traffic table (FETCH_SIZE x 2 + WRITE_SIZE per launch, separate
passes) and the algorithmic bytes per launch of the headline workload (--replicas per launch x 2269 atoms, 64^3 mesh unless --mesh says otherwise).

usage: python tools/kernel_roofs.py <kernel_stats.md> <pmc_traffic.json> [--stream GBs] [--replicas per launch] > profiles/rNN_x_kernel_roofs.md"""
import json
import sys

# (round 6: a launch covers one PHASE's share of the replicas -- remd_set_phases -- : --replicas 12 for the phased headline run)
R = int(float(sys.argv[sys.argv.index("--replicas") + 1])) if "--replicas" in sys.argv else 24
N, NPAD = 2269, 2304
NX, NY, NZ = [int(v) for v in (sys.argv[sys.argv.index("--mesh") + 1].split("x") if "--mesh" in sys.argv else ("64", "64", "64"))]   # round 4: the rebalanced split
HALF = (NZ // 2 + 1) * NX * NY * R            # complex points of the half spectrum, all replicas
REAL = NX * NY * NZ * R
# algorithmic bytes per launch (what the kernel must move at least once) and the SURVEY 8(d) row they come from
ALGO = {
    'integrate_chain_kernel': (64.0 * N * R, 'fused V/R/O bound: read x, v, f, 1/m; write x, v = 64 B/atom'),
    'pme_spread_zfwd': (16.0 * N * R + 8.0 * HALF, 'read x, q per atom; write the half spectrum (8 B/point)'),
    'pme_xy_fused_kernel': (16.0 * HALF + 4.0 * HALF, 'read + write the half spectrum, read the influence table'),
    'pme_xy_pow2_kernel': (16.0 * HALF + 4.0 * HALF, 'read + write the half spectrum, read the influence table (register transforms, pme_pow2.h)'),
    'pme_unbin_forces_kernel': (2 * 24.0 * N * R + 24.0 * N * R, 'read + zero the sums by bin position, add 24 B/atom into the per-atom accumulator'),
    'pme_zinv': (8.0 * HALF + 40.0 * N * R, 'read the half spectrum; x, q in, 24 B/atom of force atomics out (the potential mesh stays in LDS)'),
    'pme_gather_kernel': (4.0 * REAL + 40.0 * N * R, 'read the potential mesh once; x, q in, 24 B/atom of force atomics out'),
    'nonbonded_sci2_kernel': (28.0 * N * R, 'x, q, sigma, eps in; f out = 28 B/atom (the kernel is FP32-VALU bound, 10 kflop/atom)'),
    'scatter_sorted_forces_kernel': (2 * 24.0 * NPAD * R, 'read the sorted accumulator, add into the per-atom accumulator'),
    'gather_positions2_kernel': (2 * 16.0 * NPAD * R + 16.0 * NPAD * R, 'x in, sorted x out (+ cluster boxes)'),
    'listed_forces_kernel': (0.0, 'latency bound (~5000 (term, atom) entries per replica)'),
    'build_sci_list2_kernel': (0.0, 'latency bound'),
    'pme_bin_kernel': (16.0 * N * R + 4.0 * N * R, 'x in, bin lists out'),
}


def main():
    stats, pmc = sys.argv[1], sys.argv[2]
    stream = float(sys.argv[sys.argv.index('--stream') + 1]) if '--stream' in sys.argv else None
    traffic = json.load(open(pmc))
    rows = []
    for line in open(stats):
        if not line.startswith('|') or line.startswith('| kernel') or line.startswith('|---'):
            continue
        c = [x.strip() for x in line.strip().strip('|').split('|')]
        name, calls, avg_us = c[0], int(c[1]), float(c[3])
        key = [k for k in ALGO if k in name]
        if not key or calls < 100:
            continue
        algo, why = ALGO[key[0]]
        # (several instantiations of a kernel share a name stem -- the energy pass's runs 4 times, the force-only one thousands:
        # the row belongs to the instantiation this line of the statistics names, i.e. the longest common prefix)
        def lcp(a, b):
            n = 0
            while n < min(len(a), len(b)) and a[n] == b[n]: n += 1
            return n
        pm = sorted([(lcp(k.replace('void ', ''), c[0].replace('void ', '')), v) for k, v in traffic.items() if key[0] in k], key=lambda e: e[0])
        pmc_mb = pm[-1][1]['hbm_mb_corrected'] if pm else float('nan')
        rows.append((name[:46], calls, avg_us, algo / 1e6, pmc_mb, algo / (avg_us * 1e-6) / 1e9, pmc_mb * 1e6 / (avg_us * 1e-6) / 1e9, why))
    print('| kernel | launches | avg us | algorithmic MB / launch | PMC HBM-side MB / launch | algorithmic GB/s | frac of 8 TB/s' +
          (' | frac of measured STREAM (%.0f GB/s)' % stream if stream else '') + ' | PMC GB/s | algorithmic bytes counted |')
    print('|---|---|---|---|---|---|---|' + ('---|' if stream else '') + '---|---|')
    for n, calls, us, amb, pmb, ag, pg, why in rows:
        print('| %s | %d | %.1f | %.2f | %.2f | %.0f | %.3f |' % (n, calls, us, amb, pmb, ag, ag / 8000.0) +
              (' %.3f |' % (ag / stream) if stream else '') + ' %.0f | %s |' % (pg, why))


if __name__ == '__main__':
    main()
