# swap-all, speculative windows with the whole-chip preparation: the window length (REMD_MIX_PERR x R attempts, default 6) re-scanned
export TMPDIR=/tmp; O=gpurun_out/r06s3_33; mkdir -p $O
for perr in 6 4 3 2 1; do echo "== REMD_MIX_PERR=$perr"; REMD_MIX_PERR=$perr timeout 300 python tools/mix_microbench.py 24 64 128 192 2>/dev/null; done | tee $O/summary.txt
