#!/bin/bash
# exclusion words in LDS for 8 < W <= 24 (DHFR, CB7:B2): parity + configs 4, 5 + headline
export TMPDIR=/tmp
O=gpurun_out/r04_z; mkdir -p $O
timeout 600 python -m pytest tests/test_forcefield_parity.py tests/test_openmm_fixture.py -m gpu -x -q > $O/pytest_excl_lds.log 2>&1; tail -3 $O/pytest_excl_lds.log
DHFR_STEPS=100 DHFR_ITERS=4 python tools/dhfr_profile.py dhfr 16 2>&1 | tail -4
python tools/bench_configs.py 4 5 > $O/bench_configs_excl_lds.jsonl 2> /dev/null; cut -c1-200 $O/bench_configs_excl_lds.jsonl
python bench.py --no-cpu-baseline > $O/bench_excl_lds.json 2> /dev/null; head -c 400 $O/bench_excl_lds.json; echo
