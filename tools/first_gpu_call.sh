#!/bin/bash
# The first GPU call a next round should make (written at the end of round 5, when the GPU budget was gone): what could not be run on the last
# tree -- the full GPU suite, smoke, the default bench line with its shapes (strong-128 alanine ran into a device poll's time-out one commit
# before the end), config 5 (DHFR x 16: hung in the last collection with the two-per-CU chain that is now opt-in), and the integrator
# chain against the reference's own step program at fp32.
# usage (repo root): gpurun --timeout 1500 -- 'bash tools/first_gpu_call.sh'
export TMPDIR=/tmp
O=gpurun_out/first; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 300 $O/bench_default.json; echo
python - <<'PY' | tee $O/shapes.txt
import json
d = json.load(open('gpurun_out/first/bench_default.json'))
print('value', d['value'], 'shapes', json.dumps(d.get('shapes'))[:600])
PY
timeout 300 python tools/gpu_check_integrator_program.py 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tee $O/integrator_program.txt
timeout 400 python tools/bench_configs.py 5 2> $O/config5.err | cut -c1-200 | tee $O/config5.txt
