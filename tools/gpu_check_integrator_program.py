"""GPU: the HIP integrator chain against the reference's own step program (tests/golden/integrator_program_reference.json, see
tests/test_integrator_program.py) at fp32 tolerances -- written at the end of round 5 when no GPU time was left to try it, hence a
tool and not yet a test: run it first, then move the call into tests/ with the tolerances it needs.
usage (GPU box, repo root): python tools/gpu_check_integrator_program.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_integrator_program as T
from openmmtools_amd._engine import HipEngine
for k in T.CASES:
    c = T.G['cases'][k]
    try:
        worst = T.check_engine(lambda: HipEngine(), k, 2e-6, 2e-4, 2e-3)
        print('%-24s dt %.4f  ok   max |dx| %.2e nm  max |dv| %.2e nm/ps' % (c['splitting'], c['timestep'], worst[0], worst[1]))
    except AssertionError as exc:
        print('%-24s dt %.4f  FAILED %s' % (c['splitting'], c['timestep'], str(exc)[:200]))
