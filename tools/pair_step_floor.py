"""Issue floor of the pair kernel's cluster-pair step: remd_roof_pair_step at 1 ... 8 wavefronts per SIMD with one
and two steps in flight, in cycles per step per SIMD at the clock the FMA roof kernel measures, next to the kernel's own figure
(stand-alone time x SIMDs x clock / cluster-pair steps, steps from SQ_INSTS_LDS).  usage: python tools/pair_step_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmmtools_amd._engine import HipEngine
eng = HipEngine()
roofs = eng.roof_microbench()
ghz = roofs.get('shader_clock_ghz_under_fma_load', 2.1)
print('clock under FMA load %.3f GHz; plain v_fma_f32 roof %.1f TFLOP/s' % (ghz, roofs['fma_f32_tflop_per_s']))
print('waves/SIMD  chains  cycles/step/SIMD   (27 VALU + 4 extra issue slots of the packed ops = 62 cycles of pure issue at 2 cycles / wave64 op)')
for chains in (1, 2):
    for w in (1, 2, 3, 4, 6, 8):
        c, us = eng.roof_pair_step(w, chains, ghz)
        print('%6d %8d %12.1f     launch %.1f us' % (w, chains, c, us))
eng.close()
