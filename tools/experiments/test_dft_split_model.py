"""CPU model of the matrix-core XY pass's arithmetic (csrc/dft_mfma.hip): operands split into two f16 (truncated hi +
residual lo, real f16 incl. subnormals), hi*hi + lo*hi + hi*lo in one f32 accumulator, plane scaled to 2^13 at load and by
2^-5 after a pass.  Pins the claim the kernel's design rests on: the whole forward - influence - inverse pipeline stays
within 1e-6 of the plane's largest element (an f32 FFT: 2.4e-7).  The GPU tests check the kernel against numpy directly."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import dft_split_precision as model


def test_split_f16_pipeline_is_f32_accurate():
    for seed in (0, 1):
        (emax, erms), (fmax, frms), info = model.run_model(seed)
        assert emax < 1e-6 and erms < 1e-6, (emax, erms)
        assert fmax < 1e-6
        assert float(info[0]) > 1.0          # the last pass's input still sits far above the f16 subnormal floor
