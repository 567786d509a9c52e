"""The store the reference ships for its own resume test (openmmtools/tests/test_sampling.py:2943-2990), copied byte for byte as
the input fixture of the reader of the reference's netCDF4 layout (openmmtools_amd/multistate/_reference_store.py):

    /root/reference/openmmtools/data/reporter-examples/alanine_dipeptide_legacy.nc             (197 KB, analysis file)
    /root/reference/openmmtools/data/reporter-examples/alanine_dipeptide_legacy_checkpoint.nc  ( 94 KB, checkpoint file)

They are DATA written by OpenMM 7.7 + yank.multistate in 2022 (one replica of AlanineDipeptideExplicit, 20 temperatures, 3
iterations, checkpoint interval 1, no velocities: the pre-0.21.3 layout), not source code.  /root/reference does not exist on the
GPU box, so the tests read the copies under tests/golden/reference_store/.     usage: python tests/golden/make_golden_reference_store.py
"""
import hashlib
import os
import shutil

SRC = '/root/reference/openmmtools/data/reporter-examples'
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_store')

if __name__ == '__main__':
    os.makedirs(DST, exist_ok=True)
    for name in ('alanine_dipeptide_legacy.nc', 'alanine_dipeptide_legacy_checkpoint.nc'):
        shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, name))
        with open(os.path.join(DST, name), 'rb') as fh:
            print(name, os.path.getsize(os.path.join(DST, name)), hashlib.sha256(fh.read()).hexdigest()[:16])
