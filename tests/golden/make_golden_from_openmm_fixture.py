"""Golden vectors PRODUCED BY OPENMM for the floating-point half of the path: the reference ships a store written by a
real run of its sampler on the headline system,

    /root/reference/openmmtools/data/reporter-examples/alanine_dipeptide_legacy.nc             (analysis file)
    /root/reference/openmmtools/data/reporter-examples/alanine_dipeptide_legacy_checkpoint.nc  (checkpoint file)

used by the reference's own test tests/test_sampling.py:2943-2990 (resume from a legacy store).  It holds, written by
OpenMM 7.7 through openmmtools' reporter (multistatereporter.py:612-668, 865-930, 1654-1737):

    /thermodynamic_states/state0    YAML of ThermodynamicState.__getstate__ (states.py:1257-1280); 'standard_system' is
                                    zlib(XmlSerializer.serialize(System)) of testsystems.AlanineDipeptideExplicit
                                    (2269 particles, PME, cutoff 1.0 nm, switch 0.85 nm, ewaldTolerance 1e-5, 2259
                                    constraints, 2345 exceptions, CMMotionRemover + the AndersenThermostat the standard
                                    system carries as its thermostat marker, states.py:1447-1490)
    /thermodynamic_states/state1..19   temperature only (+ '_Reporter__compatible_state')
    /energies  f8[3, 1, 20]         reduced potentials u_kl of the ONE replica at the 20 temperatures, iterations 0..2
                                    (states.py:1908-1917: u = beta * U, NVT)
    /states i4[3,1], /accepted, /proposed i4[3,20,20], /neighborhoods i1[3,1,20], /mcmc_moves/move0..19, /options
    checkpoint: /positions f4[3,1,2269,3] nm, /box_vectors f4[3,1,3,3] nm, /volumes f8[3,1]

The files are HDF5 (netCDF4); h5py / netCDF4 are not in this image, so they are read through ctypes on the system's
libhdf5 (openmmtools_amd/multistate/_hdf5.py).  Output: tests/golden/openmm_alanine_fixture.npz (committed, ~110 KB;
the System XML stays zlib-compressed exactly as stored).  /root/reference does not exist on the GPU box: the tests read
only the npz.      usage:  python tests/golden/make_golden_from_openmm_fixture.py
"""
import os
import sys
import zlib

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openmmtools_amd.multistate import _hdf5

REF = '/root/reference/openmmtools/data/reporter-examples'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'openmm_alanine_fixture.npz')


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_constructor('!Quantity', lambda loader, node: loader.construct_mapping(node))


def main():
    ana = _hdf5.File(os.path.join(REF, 'alanine_dipeptide_legacy.nc'))
    chk = _hdf5.File(os.path.join(REF, 'alanine_dipeptide_legacy_checkpoint.nc'))
    n_states = len(ana.keys('/thermodynamic_states')[1])
    temperatures, xml_zlib = [], None
    for k in range(n_states):
        doc = yaml.load(ana.read('/thermodynamic_states/state%d' % k).tobytes().decode(), Loader=_Loader)
        assert doc['_serialized__class_name'] == 'ThermodynamicState' and doc['pressure'] is None
        assert doc['temperature']['unit'] == 'kelvin'
        temperatures.append(float(doc['temperature']['value']))
        if k == 0:
            xml_zlib = doc['standard_system']
        else:
            assert doc['_Reporter__compatible_state'] == 'thermodynamic_states/0'
    xml = zlib.decompress(xml_zlib).decode()
    assert xml.lstrip().startswith('<?xml') and 'openmmVersion="7.7"' in xml
    moves = [str(ana.read('/mcmc_moves/move%d' % k)[0]) for k in range(n_states)]
    out = dict(
        system_xml_zlib=np.frombuffer(xml_zlib, dtype=np.uint8),
        temperatures=np.array(temperatures),
        energies=ana.read('/energies')[:, 0, :].astype(np.float64),                  # [iteration][state]
        states=ana.read('/states')[:, 0].astype(np.int32),
        neighborhoods=ana.read('/neighborhoods')[:, 0, :].astype(np.int8),
        accepted=ana.read('/accepted').astype(np.int32), proposed=ana.read('/proposed').astype(np.int32),
        last_iteration=ana.read('/last_iteration').astype(np.int64),
        positions=chk.read('/positions')[:, 0].astype(np.float32),                   # [iteration][atom][3] nm
        box_vectors=chk.read('/box_vectors')[:, 0].astype(np.float32),               # [iteration][3][3] nm
        volumes=chk.read('/volumes')[:, 0].astype(np.float64),
        mcmc_move0=np.array(moves[0]), options=np.array(str(ana.read('/options')[0])),
        title=np.array(ana.attr('title')), convention_version=np.array(ana.attr('ConventionVersion')),
    )
    assert all(m == moves[0] for m in moves)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', out['positions'].shape, out['energies'].shape,
          'T = %.1f .. %.1f K' % (temperatures[0], temperatures[-1]))


if __name__ == '__main__':
    main()
