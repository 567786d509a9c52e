"""CPU-only: how the Ewald split travels from the engine to the descriptor (round 4).

The device engine asks the host classes for a longer Coulomb range and the smaller mesh the reference's own tolerance rule then
gives (system.rebalanced_coulomb_cutoff; alchemy.py:1528-1532 quotes the rule); another build of the ABI (the CPU baseline) and the
oracle engine keep OpenMM's split.  Parity of the rebalanced split against OpenMM's numbers: tests/test_openmm_fixture.py."""
import math
import os

import numpy as np
import pytest

import oracle
from openmmtools_amd import _engine, mcmc, states, testsystems as ts, unit
from openmmtools_amd.multistate import ParallelTemperingSampler
from openmmtools_amd.system import ewald_parameters, rebalanced_coulomb_cutoff, system_to_desc
from oracle_engine import OracleEngine

CPU_LIB = os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), '_build', 'libremd_cpu.so')


def test_rule_is_openmms_applied_to_the_coulomb_range():
    box = [3.2852863, 3.2861648, 3.1855098]
    alpha, grid = ewald_parameters(1.0, 1e-5, box)
    assert alpha == pytest.approx(math.sqrt(-math.log(2e-5)), rel=1e-15) and grid == [75, 75, 72]
    rcc = rebalanced_coulomb_cutoff(1.0, 1e-5, box)
    a2, g2 = ewald_parameters(rcc, 1e-5, box)
    assert g2 == [64, 64, 64] and a2 * rcc == pytest.approx(alpha * 1.0, rel=1e-14)        # the same erfc(alpha r) at the range's end
    assert 1.0 < rcc <= 1.25 and 2.0 * rcc < min(box)
    # nothing to gain: the reference mesh is already a plane-friendly size, or the box is too small for a longer range
    assert rebalanced_coulomb_cutoff(1.0, 1e-5, [2.9, 2.9, 2.9]) == 1.0 or max(ewald_parameters(rebalanced_coulomb_cutoff(1.0, 1e-5, [2.9, 2.9, 2.9]), 1e-5, [2.9] * 3)[1]) < 64
    assert rebalanced_coulomb_cutoff(1.0, 1e-5, [2.1, 2.1, 2.1]) == 1.0


@pytest.mark.parametrize('name, mesh', [('HostGuestExplicit', 80), ('DHFRExplicit', 128)])
def test_auto_split_of_the_other_pme_systems(name, mesh):
    system = getattr(ts, name)().system
    ref, auto = system_to_desc(system), system_to_desc(system, ewald_split='auto')
    assert 'coulomb_cutoff' not in ref and max(auto['pme_grid']) == mesh < max(ref['pme_grid'])
    assert auto['cutoff'] == ref['cutoff'] == 1.0 and 1.0 < auto['coulomb_cutoff'] < 1.1
    with pytest.raises(ValueError):
        system_to_desc(system, ewald_split=0.9)


def test_which_engine_asks_for_which_split():
    assert _engine.DEFAULT_EWALD_SPLIT == os.environ.get('REMD_EWALD_SPLIT', 'auto')
    if not os.path.exists(CPU_LIB):
        oracle.build()
    cpu = _engine.HipEngine(lib_path=CPU_LIB)                      # another build of the ABI: the reference's own split
    try:
        assert cpu.ewald_split == 'reference'
        assert _engine.HipEngine(lib_path=CPU_LIB, ewald_split=1.2).ewald_split == 1.2
    finally:
        cpu.close()


def test_sampler_hands_the_engines_split_to_the_descriptor():
    class Capture(OracleEngine):
        ewald_split = 'auto'

        def set_system(self, desc):
            self.seen = desc
            raise StopIteration                                    # the descriptor is all this test needs

    al = ts.AlanineDipeptideExplicit()
    move = mcmc.LangevinSplittingDynamicsMove(timestep=2.0 * unit.femtosecond, n_steps=1, splitting='V R O R V')
    eng = Capture()
    s = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=1, engine=eng, seed=1)
    with pytest.raises(StopIteration):
        s.create(states.ThermodynamicState(al.system, 300.0 * unit.kelvin),
                 [states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())], storage=None,
                 min_temperature=300.0 * unit.kelvin, max_temperature=400.0 * unit.kelvin, n_temperatures=2)
    assert list(eng.seen['pme_grid']) == [64, 64, 64] and eng.seen['coulomb_cutoff'] == pytest.approx(1.126, abs=1e-3)
    class Plain(Capture):                                          # no preference: OpenMM's split
        ewald_split = None
    eng2 = Plain()
    s2 = ParallelTemperingSampler(mcmc_moves=move, number_of_iterations=1, engine=eng2, seed=1)
    with pytest.raises(StopIteration):
        s2.create(states.ThermodynamicState(al.system, 300.0 * unit.kelvin),
                  [states.SamplerState(al.positions, box_vectors=al.system.getDefaultPeriodicBoxVectors())], storage=None,
                  min_temperature=300.0 * unit.kelvin, max_temperature=400.0 * unit.kelvin, n_temperatures=2)
    assert list(eng2.seen['pme_grid']) == [75, 75, 72] and 'coulomb_cutoff' not in eng2.seen


def test_cpu_library_refuses_a_coulomb_range_inside_the_cutoff():
    if not os.path.exists(CPU_LIB):
        oracle.build()
    al = ts.AlanineDipeptideExplicit()
    d = system_to_desc(al.system, ewald_split='auto')
    d['coulomb_cutoff'] = 0.9
    eng = _engine.HipEngine(lib_path=CPU_LIB)
    try:
        with pytest.raises(RuntimeError, match='shorter than the NonbondedForce cutoff'):
            eng.set_system(d)
    finally:
        eng.close()
