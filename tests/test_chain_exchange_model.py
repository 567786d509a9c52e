"""The protocol of the integrator chain's momentum exchange ('M' token, openmmtools_amd/csrc/integrate.hip) as a CPU model under
arbitrary schedules and store delays (tools/experiments/chain_exchange_model.py): every workgroup obtains the true sum of every epoch;
the parity halves and the 16-bit tag never let a stale word pass, also when the tag wraps.  The device code itself is held by the GPU
tests (merged against split momentum sum bit for bit: tests/test_forcefield_parity.py)."""
import importlib.util
import os
import random

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('chain_exchange_model', os.path.join(HERE, '..', 'tools', 'experiments', 'chain_exchange_model.py'))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def test_pack_unpack_round_trip():
    rng = random.Random(1)
    for _ in range(1000):
        v = rng.randrange(-(1 << 46), 1 << 46)
        e = rng.randrange(1, 1 << 20)
        w = model.pack(v, e)
        assert 0 <= w < (1 << 64) and model.unpack(w) == v and (w & 0xffff) == (e & 0xffff)


@pytest.mark.parametrize('G', [1, 2, 3, 32])
def test_every_workgroup_sees_the_true_sum(G):
    for seed in range(25):
        model.run(G, 30, seed=seed)


def test_tag_wrap_and_slow_stores():
    for seed in range(25):
        model.run(3, 14, seed=seed, first_epoch=65529)          # epochs 65529 ... 65542: the tag passes through 0
        model.run(5, 20, seed=seed, max_delay=200)               # stores that stay invisible for a long time


def test_a_single_buffer_would_fail():
    """the model is able to see the failure it is there to exclude: without the parity halves a fast workgroup overwrites its word
    while a slow one has not read it yet, and the slow one never finds the tag it waits for"""
    src = open(os.path.join(HERE, '..', 'tools', 'experiments', 'chain_exchange_model.py')).read()
    src = src.replace('par = epoch & 1', 'par = 0').replace("assert step < 10_000_000, 'no progress: deadlock'", "assert step < 200_000, 'no progress: deadlock'")
    ns = {}
    exec(compile(src.split("if __name__ == '__main__':")[0], 'one_buffer_model', 'exec'), ns)
    failures = 0
    for seed in range(10):
        try:
            ns['run'](3, 30, seed=seed)
        except AssertionError:
            failures += 1
    assert failures > 0
