"""GPU parity: the HIP mixing kernels (csrc/mix.hip) against the sequential C oracle, through the
C ABI (remd_mix_host / remd_mix).  Bar: bit-exact labels and both count matrices."""
import numpy as np
import pytest
import oracle

pytestmark = pytest.mark.gpu
SEED = 0xC0FFEE


@pytest.fixture(params=['speculative', 'speculative-one-kernel', 'dataflow', 'by-acceptance'], autouse=True)
def swap_all_kernel(request, monkeypatch):
    """swap-all has two kernels with the same (sequential) result: speculative windows and the dataflow over per-slot
    tickets; a handle picks by the acceptance of its previous call, REMD_MIX_FLOW pins one (mix.hip: remd_mix_launch).
    The speculative one runs with its label-independent part hoisted into a whole-chip kernel (round 4, the default) or
    entirely inside the serial workgroup (REMD_MIX_PRE=0)."""
    monkeypatch.delenv('REMD_MIX_PRE', raising=False)
    if request.param == 'by-acceptance':
        monkeypatch.delenv('REMD_MIX_FLOW', raising=False)
    else:
        monkeypatch.setenv('REMD_MIX_FLOW', '1' if request.param == 'dataflow' else '0')
        if request.param == 'speculative-one-kernel':
            monkeypatch.setenv('REMD_MIX_PRE', '0')
    return request.param


def _ukl(R, K, scale, rng):
    # PT-like structure (outer product) plus noise so that acceptance is neither 0 nor 1
    return np.outer(rng.normal(scale=scale, size=R), np.linspace(0.5, 1.5, K)) + rng.normal(scale=0.5, size=(R, K))


@pytest.mark.parametrize('R', [2, 4, 16, 24, 64, 128])
def test_swap_all_bit_exact(hip_engine_factory, R):
    eng = hip_engine_factory()
    eng.seed(SEED)
    rng = np.random.default_rng(R)
    u = _ukl(R, R, 3.0, rng)
    labels = rng.permutation(R).astype(np.int64)
    n_att = -1 if R <= 64 else 200000       # the C oracle handles 2M attempts fine, keep the test quick
    for it in (0, 1, 5):
        got = eng.mix_host('swap-all', it, u, labels, n_attempts=n_att)
        ref = oracle.mix('swap-all', SEED, it, u, labels, n_attempts=n_att)
        assert np.array_equal(got[0], ref[0])
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        assert got[2].sum() == 2 * (R ** 3 if n_att < 0 else n_att)
        labels = got[0]


def test_swap_all_full_r128(hip_engine_factory):
    """BASELINE config 5 size: 128^3 = 2,097,152 attempts."""
    eng = hip_engine_factory()
    eng.seed(SEED)
    rng = np.random.default_rng(5)
    u = _ukl(128, 128, 2.0, rng)
    labels = np.arange(128, dtype=np.int64)
    got = eng.mix_host('swap-all', 9, u, labels)
    ref = oracle.mix('swap-all', SEED, 9, u, labels)
    for a, b in zip(got[:3], ref[:3]):
        assert np.array_equal(a, b)


def test_swap_all_r192_ukl_in_global_memory(hip_engine_factory):
    """8 GPUs x 24 replicas (bench.py weak scaling): u_kl (295 KB) no longer fits in LDS; PT-like low acceptance."""
    eng = hip_engine_factory()
    eng.seed(SEED)
    R = 192
    rng = np.random.default_rng(6)
    kT = 0.0083145 * np.geomspace(300.0, 600.0, R)
    u = np.outer(-30000.0 + 0.5 * kT * 4500 + rng.normal(size=R) * np.sqrt(2250.0) * kT, 1.0 / kT)
    labels = rng.permutation(R).astype(np.int64)
    got = eng.mix_host('swap-all', 2, u, labels, n_attempts=1500000)
    ref = oracle.mix('swap-all', SEED, 2, u, labels, n_attempts=1500000)
    for a, b in zip(got[:3], ref[:3]):
        assert np.array_equal(a, b)
    assert 0.01 < got[1].sum() / got[2].sum() < 0.3


@pytest.mark.parametrize('R', [3, 8, 48, 96])
def test_swap_all_speculation_stress(hip_engine_factory, R):
    """Cases that stress the speculative window: every swap accepted (u = 0, longest accepted chains), duplicate
    labels (swaps that do not change anything), and windows cut short by crowded replica slots (small R)."""
    eng = hip_engine_factory()
    eng.seed(SEED + R)
    rng = np.random.default_rng(100 + R)
    cases = [(np.zeros((R, R)), rng.permutation(R)),
             (_ukl(R, R, 1.0, rng), rng.integers(0, R, R)),                 # duplicates allowed by the kernel contract
             (_ukl(R, R, 0.2, rng), np.arange(R))]
    for u, labels in cases:
        labels = labels.astype(np.int64)
        for it in (0, 3):
            got = eng.mix_host('swap-all', it, u, labels)
            ref = oracle.mix('swap-all', SEED + R, it, u, labels)
            for a, b in zip(got[:3], ref[:3]):
                assert np.array_equal(a, b)
            labels = got[0]


def test_swap_all_edge_cases(hip_engine_factory):
    eng = hip_engine_factory()
    eng.seed(1)
    # R = 1: every attempt is (0, 0) -> accepted no-op, counts 2 per attempt on [0, 0]
    got = eng.mix_host('swap-all', 0, np.zeros((1, 1)), np.zeros(1, np.int64))
    assert got[0][0] == 0 and got[1][0, 0] == 2 and got[2][0, 0] == 2
    # zero attempts leaves labels untouched, zeroed statistics
    lab = np.array([2, 0, 1], np.int64)
    got = eng.mix_host('swap-all', 0, np.zeros((3, 3)), lab, n_attempts=0)
    assert np.array_equal(got[0], lab) and got[2].sum() == 0
    # huge energy gaps: exp underflow branch (log_p < -700) must reject identically
    u = np.diag([0.0, -1e6, -2e6, -3e6]) + 0.0
    ref = oracle.mix('swap-all', 1, 3, u, np.arange(4))
    got = eng.mix_host('swap-all', 3, u, np.arange(4))
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # NaN energies: comparisons are false on both sides -> identical (all rejected)
    u = np.full((4, 4), np.nan)
    ref = oracle.mix('swap-all', 1, 0, u, np.arange(4))
    got = eng.mix_host('swap-all', 0, u, np.arange(4))
    assert np.array_equal(got[0], ref[0]) and got[1].sum() == ref[1].sum() == 0


def test_uniform_mixing_chi_square_device(hip_engine_factory):
    """openmmtools/tests/test_mixing.py:76-92 on the device kernel."""
    import scipy.stats
    eng = hip_engine_factory()
    eng.seed(42)
    n = 16
    u = np.zeros((n, n))
    labels = np.arange(n, dtype=np.int64)
    counts = np.zeros((n, n))
    for call in range(400):
        labels = eng.mix_host('swap-all', call, u, labels)[0]
        counts[np.arange(n), labels] += 1
    for r in range(n):
        assert scipy.stats.chisquare(counts[r])[1] > 0.001 / n


@pytest.mark.parametrize('R', [2, 3, 9, 24, 128])
def test_swap_neighbors_bit_exact(hip_engine_factory, R):
    eng = hip_engine_factory()
    eng.seed(SEED)
    rng = np.random.default_rng(R + 100)
    u = _ukl(R, R, 1.0, rng)
    labels = rng.permutation(R).astype(np.int64)
    for it in range(6):
        got = eng.mix_host('swap-neighbors', it, u, labels)
        ref = oracle.mix('swap-neighbors', SEED, it, u, labels)
        for a, b in zip(got[:3], ref[:3]):
            assert np.array_equal(a, b)
        labels = got[0]


@pytest.mark.parametrize('R,K', [(1, 5), (5, 7), (16, 128), (128, 128)])
def test_sams_global_jump_parity(hip_engine_factory, R, K):
    eng = hip_engine_factory()
    eng.seed(SEED)
    rng = np.random.default_rng(R * 1000 + K)
    u = rng.normal(scale=3.0, size=(R, K))
    logw = rng.normal(size=K)
    labels = rng.integers(0, K, R)
    for it in range(4):
        got = eng.mix_host('sams-global-jump', it, u, labels, log_weights=logw)
        ref = oracle.mix('sams-global-jump', SEED, it, u, labels, log_weights=logw)
        assert np.array_equal(got[0], ref[0])                       # drawn states: bit exact
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        assert np.allclose(got[3], ref[3], rtol=0, atol=1e-12)      # log P_k: libm log differs by <= ulps
        labels = got[0]


def test_device_reproduces_the_committed_golden_vectors(hip_engine_factory):
    """tests/golden/mix_reference_arith.json: transcription of replicaexchange.py:294-349 / :382-406 (tests/golden/make_golden_mix.py)."""
    import json, os
    path = os.path.join(os.path.dirname(__file__), 'golden', 'mix_reference_arith.json')
    for c in json.load(open(path))['cases']:
        eng = hip_engine_factory()
        eng.seed(c['seed'])
        got = eng.mix_host('swap-all', c['iteration'], np.array(c['u_kl']), np.array(c['labels_in'], dtype=np.int64))
        assert got[0].tolist() == c['labels_out']
        assert got[1].tolist() == c['n_accepted'] and got[2].tolist() == c['n_proposed']

