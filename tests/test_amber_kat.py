"""Known-answer test of the Amber -> System conversion on the host-guest system (config 4), VERDICT r3 item 6a.

The OpenMM fixture (tests/test_openmm_fixture.py) pins the conversion on the alanine dipeptide prmtop, an OLD-style topology
without SCEE_SCALE_FACTOR / SCNB_SCALE_FACTOR sections; cb7-b2/complex-explicit.prmtop carries them per dihedral type and takes
the other branch of openmmtools_amd/amber.py (create_system: `scee[t] if scee else 1.2`).  The literals below were computed from
the raw prmtop fields by tests/golden/make_kat_cb7_exceptions.py (a parser of its own; it prints the fields and the arithmetic,
checked by hand for the first pair: q = 1.36120581 / 18.2223 = 0.074700 e, 5.18242212 / 18.2223 = 0.284400 e, product / 1.2 =
0.0177039; sigma = ((1328.0125 / 9.13231543)^(1/6) + (1043080.23 / 675.612247)^(1/6)) / 2 * 0.1 = (2.29317 + 3.39967) / 20;
epsilon = sqrt(9.13231543^2 / (4 * 1328.0125) * 675.612247^2 / (4 * 1043080.23)) * 4.184 / 2.0).
Reference semantics: OpenMM's AmberPrmtopFile.createSystem as testsystems.HostGuestExplicit calls it (testsystems.py:3826-3835)."""
import numpy as np
import pytest
from openmmtools_amd import testsystems
from openmmtools_amd.system import NonbondedForce, PeriodicTorsionForce

# (i, j): chargeProd [e^2], sigma [nm], epsilon [kJ/mol] -- dihedral types 1, 7, 2 (SCEE 1.2, SCNB 2.0 read per type)
ONE_FOUR = {
    (105, 106): (1.770390000000e-02, 2.846421404264e-01, 8.670021361863e-02),     # host: CB7 n - c pair
    (147, 148): (1.579268407898e-03, 2.649532787260e-01, 3.284440005489e-02),     # guest: two hydrogens of equal type
    (133, 137): (3.302293321187e-02, 3.233071447954e-01, 3.173899778539e-01),     # guest: heavy - heavy
}
# GAFF improper (i, j, k, l) as listed (negative third / fourth pointers in the prmtop): k 10.5 kcal/mol, n = 2, phase pi
IMPROPER = ((25, 27, 101, 116), 2, 3.141594, 10.5 * 4.184)


@pytest.fixture(scope='module')
def hostguest():
    return testsystems.HostGuestExplicit().system


def test_one_four_exceptions_of_the_per_dihedral_scale_factor_branch(hostguest):
    nb = [f for f in hostguest.getForces() if isinstance(f, NonbondedForce)][0]
    found = {}
    for e in nb.exceptions:
        key = (min(e[0], e[1]), max(e[0], e[1]))
        if key in ONE_FOUR:
            found[key] = e[2:]
    assert set(found) == set(ONE_FOUR)
    for key, (qq, sig, eps) in ONE_FOUR.items():
        got = found[key]
        assert got[0] == pytest.approx(qq, rel=2e-6), key          # (AMBER_CHARGE = 18.2223 to the digits OpenMM uses)
        assert got[1] == pytest.approx(sig, rel=1e-9), key
        assert got[2] == pytest.approx(eps, rel=1e-9), key
        # a swapped SCEE / SCNB index would give q q / 2.0 and eps / 1.2
        assert abs(got[0] / (qq * 1.2 / 2.0) - 1.0) > 0.3 and abs(got[2] / (eps * 2.0 / 1.2) - 1.0) > 0.3


def test_gaff_improper_keeps_atom_order_periodicity_and_phase(hostguest):
    tf = [f for f in hostguest.getForces() if isinstance(f, PeriodicTorsionForce)][0]
    atoms, n, phase, k = IMPROPER
    hits = [t for t in tf.torsions if tuple(t[:4]) == atoms]
    assert len(hits) == 1
    assert int(hits[0][4]) == n and hits[0][5] == pytest.approx(phase, abs=1e-9) and hits[0][6] == pytest.approx(k, rel=1e-12)
    # U = k (1 + cos(2 phi - pi)) = k (1 - cos 2 phi): minimum at the planar geometry phi = 0 or pi -- the sign of an improper
    phi = np.array([0.0, np.pi / 2])
    U = k * (1.0 + np.cos(n * phi - phase))
    assert U[0] < 1e-9 and U[1] == pytest.approx(2.0 * k, rel=1e-9)
