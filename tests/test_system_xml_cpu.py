"""OpenMM System XML <-> the engine's System shim (openmmtools_amd/system_xml.py; SURVEY 8(f) rank 4 adapter).
OpenMM itself is absent: round trips through the module's own writer + one hand-written document in OpenMM's layout."""
import numpy as np
import pytest
from openmmtools_amd import testsystems as ts, unit
from openmmtools_amd.system import system_to_desc, NonbondedForce
from openmmtools_amd import system_xml


def _same_desc(a, b):
    assert set(a) == set(b)
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize('cls', [ts.HarmonicOscillator, ts.AlanineDipeptideExplicit,
                                 lambda: ts.LennardJonesFluid(nparticles=64)])
def test_round_trip_reproduces_the_engine_description(cls):
    """What the engine is programmed with (system_to_desc: every parameter table, constraints, PME mesh) is identical
    for a system and for its XML round trip; repr() floats make the text lossless."""
    t = cls()
    text = system_xml.to_xml(t.system)
    assert text.startswith('<?xml') and '<System' in text
    back, baro = system_xml.from_xml(text)
    assert baro is None
    assert back.getNumParticles() == t.system.getNumParticles() and back.getNumForces() == t.system.getNumForces()
    _same_desc(system_to_desc(t.system), system_to_desc(back))
    assert system_xml.to_xml(back) == text


def test_reads_a_document_in_openmm_layout(tmp_path):
    """Two TIP3P-like waters written the way OpenMM's XmlSerializer lays a System out (extra attributes are ignored)."""
    doc = '''<?xml version="1.0" ?>
<System openmmVersion="8.1" type="System" version="1">
	<PeriodicBoxVectors>
		<A x="2.5" y="0" z="0"/>
		<B x="0" y="2.5" z="0"/>
		<C x="0" y="0" z="2.5"/>
	</PeriodicBoxVectors>
	<Particles>
		<Particle mass="15.99943"/>
		<Particle mass="1.007947"/>
		<Particle mass="1.007947"/>
		<Particle mass="15.99943"/>
		<Particle mass="1.007947"/>
		<Particle mass="1.007947"/>
	</Particles>
	<Constraints>
		<Constraint d=".09572" p1="0" p2="1"/>
		<Constraint d=".09572" p1="0" p2="2"/>
		<Constraint d=".15139" p1="1" p2="2"/>
		<Constraint d=".09572" p1="3" p2="4"/>
		<Constraint d=".09572" p1="3" p2="5"/>
		<Constraint d=".15139" p1="4" p2="5"/>
	</Constraints>
	<Forces>
		<Force forceGroup="0" name="HarmonicBondForce" type="HarmonicBondForce" usesPeriodic="0" version="2">
			<Bonds/>
		</Force>
		<Force alpha="0" cutoff="1" dispersionCorrection="1" ewaldTolerance=".0005" exceptionsUsePeriodic="0" forceGroup="0" includeDirectSpace="1" ljAlpha="0" ljnx="0" ljny="0" ljnz="0" method="4" name="NonbondedForce" nx="0" ny="0" nz="0" recipForceGroup="-1" rfDielectric="78.3" switchingDistance=".9" type="NonbondedForce" useSwitchingFunction="1" version="4">
			<GlobalParameters/>
			<ParticleOffsets/>
			<ExceptionOffsets/>
			<Particles>
				<Particle eps=".635968" q="-.834" sig=".3150752406575124"/>
				<Particle eps="0" q=".417" sig="1"/>
				<Particle eps="0" q=".417" sig="1"/>
				<Particle eps=".635968" q="-.834" sig=".3150752406575124"/>
				<Particle eps="0" q=".417" sig="1"/>
				<Particle eps="0" q=".417" sig="1"/>
			</Particles>
			<Exceptions>
				<Exception eps="0" p1="0" p2="1" q="0" sig="1"/>
				<Exception eps="0" p1="0" p2="2" q="0" sig="1"/>
				<Exception eps="0" p1="1" p2="2" q="0" sig="1"/>
				<Exception eps="0" p1="3" p2="4" q="0" sig="1"/>
				<Exception eps="0" p1="3" p2="5" q="0" sig="1"/>
				<Exception eps="0" p1="4" p2="5" q="0" sig="1"/>
			</Exceptions>
		</Force>
		<Force forceGroup="0" frequency="1" name="CMMotionRemover" type="CMMotionRemover" version="1"/>
		<Force forceGroup="0" frequency="25" name="MonteCarloBarostat" pressure="1.01325" randomSeed="0" temperature="298" type="MonteCarloBarostat" version="1"/>
	</Forces>
</System>
'''
    path = tmp_path / 'system.xml'
    path.write_text(doc)
    s, baro = system_xml.from_xml(str(path))
    assert baro == dict(pressure=1.01325, temperature=298.0, frequency=25)
    assert s.getNumParticles() == 6 and s.getNumConstraints() == 6 and s.getNumForces() == 3
    nb = [f for f in s.getForces() if isinstance(f, NonbondedForce)][0]
    assert nb.getNonbondedMethod() == NonbondedForce.PME and nb.getUseSwitchingFunction() and nb.getSwitchingDistance() == 0.9
    assert nb.getParticleParameters(3) == (-0.834, 0.3150752406575124, 0.635968) and nb.getNumExceptions() == 6
    d = system_to_desc(s)
    assert d['settle_atoms'].shape == (2, 3) and d['n_atoms'] == 6 and d['cmm_frequency'] == 1 and d['nb_method'] == 2
    # the pressure rides on the thermodynamic state, as in the reference (states.py:1020-1068)
    from openmmtools_amd import states
    st = states.ThermodynamicState(s, baro['temperature'] * unit.kelvin, pressure=baro['pressure'] * unit.bar)
    assert st.pressure is not None and st.barostat_frequency == 25
    text = system_xml.to_xml(s, pressure=baro['pressure'], temperature=baro['temperature'])
    s2, baro2 = system_xml.from_xml(text)
    assert baro2 == baro
    _same_desc(d, system_to_desc(s2))


def test_unsupported_content_is_refused_by_name():
    with pytest.raises(NotImplementedError, match='CustomGBForce'):
        system_xml.from_xml('<System><Particles/><Forces><Force type="CustomGBForce"/></Forces></System>')
    # offsets are understood as the alchemical factory's lambda_electrostatics only (_alchemical_xml.py)
    with pytest.raises(NotImplementedError, match="offset parameter 'l'"):
        system_xml.from_xml('<System><Forces><Force type="NonbondedForce" method="0" cutoff="1"><ParticleOffsets>'
                            '<Offset parameter="l" q="1" sig="0" eps="0" particle="0"/></ParticleOffsets></Force></Forces></System>')
    with pytest.raises(ValueError):
        system_xml.from_xml('<State/>')
