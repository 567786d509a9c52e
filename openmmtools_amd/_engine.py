"""ctypes binding of libremd_hip.so (the C ABI in include/remd_hip.h) and the HipEngine wrapper.

This is the product path.  There is deliberately NO CPU fallback: if the shared library is
missing, or no gfx950 device is visible, construction raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libremd_hip.so')

MIX_SCHEMES = {None: 0, 'none': 0, 'swap-all': 1, 'swap-neighbors': 2, 'sams-global-jump': 3}

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class RemdSystemDesc(C.Structure):
    _fields_ = [
        ('n_atoms', C.c_int32), ('mass', c_double_p),
        ('n_ext', C.c_int32), ('ext_atoms', c_int32_p),
        ('ext_K', C.c_double), ('ext_x0', C.c_double), ('ext_U0', C.c_double),
        ('n_bonds', C.c_int32), ('bond_atoms', c_int32_p), ('bond_params', c_double_p),
        ('n_angles', C.c_int32), ('angle_atoms', c_int32_p), ('angle_params', c_double_p),
        ('n_torsions', C.c_int32), ('torsion_atoms', c_int32_p), ('torsion_params', c_double_p),
        ('nb_method', C.c_int32), ('cutoff', C.c_double), ('switch_distance', C.c_double),
        ('rf_dielectric', C.c_double), ('ewald_alpha', C.c_double), ('pme_grid', C.c_int32 * 3),
        ('use_dispersion_correction', C.c_int32),
        ('charge', c_double_p), ('sigma', c_double_p), ('epsilon', c_double_p),
        ('n_exceptions', C.c_int32), ('exception_atoms', c_int32_p), ('exception_params', c_double_p),
        ('n_settle', C.c_int32), ('settle_atoms', c_int32_p), ('settle_dOH', C.c_double), ('settle_dHH', C.c_double),
        ('n_shake', C.c_int32), ('shake_atoms', c_int32_p), ('shake_dist', c_double_p),
        ('cmm_frequency', C.c_int32),
        ('n_alch', C.c_int32), ('alch_atoms', c_int32_p),
        ('softcore_alpha', C.c_double), ('softcore_a', C.c_double), ('softcore_b', C.c_double), ('softcore_c', C.c_double),
    ]


class RemdAlchRegionsDesc(C.Structure):
    """remd_alch_regions_desc of include/remd_hip.h (general alchemical regions)."""
    _fields_ = [
        ('n_atoms', C.c_int32), ('n_regions', C.c_int32), ('region_of_atom', c_int32_p), ('softcore', c_double_p), ('annihilate', c_int32_p),
        ('n_interactions', C.c_int32), ('interactions', c_int32_p),
        ('charge', c_double_p), ('sigma', c_double_p), ('epsilon', c_double_p),
        ('n_exceptions', C.c_int32), ('exception_atoms', c_int32_p), ('exception_params', c_double_p),
        ('electrostatics', C.c_int32),
        ('elec_alpha', C.c_double), ('elec_krf', C.c_double), ('elec_crf', C.c_double), ('elec_switch_distance', C.c_double),
        ('exact_pme', C.c_int32),
        ('n_bonds', C.c_int32), ('bond_atoms', c_int32_p), ('bond_params', c_double_p), ('bond_region', c_int32_p),
        ('n_angles', C.c_int32), ('angle_atoms', c_int32_p), ('angle_params', c_double_p), ('angle_region', c_int32_p),
        ('n_torsions', C.c_int32), ('torsion_atoms', c_int32_p), ('torsion_params', c_double_p), ('torsion_region', c_int32_p),
        ('consistent_exceptions', C.c_int32),
    ]


class RemdGbsaDesc(C.Structure):
    """remd_gbsa_desc of include/remd_hip.h (GBSA OBC2 + ACE of a NoCutoff system)."""
    _fields_ = [('n_atoms', C.c_int32), ('charge', c_double_p), ('radius', c_double_p), ('scale', c_double_p), ('alchemical', c_int32_p),
                ('solute_dielectric', C.c_double), ('solvent_dielectric', C.c_double), ('surface_area', C.c_int32)]


EXPORTS = [
    'remd_create', 'remd_destroy', 'remd_last_error', 'remd_version', 'remd_set_system', 'remd_set_coulomb_cutoff', 'remd_set_reaction_field', 'remd_set_alchemical_options', 'remd_set_alchemical_regions',
    'remd_set_region_lambdas', 'remd_set_region_bonded_lambdas', 'remd_set_gbsa', 'remd_set_states',
    'remd_set_integrator', 'remd_set_replicas', 'remd_set_replica_ids', 'remd_copy_replicas', 'remd_set_labels', 'remd_seed', 'remd_propagate',
    'remd_compute_energies', 'remd_ukl_device_ptr', 'remd_mix', 'remd_mix_host', 'remd_get_replicas',
    'remd_get_forces', 'remd_propagate_many', 'remd_set_phases', 'remd_get_phases', 'remd_get_constraint_stats', 'remd_step', 'remd_sync', 'remd_last_timing', 'remd_profile_enable',
    'remd_profile_get', 'remd_profile_reset', 'remd_test_fft3d', 'remd_get_energy_components', 'remd_profile_filter',
    'remd_set_restart_attempts', 'remd_set_force_groups', 'remd_set_work_measurement', 'remd_get_work', 'remd_reset_work', 'remd_minimize', 'remd_set_barostat', 'remd_get_boxes', 'remd_get_barostat_stats',
    'remd_barostat_attempts',
    'remd_set_energy_const_volume', 'remd_roof_microbench', 'remd_roof_clock_ghz', 'remd_roof_pair_step', 'remd_test_coulomb_table',
    'remd_comm_unique_id', 'remd_comm_init', 'remd_comm_all_gather_energies', 'remd_comm_finalize',
]

_lib = None

# the split of the Ewald sum a HipEngine on the device asks the host classes for when nothing else is said (HipEngine.__init__):
# 'auto' = system.rebalanced_coulomb_cutoff.  Measured on the three PME systems of BASELINE.json (profiles/r04_a_ewald_split_sweep.txt,
# ms per 500 MD steps, reference -> auto): 24 x alanine dipeptide 104.5 -> 95.0, 8 x host-guest 99.7 -> 91.4, 16 x DHFR 836 -> 790.
DEFAULT_EWALD_SPLIT = os.environ.get('REMD_EWALD_SPLIT', 'auto')


def load_library(path=None):
    """dlopen libremd_hip.so and declare every prototype of include/remd_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH        # another build of the same ABI only by an explicit lib_path (tests, tools): no environment hook
    if not os.path.exists(path):
        raise RuntimeError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 + libhsa-runtime64 (used for the RCCL
    # all-gather), and a process that initialises the system ROCm runtime first leaves torch with "No HIP GPUs are
    # available".  Loading torch first makes libremd_hip.so's NEEDED libamdhip64.so.7 resolve to the already loaded copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.remd_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    lib.remd_destroy.argtypes = [vp]
    lib.remd_last_error.argtypes = [vp]
    lib.remd_last_error.restype = C.c_char_p
    lib.remd_version.argtypes = []
    lib.remd_set_system.argtypes = [vp, C.POINTER(RemdSystemDesc)]
    lib.remd_set_coulomb_cutoff.argtypes = [vp, C.c_double]
    lib.remd_set_reaction_field.argtypes = [vp, C.c_int, C.c_double]
    lib.remd_set_alchemical_options.argtypes = [vp, C.c_int]
    lib.remd_set_alchemical_regions.argtypes = [vp, C.POINTER(RemdAlchRegionsDesc)]
    lib.remd_set_region_lambdas.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p]
    lib.remd_set_region_bonded_lambdas.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]
    lib.remd_set_gbsa.argtypes = [vp, C.POINTER(RemdGbsaDesc)]
    lib.remd_set_states.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.remd_set_integrator.argtypes = [vp, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double]
    lib.remd_set_restart_attempts.argtypes = [vp, C.c_int]
    lib.remd_set_force_groups.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.remd_set_work_measurement.argtypes = [vp, C.c_int, C.c_int]
    lib.remd_get_work.argtypes = [vp, c_double_p, c_double_p, c_int64_p, c_int64_p]
    lib.remd_reset_work.argtypes = [vp]
    lib.remd_minimize.argtypes = [vp, C.c_double, C.c_int, c_int32_p, c_int32_p]
    lib.remd_set_barostat.argtypes = [vp, C.c_int, c_double_p, C.c_int]
    lib.remd_get_boxes.argtypes = [vp, c_double_p]
    lib.remd_set_energy_const_volume.argtypes = [vp, C.c_double]
    lib.remd_get_barostat_stats.argtypes = [vp, c_double_p, c_int64_p, c_int64_p]
    lib.remd_barostat_attempts.argtypes = [vp, C.c_int]
    lib.remd_set_replicas.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int64_p]
    lib.remd_set_replica_ids.argtypes = [vp, c_int64_p]
    lib.remd_copy_replicas.argtypes = [vp, c_int32_p, vp, c_int32_p, C.c_int32, C.c_int32]
    lib.remd_set_labels.argtypes = [vp, c_int64_p]
    lib.remd_seed.argtypes = [vp, C.c_uint64]
    lib.remd_propagate.argtypes = [vp, C.c_int64, c_int32_p]
    lib.remd_compute_energies.argtypes = [vp, vp, c_double_p, c_double_p]
    lib.remd_ukl_device_ptr.argtypes = [vp, C.POINTER(vp)]
    lib.remd_mix.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp, C.c_int, c_int64_p, c_int64_p, c_int64_p,
                             c_double_p, c_double_p]
    lib.remd_mix_host.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, c_double_p, c_int64_p, c_int64_p,
                                  c_int64_p, c_double_p, c_double_p, C.c_int64]
    lib.remd_get_replicas.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.remd_get_forces.argtypes = [vp, c_double_p]
    lib.remd_propagate_many.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.POINTER(C.c_int32)]
    lib.remd_get_constraint_stats.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.remd_set_phases.argtypes = [vp, C.c_int32]
    lib.remd_get_phases.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.remd_step.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64, C.c_int]
    lib.remd_sync.argtypes = [vp]
    lib.remd_get_energy_components.argtypes = [vp, c_double_p]
    lib.remd_test_fft3d.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.remd_last_timing.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    lib.remd_profile_enable.argtypes = [vp, C.c_int]
    lib.remd_profile_get.argtypes = [vp, C.c_char_p, c_int64_p, c_double_p]
    lib.remd_profile_reset.argtypes = [vp]
    lib.remd_profile_filter.argtypes = [vp, C.c_char_p]
    lib.remd_roof_microbench.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    lib.remd_roof_clock_ghz.argtypes = [vp, c_double_p]
    lib.remd_roof_pair_step.argtypes = [vp, C.c_int, C.c_int, C.c_double, c_double_p, c_double_p]
    lib.remd_test_coulomb_table.argtypes = [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for name in EXPORTS:
        if name not in ('remd_last_error',):
            getattr(lib, name).restype = C.c_int
    if path == LIB_PATH:
        _lib = lib
    return lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def _lp(a):
    return None if a is None else a.ctypes.data_as(c_int64_p)


def build_desc(d):
    """Turn the dict from system.system_to_desc into a RemdSystemDesc (+ keep-alive list)."""
    keep = []

    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return _dp(a)

    def i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return _ip(a)

    s = RemdSystemDesc()
    s.n_atoms = int(d['n_atoms']); s.mass = f64(d['mass'])
    s.n_ext = int(d['n_ext']); s.ext_atoms = i32(d['ext_atoms'])
    s.ext_K, s.ext_x0, s.ext_U0 = float(d['ext_K']), float(d['ext_x0']), float(d['ext_U0'])
    s.n_bonds = len(d['bond_atoms']); s.bond_atoms = i32(d['bond_atoms']); s.bond_params = f64(d['bond_params'])
    s.n_angles = len(d['angle_atoms']); s.angle_atoms = i32(d['angle_atoms']); s.angle_params = f64(d['angle_params'])
    s.n_torsions = len(d['torsion_atoms']); s.torsion_atoms = i32(d['torsion_atoms']); s.torsion_params = f64(d['torsion_params'])
    s.nb_method = int(d['nb_method']); s.cutoff = float(d['cutoff']); s.switch_distance = float(d['switch_distance'])
    s.rf_dielectric = float(d['rf_dielectric']); s.ewald_alpha = float(d['ewald_alpha'])
    for k in range(3):
        s.pme_grid[k] = int(d['pme_grid'][k])
    s.use_dispersion_correction = int(d['use_dispersion_correction'])
    s.charge = f64(d['charge']); s.sigma = f64(d['sigma']); s.epsilon = f64(d['epsilon'])
    s.n_exceptions = len(d['exception_atoms']); s.exception_atoms = i32(d['exception_atoms'])
    s.exception_params = f64(d['exception_params'])
    s.n_settle = len(d['settle_atoms']); s.settle_atoms = i32(d['settle_atoms'])
    s.settle_dOH, s.settle_dHH = float(d['settle_dOH']), float(d['settle_dHH'])
    s.n_shake = len(d['shake_atoms']); s.shake_atoms = i32(d['shake_atoms']); s.shake_dist = f64(d['shake_dist'])
    s.cmm_frequency = int(d['cmm_frequency'])
    s.n_alch = len(d['alch_atoms']); s.alch_atoms = i32(d['alch_atoms'])
    s.softcore_alpha, s.softcore_a, s.softcore_b, s.softcore_c = [float(v) for v in d['softcore']]
    return s, keep


class HipEngine:
    """One libremd_hip.so handle = the batched device-state pool of one GPU (replaces cache.ContextCache,
    openmmtools/cache.py:378-461, for the replica-exchange hot path)."""

    is_device = True

    def __init__(self, device=0, stream=None, lib_path=None, ewald_split=None):
        """ewald_split: how the host classes split the Ewald sum of a PME System for this engine (system.system_to_desc):
        'reference' = OpenMM's rule on the NonbondedForce cutoff, 'auto' = a longer Coulomb range that buys a plane-friendly
        mesh (system.rebalanced_coulomb_cutoff), or a Coulomb range in nm.  Potentials agree to the Ewald tolerance either way.
        None: DEFAULT_EWALD_SPLIT on the device library; 'reference' for another build of the ABI (lib_path: the CPU baseline keeps
        the reference's own split)."""
        if ewald_split is None:
            ewald_split = DEFAULT_EWALD_SPLIT if lib_path is None else 'reference'
        self.ewald_split = ewald_split
        self.lib = load_library(lib_path)
        self.h = C.c_void_p()
        rc = self.lib.remd_create(C.byref(self.h), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError('remd_create failed (%d): %s' % (rc, self.lib.remd_last_error(None).decode()))
        self.device = device
        self._ctor = (device, stream, lib_path, ewald_split)
        self.N = self.K = self.R = self.R_global = self.r_begin = 0
        self._keep = None

    def spawn(self):
        """Another handle of the same library on the same device and stream (one per group of compatible states:
        multistate/_engine_pool.py)."""
        return type(self)(*self._ctor)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.remd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (what, rc, self.lib.remd_last_error(self.h).decode()))

    # ---- set-up ---------------------------------------------------------------------
    def set_system(self, desc_dict):
        s, keep = build_desc(desc_dict)
        # Ewald split (system.system_to_desc(ewald_split=...)): range of the direct-space Coulomb sum, 0 = the cutoff
        self._check(self.lib.remd_set_coulomb_cutoff(self.h, float(desc_dict.get('coulomb_cutoff', 0.0))), 'remd_set_coulomb_cutoff')
        # reaction field as the alchemical factory re-writes it (alchemical_rf_treatment='switched'): c_rf = 0 + switch
        rfw = desc_dict.get('rf_unshifted_switch_width')
        self._check(self.lib.remd_set_reaction_field(self.h, int(rfw is not None), float(rfw or 0.0)), 'remd_set_reaction_field')
        self._check(self.lib.remd_set_alchemical_options(self.h, int(bool(desc_dict.get('annihilate_sterics', False)))), 'remd_set_alchemical_options')
        self._check(self.lib.remd_set_system(self.h, C.byref(s)), 'remd_set_system')
        self.N = int(desc_dict['n_atoms'])
        self.n_regions = 0
        regions = desc_dict.get('alch_regions')
        if regions is not None:                          # general alchemical regions: the factory's custom forces (alchemy.py:1539-2038)
            keep = []

            def f64(a):
                keep.append(np.ascontiguousarray(a, dtype=np.float64)); return _dp(keep[-1])

            def i32(a):
                keep.append(np.ascontiguousarray(a, dtype=np.int32)); return _ip(keep[-1])
            r = RemdAlchRegionsDesc()
            r.n_atoms = self.N; r.n_regions = len(regions['softcore'])
            r.region_of_atom = i32(regions['region_of_atom']); r.softcore = f64(regions['softcore']); r.annihilate = i32(regions['annihilate'])
            r.n_interactions = len(regions['interactions']); r.interactions = i32(np.asarray(regions['interactions'], dtype=np.int32).reshape(-1, 2))
            r.charge, r.sigma, r.epsilon = f64(regions['charge']), f64(regions['sigma']), f64(regions['epsilon'])
            r.n_exceptions = len(regions['exception_atoms']); r.exception_atoms = i32(np.asarray(regions['exception_atoms'], dtype=np.int32).reshape(-1, 2))
            r.exception_params = f64(np.asarray(regions['exception_params'], dtype=np.float64).reshape(-1, 3))
            r.electrostatics = int(regions['electrostatics'])
            r.elec_alpha, r.elec_krf, r.elec_crf = float(regions['elec_alpha']), float(regions['elec_krf']), float(regions['elec_crf'])
            r.elec_switch_distance = float(regions['elec_switch_distance'])
            r.exact_pme = int(regions.get('exact_pme', 0))
            r.consistent_exceptions = int(regions.get('consistent_exceptions', 0))
            for kind, width, npar in (('bond', 2, 2), ('angle', 3, 2), ('torsion', 4, 3)):
                atoms = np.asarray(regions.get(kind + '_atoms', np.zeros((0, width))), dtype=np.int32).reshape(-1, width)
                setattr(r, 'n_%ss' % kind, len(atoms))
                setattr(r, kind + '_atoms', i32(atoms))
                setattr(r, kind + '_params', f64(np.asarray(regions.get(kind + '_params', np.zeros((0, npar))), dtype=np.float64).reshape(-1, npar)))
                setattr(r, kind + '_region', i32(np.asarray(regions.get(kind + '_region', np.zeros(0)), dtype=np.int32)))
            self._check(self.lib.remd_set_alchemical_regions(self.h, C.byref(r)), 'remd_set_alchemical_regions')
            self.n_regions = int(r.n_regions)
        gb = desc_dict.get('gbsa')
        if gb is not None:                               # GBSAOBCForce of an implicit-solvent system (alchemy.py:2144-2225)
            arrs = [np.ascontiguousarray(gb[k], dtype=np.float64) for k in ('charge', 'radius', 'scale')]
            alch = np.ascontiguousarray(gb.get('alchemical', np.zeros(self.N)), dtype=np.int32)
            g = RemdGbsaDesc(self.N, _dp(arrs[0]), _dp(arrs[1]), _dp(arrs[2]), _ip(alch), float(gb['solute_dielectric']), float(gb['solvent_dielectric']),
                             int(gb.get('surface_area', 1)))
            self._check(self.lib.remd_set_gbsa(self.h, C.byref(g)), 'remd_set_gbsa')
        if 'force_groups' in desc_dict:                  # Force.getForceGroup() of the force classes (V<g> substeps)
            self.set_force_groups(desc_dict['force_groups'])

    def set_states(self, beta, lambda_sterics=None, lambda_electrostatics=None, energy_const=None):
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64)
                for a in (lambda_sterics, lambda_electrostatics, energy_const)]
        self._check(self.lib.remd_set_states(self.h, len(beta), _dp(beta), *[_dp(a) for a in arrs]), 'remd_set_states')
        self.K = len(beta)

    def set_region_lambdas(self, lambda_sterics, lambda_electrostatics):
        """[K][n_regions] lambdas of general alchemical regions (after set_states)."""
        ls = np.ascontiguousarray(lambda_sterics, dtype=np.float64).reshape(self.K, -1)
        le = np.ascontiguousarray(lambda_electrostatics, dtype=np.float64).reshape(self.K, -1)
        self._check(self.lib.remd_set_region_lambdas(self.h, self.K, ls.shape[1], _dp(ls), _dp(le)), 'remd_set_region_lambdas')

    def set_region_bonded_lambdas(self, lambda_bonds=None, lambda_angles=None, lambda_torsions=None):
        """[K][n_regions] lambdas of the alchemically softened bonded terms (after set_region_lambdas; None: 1)."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(self.K, -1) for a in (lambda_bonds, lambda_angles, lambda_torsions)]
        n = next((a.shape[1] for a in arrs if a is not None), self.n_regions)
        self._check(self.lib.remd_set_region_bonded_lambdas(self.h, self.K, n, *[_dp(a) for a in arrs]), 'remd_set_region_bonded_lambdas')

    def set_integrator(self, splitting, timestep, collision_rate, n_steps, reassign_velocities=True,
                       constraint_tolerance=1e-8):
        self._check(self.lib.remd_set_integrator(self.h, splitting.encode(), float(timestep), float(collision_rate),
                                                 int(n_steps), int(bool(reassign_velocities)),
                                                 float(constraint_tolerance)), 'remd_set_integrator')

    def set_work_measurement(self, measure_heat=False, measure_shadow_work=False):
        """LangevinIntegrator(measure_heat=, measure_shadow_work=) (integrators.py:1077-1125)."""
        self._check(self.lib.remd_set_work_measurement(self.h, int(bool(measure_heat)), int(bool(measure_shadow_work))), 'remd_set_work_measurement')

    def get_work(self):
        """dict(heat, shadow_work [kJ/mol], n_accepted, n_trials) per local replica, accumulated since the last reset_work."""
        heat, sw = np.zeros(self.R), np.zeros(self.R)
        na, nt = np.zeros(self.R, np.int64), np.zeros(self.R, np.int64)
        self._check(self.lib.remd_get_work(self.h, _dp(heat), _dp(sw), _lp(na), _lp(nt)), 'remd_get_work')
        return dict(heat=heat, shadow_work=sw, n_accepted=na, n_trials=nt)

    def reset_work(self):
        self._check(self.lib.remd_reset_work(self.h), 'remd_reset_work')

    def set_force_groups(self, groups):
        """Force groups of (external, bonds, angles, torsions, nonbonded direct, PME reciprocal) for V<g> substeps."""
        g = (C.c_int32 * 6)(*[int(x) for x in groups])
        self._check(self.lib.remd_set_force_groups(self.h, g), 'remd_set_force_groups')

    # ---- sharding through RCCL inside the library (include/remd_hip.h; a torch.distributed host may use multistate/comm.py) ----
    def comm_unique_id(self):
        """Rank 0: the 128 opaque bytes every other rank needs for ``comm_init`` (passed on by the host's own channel)."""
        buf = (C.c_ubyte * 128)()
        self._check(self.lib.remd_comm_unique_id(buf), 'remd_comm_unique_id')
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        if len(unique_id) != 128:
            raise ValueError('the communicator id is 128 bytes')
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._check(self.lib.remd_comm_init(self.h, int(rank), int(world), buf), 'remd_comm_init')

    def comm_all_gather_energies(self):
        """After ``compute_energies`` into the handle's own matrix: every other rank's rows, on the handle's stream."""
        self._check(self.lib.remd_comm_all_gather_energies(self.h), 'remd_comm_all_gather_energies')

    def comm_finalize(self):
        self._check(self.lib.remd_comm_finalize(self.h), 'remd_comm_finalize')

    def set_restart_attempts(self, n):
        """mcmc.py:706-759: retries of a move whose result holds a NaN (restored start state, fresh noise)."""
        self._check(self.lib.remd_set_restart_attempts(self.h, int(n)), 'remd_set_restart_attempts')

    def set_barostat(self, pressure, frequency=25):
        """Monte Carlo barostat of the NPT states (states.py:1177-1181): pressure per state in kJ/mol/nm^3, or None."""
        if pressure is None:
            self._check(self.lib.remd_set_barostat(self.h, 0, None, 0), 'remd_set_barostat')
            return
        p = np.ascontiguousarray(pressure, dtype=np.float64)
        self._check(self.lib.remd_set_barostat(self.h, len(p), _dp(p), int(frequency)), 'remd_set_barostat')

    def set_energy_const_volume(self, volume):
        """The energy constants of set_states scale as volume / V with the replica's box (NPT + alchemical states)."""
        self._check(self.lib.remd_set_energy_const_volume(self.h, float(volume)), 'remd_set_energy_const_volume')

    def get_boxes(self):
        box = np.zeros((self.R, 3), dtype=np.float64)
        self._check(self.lib.remd_get_boxes(self.h, _dp(box)), 'remd_get_boxes')
        return box

    def barostat_attempts(self, n_attempts):
        """mcmc.py:1597-1700 MonteCarloBarostatMove: n volume moves of every local replica outside the integrator."""
        self._check(self.lib.remd_barostat_attempts(self.h, int(n_attempts)), 'remd_barostat_attempts')

    def barostat_stats(self):
        vs = np.zeros(self.R); na = np.zeros(self.R, np.int64); nc = np.zeros(self.R, np.int64)
        self._check(self.lib.remd_get_barostat_stats(self.h, _dp(vs), na.ctypes.data_as(c_int64_p), nc.ctypes.data_as(c_int64_p)),
                    'remd_get_barostat_stats')
        return vs, na, nc

    def minimize(self, tolerance=1.0, max_iterations=0):
        """FIRE minimisation of every local replica (multistatesampler.py:611-647).  tolerance in kJ/mol/nm.
        Returns (converged flags, FIRE steps taken)."""
        conv = np.zeros(self.R, dtype=np.int32)
        n = C.c_int32(0)
        self._check(self.lib.remd_minimize(self.h, float(tolerance), int(max_iterations), conv.ctypes.data_as(c_int32_p),
                                           C.byref(n)), 'remd_minimize')
        return conv, n.value

    def set_replicas(self, R_global, r_begin, x, v, box, labels):
        """x = None sizes the handle for len(box) replicas whose coordinates follow through copy_replicas."""
        box = np.ascontiguousarray(box, dtype=np.float64)
        R_local = box.reshape(-1, 3).shape[0] if x is None else np.shape(x)[0]
        x = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        v = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        box = box.reshape(R_local, 3)
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        assert (x is None or x.shape == (R_local, self.N, 3)) and labels.shape == (R_global,)
        self._check(self.lib.remd_set_replicas(self.h, int(R_global), int(r_begin), R_local, _dp(x), _dp(v),
                                               _dp(box), _lp(labels)), 'remd_set_replicas')
        self.R, self.R_global, self.r_begin = R_local, int(R_global), int(r_begin)

    POSITIONS, VELOCITIES, BOXES = 1, 2, 4

    def copy_replicas(self, slots, source, source_slots, what=7):
        """Positions / velocities / boxes (bits 1 / 2 / 4 of ``what``) of ``source``'s local replicas ``source_slots`` into this
        handle's local replicas ``slots``, device to device (remd_copy_replicas; one handle per compatibility group)."""
        d = np.ascontiguousarray(slots, dtype=np.int32)
        s = np.ascontiguousarray(source_slots, dtype=np.int32)
        if d.shape != s.shape or d.ndim != 1:
            raise ValueError('one source slot per destination slot')
        self._check(self.lib.remd_copy_replicas(self.h, d.ctypes.data_as(c_int32_p), source.h, s.ctypes.data_as(c_int32_p),
                                                int(len(d)), int(what)), 'remd_copy_replicas')

    def set_replica_ids(self, ids):
        """Global replica indices that key the local replicas' random streams (a handle holding a non-contiguous subset of the
        ensemble: multistate/_engine_pool.py); None restores r_begin + r.  Call after set_replicas."""
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        if ids is not None and ids.shape != (self.R,):
            raise ValueError('one id per local replica')
        self._check(self.lib.remd_set_replica_ids(self.h, _lp(ids)), 'remd_set_replica_ids')

    def set_labels(self, labels):
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        self._check(self.lib.remd_set_labels(self.h, _lp(labels)), 'remd_set_labels')

    def seed(self, seed):
        self._check(self.lib.remd_seed(self.h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)), 'remd_seed')

    # ---- hot path ----------------------------------------------------------------------
    def propagate(self, iteration):
        flags = np.zeros(self.R, dtype=np.int32)
        self._check(self.lib.remd_propagate(self.h, int(iteration), _ip(flags)), 'remd_propagate')
        return flags

    def constraint_stats(self):
        """(most Newton updates an X-H position solve has needed, whether one ever reached the bound unconverged)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.remd_get_constraint_stats(self.h, C.byref(a), C.byref(b)), 'remd_get_constraint_stats')
        return int(a.value), bool(b.value)

    def set_phases(self, n):
        """remd_set_phases: 0 = by rule (two blocks of replicas whose MD steps take turns when the process keeps few hardware queues),
        1 = one block, 2 = two blocks"""
        self._check(self.lib.remd_set_phases(self.h, int(n)), 'remd_set_phases')

    def phases_active(self):
        """blocks the last propagate ran as"""
        n = C.c_int32(1)
        self._check(self.lib.remd_get_phases(self.h, C.byref(n)), 'remd_get_phases')
        return int(n.value)

    @staticmethod
    def propagate_many(engines, iteration):
        """remd_propagate_many: the engines' MD steps take turns on one device from one host thread; returns the NaN flags per engine"""
        lib = engines[0].lib
        hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
        flags = np.zeros(sum(e.R for e in engines), dtype=np.int32)
        engines[0]._check(lib.remd_propagate_many(hs, len(engines), int(iteration), _ip(flags)), 'remd_propagate_many')
        return np.split(flags, np.cumsum([e.R for e in engines])[:-1])

    def compute_energies(self, d_rows=None, want_host=True, want_potential=False):
        """Rows [R_local, K] of u_kl.  d_rows: device pointer (int) or None for the handle's matrix."""
        ukl = np.empty((self.R, self.K), dtype=np.float64) if want_host else None
        pot = np.empty(self.R, dtype=np.float64) if want_potential else None
        self._check(self.lib.remd_compute_energies(self.h, C.c_void_p(d_rows) if d_rows else None, _dp(ukl), _dp(pot)),
                    'remd_compute_energies')
        return (ukl, pot) if want_potential else ukl

    def ukl_device_ptr(self):
        p = C.c_void_p()
        self._check(self.lib.remd_ukl_device_ptr(self.h, C.byref(p)), 'remd_ukl_device_ptr')
        return p.value

    def mix(self, scheme, iteration, labels, d_ukl=None, R=None, K=None, ld=0, log_weights=None):
        R = self.R_global if R is None else R
        K = self.K if K is None else K
        labels = np.ascontiguousarray(labels, dtype=np.int64).copy()
        nacc = np.zeros((K, K), dtype=np.int64)
        nprop = np.zeros((K, K), dtype=np.int64)
        sid = MIX_SCHEMES[scheme]
        logw = None if log_weights is None else np.ascontiguousarray(log_weights, dtype=np.float64)
        logP = np.zeros((R, K), dtype=np.float64) if sid == 3 else None
        self._check(self.lib.remd_mix(self.h, sid, int(iteration), R, K, C.c_void_p(d_ukl) if d_ukl else None,
                                      int(ld), _lp(labels), _lp(nacc), _lp(nprop), _dp(logw), _dp(logP)), 'remd_mix')
        return labels, nacc, nprop, logP

    def mix_host(self, scheme, iteration, ukl, labels, log_weights=None, n_attempts=-1):
        ukl = np.ascontiguousarray(ukl, dtype=np.float64)
        R, K = ukl.shape
        labels = np.ascontiguousarray(labels, dtype=np.int64).copy()
        nacc = np.zeros((K, K), dtype=np.int64)
        nprop = np.zeros((K, K), dtype=np.int64)
        sid = MIX_SCHEMES[scheme]
        logw = None if log_weights is None else np.ascontiguousarray(log_weights, dtype=np.float64)
        logP = np.zeros((R, K), dtype=np.float64) if sid == 3 else None
        self._check(self.lib.remd_mix_host(self.h, sid, int(iteration), R, K, _dp(ukl), _lp(labels), _lp(nacc),
                                           _lp(nprop), _dp(logw), _dp(logP), int(n_attempts)), 'remd_mix_host')
        return labels, nacc, nprop, logP

    # ---- snapshots / hooks ---------------------------------------------------------------
    def get_replicas(self, positions=True, velocities=True, potential=False, kinetic=False):
        x = np.empty((self.R, self.N, 3)) if positions else None
        v = np.empty((self.R, self.N, 3)) if velocities else None
        u = np.empty(self.R) if potential else None
        k = np.empty(self.R) if kinetic else None
        self._check(self.lib.remd_get_replicas(self.h, _dp(x), _dp(v), _dp(u), _dp(k)), 'remd_get_replicas')
        return x, v, u, k

    def get_forces(self):
        f = np.empty((self.R, self.N, 3))
        self._check(self.lib.remd_get_forces(self.h, _dp(f)), 'remd_get_forces')
        return f

    def step(self, splitting, iteration=0, first_step=0, n_steps=1):
        self._check(self.lib.remd_step(self.h, splitting.encode(), int(iteration), int(first_step), int(n_steps)),
                    'remd_step')

    def energy_components(self):
        names = ['external', 'bonds', 'angles', 'torsions', 'exceptions', 'ewald_exclusions', 'pme_reciprocal',
                 'constants', 'nonbonded_direct']
        out = np.zeros((self.R, 9))
        self._check(self.lib.remd_get_energy_components(self.h, _dp(out)), 'remd_get_energy_components')
        return [dict(zip(names, row)) for row in out]

    def test_fft3d(self, data, inverse=False):
        a = np.ascontiguousarray(data, dtype=np.complex64).copy()
        nx, ny, nz = a.shape
        self._check(self.lib.remd_test_fft3d(self.h, nx, ny, nz, a.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)),
                                             int(bool(inverse))), 'remd_test_fft3d')
        return a

    def sync(self):
        self._check(self.lib.remd_sync(self.h), 'remd_sync')

    def last_timing(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.lib.remd_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(propagate_ms=a.value, energies_ms=b.value, mix_ms=c.value)

    def roof_microbench(self):
        """Achievable roofs of this box: STREAM triad GB/s, v_fma_f32 and v_pk_fma_f32 TFLOP/s (include/remd_hip.h)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.remd_roof_microbench(self.h, C.byref(a), C.byref(b), C.byref(c)), 'remd_roof_microbench')
        out = dict(stream_triad_gb_per_s=a.value, fma_f32_tflop_per_s=b.value, pk_fma_f32_tflop_per_s=c.value)
        g = C.c_double()
        if self.lib.remd_roof_clock_ghz(self.h, C.byref(g)) == 0 and g.value > 0:
            # 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 flop at 2.4 GHz: what the same issue rate gives at the measured clock
            out['shader_clock_ghz_under_fma_load'] = g.value
            out['fma_f32_peak_at_that_clock_tflop_per_s'] = 157.3 * g.value / 2.4
        return out

    def roof_pair_step(self, waves_per_simd, chains=1, ghz=2.4):
        """(cycles per cluster-pair step per SIMD at `ghz`, microseconds of the launch): the pair kernel's step replayed from registers."""
        c, us = C.c_double(), C.c_double()
        self._check(self.lib.remd_roof_pair_step(self.h, int(waves_per_simd), int(chains), float(ghz), C.byref(c), C.byref(us)), 'remd_roof_pair_step')
        return c.value, us.value

    def profile_enable(self, on=1, kernel_class=None):
        if kernel_class is not None:
            self.lib.remd_profile_filter(self.h, kernel_class.encode())
        self.lib.remd_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.lib.remd_profile_reset(self.h)

    def profile_get(self, name):
        n, ms = C.c_int64(), C.c_double()
        self.lib.remd_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms))
        return n.value, ms.value
