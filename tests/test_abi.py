"""CPU-only: the C-ABI library loads and exports every symbol include/remd_hip.h declares; the
product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re
import ctypes
import pytest
from openmmtools_amd import _engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'remd_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(remd_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported():
    if not os.path.exists(_engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_engine.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libremd_hip.so does not export %s' % name
    assert sorted(_engine.EXPORTS) == declared


def test_prototypes_bind():
    lib = _engine.load_library()
    assert lib.remd_version() >= 1


def test_desc_struct_matches_header_field_order():
    text = open(os.path.join(ROOT, 'include', 'remd_hip.h')).read()
    body = text[text.index('typedef struct remd_system_desc {'):text.index('} remd_system_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S).replace('typedef struct remd_system_desc {', '')
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(','):
            m = re.search(r'([A-Za-z_][A-Za-z_0-9]*)\s*(\[\d+\])?\s*$', part.strip())
            names.append(m.group(1))
    assert names == [f[0] for f in _engine.RemdSystemDesc._fields_]


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='a GPU is present')
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(RuntimeError):
        _engine.HipEngine()
