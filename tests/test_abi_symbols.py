"""CPU: the C-ABI library loads and exports every symbol include/pmbrl.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from tests import common


def test_header_symbols_exported():
    from prob_mbrl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    header = open(os.path.join(common.ROOT, 'include', 'pmbrl.h')).read()
    declared = set(re.findall(r'\b(pmbrl_[a-z_0-9]+)\s*\(', header))
    assert len(declared) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), 'missing export %s' % name
    for name in _lib.EXPORTS:
        assert name in declared


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (int32/float arrays, natural alignment)."""
    from prob_mbrl_amd import _lib
    assert ctypes.sizeof(_lib.MLP) == 4 + 4 * 9 + 4 * 8
    assert ctypes.sizeof(_lib.Reward) == 4 * (3 + 8 + 1) + 4 * (8 * 64 + 8 + 2 + 64 + 64 * 64)
    lib = _lib.load()
    assert lib.pmbrl_version() >= 6
    assert re.fullmatch(rb'[0-9a-f]{16}', lib.pmbrl_build_id())
    assert lib.pmbrl_last_error() is not None


def test_plan_rejects_bad_shapes_without_gpu():
    """Validation errors come back as negative codes + message (no exceptions cross the ABI)."""
    from prob_mbrl_amd import _lib
    lib = _lib.load()
    cfg = _lib.Config()
    plan = ctypes.c_void_p()
    rc = lib.pmbrl_plan_create(ctypes.byref(cfg), 0, ctypes.byref(plan))
    assert rc < 0
    assert b'must be' in lib.pmbrl_last_error()
