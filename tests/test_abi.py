"""The C-ABI library loads on a GPU-less host and exports every symbol include/rp_playroom.h declares."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, 'include', 'rp_playroom.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rp_[a-z_]+)\s*\(', src)))


def test_header_symbols_are_exported():
    from roboticsplayroompybullet_amd import _lib
    _lib.build()
    names = declared_functions()
    assert len(names) >= 15 and 'rp_step' in names and 'rp_create' in names
    for path in (_lib.LIB_PATH, _lib.WIDE_LIB_PATH):          # the library and its RP_WIDE build (two-object play ids)
        lib = ctypes.CDLL(path)
        for n in names:
            assert hasattr(lib, n), 'missing export %s in %s' % (n, path)
        lib.rp_version.restype = ctypes.c_char_p
        assert b'gfx950' in lib.rp_version()
    assert set(_lib.EXPORTS) == set(names)


def test_no_compute_without_gpu_and_no_cpu_fallback():
    import pytest
    import torch
    from roboticsplayroompybullet_amd import VecPlayEnv
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        VecPlayEnv('UR5PlayAbsRPY1Obj-v0', 4)
    with pytest.raises(NotImplementedError):
        VecPlayEnv('pointMass3D-v0', 4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'roboticsplayroompybullet_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cuh', '.h')) and 'generated' not in root:
                text = open(os.path.join(root, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'librp_oracle' not in text, f
