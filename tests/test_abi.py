"""The C-ABI library loads on a GPU-less host and exports every symbol include/rp_playroom.h declares."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header='rp_playroom.h'):
    src = open(os.path.join(REPO, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rp_[a-z_0-9]+)\s*\(', src)))


def exported_functions(path):
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], check=True, capture_output=True, text=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == 'T' and ln.split()[-1].startswith('rp_')})


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


def test_every_exported_symbol_is_declared_in_a_header():
    """the converse: the library exports no rp_* function that neither include/rp_playroom.h nor include/rp_playroom_debug.h declares"""
    from roboticsplayroompybullet_amd import _lib
    _lib.build()
    public, debug = set(declared_functions()), set(declared_functions('rp_playroom_debug.h'))
    assert set(_lib.DEBUG_EXPORTS) == debug and not (public & debug)
    for path in (_lib.LIB_PATH, _lib.WIDE_LIB_PATH):
        exported = set(exported_functions(path))
        assert exported, path
        assert exported <= public | debug, 'undeclared exports in %s: %s' % (path, sorted(exported - public - debug))
        assert public | debug <= exported


def test_config_struct_matches_the_header():
    """the ctypes mirror of rp_config / rp_out has the field order of include/rp_playroom.h"""
    from roboticsplayroompybullet_amd import _lib
    src = open(os.path.join(REPO, 'include', 'rp_playroom.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    for name, cls in (('rp_config', _lib.RpConfig), ('rp_out', _lib.RpOut), ('rp_dims', _lib.RpDims)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(',')[0:1] + decl.split(',')[1:]:
                fields.append(re.sub(r'\[.*?\]', '', part.replace('*', ' ')).split()[-1])
        assert fields == [f for f, _ in cls._fields_], (name, fields)


def test_enum_values_match_the_header():
    """the Python mirror of the header's enumerations: rp_config_flags (CFG_*), rp_env_kind (ENV_KINDS covers 0 .. RP_ENV_COUNT - 1), rp_action_type"""
    from roboticsplayroompybullet_amd import _lib
    src = open(os.path.join(REPO, 'include', 'rp_playroom.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    enums = {k: int(v) for k, v in re.findall(r'\b(RP_[A-Z0-9_]+)\s*=\s*(-?\d+)', src)}
    flags = {k[len('RP_CFG_'):]: v for k, v in enums.items() if k.startswith('RP_CFG_')}
    assert len(flags) >= 9
    for name, value in flags.items():
        assert getattr(_lib, 'CFG_' + name) == value, name
    assert sorted(_lib.ENV_KINDS.values()) == list(range(enums['RP_ENV_COUNT']))
    actions = {k[len('RP_ACTION_'):].lower(): v for k, v in enums.items() if k.startswith('RP_ACTION_')}
    assert actions == _lib.ACTION_TYPE_CODES


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


def test_pair_env_entries_are_decoded_through_the_masked_helper_only():
    """a pair_env entry carries its env's contact count in the top byte; hipcc 7.2 miscompiled the plain-C mask in front of a 64-bit address
    multiply once (rp_kernels.cuh pair_env_id), so every read of the table must go through pair_env_id() (the env) or `>> 24` (the count),
    and nothing else may touch the raw word"""
    import re
    src = open(os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'rp_kernels.cuh')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)                                              # comments
    src = re.sub(r'__device__ __forceinline__ int pair_env_id\(int pe\) \{.*?\n\}', '', src, flags=re.S)      # the helper itself
    reads = re.findall(r'const int (\w+) = [^;]*(?:pair_env|hv_list)\[[^;]*;', src)      # (round 6: k_solve2's list of heavy envs carries the same words)
    assert len(reads) >= 3, reads
    for name in set(reads):
        for line in src.splitlines():
            if 'pair_env[' in line or 'hv_list[' in line or not re.search(r'\b%s\b' % name, line):
                continue
            rest = line
            for allowed in (r'pair_env_id\(%s\)', r'\(?%s >> 24\)?', r'%s < 0', r'%s >= 0'):
                rest = re.sub(allowed % name, '', rest)
            assert not re.search(r'\b%s\b' % name, rest), 'raw use of a pair_env word: %s' % line.strip()
