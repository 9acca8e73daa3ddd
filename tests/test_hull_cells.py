"""CPU: the support-vertex candidate tables of the arm links' hulls (csrc/generated/rp_hullcells_gen.h, tools/bake_hull_cells.py) are EXACT: scanning the cube-map cell of a
direction returns the very vertex a scan of the whole hull returns - the largest (smallest) computed coordinate in the library's fp32 arithmetic (hull_coord's fused
sequence), the lowest vertex number among equals - for every direction.  The HIP library's hull contacts and GJK support queries read the tables (rp_kernels.cuh
hull_item); the oracle keeps scanning whole hulls, so the device parity tests check the same thing end to end; this test checks it directly, on the library's own lookup
arithmetic restated in C (rp_oracle.c rpo_hullcell_of / rpo_hull_support), over
  * random directions, unit length with a random box-centre coordinate subtracted (the face scan, the probe) and of length 1e-5 .. 1 without (GJK's directions),
  * the coordinate axes and steps of 1e-8 .. 1 off them (a link lying flat: whole rim circles tie),
  * directions on and within 1e-7 / 3e-6 of the borders of the cube map's cells and faces (the lookup's own rounding),
each as a maximum and as a minimum query, every hull of both arms: 2.4 million queries per arm.  Also: the committed header is what the bake tool writes."""
import ctypes as C
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import oracle  # noqa: E402


@pytest.mark.parametrize('kind,name', [(0, 'UR5'), (2, 'Panda')])
def test_candidate_tables_return_the_full_scans_vertex(kind, name):
    lib = oracle.load(f32=True)
    lib.rpo_hullcell_selftest.restype = C.c_long
    lib.rpo_hullcell_selftest.argtypes = [C.c_int, C.c_long, C.c_int, C.c_ulonglong, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    total = 0
    for mode, what in ((0, 'random'), (1, 'axes'), (2, 'cell borders')):
        q, mc = C.c_long(), C.c_double()
        per_hull = 40000 if kind == 0 else 55000
        bad = lib.rpo_hullcell_selftest(kind, per_hull, mode, 20240 + mode, C.byref(q), C.byref(mc))
        print('%s, %s directions: %d queries, %d mismatches, %.1f candidates looked at per query' % (name, what, q.value, bad, mc.value))
        assert bad == 0, (name, what, bad)
        total += q.value
    assert total >= 2_400_000


def test_committed_header_is_what_the_bake_writes(tmp_path):
    header = os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'generated', 'rp_hullcells_gen.h')
    fresh = str(tmp_path / 'rp_hullcells_gen.h')
    subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'bake_hull_cells.py'), '--out', fresh], check=True, capture_output=True)
    assert open(fresh).read() == open(header).read(), 'csrc/generated/rp_hullcells_gen.h is stale: run tools/bake_hull_cells.py'


def test_committed_plane_header_is_what_the_bake_writes(tmp_path):
    """csrc/generated/rp_hullplanes_gen.h (the hulls' face planes for the ray caster: rp_render / rp_ray_test draw an arm link as the hull it collides as) against a
    fresh run of tools/bake_hull_planes.py; and every plane keeps every vertex of its hull on its inner side"""
    import re
    import numpy as np
    header = os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'generated', 'rp_hullplanes_gen.h')
    fresh = str(tmp_path / 'rp_hullplanes_gen.h')
    subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'bake_hull_planes.py'), '--out', fresh], check=True, capture_output=True)
    assert open(fresh).read() == open(header).read(), 'csrc/generated/rp_hullplanes_gen.h is stale: run tools/bake_hull_planes.py'
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import bake_hull_cells as bc
    table, ints = bc.parse(open(bc.SRC).read())
    src = open(header).read()
    for arm, kind in (('UR5', 'U'), ('PANDA', 'P')):
        m = re.search(r'static const float rp_hplane_%s\[(\d+)\]\[4\] = \{(.*?)\};' % arm, src, re.S)
        pl = np.array([[float(x.rstrip('f')) for x in v.split(',')] for v in re.findall(r'\{([^{}]*)\}', m.group(2))])
        poff = [int(x) for x in re.search(r'rp_hplane_off_%s\[64\] = \{([^}]*)\}' % kind, src).group(1).split(',')]
        pcnt = [int(x) for x in re.search(r'rp_hplane_cnt_%s\[64\] = \{([^}]*)\}' % kind, src).group(1).split(',')]
        V, off, cnt = table('rp_hullv_%s' % arm), ints('rp_hull_off_%s' % kind), ints('rp_hull_cnt_%s' % kind)
        for c in range(64):
            if cnt[c] == 0:
                assert pcnt[c] == 0
                continue
            P = V[off[c]:off[c] + cnt[c]].astype(np.float64)
            q = pl[poff[c]:poff[c] + pcnt[c]]
            assert pcnt[c] >= 4 and np.abs(np.linalg.norm(q[:, :3], axis=1) - 1).max() < 1e-5
            side = P @ q[:, :3].T + q[:, 3]                      # [vertex, plane]
            assert side.max() < 2e-6, (arm, c, side.max())       # inside or on every plane
            assert (np.abs(side) < 2e-6).sum(axis=0).min() >= 3  # every plane passes through at least three vertices
