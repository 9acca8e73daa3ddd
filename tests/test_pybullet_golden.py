"""Consumes PyBullet fixtures written by tools/pybullet_replay.py --dump (the reference's own env classes run on a real PyBullet) when
any are present under tests/golden/pybullet_*.json: the one way this repo's physics can be PINNED to the reference.  None is committed -
PyBullet exists neither in the build image nor on the GPU boxes - so the pinning tests skip and say so; the pipeline itself (fixture
format, state hand-over, replay, comparison) is exercised with a fixture written from the CPU oracle."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from oracle import OracleEnv

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import pybullet_replay  # noqa: E402

FIXTURES = sorted(glob.glob(os.path.join(REPO, 'tests', 'golden', 'pybullet_*.json')))


def replay_on_oracle(fx, **kw):
    """start the oracle from the fixture's initial state, replay its actions; returns per-step relative joint divergence and block error"""
    o = OracleEnv(fx['env'], seed=fx['seed'], env_index=0, **kw)
    o.reset()
    o.set_state(pybullet_replay.state_vector_from_snapshot(o.kind, o.n_arm, fx['initial_state']))
    o.set_goal(np.array(fx['initial_obs']['desired_goal'], dtype=np.float64))
    o.clear_quat_memory()
    o.calc_state()
    rel, blk, tp = [], [], []
    for s in fx['trajectory']:
        _, _, _, info = o.step(np.array(s['action']))
        st = o.get_state()
        q = np.array(s['q'])
        rel.append(float((np.abs(st[:o.n_arm] - q) / np.maximum(1.0, np.abs(q))).max()))
        if 'block_pos' in s:
            blk.append(float(np.abs(st[2 * o.n_arm:2 * o.n_arm + 3] - np.array(s['block_pos'])).max()))
        tp.append(float(np.abs(np.array(info['target_poses']) - np.array(s['target_poses'])).max()))
    return np.array(rel), np.array(blk), np.array(tp)


def check_known_answers(fx):
    """single-hypothesis pins, present only in fixtures written from PyBullet"""
    o = OracleEnv(fx['env'], seed=fx['seed'], env_index=0)
    o.reset()
    na = o.n_arm
    if 'mass_matrix' in fx:                         # H2 / H3: link masses and inertias as Bullet computed them
        st = o.get_state()
        st[:na] = fx['mass_matrix']['q']
        st[na:2 * na] = 0
        o.set_state(st)
        Minv = np.zeros((na, na))
        o.lib.rpo_mass_matrix_inv(o.h, Minv.ctypes.data_as(oracle.C.POINTER(oracle.C.c_double)))
        M = np.linalg.inv(Minv)
        ref = np.array(fx['mass_matrix']['M'])[:na, :na]
        np.testing.assert_allclose(M, ref, rtol=1e-3, atol=1e-6 * np.abs(ref).max(), err_msg='mass matrix at the recorded pose')
    for pr in fx.get('ik_probes', []):              # H9: damping, step cap, residual rule
        st = o.get_state()
        st[:na] = pr['q']
        o.set_state(st)
        for key, iters in (('result_1_iteration', 1), ('result_default', 20)):
            got = o.ik(np.array(pr['target_pos']), np.array(pr['target_orn']), np.array(pr['q']), iters)
            np.testing.assert_allclose(got[:len(pr[key])][:6], np.array(pr[key])[:6], atol=1e-4, err_msg='calculateInverseKinematics, %s' % key)


@pytest.mark.skipif(bool(FIXTURES), reason='PyBullet fixtures are present: the pinning tests below run')
def test_parity_is_unpinned_without_a_pybullet_fixture():
    pytest.skip('PARITY UNPINNED: no tests/golden/pybullet_*.json - run tools/pybullet_replay.py --dump on a machine with PyBullet and the '
                'reference repo, commit the fixture, and these tests hold the oracle and the HIP library to it')


@pytest.mark.parametrize('path', FIXTURES)
def test_oracle_against_the_pybullet_fixture(path):
    fx = json.load(open(path))
    assert fx['format'] == pybullet_replay.FORMAT and fx['source'] == 'pybullet'
    check_known_answers(fx)
    rel, blk, tp = replay_on_oracle(fx)
    relB, blkB, _ = replay_on_oracle(fx, bullet_ref=True)
    print('%s: fast model vs PyBullet: joints max %.3e block max %.3e target poses max %.3e | reference step: joints max %.3e block max %.3e'
          % (os.path.basename(path), rel.max(), blk.max() if len(blk) else 0, tp.max(), relB.max(), blkB.max() if len(blkB) else 0))
    assert rel.max() <= 1e-3, 'north_star: <= 1e-3 relative joint-state divergence over %d steps' % len(rel)


@pytest.mark.gpu
@pytest.mark.parametrize('path', FIXTURES)
def test_hip_library_against_the_pybullet_fixture(path):
    torch = pytest.importorskip('torch')
    from gpu_debug import record_from_oracle
    from roboticsplayroompybullet_amd import VecPlayEnv
    fx = json.load(open(path))
    o = OracleEnv(fx['env'], seed=fx['seed'], env_index=0)
    o.reset()
    o.set_state(pybullet_replay.state_vector_from_snapshot(o.kind, o.n_arm, fx['initial_state']))
    o.set_goal(np.array(fx['initial_obs']['desired_goal'], dtype=np.float64))
    env = VecPlayEnv(fx['env'], 1, seed=fx['seed'])
    env.reset()
    env.set_state(torch.tensor(record_from_oracle(o)[None]))
    worst = 0.0
    for s in fx['trajectory']:
        env.step(torch.tensor(np.array(s['action'])[None], dtype=torch.float32))
        q = env.get_state()[0, :o.n_arm].cpu().numpy()
        ref = np.array(s['q'])
        worst = max(worst, float((np.abs(q - ref) / np.maximum(1.0, np.abs(ref))).max()))
    assert worst <= 1e-3, worst


def test_fixture_pipeline_with_an_oracle_written_fixture(tmp_path):
    """the whole path - dump, state hand-over, replay, comparison - on a fixture the CPU oracle wrote (source = "oracle": a format
    check, never a pin): the replay reproduces it to rounding, and the frozen reference step can be replayed from the same file"""
    for env_id, scenario in (('UR5PlayAbsRPY1Obj-v0', 'grasp'), ('pandaPick-v0', 'random')):
        path = str(tmp_path / ('oracle_%s.json' % env_id))
        subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pybullet_replay.py'), '--from-oracle', '--env', env_id, '--steps', '30',
                        '--seed', '3', '--scenario', scenario, '--dump', path], check=True, capture_output=True)
        fx = json.load(open(path))
        assert fx['source'] == 'oracle' and fx['format'] == pybullet_replay.FORMAT and len(fx['trajectory']) == 30
        rel, blk, tp = replay_on_oracle(fx)
        assert rel.max() < 1e-9 and tp.max() < 1e-9 and (len(blk) == 0 or blk.max() < 1e-9)
        relB, _, _ = replay_on_oracle(fx, bullet_ref=True)
        assert np.isfinite(relB).all()


FAKE_RUN = r'''
import json, os, sys, types
sys.path.insert(0, os.path.join(%(repo)r, 'tests', 'golden')); sys.path.insert(0, os.path.join(%(repo)r, 'tools'))
import fake_bullet
gold = json.load(open(os.path.join(%(repo)r, 'tests', 'golden', 'assets_independent.json')))
def joint_types(model):       # Bullet's joint order: depth-first, children in XML order (pybullet: 0 revolute, 1 prismatic, 4 fixed)
    out = []
    def visit(link):
        for j in model['joints']:
            if j['parent'] == link:
                out.append({'revolute': 0, 'continuous': 0, 'prismatic': 1}.get(j['type'], 4)); visit(j['child'])
    visit(model['root'])
    return out
fake_bullet.UR5_JOINT_TYPES, fake_bullet.PANDA_JOINT_TYPES = joint_types(gold['ur5']), joint_types(gold['panda'])
clients = []
fake_bullet.install_stubs(clients, fake_bullet.StaticWorldClient)
import pybullet_replay
args = types.SimpleNamespace(env=%(env)r, steps=4, seed=0, scenario=%(scenario)r, reference_root='/root/reference', dump=%(out)r, source_label='fake')
pybullet_replay.dump_from_pybullet(args)
'''


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference repo (build container only)')
@pytest.mark.parametrize('env_id,scenario', [('UR5PlayAbsRPY1Obj-v0', 'random'), ('pandaPick-v0', 'grasp')])
def test_the_pybullet_half_of_the_replay_tool_runs_end_to_end_on_a_fake_client(env_id, scenario, tmp_path):
    """tools/pybullet_replay.py --dump has to work on the first try on the one machine with a real PyBullet: here its PyBullet half drives the
    REFERENCE'S OWN env class against tests/golden/fake_bullet.StaticWorldClient (no physics: format only, `source` = "fake", never a pin) and
    the fixture it writes has every section the consumers read"""
    out = str(tmp_path / 'fx.json')
    r = subprocess.run([sys.executable, '-c', FAKE_RUN % {'repo': REPO, 'env': env_id, 'scenario': scenario, 'out': out}], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    fx = json.load(open(out))
    assert fx['source'] == 'fake' and fx['format'] == pybullet_replay.FORMAT and fx['env'] == env_id
    n = 12 if env_id.startswith('UR5') else 9
    for key in ('joint_info', 'dynamics_info', 'mass_matrix', 'ik_probes', 'link_states_at_reset', 'contact_points_at_reset', 'physics_engine_parameters',
                'pybullet_api_version', 'initial_state', 'initial_obs', 'trajectory'):
        assert key in fx, key
    assert len(fx['trajectory']) == 4 and len(fx['initial_state']['q']) == n and len(fx['mass_matrix']['M']) == n
    assert len(fx['dynamics_info']) == len(fx['joint_info']) + 1
    s = fx['trajectory'][-1]
    assert {'q', 'qd', 'action', 'obs_quat', 'reward', 'target_poses', 'block_pos'} <= set(s)
    if env_id.startswith('UR5Play'):
        assert len(s['scene_joints']) == 3 and len(s['drawer_pos']) == 3
    # the consumers' state hand-over accepts it
    o = OracleEnv(env_id, seed=0)
    v = pybullet_replay.state_vector_from_snapshot(o.kind, o.n_arm, fx['initial_state'])
    assert v.shape == (o.lib.rpo_state_size(o.h),)
