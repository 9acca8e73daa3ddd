"""C-ABI boundary on the MI355X: rp_config kwargs, rp_out.pack, several handles in one process, the N = 32768 shard layout of
BASELINE.json's config C4 on one GPU, error paths.  Run with -m gpu."""
import ctypes as C

import numpy as np
from tolerances import obs_atol
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

U = 'UR5PlayAbsRPY1Obj-v0'
LO = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0])
HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])


def acts(steps, n, seed):
    rng = np.random.default_rng(seed)
    return torch.tensor(LO + (HI - LO) * rng.random((steps, n, 7)), dtype=torch.float32)


def test_pack_output_is_the_gather_message_bitwise():
    """rp_out.pack == cat(obs_quat, achieved_goal, reward, is_success) (what sharding.pack_observations builds), written by
    k_calc_state itself, for step, reset and calc_state; alternates between two buffers"""
    from roboticsplayroompybullet_amd import VecPlayEnv, sharding
    for gid in (U, 'pandaPick-v0', 'pandaPlay-v0'):
        env = VecPlayEnv(gid, 37, seed=3)
        obs = env.reset()
        a = torch.zeros((37, env.dims['action']))
        a[:, :3] = torch.tensor([0.0, 0.1, 0.2])
        prev = None
        for t in range(3):
            obs, r, _, info = env.step(a)
            torch.cuda.synchronize()
            want = sharding.pack_observations(obs, r, info['is_success'])
            assert torch.equal(env.pack, want), gid
            assert prev is None or env.pack.data_ptr() != prev
            prev = env.pack.data_ptr()
        u = sharding.unpack_observations(env.pack, env.dims['obs_quat'], env.dims['achieved_goal'])
        assert torch.equal(u['obs_quat'], obs['obs_quat']) and torch.equal(u['is_success'], info['is_success'])


def test_every_writer_of_the_pack_takes_the_other_buffer():
    """step, reset(mask), calc_state all write rp_out.pack; an asynchronous all-gather of the previous pack may still be reading it on RCCL's
    stream (sharding.gather_observations(async_op=True): the auto-reset loop is step -> async gather -> masked reset), so every writer gets the
    OTHER of the two buffers and the one handed to the gather stays as it was; a masked reset carries the rows it does not rewrite over."""
    import torch
    from roboticsplayroompybullet_amd import VecPlayEnv
    n = 8
    env = VecPlayEnv('UR5PlayAbsRPY1Obj-v0', n, seed=3)
    env.reset()
    a = torch.zeros((n, 7)); a[:, :3] = torch.tensor([0.0, 0.15, 0.2])
    obs, r, _, info = env.step(a)
    sent = env.pack                         # what a gather of this step would be reading
    want = sent.clone()
    mask = torch.zeros(n, dtype=torch.uint8); mask[[1, 6]] = 1
    ob2 = env.reset(mask=mask)
    torch.cuda.synchronize()
    assert env.pack.data_ptr() != sent.data_ptr()
    assert torch.equal(sent, want)          # untouched by the reset
    keep = (mask == 0).to(sent.device)
    w = env.dims['obs_quat']
    assert torch.equal(env.pack[keep], want[keep])                                   # rows of the envs that were not reset: carried over
    assert torch.equal(env.pack[~keep][:, :w], ob2['obs_quat'][~keep])               # rows of the reset envs: their fresh observation
    assert not torch.equal(env.pack[~keep], want[~keep])
    sent2, want2 = env.pack, env.pack.clone()
    env.calc_state()
    torch.cuda.synchronize()
    assert env.pack.data_ptr() != sent2.data_ptr() and torch.equal(sent2, want2)
    sent3, want3 = env.pack, env.pack.clone()
    env.reset()                             # unmasked: every row rewritten, still the other buffer
    torch.cuda.synchronize()
    assert env.pack.data_ptr() != sent3.data_ptr() and torch.equal(sent3, want3)


def test_two_handles_one_process_distinct_streams():
    """two handles driven alternately on their own torch streams (and with another device-current state around the calls) ==
    the same two handles driven one after the other: handles share nothing"""
    from roboticsplayroompybullet_amd import VecPlayEnv
    n = 65
    a = acts(6, n, 1)
    ref = []
    for seed in (5, 6):
        e = VecPlayEnv(U, n, seed=seed)
        e.reset()
        for t in range(6):
            e.step(a[t])
        torch.cuda.synchronize()
        ref.append(e.get_state().clone())
        e.close()
    e1, e2 = VecPlayEnv(U, n, seed=5), VecPlayEnv(U, n, seed=6)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ad = a.cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        e1.reset()
    with torch.cuda.stream(s2):
        e2.reset()
    for t in range(6):
        with torch.cuda.stream(s1):
            e1.step(ad[t])
        with torch.cuda.stream(s2):
            e2.step(ad[t])
    torch.cuda.synchronize()
    assert torch.equal(e1.get_state(), ref[0]) and torch.equal(e2.get_state(), ref[1])


def test_config_c4_32768_envs_equals_eight_shards_of_4096():
    """BASELINE.json config C4 on one GPU: one handle of N = 32768 == 8 handles of 4096 with env_offset = 4096 r (what the 8 ranks
    of the multi-GPU run hold), compared on the rows of each shard's first, last and a few sampled envs, bit for bit"""
    from roboticsplayroompybullet_amd import VecPlayEnv
    n, shards, steps = 4096, 8, 4
    full = VecPlayEnv(U, n * shards, seed=11)
    full.reset()
    a = acts(steps, n * shards, 2).cuda()
    for t in range(steps):
        full.step(a[t])
    torch.cuda.synchronize()
    fs, fp = full.get_state(), full.pack.clone()
    assert int((full.buf['status'] & 1).sum()) == 0
    rows = torch.tensor([0, 1, 63, 64, 1000, 2047, 4095])
    for r in range(shards):
        sh = VecPlayEnv(U, n, seed=11, env_offset=n * r)
        sh.reset()
        for t in range(steps):
            sh.step(a[t, n * r:n * (r + 1)])
        torch.cuda.synchronize()
        assert torch.equal(sh.get_state()[rows], fs[n * r + rows]), 'shard %d' % r
        assert torch.equal(sh.pack[rows], fp[n * r + rows])
        sh.close()


def test_constructor_kwargs_reach_the_device():
    """envList.py kwargs through rp_config: goal / object ranges move the draws, sparse_rew_thresh and sparse=False change the
    reward, action_type changes the action width - each against the oracle given the same kwargs"""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    n = 8
    gl, gh = [-0.05, -0.05, 0.02], [0.05, 0.05, 0.04]
    ol, oh = [-0.02, -0.02, 0.0], [0.02, 0.02, 0.01]
    gl, gh = [-0.05, -0.05, 0.12], [0.05, 0.05, 0.14]             # goals (and the arm's reset target) well above the block
    env = VecPlayEnv('pandaPick-v0', n, seed=4, goal_range_low=gl, goal_range_high=gh, obj_lower_bound=ol, obj_upper_bound=oh,
                     env_range_high=[0.18, 0.18, 0.2])
    obs = env.reset()
    dg = obs['desired_goal'].cpu().numpy()
    assert (dg >= np.float32(gl) - 1e-6).all() and (dg <= np.float32(gh) + 1e-6).all()
    blk = obs['achieved_goal'].cpu().numpy()
    assert (np.abs(blk[:, :2]) < 0.03).all()                      # spawned in the small object range, settled where it fell
    for e in range(n):
        o = OracleEnv('P', seed=4, env_index=e, f32=True, ranges=(gl, gh, ol, oh, [0.18, 0.18, 0.2]))
        oo = o.reset()
        for k in ('obs_quat', 'desired_goal'):
            np.testing.assert_allclose(obs[k][e].cpu().numpy(), oo[k], atol=1e-4, rtol=0, err_msg=k)
    # reward threshold 0.2: distances between 0.05 and 0.2 now give -distance
    env = VecPlayEnv('pandaPick-v0', n, seed=4, sparse_rew_thresh=0.2)
    ag = torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    g2 = torch.tensor([[0.1, 0.0, 0.0], [0.3, 0.0, 0.0]])
    r = env.compute_reward(ag, g2).cpu().numpy()
    np.testing.assert_allclose(r, [-0.1, -1.0], atol=1e-7)
    orc = OracleEnv('P', seed=4, env_index=0, sparse_rew_thresh=0.2)
    assert orc.compute_reward(np.zeros(3), np.array([0.1, 0, 0])) == pytest.approx(-0.1)
    # dense reward (sparse=False): -||ag - dg|| on every id, over the whole goal vector for the play ids (environments.py:269-275)
    dense = VecPlayEnv(U, n, seed=4, sparse=False)
    obs = dense.reset()
    # a dense env keeps the first draw of its reset (the reference's `while r > -1` would not end: INTEGRATION.md): rounds = object re-samples only, not 64 attempts
    assert dense.lib.rp_debug_reset_rounds(dense.h) <= 9
    o_d = OracleEnv('U', seed=4, env_index=0, f32=True, dense_reward=True).reset()
    np.testing.assert_allclose(obs['obs_quat'][0].cpu().numpy()[:7], o_d['obs_quat'][:7], atol=1e-4, rtol=0)
    o2, r2, _, info = dense.step(acts(1, n, 0)[0])
    want = -torch.linalg.vector_norm(o2['achieved_goal'] - o2['desired_goal'], dim=1)
    torch.testing.assert_close(r2, want, atol=1e-6, rtol=1e-6)
    assert torch.equal(info['is_success'], (r2 >= 0).to(torch.int32))
    orc = OracleEnv('U', seed=4, env_index=0, f32=True, dense_reward=True)
    orc.reset()
    assert orc.compute_reward(np.zeros(11), np.full(11, 0.5)) == pytest.approx(-np.sqrt(11 * 0.25))
    # another action type on an id that registers absolute_rpy
    rel = VecPlayEnv('UR5Reach-v0', n, seed=4, action_type='relative_joints')
    assert rel.dims['action'] == 7 and rel.action_type == 'relative_joints'
    rel.reset()
    q0 = rel.get_state()[:, :6].clone()
    da = torch.zeros((n, 7))
    da[:, 0] = 0.05
    _, _, _, info = rel.step(da)
    torch.testing.assert_close(info['target_poses'][:, 0], q0[:, 0] + 0.05, atol=1e-6, rtol=0)


def test_single_env_adapter_honours_kwargs_and_refuses_layout_changes():
    import roboticsplayroompybullet_amd as rp
    env = rp.make('pandaPick-v0', goal_range_low=(-0.05, -0.05, 0.02), goal_range_high=(0.05, 0.05, 0.04), seed=9)
    o = env.reset()
    assert (o['desired_goal'] >= np.float32([-0.05, -0.05, 0.02]) - 1e-6).all() and (o['desired_goal'] <= np.float32([0.05, 0.05, 0.04]) + 1e-6).all()
    env.close()
    dense = rp.make('UR5Reach-v0', sparse=False, seed=9)
    o = dense.reset()
    o, r, _, info = dense.step(np.array([0.0, 0.0, 0.1, 0, 0, 0, 0]))
    assert r == pytest.approx(-np.linalg.norm(o['achieved_goal'] - o['desired_goal']), abs=1e-6)
    # compute_reward_sparse stays the sparse formula on a dense env (the reference rebinds only compute_reward, environments.py:169-170)
    assert dense.compute_reward_sparse(np.zeros(3), np.array([0.2, 0, 0])) == -1.0
    assert dense.compute_reward_sparse(np.zeros(3), np.array([0.03, 0, 0])) == pytest.approx(-0.03, abs=1e-7)
    assert dense.compute_reward(np.zeros(3), np.array([0.2, 0, 0])) == pytest.approx(-0.2, abs=1e-7)
    dense.close()
    bad = rp.make('UR5Reach-v0', num_objects=1)
    with pytest.raises(NotImplementedError, match='cannot be overridden'):
        bad.reset()
    # unseeded envs draw their seed from the global numpy RNG, like the reference's np.random calls
    np.random.seed(123)
    a = rp.make('UR5Reach-v0').reset()['desired_goal']
    b = rp.make('UR5Reach-v0').reset()['desired_goal']
    np.random.seed(123)
    c = rp.make('UR5Reach-v0').reset()['desired_goal']
    assert not np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize('margin', [0.0, 0.005, 0.02, 0.05])      # 0.05: every cap (64 pairs, 64 candidate points, 21 contacts) is hit
def test_contact_margin_is_a_parameter_shared_with_the_oracle(margin):
    """rp_config.contact_margin: reset (100 settle substeps) and a grasp-like rollout at each margin, device vs the fp32 oracle at
    the same margin"""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    n = 6
    env = VecPlayEnv(U, n, seed=8, contact_margin=margin)
    obs = env.reset()
    oracles = [OracleEnv('U', seed=8, env_index=e, f32=True, margin=margin) for e in range(n)]
    for e, o in enumerate(oracles):
        oo = o.reset()
        assert (np.abs(obs['obs_quat'][e].cpu().numpy() - oo['obs_quat']) <= obs_atol('U', len(oo['obs_quat']), 1e-4, rest=True)).all()      # (the gripper entry: its joints sit AT their limits after a reset, tests/tolerances.py)
    a = acts(6, n, 3)
    a[:, :, 2] = 0.03                                  # low: fingers near the table and the block
    for t in range(6):
        obs, r, _, info = env.step(a[t])
        for e, o in enumerate(oracles):
            oo, _, _, _ = o.step(a[t, e].numpy().astype(np.float64))
            err = np.abs(obs['obs_quat'][e].cpu().numpy() - oo['obs_quat'])      # (the gripper entry: the phase of its limit sawtooth is decided at rounding level, tests/tolerances.py)
            assert (err <= obs_atol('U', len(err), 1e-3)).all(), 'step %d env %d: %s' % (t, e, err)
    with pytest.raises(RuntimeError, match='contact_margin'):
        VecPlayEnv(U, 2, contact_margin=0.5)


def test_status_bits_and_argument_errors():
    from roboticsplayroompybullet_amd import VecPlayEnv, _lib
    env = VecPlayEnv(U, 4, seed=1)
    env.reset()
    s = env.get_state()
    s[2, 24 + 2] = -1.0                              # block of env 2 far below the ground plate (z = -0.27)
    env.set_state(s)
    _, _, _, info = env.step(torch.zeros((4, 7)))
    st = info['status'].cpu().numpy()
    assert st[2] & 2 and not (st[[0, 1, 3]] & 2).any() and not (st & 1).any()
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.RpConfig(0, (1 << 22) + 1, 0, 0, 0)
    assert lib.rp_create(C.byref(cfg), C.byref(h)) == -1 and b'num_envs' in lib.rp_last_error(None)
    cfg = _lib.RpConfig(0, 4, 99, 0, 0)
    assert lib.rp_create(C.byref(cfg), C.byref(h)) in (-1, -2)
    assert lib.rp_debug_reset_rounds(env.h) >= 1
