"""The playroom's articulated fixtures driven on the MI355X against the CPU oracle (scenes.py:117-426, environments.py:469-483,
767-793, 868-894): the drawer pulled and pushed to both stops and the door slid by the arm, the button pressed by a dropped block,
non-rest door / button / dial / drawer values through calc_state and the reward, the quaternion sign memory.  Run with -m gpu.

Contact-rich rollouts are compared with the fp32 oracle step by step; where the fp32 and the fp64 CPU oracle themselves drift apart
(a pushed body is sensitive to rounding), the device - a third evaluation order of the same fp32 arithmetic - is held to three
times their gap."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
from tolerances import Followers, obs_atol  # noqa: E402

pytestmark = pytest.mark.gpu
U = 'UR5PlayAbsRPY1Obj-v0'
FREE0, JQ = 24, 50                      # VecPlayEnv.STATE_LAYOUT: free0 (block), jq (door, button, dial); free1 = drawer at 37
EXTRA = {}                              # id(fp32 oracle of an env) -> its nudged fp32 companions (make / drive)


def drive(env, oracles32, oracles64, script, atol=1e-3, check=None, kind='U', loose=None, branch=None):
    """script: list of (target xyz, grip, steps); every env gets the same commands"""
    n = env.num_envs
    worst = 0.0
    last = None
    branched = np.zeros(n, bool)        # branch(previous fp32 obs, fp32 obs) said so: from then on the env is judged by what the caller checks afterwards
    prev = [None] * n
    for target, grip, steps in script:
        a = np.array(list(target) + [0, 0, 0, grip], dtype=np.float64)
        for _ in range(steps):
            obs, r, _, info = env.step(torch.tensor(np.tile(a, (n, 1)), dtype=torch.float32))
            got = obs['obs_quat'].cpu().numpy()
            for e in range(n):
                o32 = oracles32[e].step(a)[0]['obs_quat']
                o64 = oracles64[e].step(a)[0]['obs_quat']
                gap = np.abs(o32 - o64)
                for x in EXTRA.get(id(oracles32[e]), ()):           # four more fp32 runs started 1e-5 .. 2e-5 off (tolerances.Followers): a single-point contact pushing
                    gap = np.maximum(gap, np.abs(x.step(a)[0]['obs_quat'] - o64))      # a 0.1 kg dial is as sensitive to the evaluation order as anything here
                tol = np.maximum(obs_atol(kind, len(o32), atol), 3 * gap)      # (the gripper entry: tests/tolerances.py)
                for k, v in (loose or {}).items():
                    tol[k] = max(tol[k], v)
                if branch is not None and prev[e] is not None and branch(prev[e], o32):
                    branched[e] = True
                prev[e] = o32
                if branched[e]:
                    continue
                err = np.abs(got[e] - o32)
                assert (err <= tol).all(), 'env %d: err %s tol %s' % (e, err, tol)
                worst = max(worst, float((err / tol).max()))
            assert int((info['status'] & 1).sum()) == 0
        last = obs
        if check:
            check(target, last)
    return last, worst


def make(n, seed):
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    env = VecPlayEnv(U, n, seed=seed)
    obs = env.reset()
    fol = [Followers('U', seed, e, extra=4) for e in range(n)]
    o32, o64 = [f.o32 for f in fol], [f.o64 for f in fol]
    for e in range(n):
        a = fol[e].reset()[0]
        EXTRA[id(o32[e])] = fol[e].more
        assert (np.abs(obs['obs_quat'][e].cpu().numpy() - a['obs_quat']) <= obs_atol('U', len(a['obs_quat']), 1e-4, rest=True)).all()
    return env, o32, o64


def test_drawer_pulled_and_pushed_to_both_stops():
    """the closed gripper goes into the drawer's handle hole, pulls it against the two front stops (-0.06 m), pushes it against the back
    stop (+0.075 m): obs_quat[15] = drawer base y (environments.py:783), every step against the oracle"""
    env, o32, o64 = make(3, 5)
    seen = {}

    def note(target, obs):
        seen[target] = obs['obs_quat'][:, 15].cpu().numpy().copy()
    script = [((-0.13, -0.165, 0.10), 1.0, 40), ((-0.13, -0.165, -0.05), 1.0, 40), ((-0.13, -0.30, -0.05), 1.0, 60),
              ((-0.13, -0.02, -0.05), 1.0, 80)]
    obs, worst = drive(env, o32, o64, script, check=note)
    pulled, pushed = seen[(-0.13, -0.30, -0.05)], seen[(-0.13, -0.02, -0.05)]
    assert (pulled < -0.055).all() and (pulled > -0.08).all(), pulled          # against the front stops
    assert (pushed > 0.07).all() and (pushed < 0.095).all(), pushed            # against the back stop
    rc = env.debug_row_counts()
    assert int(rc[:, 1].max()) >= 4                                             # contacts were there to be solved
    print('drawer scenario: worst error / tolerance = %.2f' % worst)


def test_door_slid_by_the_arm():
    """a finger beside the door's loop handle pushes it along world x (door joint: prismatic, scenes.py:117-182) by more than the reward's
    0.04 tolerance, then back: obs_quat[16] = door joint position"""
    env, o32, o64 = make(3, 6)
    seen = {}

    def note(target, obs):
        seen[target] = obs['obs_quat'][:, 16].cpu().numpy().copy()
    script = [((-0.06, 0.322, 0.15), 1.0, 40), ((-0.06, 0.322, 0.10), 1.0, 30), ((0.15, 0.322, 0.10), 1.0, 80),
              ((0.15, 0.322, 0.2), 1.0, 20), ((0.28, 0.322, 0.2), 1.0, 20), ((0.28, 0.322, 0.10), 1.0, 30), ((0.05, 0.322, 0.10), 1.0, 70)]
    # The finger travels back to the handle at the per-step clip (0.1 - 0.2 rad a step at half a metre: 4 - 8 mm per SUBSTEP) and where the door comes to rest afterwards hangs
    # on the substep in which it touches: two evaluation orders of the same arithmetic part by up to a substep's travel there, which the nudged CPU followers - the
    # oracle's own evaluation order - do not draw (measured in round 5, tools/dbg_door.py: joint targets and arm agree to 1.2e-7 for 222 steps, the contact of step 223
    # leaves the door 1.8 mm elsewhere for good).  The door's entry gets half a substep's travel; the arm pressed against the handle follows it (with the penetration contacts of
    # RPO_RULE_EPA in the model the end effector's pose parts by 1.5 - 2.3 mm from step 228 on, where the fp32 and fp64 oracles' doors are 2.3 - 4.7 mm apart): 3 mm.
    obs, worst = drive(env, o32, o64, script, atol=3e-3, check=note, loose={16: 4e-3})
    out, back = seen[(0.15, 0.322, 0.10)], seen[(0.05, 0.322, 0.10)]
    assert (out > 0.1).all(), out
    assert (back < out - 0.05).all(), (out, back)
    print('door scenario: worst error / tolerance = %.2f' % worst)


def test_dial_turned_by_the_closed_gripper():
    """the dial (scenes.py:345-426: a 0.1 kg box link on a revolute joint about world -y at the table's front face, held only by Bullet's default
    velocity motor, max impulse 1 per substep) with the closed gripper pressed down on its rim, 15 mm off its axis: the joint turns by several tenths of
    a radian (the arm's push outweighs the motor), obs_quat[18] = dial_to_0_1_range(q) follows through the wrap of scenes.py:342-343 - every step of
    the device against the fp32 oracle inside the fp32 / fp64 envelope, and all three agree on how far it went"""
    env, o32, o64 = make(3, 8)
    seen = {}

    def note(target, obs):
        seen[target] = obs['obs_quat'][:, 18].cpu().numpy().copy()
    script = [((0.215, -0.055, 0.10), 1.0, 30), ((0.215, -0.055, -0.02), 1.0, 40), ((0.215, -0.055, -0.09), 1.0, 40)]
    # Once the gripper SPINS the dial (15 rad/s through ten spanning contacts - hull vertices of the gripper links on the dial's box since RPO_RULE_HULLMOV -:
    # more than 0.1 rad per step) the rollouts branch: which cached point a candidate replaces is decided at rounding level and two evaluation orders part by
    # 0.05 rad within a step.  From that step on the env is judged by where the dial comes to rest (below).
    obs, worst = drive(env, o32, o64, script, atol=2e-3, check=note, branch=lambda a, b: abs(b[18] - a[18]) > 0.05 and abs(b[18] - a[18]) < 0.4)
    before, after = seen[(0.215, -0.055, -0.02)], seen[(0.215, -0.055, -0.09)]
    assert (before == 0).all(), before                                          # nothing touched it on the way down
    q = env.get_state()[:, JQ + 2].cpu().numpy()
    q32 = np.array([o.get_state()[2 * o.n_arm + 26 + 2] for o in o32])
    assert (np.abs(q) > 0.1).all() and (np.abs(q) < 2.0).all(), q                # turned, not spun
    # where it comes to rest after the branch: TWO rest angles 0.10 rad apart, -1.411 and -1.307, and runs of one arithmetic land on either - measured in round 6: the fp32 oracle
    # -1.411 / -1.411 / -1.307 and the fp64 oracle -1.411 (x3) with the motor row's number in the arm lanes of the residual form; both oracles -1.307 (x3) with s in
    # those lanes (the same sweeps to 1e-14 per step until step 73, where ONE step of the spinning dial - 0.4 rad - parts them by 5e-2: tools/residual_check.py's kind of
    # event); the device -1.411 / -1.307 / -1.410.  Until then this line held the device to the fp32 oracle's angle within 2e-2, which is a statement about that draw.
    np.testing.assert_allclose(q, q32, atol=max(0.15, 3 * float(np.abs(q32 - np.array([o.get_state()[2 * o.n_arm + 26 + 2] for o in o64])).max())))
    want = (q - 2.0 * np.floor(q / 2.0)) / 2.2                                   # dial_to_0_1_range: (q mod 2) / 2.2 with Python's modulo
    np.testing.assert_allclose(after, want, atol=1e-5)
    print('dial scenario: q = %s, worst error / tolerance = %.2f' % (q, worst))


def test_button_pressed_by_a_dropped_block():
    """the block (0.3 kg) dropped on the button (spring = position motor, target 0.03, force 1 N: scenes.py:238) presses it below the
    toggle threshold q < 0.025 (environments.py:474): obs_quat[17] against the oracle through the contact phase"""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    n = 3
    env = VecPlayEnv(U, n, seed=7)
    env.reset()
    o32 = [OracleEnv('U', seed=7, env_index=e, f32=True) for e in range(n)]
    o64 = [OracleEnv('U', seed=7, env_index=e) for e in range(n)]
    s = env.get_state()
    for e in range(n):
        o32[e].reset()
        o64[e].reset()
    s[:, FREE0:FREE0 + 13] = torch.tensor([-0.25, 0.45, 0.09, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=torch.float32)
    env.set_state(s)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import oracle_state_from_record
    rec = s.cpu().numpy()
    for e in range(n):
        for o in (o32[e], o64[e]):
            o.set_state(oracle_state_from_record(o, rec[e]))
    park = ((0.1, 0.1, 0.25), 0.0, 40)                       # the arm stays away
    obs, worst = drive(env, o32, o64, [park], atol=1e-3)
    q = obs['obs_quat'][:, 17].cpu().numpy()
    assert (q < 0.025).all() and (q > -0.04).all(), q        # pressed: the toggle condition of updateToggles
    assert (obs['obs_quat'][:, 10].cpu().numpy() < 0.04).all()      # the block came down with it
    print('button scenario: worst error / tolerance = %.2f' % worst)


def test_panda_pick_grasp_and_lift():
    """pandaPick-v0 (config C3): open fingers onto the block, close, lift to z = 0.15 (environments.py:915-1073 panda branch; the fingers'
    soft <contact> pads, the finger gear, arm-against-block rows in the solver's folded slots).  Approach and closing (60 steps): every step
    of every env against the fp32 oracle inside the running fp32 / fp64 sensitivity envelope.  The lift itself is chaotic (DESIGN.md
    section 2: the fp32 and the fp64 oracle part ways by centimetres too, and which envs end with the block in the air differs between any two
    runs), so it is judged by its outcome: the device holds the block in the air in as many of the envs as the eight CPU followers
    (tolerances.Followers: fp64, fp32, six nudged fp32 runs) do, give or take ONE (round 5's final build and round 6: the followers 3 - 4 of 6, the device 4 of 6; the
    "device 6 of 6" that once widened this bound to two was measured before the oracle and the device shared one fma convention).  The lift RATE on 64 envs is the next test's;
    the lift's physics is held to the oracle step by step in
    tests/test_gpu_dist_a.py::test_distribution_a_lockstep_with_contact_history[P-12-110-grasp]: from the same state, cache and joint targets every step of it agrees."""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    n, seed = 6, 3
    env = VecPlayEnv('pandaPick-v0', n, seed=seed)
    obs = env.reset()
    fol = [Followers('P', seed, e, extra=6) for e in range(n)]
    ob32 = [f.reset()[0] for f in fol]
    for e in range(n):
        assert (np.abs(obs['obs_quat'][e].cpu().numpy() - ob32[e]['obs_quat']) <= obs_atol('P', 13, 1e-4, rest=True)).all()
    worst, folded = 0.0, 0
    gap = [np.zeros(13) for _ in range(n)]
    kicked = np.zeros(n, bool)
    for t in range(110):
        a = np.zeros((n, 7))
        for e in range(n):
            blk = ob32[e]['achieved_goal'][:3]
            a[e, 0:3] = blk
            a[e, 2] = blk[2] if t < 60 else 0.15
            a[e, 6] = -1.0 if t < 30 else 1.0
        obs, r, _, info = env.step(torch.tensor(a, dtype=torch.float32))
        folded += int((env.debug_row_counts()[:, 3] > 0).sum())
        got = obs['obs_quat'].cpu().numpy()
        for e in range(n):
            r32, r64, allres = fol[e].step(a[e])
            ob32[e] = r32[0]
            if t >= 60:
                continue
            # trajectories that have separated need not meet again: the envelope is the running maximum of the fp32 followers' distance from the fp64 one
            gap[e] = np.maximum(gap[e], np.max([np.abs(x[0]['obs_quat'] - r64[0]['obs_quat']) for x in allres[1:]], axis=0))
            if gap[e][10:13].max() > 0.1:                   # a finger landed on the block's edge and the CPU runs themselves disagree about the kick by > 0.1 m/s:
                kicked[e] = True                            # this env has branched, it is judged by the outcome only
            if kicked[e]:
                continue
            tol = np.maximum(1e-3, 3 * gap[e])
            tol[3:6] = np.maximum(tol[3:6], 1e-2)           # the EE's velocity feels the fingers' limit chatter while they are commanded open past their limits (tests/tolerances.py)
            tol[6] = max(tol[6], obs_atol('P', 13, 1e-3)[6])
            tol[7:10] = np.maximum(tol[7:10], 5e-3)         # the block between the soft pads slides by millimetres between evaluation orders of the same fp32 arithmetic
            tol[10:13] = np.maximum(tol[10:13], 5e-2)       # ... and its velocity (obs_quat[10:13]) jitters by centimetres per second
            err = np.abs(got[e] - ob32[e]['obs_quat'])
            bad = np.where(err > tol)[0]
            assert bad.size == 0, 'step %d env %d: components %s err %s tol %s' % (t, e, bad, err[bad], tol[bad])
            worst = max(worst, float((err[:7] / tol[:7]).max()))
        assert int((info['status'] & 1).sum()) == 0
    zs = np.array([[o.calc_state()['achieved_goal'][2] for o in f.all()] for f in fol])      # [env, follower]
    z_dev = obs['achieved_goal'][:, 2].cpu().numpy()
    lifted = (zs > 0.05).sum(axis=0)                      # per follower: in how many envs it holds the block in the air
    n_dev = int((z_dev > 0.05).sum())
    assert folded > 0, 'the scenario must exercise arm-against-block rows'
    assert kicked.sum() <= n // 3, kicked
    assert lifted.max() >= 1, zs
    assert lifted.min() - 1 <= n_dev <= lifted.max() + 1, (z_dev, zs)
    print('panda pick scenario: worst arm error / tolerance before the lift = %.2f, lifted %d of %d (the eight CPU followers: %s)' % (worst, n_dev, n, lifted))


def test_panda_pick_lift_rate_64_envs():
    """The grasp-and-lift script of the test above on 64 envs: the lift is a chaotic outcome per env, its RATE is not.  Device against the fp32 and the fp64 CPU oracle from
    their own resets (same seeds), every env driven by its own observation as a learner would: the fraction of envs that hold the block above 5 cm at the end, printed for all
    three and the device held between the CPU runs' rates widened by 0.15 (64 draws: a standard deviation of 0.06 at a rate of one half)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    n, seed, steps = 64, 5, 110

    def script(blk, t):
        a = np.zeros(7)
        a[0:3] = blk
        a[2] = blk[2] if t < 60 else 0.15
        a[6] = -1.0 if t < 30 else 1.0
        return a

    env = VecPlayEnv('pandaPick-v0', n, seed=seed)
    obs = env.reset()
    for t in range(steps):
        blk = obs['achieved_goal'][:, :3].cpu().numpy()
        a = np.stack([script(blk[e], t) for e in range(n)])
        obs, r, _, info = env.step(torch.tensor(a, dtype=torch.float32))
    assert int((info['status'] & 1).sum()) == 0
    z_dev = obs['achieved_goal'][:, 2].cpu().numpy()

    def cpu(args):
        e, f32 = args
        o = OracleEnv('P', seed=seed, env_index=e, f32=f32)
        ob = o.reset()
        for t in range(steps):
            ob = o.step(script(ob['achieved_goal'][:3], t))[0]
        return ob['achieved_goal'][2]
    with ThreadPoolExecutor(16) as pool:
        z32 = np.array(list(pool.map(cpu, [(e, True) for e in range(n)])))
        z64 = np.array(list(pool.map(cpu, [(e, False) for e in range(n)])))
    rd, r32, r64 = float((z_dev > 0.05).mean()), float((z32 > 0.05).mean()), float((z64 > 0.05).mean())
    print('pandaPick grasp-and-lift, 64 envs: block in the air at the end: device %.2f, fp32 CPU oracle %.2f, fp64 CPU oracle %.2f; the same outcome as the fp32 oracle in %d of 64 envs (fp64 vs fp32: %d)'
          % (rd, r32, r64, int(((z_dev > 0.05) == (z32 > 0.05)).sum()), int(((z64 > 0.05) == (z32 > 0.05)).sum())))
    assert min(r32, r64) > 0.1, 'the script no longer lifts anything'
    assert min(r32, r64) - 0.15 <= rd <= max(r32, r64) + 0.15, (rd, r32, r64)


def test_fixture_values_through_calc_state_and_reward(golden):
    """door / button / dial / drawer at non-rest values (set through rp_set_state): calc_state's environment half (environments.py:767-793,
    dial_to_0_1_range over several turns incl. negative angles) and the reward / success truth table (playRewardFunc.py:16-77) against the
    oracle, whose arithmetic the reference goldens pin (tests/test_oracle_golden.py)"""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    vals = [(0.0, 0.0, 0.03, 0.0), (0.05, 0.12, 0.01, 1.0), (-0.04, -0.1, 0.02, 3.0), (0.07, 0.03, -0.01, -0.5), (0.0, 0.2, 0.03, 7.3),
            (-0.06, -0.15, 0.029, 1.99), (0.02, 0.0, 0.0, 2.01), (0.0, 0.05, 0.03, -3.7)]
    n = len(vals)
    env = VecPlayEnv(U, n, seed=2)
    env.reset()
    s = env.get_state()
    for e, (dy, door, button, dial) in enumerate(vals):
        s[e, 37 + 1] = dy
        s[e, JQ:JQ + 3] = torch.tensor([door, button, dial])
    env.set_state(s)
    obs = env.calc_state()
    torch.cuda.synchronize()
    o = OracleEnv('U', seed=2, env_index=0, f32=True)
    o.reset()
    for e, (dy, door, button, dial) in enumerate(vals):
        want_dial = o.lib.rpo_dial_to_0_1_range(float(np.float32(dial)))
        got = obs['obs_quat'][e, 15:19].cpu().numpy()
        np.testing.assert_allclose(got, [dy, door, button, want_dial], atol=2e-7, rtol=0)
        np.testing.assert_array_equal(obs['achieved_goal'][e, 7:11].cpu().numpy(), got)
        assert 0.0 <= got[3] < 0.9091
    # the dial mapping's golden samples (scenes.py:342-343 = (x mod 2) / 2.2)
    dial_gold = golden('rewards.json')['dial']
    fo = OracleEnv('U', seed=2, env_index=0)
    for d in dial_gold:
        assert fo.lib.rpo_dial_to_0_1_range(float(d['x'])) == pytest.approx(d['y'], abs=1e-12)
    # ... and the device at the golden samples themselves
    m = min(n, len(dial_gold))
    for e in range(m):
        s[e, JQ + 2] = float(dial_gold[e * (len(dial_gold) // m)]['x'])
    env.set_state(s)
    obs2 = env.calc_state()
    for e in range(m):
        assert float(obs2['obs_quat'][e, 18]) == pytest.approx(dial_gold[e * (len(dial_gold) // m)]['y'], abs=2e-7)
    # reward: goal = achieved with one fixture entry moved just inside / outside its tolerance
    ag = obs['achieved_goal'].clone()
    tol = {7: 0.025, 8: 0.04, 9: 0.01, 10: 0.3}
    for idx, t in tol.items():
        for f, want in ((0.9, 0.0), (1.1, -1.0)):
            g = ag.clone()
            g[:, idx] += f * t
            r = env.compute_reward(ag, g).cpu().numpy()
            assert (r == want).all(), (idx, f, r)
            for e in range(n):
                assert o.compute_reward(ag[e].cpu().numpy().astype(np.float64), g[e].cpu().numpy().astype(np.float64)) == want


def test_quaternion_sign_memory_sequence():
    """quaternion_safe_the_obs (environments.py:868-894): a quaternion whose four signs are all opposite to the previous observation's is
    negated, any other is passed through, zeros count as equal signs; the memory follows what was returned.  Block orientation set
    through rp_set_state, observed through calc_state, against the oracle fed the same sequence."""
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import oracle_state_from_record
    q0 = np.array([0.1, -0.2, 0.3, 0.9]); q0 /= np.linalg.norm(q0)
    seq = [q0, -q0, -q0, q0, np.array([0.1, 0.2, -0.3, -0.9]) / np.linalg.norm(q0), -q0, np.array([0.0, -0.2, 0.3, 0.9]), np.array([0.0, 0.2, -0.3, -0.9]),
           np.array([1e-9, 0.2, -0.3, -0.9])]
    env = VecPlayEnv(U, 2, seed=3)
    env.reset()
    o = OracleEnv('U', seed=3, env_index=0, f32=True)
    o.reset()
    env.calc_state()
    flips = 0
    for q in seq:
        s = env.get_state()
        s[0, FREE0 + 3:FREE0 + 7] = torch.tensor(q, dtype=torch.float32)
        env.set_state(s)
        rec = s[0].cpu().numpy()
        o.set_state(oracle_state_from_record(o, rec))
        got = env.calc_state()
        want = o.calc_state()
        g = got['obs_quat'][0, 11:15].cpu().numpy()
        np.testing.assert_allclose(g, want['obs_quat'][11:15], atol=1e-7, rtol=0)
        np.testing.assert_allclose(got['achieved_goal'][0, 3:7].cpu().numpy(), want['achieved_goal'][3:7], atol=1e-7, rtol=0)
        if not np.allclose(g, np.float32(q), atol=1e-7):
            flips += 1
            np.testing.assert_allclose(g, -np.float32(q), atol=1e-7)
    assert flips >= 2           # the sequence exercises both branches
