"""CPU: the oracle under the LITERAL random-action distribution (SURVEY.md 8d distribution A: a ~ U(action_space.low, action_space.high) = U(-6, 6)^6 x U(-1, 1),
environments.py:104-110) - the workload BASELINE.json's metric names.  Targets lie metres outside the workspace, every IK runs out of iterations, the arm slews at the
per-step clip through the furniture: hull scans, GJK with its cached simplices, first-contact insertions into the contact cache and the slot aliasing of the GJK
cache (pair index mod 16) run all the time, where under distribution B they run in a few per cent of the steps.

  * the contact cache's row form (rp_kernels.cuh PMC_*; OracleEnv.get_cache_row / set_cache_row) carries the WHOLE contact history: an fp32 oracle that takes
    another's state and cache row continues bit for bit - the property the GPU lock-step test (tests/test_gpu_dist_a.py) rests on;
  * the CPU twin of the GPU rollout test: the fp32 build against the fp64 build under A, same measure (arm joints, relative), with the caches compared field by field
    while the two are on one trajectory.
The GPU tests of the same name hold the HIP library to the same oracle."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cache_rows  # noqa: E402
from oracle import OracleEnv  # noqa: E402
from tolerances import N_MAIN  # noqa: E402


def actions_a(kind, steps, n, seed):
    """distribution A: uniform over the declared action space (environments.py:104-110: high = [6, 6, 6, 6, 6, 6, 1] for absolute_rpy)"""
    hi = OracleEnv(kind).action_high()
    rng = np.random.default_rng(seed)
    return (2 * rng.random((steps, n, len(hi))) - 1) * hi


@pytest.mark.parametrize('kind', ['U', 'P', 'V'])
def test_cache_row_carries_the_whole_contact_history(kind):
    """30 steps of A, then a second fp32 oracle takes state + cache row (motors are re-commanded by every action) and both run on: same bits, caches included.
    Without the row (set_state alone clears the history) a copy parts from the original in at least one of the envs tried - the test would be vacuous otherwise
    (asserted for the headline id; a Panda's block on its tray regenerates the same four points from nothing)."""
    parted = False
    for ei in (2, 5, 11, 14):
        acts = actions_a(kind, 45, 1, 3 + ei)[:, 0]
        a = OracleEnv(kind, seed=5, env_index=ei, f32=True)
        a.reset()
        for t in range(30):
            a.step(acts[t])
        row = a.get_cache_row()
        assert sum(m['n'] for m in cache_rows.decode(row)['manifolds']) > 0, 'the rollout has no contact history to carry'
        b = OracleEnv(kind, seed=5, env_index=ei, f32=True)
        c = OracleEnv(kind, seed=5, env_index=ei, f32=True)
        for o in (b, c):
            o.reset()
            o.step(acts[0])                    # (motor modes: every action re-commands every motor, environments.py:1010-1073)
            o.set_state(a.get_state())
        b.set_cache_row(row)
        assert np.array_equal(b.get_cache_row().view(np.int32), row.view(np.int32))
        for t in range(30, 45):
            a.step(acts[t]); b.step(acts[t]); c.step(acts[t])
            assert np.array_equal(a.get_state(), b.get_state()), 'env %d step %d: the copy with the cache row left the original' % (ei, t)
            assert np.array_equal(a.get_cache_row().view(np.int32), b.get_cache_row().view(np.int32)), 'env %d step %d: caches differ' % (ei, t)
            parted |= not np.array_equal(a.get_state(), c.get_state())
    assert parted or kind != 'U', 'copies WITHOUT the contact history follow the originals bit for bit: the scenario does not exercise the cache'


def positions(o, s):
    """the position-level part of an OracleEnv.get_state vector: arm joints, free bodies' position and quaternion, scene joints"""
    na, nf = o.n_arm, (len(s) - 2 * o.n_arm) // 13
    nj = (len(s) - 2 * na - 13 * nf) // 2
    idx = list(range(na)) + [2 * na + 13 * k + i for k in range(nf) for i in range(7)] + [2 * na + 13 * nf + i for i in range(nj)]
    return s[idx]


def leave_step(trace, bound):
    """first step (0-based) at which a divergence trace exceeds `bound`; len(trace) if it never does"""
    over = np.nonzero(np.asarray(trace) > bound)[0]
    return int(over[0]) if over.size else len(trace)


@pytest.mark.parametrize('kind,n,steps', [('U', 16, 100), ('P', 8, 100), ('V', 8, 100)])
def test_fp32_oracle_follows_fp64_oracle_under_distribution_a(kind, n, steps):
    """the CPU twin of test_gpu_dist_a.py::test_distribution_a_rollout_vs_fp64_oracle: how far the SAME algorithm in fp32 drifts from fp64 under A.

    MEASURED (round 5), and the reason the GPU tests under A are built the way they are: the literal random-action rollout is CHAOTIC at rounding level.  The fp32
    build of the oracle leaves its own fp64 build by more than 1e-3 (arm joints, relative) after a median of ~37 steps on the playroom id, every env within 100 - each
    IK runs out of its 4 x 20 iterations (targets metres outside the workspace: the joint targets then hang on the measured joints' last bits) and the arm strikes
    furniture at the per-step clip.  north_star's "<= 1e-3 over 200 steps" is a statement about distribution B (tests/test_gpu_parity.py); under A no fp32
    implementation can meet it against an fp64 run of the same algorithm.  What CAN be held: rounding-level agreement until the first such event (>= 10 steps
    here), the time an env stays within 1e-3 (reported; the GPU test compares the device's with these CPU runs'), and field-by-field equal contact caches while two
    runs still share a trajectory: same manifolds in creation order, same points in slot order, same cached GJK pairs (the cached SIMPLEX of a pair may differ between
    fp32 and fp64 - another triangulation of the same features, another of several separating simplices of a pair that is apart: equivalent warm starts; counted)."""
    acts = actions_a(kind, steps, n, 17)
    o64 = [OracleEnv(kind, seed=9, env_index=e) for e in range(n)]
    o32 = [OracleEnv(kind, seed=9, env_index=e, f32=True) for e in range(n)]
    for a, b in zip(o64, o32):
        a.reset(); b.reset()
        b.set_state(a.get_state())
        a.set_state(a.get_state())      # (both start without contact history)
    nm, na = N_MAIN[kind], o64[0].n_arm

    def run(e):
        trace, res = [], []
        for t in range(steps):
            o64[e].step(acts[t, e]); o32[e].step(acts[t, e])
            s64, s32 = o64[e].get_state(), o32[e].get_state()
            trace.append(float((np.abs(s32[:na] - s64[:na]) / np.maximum(1.0, np.abs(s64[:na])))[:nm].max()))
            if np.abs(positions(o64[e], s32) - positions(o64[e], s64)).max() <= 1e-5:
                ra, rb = o64[e].get_cache_row(), o32[e].get_cache_row()
                res.append((cache_rows.manifolds(ra) == cache_rows.manifolds(rb) and cache_rows.gjk_tags(ra) == cache_rows.gjk_tags(rb), cache_rows.features(ra) == cache_rows.features(rb),
                            cache_rows.integers(ra) == cache_rows.integers(rb)))
        return trace, res

    with ThreadPoolExecutor(8) as ex:
        out = list(ex.map(run, range(n)))
    traces = np.array([o[0] for o in out])
    checked, same, feat, strict = sum(len(o[1]) for o in out), sum(sum(r[0] for r in o[1]) for o in out), sum(sum(r[1] for r in o[1]) for o in out), sum(sum(r[2] for r in o[1]) for o in out)
    leave3 = np.array([leave_step(tr, 1e-3) for tr in traces]); leave5 = np.array([leave_step(tr, 1e-5) for tr in traces])
    print('%s under distribution A, %d envs x %d steps, fp32 vs fp64 oracle (arm joints): within 1e-5 for a median of %d steps (min %d), within 1e-3 for a median of %d steps '
          '(min %d, %d envs to the end); final divergence median %.1e; caches on a shared trajectory (positions within 1e-5), %d checks: same manifolds, points and cached GJK pairs in %d, also the same simplex features in %d, the very same simplices in %d'
          % (kind, n, steps, np.median(leave5), leave5.min(), np.median(leave3), leave3.min(), int((leave3 == steps).sum()), np.median(traces[:, -1]), checked, same, feat, strict))
    # rounding level until an env's first event: the playroom arm starts clear of everything (every env, eight steps); the Panda's one 200-iteration IK call on an
    # unreachable target is itself rounding-sensitive at the 1e-4 level from the first step on (status bit 8 on the device): a third of the envs
    if kind == 'U':
        assert traces[:, :8].max() <= 1e-5, traces[:, :8].max(axis=1)
    else:
        assert (traces[:, :3].max(axis=1) <= 1e-5).sum() >= n // 3, traces[:, :3].max(axis=1)
    assert np.median(leave3) >= {'U': 20, 'P': 5, 'V': 5}[kind], leave3             # measured: see the printed line (U 37, P 9.5)
    assert checked >= 3 * n and same >= checked - max(2, checked // 20), (same, checked)
