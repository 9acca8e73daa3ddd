#!/usr/bin/env python3
"""CPU: what the arm lanes of the residual form carry, measured (round 6).  Env-steps of the literal random-action distribution A (far targets: motor rows with
|rhs| ~ 1e3) are replayed ONE STEP from identical states and contact caches by the fp64 and the fp32 oracle; the gap of the arm's joint velocities is the fp32 error of that
step.  Two populations: the env-steps that take the residual form in the shipped model (a coupled or crowded env), and - RPO_FORCE_RESIDUAL=1 - every env-step in the form.  With the motor
row's number in an arm dof's lane (RPO_NO_SLANE=1: the form until the middle of round 6) the dof's limit rows read (rhs_motor + s) + (rhs_limit - rhs_motor) and lose three
digits; with s = -Jd . dv in the lane (the default) every row adds its own rhs.  Prints both distributions (the other mode in a child process: the switch is read once).
    python tools/slane_check.py [envs=12] [steps=60]"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))


def gaps(n, steps):
    from oracle import OracleEnv
    out = []
    for e in range(n):
        a = OracleEnv('U', seed=9, env_index=e)
        b = OracleEnv('U', seed=9, env_index=e, f32=True)
        a.reset(); b.reset()
        rng = np.random.default_rng(300 + e)
        hi = np.array([6.0] * 6 + [1.0])
        na = a.n_arm
        for t in range(steps):
            act = rng.uniform(-hi, hi)
            s = a.get_state()
            row = a.get_cache_row()
            b.set_state(s); b.set_cache_row(row)             # both from the fp64 run's state and contact cache
            r0 = a.lib.rpo_residual_substeps(a.h)
            a.step(act); b.step(act)
            heavy = a.lib.rpo_residual_substeps(a.h) - r0
            if heavy > 0 and (os.environ.get('RPO_FORCE_RESIDUAL') is None or True):
                sa, sb = a.get_state(), b.get_state()
                out.append((heavy, float(np.abs(sa[na:2 * na] - sb[na:2 * na])[:6].max()), float(np.abs(sa[:na] - sb[:na])[:6].max())))
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        print(json.dumps(gaps(int(sys.argv[2]), int(sys.argv[3]))))
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    res = {}
    for mode, env in (('heavy env-steps, s in the arm lanes (default)', {}), ('heavy env-steps, the motor row\'s number in the lanes (RPO_NO_SLANE=1)', {'RPO_NO_SLANE': '1'}),
                      ('every env-step in the form, s in the arm lanes', {'RPO_FORCE_RESIDUAL': '1'}), ('every env-step in the form, the motor row\'s number', {'RPO_FORCE_RESIDUAL': '1', 'RPO_NO_SLANE': '1'})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(n), str(steps)], env=dict(os.environ, **env), capture_output=True, text=True, check=True)
        g = np.array(json.loads(p.stdout.strip().splitlines()[-1]))
        res[mode] = g
        print('%-78s %4d env-steps of U under A: fp32 against fp64 after ONE step, velocities of the six arm joints (the light links of the gripper chatter at their limits in either mode): median %.1e p90 %.1e p99 %.1e max %.1e rad/s; beyond 1e-2 in %d; joints: max %.1e rad'
              % (mode + ':', len(g), np.median(g[:, 1]), np.quantile(g[:, 1], 0.9), np.quantile(g[:, 1], 0.99), g[:, 1].max(), int((g[:, 1] > 1e-2).sum()), g[:, 2].max()))
    return res


if __name__ == '__main__':
    main()
