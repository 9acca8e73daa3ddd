"""The fidelity table of DESIGN.md section 2, held by a test (CPU, fp64): the shipped fast model (oracle/rp_oracle.c with the product's default rule) against the
FROZEN Bullet-like reference step (oracle/rp_bullet_ref.c) on the ids whose arm touches the scene - the headline playroom id and pandaPick - with
tools/model_divergence.py's own rollouts: 12 envs x 200 steps, both models from the reference step's post-reset state, the same random actions
(environments.py:485-490: 12 x stepSimulation per step; runSimulation is what the two models restate).

PARITY UNPINNED like all physics here (PyBullet is absent): the reference step is a recollection of Bullet, the only independent evidence there is.  What the
test pins is that a change to the oracle's contact model cannot walk the table back unnoticed: the bounds sit one notch above the measured values, and switching
off any of rule bits 4 (hull vertices), 256 (persistent manifolds) or 1024 (GJK beside the face) breaks at least one of them."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))

DEFAULT_RULE = 2039      # rp_oracle.c rpo_create: 1 | 2 | 4 | 16 | 32 | 64 | 128 | 256 | 512 | 1024

# measured (tools/fidelity_rows.py, round 4): U arm median 4.2e-4, p75 6.2e-3, 7 of 12 within 1e-3, block median 1.9e-3 m;
#                                             P arm median 1.3e-14, p75 8.8e-9, max 1.2e-3, 11 of 12, block median 1.5e-16 m
BOUNDS = {
    'U': dict(arm_median=6e-4, arm_p75=8e-3, within_1e3=7, block_median=3e-3),
    'P': dict(arm_median=1e-12, arm_p75=1e-7, arm_max=2e-3, within_1e3=11, block_median=1e-12),
}


def violations(kind, r):
    b = BOUNDS[kind]
    bad = []
    for k, v in b.items():
        if k == 'within_1e3':
            if r[k] < v:
                bad.append('%s %d < %d' % (k, r[k], v))
        elif r[k] > v:
            bad.append('%s %.2e > %.2e' % (k, r[k], v))
    return bad


@pytest.mark.parametrize('kind', ['U', 'P'])
def test_fast_model_vs_reference_step(kind):
    import fidelity_rows
    from oracle import OracleEnv
    assert OracleEnv(kind, seed=1, env_index=0).lib.rpo_get_rule(OracleEnv(kind, seed=1, env_index=0).h) == DEFAULT_RULE
    r = fidelity_rows.rows(kind, DEFAULT_RULE)
    print(kind, r)
    assert not violations(kind, r), (violations(kind, r), r)


@pytest.mark.parametrize('bit', [4, 256, 1024])
def test_the_bounds_notice_a_missing_contact_feature(bit):
    """without hull vertices / the contact cache / GJK beside the face the playroom id or pandaPick leaves the table"""
    import fidelity_rows
    bad = []
    for kind in ('U', 'P'):
        bad += violations(kind, fidelity_rows.rows(kind, DEFAULT_RULE & ~bit))
        if bad:
            break
    assert bad, 'rule bit %d off and every bound still holds' % bit
