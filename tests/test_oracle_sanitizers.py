"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the host build; GPU sanitizers do not exist on this pool).

`make -C oracle asan` builds oracle/rpo_asan (fp64) and rpo_asan_f32 from rp_oracle.c + asan_main.c; the driver rolls all six models through resets and steps under both
action distributions - the literal random-action one drives the hull scans, GJK with cached simplices, the contact cache's insertions and the cache-row export / import -
and a threaded rpo_bench_rollout.  Any finding aborts the child process; this test fails with its report."""
import os
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')


def test_oracle_rollouts_clean_under_asan_ubsan():
    b = subprocess.run(['make', '-C', ORACLE, '-s', 'asan'], capture_output=True, text=True)
    if b.returncode != 0 and 'sanitize' in (b.stderr + b.stdout) and 'unrecognized' in (b.stderr + b.stdout):
        pytest.skip('this compiler has no -fsanitize=address,undefined')
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    for exe, steps in (('rpo_asan', 25), ('rpo_asan_f32', 25)):
        r = subprocess.run([os.path.join(ORACLE, exe), str(steps), '2'], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, '%s:\n%s\n%s' % (exe, r.stdout[-1500:], r.stderr[-4000:])
        assert 'threads:' in r.stdout and r.stdout.count('clean') >= 7, r.stdout
        assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
