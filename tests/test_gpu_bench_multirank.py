"""bench.py's N > 1 branch, run for real: two FRESH child processes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment, exactly what
torch.distributed.run hands a rank), both on device 0, RP_BENCH_BACKEND=gloo - the documented control-flow path of bench.py for a box with fewer GPUs
than ranks.  Never a measurement (two ranks share one GPU and the gather goes through the host): what is checked is that the branch runs and that its
JSON line says what the driver reads off it.  The 1 -> 8 GPU curve itself stays unmeasured (no 8-GPU node was available to the builder).

The children are started before this process touches the GPU (a GPU-initialised parent must never be replaced by another program, and need not be:
subprocess.Popen forks a child, which execs; the parent goes on)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo_on_one_device():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   RP_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1', '--no-extras',
                                       '--no-cpu-baseline', '--repeats', '1', '--envs-per-gpu', '512'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, err[-2000:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][1].splitlines() if l.startswith('{')], 'rank 0 alone prints the line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['scaling'] == 'weak'
    assert d['config']['collective'].startswith('all_gather(') and d['config']['collective_backend'] == 'gloo'
    ranges = sorted(tuple(r['envs']) for r in d['config']['ranks'])
    assert ranges == [(0, 512), (512, 1024)], ranges
    assert sorted(r['rank'] for r in d['config']['ranks']) == [0, 1]
    assert d['config']['global_envs'] == 1024 and d['steps'] == 5 and d['warmup'] == 1
    v = d['value']
    assert v == v and 0 < v < 1e9 and abs(v - 1024 * 5 / (d['ms_per_step'] * 5e-3)) < 1e-6 * v
    assert d['non_finite_envs'] == 0
    assert 'cpu_baseline' not in d and d['roofline']['traffic'] is None      # (not the headline batch size: no PMC profile quoted)


def run_bench(extra, ranks=1, timeout=900):
    port = free_port()
    procs = []
    for rank in range(ranks):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        if ranks > 1:
            env.update(RANK=str(rank), WORLD_SIZE=str(ranks), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RP_BENCH_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(ranks)] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, cwd=REPO))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    lines = [l for l in outs[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_strong_scaling_two_ranks_gloo_on_one_device():
    """--scaling strong: the config's envs IN TOTAL, split over the ranks in contiguous shards ("4096 total for strong scaling"); 1025 envs over two ranks
    = shards of 513 and 512, the short one padded for the collective.  A control-flow test like the weak-scaling one above."""
    d = run_bench(['--steps', '4', '--warmup', '1', '--no-extras', '--no-cpu-baseline', '--repeats', '1', '--scaling', 'strong', '--envs-per-gpu', '1025'], ranks=2)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['config']['global_envs'] == 1025
    assert sorted(tuple(r['envs']) for r in d['config']['ranks']) == [(0, 513), (513, 1025)]
    assert abs(d['value'] - 1025 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-6 * d['value'] and d['non_finite_envs'] == 0


@pytest.mark.parametrize('config,env_id,envs', [('C2', 'UR5PlayAbsRPY1Obj-v0', 1024), ('C3', 'pandaPick-v0', 4096), ('C5', 'UR5PlayAbsRPY1Obj-v0', 16384)])
def test_bench_prints_the_same_line_for_every_baseline_config(config, env_id, envs):
    """bench.py --config {C2, C3, C5}: BASELINE.json's other configs through the same protocol, the same JSON shape (roofline with that config's algorithmic bytes,
    per-launch times, the dominant kernel); C5 adds the CEM-MPC block.  Short regions: a shape test, the numbers are taken by tools/collect_profiles.sh."""
    d = run_bench(['--config', config, '--steps', '50' if config == 'C5' else '6', '--warmup', '2', '--no-cpu-baseline', '--repeats', '1', '--no-extras'])
    assert d['config']['name'] == config and env_id in d['config']['workload'] and d['config']['envs_per_gpu'] == envs and d['n_gpus'] == 1
    assert d['unit'] == 'env-steps/s' and d['value'] > 0 and d['non_finite_envs'] == 0 and d['roofline']['traffic'] is None
    alg = 552 if config == 'C3' else 1036
    assert d['roofline']['algorithmic_bytes_per_step'] == alg * envs
    assert abs(d['roofline']['achieved'] / d['roofline']['peak'] - d['roofline']['frac']) < 1e-12 and 0 < d['roofline']['frac'] < 1
    assert d['roofline']['dominant_kernel']['kernel'] == 'k_solve2' and d['roofline']['per_launch_ms']['k_solve2'] > 0
    if config == 'C5':
        c = d['cem_mpc']
        assert (c['start_states'], c['candidates'], c['horizon'], c['iterations']) == (32, 512, 50, 1) and d['steps'] == 50 and c['plans_per_s'] > 0
    else:
        assert 'cem_mpc' not in d and d['steps'] == 6
