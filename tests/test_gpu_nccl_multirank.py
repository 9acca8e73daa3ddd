"""The N > 1 path on REAL devices over RCCL (backend "nccl"), for boxes that show at least two GPUs; skipped elsewhere (the gpurun pool's boxes show one - the 1 -> 8 GPU
curve stays unmeasured; tests/test_gpu_bench_multirank.py and tests/test_sharding_gloo.py cover the same control flow on one device / on the CPU).

Children are fresh processes started before this one touches a GPU (torch.cuda.device_count() does not initialise it on this image), one rank per device:
  * bench.py --gpus 2 as torch.distributed.run would start it: the JSON line's multi-rank fields, collective_backend == nccl;
  * shard equivalence of the gathered observation pack: two ranks x 128 envs over RCCL == one rank x 256 envs, bit for bit, every step (contiguous shards keyed by the
    global env index: sharding.py; north_star: "an RCCL-over-xGMI gather of observations only")."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two visible GPUs (RCCL between real devices)')]

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(argv, ranks):
    port = free_port()
    procs = []
    for rank in range(ranks):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('RP_BENCH_BACKEND', None)
        if ranks > 1:
            env.update(RANK=str(rank), WORLD_SIZE=str(ranks), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    return outs


def test_bench_two_ranks_over_rccl():
    outs = launch([os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '2', '--no-extras', '--no-cpu-baseline', '--repeats', '1'], 2)
    lines = [l for l in outs[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1].splitlines() if l.startswith('{')]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['scaling'] == 'weak' and d['config']['collective_backend'] == 'nccl'
    assert sorted(tuple(r['envs']) for r in d['config']['ranks']) == [(0, 4096), (4096, 8192)]
    assert len({r['uuid'] for r in d['config']['ranks']}) == 2, 'two ranks, two devices'
    assert d['non_finite_envs'] == 0 and d['value'] > 0


def test_gathered_pack_of_two_rccl_ranks_equals_one_rank(tmp_path):
    worker = os.path.join(REPO, 'tests', 'nccl_shard_worker.py')
    two, one = str(tmp_path / 'two.npy'), str(tmp_path / 'one.npy')
    launch([worker, '256', '12', two], 2)
    launch([worker, '256', '12', one], 1)
    a, b = np.load(two), np.load(one)
    assert a.shape == b.shape == (12, 256, a.shape[2]) and np.isfinite(a).all()
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), 'the ranks\' gathered observation pack differs from the single-device run'
