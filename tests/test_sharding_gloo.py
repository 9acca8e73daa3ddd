"""N > 1 host path on CPU: two gloo ranks shard the env range, pack and all-gather observations in rank order, and the
per-env RNG keyed by the global env index makes shards equal to the single-device run (checked on the CPU oracle)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from roboticsplayroompybullet_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d_obs, d_ag, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = sharding.shard_range(rank, world, n)
    g = torch.arange(lo, hi, dtype=torch.float32)
    obs = {'obs_quat': g[:, None] + torch.arange(d_obs)[None] * 0.01, 'achieved_goal': g[:, None] * 2 + torch.arange(d_ag)[None]}
    pack = sharding.pack_observations(obs, -g, (g.to(torch.int32) % 2))
    full = sharding.gather_observations(pack)
    out2, work = sharding.gather_observations(pack * 2, async_op=True)      # the overlapped form bench.py uses
    work.wait()
    assert torch.equal(out2, full * 2)
    dist.barrier()
    if rank == 0:
        ret['full'] = full.numpy().copy()
    dist.destroy_process_group()


def test_two_rank_gather_is_rank_ordered_concatenation():
    world, n, d_obs, d_ag = 2, 5, 19, 11
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d_obs, d_ag, ret), nprocs=world, join=True)
    full = torch.tensor(ret['full'])
    assert full.shape == (world * n, d_obs + d_ag + 2)
    u = sharding.unpack_observations(full, d_obs, d_ag)
    g = torch.arange(world * n, dtype=torch.float32)
    assert torch.equal(u['obs_quat'][:, 0], g)                    # global env order preserved
    assert torch.equal(u['achieved_goal'][:, 0], 2 * g)
    assert torch.equal(u['reward'], -g)
    assert torch.equal(u['is_success'], (g.to(torch.int32) % 2))


def _strong_worker(rank, world, port, total, w, ret):
    """strong scaling (bench.py --scaling strong): `total` envs in all, shards of unequal size when world does not divide it; the gather is an all_gather into views of
    one buffer, rank order = global env order"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = sharding.shard_range(rank, world, total, total=True)
    pack = torch.arange(lo, hi, dtype=torch.float32)[:, None] + 0.001 * torch.arange(w)[None]
    sizes = [b - a for a, b in (sharding.shard_range(r, world, total, total=True) for r in range(world))]
    out, work = sharding.gather_observations(sharding.pad_rows(pack, max(sizes)), async_op=True)      # (collectives want equal shards: the short ones are padded)
    work.wait()
    full = sharding.strip_padding(out, sizes)
    dist.barrier()
    if rank == 0:
        ret['full'] = full.numpy().copy()
        ret['sizes'] = sizes
    dist.destroy_process_group()


def test_strong_scaling_shards_cover_the_env_range_in_order():
    for world, total in ((2, 7), (2, 8)):
        ranges = [sharding.shard_range(r, world, total, total=True) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1)), ranges
        assert max(b - a for a, b in ranges) - min(b - a for a, b in ranges) <= 1
    assert [sharding.shard_range(r, 8, 4096, total=True) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_strong_worker, args=(2, _free_port(), 7, 5, ret), nprocs=2, join=True)
    assert list(ret['sizes']) == [4, 3]
    np.testing.assert_allclose(ret['full'][:, 0], np.arange(7))


def test_global_env_index_keys_the_rng_so_shards_equal_one_run():
    """oracle-side statement of the shard-equivalence property (the GPU version is tests/test_gpu_parity.py)."""
    from oracle import OracleEnv
    single = [OracleEnv('R', seed=5, env_index=e).reset()['obs_quat'] for e in range(4)]
    for rank in range(2):
        lo, hi = sharding.shard_range(rank, 2, 2)
        shard = [OracleEnv('R', seed=5, env_index=e).reset()['obs_quat'] for e in range(lo, hi)]
        np.testing.assert_array_equal(np.stack(shard), np.stack(single[lo:hi]))
    assert not np.array_equal(single[0], single[1])               # different envs draw different numbers
