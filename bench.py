#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of UR5PlayAbsRPY1Obj-v0 at N = 4096 parallel envs per MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 200 --warmup 20

One "step" = one playEnv.step() for every env of the batch (one rp_step call): clip -> AbsRPY IK -> motor targets
(k_action; in the default pipeline fused with the first k_prep2 into one launch, k_action_prep) -> 12 physics substeps at 300 Hz
(k_prep2 + k_solve2 each) -> observation + reward (k_calc_state).
Per-launch durations are measured with hipEvents recorded on the launch stream inside rp_step over the whole timed
region (rp_enable_timers / rp_get_timers); the roofline object is for the dominant kernel, k_solve2.  Actions are synthetic
(distribution B of SURVEY.md §8d: workspace-uniform, resampled every step), pre-generated on the device so the timed
region holds only the hot path (and, for N > 1 GPUs, the RCCL all-gather of observations, enqueued asynchronously so that it
overlaps the next step's physics; every gather is complete before the region ends).  Envs shard across ranks
with no data-path collective (weak scaling: 4096 envs per GPU).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ENV_ID = 'UR5PlayAbsRPY1Obj-v0'
ENVS_PER_GPU = 4096
ALG_BYTES_PER_ENV_STEP = 1036          # SURVEY.md §8d / BASELINE.md §4: action 28 + goal 44 + state 272 r + 272 w + outputs 420
ALG_BYTES_PER_ENV_SUBSTEP = 544        # k_solve, one launch = one substep of every env: state S = 272 B read + 272 B written
HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_actions(n, steps, device, seed):
    """distribution B: xyz ~ U(goal_lo, goal_hi + [0,0,0.2]), rpy ~ U(-0.5, 0.5)^3, gripper ~ U(-1, 1)"""
    g = torch.Generator(device=device).manual_seed(seed)
    lo = torch.tensor([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0], device=device)
    hi = torch.tensor([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0], device=device)
    u = torch.rand((steps, n, 7), generator=g, device=device)
    return lo + (hi - lo) * u


def cpu_baseline(seed, budget_s=12.0):
    """The CPU oracle (a port of the same semantics, NOT PyBullet) on one host core: bounded sample of the workload."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import numpy as np
    from oracle import OracleEnv
    rng = np.random.default_rng(seed)
    lo = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0])
    hi = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])
    n_env, n_steps, done = 0, 100, 0
    t_total = 0.0
    while t_total < budget_s and n_env < 256:
        env = OracleEnv(ENV_ID, seed=seed, env_index=n_env)
        env.reset()
        acts = lo + (hi - lo) * rng.random((n_steps, 7))
        t0 = time.perf_counter()
        for a in acts:
            env.step(a)
        t_total += time.perf_counter() - t0
        done += n_steps
        n_env += 1
    return {'value': done / t_total, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': '%d envs x %d steps of %s (distribution B) on the fp64 CPU oracle, 1 thread, reset excluded; '
                      'PyBullet is not installed on this box' % (n_env, n_steps, ENV_ID),
            'host_cpus': os.cpu_count()}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc_summary.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench; KB units; reads doubled per the gfx950 FETCH_SIZE
    correction in MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process, so this is the profile's
    number, not a live one; None if the file is absent."""
    path = os.path.join(REPO, 'profiles', 'r01_pmc_summary.json')
    if not os.path.exists(path):
        return None
    k = json.load(open(path)).get(kernel)
    if not k:
        return None
    return (2.0 * k['FETCH_SIZE_KB_avg'] + k['WRITE_SIZE_KB_avg']) * 1024.0


def sharding_offset(rank, world, n):
    from roboticsplayroompybullet_amd import sharding
    return sharding.shard_range(rank, world, n)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--groups', type=int, default=0, help='env groups per rp_step (0 = library default)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # one rank per GPU; RP_BENCH_BACKEND=gloo is a control-flow check of the N > 1 path on a box with fewer GPUs than ranks (ranks
    # then share devices and the gather goes through the host) - never a measurement
    backend = os.environ.get('RP_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from roboticsplayroompybullet_amd import VecPlayEnv
    n = args.envs_per_gpu
    env = VecPlayEnv(ENV_ID, n, device=dev_index, seed=1234, env_offset=sharding_offset(rank, world, n))
    if args.groups:
        env.set_groups(args.groups)
    env.reset()
    actions = make_actions(n, args.steps + args.warmup, device, 1234 + rank)
    pack_w = env.dims['obs_quat'] + env.dims['achieved_goal'] + 2
    gathered = torch.empty((world * n, pack_w), dtype=torch.float32, device=device) if world > 1 else None

    from roboticsplayroompybullet_amd import sharding

    def one_step(k):
        obs, r, done, info = env.step(actions[k])
        if world > 1:   # the only collective on the path: gather observations for a single consumer
            sharding.gather_observations(sharding.pack_observations(obs, r, info['is_success']), out=gathered)
        return info

    for k in range(args.warmup):
        one_step(k)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for k in range(args.steps):
        ev0[k].record()
        obs, r, done, info = env.step(actions[args.warmup + k])
        ev1[k].record()
        if world > 1:      # the gather of step k runs on RCCL's stream while step k + 1's physics runs on ours
            pack = sharding.pack_observations(obs, r, info['is_success'])
            if pending is not None:
                pending.wait()
            _, pending = sharding.gather_observations(pack, out=gathered, async_op=True)
    if pending is not None:
        pending.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps
    # per-launch kernel durations: hipEvent pairs recorded on the launch stream inside rp_step; bracketing single kernels
    # needs the env groups (concurrent streams) switched off, so this is a second region over the same actions
    n_kt = min(args.steps, 50)
    env.enable_timers(n_kt)
    for k in range(n_kt):
        env.step(actions[args.warmup + k])
    torch.cuda.synchronize()
    tm = env.timers()
    env.enable_timers(0)
    bad = int(info['status'].sum().item())
    success = float(info['is_success'].float().mean().item())

    if rank == 0:
        value = world * n * args.steps / elapsed
        solve_ms = tm['avg_solve_ms']
        achieved = ALG_BYTES_PER_ENV_SUBSTEP * n / (solve_ms * 1e-3) / 1e9
        step_achieved = ALG_BYTES_PER_ENV_STEP * n / (step_ms * 1e-3) / 1e9
        line = {
            'metric': 'env-steps/sec at N=4096 parallel UR5PlayAbsRPY1Obj-v0 envs per MI355X',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %d envs per GPU, 12 substeps x 50 PGS sweeps per step, random actions '
                                   '(distribution B, resampled every step), reset excluded' % (ENV_ID, n),
                       'envs_per_gpu': n, 'global_envs': world * n, 'parallelism': 'env-shard x%d' % world,
                       'collective': 'all_gather(obs_quat+achieved_goal+reward+is_success) per step' if world > 1 else 'none'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic('k_solve2') if n == ENVS_PER_GPU else None, 'kernel': 'k_solve2', 'kernel_ms': solve_ms, 'launches_per_step': 12,
                         'traffic_note': 'bytes per k_solve2 launch from profiles/r01_pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE); '
                                         'it is ~13x the algorithmic bytes because the constraint rows (~5 KB per env-substep) are '
                                         'handed from k_prep2 to k_solve2 through an Infinity-Cache-resident workspace',
                         'algorithmic_bytes_per_launch': ALG_BYTES_PER_ENV_SUBSTEP * n,
                         'whole_step': {'achieved': step_achieved, 'frac': step_achieved / HBM_PEAK_GBS, 'ms': step_ms,
                                        'algorithmic_bytes': ALG_BYTES_PER_ENV_STEP * n},
                         'per_launch_ms': {'k_action': tm['avg_action_ms'], 'k_prep2': tm['avg_prep_ms'], 'k_solve2': solve_ms,
                                           'k_calc_state': tm['avg_obs_ms'], 'steps_timed': tm['steps_timed'],
                                           'how': 'hipEvent pair around every launch on the launch stream (rp_enable_timers), separate '
                                                  'region right after the timed one with the env-group streams switched off'},
                         'note': 'latency-bound path (50 sweeps of dependent PGS row updates per launch; the launch lasts as long as its '
                                 'heaviest wave), not bandwidth-bound; advisory FLOP model '
                                 '9e6 FLOP/env-step => %.3g of the 157.3 TFLOP/s fp32 vector peak' % (9e6 * n / (step_ms * 1e-3) / 157.3e12)},
            'non_finite_envs': bad, 'success_rate_last_step': success,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(1234)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
