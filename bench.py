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
# BASELINE.json's configs as concrete inputs.  `--config headline` (the default, what the driver runs) is the metric's own workload; C2 / C3 / C5 print
# the same JSON shape for theirs (C1 is the CPU-only plumbing case = the cpu_baseline leg of UR5Reach; C4 = the headline at --gpus 8).  alg = algorithmic bytes per
# env-step / per env-substep of the dominant kernel (state read + written), SURVEY.md §8d.
CONFIGS = {
    'headline': dict(env_id='UR5PlayAbsRPY1Obj-v0', envs=4096, alg=1036, alg_sub=544, what='the headline: 4096 playroom envs per GPU'),
    'C2': dict(env_id='UR5PlayAbsRPY1Obj-v0', envs=1024, alg=1036, alg_sub=544, what='BASELINE config C2: 1024 playroom envs on one GPU'),
    'C3': dict(env_id='pandaPick-v0', envs=4096, alg=552, alg_sub=248, what='BASELINE config C3: 4096 pandaPick envs on one GPU (second arm, grasp path)'),
    'C5': dict(env_id='UR5PlayAbsRPY1Obj-v0', envs=16384, alg=1036, alg_sub=544,
               what='BASELINE config C5: CEM-MPC, 32 start states x 512 candidate action sequences = 16384 envs, horizon 50, open loop, returns reduced per candidate on the device'),
}
CEM_GOALS, CEM_CANDIDATES, CEM_HORIZON, CEM_ELITES = 32, 512, 50, 64


B_RANGES = {'UR5PlayAbsRPY1Obj-v0': ([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0], [0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0]),
            # pandaPick-v0 (envList.py:18-22): goal range [-0.18, -0.18, 0] .. [0.18, 0.18, 0.2] in its own frame
            'pandaPick-v0': ([-0.18, -0.18, 0.0, -0.5, -0.5, -0.5, -1.0], [0.18, 0.18, 0.2, 0.5, 0.5, 0.5, 1.0])}


def make_actions(n, steps, device, seed, env_id=ENV_ID):
    """distribution B: xyz ~ U(goal_lo, goal_hi + [0,0,0.2]), rpy ~ U(-0.5, 0.5)^3, gripper ~ U(-1, 1)"""
    g = torch.Generator(device=device).manual_seed(seed)
    lo, hi = (torch.tensor(v, device=device) for v in B_RANGES[env_id])
    u = torch.rand((steps, n, 7), generator=g, device=device)
    return lo + (hi - lo) * u


def cpu_quota():
    """CPUs' worth of time the cgroup of this process may use (cgroup v2 cpu.max, v1 cfs quota), None = unlimited / unknown"""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(seed, margin=None, env_id=ENV_ID):
    """The CPU oracle (a port of the same semantics, NOT PyBullet) on the box's host cores: a bounded sample of the workload on one thread and on all of them (envs over
    threads, static partition - SURVEY.md 8d).  Sized so that every thread steps for seconds, not a start-up transient: 32 envs x 100 steps per thread (round 4's 8 x 50 was
    0.2 s of work per thread, and its worker threads shared one line of GJK counters: it read 7.2 x one core on 256 threads).  Reports the parallel efficiency beside it."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import numpy as np
    import oracle
    rng = np.random.default_rng(seed)
    lo, hi = (np.array(v) for v in B_RANGES[env_id])
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = cpu_quota()
    if quota is not None:                              # a container may see 256 logical CPUs and be allowed the time of eight (round 4's "7.2 x on 256 threads")
        cores = max(1, min(cores, int(quota + 0.999)))
    n_steps, per_thread = 100, 32
    a1 = lo + (hi - lo) * rng.random((per_thread, n_steps, 7))
    one = oracle.bench_rollout(env_id, seed, a1, 1, margin=margin)
    threads = min(cores, 1024)
    # a short all-thread probe first: the sample is then sized for ~12 s of wall time whatever the box's real parallelism is (32 envs per thread at most, 2 at least)
    probe = oracle.bench_rollout(env_id, seed, lo + (hi - lo) * rng.random((2 * threads, 20, 7)), threads, margin=margin)
    per_thread_all = int(min(32, max(2, round(probe * 12.0 / (threads * n_steps)))))
    aN = lo + (hi - lo) * rng.random((per_thread_all * threads, n_steps, 7))
    allc = oracle.bench_rollout(env_id, seed, aN, threads, margin=margin)
    per_thread_one, per_thread = per_thread, per_thread_all
    return {'value': allc, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'one_core': {'value': one, 'cores': 1, 'sample': '%d envs x %d steps' % (per_thread_one, n_steps)},
            'parallel_efficiency': allc / (one * threads), 'cpu_quota': quota,
            'sample': '%d envs x %d steps of %s (distribution B) on the fp64 CPU oracle, %d threads (= the logical CPUs this process may run on, capped by its cgroup\'s CPU quota; os.cpu_count() = %s) with the '
                      'envs statically partitioned (%d per thread, sized by a short probe for ~12 s of wall time), resets excluded; the one_core leg runs %d envs x %d steps on one thread; PyBullet is not installed '
                      'on this box' % (per_thread * threads, n_steps, env_id, threads, os.cpu_count(), per_thread, per_thread_one, n_steps),
            'host_cpus': os.cpu_count()}


STEP_KERNELS = ('k_action_prep', 'k_prep2', 'k_solve2', 'k_calc_state')     # what one rp_step launches (default pipeline)
PMC_SUMMARY = 'r06_pmc_summary.json'      # (the round's profile; pmc_traffic refuses it for another build of the library)
WARMUP_FLOOR = 200     # untimed steps before the first timed region whatever --warmup says: the rollout has reached its steady contact statistics by then (the first ~100 steps after a reset run ~3 % slower: arms still travelling from the rest pose), and clocks, caches and the load-sorted env pairing have settled


def pmc_traffic(kernel=None):
    """HBM-side bytes per launch of `kernel` (None: per env step, all of STEP_KERNELS weighted by their launches per step) from the
    committed rocprofv3 PMC passes (profiles/r06_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench;
    KB units; reads doubled per the gfx950 FETCH_SIZE correction in MI355X_MICROARCH.md).  PMC counters cannot be read from
    inside this process, so this is the profile's number, not a live one.  None if the file is absent, was taken with another
    library version, or lacks one of the kernels rp_step launches today (a stale profile is refused, not quoted)."""
    path = os.path.join(REPO, 'profiles', PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    prof = json.load(open(path))
    from roboticsplayroompybullet_amd import _lib
    if prof.get('library_version') != _lib.load().rp_version().decode():
        return None
    if any(k not in prof for k in STEP_KERNELS):
        return None
    one = lambda k: (2.0 * prof[k]['FETCH_SIZE_KB_avg'] + prof[k]['WRITE_SIZE_KB_avg']) * 1024.0   # noqa: E731
    if kernel is not None:
        return one(kernel)
    per_step = {'k_action_prep': 1, 'k_prep2': 11, 'k_solve2': 12, 'k_calc_state': 1}
    return sum(n * one(k) * prof[k].get('launch_fraction_of_envs', 1.0) for k, n in per_step.items())


def sharding_offset(rank, world, n):
    from roboticsplayroompybullet_amd import sharding
    return sharding.shard_range(rank, world, n)[0]


def cem_region(env, starts, iterations, device, seed):
    """BASELINE config C5, one timed region: `iterations` CEM iterations over CEM_GOALS start states (rows of rp_get_state, contact caches included) x CEM_CANDIDATES
    candidate action sequences of CEM_HORIZON steps.  Per iteration: every start state is broadcast to its 512 envs (rp_set_state), the candidates are drawn around the
    running mean / deviation of their start state, rolled out open loop, the rewards summed per candidate ON THE DEVICE, the best CEM_ELITES refit mean and deviation.
    Returns (env steps taken, seconds, the last iteration's best return per start state)."""
    n = env.num_envs
    g = torch.Generator(device=device).manual_seed(seed)
    lo, hi = (torch.tensor(v, device=device) for v in B_RANGES[ENV_ID])
    mu = (0.5 * (lo + hi)).expand(CEM_GOALS, CEM_HORIZON, 7).clone()
    sd = (0.25 * (hi - lo)).expand(CEM_GOALS, CEM_HORIZON, 7).clone()
    rows = starts.repeat_interleave(CEM_CANDIDATES, 0).contiguous()                  # [n, state words]: start state k in envs [512 k, 512 (k + 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best = None
    for _ in range(iterations):
        env.set_state(rows)
        eps = torch.randn((CEM_GOALS, CEM_CANDIDATES, CEM_HORIZON, 7), generator=g, device=device)
        acts = torch.clamp(mu[:, None] + sd[:, None] * eps, lo, hi)                # [goal, candidate, step, 7]
        plan = acts.permute(2, 0, 1, 3).reshape(CEM_HORIZON, n, 7).contiguous()
        ret = torch.zeros(n, device=device)
        for t in range(CEM_HORIZON):
            _, r, _, _ = env.step(plan[t])
            ret += r                                                                 # (the reward buffer is overwritten by the next step)
        ret = ret.view(CEM_GOALS, CEM_CANDIDATES)
        top = ret.topk(CEM_ELITES, dim=1).indices                                   # [goal, elite]
        elite = torch.gather(acts, 1, top[:, :, None, None].expand(-1, -1, CEM_HORIZON, 7))
        mu, sd = elite.mean(1), elite.std(1) + 1e-3
        best = ret.max(1).values
    torch.cuda.synchronize()
    return iterations * CEM_HORIZON, time.perf_counter() - t0, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='headline', help='BASELINE.json config (SURVEY.md 8d); the default is the metric\'s own workload')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak', help='N > 1 GPUs: weak = the config\'s envs PER GPU, strong = the config\'s envs in total, split over the ranks')
    ap.add_argument('--envs-per-gpu', type=int, default=0, help='override the config\'s batch size')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--groups', type=int, default=0, help='env groups per rp_step (0 = library default)')
    ap.add_argument('--contact-margin', type=float, default=None, help='rp_config.contact_margin in metres (default: the library default)')
    ap.add_argument('--stateless-contacts', action='store_true', help='RP_CFG_STATELESS_CONTACTS: no contact cache (round 3\'s first model)')
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps steps; `value` is the first one, the median is reported beside it')
    ap.add_argument('--no-extras', action='store_true', help='skip distribution A and the second contact margin')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    env_id = cfg['env_id']

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

    from roboticsplayroompybullet_amd import VecPlayEnv, sharding
    total = args.envs_per_gpu or cfg['envs']
    if args.scaling == 'strong' and world > 1:
        # strong scaling ("4096 total"): the config's envs split over the ranks in contiguous shards (sharding.shard_range: the remainder goes to the first ranks)
        lo_e, hi_e = sharding.shard_range(rank, world, total, total=True)
        n, offset = hi_e - lo_e, lo_e
        global_envs = total
    else:
        n, offset = total, sharding_offset(rank, world, total)
        global_envs = world * total
    env = VecPlayEnv(env_id, n, device=dev_index, seed=1234, env_offset=offset, contact_margin=args.contact_margin,
                     persistent_manifolds=not args.stateless_contacts)
    if args.groups:
        env.set_groups(args.groups)
    env.reset()
    pre = max(0, WARMUP_FLOOR - args.warmup)          # uncounted steps in front of the --warmup ones
    actions = make_actions(n, args.steps + args.warmup + pre, device, 1234 + rank, env_id)
    actions, pre_actions = actions[pre:], actions[:pre]
    pack_w = env.dims['obs_quat'] + env.dims['achieved_goal'] + 2
    # the all-gather's receive buffer: equal shards (weak scaling; strong scaling when the ranks divide the batch); a remainder's short shards are padded to the longest
    shard_rows = n
    if world > 1 and args.scaling == 'strong':
        shard_rows = max(b - a for a, b in (sharding.shard_range(r, world, total, total=True) for r in range(world)))
    gathered = torch.empty((world * shard_rows, pack_w), dtype=torch.float32, device=device) if world > 1 else None

    def gather(async_op):
        return sharding.gather_observations(sharding.pad_rows(env.pack, shard_rows), out=gathered, async_op=async_op)

    def timed_region(env, acts, first, steps, events=None):
        """EXACTLY `steps` env steps between barrier + synchronize on both sides; returns the max over ranks of the wall seconds.
        For N > 1 the only collective on the path - the all-gather of env.pack (obs_quat | achieved_goal | reward | is_success,
        written in that layout by k_calc_state itself) - is enqueued asynchronously after every step, so the gather of step k runs
        on RCCL's stream while step k + 1's physics runs on ours; every gather is complete inside the region."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = None
        if events:
            events[0].record()                 # ONE pair around the whole region (a pair per step puts two marker packets into the queue per step: 2 % of the step)
        for k in range(steps):
            obs, r, done, info = env.step(acts[first + k])
            if world > 1:
                if pending is not None:
                    pending.wait()
                _, pending = gather(True)
        if pending is not None:
            pending.wait()
        if events:
            events[1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, info

    for a in list(pre_actions) + [actions[k] for k in range(args.warmup)]:
        env.step(a)
        if world > 1:
            gather(False)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cem = None
    if args.config == 'C5':
        # the CEM-MPC shape: 32 start states = the states 32 of the warmed-up envs are in now; a step = one env.step of all 16384 candidates; --steps is rounded to whole
        # iterations of CEM_HORIZON steps (at least one); the region includes the broadcasts, the sampling, the return reduction and the refit
        assert world == 1 and n == CEM_GOALS * CEM_CANDIDATES, 'C5 is a one-GPU config of 32 x 512 envs'
        starts = env.get_state()[:: CEM_CANDIDATES].clone()
        iters = max(1, args.steps // CEM_HORIZON)
        cem_region(env, starts, 1, device, 99)                  # (one untimed iteration: allocator, clocks)
        ev0.record()
        steps_done, elapsed, best = cem_region(env, starts, iters, device, 100)
        ev1.record()
        torch.cuda.synchronize()
        info = {'status': env.buf['status'], 'is_success': env.buf['is_success']}
        cem = {'start_states': CEM_GOALS, 'candidates': CEM_CANDIDATES, 'horizon': CEM_HORIZON, 'elites': CEM_ELITES, 'iterations': iters,
               'plans_per_s': CEM_GOALS * iters / elapsed, 'ms_per_iteration': 1e3 * elapsed / iters, 'best_return_mean': float(best.mean().item()),
               'what': 'one iteration = rp_set_state broadcast of the 32 start rows (records + contact caches) to 512 envs each, 50 open-loop env steps of all 16384 candidates, '
                       'per-candidate return summed on the device, top-64 refit of mean and deviation; `steps` = iterations x 50'}
        args.steps = steps_done
    else:
        elapsed, info = timed_region(env, actions, args.warmup, args.steps, (ev0, ev1))       # the contract's region: `value`
    torch.cuda.synchronize()
    step_ms = ev0.elapsed_time(ev1) / args.steps      # device-side time of the same region (events on rp_step's stream)
    # per-launch kernel durations: hipEvent pairs recorded on the launch stream inside rp_step; bracketing single kernels
    # needs the env groups (concurrent streams) switched off, so this is a second region over the same actions
    n_kt = min(args.steps, 50)
    env.enable_timers(n_kt)
    for k in range(n_kt):
        env.step(actions[args.warmup + k % max(1, actions.shape[0] - args.warmup)])
    torch.cuda.synchronize()
    tm = env.timers()
    env.enable_timers(0)
    bad = int((info['status'] & 1).sum().item())
    fell = int(((info['status'] & 2) != 0).sum().item())
    success = float(info['is_success'].float().mean().item())
    # repeats of the same region (same actions; the state keeps evolving) for a median, then the extras: distribution A (the literal
    # random-action distribution of SURVEY.md 8d: U(action_space.low, high)) and the other contact margin
    rep_values = [global_envs * args.steps / elapsed]
    n_rep = max(0, args.repeats - 1) if cem is None else 0
    if n_rep:      # FRESH actions per repeat (round 5: replaying the first region's twenty actions let the arms settle near the same targets - each repeat 1 - 2 % faster than the one before)
        more = make_actions(n, n_rep * args.steps, device, 9000 + rank, env_id)
    for i in range(n_rep):
        t_rep, _ = timed_region(env, more, i * args.steps, args.steps)
        rep_values.append(global_envs * args.steps / t_rep)
    extras = {}
    if not args.no_extras and world == 1 and cem is None:
        g = torch.Generator(device=device).manual_seed(4321 + rank)
        hi = env.action_high
        acts_a = (2 * torch.rand((pre + args.warmup + args.steps, n, 7), generator=g, device=device) - 1) * hi
        for k in range(pre + args.warmup):          # the same floor of untimed steps for this distribution (other actions, other contact sets), every step on actions of its own
            env.step(acts_a[k])
        t_a, info_a = timed_region(env, acts_a, pre + args.warmup, args.steps)
        extras['distribution_A'] = {'value': n * args.steps / t_a, 'ms_per_step': 1e3 * t_a / args.steps,
                                    'what': 'a ~ U(action_space.low, action_space.high) = U(-6, 6)^6 x U(-1, 1), resampled every step (the '
                                            'literal random-action rollout: targets mostly unreachable, the arm slews at the per-step clip); continues from the headline rollout, %d untimed steps of this '
                                            'distribution first' % (pre + args.warmup),
                                    'non_finite_envs': int((info_a['status'] & 1).sum().item())}
        env.close()
        for other in ((0.005, 0.02) if args.config == 'headline' else ()):      # uniform margins beside the default (per pair: Bullet's relative breaking thresholds)
            if args.contact_margin is not None and abs(args.contact_margin - other) < 1e-9:
                continue
            env2 = VecPlayEnv(env_id, n, device=dev_index, seed=1234, contact_margin=other)
            env2.reset()
            for a in list(pre_actions) + [actions[k] for k in range(args.warmup)]:
                env2.step(a)
            t_m, _ = timed_region(env2, actions, args.warmup, args.steps)
            extras['contact_margin_%g' % other] = {'value': n * args.steps / t_m, 'ms_per_step': 1e3 * t_m / args.steps,
                                                   'what': 'the same workload with rp_config.contact_margin = %g m for every pair' % other}
            env2.close()

    # who took part: every rank's device as RCCL / torch saw it, gathered so that "did N ranks on N GPUs run" can be read off the line
    props = torch.cuda.get_device_properties(dev_index)
    me = {'rank': rank, 'local_rank': local_rank, 'device_index': dev_index, 'name': props.name, 'uuid': str(getattr(props, 'uuid', '')),
          'envs': [offset, offset + n]}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if rank == 0:
        value = global_envs * args.steps / elapsed
        margin_used = ('stateless, margin %g m for every pair' % args.contact_margin) if args.contact_margin is not None else (
            "stateless (RP_CFG_STATELESS_CONTACTS), per-pair margins = Bullet's relative breaking thresholds" if args.stateless_contacts else
            "persistent manifolds (per-env contact cache) and GJK on the arm links' hulls beside box faces (library defaults), per-pair breaking thresholds = Bullet's relative ones")
        solve_ms = tm['avg_solve_ms']
        alg, alg_sub = cfg['alg'], cfg['alg_sub']
        achieved = alg_sub * n / (solve_ms * 1e-3) / 1e9
        step_achieved = alg * n / (step_ms * 1e-3) / 1e9
        headline = args.config == 'headline'
        line = {
            'metric': 'env-steps/sec at N=4096 parallel UR5PlayAbsRPY1Obj-v0 envs per MI355X' if headline and args.scaling == 'weak' else
                      'env-steps/sec, %s (%d envs %s)' % (cfg['what'], total, 'in total, strong scaling' if args.scaling == 'strong' else 'per GPU'),
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %d envs %s, 12 substeps x 50 PGS sweeps per step, %s, reset excluded, contacts: %s'
                                   % (env_id, total, 'in total over the ranks' if args.scaling == 'strong' and world > 1 else 'per GPU',
                                      'CEM-MPC candidate rollouts (open loop, sampled around a refitted mean)' if cem else 'random actions (distribution B, resampled every step)', margin_used),
                       'name': args.config, 'envs_per_gpu': n, 'global_envs': global_envs, 'parallelism': 'env-shard x%d' % world,
                       'collective': 'all_gather(obs_quat+achieved_goal+reward+is_success) per step' if world > 1 else 'none',
                       'collective_backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')) if world > 1 else None,
                       'ranks_seen': dist.get_world_size() if world > 1 else 1, 'ranks': ranks,
                       'warmup_untimed_steps': pre + args.warmup},
            # frac = the strict SURVEY.md 8d figure: algorithmic bytes of a whole env step / measured step time / HBM peak; the dominant
            # kernel's own per-launch figure sits in `dominant_kernel`
            'roofline': {'bound': 'latency/issue', 'nominal_bound': 'hbm', 'hbm_frac': step_achieved / HBM_PEAK_GBS, 'achieved': step_achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': step_achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic() if (headline and n == ENVS_PER_GPU) else None,
                         'what': 'whole env step: %d B algorithmic per env-step (SURVEY.md 8d) x %d envs / %.3f ms (torch events on the stream of rp_step around the timed region)'
                                 % (alg, n, step_ms),
                         'algorithmic_bytes_per_step': alg * n,
                         'limiter': 'latency / issue, not bandwidth: 50 sweeps of dependent PGS row updates per k_solve2 launch (the launch lasts as '
                                    'long as its heaviest wave) - the HBM fraction is reported because the contract asks for it, not because it '
                                    'is the ceiling; advisory FLOP model 9e6 FLOP/env-step => %.3g of the 157.3 TFLOP/s fp32 vector peak'
                                    % (9e6 * n / (step_ms * 1e-3) / 157.3e12),
                         'dominant_kernel': {'kernel': 'k_solve2', 'kernel_ms': solve_ms, 'launches_per_step': 12, 'achieved': achieved,
                                             'frac': achieved / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': alg_sub * n,
                                             'traffic': pmc_traffic('k_solve2') if (headline and n == ENVS_PER_GPU) else None},
                         'traffic_note': 'bytes per env step from profiles/%s (sum over the step\'s launches of 2*FETCH_SIZE + WRITE_SIZE); null when '
                                         'that profile was taken with another library version or another config' % PMC_SUMMARY,
                         'per_launch_ms': {'k_action': tm['avg_action_ms'], 'k_prep2': tm['avg_prep_ms'], 'k_solve2': solve_ms,
                                           'k_calc_state': tm['avg_obs_ms'], 'steps_timed': tm['steps_timed'],
                                           'how': 'hipEvent pair around every launch on the launch stream (rp_enable_timers), separate '
                                                  'region right after the timed one with the env-group streams switched off'}},
            'repeats': {'values': rep_values, 'median': sorted(rep_values)[len(rep_values) // 2],
                        'what': '%d timed regions of %d steps each, every region on actions of its own; `value` is the first' % (len(rep_values), args.steps)},
            'non_finite_envs': bad, 'fallen_objects': fell, 'success_rate_last_step': success,
        }
        if cem:
            line['cem_mpc'] = cem
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(1234, args.contact_margin, env_id)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
