"""Multi-GPU: contiguous env shards per rank and the single collective on the path.

Envs never interact, so physics needs no exchange.  Rank r owns global envs [r*n, (r+1)*n); the per-env counter RNG is
keyed by the GLOBAL env index (rp_config.env_offset), so concatenating the ranks' outputs equals a single-device run
bit for bit.  The only collective is an all-gather of the per-step observation pack for a single consumer
(RCCL over xGMI on GPUs: backend "nccl"; the same code runs on gloo for the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world_size, envs, total=False):
    """global env indices [lo, hi) owned by `rank`; env_offset for rp_create is the first one.  Weak scaling (default): `envs` per rank.  total=True (strong
    scaling: `envs` in all): contiguous shards of envs // world_size, the first envs % world_size ranks one env larger."""
    if not total:
        lo = rank * envs
        return lo, lo + envs
    base, rem = divmod(envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_observations(obs, reward, is_success):
    """[n, D+G+2] float32: obs_quat | achieved_goal | reward | is_success — what a learner consumes each step"""
    return torch.cat([obs['obs_quat'], obs['achieved_goal'], reward[:, None], is_success.to(torch.float32)[:, None]], dim=1)


def unpack_observations(pack, d_obs, d_ag):
    return {'obs_quat': pack[:, :d_obs], 'achieved_goal': pack[:, d_obs:d_obs + d_ag], 'reward': pack[:, d_obs + d_ag],
            'is_success': pack[:, d_obs + d_ag + 1].to(torch.int32)}


def pad_rows(pack, rows):
    """strong scaling with a remainder: a rank's pack padded with zero rows to the longest shard (collectives want equal contributions)"""
    if pack.shape[0] == rows:
        return pack
    return torch.cat([pack, pack.new_zeros((rows - pack.shape[0], pack.shape[1]))], 0)


def strip_padding(gathered, sizes):
    """the valid rows of a gather of padded packs, in global env order: [sum(sizes), W]"""
    rows = gathered.shape[0] // len(sizes)
    if all(s == rows for s in sizes):
        return gathered
    return torch.cat([gathered[r * rows:r * rows + s] for r, s in enumerate(sizes)], 0)


def gather_observations(pack, out=None, group=None, async_op=False):
    """all-gather the ranks' packs in rank order -> [world*n, W].  `out` may be a preallocated buffer.  With async_op the
    collective is only enqueued (RCCL runs it on its own stream while the caller's stream goes on with the next step's
    physics) and (out, work) is returned: work.wait() before `out` is read or handed to the next gather."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * pack.shape[0], pack.shape[1]), dtype=pack.dtype, device=pack.device)
    if dist.get_backend(group) == 'nccl':
        work = dist.all_gather_into_tensor(out, pack.contiguous(), group=group, async_op=async_op)
    else:
        work = dist.all_gather(list(out.chunk(world, dim=0)), pack.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out
