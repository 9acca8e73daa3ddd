"""Multi-GPU: contiguous env shards per rank and the single collective on the path.

Envs never interact, so physics needs no exchange.  Rank r owns global envs [r*n, (r+1)*n); the per-env counter RNG is
keyed by the GLOBAL env index (rp_config.env_offset), so concatenating the ranks' outputs equals a single-device run
bit for bit.  The only collective is an all-gather of the per-step observation pack for a single consumer
(RCCL over xGMI on GPUs: backend "nccl"; the same code runs on gloo for the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world_size, envs_per_rank):
    """global env indices owned by `rank`; env_offset for rp_create is the first one"""
    lo = rank * envs_per_rank
    return lo, lo + envs_per_rank


def pack_observations(obs, reward, is_success):
    """[n, D+G+2] float32: obs_quat | achieved_goal | reward | is_success — what a learner consumes each step"""
    return torch.cat([obs['obs_quat'], obs['achieved_goal'], reward[:, None], is_success.to(torch.float32)[:, None]], dim=1)


def unpack_observations(pack, d_obs, d_ag):
    return {'obs_quat': pack[:, :d_obs], 'achieved_goal': pack[:, d_obs:d_obs + d_ag], 'reward': pack[:, d_obs + d_ag],
            'is_success': pack[:, d_obs + d_ag + 1].to(torch.int32)}


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
