"""Multi-process helpers of the data-parallel path (one process per GPU; torch.distributed backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU).  Mirrors the pieces of slowfast/utils/distributed.py that sit on the hot loop:
``all_reduce`` of the per-iteration scalars (:98-114, used at tools/train_net.py:284-287) -- batched into ONE
collective here instead of three -- and the rank-strided sample sharding of DistributedSampler
(slowfast/datasets/utils.py:326-341) used to shard clips / sliding windows over ranks."""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce(tensors, average=True):
    """All-reduce a list of same-device scalars/tensors in one collective; returns new tensors (mean if average)."""
    world = get_world_size()
    if world == 1:
        return [t.clone() for t in tensors]
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat, async_op=False)
    if average:
        flat.mul_(1.0 / world)
    out, off = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[off:off + n].reshape(t.shape))
        off += n
    return out


def all_gather_cat(t):
    """Concatenate a [n, ...] tensor from every rank along dim 0 (tools/test_net.py:119-122)."""
    world = get_world_size()
    if world == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, 0)


def shard_indices(n, rank=None, world=None, pad=True):
    """Indices of the items rank `rank` processes: rank-strided like DistributedSampler(shuffle=False); with pad the list is
    padded by wrapping so every rank gets ceil(n/world) items (same collective count on every rank)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    idx = list(range(n))
    if pad and n and n % world:
        idx = [i % n for i in range(-(-n // world) * world)]      # wraps as often as needed (fewer items than ranks included)
    return idx[rank::world]
