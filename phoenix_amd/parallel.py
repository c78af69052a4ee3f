"""Data parallelism over trajectories: parameters replicated, samples sharded,
ONE flat all-reduce of the P = 4HN+2H+N gradient floats per optimizer step over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """contiguous chunk of `n_items` owned by `rank` (remainder spread over the first ranks)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flat_grads(module):
    ps = [p for p in module.parameters() if p.requires_grad]
    return ps, torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])


def allreduce_grads(module, scale=None):
    """sum-all-reduce every parameter gradient as one flat fp32 buffer; `scale` (e.g. 1/world for a
    mean loss over the global batch) is applied after the reduction."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    ps, flat = flat_grads(module)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if scale is not None:
        flat.mul_(scale)
    views, off = [], 0
    for p in ps:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p))
        off += n
    have = [p.grad is not None for p in ps]
    if all(have):
        torch._foreach_copy_([p.grad for p in ps], views)     # one fused copy-back instead of one kernel per tensor
    else:
        for p, g in zip(ps, views):
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)


def allreduce_scalars(*vals):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vals
    t = torch.stack([v.detach().reshape(()) for v in vals])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(t.unbind(0))
