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


def allreduce_grads(module, scale=None):
    """sum-all-reduce every parameter gradient in ONE grouped collective (RCCL group call: no flat copy of the
    P = 4HN+2H+N floats and no copy back -- the gradients are reduced where autograd left them); `scale` (e.g. 1/world
    for a mean loss over the global batch) is applied after the reduction."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in ps]
    if not all(g.is_contiguous() for g in grads):
        for p in ps:
            p.grad = p.grad.contiguous()
        grads = [p.grad for p in ps]
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dist.all_reduce_coalesced(grads, op=dist.ReduceOp.SUM)
    except (AttributeError, RuntimeError):   # backend without the grouped form: one collective per tensor
        for g in grads:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
    if scale is not None:
        torch._foreach_mul_(grads, scale)


def allreduce_scalars(*vals):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vals
    t = torch.stack([v.detach().reshape(()) for v in vals])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(t.unbind(0))
