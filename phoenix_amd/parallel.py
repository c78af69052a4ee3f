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


def _has_coalesced():
    return hasattr(dist, "all_reduce_coalesced")


def allreduce_grads(module, scale=None, extra=()):
    """sum-all-reduce every parameter gradient in ONE grouped collective (RCCL group call: no flat copy of the
    P = 4HN+2H+N floats and no copy back -- the gradients are reduced where autograd left them); `scale` (e.g. 1/world
    for a mean loss over the global batch) is applied after the reduction.  `extra`: further tensors summed in the same
    group call (not scaled)."""
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
    group = grads + list(extra)
    # The collective sequence must be the same on every rank: whether the grouped form is used is decided from what
    # the installed torch offers (identical on all ranks), never from an exception raised inside a collective (a rank
    # that fell back alone would issue a different sequence than its peers: hang or double reduction).
    if _has_coalesced():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dist.all_reduce_coalesced(group, op=dist.ReduceOp.SUM)
    else:
        for g in group:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
    if scale is not None:
        torch._foreach_mul_(grads, scale)


def zero_grads_where_failed(module, flag):
    """device-side guard of the deferred mode: where the summed failure flag is non-zero every gradient becomes 0 (no
    host round trip).  A select, not a multiply: the engine writes NaN into outputs a failed solve never reached, so the
    summed gradients of a failed step may hold NaN / Inf, and 0 * NaN is NaN."""
    ok = flag.reshape(()) == 0
    for p in module.parameters():
        if p.requires_grad and p.grad is not None:
            p.grad.copy_(torch.where(ok, p.grad, torch.zeros((), dtype=p.grad.dtype, device=p.grad.device)))


class GradSync:
    """`grad_sync` of `training_step` for data-parallel runs: the gradient all-reduce, plus agreement on failure.

    A solve that fails on ONE rank (max_num_steps, dt underflow, a lost co-residency check) raises there and would
    leave the other ranks waiting in the collective.  training_step hands the exception to this object instead; a
    failure flag rides in the same group call as the gradients, the failing rank re-raises after the collective and
    every other rank raises `RuntimeError` at a point that is the same on all of them: at once when the engine reads
    statuses immediately (or on CPU), else at its next call / `finish()` -- the flag then comes back through pinned
    memory like the deferred solver status (phoenix_amd.engine.set_status_mode) and costs no host synchronisation.

    Deferred mode and the optimizer step: the healthy ranks learn of the failure one call late, i.e. AFTER their
    `opt.step()` of the failed step.  So that they do not apply the incomplete sum, the reduced gradients are replaced
    by zeros where `flag != 0` on the device (`torch.where`: no host round trip, and non-finite sums do not survive as
    0 * NaN would): on a failed step every rank steps with all-zero gradients.  A
    stateless optimizer then changes nothing; one with momentum (Adam) still moves the parameters by its running
    averages on the healthy ranks while the failing rank does not step at all -- after this error the replicas must be
    restored from the last checkpoint on ALL ranks before training continues."""
    collective_errors = True

    def __init__(self, scale=None):
        self.scale = scale
        self._pending = None     # (pinned host flag, event) of the previous call

    def check(self):
        if self._pending is not None:
            host, ev = self._pending
            self._pending = None
            ev.synchronize()
            if float(host[0]) > 0:
                raise RuntimeError("phoenix_amd: a solve failed on another rank in the previous step")

    finish = check

    def __call__(self, module, error=None):
        from . import engine
        self.check()
        dev = next(module.parameters()).device
        flag = torch.full((1,), 0.0 if error is None else 1.0, device=dev)
        allreduce_grads(module, self.scale, extra=[flag])
        if error is not None:
            raise error
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dev.type != "cuda" or engine.status_mode() == "immediate":
            if float(flag[0]) > 0:
                raise RuntimeError("phoenix_amd: a solve failed on another rank")
            return
        zero_grads_where_failed(module, flag)
        host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (host, ev)


def allreduce_scalars(*vals):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vals
    t = torch.stack([v.detach().reshape(()) for v in vals])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(t.unbind(0))
