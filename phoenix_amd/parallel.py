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


def global_mean_weights(b_local, k_local, device=None):
    """(B_local / B_global, K_local / K_global): the factors that turn a rank's local `torch.mean` losses into its share of
    the reference's means over the WHOLE batch (train_insilico.py:132,136) when the ranks' shards are not equal (23 yeast
    pairs over 8 ranks).  One small SUM all-reduce of the two counts; every rank must call it in the same step."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0, 1.0
    cnt = torch.tensor([float(b_local), float(k_local)], dtype=torch.float64, device=device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    bg, kg = cnt.tolist()
    return (b_local / bg if bg > 0 else 0.0), (k_local / kg if kg > 0 else 0.0)


def _reduce_list(tensors, async_op=False):
    """SUM all-reduce of a list of tensors in one grouped call where torch offers it (same decision on every rank)"""
    if _has_coalesced():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return dist.all_reduce_coalesced(tensors, op=dist.ReduceOp.SUM, async_op=async_op)
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op) for g in tensors]
    if not async_op:
        return None

    class _All:
        def wait(self):
            for w in works:
                w.wait()
    return _All()


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

    def __init__(self, scale=None, global_batch=None, global_prior=None, weighted=False, overlap=False):
        """`scale`: factor applied to the summed gradients (1 / world for EQUAL shards of a mean loss).
        Unequal shards -- the reference's mean is over the global batch (train_insilico.py:132) -- need per-rank weights
        instead: `global_batch` / `global_prior` (the global numbers of trajectories and prior rows, when the caller knows
        them) or `weighted=True` (the counts are summed over the ranks every step, one small collective): training_step
        then scales each rank's two losses by B_local / B_global and K_local / K_global and the gradients are SUMMED
        (`scale` must stay None).
        `overlap=True`: training_step differentiates the two losses separately and the all-reduce of the data-loss
        gradients runs (async, on the process group's stream) while the prior branch's backward still computes; the two
        reduced gradient sets are added afterwards."""
        self.scale = scale
        self.global_batch, self.global_prior, self.weighted = global_batch, global_prior, weighted
        self.overlap = overlap
        if (weighted or global_batch is not None) and scale is not None:
            raise ValueError("GradSync: per-rank loss weights already make the sum of the gradients the global gradient; "
                             "scale must be None")
        self._pending = None     # (pinned host flag, event) of the previous call

    def loss_weights(self, b_local, k_local, device=None):
        """factors of this rank's (data, prior) mean losses; (1, 1) for equal shards (`scale` does the averaging)"""
        if self.global_batch is not None:
            kg = self.global_prior if self.global_prior is not None else k_local
            return b_local / float(self.global_batch), (k_local / float(kg) if kg else 0.0)
        if self.weighted:
            return global_mean_weights(b_local, k_local, device)
        return 1.0, 1.0

    def check(self):
        if self._pending is not None:
            host, ev = self._pending
            self._pending = None
            ev.synchronize()
            if float(host[0]) > 0:
                raise RuntimeError("phoenix_amd: a solve failed on another rank in the previous step")

    finish = check

    def reduce_async(self, grads):
        """starts the SUM all-reduce of a list of gradient tensors (stream ordered behind the kernels that wrote them) and
        returns a handle whose wait() orders the current stream behind it; None without a process group"""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        return _reduce_list([g for g in grads if g is not None], async_op=True)

    def __call__(self, module, error=None, reduced=False):
        """`reduced=True`: the gradients in `.grad` are already summed over the ranks (the overlapped path of
        training_step): only the failure flag travels"""
        from . import engine
        self.check()
        dev = next(module.parameters()).device
        flag = torch.full((1,), 0.0 if error is None else 1.0, device=dev)
        if reduced:
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        else:
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
