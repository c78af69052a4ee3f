"""The reference's `training_step` (ode_net/code/train_insilico.py:124-140, = train_breast.py:161-187)
on the engine: same inputs, same two losses, same composed gradient, same optimizer step -- but the
B per-sample `odeint_adjoint` calls are one launch and the prior branch runs on the HIP RHS kernels."""
import torch

from .odeint import odeint_adjoint


def training_step(odenet, data_handler, opt, method, batch_size, explicit_time, relative_error, batch_for_prior,
                  prior_grad, loss_lambda, grad_sync=None):
    """`grad_sync`: optional callable run between backward() and opt.step() (data-parallel all-reduce,
    phoenix_amd.parallel.allreduce_grads, or a phoenix_amd.parallel.GradSync, which also makes every rank fail
    together when one rank's solve fails); everything else is the reference signature."""
    batch, t, target = data_handler.get_batch(batch_size)            # [B,1,N], [B,2], [B,1,N]
    opt.zero_grad()
    failure = None
    # Data-parallel shards of unequal size: the reference's losses are means over the WHOLE batch (train_insilico.py:132,136),
    # so a rank's local means enter with B_local / B_global and K_local / K_global and the gradients are summed.  Asked
    # before the solve: a count collective (GradSync(weighted=True)) must be met by every rank, also one whose solve fails.
    wd = wp = 1.0
    if grad_sync is not None and hasattr(grad_sync, "loss_weights"):
        wd, wp = grad_sync.loss_weights(batch.shape[0], batch_for_prior.shape[0], batch.device)
    overlap = bool(getattr(grad_sync, "overlap", False))
    issued = [0]          # gradient collectives the overlapped path has started (a failing rank tops them up below)
    try:
        # reference: python loop of odeint(odenet, batch_point, time)[1]; here one launch, per-sample control
        predictions = odeint_adjoint(odenet, batch, t, method=method)[1]
        loss_data = torch.mean((predictions - target) ** 2)
        if hasattr(odenet, "prior_mse"):     # phoenix_amd.ODENet: fused on the engine (no [K,1,N] prediction tensor)
            loss_prior = odenet.prior_mse(t, batch_for_prior, prior_grad)
        else:                                # the reference's own ODENet class
            pred_grad = odenet.prior_only_forward(t, batch_for_prior)
            loss_prior = torch.mean((pred_grad - prior_grad) ** 2)
        if overlap:
            _backward_overlapped(odenet, grad_sync, loss_lambda * loss_data if wd == 1.0 else (loss_lambda * wd) * loss_data,
                                 (1 - loss_lambda) * loss_prior if wp == 1.0 else ((1 - loss_lambda) * wp) * loss_prior,
                                 issued)
        elif wd == 1.0 and wp == 1.0:
            composed_loss = loss_lambda * loss_data + (1 - loss_lambda) * loss_prior
            composed_loss.backward()
        else:
            composed_loss = (loss_lambda * wd) * loss_data + ((1 - loss_lambda) * wp) * loss_prior
            composed_loss.backward()
    except (AssertionError, RuntimeError) as err:   # a failed solve (the reference's asserts, rk_common.py:174-176) or launch
        if not getattr(grad_sync, "collective_errors", False):
            raise
        failure = err                        # the other ranks are on their way into the collective: meet them there
    if grad_sync is not None:
        if overlap:
            # every rank issues the same sequence of collectives: a rank whose step failed part-way sends zeros for the
            # gradient reductions it did not reach (the flag below then stops every rank in this step)
            while failure is not None and issued[0] < 2:
                h = grad_sync.reduce_async([torch.zeros_like(p) for p in odenet.parameters() if p.requires_grad])
                issued[0] += 1
                if h is not None:
                    h.wait()
            grad_sync(odenet, error=failure, reduced=True)
        elif failure is not None:
            grad_sync(odenet, error=failure)
        else:
            grad_sync(odenet)
    opt.step()
    return [loss_data, loss_prior]


def _backward_overlapped(odenet, grad_sync, term_data, term_prior, issued):
    """The two loss terms are differentiated separately into separate gradient buffers: the all-reduce of the data-loss
    gradients (ready once the backward solve and its reduction kernel have run) is in flight while the prior branch's
    backward chain computes; both sets are summed over the ranks, then added into `.grad`.  The persistent solve kernels
    never share the device with the collective: the next step's forward solve is ordered behind both waits."""
    params = [p for p in odenet.parameters() if p.requires_grad]
    g_data = torch.autograd.grad(term_data, params, allow_unused=True)
    g_data = [g.contiguous() if g is not None else torch.zeros_like(p) for g, p in zip(g_data, params)]
    h_data = grad_sync.reduce_async(g_data)
    issued[0] += 1
    g_prior = torch.autograd.grad(term_prior, params, allow_unused=True)
    g_prior = [g.contiguous() if g is not None else torch.zeros_like(p) for g, p in zip(g_prior, params)]
    h_prior = grad_sync.reduce_async(g_prior)
    issued[0] += 1
    for h in (h_data, h_prior):
        if h is not None:
            h.wait()
    scale = getattr(grad_sync, "scale", None)
    for p, a, b in zip(params, g_data, g_prior):
        a.add_(b)
        if scale is not None:
            a.mul_(scale)
        p.grad = a
