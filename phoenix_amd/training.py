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
    try:
        # reference: python loop of odeint(odenet, batch_point, time)[1]; here one launch, per-sample control
        predictions = odeint_adjoint(odenet, batch, t, method=method)[1]
        loss_data = torch.mean((predictions - target) ** 2)
        if hasattr(odenet, "prior_mse"):     # phoenix_amd.ODENet: fused on the engine (no [K,1,N] prediction tensor)
            loss_prior = odenet.prior_mse(t, batch_for_prior, prior_grad)
        else:                                # the reference's own ODENet class
            pred_grad = odenet.prior_only_forward(t, batch_for_prior)
            loss_prior = torch.mean((pred_grad - prior_grad) ** 2)
        composed_loss = loss_lambda * loss_data + (1 - loss_lambda) * loss_prior
        composed_loss.backward()
    except (AssertionError, RuntimeError) as err:   # a failed solve (the reference's asserts, rk_common.py:174-176) or launch
        if not getattr(grad_sync, "collective_errors", False):
            raise
        failure = err                        # the other ranks are on their way into the collective: meet them there
    if grad_sync is not None:
        if failure is not None:
            grad_sync(odenet, error=failure)
        else:
            grad_sync(odenet)
    opt.step()
    return [loss_data, loss_prior]
