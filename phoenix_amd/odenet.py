"""Host-side mirror of the reference's RHS model (ode_net/code/odenet.py:38-151).

Same constructor, attribute names, parameter shapes, init and `parameters()` order as the reference
`ODENet`, so a reference training loop (optimizer param groups by attribute, `save`) runs unchanged.
`forward` / `prior_only_forward` run on the HIP engine through torch.autograd.Functions."""
import os

import torch
import torch.nn as nn

from . import engine


class SoftsignMod(nn.Module):
    """placeholder with the reference's module name (odenet.py:16-25); the activation itself is
    fused into the HIP kernels (phx_device.hpp act_pair)."""

    def forward(self, x):
        raise RuntimeError("phoenix_amd: activations are fused into the HIP RHS kernel; call ODENet.forward")


class LogShiftedSoftSignMod(SoftsignMod):
    """odenet.py:27-35"""


def params_of(func):
    """Structural recognition of a PHOENIX ODENet (reference class or this mirror):
    net_sums.linear_out, net_prods.linear_out, net_alpha_combine.linear_out, gene_multipliers."""
    try:
        ws = func.net_sums.linear_out.weight
        bs = func.net_sums.linear_out.bias
        wp = func.net_prods.linear_out.weight
        bp = func.net_prods.linear_out.bias
        wa = func.net_alpha_combine.linear_out.weight
        g = func.gene_multipliers
    except AttributeError:
        raise TypeError("phoenix_amd.odeint accelerates PHOENIX's ODENet only (needs net_sums/net_prods/"
                        "net_alpha_combine.linear_out and gene_multipliers); got %s" % type(func).__name__)
    if wa.shape != (ws.shape[1], 2 * ws.shape[0]) or getattr(func.net_alpha_combine.linear_out, "bias", None) is not None:
        raise TypeError("phoenix_amd: module does not have the PHOENIX ODENet structure")
    return ws, bs, wp, bp, wa, g


class _RhsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, prior_only, ws, bs, wp, bp, wa, g):
        p = engine.params_cached(ws, bs, wp, bp, wa, g)   # one re-layout per parameter version, shared with backward
        ctx.prior_only = prior_only
        ctx.save_for_backward(y, ws, bs, wp, bp, wa, g)
        return engine.rhs_forward(p, y, prior_only)

    @staticmethod
    def backward(ctx, grad_out):
        y, ws, bs, wp, bp, wa, g = ctx.saved_tensors
        p = engine.params_cached(ws, bs, wp, bp, wa, g)
        need_p = any(ctx.needs_input_grad[2:])
        vjp, grads = engine.rhs_vjp(p, y, grad_out.contiguous(), ctx.prior_only, want_grads=need_p,
                                    want_vjp_y=ctx.needs_input_grad[0])
        if need_p:
            gws, gbs, gwp, gbp, gwa, gg = grads.as_reference_layout(g.shape)
        else:
            gws = gbs = gwp = gbp = gwa = gg = None
        return vjp, None, gws, gbs, gwp, gbp, gwa, gg


class _PriorMSEFn(torch.autograd.Function):
    """loss_prior = mean((prior_only_forward(X) - prior_grad)^2) (train_insilico.py:134-135) as one engine call: the
    expansion kernel writes the loss cotangent directly, the prediction is never materialised."""

    @staticmethod
    def forward(ctx, X, target, ws, bs, wp, bp, wa, g):
        p = engine.params_cached(ws, bs, wp, bp, wa, g)
        res = engine.prior_mse(p, X, target, keep_hidden=True)
        if res is None:
            raise RuntimeError("phoenix_amd: the fused prior loss is not available for this shape / engine mode")
        loss, cot, z = res
        ctx.has_z = z is not None          # the forward chain's hidden rows, kept for the backward (no recomputation)
        ctx.save_for_backward(X, cot, ws, bs, wp, bp, wa, g, *([z] if z is not None else []))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        X, cot, ws, bs, wp, bp, wa, g = ctx.saved_tensors[:8]
        p = engine.params_cached(ws, bs, wp, bp, wa, g)
        if ctx.has_z:
            grads = engine.prior_vjp_saved(p, X, cot, ctx.saved_tensors[8])
        else:
            _, grads = engine.rhs_vjp(p, X, cot, True, want_grads=True, want_vjp_y=False)
        grads.flat.mul_(grad_out)          # the cotangent was formed for d loss = 1
        gws, gbs, gwp, gbp, gwa, gg = grads.as_reference_layout(g.shape)
        return None, None, gws, gbs, gwp, gbp, gwa, gg


class ODENet(nn.Module):
    """ODE-Net (reference odenet.py:38-98)."""

    def __init__(self, device, ndim, explicit_time=False, neurons=100):
        super().__init__()
        self.ndim = ndim
        self.explicit_time = explicit_time
        # same construction order as the reference so that parameters() enumerates identically
        self.net_prods = nn.Sequential()
        self.net_prods.add_module("activation_0", LogShiftedSoftSignMod())
        self.net_prods.add_module("linear_out", nn.Linear(ndim, neurons, bias=True))
        self.net_sums = nn.Sequential()
        self.net_sums.add_module("activation_0", SoftsignMod())
        self.net_sums.add_module("linear_out", nn.Linear(ndim, neurons, bias=True))
        self.net_alpha_combine = nn.Sequential()
        self.net_alpha_combine.add_module("linear_out", nn.Linear(2 * neurons, ndim, bias=False))
        self.gene_multipliers = nn.Parameter(torch.rand(1, ndim), requires_grad=True)
        for seq in (self.net_sums, self.net_prods, self.net_alpha_combine):   # odenet.py:64-75
            nn.init.sparse_(seq.linear_out.weight, sparsity=0.95, std=0.05)
        self.to(device)   # the reference's .to(device) of gene_multipliers is a no-op bug (odenet.py:80); do it right

    def forward(self, t, y):
        """odenet.py:85-91 (t is ignored: autonomous system)"""
        return _RhsFn.apply(y, False, *params_of(self))

    def prior_only_forward(self, t, y):
        """odenet.py:93-98"""
        return _RhsFn.apply(y, True, *params_of(self))

    def prior_mse(self, t, batch_for_prior, prior_grad):
        """`torch.mean((self.prior_only_forward(t, batch_for_prior) - prior_grad) ** 2)` (train_insilico.py:134-135);
        fused on the engine for large batches whose input needs no gradient, the plain formula otherwise."""
        fused_ok = (batch_for_prior.is_cuda and not batch_for_prior.requires_grad and not prior_grad.requires_grad and
                    batch_for_prior.shape == prior_grad.shape and
                    batch_for_prior.numel() // self.ndim >= engine.PRIOR_MSE_MIN_ROWS and
                    self.net_sums.linear_out.weight.shape[0] <= 256 and
                    os.environ.get("PHX_ENGINE") != "v0" and os.environ.get("PHX_PGRAD") != "v1")
        if fused_ok:
            return _PriorMSEFn.apply(batch_for_prior, prior_grad, *params_of(self))
        return torch.mean((self.prior_only_forward(t, batch_for_prior) - prior_grad) ** 2)

    def save(self, fp):
        """four-file pickle of whole modules, as the reference (odenet.py:100-111)"""
        idx = fp.index(".")
        torch.save(self.net_prods, fp[:idx] + "_prods" + fp[idx:])
        torch.save(self.net_sums, fp[:idx] + "_sums" + fp[idx:])
        torch.save(self.net_alpha_combine, fp[:idx] + "_alpha_comb" + fp[idx:])
        torch.save(self.gene_multipliers, fp[:idx] + "_gene_multipliers" + fp[idx:])

    def load_model(self, fp):
        """odenet.py:118-133 (without the forced .to('cpu'))"""
        idx = fp.index(".pt")
        dev = self.gene_multipliers.device
        self.net_prods = torch.load(fp[:idx] + "_prods" + fp[idx:], weights_only=False).to(dev)
        self.net_sums = torch.load(fp[:idx] + "_sums" + fp[idx:], weights_only=False).to(dev)
        self.gene_multipliers = nn.Parameter(torch.load(fp[:idx] + "_gene_multipliers" + fp[idx:],
                                                        weights_only=False).to(dev))
        self.net_alpha_combine = torch.load(fp[:idx] + "_alpha_comb" + fp[idx:], weights_only=False).to(dev)

    def load(self, fp):
        self.load_model(fp)
