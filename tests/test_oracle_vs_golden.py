"""Pins the CPU oracle (oracle/phx_oracle.c) against goldens captured from the reference itself
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, net_from, relerr, sub

KEYS = ("Ws", "bs", "Wp", "bp", "Wa", "g")
# fp32 paths that differ only in summation order (MKL vs plain C): a few ulp on O(N) sums
TOL_RHS = 5e-6
TOL_FIXED = 5e-6
# north_star tolerance on trajectories of converged adaptive solves: 1e-5 relative
TOL_DOPRI = 1e-5
# gradients of the adaptive adjoint: rtol=1e-7 is below fp32 epsilon, so accept/reject decisions are
# rounding noise and two fp32 implementations land on different step sequences.  Measured here
# (DESIGN.md "parity tolerances"): the reference's own fp32 gradient is 4e-6..1.8e-5 from the fp64
# tight-tolerance truth when rtol is jittered by +-10%; the oracle spans 5e-6..2.6e-5.
TOL_DOPRI_GRAD = 5e-5


@pytest.mark.parametrize("case", ["sparse", "dense", "odd"])
def test_g1_rhs(oracle, case):
    g = sub(load_golden("g1_g2_rhs"), case + "/")
    net = net_from(oracle, g)
    assert relerr(oracle.rhs(net, g["y"]), g["f"]) < TOL_RHS
    assert relerr(oracle.rhs(net, g["y"], prior_only=True), g["f_prior"]) < TOL_RHS
    assert relerr(oracle.rhs(net, g["y1"]), g["f1"]) < TOL_RHS


@pytest.mark.parametrize("case", ["sparse", "dense", "odd"])
def test_g2_rhs_vjp(oracle, case):
    g = sub(load_golden("g1_g2_rhs"), case + "/")
    net = net_from(oracle, g)
    vjp, grads, f = oracle.rhs_vjp(net, g["y"], g["cot"])
    assert relerr(f, g["f"]) < TOL_RHS
    assert relerr(vjp, g["vjp_y"]) < TOL_RHS
    for k in KEYS:
        assert relerr(grads[k], g["vjp_" + k]) < TOL_RHS, k
    vjp, grads, f = oracle.rhs_vjp(net, g["y"], g["cot"], prior_only=True)
    assert relerr(vjp, g["vjp_y_prior"]) < TOL_RHS
    for k in KEYS:
        ref = g["vjpprior_" + k]
        if np.max(np.abs(ref)) == 0:
            assert np.max(np.abs(grads[k])) == 0
        else:
            assert relerr(grads[k], ref) < TOL_RHS, k


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("tname", ["t2", "t5", "t5_64", "t_dec"])
@pytest.mark.parametrize("yname", ["single", "batch"])
def test_g3_fixed(oracle, method, tname, yname):
    g = load_golden("g3_fixed")
    net = net_from(oracle, g)
    c = sub(g, "%s/%s/%s/" % (method, tname, yname))
    y0 = g["y0_" + yname]
    sol = oracle.odeint(net, y0, g[tname], method=method)
    assert relerr(sol, c["sol"]) < TOL_FIXED
    if "G" in c:
        adj, grads = oracle.adjoint_backward(net, g[tname], c["sol"], c["G"], method=method)
        assert relerr(adj, c["grad_y0"]) < TOL_FIXED
        for k in KEYS:
            assert relerr(grads[k], c["grad_" + k]) < TOL_FIXED, k


@pytest.mark.parametrize("tname", ["t2", "t4", "t10_64", "t_dec"])
@pytest.mark.parametrize("yname", ["single", "batch"])
def test_g4_dopri5(oracle, tname, yname):
    g = load_golden("g4_dopri5")
    net = net_from(oracle, g)
    c = sub(g, "%s/%s/" % (tname, yname))
    sol = oracle.odeint(net, g["y0_" + yname], g[tname], method="dopri5")
    assert relerr(sol, c["sol"]) < TOL_DOPRI
    # and both agree with the fp64 tight-tolerance truth
    assert relerr(sol, c["truth64"]) < TOL_DOPRI
    if "G" in c:
        for theta_in_norm in (True, False):
            adj, grads = oracle.adjoint_backward(net, g[tname], c["sol"], c["G"], method="dopri5",
                                                 theta_in_norm=theta_in_norm)
            assert relerr(adj, c["grad_y0"]) < TOL_DOPRI_GRAD
            for k in KEYS:
                assert relerr(grads[k], c["grad_" + k]) < TOL_DOPRI_GRAD, (k, theta_in_norm)


def test_g4_per_sample_loop(oracle):
    g = load_golden("g4_dopri5")
    net = net_from(oracle, g)
    c = sub(g, "loop/")
    y0 = g["y0_batch"].reshape(4, -1)
    sol = oracle.odeint_per_sample(net, y0, c["t"], method="dopri5", nthreads=1)
    assert relerr(sol[:, 1], c["pred"].reshape(4, -1)) < TOL_DOPRI
    gy = np.zeros_like(sol)
    gy[:, 1] = c["G"].reshape(4, -1)
    for theta_in_norm in (True, False):
        adj, grads = oracle.adjoint_backward_per_sample(net, c["t"], sol, gy, method="dopri5",
                                                        theta_in_norm=theta_in_norm, nthreads=1)
        assert relerr(adj, c["grad_y0"].reshape(4, -1)) < TOL_DOPRI_GRAD
        for k in KEYS:
            assert relerr(grads[k], c["grad_" + k]) < TOL_DOPRI_GRAD, k


def test_g6_controller(oracle):
    g = load_golden("g6_controller")
    assert abs(oracle.rms_norm(g["norm/x"]) - g["norm/rms"]) < 1e-6 * g["norm/rms"]
    assert abs(oracle.mixed_norm(g["norm/x"], g["norm/blocks"]) - g["norm/mixed"]) < 1e-6 * g["norm/mixed"]
    for er, want in zip(g["step/error_ratio"], g["step/dt_next"]):
        got = oracle.optimal_step_size(0.125, float(er))
        assert abs(got - want) <= 1e-12 * abs(want), (er, got, want)
    coef = oracle.interp_fit(g["interp/y0"], g["interp/y1"], g["interp/ym"], g["interp/f0"], g["interp/f1"],
                             float(g["interp/dt"]))
    assert relerr(coef, g["interp/coef"]) < 2e-6
    for tt, want in zip(g["interp/ts"], g["interp/vals"]):
        assert relerr(oracle.interp_eval(g["interp/coef"], 1.0, 1.37, float(tt)), want) < 2e-6
    # error ratio == rms(err / (atol + rtol*max|y|))
    rt, at = np.float32(1e-7), np.float32(1e-9)
    tol = at + rt * np.maximum(np.abs(g["ratio/y0"]), np.abs(g["ratio/y1"]))
    got = oracle.rms_norm((g["ratio/err"] / tol).astype(np.float32))
    assert abs(got - g["ratio/value"]) < 1e-5 * g["ratio/value"]


def test_g6_initial_step_via_nfe(oracle):
    """_select_initial_step is internal to the oracle's dopri5; pin it through a solve whose first
    step is the initial step: t1 - t0 far below h => exactly one accepted step of size h."""
    g = load_golden("g6_controller")
    net = net_from(oracle, g, "init/p_")
    h = float(g["init/h"])
    sol, nfe, nsteps = oracle.odeint(net, g["init/y0"], np.array([0.0, h * 0.5]), method="dopri5",
                                     return_stats=True)
    assert nfe == 2 + 6 * nsteps and nsteps >= 1


@pytest.mark.parametrize("name", ["yeast", "breast"])
def test_g7_realdata(oracle, name):
    g = sub(load_golden("g7_realdata"), name + "/")
    net = net_from(oracle, g)
    for i in range(2):
        c = sub(g, "pair%d/" % i)
        y0 = g["Y"][i:i + 1]
        t = g["t"][i:i + 2]
        sol = oracle.odeint(net, y0, t, method="dopri5")
        assert relerr(sol, c["sol"]) < TOL_DOPRI
        adj, grads = oracle.adjoint_backward(net, t, c["sol"], c["G"], method="dopri5")
        assert relerr(adj, c["grad_y0"]) < TOL_DOPRI_GRAD
        for k in KEYS:
            assert relerr(grads[k], c["grad_" + k]) < TOL_DOPRI_GRAD, k
