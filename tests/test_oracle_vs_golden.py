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
# rounding noise and two fp32 implementations land on different step sequences.  Measured (golden G12, captured
# from the reference): its own fp32 gradient is 2.6e-6..1.3e-5 (median 7.7e-6) from the fp64 tight-tolerance truth
# when rtol is jittered by +-15 %, so two such gradients differ by up to ~2.5e-5; test_g12_* below holds the oracle
# to the reference's own distance from the truth, this constant bounds oracle-vs-reference differences.
TOL_DOPRI_GRAD = 3e-5


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


def test_g12_gradients_are_as_close_to_the_truth_as_the_references_own(oracle):
    """G12: the reference's own fp32 gradients at rtol * {0.85 ... 1.15} against an fp64 tight-tolerance truth of the
    same six G4 problems: the noise floor of this solver at rtol = 1e-7 < fp32 epsilon, MEASURED (median 7.7e-6, max
    1.3e-5 relative).  The oracle must sit in the same distribution: median no worse than 1.5 x the reference's median (six cases against forty-two reference runs),
    worst case within 2 x the reference's worst."""
    g, sp = load_golden("g4_dopri5"), load_golden("g12_spread")
    net = net_from(oracle, g)
    ref_err, our_err = [], []
    for tname in ("t2", "t4", "t_dec"):
        for yname in ("single", "batch"):
            c, s = sub(g, "%s/%s/" % (tname, yname)), sub(sp, "%s/%s/" % (tname, yname))
            names = ["grad_y0"] + ["grad_" + k for k in KEYS]
            ref_err += [max(relerr(s["jit%d/%s" % (j, n)], s["truth64/" + n]) for n in names) for j in range(7)]
            adj, grads = oracle.adjoint_backward(net, g[tname], c["sol"], c["G"], method="dopri5", theta_in_norm=False)
            our_err.append(max([relerr(adj, s["truth64/grad_y0"])] + [relerr(grads[k], s["truth64/grad_" + k]) for k in KEYS]))
    assert np.median(our_err) <= 1.5 * np.median(ref_err), (np.median(our_err), np.median(ref_err))
    assert max(our_err) <= 2.0 * max(ref_err), (max(our_err), max(ref_err))


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


def _fullsize_case(g):
    """inputs of goldens G14 / G15 in the per-sample form: y0 [B,N], t [B,2], the training loss's cotangent [B,2,N]"""
    Y, tt = g["Y"], g["t"]
    B, N = Y.shape[0] - 1, Y.shape[1]
    t = np.stack([tt[:-1], tt[1:]], 1).astype(np.float32)
    G = np.zeros((B, 2, N), np.float32)
    G[:, 1] = (2.0 * (g["pred"].astype(np.float64) - Y[1:]) / (B * N)).astype(np.float32)   # d mean((pred - target)^2) / d pred
    return Y[:-1].copy(), t, G, B, N


def _sampled_grad_err(g, got, k):
    """max error of a weight gradient at the golden's 4 096 stored positions, normalised by the reference tensor's max"""
    return float(np.max(np.abs(got.reshape(-1)[g["gidx_" + k]].astype(np.float64) - g["gval_" + k]))) / float(g["gmax_" + k])


@pytest.mark.parametrize("name,gtol", [("g14_breast11165", TOL_DOPRI), ("g15_yeast3551", TOL_DOPRI_GRAD)])
def test_g14_g15_full_size_reference_runs_on_shipped_data(oracle, name, gtol):
    """The reference ITSELF at full size (N = 11 165, H = 40: seven pairs of the shipped breast-cancer test sample, one
    accepted step each; N = 3 551, H = 120: the 23 pairs of the shipped yeast sample, ~27 steps each): predictions,
    loss, dL/dy0 of every pair, bias / gene-multiplier gradients in full, weight gradients at 4 096 positions."""
    g = load_golden(name)
    net = net_from(oracle, g)
    y0, t, G, B, N = _fullsize_case(g)
    sol = oracle.odeint_per_sample(net, y0, t, method="dopri5", nthreads=8)
    assert relerr(sol[:, 1], g["pred"]) < TOL_DOPRI
    assert abs(float(np.mean((sol[:, 1].astype(np.float64) - g["Y"][1:]) ** 2)) - float(g["loss"])) < 1e-5 * float(g["loss"])
    adj, grads = oracle.adjoint_backward_per_sample(net, t, sol, G, method="dopri5", theta_in_norm=True, nthreads=8)
    assert relerr(adj, g["grad_y0"]) < gtol
    for k in ("bs", "bp", "g"):
        assert relerr(grads[k].reshape(-1), g["grad_" + k].reshape(-1)) < gtol, k
    for k in ("Ws", "Wp", "Wa"):
        assert _sampled_grad_err(g, grads[k], k) < gtol, k


@pytest.mark.parametrize("name,N,H,t1,sparse", [("yeast-like C3", 2000, 120, 5.0, True), ("breast-like C4", 11165, 40, 0.0051, False)])
def test_theta_block_of_the_adjoint_norm_is_inert_at_training_scale(oracle, name, N, H, t1, sparse):
    """adjoint.py:72-78 puts a fourth block -- the accumulated parameter gradient -- into the mixed Linf/RMS norm of
    the backward solve; the engine's controllers see [t, y, a] only (a per-trajectory P-sized state would be needed).
    MEASUREMENT behind that choice, at full C3 / C4 size with the oracle, which implements both: with the cotangent of
    the training loss (mean over B*N elements, train_insilico.py:132: |dL/dy| ~ 1/(B N)) the theta block never wins
    the max -- step sequence AND results are bit-identical with and without it, because atol = 1e-9 dominates the
    tolerance of every adjoint-sized quantity.  (With O(1) cotangents it does win: a few more steps, results within
    5e-6 of each other -- see DESIGN.md; the B-cell shape shows 1e-7 even at training scale.)"""
    r = np.random.RandomState(10)
    std = 0.05 if sparse else 0.02
    p = {"Ws": r.randn(H, N) * std, "bs": r.randn(H) * 0.1, "Wp": r.randn(H, N) * std, "bp": r.randn(H) * 0.1,
         "Wa": r.randn(N, 2 * H) * std, "g": r.rand(1, N)}
    if sparse:
        for k in ("Ws", "Wp", "Wa"):
            p[k] = p[k] * (r.rand(*p[k].shape) < 0.05)
    p = {k: v.astype(np.float32) for k, v in p.items()}
    net = oracle.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
    B = 2
    y0 = (np.clip(r.randn(B, N) * 0.4, -2.5, 4.0) if sparse else np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07)).astype(np.float32)
    t = np.tile(np.array([[0.0, t1]], np.float32), (B, 1))
    ref = oracle.odeint_per_sample(net, y0, t, method="dopri5")
    G = np.zeros((B, 2, N), np.float32)
    G[:, 1] = 2.0 * r.randn(B, N) * 0.1 / (256 * N)          # d mean((pred - target)^2) / d pred for a 256-sample batch
    out = {}
    for th in (True, False):
        out[th] = oracle.adjoint_backward_per_sample(net, t, ref, G, method="dopri5", theta_in_norm=th, return_stats=True)
    assert out[True][2] == out[False][2] and out[True][3] == out[False][3]           # nfe, nsteps
    assert np.array_equal(out[True][0], out[False][0])
    for k in KEYS:
        assert np.array_equal(out[True][1][k], out[False][1][k]), k
