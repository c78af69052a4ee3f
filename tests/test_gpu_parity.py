"""GPU parity tests: the HIP engine (through the C ABI, via phoenix_amd) against
  (1) the golden vectors captured from the reference, and
  (2) the CPU oracle on fresh seeded inputs at larger sizes,
plus size-independent properties at BASELINE.json's full sizes.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import GRAD_ERRORS, grad_err, load_golden, net_from, relerr, sub

pytestmark = pytest.mark.gpu

KEYS = ("Ws", "bs", "Wp", "bp", "Wa", "g")
TOL_RHS = 5e-6        # fp32 sums in a different order (tree/shuffle vs MKL)
TOL_FIXED = 1e-5      # north_star tolerance on trajectories
TOL_DOPRI = 1e-5
TOL_DOPRI_GRAD = 2.5e-5  # see tests/test_oracle_vs_golden.py: adaptive fp32 noise floor of the gradients (round 2: 3e-5;
                         # 2e-5 is crossed by the VALU engine and the MFMA engines alike on one multi-step case: 2.2e-5)
TOL_DOPRI_GRAD_1STEP = 1e-5   # intervals one accepted step long (the breast-cancer pseudotime grid): no accept/reject noise


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pa():
    import phoenix_amd
    return phoenix_amd


def make_net(pa, dev, p):
    """phoenix_amd.ODENet carrying the given parameter arrays (reference layouts)"""
    H, N = p["Ws"].shape
    net = pa.ODENet(dev, N, neurons=H)
    with torch.no_grad():
        net.net_sums.linear_out.weight.copy_(torch.from_numpy(p["Ws"]))
        net.net_sums.linear_out.bias.copy_(torch.from_numpy(p["bs"]))
        net.net_prods.linear_out.weight.copy_(torch.from_numpy(p["Wp"]))
        net.net_prods.linear_out.bias.copy_(torch.from_numpy(p["bp"]))
        net.net_alpha_combine.linear_out.weight.copy_(torch.from_numpy(p["Wa"]))
        net.gene_multipliers.copy_(torch.from_numpy(p["g"]).reshape(1, N))
    return net


def grads_of(net):
    def g(p):
        return (torch.zeros_like(p) if p.grad is None else p.grad).detach().cpu().numpy()
    return {"Ws": g(net.net_sums.linear_out.weight), "bs": g(net.net_sums.linear_out.bias),
            "Wp": g(net.net_prods.linear_out.weight), "bp": g(net.net_prods.linear_out.bias),
            "Wa": g(net.net_alpha_combine.linear_out.weight), "g": g(net.gene_multipliers).reshape(-1)}


def zero_grads(net):
    for p in net.parameters():
        p.grad = None


def rand_params(N, H, seed, std=0.1, neg=0.1):
    r = np.random.RandomState(seed)
    g = r.rand(N).astype(np.float32)
    g[r.rand(N) < neg] *= -1
    return {"Ws": (r.randn(H, N) * std).astype(np.float32), "bs": r.uniform(-.2, .2, H).astype(np.float32),
            "Wp": (r.randn(H, N) * std).astype(np.float32), "bp": r.uniform(-.2, .2, H).astype(np.float32),
            "Wa": (r.randn(N, 2 * H) * std).astype(np.float32), "g": g}


def onet_of(oracle, p):
    return oracle.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])


# --------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("case", ["sparse", "dense", "odd"])
def test_g1_g2_rhs_and_vjp(pa, dev, case):
    g = sub(load_golden("g1_g2_rhs"), case + "/")
    net = make_net(pa, dev, sub(g, "p_"))
    y = torch.from_numpy(g["y"]).to(dev)
    t0 = torch.tensor(0.0, device=dev)
    assert relerr(net(t0, y).detach().cpu().numpy(), g["f"]) < TOL_RHS
    assert relerr(net.prior_only_forward(t0, y).detach().cpu().numpy(), g["f_prior"]) < TOL_RHS
    assert relerr(net(t0, torch.from_numpy(g["y1"]).to(dev)).detach().cpu().numpy(), g["f1"]) < TOL_RHS
    cot = torch.from_numpy(g["cot"]).to(dev)
    for prior, pre, fy in ((False, "vjp_", "vjp_y"), (True, "vjpprior_", "vjp_y_prior")):
        zero_grads(net)
        yv = y.clone().requires_grad_(True)
        out = net.prior_only_forward(t0, yv) if prior else net(t0, yv)
        out.backward(cot)
        assert relerr(yv.grad.cpu().numpy(), g[fy]) < TOL_RHS
        got = grads_of(net)
        for k in KEYS:
            ref = g[pre + k]
            if np.max(np.abs(ref)) == 0:
                assert np.max(np.abs(got[k])) == 0, k
            else:
                assert relerr(got[k], ref) < TOL_RHS, (k, prior)


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("tname", ["t2", "t5", "t5_64", "t_dec"])
@pytest.mark.parametrize("yname", ["single", "batch"])
def test_g3_fixed(pa, dev, method, tname, yname):
    g = load_golden("g3_fixed")
    net = make_net(pa, dev, sub(g, "p_"))
    c = sub(g, "%s/%s/%s/" % (method, tname, yname))
    y0 = torch.from_numpy(g["y0_" + yname]).to(dev)
    t = torch.from_numpy(g[tname]).to(dev)
    sol = pa.odeint(net, y0, t, method=method).detach()   # (autograd-tracked like the reference's when parameters require grad)
    assert sol.shape == c["sol"].shape
    assert relerr(sol.cpu().numpy(), c["sol"]) < TOL_FIXED
    if "G" in c:
        zero_grads(net)
        y0r = y0.clone().requires_grad_(True)
        s2 = pa.odeint_adjoint(net, y0r, t, method=method)
        (s2 * torch.from_numpy(c["G"]).to(dev)).sum().backward()
        assert relerr(y0r.grad.cpu().numpy(), c["grad_y0"]) < TOL_FIXED
        got = grads_of(net)
        for k in KEYS:
            assert relerr(got[k], c["grad_" + k]) < TOL_FIXED, k


@pytest.mark.parametrize("tname", ["t2", "t4", "t10_64", "t_dec"])
@pytest.mark.parametrize("yname", ["single", "batch"])
def test_g4_dopri5(pa, dev, tname, yname):
    g = load_golden("g4_dopri5")
    net = make_net(pa, dev, sub(g, "p_"))
    c = sub(g, "%s/%s/" % (tname, yname))
    y0 = torch.from_numpy(g["y0_" + yname]).to(dev)
    t = torch.from_numpy(g[tname]).to(dev)
    sol = pa.odeint(net, y0, t).detach()   # default dopri5 / 1e-7 / 1e-9; shared step control like the reference
    assert relerr(sol.cpu().numpy(), c["sol"]) < TOL_DOPRI
    assert relerr(sol.cpu().numpy(), c["truth64"]) < TOL_DOPRI
    if "G" in c:
        zero_grads(net)
        y0r = y0.clone().requires_grad_(True)
        s2 = pa.odeint_adjoint(net, y0r, t)
        (s2 * torch.from_numpy(c["G"]).to(dev)).sum().backward()
        assert grad_err(y0r.grad.cpu().numpy(), c["grad_y0"]) < TOL_DOPRI_GRAD
        got = grads_of(net)
        for k in KEYS:
            assert grad_err(got[k], c["grad_" + k]) < TOL_DOPRI_GRAD, k


def test_g12_gradients_are_as_close_to_the_truth_as_the_references_own(pa, dev):
    """The engine's dopri5 gradients against the fp64 tight-tolerance truth of the six G4 problems, held to the
    reference's OWN measured distance from that truth (golden G12: its fp32 gradients at rtol * {0.85 ... 1.15};
    rtol = 1e-7 is below fp32 epsilon, so this distance -- median 7.7e-6, max 1.3e-5 -- is the noise floor of the
    algorithm, not of an implementation): median within 1.25 x the reference's median (six cases against forty-two
    reference runs), worst case within 1.5 x its worst (round 2: 1.5 x / 2 x)."""
    g, sp = load_golden("g4_dopri5"), load_golden("g12_spread")
    net = make_net(pa, dev, sub(g, "p_"))
    ref_err, our_err = [], []
    for tname in ("t2", "t4", "t_dec"):
        for yname in ("single", "batch"):
            c, s = sub(g, "%s/%s/" % (tname, yname)), sub(sp, "%s/%s/" % (tname, yname))
            names = ["grad_y0"] + ["grad_" + k for k in KEYS]
            ref_err += [max(relerr(s["jit%d/%s" % (j, n)], s["truth64/" + n]) for n in names) for j in range(7)]
            zero_grads(net)
            y0r = torch.from_numpy(g["y0_" + yname]).to(dev).requires_grad_(True)
            s2 = pa.odeint_adjoint(net, y0r, torch.from_numpy(g[tname]).to(dev))
            (s2 * torch.from_numpy(c["G"]).to(dev)).sum().backward()
            got = grads_of(net)
            our_err.append(max([relerr(y0r.grad.cpu().numpy(), s["truth64/grad_y0"])] +
                               [relerr(got[k], s["truth64/grad_" + k]) for k in KEYS]))
    print("gradient error vs fp64 truth: engine median %.2e max %.2e | reference median %.2e max %.2e" %
          (np.median(our_err), max(our_err), np.median(ref_err), max(ref_err)))
    assert np.median(our_err) <= 1.25 * np.median(ref_err), (np.median(our_err), np.median(ref_err))
    assert max(our_err) <= 1.5 * max(ref_err), (max(our_err), max(ref_err))


def test_g4_per_sample_loop_as_one_launch(pa, dev):
    g = load_golden("g4_dopri5")
    net = make_net(pa, dev, sub(g, "p_"))
    c = sub(g, "loop/")
    y0 = torch.from_numpy(g["y0_batch"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(c["t"]).to(dev)
    pred = pa.odeint_adjoint(net, y0, t)[1]
    assert relerr(pred.detach().cpu().numpy(), c["pred"]) < TOL_DOPRI
    (pred * torch.from_numpy(c["G"]).to(dev)).sum().backward()
    assert grad_err(y0.grad.cpu().numpy(), c["grad_y0"]) < TOL_DOPRI_GRAD
    got = grads_of(net)
    for k in KEYS:
        assert grad_err(got[k], c["grad_" + k]) < TOL_DOPRI_GRAD, k


class _Handler:
    def __init__(self, b, t, y):
        self.b = (b, t, y)
        self.device = b.device

    def get_batch(self, bs):
        return self.b


@pytest.mark.parametrize("method", ["dopri5", "rk4"])
@pytest.mark.parametrize("lam", [0.99, 1.0])
def test_g5_training_step(pa, dev, method, lam):
    g = sub(load_golden("g5_training_step"), "%s/lam%g/" % (method, lam))
    net = make_net(pa, dev, sub(g, "p_"))
    lr = float(g["lr"])
    opt = torch.optim.Adam([
        {"params": net.net_sums.linear_out.weight}, {"params": net.net_sums.linear_out.bias},
        {"params": net.net_prods.linear_out.weight}, {"params": net.net_prods.linear_out.bias},
        {"params": net.net_alpha_combine.linear_out.weight},
        {"params": net.gene_multipliers, "lr": 5 * lr}], lr=lr, weight_decay=0)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    h = _Handler(T("batch"), T("t"), T("target"))
    with torch.no_grad():
        pred = pa.odeint(net, h.b[0], h.b[1], method=method)[1].detach()
    tol = TOL_DOPRI if method == "dopri5" else TOL_FIXED
    gtol = TOL_DOPRI_GRAD if method == "dopri5" else TOL_FIXED
    assert relerr(pred.cpu().numpy(), g["pred"]) < tol
    loss_data, loss_prior = pa.training_step(net, h, opt, method, 4, False, False, T("X"), T("prior_grad"), lam)
    assert abs(loss_data.item() - float(g["loss_data"])) < 1e-5 * abs(float(g["loss_data"]))
    assert abs(loss_prior.item() - float(g["loss_prior"])) < 1e-5 * abs(float(g["loss_prior"]))
    got = grads_of(net)
    for k in KEYS:
        assert relerr(got[k], g["grad_" + k]) < gtol, k
    after = {"Ws": net.net_sums.linear_out.weight, "bs": net.net_sums.linear_out.bias,
             "Wp": net.net_prods.linear_out.weight, "bp": net.net_prods.linear_out.bias,
             "Wa": net.net_alpha_combine.linear_out.weight, "g": net.gene_multipliers.reshape(-1)}
    for k in KEYS:
        # Adam's first step moves every touched weight by +-lr regardless of gradient size: compare directly
        assert np.max(np.abs(after[k].detach().cpu().numpy() - g["after_" + k])) < 2.1 * lr * 5, k
        big = np.abs(g["grad_" + k]) > 1e-3 * np.abs(g["grad_" + k]).max()
        d = np.abs(after[k].detach().cpu().numpy() - g["after_" + k])[big]
        assert d.size == 0 or d.max() < 1e-5, k


@pytest.mark.parametrize("N,H", [(350, 40), (777, 64), (1100, 131), (500, 200), (333, 256)])
def test_layout_params_and_pack_weight_images_write_the_same_images(pa, dev, N, H):
    """The two producers of the packed LDS weight images promise the same bytes (include/phoenix_hip.h): `phx_layout_params`
    (what the Python path calls once per parameter version, straight from W_alpha [N, 2H]) and `phx_pack_weight_images` (what a C
    caller without the former, and the library itself when `phx_params.wimg` is NULL, use).  Compared: every live column of
    every row of every (gene block, hidden chunk) image incl. the zero rows and relu(gene_multipliers), for one- and
    two-chunk layers and a ragged last gene block; and WaT against Wa.t().  (The 33rd float of a row is padding nobody reads.)"""
    import ctypes as C
    from phoenix_amd import _lib, engine
    net = make_net(pa, dev, rand_params(N, H, seed=N + H))
    P = engine.Params(*pa.odenet.params_of(net))                      # runs phx_layout_params
    torch.cuda.synchronize()
    nbytes = _lib.load().phx_weight_image_bytes(N, H)
    assert nbytes > 0 and P.wimg is not None and P.wimg.numel() == nbytes
    other = torch.full((nbytes,), 0x7F, dtype=torch.uint8, device=dev)
    assert _lib.load().phx_pack_weight_images(C.byref(P.c), C.c_void_p(other.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    HC = (H + 127) // 128
    Hc = (H + HC - 1) // HC
    HT = 3 if Hc <= 48 else (7 if (HC > 1 and Hc <= 112) else 8)       # solve_ht (phx_engine.hip)
    BF = (((4 * Hc + (16 * HT - Hc)) * 33 + 32) + 3) & ~3              # blk_floats_ch
    nblk = (N + 31) // 32
    a = P.wimg.view(torch.float32).reshape(nblk, HC, BF).cpu().numpy()
    b = other.view(torch.float32).reshape(nblk, HC, BF).cpu().numpy()
    for ch in range(HC):
        hc = min(Hc, H - ch * Hc)
        rows = 4 * hc + (16 * HT - hc)
        ia = a[:, ch, :rows * 33].reshape(nblk, rows, 33)[:, :, :32]
        ib = b[:, ch, :rows * 33].reshape(nblk, rows, 33)[:, :, :32]
        assert np.array_equal(ia, ib), (ch, "weight rows")
        assert np.array_equal(a[:, ch, BF - 32:], b[:, ch, BF - 32:]), (ch, "relu(g)")
        assert not ia[:, 4 * hc:].any(), (ch, "zero rows")
    assert torch.equal(P.WaT, net.net_alpha_combine.linear_out.weight.detach().t().contiguous())


def test_parameter_cache_follows_the_optimizer(pa, dev):
    """The engine caches its layout of the parameters (transposed W_alpha, packed LDS weight images) per parameter
    version.  Five SGD steps of the reference's training_step with the cache, and the same five steps with the cache
    dropped before every engine call, must take the network to bitwise the same parameters -- including a validation
    solve between steps (no_grad, same weights: a cache hit) and a `load_state_dict` in the middle."""
    from phoenix_amd import engine
    g = sub(load_golden("g5_training_step"), "dopri5/lam0.99/")
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    h = _Handler(T("batch"), T("t"), T("target"))

    def train(drop_cache):
        net = make_net(pa, dev, sub(g, "p_"))
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        snapshot = None
        vals = []
        for step in range(5):
            if drop_cache:
                engine.invalidate_params()
            pa.training_step(net, h, opt, "dopri5", 4, False, False, T("X"), T("prior_grad"), 0.99)
            if drop_cache:
                engine.invalidate_params()
            with torch.no_grad():
                vals.append(pa.odeint_adjoint(net, h.b[0], h.b[1], method="dopri5")[1].clone())
            if step == 1:
                snapshot = {k: v.clone() for k, v in net.state_dict().items()}
            if step == 3:
                net.load_state_dict(snapshot)          # back to the weights after step 2
        return [p.detach().clone() for p in net.parameters()], vals

    p1, v1 = train(False)
    p2, v2 = train(True)
    for a, b in zip(p1 + v1, p2 + v2):
        assert torch.equal(a, b)
    assert not torch.equal(v1[0], v1[1])               # the weights did move between the validation solves


@pytest.mark.parametrize("name", ["yeast", "breast"])
def test_g7_realdata(pa, dev, name):
    g = sub(load_golden("g7_realdata"), name + "/")
    net = make_net(pa, dev, sub(g, "p_"))
    for i in range(2):
        c = sub(g, "pair%d/" % i)
        zero_grads(net)
        y0 = torch.from_numpy(g["Y"][i:i + 1]).to(dev).requires_grad_(True)
        t = torch.from_numpy(g["t"][i:i + 2]).to(dev)
        sol = pa.odeint_adjoint(net, y0, t)
        assert relerr(sol.detach().cpu().numpy(), c["sol"]) < TOL_DOPRI
        (sol * torch.from_numpy(c["G"]).to(dev)).sum().backward()
        assert grad_err(y0.grad.cpu().numpy(), c["grad_y0"]) < TOL_DOPRI_GRAD
        got = grads_of(net)
        for k in KEYS:
            assert grad_err(got[k], c["grad_" + k]) < TOL_DOPRI_GRAD, k


@pytest.mark.parametrize("name,gtol", [("g14_breast11165", TOL_DOPRI_GRAD_1STEP), ("g15_yeast3551", TOL_DOPRI_GRAD)])
def test_g14_g15_full_size_reference_runs_on_shipped_data(pa, dev, name, gtol):
    """Goldens G14 / G15: the reference itself (its ODENet with the sparse_ init, its per-sample odeint_adjoint loop, its
    loss_data, train_insilico.py:128-132) at FULL size on the shipped data -- the 7 pairs of the 11 165-gene breast-cancer
    test sample (H = 40, one accepted step each) and the 23 pairs of the 3 551-gene yeast sample (H = 120, dt = 5, ~27
    steps each).  The engine integrates all pairs in one launch (per-sample control) and must reproduce the predictions,
    the loss, dL/dy0 of every pair, the bias / gene-multiplier gradients in full and the weight gradients at the 4 096
    stored positions."""
    g = load_golden(name)
    net = make_net(pa, dev, sub(g, "p_"))
    Y, tt = g["Y"], g["t"]
    B, N = Y.shape[0] - 1, Y.shape[1]
    y0 = torch.from_numpy(Y[:-1].copy()).to(dev).reshape(B, 1, N).requires_grad_(True)
    t = torch.from_numpy(np.stack([tt[:-1], tt[1:]], 1).astype(np.float32)).to(dev)
    target = torch.from_numpy(Y[1:].copy()).to(dev).reshape(B, 1, N)
    zero_grads(net)
    pred = pa.odeint_adjoint(net, y0, t, method="dopri5")[1]
    loss = torch.mean((pred - target) ** 2)
    loss.backward()
    assert relerr(pred.detach().cpu().numpy().reshape(B, N), g["pred"]) < TOL_DOPRI
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert relerr(y0.grad.cpu().numpy().reshape(B, N), g["grad_y0"]) < gtol
    got = grads_of(net)
    for k in ("bs", "bp", "g"):
        assert relerr(got[k].reshape(-1), g["grad_" + k].reshape(-1)) < gtol, k
    for k in ("Ws", "Wp", "Wa"):
        err = float(np.max(np.abs(got[k].reshape(-1)[g["gidx_" + k]].astype(np.float64) - g["gval_" + k]))) / float(g["gmax_" + k])
        assert err < gtol, (k, err)


# --------------------------------------------------------------------------- oracle, larger sizes
@pytest.mark.parametrize("N,H,B", [(350, 40, 64), (2000, 120, 4), (1537, 24, 9), (513, 7, 3)])
def test_rhs_vjp_vs_oracle(pa, dev, oracle, N, H, B):
    p = rand_params(N, H, seed=N + H)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(1)
    y = (r.rand(B, 1, N) * 3 - 1).astype(np.float32)
    cot = r.randn(B, 1, N).astype(np.float32)
    yt = torch.from_numpy(y).to(dev).requires_grad_(True)
    out = net(torch.tensor(0.0), yt)
    vjp_ref, gr_ref, f_ref = oracle.rhs_vjp(onet, y, cot)
    assert relerr(out.detach().cpu().numpy(), f_ref) < TOL_RHS
    out.backward(torch.from_numpy(cot).to(dev))
    assert relerr(yt.grad.cpu().numpy(), vjp_ref) < TOL_RHS
    got = grads_of(net)
    for k in KEYS:
        assert relerr(got[k], gr_ref[k]) < 2 * TOL_RHS, k
    # prior branch
    zero_grads(net)
    out = net.prior_only_forward(torch.tensor(0.0), yt)
    assert relerr(out.detach().cpu().numpy(), oracle.rhs(onet, y, prior_only=True)) < TOL_RHS


@pytest.mark.parametrize("prior_only", [False, True])
@pytest.mark.parametrize("N,H,B", [(11165, 40, 300), (700, 100, 37), (513, 7, 3), (2000, 128, 130),
                                   (700, 200, 37), (1100, 131, 130), (3000, 256, 70)])   # the last three: two hidden chunks
def test_rhs_vjp_kernel_chain_vs_oracle_rows_and_valu_engine(pa, dev, oracle, monkeypatch, N, H, B, prior_only):
    """phx_rhs_vjp with dL/dy (the adjoint.py:116-119 shape on a whole batch) runs as the MFMA kernel chain; checked
    against the oracle on sampled rows (input VJP), against the VALU engine for everything (input VJP, all six
    parameter gradients), for every combination of requested outputs."""
    from phoenix_amd import engine
    p = rand_params(N, H, seed=3 * N + H, std=0.5 / np.sqrt(N))
    net = make_net(pa, dev, p)
    P = engine.params_cached(*pa.odenet.params_of(net))
    r = np.random.RandomState(2)
    y = torch.from_numpy((r.rand(B, N) * 3 - 1).astype(np.float32)).to(dev)
    cot = torch.from_numpy(r.randn(B, N).astype(np.float32)).to(dev)
    vjp, grads = engine.rhs_vjp(P, y, cot, prior_only=prior_only)
    monkeypatch.setenv("PHX_ENGINE", "v0")
    vjp0, grads0 = engine.rhs_vjp(P, y, cot, prior_only=prior_only)
    monkeypatch.delenv("PHX_ENGINE")
    assert relerr(vjp.cpu().numpy(), vjp0.cpu().numpy()) < TOL_RHS
    for k in ("Ws", "bs", "Wp", "bp", "Wa", "g"):
        a, b = getattr(grads, k).cpu().numpy(), getattr(grads0, k).cpu().numpy()
        assert relerr(a, b) < 2 * TOL_RHS or (np.abs(b).max() == 0 and np.abs(a).max() == 0), k
    # f(y) from the same chain (the C entry point's f_out) equals the forward entry point
    fo = torch.empty_like(y)
    v_f, _ = engine.rhs_vjp(P, y, cot, prior_only=prior_only, want_grads=False, f_out=fo)
    assert torch.equal(v_f, vjp)
    assert relerr(fo.cpu().numpy(), engine.rhs_forward(P, y, prior_only=prior_only).cpu().numpy()) < TOL_RHS
    # single outputs equal the combined call
    v_only, none = engine.rhs_vjp(P, y, cot, prior_only=prior_only, want_grads=False)
    assert none is None and torch.equal(v_only, vjp)
    _, g_only = engine.rhs_vjp(P, y, cot, prior_only=prior_only, want_vjp_y=False)
    assert relerr(g_only.Ws.cpu().numpy(), grads.Ws.cpu().numpy()) < 1e-6
    assert relerr(g_only.g.cpu().numpy(), grads.g.cpu().numpy()) < 1e-6 or prior_only
    # oracle on sampled rows (the input VJP of a row does not depend on the other rows)
    if not prior_only:
        rows = [0, B // 2, B - 1]
        onet = onet_of(oracle, p)
        vref, _, _ = oracle.rhs_vjp(onet, y[rows].cpu().numpy().reshape(len(rows), 1, N),
                                    cot[rows].cpu().numpy().reshape(len(rows), 1, N))
        assert relerr(vjp[rows].cpu().numpy(), vref.reshape(len(rows), N)) < TOL_RHS


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
@pytest.mark.parametrize("N,H,B", [(350, 40, 64), (2000, 120, 4), (1100, 16, 5),
                                   (350, 40, 150), (1100, 16, 37)])   # the last two: batches that do not fill their last group
def test_per_sample_solve_and_adjoint_vs_oracle(pa, dev, oracle, method, N, H, B):
    p = rand_params(N, H, seed=7 * N + H, std=0.05)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(2)
    y0 = r.rand(B, N).astype(np.float32)
    # interval lengths 0.4 .. 3.55 (longer ones make the single-step rk4 solve ill-conditioned in fp32)
    t = np.stack([np.array([0.1 * b, 0.1 * b + 0.4 + 0.05 * (b % 64)]) for b in range(B)]).astype(np.float32)
    G = r.randn(B, 2, N).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method=method)                       # [B,2,N]
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method=method, theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method=method)    # [2,B,1,N]
    got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
    tol = TOL_DOPRI if method == "dopri5" else TOL_FIXED
    gtol = TOL_DOPRI_GRAD if method == "dopri5" else TOL_FIXED
    assert relerr(got, ref) < tol
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
    assert relerr(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < gtol
    gg = grads_of(net)
    for k in KEYS:
        assert relerr(gg[k], gr_ref[k]) < gtol, k


def test_shared_control_multi_output_vs_oracle(pa, dev, oracle):
    """validation()/find_gene_influences-style call: y0 [B,1,N], t [10] float64, one controller."""
    N, H, B = 700, 20, 6
    p = rand_params(N, H, seed=5, std=0.08)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    y0 = np.random.RandomState(3).rand(B, 1, N).astype(np.float32) - 0.25
    t = np.arange(0, 1, 0.1)
    ref, nfe_ref, _ = oracle.odeint(onet, y0, t, method="dopri5", return_stats=True)
    sol, nfe, nsteps = pa.odeint(net, torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev), return_stats=True)
    assert sol.shape == (10, B, 1, N)
    assert relerr(sol.cpu().numpy(), ref) < TOL_DOPRI
    assert int(nfe[0]) == 2 + 6 * int(nsteps[0])
    assert abs(int(nfe[0]) - nfe_ref) <= 0.25 * nfe_ref   # same controller => similar step count


# --------------------------------------------------------------------------- edge cases / errors
def test_edge_cases(pa, dev):
    p = rand_params(64, 8, seed=1)
    net = make_net(pa, dev, p)
    y0 = torch.rand(3, 1, 64, device=dev)
    # single time point: solution is y0
    sol = pa.odeint(net, y0, torch.tensor([0.5], device=dev)).detach()
    assert torch.equal(sol[0], y0)
    # y0 row of the output is exactly y0
    sol = pa.odeint(net, y0, torch.tensor([0.0, 0.3, 0.9], device=dev), method="rk4").detach()
    assert torch.equal(sol[0], y0)
    # non-monotone t -> AssertionError (misc.py:114-115)
    with pytest.raises(AssertionError):
        pa.odeint(net, y0, torch.tensor([0.0, 1.0, 0.5], device=dev))
    # NaN state -> the reference trips 'underflow in dt nan' (rk_common.py:175)
    bad = y0.clone()
    bad[1, 0, 5] = float("nan")
    with pytest.raises(AssertionError, match="underflow in dt"):
        pa.odeint(net, bad, torch.tensor([0.0, 1.0], device=dev))
    # max_num_steps
    with pytest.raises(AssertionError, match="max_num_steps"):
        pa.odeint(net, y0, torch.tensor([0.0, 50.0], device=dev), options={"max_num_steps": 2})
    # training path: a failed forward solve surfaces with the forward's message from backward(), where its status
    # is read together with the backward solve's (one round trip per step); per-sample grids too
    for t in (torch.tensor([0.0, 50.0], device=dev), torch.tensor([[0.0, 50.0]] * 3, device=dev)):
        yg = y0.clone().requires_grad_(True)
        sol = pa.odeint_adjoint(net, yg, t, options={"max_num_steps": 2})
        with pytest.raises(AssertionError, match="max_num_steps"):
            sol.sum().backward()
    # ... and the outputs that trajectory never reached are NaN, the others are complete (both engines)
    import os
    from phoenix_amd import _lib, engine
    pp = engine.Params(net.net_sums.linear_out.weight, net.net_sums.linear_out.bias, net.net_prods.linear_out.weight,
                       net.net_prods.linear_out.bias, net.net_alpha_combine.linear_out.weight, net.gene_multipliers)
    t3 = torch.tensor([[0.0, 0.5, 1.0]] * 3, device=dev, dtype=torch.float64)
    for eng in ("v1", "v0"):
        if eng == "v0":
            os.environ["PHX_ENGINE"] = "v0"
        try:
            s3, st3, _, _ = engine.solve_forward(pp, bad.reshape(3, 64).contiguous(), t3, "dopri5", _lib.CTRL_PER_TRAJECTORY,
                                                 1e-7, 1e-9, True, False)
        finally:
            os.environ.pop("PHX_ENGINE", None)
        assert st3.tolist() == [0, 2, 0], eng
        assert torch.isnan(s3[1:, 1]).all() and torch.isfinite(s3[:, [0, 2]]).all(), eng
    yg = bad.clone().requires_grad_(True)
    sol = pa.odeint_adjoint(net, yg, torch.tensor([[0.0, 1.0]] * 3, device=dev))
    with pytest.raises(AssertionError, match="underflow in dt .*trajectory 1"):
        sol.sum().backward()


# --------------------------------------------------------------------------- full-size properties
def test_full_size_breast_properties(pa, dev, oracle):
    """BASELINE config C4 (N=11165, H=40, B=256, dopri5 + adjoint): oracle on a row sample, plus
    size-independent properties (batch invariance, VJP linearity, duplicated rows)."""
    N, H, B = 11165, 40, 256
    p = rand_params(N, H, seed=11, std=0.02)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(4)
    y0 = np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)
    y0[7] = y0[3]                                   # duplicated trajectory
    t = np.tile(np.array([[0.0, 0.0051]], np.float32), (B, 1))
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    tt = torch.from_numpy(t).to(dev)
    sol = pa.odeint_adjoint(net, y0t, tt)
    G = torch.from_numpy(r.randn(2, B, 1, N).astype(np.float32)).to(dev)
    G[:, 7] = G[:, 3]
    (sol * G).sum().backward()
    s = sol.detach()
    assert torch.equal(s[0], y0t.detach())
    assert torch.equal(s[1, 7], s[1, 3]) and torch.equal(y0t.grad[7], y0t.grad[3])
    # oracle on 3 sampled rows (per-sample semantics => rows are independent)
    rows = [0, 100, 255]
    ref = oracle.odeint_per_sample(onet, y0[rows], t[rows], method="dopri5")
    assert relerr(s[:, rows, 0].cpu().numpy().transpose(1, 0, 2), ref) < TOL_DOPRI
    adj_ref, gr_rows = oracle.adjoint_backward_per_sample(onet, t[rows], ref, G[:, rows, 0].cpu().numpy().transpose(1, 0, 2),
                                                          method="dopri5", theta_in_norm=True)
    assert grad_err(y0t.grad[rows, 0].cpu().numpy(), adj_ref) < TOL_DOPRI_GRAD
    full = grads_of(net)
    # parameter gradients at full size against the oracle: the engine on exactly the sampled rows (the quadrature
    # kernel at N = 11165; linearity over sub-batches below ties the 256-row gradient to such pieces)
    zero_grads(net)
    yr = y0t.detach()[rows].clone().requires_grad_(True)
    sr = pa.odeint_adjoint(net, yr, tt[rows])
    (sr * G[:, rows]).sum().backward()
    g_rows = grads_of(net)
    for k in KEYS:
        assert grad_err(g_rows[k], gr_rows[k]) < TOL_DOPRI_GRAD, k
    # batch invariance: a sub-batch gives the same rows
    sub_rows = list(range(16, 48))
    s2 = pa.odeint(net, y0t.detach()[sub_rows], tt[sub_rows]).detach()
    assert relerr(s2[1].cpu().numpy(), s[1, sub_rows].cpu().numpy()) < 2e-6
    # parameter gradients == sum over sub-batches (linearity in the batch)
    acc = {k: np.zeros_like(v) for k, v in full.items()}
    for lo in range(0, B, 64):
        zero_grads(net)
        yy = y0t.detach()[lo:lo + 64].clone().requires_grad_(True)
        ss = pa.odeint_adjoint(net, yy, tt[lo:lo + 64])
        (ss * G[:, lo:lo + 64]).sum().backward()
        for k, v in grads_of(net).items():
            acc[k] += v
    for k in KEYS:
        assert relerr(acc[k], full[k]) < 2e-5, k


@pytest.mark.parametrize("B", [256, 200])
def test_full_size_breast_every_row_vs_oracle(pa, dev, oracle, B):
    """BASELINE config C4 at full size (N=11165, H=40, dopri5 + adjoint, the pseudotime interval of the bench): EVERY
    row of the batch -- trajectories, dL/dy0 -- and the parameter gradient of the whole batch against the oracle running
    the reference's per-sample loop WITH the parameter block in the adjoint's step-control norm (adjoint.py:72-78,
    `theta_in_norm=True`): cotangents at the scale `torch.mean((pred - target)**2)` produces, where the engine's
    deviation (no parameter block on the device) must be inert.  B = 256 is the configuration; B = 200 leaves the last
    batch group with one ragged tile and three padding-only ones."""
    N, H = 11165, 40
    p = rand_params(N, H, seed=11, std=0.02)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(4 + B)
    y0 = np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)
    t = np.tile(np.array([[0.0, 0.0051]], np.float32), (B, 1))
    G = (r.randn(B, 2, N) * (2.0 / (B * N))).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
    got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
    worst_row = np.abs(got - ref).reshape(B, -1).max(1) / np.abs(ref).max()
    assert worst_row.max() < TOL_DOPRI, int(worst_row.argmax())
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
    ga = y0t.grad.cpu().numpy().reshape(B, N)
    worst_row = np.abs(ga - adj_ref).max(1) / np.abs(adj_ref).max()
    assert worst_row.max() < TOL_DOPRI_GRAD_1STEP, int(worst_row.argmax())
    gg = grads_of(net)
    for k in KEYS:
        assert relerr(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD_1STEP, k


def test_bench_inputs_breast_vs_oracle(pa, dev, oracle):
    """The EXACT inputs `bench.py` times at its default workload (C4: `bench.make_problem(WORKLOADS["breast"], seed 0)` --
    trained-like dense N(0, 0.05^2) factors, not the std 0.02 of the other full-size tests -- and the bench's own
    cotangent): 16 rows of the trajectories and of dL/dy0, and all six parameter gradients of the whole 256-row batch,
    against the oracle running the reference's per-sample loop with the parameter block in the adjoint norm."""
    import bench
    wl = bench.WORKLOADS["breast"]
    N, H, B = wl["N"], wl["H"], wl["B"]
    net, y0, t = bench.make_problem(wl, dev, seed=0)
    T = t.shape[1]
    gg = torch.Generator(device="cpu").manual_seed(100)              # bench.py: problem(): rank 0's cotangent
    G = (torch.randn(T, B, 1, N, generator=gg) / (B * N)).to(dev)
    G[0].zero_()
    p = {"Ws": net.net_sums.linear_out.weight, "bs": net.net_sums.linear_out.bias, "Wp": net.net_prods.linear_out.weight,
         "bp": net.net_prods.linear_out.bias, "Wa": net.net_alpha_combine.linear_out.weight, "g": net.gene_multipliers}
    p = {k: v.detach().cpu().numpy().reshape(-1) if k == "g" else v.detach().cpu().numpy() for k, v in p.items()}
    onet = onet_of(oracle, p)
    zero_grads(net)
    y = y0.detach().requires_grad_(True)
    sol = pa.odeint_adjoint(net, y, t, method=wl["method"])
    torch.autograd.backward(sol, G)                                   # the bench's step
    y0n, tn = y0.cpu().numpy().reshape(B, N), t.cpu().numpy()
    Gn = G.cpu().numpy().reshape(T, B, N).transpose(1, 0, 2).copy()
    ref = oracle.odeint_per_sample(onet, y0n, tn, method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, tn, ref, Gn, method="dopri5", theta_in_norm=True)
    rows = list(range(0, B, 16))
    got = sol.detach().cpu().numpy().reshape(T, B, N).transpose(1, 0, 2)
    assert relerr(got[rows], ref[rows]) < TOL_DOPRI
    assert relerr(y.grad.cpu().numpy().reshape(B, N)[rows], adj_ref[rows]) < TOL_DOPRI_GRAD_1STEP
    gnet = grads_of(net)
    for k in KEYS:
        assert relerr(gnet[k], gr_ref[k]) < TOL_DOPRI_GRAD_1STEP, k


def test_full_size_bcell_rows_of_every_group_vs_oracle(pa, dev, oracle):
    """BASELINE config C5 at full size (N=14691, H=200 -> two hidden chunks, B=256, dopri5 + adjoint over [0, 1]): 32 rows
    that cover every batch group and the first and last trajectory tile of each, against the oracle with the parameter
    block in the norm (training-scale cotangents).  The engine runs the whole 256-row batch; its rows are compared one by
    one, and its parameter gradient on exactly those rows (a second launch of the 32 rows) with the oracle's."""
    N, H, B = 14691, 200, 256
    p = rand_params(N, H, seed=21, std=0.004)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(19)
    y0 = np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)
    t = np.tile(np.array([[0.0, 1.0]], np.float32), (B, 1))
    G = (r.randn(B, 2, N) * (2.0 / (B * N))).astype(np.float32)
    rows = sorted(set(list(range(0, 4)) + list(range(60, 68)) + list(range(124, 132)) + list(range(188, 196)) +
                      list(range(252, 256))))
    assert len(rows) == 32
    ref = oracle.odeint_per_sample(onet, y0[rows], t[rows], method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t[rows], ref, G[rows], method="dopri5", theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    tt = torch.from_numpy(t).to(dev)
    Gt = torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)
    sol = pa.odeint_adjoint(net, y0t, tt)
    (sol * Gt).sum().backward()
    got = sol.detach()[:, rows, 0].cpu().numpy().transpose(1, 0, 2)
    assert relerr(got, ref) < TOL_DOPRI
    assert grad_err(y0t.grad[rows, 0].cpu().numpy(), adj_ref) < TOL_DOPRI_GRAD
    zero_grads(net)
    yr = y0t.detach()[rows].clone().requires_grad_(True)
    sr = pa.odeint_adjoint(net, yr, tt[rows])
    (sr * Gt[:, rows]).sum().backward()
    assert grad_err(yr.grad[:, 0].cpu().numpy(), adj_ref) < TOL_DOPRI_GRAD
    gg = grads_of(net)
    # This problem sits where the adjoint's error control is atol-dominated (|a| ~ 2 / (B N) = 5e-7 against atol = 1e-9: a
    # relative tolerance of 2e-3 per step), so the reference algorithm's OWN weight gradients move by 2.4e-5 ... 4e-5 when
    # rtol is changed by 0.1 % ... 10 % (measured in round 5: tools/_exp) -- above TOL_DOPRI_GRAD.  As with
    # golden G12 the engine is held to the reference algorithm's measured sensitivity: the oracle re-run at rtol x 0.9 / 1.1.
    spread = {k: 0.0 for k in KEYS}
    for f in (0.9, 1.1):
        _, gr_j = oracle.adjoint_backward_per_sample(onet, t[rows], ref, G[rows], method="dopri5", theta_in_norm=True,
                                                     rtol=1e-7 * f)
        for k in KEYS:
            spread[k] = max(spread[k], relerr(gr_j[k], gr_ref[k]))
    for k in KEYS:
        assert grad_err(gg[k], gr_ref[k]) < max(TOL_DOPRI_GRAD, 1.25 * spread[k]), (k, spread[k])


def test_full_size_insilico_vs_oracle(pa, dev, oracle):
    """BASELINE config C2 at full size (N=350, H=40, 1024 trajectories x 4 intervals, fused 3/8-rule rk4 + adjoint):
    every row against the oracle (per-sample semantics), gradients summed over the batch."""
    N, H, B = 350, 40, 1024
    p = rand_params(N, H, seed=2, std=0.05)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(3)
    y0 = np.clip(r.beta(2, 2, size=(B, N)) + r.uniform(-0.25, 0.25, size=(1, N)), 0, 1).astype(np.float32)
    t = np.tile(np.array([[0.0, 2.0, 3.0, 7.0, 9.0]], np.float32), (B, 1))
    G = (r.randn(B, 5, N) / (B * N)).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="rk4")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="rk4", theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method="rk4")
    got = sol.detach().cpu().numpy().reshape(5, B, N).transpose(1, 0, 2)
    assert relerr(got, ref) < TOL_FIXED
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(5, B, 1, N).copy()).to(dev)).sum().backward()
    assert relerr(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < TOL_FIXED
    gg = grads_of(net)
    for k in KEYS:
        assert relerr(gg[k], gr_ref[k]) < TOL_FIXED, k


def test_full_size_yeast_vs_oracle(pa, dev, oracle):
    """BASELINE config C3 at full size (N=2000, H=120, all 23 consecutive pairs, dopri5 + adjoint, reference-like
    95 %-sparse weights, yeast value range): every row against the oracle."""
    N, H, B = 2000, 120, 23
    p = rand_params(N, H, seed=8, std=0.05)
    r = np.random.RandomState(10)
    for k in ("Ws", "Wp", "Wa"):
        p[k] = (p[k] * (r.rand(*p[k].shape) < 0.05)).astype(np.float32)      # nn.init.sparse_(0.95) like
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    y0 = np.clip(r.randn(B, N) * 0.4, -2.5, 4.0).astype(np.float32)
    t = np.tile(np.array([[0.0, 5.0]], np.float32), (B, 1))
    G = (r.randn(B, 2, N) / (B * N)).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    # training-scale cotangents: the oracle keeps the reference's parameter block in the adjoint norm (adjoint.py:72-78)
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
    got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
    assert relerr(got, ref) < TOL_DOPRI
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
    assert grad_err(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < TOL_DOPRI_GRAD
    gg = grads_of(net)
    for k in KEYS:
        assert grad_err(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD, k


def test_full_size_bcell_properties(pa, dev, oracle):
    """BASELINE config C5 (N=14691, H=200 -> two hidden chunks, B=256, dopri5 + adjoint) at full size: oracle on two
    sampled rows, duplicated rows, sub-batch invariance, parameter gradients additive over sub-batches."""
    N, H, B = 14691, 200, 256
    p = rand_params(N, H, seed=21, std=0.004)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(9)
    y0 = np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)
    y0[200] = y0[5]
    t = np.tile(np.array([[0.0, 1.0]], np.float32), (B, 1))      # the interval bench.py --workload bcell integrates
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    tt = torch.from_numpy(t).to(dev)
    sol = pa.odeint_adjoint(net, y0t, tt)
    G = torch.from_numpy(r.randn(2, B, 1, N).astype(np.float32)).to(dev)
    G[:, 200] = G[:, 5]
    (sol * G).sum().backward()
    s = sol.detach()
    assert torch.equal(s[0], y0t.detach())
    assert torch.equal(s[1, 200], s[1, 5]) and torch.equal(y0t.grad[200], y0t.grad[5])
    rows = [1, 254]
    ref = oracle.odeint_per_sample(onet, y0[rows], t[rows], method="dopri5")
    assert relerr(s[:, rows, 0].cpu().numpy().transpose(1, 0, 2), ref) < TOL_DOPRI
    adj_ref, gr_rows = oracle.adjoint_backward_per_sample(onet, t[rows], ref, G[:, rows, 0].cpu().numpy().transpose(1, 0, 2),
                                                          method="dopri5", theta_in_norm=True)
    assert grad_err(y0t.grad[rows, 0].cpu().numpy(), adj_ref) < TOL_DOPRI_GRAD
    full = grads_of(net)
    zero_grads(net)                                  # parameter gradients of exactly the sampled rows against the oracle
    yr = y0t.detach()[rows].clone().requires_grad_(True)
    sr = pa.odeint_adjoint(net, yr, tt[rows])
    (sr * G[:, rows]).sum().backward()
    g_rows = grads_of(net)
    for k in KEYS:
        assert grad_err(g_rows[k], gr_rows[k]) < TOL_DOPRI_GRAD, k
    sub_rows = list(range(100, 132))
    s2 = pa.odeint(net, y0t.detach()[sub_rows], tt[sub_rows]).detach()
    assert relerr(s2[1].cpu().numpy(), s[1, sub_rows].cpu().numpy()) < 2e-6
    acc = {k: np.zeros_like(v) for k, v in full.items()}
    for lo in range(0, B, 128):
        zero_grads(net)
        yy = y0t.detach()[lo:lo + 128].clone().requires_grad_(True)
        ss = pa.odeint_adjoint(net, yy, tt[lo:lo + 128])
        (ss * G[:, lo:lo + 128]).sum().backward()
        for k, v in grads_of(net).items():
            acc[k] += v
    for k in KEYS:
        assert relerr(acc[k], full[k]) < 2e-5, k


# --------------------------------------------------------------------------- error behaviour of the step budget
def _stiffish_problem(pa, dev, N=96, H=8, B=3, seed=5):
    p = rand_params(N, H, seed=seed, std=0.5 / np.sqrt(N))
    net = make_net(pa, dev, p)
    y0 = torch.rand(B, 1, N, device=dev)
    return net, y0


def test_forward_failure_raises_under_no_grad(pa, dev):
    """odeint_adjoint inside torch.no_grad() (the reference's validation code, train_insilico.py:77-106) never gets a
    backward pass: a failed forward solve must raise right away even though the parameters require grad."""
    net, y0 = _stiffish_problem(pa, dev)
    t = torch.tensor([0.0, 3.0], device=dev)
    with torch.no_grad():
        with pytest.raises(AssertionError, match="max_num_steps"):
            pa.odeint_adjoint(net, y0, t, method="dopri5", options={"max_num_steps": 2})
        pa.odeint_adjoint(net, y0, t, method="dopri5")      # and the unrestricted call still works
    # grad mode, nothing requires grad: also immediate
    for q in net.parameters():
        q.requires_grad_(False)
    with pytest.raises(AssertionError, match="max_num_steps"):
        pa.odeint_adjoint(net, y0, t, method="dopri5", options={"max_num_steps": 2})


@pytest.mark.parametrize("variant", ["default", "v0", "adj1", "adj2_np2", "adj2_np4", "adj3"])
def test_max_num_steps_is_a_budget_per_output_time(pa, dev, monkeypatch, variant):
    """The reference restarts n_steps in every _advance(next_t) (rk_common.py:152-156) and the adjoint solves interval
    by interval (adjoint.py:136-154): a budget between the largest per-interval count and the total must pass."""
    if variant == "v0":
        monkeypatch.setenv("PHX_ENGINE", "v0")
    elif variant == "adj1":
        monkeypatch.setenv("PHX_ADJ", "v1")
    elif variant == "adj3":
        monkeypatch.setenv("PHX_ADJ", "v3")
    elif variant.startswith("adj2"):
        monkeypatch.setenv("PHX_ADJ", "v2")
        monkeypatch.setenv("PHX_ADJ2_NP", variant[-1])
    net, y0 = _stiffish_problem(pa, dev)
    t3 = torch.tensor([0.0, 1.5, 3.0], device=dev)
    _, _, ns_a = pa.odeint(net, y0, t3[:2], method="dopri5", return_stats=True)
    _, _, ns_tot = pa.odeint(net, y0, t3, method="dopri5", return_stats=True)
    n_a, n_tot = int(ns_a.max()), int(ns_tot.max())
    assert n_tot > n_a + 3, "the fixture should need several steps in each interval"
    budget = max(n_a, n_tot - n_a) + 2           # enough for either interval, not for both
    assert budget < n_tot
    sol = pa.odeint(net, y0, t3, method="dopri5", options={"max_num_steps": budget}).detach()
    assert torch.isfinite(sol).all()
    # the backward solve restarts its count per interval too (adjoint.py:136-154): its total over both intervals, then a
    # budget of 80 % of it (the two intervals are roughly balanced)
    from phoenix_amd import engine, _lib
    pe = engine.params_cached(*pa.odenet.params_of(net))
    y2 = y0.reshape(-1, y0.shape[-1]).contiguous()
    sol, st, _, _ = engine.solve_forward(pe, y2, t3.double(), "dopri5", _lib.CTRL_SHARED, 1e-7, 1e-9, False, 0)
    Gc = torch.ones_like(sol)
    _, _, st_b, _, ns_b = engine.solve_adjoint(pe, t3.double(), sol, Gc, "dopri5", _lib.CTRL_SHARED, 1e-7, 1e-9, False, 0)
    assert int(st.max()) == 0 and int(st_b.max()) == 0
    nb_tot = int(ns_b.max())
    _, _, st_b2, _, ns_b2 = engine.solve_adjoint(pe, t3.double(), sol, Gc, "dopri5", _lib.CTRL_SHARED, 1e-7, 1e-9, False, 0,
                                                 max_num_steps=int(0.8 * nb_tot))
    assert int(st_b2.max()) == 0 and int(ns_b2.max()) == nb_tot
    with pytest.raises(AssertionError, match="max_num_steps"):
        pa.odeint(net, y0, t3, method="dopri5", options={"max_num_steps": min(n_a, n_tot - n_a) - 1})


def test_deferred_status_mode(pa, dev):
    """set_status_mode("deferred"): backward() returns without a device round trip, the same AssertionError comes out
    of a later engine call / check_pending_status(wait=True)."""
    net, y0 = _stiffish_problem(pa, dev)
    t = torch.tensor([0.0, 3.0], device=dev)
    pa.set_status_mode("deferred")
    try:
        y0g = y0.clone().requires_grad_(True)
        out = pa.odeint_adjoint(net, y0g, t, method="dopri5", options={"max_num_steps": 2})
        out.sum().backward()                     # no exception here
        with pytest.raises(AssertionError, match="max_num_steps"):
            pa.check_pending_status(wait=True)
        y0g = y0.clone().requires_grad_(True)    # a healthy step leaves nothing behind
        pa.odeint_adjoint(net, y0g, t, method="dopri5").sum().backward()
        pa.check_pending_status(wait=True)
    finally:
        pa.set_status_mode("immediate")


def test_solve_beside_a_busy_second_stream(pa, dev):
    """The persistent kernels wait for each other's rows, so every workgroup of a launch must become resident.  The
    library refuses a grid the device cannot hold (occupancy query -> PHX_ERR_LAUNCH); work on ANOTHER stream can only
    delay residency until its own workgroups drain.  A solve launched beside a stream that keeps the chip busy with
    GEMMs must finish with the same result, far inside the 5 s spin limit."""
    import time
    N, H, B = 2000, 40, 64
    p = rand_params(N, H, seed=3, std=0.03)
    net = make_net(pa, dev, p)
    y0 = torch.rand(B, 1, N, device=dev)
    t = torch.tensor([[0.0, 0.3]], device=dev).repeat(B, 1)

    def run():
        zero_grads(net)
        yy = y0.clone().requires_grad_(True)
        sol = pa.odeint_adjoint(net, yy, t)
        sol.sum().backward()
        return sol.detach().clone(), yy.grad.clone(), grads_of(net)

    quiet = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    stop = time.perf_counter() + 0.6
    results = []
    with torch.cuda.stream(side):
        for _ in range(40):                      # ~40 x 1.5 ms of 256-workgroup-plus GEMMs queued on the side stream
            a = (a @ a).clamp_(-1, 1)
    t0 = time.perf_counter()
    for _ in range(5):
        results.append(run())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert elapsed < 3.0, elapsed
    for sol, gy, gp in results:
        assert torch.equal(sol, quiet[0]) and torch.equal(gy, quiet[1])
        for k in KEYS:
            assert np.array_equal(gp[k], quiet[2][k]), k


# --------------------------------------------------------------------------- engine variants
@pytest.mark.parametrize("variant", ["v0", "v1_nw1", "v1_nw2", "adj1", "adj2_np2", "adj2_np4", "adj3", "fwd1"])
@pytest.mark.parametrize("method", ["rk4", "dopri5"])
def test_engine_variants_agree_with_oracle(pa, dev, oracle, monkeypatch, variant, method):
    """The v0 (VALU, grid-barrier) kernels remain the fallback for shapes the v1 (MFMA) plan rejects, v1 has several
    workgroup geometries, and the backward solve has three kernels (the second in a four- and an eight-wave form, the
    third for dopri5 with a narrow hidden layer) chosen by shape: every variant must pass the same oracle check."""
    if variant == "v0":
        monkeypatch.setenv("PHX_ENGINE", "v0")
    elif variant == "adj3":
        monkeypatch.setenv("PHX_ADJ", "v3")       # dopri5 only: rk4 falls through to the shape's usual kernel
    elif variant == "fwd1":
        monkeypatch.setenv("PHX_FWD", "v1")       # first-generation forward kernel where the third would run
    elif variant == "adj1":
        monkeypatch.setenv("PHX_ADJ", "v1")
    elif variant.startswith("adj2"):
        monkeypatch.setenv("PHX_ADJ", "v2")
        monkeypatch.setenv("PHX_ADJ2_NP", variant[-1])
    else:
        monkeypatch.setenv("PHX_V1_MAXNW", variant[-1])
    N, H, B = 777, 12, 21
    p = rand_params(N, H, seed=99, std=0.06)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(8)
    y0 = r.rand(B, N).astype(np.float32)
    t = np.stack([np.array([0.05 * b, 0.05 * b + 0.5]) for b in range(B)]).astype(np.float32)
    G = r.randn(B, 2, N).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method=method)
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method=method, theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method=method)
    got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
    tol = TOL_DOPRI if method == "dopri5" else TOL_FIXED
    gtol = TOL_DOPRI_GRAD if method == "dopri5" else TOL_FIXED
    assert relerr(got, ref) < tol
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
    assert relerr(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < gtol
    gg = grads_of(net)
    for k in KEYS:
        assert relerr(gg[k], gr_ref[k]) < gtol, k


def test_repeated_solves_on_a_kept_workspace_are_bitwise_the_first(pa, dev):
    """phx_solve_opts.ws_keep: from the second identical call on, the engine vouches for its cached workspace and the
    third-generation kernels skip the fill of their exchange buffers (two sets used alternately, the idle one cleaned by
    the running launch).  Every repetition -- an odd and an even number of launches since the last fill, after another
    shape used the same workspace, after a launch in which every trajectory failed at once -- must return bitwise what a
    freshly filled workspace returns."""
    from phoenix_amd import _lib, engine
    N, H = 2300, 24
    p = rand_params(N, H, seed=77, std=0.05)
    net = make_net(pa, dev, p)
    pe = engine.params_cached(*pa.odenet.params_of(net))
    r = np.random.RandomState(6)

    def problem(B):
        y0 = torch.from_numpy(np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)).to(dev)
        t = torch.from_numpy(np.stack([np.array([0.0, 0.2 + 0.01 * (b % 7)]) for b in range(B)]).astype(np.float32)).to(dev)
        G = torch.from_numpy(r.randn(2, B, N).astype(np.float32)).to(dev)
        return y0, t, G

    def run(prob, max_steps=0):
        y0, t, G = prob
        sol, st, _, ns = engine.solve_forward(pe, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 2, max_steps)
        adj, gr, st2, _, ns2 = engine.solve_adjoint(pe, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 2,
                                                    max_num_steps=max_steps)
        return sol.clone(), adj.clone(), gr.flat.clone(), int(st.max()), int(st2.max())

    assert _lib.load().phx_debug_adjoint_kernel_m(N, H, 40, 2, _lib.CTRL_PER_TRAJECTORY, _lib.METHODS["dopri5"]) == 3
    pa_, pb_ = problem(40), problem(70)
    engine.forget_workspaces()
    ref_a = run(pa_)
    assert ref_a[3] == 0 and ref_a[4] == 0
    for _ in range(3):                      # kept: sets 1, 0, 1
        got = run(pa_)
        assert all(torch.equal(x, y) for x, y in zip(got[:3], ref_a[:3]))
    engine.forget_workspaces()
    ref_b = run(pb_)
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(run(pb_)[:3], ref_b[:3]))
    for _ in range(2):                      # back to the first shape: filled once, then kept
        assert all(torch.equal(x, y) for x, y in zip(run(pa_)[:3], ref_a[:3]))
    failed = run(pa_, max_steps=1)          # every trajectory stops after one step: status max_num_steps exceeded
    assert failed[3] == 1
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(run(pa_)[:3], ref_a[:3]))


@pytest.mark.parametrize("B", [64, 200])
def test_step_counts_of_the_third_generation_kernels_match_the_first(pa, dev, oracle, monkeypatch, B):
    """Canary for the wrong-step-size signature of DESIGN.md section 2 ("one signature"): trajectories 12..15 of every
    tile of a third-generation kernel took thousands of dopri5 steps (a first-order error estimate from the second step
    on), results stayed plausible and no status was set.  Round 4 found the cause: a gfx950 store-data hazard the
    compiler does not cover (a VALU write to the data VGPRs of a 128-bit buffer store with an SGPR soffset within the next
    wait state corrupts lanes 12..15 of every 16 of the stored tuple -- here the FSAL slope tile k_7; reproducer
    tools/membench/store_war.hip), whether a build was hit depended on register allocation.  Every such store is guarded
    now (phx_mfma_v3common.inc: bstore_guard) and tests/test_abi_cpu.py checks the built ISA statically; this test is the
    dynamic side: multi-step per-trajectory dopri5 solves on shapes whose groups have MORE than 64 gene tiles (N =
    11 165: 175 tiles at B = 64, 88 at B = 200) must take the same number of steps per trajectory through k1_solve_fwd3 /
    k1_solve_adj3 as through the first-generation kernels (accept/reject noise aside), with no lane class standing
    out, and sampled rows must agree with the oracle."""
    from phoenix_amd import _lib, engine
    N, H = 11165, 40
    p = rand_params(N, H, seed=41, std=0.05)
    net = make_net(pa, dev, p)
    r = np.random.RandomState(3)
    y0 = np.clip(r.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)
    # interval lengths are spread over the lanes of a tile ((7 b) mod 16), not ordered by lane
    t = np.stack([np.array([0.0, 0.25 + 0.01 * ((7 * b) % 16)]) for b in range(B)]).astype(np.float32)
    # O(1) cotangents: with 1/(B N)-scaled ones atol = 1e-9 dominates the adjoint's error tolerance and two correct
    # implementations sit 1e-3 apart
    G = r.randn(B, 2, N).astype(np.float32)
    G[:, 0] = 0
    pe = engine.params_cached(*pa.odenet.params_of(net))
    y0d, td = torch.from_numpy(y0).to(dev), torch.from_numpy(t).to(dev).contiguous()
    Gd = torch.from_numpy(np.ascontiguousarray(G.transpose(1, 0, 2))).to(dev)
    assert _lib.load().phx_debug_adjoint_kernel_m(N, H, B, 2, _lib.CTRL_PER_TRAJECTORY, _lib.METHODS["dopri5"]) == 3

    def run():
        sol, st, nfe, ns = engine.solve_forward(pe, y0d, td, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 2)
        adj, gr, st2, nfe2, ns2 = engine.solve_adjoint(pe, td, sol, Gd, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9,
                                                       True, 2)
        assert int(st.max()) == 0 and int(st2.max()) == 0
        return sol.cpu().numpy(), ns.cpu().numpy(), adj.cpu().numpy(), ns2.cpu().numpy(), gr.flat.cpu().numpy()

    sol3, nsf3, adj3, nsb3, gr3 = run()
    monkeypatch.setenv("PHX_FWD", "v1")
    monkeypatch.setenv("PHX_ADJ", "v1")
    sol1, nsf1, adj1, nsb1, gr1 = run()
    monkeypatch.delenv("PHX_FWD")
    monkeypatch.delenv("PHX_ADJ")
    assert nsf1.min() >= 3 and nsb1.min() >= 3, "the canary needs multi-step solves"
    for name, a, b in (("forward", nsf3, nsf1), ("backward", nsb3, nsb1)):
        # the signature is gross (the maximal growth factor after every step: thousands of attempts, or a handful);
        # accept / reject noise at rtol = 1e-7 moves a 45-step solve by up to fifteen attempts between two kernels
        assert (np.abs(a.astype(np.int64) - b) <= np.maximum(5, 0.6 * b)).all(), (name, a.tolist(), b.tolist())
        lanes = np.arange(B) % 16
        bad, good = a[lanes >= 12], a[lanes < 12]
        bad1, good1 = b[lanes >= 12], b[lanes < 12]
        assert abs(bad.mean() / good.mean() - bad1.mean() / good1.mean()) <= 0.1, \
            (name, "trajectories 12-15 of the tiles step differently", a.tolist(), b.tolist())
    assert relerr(sol3, sol1) < TOL_DOPRI and relerr(adj3, adj1) < TOL_DOPRI_GRAD and relerr(gr3, gr1) < TOL_DOPRI_GRAD
    onet = onet_of(oracle, p)
    rows = [12, 13, 15, 28, 47, 63, 5][:7] if B >= 64 else list(range(B))
    ref = oracle.odeint_per_sample(onet, y0[rows], t[rows], method="dopri5")
    assert relerr(sol3.transpose(1, 0, 2)[rows], ref) < TOL_DOPRI


@pytest.mark.parametrize("N,H,B,adj", [(300, 10, 5, None), (300, 10, 5, "v3"), (1500, 12, 20, None), (1500, 12, 37, "v1")])
def test_shared_control_adjoint_multi_interval_vs_oracle(pa, dev, oracle, monkeypatch, N, H, B, adj):
    """odeint_adjoint on a batched y0 with ONE controller (reference semantics) over several output times:
    exercises the per-interval restarts, the jump a += grad_y[i-1] and the multi-step quadrature -- on the kernel the
    shape gets (300 genes: the wave-pair kernel; 1500 genes = 47 gene tiles: the third kernel) and on a forced one."""
    if adj:
        monkeypatch.setenv("PHX_ADJ", adj)
    p = rand_params(N, H, seed=17, std=0.15 if N == 300 else 0.05)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(5)
    y0 = (r.rand(B, 1, N) * 1.4 - 0.2).astype(np.float32)
    t = np.array([0.0, 0.6, 1.1, 2.5], np.float32)
    G = r.randn(4, B, 1, N).astype(np.float32)
    ref = oracle.odeint(onet, y0, t, method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    y0t = torch.from_numpy(y0).to(dev).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
    assert relerr(sol.detach().cpu().numpy(), ref) < TOL_DOPRI
    (sol * torch.from_numpy(G).to(dev)).sum().backward()
    assert grad_err(y0t.grad.cpu().numpy(), adj_ref) < TOL_DOPRI_GRAD
    gg = grads_of(net)
    for k in KEYS:
        assert grad_err(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD, k


@pytest.mark.parametrize("N,H,B,method", [(700, 250, 5, "dopri5"), (300, 131, 20, "rk4"), (300, 131, 4, "dopri5")])
def test_chunked_hidden_layer_shared_control_vs_oracle(pa, dev, oracle, N, H, B, method):
    """H > 128: the hidden layer is processed in chunks whose weights take turns in LDS (H = 250 -> 125 + 125,
    H = 131 -> 66 + 65; the library-wide limit is H <= 256).  Batched y0 with one controller over several output times, forward + adjoint, against the
    oracle; and the VALU engine (PHX_ENGINE=v0) must agree with the chunked MFMA path."""
    import os
    p = rand_params(N, H, seed=H + B, std=0.04)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(5)
    y0 = (r.rand(B, 1, N) * 1.2 - 0.1).astype(np.float32)
    t = np.array([0.0, 0.4, 0.9, 1.3], np.float32)
    G = r.randn(4, B, 1, N).astype(np.float32)
    ref = oracle.odeint(onet, y0, t, method=method)
    adj_ref, gr_ref = oracle.adjoint_backward(onet, t, ref, G, method=method, theta_in_norm=True)
    tol, gtol = (TOL_DOPRI, TOL_DOPRI_GRAD) if method == "dopri5" else (TOL_FIXED, TOL_FIXED)
    results = {}
    for engine in ("v1", "v0"):
        if engine == "v0":
            os.environ["PHX_ENGINE"] = "v0"
        try:
            for rep in range(2):     # the second pass runs on workspaces that exist and were filled with junk
                from phoenix_amd import engine as _eng
                _eng.forget_workspaces()               # this test writes into the workspaces behind the engine's back
                for buf in _eng._ws_cache.values():
                    buf.view(torch.float32)[: buf.numel() // 4].fill_(777.0)
                zero_grads(net)
                y0t = torch.from_numpy(y0).to(dev).requires_grad_(True)
                sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method=method)
                (sol * torch.from_numpy(G).to(dev)).sum().backward()
        finally:
            os.environ.pop("PHX_ENGINE", None)
        assert relerr(sol.detach().cpu().numpy(), ref) < tol, engine
        assert relerr(y0t.grad.cpu().numpy(), adj_ref) < gtol, engine
        gg = grads_of(net)
        for k in KEYS:
            assert relerr(gg[k], gr_ref[k]) < gtol, (engine, k)
        results[engine] = gg
    for k in KEYS:
        assert relerr(results["v1"][k], results["v0"][k]) < gtol, k


def test_wide_hidden_layer_paths(pa, dev, oracle):
    """H = 120 (yeast config: 8 hidden tiles, weights LDS resident) and H = 200 (B-cell config: two chunks of 100
    hidden rows, re-staged per evaluation), per-sample control."""
    for N, H, B in ((600, 120, 3), (400, 200, 2)):
        p = rand_params(N, H, seed=H, std=0.03)
        net, onet = make_net(pa, dev, p), onet_of(oracle, p)
        r = np.random.RandomState(6)
        y0 = r.rand(B, N).astype(np.float32)
        t = np.tile(np.array([[0.0, 0.8]], np.float32), (B, 1))
        G = r.randn(B, 2, N).astype(np.float32)
        ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
        adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
        y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
        sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
        got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
        assert relerr(got, ref) < TOL_DOPRI, (N, H)
        (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
        assert grad_err(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < TOL_DOPRI_GRAD, (N, H)
        gg = grads_of(net)
        for k in KEYS:
            assert grad_err(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD, (k, N, H)


@pytest.mark.parametrize("N,H,B", [(322, 120, 37), (500, 200, 20), (250, 120, 70), (777, 120, 12), (400, 64, 10)])
def test_half_block_gene_tiles_and_helper_waves_vs_oracle(pa, dev, oracle, monkeypatch, N, H, B):
    """Chunked third-generation kernels, HALF-BLOCK gene tiles (two workgroups per 32-gene block):
    three tiles and a last block whose second half is padding only (322 genes), two tiles with one helper wave each
    (B = 20), two batch groups with at most 16 members (250 genes), one tile with three helper waves (B = 12; with two
    chunks, B = 10: more helpers than chunks) -- each
    against the oracle, and against whole-block tiles (PHX_V3C_HB=0: same steps, other summation order)."""
    import ctypes as C
    from phoenix_amd import _lib
    p = rand_params(N, H, seed=3 * N + H, std=0.03)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(9)
    y0 = r.rand(B, N).astype(np.float32)
    t = np.stack([np.array([0.0, 0.5 + 0.01 * (b % 7)]) for b in range(B)]).astype(np.float32)
    G = r.randn(B, 2, N).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    out = {}
    # "1": the default plan (where the chip has room: batch groups of one or two tiles, their helper waves);
    # "4": half-block tiles in the four-tile groups of the plan before that rule; "0": whole-block tiles
    for hb in ("1", "4", "0"):
        monkeypatch.setenv("PHX_V3C_HB", "0" if hb == "0" else "1")
        if hb == "1":
            monkeypatch.delenv("PHX_V3C_NTG", raising=False)
        else:
            monkeypatch.setenv("PHX_V3C_NTG", "4")
        lib = _lib.load()
        plan = (C.c_int * 6)()
        off, nwg = C.c_size_t(0), C.c_int(0)
        for op in (_lib.OP_ODEINT, _lib.OP_ADJOINT):
            assert lib.phx_debug_profile_region(op, N, H, B, 2, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan) == 0
            assert (plan[5] % 10) // 2 == (0 if hb == "0" else 1), (op, list(plan))   # the half-block form runs where the test says it does
        for q in net.parameters():
            q.grad = None
        y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
        sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
        got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
        assert relerr(got, ref) < TOL_DOPRI, hb
        (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
        assert grad_err(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < TOL_DOPRI_GRAD, hb
        gg = grads_of(net)
        for k in KEYS:
            assert grad_err(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD, (k, hb)
        out[hb] = (got, gg)
    for v in ("1", "4"):
        assert relerr(out[v][0], out["0"][0]) < TOL_DOPRI
        for k in KEYS:
            assert relerr(out[v][1][k], out["0"][1][k]) < TOL_DOPRI_GRAD, (k, v)


@pytest.mark.parametrize("N,H,K", [(350, 40, 1500), (1537, 24, 1100), (600, 120, 1100), (11165, 40, 2100), (500, 200, 1100)])
def test_prior_backward_from_saved_hidden_rows_equals_recomputation(pa, dev, N, H, K):
    """phx_prior_mse_save + phx_prior_vjp_saved (the forward chain's hidden rows kept for the backward) give the bits of
    phx_prior_mse + phx_rhs_vjp (which recomputes them); for H > 128 there is no saved path and the mirror falls back."""
    from phoenix_amd import engine
    p = rand_params(N, H, seed=5 * N + H, std=0.08)
    net = make_net(pa, dev, p)
    P = engine.params_cached(*pa.odenet.params_of(net))
    r = np.random.RandomState(3)
    X = torch.from_numpy((r.rand(K, N) - 0.5).astype(np.float32)).to(dev)
    tgt = torch.from_numpy((r.randn(K, N) * 0.1).astype(np.float32)).to(dev)
    loss0, cot0 = engine.prior_mse(P, X, tgt)
    loss1, cot1, z = engine.prior_mse(P, X, tgt, keep_hidden=True)
    assert torch.equal(loss0, loss1) and torch.equal(cot0, cot1)
    assert (z is None) == (H > 128)
    _, g0 = engine.rhs_vjp(P, X, cot0, True, want_grads=True, want_vjp_y=False)
    if z is not None:
        g1 = engine.prior_vjp_saved(P, X, cot1, z)
        for k in ("Ws", "bs", "Wp", "bp", "Wa", "g"):
            assert torch.equal(getattr(g0, k), getattr(g1, k)), k


@pytest.mark.parametrize("N,H,K", [(350, 40, 1500), (1537, 24, 333), (600, 120, 70), (600, 120, 1100), (11165, 40, 2100), (500, 200, 1100), (300, 131, 90),
                                   (11165, 40, 10000)])   # the last one: the prior batch of config C4 at its full size
def test_prior_branch_vs_oracle(pa, dev, oracle, N, H, K):
    """prior_only_forward on a large batch and its parameter gradients (train_insilico.py:134-138): the exchange-free
    MFMA chains of phx_mfma_batch.inc (forward and parameter gradients for any batch when the input needs no gradient;
    the pass-structured k1_eval_fwd: test_small_batch_evaluation_both_routes); H <= 48 and H = 120 tilings, a ragged
    last gene block, several partial-buffer chunks at breast scale."""
    p = rand_params(N, H, seed=3 * N + H, std=0.08)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(12)
    X = (r.rand(K, 1, N) - 0.5).astype(np.float32)
    tgt = (r.randn(K, 1, N) * 0.1).astype(np.float32)
    Xt, tt = torch.from_numpy(X).to(dev), torch.from_numpy(tgt).to(dev)
    pred = net.prior_only_forward(torch.tensor(0.0), Xt)
    ref = oracle.rhs(onet, X, prior_only=True)
    assert relerr(pred.detach().cpu().numpy(), ref) < TOL_RHS
    loss = torch.mean((pred - tt) ** 2)
    loss.backward()
    cot = 2.0 * (ref - tgt) / ref.size
    _, gr_ref, _ = oracle.rhs_vjp(onet, X, cot.astype(np.float32), prior_only=True)
    got = grads_of(net)
    for k in KEYS:
        if np.max(np.abs(gr_ref[k])) == 0:
            assert np.max(np.abs(got[k])) == 0, k
        else:
            assert relerr(got[k], gr_ref[k]) < 4 * TOL_RHS, k
    # full RHS on the same batch (forward kernel, non-prior mode)
    f = net(torch.tensor(0.0), Xt)
    assert relerr(f.detach().cpu().numpy(), oracle.rhs(onet, X)) < TOL_RHS
    # fused loss head (ODENet.prior_mse, used by training_step): same loss, same gradients
    zero_grads(net)
    loss2 = net.prior_mse(torch.tensor(0.0), Xt, tt)
    assert abs(float(loss2.detach()) - float(loss.detach())) <= 2e-6 * abs(float(loss.detach()))
    (0.01 * loss2).backward()            # a loss weight like (1 - lambda) must scale the gradients
    got2 = grads_of(net)
    for k in KEYS:
        if np.max(np.abs(gr_ref[k])) == 0:
            assert np.max(np.abs(got2[k])) == 0, k
        else:
            assert relerr(got2[k], 0.01 * gr_ref[k]) < 4 * TOL_RHS, k


@pytest.mark.parametrize("N,H,K", [(350, 40, 16), (1537, 24, 333), (600, 120, 70), (11165, 40, 200), (300, 131, 90)])
def test_small_batch_evaluation_both_routes(pa, dev, oracle, monkeypatch, N, H, K):
    """Standalone RHS evaluation of a small batch: with the packed weight images the library takes the kernel chain at
    every batch size (round 4), without them -- or with PHX_BATCH_MIN_ROWS=1024 -- the pass-structured k1_eval_fwd.
    Both routes against the oracle, full RHS and prior_only."""
    p = rand_params(N, H, seed=7 * N + H, std=0.08)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    X = (np.random.RandomState(4).rand(K, 1, N) - 0.3).astype(np.float32)
    Xt = torch.from_numpy(X).to(dev)
    refs = {True: oracle.rhs(onet, X, prior_only=True), False: oracle.rhs(onet, X)}
    outs = {}
    for route in ("default", "1024"):
        if route == "1024":
            monkeypatch.setenv("PHX_BATCH_MIN_ROWS", "1024")
        with torch.no_grad():
            outs[route] = (net.prior_only_forward(torch.tensor(0.0), Xt).cpu().numpy(), net(torch.tensor(0.0), Xt).cpu().numpy())
        assert relerr(outs[route][0], refs[True]) < TOL_RHS, route
        assert relerr(outs[route][1], refs[False]) < TOL_RHS, route
    assert relerr(outs["default"][0], outs["1024"][0]) < TOL_RHS


def test_prior_branch_pass_structured_fallback(pa, dev, oracle, monkeypatch):
    """PHX_PGRAD=v1: the pass-structured k1_eval_pgrad (kept as the fallback of the kernel chain) still agrees."""
    monkeypatch.setenv("PHX_PGRAD", "v1")
    N, H, K = 1537, 24, 333
    p = rand_params(N, H, seed=5, std=0.08)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(2)
    X = (r.rand(K, 1, N) - 0.5).astype(np.float32)
    tgt = (r.randn(K, 1, N) * 0.1).astype(np.float32)
    pred = net.prior_only_forward(torch.tensor(0.0), torch.from_numpy(X).to(dev))
    ref = oracle.rhs(onet, X, prior_only=True)
    assert relerr(pred.detach().cpu().numpy(), ref) < TOL_RHS
    torch.mean((pred - torch.from_numpy(tgt).to(dev)) ** 2).backward()
    _, gr_ref, _ = oracle.rhs_vjp(onet, X, (2.0 * (ref - tgt) / ref.size).astype(np.float32), prior_only=True)
    got = grads_of(net)
    for k in KEYS:
        if np.max(np.abs(gr_ref[k])) == 0:
            assert np.max(np.abs(got[k])) == 0, k
        else:
            assert relerr(got[k], gr_ref[k]) < 4 * TOL_RHS, k


def test_batches_larger_than_one_residency_are_chunked(pa, dev):
    """B = 2000 trajectories at N = 11165 exceed what one persistent launch can hold co-resident; the host then
    walks the batch in chunks (per-trajectory control => trajectories are independent).  Property: every row
    equals the same row integrated in a small batch, forward and backward, and parameter gradients add up."""
    N, H, B = 11165, 40, 2000
    p = rand_params(N, H, seed=21, std=0.02)
    net = make_net(pa, dev, p)
    g = torch.Generator(device="cpu").manual_seed(3)
    y0 = (torch.randn(B, 1, N, generator=g) * 0.15 + 0.5).clamp_(0.03, 1.07).to(dev)
    t = torch.tensor([[0.0, 0.0051]], dtype=torch.float32).repeat(B, 1).to(dev)
    G = (torch.randn(2, B, 1, N, generator=g) / N).to(dev)
    yb = y0.clone().requires_grad_(True)
    sol = pa.odeint_adjoint(net, yb, t)
    (sol * G).sum().backward()
    full = grads_of(net)
    rows = [0, 1023, 1024, 1999]            # both sides of the chunk boundary
    zero_grads(net)
    ys = y0[rows].clone().requires_grad_(True)
    s2 = pa.odeint_adjoint(net, ys, t[rows])
    (s2 * G[:, rows]).sum().backward()
    assert relerr(s2.detach().cpu().numpy(), sol.detach()[:, rows].cpu().numpy()) < 2e-6
    assert relerr(ys.grad.cpu().numpy(), yb.grad[rows].cpu().numpy()) < 2e-5
    # gradient additivity over two halves
    acc = {k: np.zeros_like(v) for k, v in full.items()}
    for lo, hi in ((0, 1000), (1000, 2000)):
        zero_grads(net)
        yy = y0[lo:hi].clone().requires_grad_(True)
        ss = pa.odeint_adjoint(net, yy, t[lo:hi])
        (ss * G[:, lo:hi]).sum().backward()
        for k, v in grads_of(net).items():
            acc[k] += v
    for k in KEYS:
        assert relerr(acc[k], full[k]) < 3e-5, k


def test_f1_prior_targets_spmm(pa, dev):
    """prior_grad = X @ P on the GPU (CSC SpMM) against the reference's dense matmul (golden G8) and, at
    breast scale, against torch's dense product of the same sparse matrix (fp32 reference of the same op)."""
    from phoenix_amd.prior import PriorMatrix, prior_targets
    g = load_golden("g8_prior")
    a = sub(g, "g350/")
    P = PriorMatrix(a["rows"], a["cols"], a["vals"], 350, dev)
    got = prior_targets(torch.from_numpy(a["X"]).to(dev), P)
    assert got.shape == a["prior_grad"].shape
    assert relerr(got.cpu().numpy(), a["prior_grad"]) < 2e-6
    b = sub(g, "trip/")
    r, c = np.nonzero(b["dense"])
    Pt = PriorMatrix(r, c, b["dense"][r, c], 40, dev)
    assert relerr(prior_targets(torch.from_numpy(b["X"]).to(dev), Pt).cpu().numpy(), b["prior_grad"]) < 2e-6
    assert relerr(prior_targets(torch.from_numpy(b["X"]).to(dev), Pt.abs()).cpu().numpy(), b["prior_grad_abs"]) < 2e-6
    # breast scale: N = 11165, 0.3 % dense {0, 0.5} prior, K = 10000 rows
    N, K = 11165, 10000
    rs = np.random.RandomState(1)
    nnz = int(0.003 * N * N)
    rr, cc = rs.randint(0, N, nnz), rs.randint(0, N, nnz)
    Pb = PriorMatrix(rr, cc, np.full(nnz, 0.5, np.float32), N, dev)
    X = torch.rand(K, 1, N, device=dev) - 0.5
    out = prior_targets(X, Pb)
    dense = Pb.to_dense().to(dev)
    ref = torch.matmul(X[:64], dense)
    assert relerr(out[:64].cpu().numpy(), ref.cpu().numpy()) < 5e-6
    # linearity property at full size: (X1 + X2) P = X1 P + X2 P
    X2 = torch.rand(K, 1, N, device=dev) - 0.5
    lhs = prior_targets(X + X2, Pb)
    assert relerr((out + prior_targets(X2, Pb)).cpu().numpy()[:256], lhs.cpu().numpy()[:256]) < 5e-6


# --------------------------------------------------------------------------- row f3: batched analysis callers
def test_f3_gene_influence_scores(pa, dev):
    """find_gene_influences.py:64-77 on the engine against the scores the reference produced (golden G10) from
    the same random draws: 2N shared-control dopri5 solves of [8,1,24] with 10 outputs on a float64 grid."""
    from phoenix_amd.analysis import gene_influence_scores
    g = sub(load_golden("g10_analysis"), "infl/")
    net = make_net(pa, dev, sub(g, "p_"))
    inits, perts = torch.from_numpy(g["inits"]), torch.from_numpy(g["perts"])
    got = gene_influence_scores(net, 24, "dopri5", n_random_inputs_per_gene=8, device=dev,
                                draws=lambda k: (inits[k], perts[k]))
    assert got.shape == g["scores"].shape
    assert np.max(np.abs(got - g["scores"]) / np.abs(g["scores"])) < 2e-5
    # default draws: runs end to end, positive finite scores, one per requested gene
    some = gene_influence_scores(net, 24, "dopri5", n_random_inputs_per_gene=8, device=dev, genes=[3, 7])
    assert some.shape == (2,) and np.all(np.isfinite(some)) and np.all(some > 0)


@pytest.mark.parametrize("N,H,B,K,method", [(350, 30, 7, 6, "dopri5"), (350, 30, 60, 4, "dopri5"), (700, 20, 20, 3, "rk4"),
                                              (11165, 40, 60, 3, "dopri5"), (96, 8, 5, 40, "dopri5"),
                                              (14691, 200, 24, 2, "dopri5")])
def test_odeint_calls_equals_separate_calls(pa, dev, oracle, N, H, B, K, method):
    """K odeint calls in one batch (one step controller per call) against the K separate calls and, for one call, the
    oracle: every call must see ITS OWN shared step size -- the calls start from differently scaled states so their
    step sequences differ."""
    p = rand_params(N, H, seed=N + K, std=0.6 / np.sqrt(N))
    net = make_net(pa, dev, p)
    rs = np.random.RandomState(K)
    y0s = np.stack([(rs.rand(B, 1, N).astype(np.float32) - 0.5) * (0.2 + 0.9 * k) for k in range(K)])
    t = torch.from_numpy(np.arange(0, 1, 0.1)).to(dev)
    y0d = torch.from_numpy(y0s).to(dev)
    out = pa.odeint_calls(net, y0d, t, method=method)
    assert out.shape == (K, 10, B, 1, N)
    steps = []
    for k in range(K):
        one, nfe, nsteps = pa.odeint(net, y0d[k], t, method=method, return_stats=True)
        steps.append(int(nsteps[0]))
        # the batched calls run on k1_solve_fwd, a separate dopri5 call with H <= 48 on k1_solve_fwd3: two adaptive solves
        # agree to the trajectory tolerance, not bit for bit (fixed grids: the same kernel, summation order only)
        assert relerr(out[k].cpu().numpy(), one.cpu().numpy()) < (TOL_DOPRI if method == "dopri5" else 5e-6), k
    if method == "dopri5" and N <= 700:
        assert len(set(steps)) > 1       # the controllers really were independent
    k = K - 1
    if N <= 2000:
        ref = oracle.odeint(onet_of(oracle, p), y0s[k], t.cpu().numpy(), method=method)
        assert relerr(out[k].cpu().numpy(), ref) < TOL_DOPRI
    # a failing call fails the batch with the reference's message
    if method == "dopri5" and N <= 700:
        with pytest.raises(AssertionError, match="max_num_steps"):
            pa.odeint_calls(net, y0d, torch.tensor([0.0, 40.0], device=dev), options={"max_num_steps": 1})


def test_odeint_calls_without_a_batched_plan_runs_the_calls_one_by_one(pa, dev, monkeypatch):
    """shapes the MFMA planner does not take (here: the VALU engine forced) have no calls-per-launch plan: the same
    answer, one launch per call"""
    p = rand_params(64, 8, seed=2)
    net = make_net(pa, dev, p)
    y0s = torch.rand(3, 5, 1, 64, device=dev) - 0.25
    t = torch.from_numpy(np.arange(0, 1, 0.25)).to(dev)
    ref = pa.odeint_calls(net, y0s, t)
    monkeypatch.setenv("PHX_ENGINE", "v0")
    got = pa.odeint_calls(net, y0s, t)
    assert got.shape == ref.shape == (3, 4, 5, 1, 64)
    assert relerr(got.cpu().numpy(), ref.cpu().numpy()) < 5e-6


def test_f3_calculate_trajectory(pa, dev):
    """DataHandler.calculate_trajectory (datahandler.py:310-340): same samples (numpy stream) and the same 300-point
    dense trajectories as the reference, all samples integrated in one launch."""
    import os
    from phoenix_amd.data import DataHandler
    g = sub(load_golden("g10_analysis"), "traj/")
    net = make_net(pa, dev, sub(g, "p_"))
    csv = os.path.join(os.path.dirname(__file__), "golden", "g9_data.csv")
    np.random.seed(404)
    h = DataHandler.fromcsv(csv, dev, 0.3, batch_type="trajectory", noise=0.0)
    trajs, samples, grid = h.calculate_trajectory(net, "dopri5", num_val_trajs=2)
    assert [int(s) for s in samples] == [int(s) for s in g["samples"]]
    assert np.array_equal(grid, g["grid"])
    got = np.stack([x.numpy() for x in trajs])
    assert got.shape == g["trajectories"].shape
    assert relerr(got, g["trajectories"]) < TOL_DOPRI


@pytest.mark.parametrize("seed", list(range(32)))
def test_random_shapes_mfma_engine_agrees_with_valu_engine(pa, dev, seed):
    """Seeded fuzz over shapes the planner treats differently (ragged gene blocks, 1..5 trajectory tiles, helper waves,
    small groups, H <= 48 / <= 128 / chunked, shared and per-trajectory control, every method, several output times):
    the MFMA engine against the independent VALU engine (PHX_ENGINE=v0), forward solution and all gradients."""
    import os
    r = np.random.RandomState(1000 + seed)
    N = int(r.choice([33, 96, 350, 777, 1500]))
    H = int(r.choice([5, 24, 40, 64, 120, 150]))
    B = int(r.choice([1, 3, 17, 40, 70, 130, 200, 330]))   # the larger ones: several batch groups, padding-only waves
    method = str(r.choice(["dopri5", "rk4", "midpoint", "euler"]))
    T = int(r.choice([2, 3, 5]))
    per_sample = bool(r.rand() < 0.5)
    p = rand_params(N, H, seed=seed, std=0.6 / np.sqrt(N))
    net = make_net(pa, dev, p)
    y0 = (r.rand(B, 1, N) * 1.2 - 0.1).astype(np.float32)
    tgrid = np.cumsum(np.concatenate([[0.0], r.uniform(0.1, 0.5, T - 1)])).astype(np.float32)
    if r.rand() < 0.25:
        tgrid = tgrid[::-1].copy()                       # decreasing time
    t = np.tile(tgrid, (B, 1)) if per_sample else tgrid
    G = r.randn(T, B, 1, N).astype(np.float32)
    res = {}
    # "alt": the backward-kernel generation / geometry the planner would NOT pick for this shape
    alt = [{"PHX_ADJ": "v2", "PHX_ADJ2_NP": "2"}, {"PHX_ADJ": "v2", "PHX_ADJ2_NP": "4"}, {"PHX_ADJ": "v1", "PHX_FWD": "v1"},
           {"PHX_ADJ": "v3"}][seed % 4]
    for eng in ("v1", "v0", "alt"):
        if eng == "v0":
            os.environ["PHX_ENGINE"] = "v0"
        if eng == "alt":
            os.environ.update(alt)
        try:
            for rep in range(2):     # the second pass runs on workspaces that exist and were filled with junk
                from phoenix_amd import engine as _eng
                _eng.forget_workspaces()               # this test writes into the workspaces behind the engine's back
                for buf in _eng._ws_cache.values():
                    buf.view(torch.float32)[: buf.numel() // 4].fill_(777.0)
                zero_grads(net)
                y0t = torch.from_numpy(y0).to(dev).requires_grad_(True)
                sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method=method)
                (sol * torch.from_numpy(G).to(dev)).sum().backward()
        finally:
            for k in ("PHX_ENGINE", "PHX_ADJ", "PHX_ADJ2_NP", "PHX_FWD"):
                os.environ.pop(k, None)
        res[eng] = (sol.detach().cpu().numpy(), y0t.grad.cpu().numpy(), grads_of(net))
    what = (N, H, B, T, method, per_sample)
    tol, gtol = (TOL_DOPRI, TOL_DOPRI_GRAD) if method == "dopri5" else (TOL_FIXED, TOL_FIXED)
    for eng in ("v1", "alt"):
        assert relerr(res[eng][0], res["v0"][0]) < tol, (eng,) + what
        assert relerr(res[eng][1], res["v0"][1]) < gtol, (eng,) + what
        for k in KEYS:
            assert relerr(res[eng][2][k], res["v0"][2][k]) < gtol, (eng, k) + what


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_new_paths_agree_with_their_references(pa, dev, monkeypatch, seed):
    """Fuzz of the round-2 paths on random shapes: (1) batched odeint calls against the separate calls, (2) the full-VJP
    kernel chain against the VALU engine.  Engine vs engine -- an extra on top of the oracle-checked cases above."""
    from phoenix_amd import engine
    rs = np.random.RandomState(1000 + seed)
    N = int(rs.choice([33, 64, 97, 350, 513, 1200, 2111]))
    H = int(rs.choice([3, 8, 17, 40, 48, 49, 100, 128]))
    p = rand_params(N, H, seed=seed, std=0.5 / np.sqrt(N))
    net = make_net(pa, dev, p)
    # (1)
    K, B = int(rs.randint(2, 7)), int(rs.randint(1, 40))
    method = str(rs.choice(["dopri5", "rk4", "euler", "midpoint"]))
    T = int(rs.randint(2, 6))
    t = torch.from_numpy(np.sort(rs.rand(T)) * (1 if rs.rand() < 0.7 else -1)).to(dev)
    y0s = torch.from_numpy((rs.rand(K, B, 1, N).astype(np.float32) - 0.5) * rs.uniform(0.2, 1.5, size=(K, 1, 1, 1)).astype(np.float32)).to(dev)
    out = pa.odeint_calls(net, y0s, t, method=method)
    for k in range(K):
        one = pa.odeint(net, y0s[k], t, method=method).detach()
        assert relerr(out[k].cpu().numpy(), one.cpu().numpy()) < 1e-5, (N, H, K, B, method, k)
    # (2)
    if H <= 128:
        P = engine.params_cached(*pa.odenet.params_of(net))
        Bv = int(rs.randint(1, 200))
        y = torch.from_numpy((rs.rand(Bv, N) * 3 - 1).astype(np.float32)).to(dev)
        cot = torch.from_numpy(rs.randn(Bv, N).astype(np.float32)).to(dev)
        po = bool(rs.rand() < 0.3)
        vjp, grads = engine.rhs_vjp(P, y, cot, prior_only=po)
        monkeypatch.setenv("PHX_ENGINE", "v0")
        vjp0, grads0 = engine.rhs_vjp(P, y, cot, prior_only=po)
        monkeypatch.delenv("PHX_ENGINE")
        assert relerr(vjp.cpu().numpy(), vjp0.cpu().numpy()) < TOL_RHS, (N, H, Bv, po)
        for k in ("Ws", "bs", "Wp", "bp", "Wa", "g"):
            a, b = getattr(grads, k).cpu().numpy(), getattr(grads0, k).cpu().numpy()
            assert (np.abs(b).max() == 0 and np.abs(a).max() == 0) or relerr(a, b) < 2 * TOL_RHS, (k, N, H, Bv, po)


@pytest.mark.parametrize("N,H,B,method", [(350, 40, 1500, "rk4"), (350, 40, 1000, "dopri5"), (2000, 120, 200, "dopri5"),
                                          (700, 20, 40, "dopri5"), (1200, 100, 300, "rk4")])
def test_results_do_not_depend_on_what_the_workspace_held_before(pa, dev, monkeypatch, N, H, B, method):
    """Every call must initialise whatever it reads from the caller's workspace.  The batch sizes leave padding-only
    wave pairs in the last batch group of the second backward kernel (their gradient partials are never written by the
    kernel: a stale partial of an earlier launch used to be summed into the gradients).  Each operation runs on a
    workspace filled with zeros and on one filled with 1000.0 -- bitwise equal outputs, for every backward kernel --
    and the default kernel's gradients agree with the first-generation kernel's."""
    from phoenix_amd import engine, _lib
    p = rand_params(N, H, seed=N + B, std=0.5 / np.sqrt(N))
    net = make_net(pa, dev, p)
    P = engine.params_cached(*pa.odenet.params_of(net))
    rs = np.random.RandomState(B)
    y0 = torch.from_numpy((rs.rand(B, N) * 0.8 + 0.1).astype(np.float32)).to(dev)
    t = torch.stack([torch.zeros(B), torch.full((B,), 0.05)], 1).double().to(dev)
    G = torch.from_numpy((rs.randn(2, B, N) / (B * N)).astype(np.float32)).to(dev)
    cot = torch.from_numpy(rs.randn(B, N).astype(np.float32)).to(dev)

    def poison(value):
        engine.forget_workspaces()                     # the engine would otherwise vouch for them (ws_keep)
        for buf in engine._ws_cache.values():
            buf.view(torch.float32)[: buf.numel() // 4].fill_(value)

    def everything():
        out = []
        sol, st, _, _ = engine.solve_forward(P, y0, t, method, _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
        adj, grads, st2, _, _ = engine.solve_adjoint(P, t, sol, G, method, _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
        assert int(st.max()) == 0 and int(st2.max()) == 0
        out += [sol, adj, grads.flat]
        vjp, g2 = engine.rhs_vjp(P, y0, cot)
        out += [vjp, g2.flat, engine.rhs_forward(P, y0)]
        _, g3 = engine.rhs_vjp(P, y0, cot, prior_only=True, want_vjp_y=False)      # the prior branch's backward chain
        out += [g3.flat, engine.rhs_forward(P, y0, prior_only=True)]
        pm = engine.prior_mse(P, y0, cot)                                          # fused loss head (large batches only)
        if pm is not None:
            out += [pm[0], pm[1]]
        with torch.no_grad():                                                      # shared step control, batched calls
            ts = torch.from_numpy(np.arange(0, 0.5, 0.1)).to(dev)
            nb = min(B, 90) // 3
            out += [pa.odeint(net, y0[:37].unsqueeze(1), ts, method=method).detach(),
                    pa.odeint_calls(net, y0[:3 * nb].reshape(3, nb, 1, N), ts, method=method)]
        return [x.clone() for x in out]

    results = {}
    for variant, env in (("default", {}), ("first kernel", {"PHX_ADJ": "v1", "PHX_FWD": "v1"}), ("third kernel", {"PHX_ADJ": "v3"})):
        for k in ("PHX_ADJ", "PHX_ADJ2_NP", "PHX_FWD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        everything()                       # allocates the workspaces of this variant
        poison(0.0)
        clean = everything()
        poison(1000.0)
        dirty = everything()
        for a, b in zip(clean, dirty):
            assert torch.equal(a, b), variant
        results[variant] = clean
    for other in ("default", "third kernel"):
        for a, b in zip(results[other][1:3], results["first kernel"][1:3]):      # adj_y0 and the six gradients
            assert relerr(a.cpu().numpy(), b.cpu().numpy()) < 2e-5, other


def test_other_stream_and_mixed_time_directions(pa, dev, oracle):
    """(1) A solve issued on a side stream right after the parameters were laid out on the default stream (the weight
    images are packed there) waits for that packing.  (2) In the per-sample loop every sample has its own time grid --
    also its own DIRECTION: half of the batch integrates backwards in time here.  Both against the oracle."""
    N, H, B = 350, 30, 6
    p = rand_params(N, H, seed=12, std=0.6 / np.sqrt(N))
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    rs = np.random.RandomState(5)
    y0 = (rs.rand(B, 1, N).astype(np.float32) - 0.25)
    t = np.stack([np.array([0.0, 0.3, 0.7]) if b % 2 == 0 else np.array([1.0, 0.6, 0.1]) for b in range(B)])
    G = rs.randn(3, B, 1, N).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    side = torch.cuda.Stream(device=dev)
    y0t = torch.from_numpy(y0).to(dev).requires_grad_(True)
    tt, Gt = torch.from_numpy(t).to(dev), torch.from_numpy(G).to(dev)
    torch.cuda.synchronize()
    with torch.no_grad():
        net.gene_multipliers.mul_(1.0)            # new parameter version: the next call lays the parameters out again
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sol = pa.odeint_adjoint(net, y0t, tt)
        (sol * Gt).sum().backward()
    side.synchronize()
    got_sol = sol.detach().cpu().numpy().reshape(3, B, N).transpose(1, 0, 2)          # oracle layout [B,T,N]
    assert relerr(got_sol, ref) < TOL_DOPRI
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G.reshape(3, B, N).transpose(1, 0, 2).copy(),
                                                         method="dopri5")[:2]
    assert grad_err(y0t.grad.cpu().numpy().reshape(B, N), np.asarray(adj_ref).reshape(B, N)) < TOL_DOPRI_GRAD
    got = grads_of(net)
    for k in KEYS:
        assert grad_err(got[k], gr_ref[k]) < TOL_DOPRI_GRAD, k



@pytest.mark.parametrize("N,H,B,T,env", [(1500, 200, 10, 2, {"PHX_V3C_NB": "4", "PHX_V3C_RES": "0"}),
                                         (777, 120, 20, 3, {"PHX_V3C_NB": "2"}),
                                         (1511, 200, 7, 2, {"PHX_V3C_NB": "2", "PHX_V3C_RES": "0"})])
def test_block_split_of_the_chunked_kernels_vs_oracle(pa, dev, oracle, monkeypatch, N, H, B, T, env):
    """Chunked third-generation kernels on multi-block gene tiles with few trajectory tiles (the B-cell shape at the reference's
    batch of 2 is the natural case; smaller problems get there with a forced tile size): the waves of a tile take its gene
    blocks, partial rows and norm partials are added in LDS (four parts on re-staged chunks, two parts x two tiles on resident
    ones, a last gene tile with fewer blocks than parts) -- against the oracle, and against the unsplit form."""
    import ctypes as C
    from phoenix_amd import _lib
    p = rand_params(N, H, seed=5 * N + H, std=0.03)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(4)
    y0 = r.rand(B, N).astype(np.float32)
    t = np.stack([np.linspace(0.0, 0.4 + 0.01 * (b % 5), T) for b in range(B)]).astype(np.float32)
    G = r.randn(B, T, N).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    adj_ref, gr_ref = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = {}
    for split in ("1", "0"):
        monkeypatch.setenv("PHX_V3C_SPLIT", split)
        plan = (C.c_int * 6)()
        off, nwg = C.c_size_t(0), C.c_int(0)
        for op in (_lib.OP_ODEINT, _lib.OP_ADJOINT):
            assert _lib.load().phx_debug_profile_region(op, N, H, B, T, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan) == 0
            assert (plan[5] % 10) // 4 == int(split), (op, list(plan))
        for q in net.parameters():
            q.grad = None
        y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
        sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
        got = sol.detach().cpu().numpy().reshape(T, B, N).transpose(1, 0, 2)
        assert relerr(got, ref) < TOL_DOPRI, split
        (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(T, B, 1, N).copy()).to(dev)).sum().backward()
        assert grad_err(y0t.grad.cpu().numpy().reshape(B, N), adj_ref) < TOL_DOPRI_GRAD, split
        gg = grads_of(net)
        for k in KEYS:
            assert grad_err(gg[k], gr_ref[k]) < TOL_DOPRI_GRAD, (k, split)
        out[split] = (got, gg)
    assert relerr(out["1"][0], out["0"][0]) < TOL_DOPRI
    for k in KEYS:
        assert relerr(out["1"][1][k], out["0"][1][k]) < TOL_DOPRI_GRAD, k


def test_small_problem_with_training_scale_cotangents_vs_oracle_with_and_without_theta(pa, dev, oracle):
    """The regime DESIGN.md section 5 once blamed on the theta block of the adjoint norm (adjoint.py:72-78; the engine's
    controllers see [t, y, a] only): 700 genes, 6 trajectories, cotangents of training scale 1/(B N), where atol dominates
    every adjoint tolerance and the backward solve is loosely converged.  Measured (round 5): the oracle WITH and WITHOUT the
    block are 2e-7 apart here -- the block is not what parts engine and oracle; the reference algorithm's own sensitivity
    is: re-run at rtol x 0.9 / 1.1 its gradients move by more than the suite's tolerance.  The engine is held to that
    measured spread (like golden G12 and the full-size B-cell test), against both oracles."""
    N, H, B = 700, 40, 6
    p = rand_params(N, H, seed=77, std=0.05)
    net, onet = make_net(pa, dev, p), onet_of(oracle, p)
    r = np.random.RandomState(12)
    y0 = r.rand(B, N).astype(np.float32)
    t = np.tile(np.array([[0.0, 1.5]], np.float32), (B, 1))
    G = (r.randn(B, 2, N) / (B * N)).astype(np.float32)
    ref = oracle.odeint_per_sample(onet, y0, t, method="dopri5")
    adj_w, gr_w = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    adj_o, gr_o = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=False)
    theta = max([relerr(adj_o, adj_w)] + [relerr(gr_o[k], gr_w[k]) for k in KEYS])
    spread = {k: 0.0 for k in KEYS + ("adj",)}
    for f in (0.9, 1.1):
        adj_j, gr_j = oracle.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=True, rtol=1e-7 * f)
        spread["adj"] = max(spread["adj"], relerr(adj_j, adj_w))
        for k in KEYS:
            spread[k] = max(spread[k], relerr(gr_j[k], gr_w[k]))
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev))
    assert relerr(sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2), ref) < TOL_DOPRI
    (sol * torch.from_numpy(G.transpose(1, 0, 2).reshape(2, B, 1, N).copy()).to(dev)).sum().backward()
    gg = grads_of(net)
    a = y0t.grad.cpu().numpy().reshape(B, N)
    err = {"adj": max(relerr(a, adj_w), relerr(a, adj_o))}
    for k in KEYS:
        err[k] = max(relerr(gg[k], gr_w[k]), relerr(gg[k], gr_o[k]))
    print("N=700 B=6, 1/(BN) cotangents: oracle with vs without theta %.2e; engine vs oracle %s; oracle's rtol x 0.9/1.1 spread %s" %
          (theta, {k: "%.1e" % v for k, v in err.items()}, {k: "%.1e" % v for k, v in spread.items()}))
    assert theta < 1e-6
    for k, v in err.items():
        assert v < max(TOL_DOPRI_GRAD, 1.25 * spread[k]), (k, v, spread[k])


@pytest.mark.parametrize("N,H,B,method", [(350, 40, 17, "dopri5"), (600, 120, 5, "dopri5"), (350, 40, 64, "rk4")])
def test_training_step_recorded_into_a_graph(pa, dev, N, H, B, method):
    """phoenix_amd.GraphedStep: a step (parameter re-layout, forward solve, backward solve, gradient reduction, an SGD
    update) recorded into ONE HIP graph replays to bitwise the eager step's parameters, with the solver status left on the
    device; a step that fails raises from check_status() like the eager one."""
    from phoenix_amd import engine
    p = rand_params(N, H, seed=N + H + B, std=0.05)
    r = np.random.RandomState(3)
    y0 = torch.from_numpy(r.rand(B, 1, N).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.stack([np.array([0.0, 0.3 + 0.01 * b]) for b in range(B)]).astype(np.float32)).to(dev)
    G = torch.from_numpy((r.randn(2, B, 1, N) / (B * N)).astype(np.float32)).to(dev)

    def make():
        net = make_net(pa, dev, p)
        opt = torch.optim.SGD(net.parameters(), lr=0.05)

        def step():
            opt.zero_grad(set_to_none=False)
            y = y0.detach().requires_grad_(True)
            sol = pa.odeint_adjoint(net, y, t, method=method)
            torch.autograd.backward(sol, G)
            opt.step()
        return net, step

    net_e, step_e = make()
    for q in net_e.parameters():
        q.grad = torch.zeros_like(q)
    for _ in range(2 + 4):               # the graph's two warm-up runs (they update the parameters too) + four replays
        step_e()
    torch.cuda.synchronize()
    net_g, step_g = make()
    for q in net_g.parameters():
        q.grad = torch.zeros_like(q)
    gs = pa.GraphedStep(step_g, warmup=2)          # (the recording itself executes nothing)
    assert engine.status_mode() == "immediate" and len(gs.status_blocks) >= 1
    for _ in range(4):
        gs()
    gs.check_status()
    torch.cuda.synchronize()
    for a, b in zip(net_g.parameters(), net_e.parameters()):
        assert torch.equal(a, b)
    # an eager call after the replays sees the parameters the graph left (the layout cache was dropped)
    sol_e = pa.odeint(net_e, y0, t, method=method).detach()
    sol_g = pa.odeint(net_g, y0, t, method=method).detach()
    assert torch.equal(sol_e, sol_g)
    if method == "dopri5":
        # failure inside a replay: NaN initial states stop every trajectory with a status, read on request
        y0.fill_(float("nan"))
        gs()
        with pytest.raises((AssertionError, RuntimeError)):
            gs.check_status()
        y0.copy_(torch.from_numpy(r.rand(B, 1, N).astype(np.float32)))
        engine.forget_workspaces()


def test_zz_median_gradient_error_of_the_suite_meets_the_north_star_bar():
    """Runs last (file order): the MEDIAN engine-vs-reference error of every multi-step dopri5 gradient comparison this
    session made (oracle with the parameter block in the norm, goldens captured from torchdiffeq) is within the
    north-star 1e-5; single cases may sit at the accept/reject noise ceiling TOL_DOPRI_GRAD (golden G12: the reference's
    own gradients scatter 7.7e-6 median / 1.3e-5 max around the fp64 truth at rtol = 1e-7)."""
    if len(GRAD_ERRORS) < 20:
        pytest.skip("only %d gradient comparisons were made in this session (a partial run)" % len(GRAD_ERRORS))
    v = np.array([e for e, _ in GRAD_ERRORS])
    assert float(np.median(v)) <= 1e-5, (float(np.median(v)), float(v.max()))
