"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol that
include/phoenix_hip.h declares; host-side argument logic mirrors the reference's errors."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "phoenix_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(phx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from phoenix_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.phx_abi_version() == 7
    assert lib.phx_status_string(2).decode() == "underflow in dt"


def test_workspace_bytes_monotone_and_positive():
    from phoenix_amd import _lib
    lib = _lib.load()
    a = lib.phx_workspace_bytes(_lib.OP_ODEINT, 350, 40, 64, 5)
    b = lib.phx_workspace_bytes(_lib.OP_ODEINT, 350, 40, 1024, 5)
    c = lib.phx_workspace_bytes(_lib.OP_ADJOINT, 350, 40, 1024, 5)
    assert 0 < a < b < c
    assert lib.phx_workspace_bytes(_lib.OP_RHS_FORWARD, 0, 40, 1, 0) == 0


def test_null_arguments_are_rejected_without_a_gpu():
    from phoenix_amd import _lib
    lib = _lib.load()
    assert lib.phx_rhs_forward(None, None, None, 1, 0, None, 0, None) == 4  # PHX_ERR_BAD_ARG
    assert lib.phx_pack_weight_images(None, None, None) == 4
    assert lib.phx_layout_params(None, None, None, None, 1, 1, None, None, None) == 4
    assert lib.phx_prior_targets_sell(None, None, None, None, None, None, 1, 1, None) == 4
    # weight images: a pure function of (N, H); 0 where no MFMA plan exists
    a, b = lib.phx_weight_image_bytes(350, 40), lib.phx_weight_image_bytes(11165, 40)
    assert 0 < a < b and lib.phx_weight_image_bytes(11165, 200) > b and lib.phx_weight_image_bytes(0, 40) == 0
    assert lib.phx_odeint_calls_workspace_bytes(350, 40, 60, 10, 7) == 0          # 60 rows are not 7 equal calls


def test_no_cpu_fallback():
    import phoenix_amd
    net = phoenix_amd.ODENet("cpu", 16, neurons=4)
    y0 = torch.rand(1, 16)
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        phoenix_amd.odeint(net, y0, torch.tensor([0.0, 1.0]))
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        net(torch.tensor(0.0), y0)


def test_argument_errors_match_reference():
    import phoenix_amd
    net = phoenix_amd.ODENet("cpu", 16, neurons=4)
    y0 = torch.rand(1, 16)
    t = torch.tensor([0.0, 1.0])
    with pytest.raises(ValueError, match='Invalid method "foo"'):        # misc.py:184-186
        phoenix_amd.odeint(net, y0, t, method="foo")
    with pytest.raises(TypeError, match="`t` must be a floating point"):   # misc.py:118-120
        phoenix_amd.odeint(net, y0, torch.tensor([0, 1]))
    with pytest.raises(TypeError, match="`y0` must be a floating point"):
        phoenix_amd.odeint(net, torch.zeros(1, 16, dtype=torch.int64), t)
    with pytest.raises(NotImplementedError):
        phoenix_amd.odeint(net, y0, t, method="bosh3")
    # a module that is not a PHOENIX ODENet is integrated by the unfused torch stepper (tests/test_generic_gpu.py); the
    # engine-level recognition still says what it wants
    with pytest.raises(TypeError, match="ODENet only"):
        phoenix_amd.odenet.params_of(torch.nn.Linear(16, 16))
    lin = torch.nn.Linear(16, 16)
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):     # device tensors only, whatever the func
        phoenix_amd.odeint(lambda tt, y: lin(y), y0, t, method="euler")


def test_odenet_mirror_has_reference_structure():
    import phoenix_amd
    net = phoenix_amd.ODENet("cpu", 30, neurons=5)
    names = [n for n, _ in net.named_parameters()]
    # parameters() order of the reference
    assert names == ["gene_multipliers", "net_prods.linear_out.weight", "net_prods.linear_out.bias",
                     "net_sums.linear_out.weight", "net_sums.linear_out.bias", "net_alpha_combine.linear_out.weight"]
    assert net.net_alpha_combine.linear_out.weight.shape == (30, 10)
    assert net.gene_multipliers.shape == (1, 30)
    w = net.net_sums.linear_out.weight
    assert (w == 0).float().mean() >= 0.9   # nn.init.sparse_(sparsity=0.95)


def test_checkpoint_round_trip_in_the_references_four_file_format(tmp_path):
    """ODENet.save / load_model (odenet.py:100-133): four pickles `<name>_{prods,sums,alpha_comb,gene_multipliers}.pt`;
    a fresh network that loads them has identical parameters in the reference's parameters() order."""
    import phoenix_amd
    torch.manual_seed(3)
    a = phoenix_amd.ODENet("cpu", 21, neurons=6)
    with torch.no_grad():                     # the 95 %-sparse init leaves a 6-row layer all zero: give it content
        for prm in a.parameters():
            prm.add_(torch.randn_like(prm))
    fp = str(tmp_path / "model.pt")
    a.save(fp)
    for suffix in ("_prods", "_sums", "_alpha_comb", "_gene_multipliers"):
        assert os.path.exists(str(tmp_path / ("model" + suffix + ".pt")))
    torch.manual_seed(4)
    b = phoenix_amd.ODENet("cpu", 21, neurons=6)
    assert not torch.equal(a.net_sums.linear_out.weight, b.net_sums.linear_out.weight)
    b.load(fp)
    for (na, pa_), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa_, pb)


def test_no_wide_buffer_store_has_its_data_registers_overwritten_at_once():
    """gfx950 store-data hazard (tools/membench/store_war.hip): a VALU write to the data VGPRs of a 128-bit buffer store
    within the next wait state corrupts lanes 12..15 of every 16 of what is stored, and the compiler does not insert the
    wait states when the store's soffset is an SGPR.  The third-generation kernels guard every such store
    (phx_mfma_v3common.inc: bstore_guard); this checks the ISA the build actually produced."""
    import subprocess
    import sys
    from phoenix_amd import build
    build.build()
    listings = [build.listing_of(s) for s in build.sources() if os.path.basename(s) in build.LISTINGS]
    assert listings and all(os.path.exists(p) for p in listings), listings
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_store_hazard.py")] + listings,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_a_failed_solve_forgets_every_vouched_workspace():
    """after a launch that ended in an error (above all PHX_ERR_SYNC_TIMEOUT, whose bail-out leaves the
    exchange-set parity / `started` counters of the workspace header undefined) no cached workspace may be vouched for:
    `raise_for_status` clears the `ws_keep` bookkeeping, so the next solve fills its exchange buffers again."""
    from phoenix_amd import engine
    engine._ws_last[(0, 0, 1)] = ("some", "plan")
    engine.raise_for_status(torch.zeros(2, 3, dtype=torch.int32))        # all OK: nothing happens
    assert engine._ws_last
    with pytest.raises(AssertionError, match="max_num_steps"):
        engine.raise_for_status(torch.tensor([[0, 0, 0], [0, 1, 0]], dtype=torch.int32))   # the reference's assert, rk_common.py:154
    assert not engine._ws_last
    engine._ws_last[(0, 0, 1)] = ("some", "plan")
    with pytest.raises(RuntimeError, match="trajectory 2"):
        engine.raise_for_status(torch.tensor([0, 0, 6], dtype=torch.int32))                # a launch-level failure
    assert not engine._ws_last


def test_odeint_on_a_differentiable_odenet_is_routed_to_the_adjoint(monkeypatch):
    """`odeint` (odeint.py:30-74) returns an autograd-tracked solution in the reference; here a call whose inputs require
    grad goes through `odeint_adjoint` (checked without a GPU by intercepting it), a call under no_grad does not."""
    import phoenix_amd
    import sys
    od = sys.modules["phoenix_amd.odeint"]          # (the package re-exports the function under the module's name)
    net = phoenix_amd.ODENet("cpu", 16, neurons=4)
    y0, t = torch.rand(2, 1, 16), torch.tensor([0.0, 1.0])
    seen = {}

    def fake_adjoint(func, y0_, t_, **kw):
        seen.update(kw)
        return "routed"

    monkeypatch.setattr(od, "odeint_adjoint", fake_adjoint)
    assert phoenix_amd.odeint(net, y0, t, method="rk4") == "routed" and seen["_via_odeint"] and seen["method"] == "rk4"
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        phoenix_amd.odeint(net, y0, t)          # not routed: the plain forward solve (which has no CPU path)


def test_plan_switches_are_read_from_the_live_environment(monkeypatch):
    """engine._plan_env (the fast read of the PHX_* plan switches that keys workspaces and plans) follows os.environ as
    the tests and tools change it, and equals the plain os.environ.get reading"""
    import os
    from phoenix_amd import engine
    monkeypatch.delenv("PHX_V3C_HB", raising=False)
    monkeypatch.delenv("PHX_ADJ", raising=False)
    base = engine._plan_env()
    assert len(base) == len(engine._PLAN_ENV)
    monkeypatch.setenv("PHX_V3C_HB", "0")
    monkeypatch.setenv("PHX_ADJ", "v2")
    now = engine._plan_env()
    assert now != base
    plain = tuple(os.environ.get(k) for k in engine._PLAN_ENV)
    assert tuple(None if v is None else os.fsdecode(v) if isinstance(v, bytes) else v for v in now) == plain
    monkeypatch.delenv("PHX_V3C_HB")
    monkeypatch.delenv("PHX_ADJ")
    assert engine._plan_env() == base


def test_status_modes():
    """"captured" keeps device status blocks for check_captured_status and never blocks; host blocks are checked at once"""
    import torch
    from phoenix_amd import engine
    assert engine.status_mode() == "immediate"
    engine.set_status_mode("captured")
    try:
        engine.check_pending_status(wait=True)                     # nothing to wait for, no event calls
        engine.raise_for_status(torch.zeros(4, dtype=torch.int32))  # a host block: read right away
        with pytest.raises(AssertionError):
            engine.raise_for_status(torch.tensor([0, 1, 0], dtype=torch.int32))
        assert engine.captured_status == []
    finally:
        engine.set_status_mode("immediate")
    with pytest.raises(AssertionError):
        engine.set_status_mode("later")

