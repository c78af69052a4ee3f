"""Row f4: compiler of the Hill-kinetics rate expressions (host logic) against golden G11 =
Python's own fp64 evaluation of the shipped 350-gene expression strings."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden



def _system(g, device, tmp_path=None):
    """HillSystem of the fixture's 350-gene network; through the CSV wire format when a directory is given."""
    from phoenix_amd.simulator import HillSystem
    names, eqns = [str(x) for x in g["names"]], [str(x) for x in g["eqns"]]
    if tmp_path is None:
        return HillSystem(names, eqns, device=device)
    import csv
    path = os.path.join(str(tmp_path), "ode_system.csv")
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_ALL)
        w.writerow(["node", "eqn"])
        w.writerows(zip(names, eqns))
    return HillSystem.from_csv(path, device=device)


def test_compiled_programs_match_the_expression_strings(tmp_path):
    g = load_golden("g11_hill")
    sys_ = _system(g, "cpu", tmp_path)
    assert sys_.N == 350 and int(sys_.is_input.sum()) == 74          # 74 "input gene" rows in the shipped network
    from oracle import hill_oracle
    got = hill_oracle.interpret_programs(sys_.code_host, sys_.consts_host, sys_.off_host, sys_.len_host, g["X"])
    assert np.max(np.abs(got - g["rates"])) < 1e-12
    assert np.all(got[:, sys_.is_input] == 0)


def test_oracle_fact_matches_the_r_definition():
    from oracle import hill_oracle
    from phoenix_amd.simulator import fact_constants
    # GraphGRN_core.R:431-436 at the default parameters and at one shipped edge
    for ec50, n, tf in ((0.5, 1.39, 0.3), (0.408142132172361, 1.79729779304471, 0.9)):
        b, k, _ = fact_constants(ec50, n)
        assert abs(hill_oracle.fAct(tf, ec50, n) - b * tf ** n / (k + tf ** n)) < 1e-15
    assert abs(hill_oracle.fAct(1.0, 0.5, 1.39) - 1.0) < 1e-12          # saturates at TF = 1


def test_unsupported_syntax_is_rejected():
    from phoenix_amd.simulator import HillSystem
    with pytest.raises(ValueError):
        HillSystem(["A", "B"], ["A ** 2", "input gene"], device="cpu")
    with pytest.raises(ValueError):
        HillSystem(["A", "B"], ["fAct(C, 0.5, 1.2) - A", "input gene"], device="cpu")


@pytest.mark.gpu
def test_hill_rhs_and_trajectories_on_device(tmp_path):
    """phx_hill_rhs / phx_hill_simulate (fp32, RK4 sub-steps of 0.01) against the fp64 oracle values."""
    from phoenix_amd.data import readcsv
    from phoenix_amd.simulator import generate_dataset
    g = load_golden("g11_hill")
    dev = torch.device("cuda:0")
    sys_ = _system(g, dev)
    rates = sys_.rhs(torch.from_numpy(g["X"]).float().to(dev)).cpu().numpy()
    assert np.max(np.abs(rates - g["rates"])) < 5e-6
    traj = sys_.simulate(torch.from_numpy(g["x0"]).float().to(dev), g["times"]).cpu().numpy()
    assert traj.shape == g["traj"].shape
    assert np.max(np.abs(traj - g["traj"])) / np.max(np.abs(g["traj"])) < 1e-5
    assert np.array_equal(traj[0], g["x0"].astype(np.float32))
    # input genes stay constant; a generated data set round-trips through the reference's CSV wire format
    assert np.array_equal(traj[-1][:, sys_.is_input], traj[0][:, sys_.is_input])
    path = str(tmp_path / "sim.csv")
    data_np, t_np = generate_dataset(sys_, 5, rng=np.random.default_rng(3), path=path)
    np.random.seed(0)
    back = readcsv(path, "cpu", 0.0, 1.0)
    assert back[4] == 350 and back[5] == 5
    assert np.allclose(back[0][2], data_np[2], rtol=0, atol=1e-6) and np.array_equal(back[2][0], t_np[0])
    assert all(np.all((d >= -1e-4) & (d <= 1.0 + 1e-4)) for d in data_np)   # Hill dynamics keep expression in [0, 1]
