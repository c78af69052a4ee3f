"""TEST INFRASTRUCTURE ONLY (see oracle/phx_oracle.h): reference semantics of the Hill-kinetics rate expressions.

The expressions in `ode_system_functions_*.csv` are R source text emitted by GraphGRN_core.R:425-486; their only
function is `fAct` (GraphGRN_core.R:431-436) and their arithmetic is also valid Python, so the reference value of a
rate is Python's own evaluation of the shipped string in fp64.  Trajectories: scipy's LSODA at tight tolerances
(the reference integrates with deSolve::ode, which is lsoda).  Pinned only against the shipped expression files --
the R run itself (RNG draws, lsoda step sequence) cannot be reproduced in this image."""
import numpy as np


def fAct(TF, EC50=0.5, n=1.39):
    B = (EC50 ** n - 1) / (2 * EC50 ** n - 1)
    K_n = B - 1
    return B * TF ** n / (K_n + TF ** n)


def rhs(names, expressions, x):
    """rates of every gene at the states x [..., N] ('input gene' rows: 0)."""
    x = np.asarray(x, np.float64)
    env = {n: x[..., i] for i, n in enumerate(names)}
    env["fAct"] = fAct
    out = np.zeros_like(x)
    for g, e in enumerate(expressions):
        if e.strip() != "input gene":
            out[..., g] = eval(compile(e, "<eqn %s>" % names[g], "eval"), {"__builtins__": {}}, env)
    return out


def simulate(names, expressions, x0, times, rtol=1e-10, atol=1e-12):
    from scipy.integrate import solve_ivp
    codes = [None if e.strip() == "input gene" else compile(e, "<eqn>", "eval") for e in expressions]

    def f(_t, y):
        env = dict(zip(names, y))
        env["fAct"] = fAct
        return [0.0 if c is None else eval(c, {"__builtins__": {}}, env) for c in codes]

    sol = solve_ivp(f, (times[0], times[-1]), np.asarray(x0, np.float64), t_eval=times, method="LSODA", rtol=rtol, atol=atol)
    return sol.y.T


def interpret_programs(code, consts, off, length, x):
    """fp64 interpreter of the postfix programs phoenix_amd.simulator compiles (same opcodes as csrc/phx_hill.inc):
    checks the COMPILER against `rhs` above without a GPU."""
    x = np.asarray(x, np.float64)
    out = np.zeros_like(x)
    for g in range(len(off)):
        st = []
        for op, arg in code[off[g]: off[g] + length[g]]:
            if op == 0:
                st.append(np.full(x.shape[:-1], consts[arg]))
            elif op == 1:
                st.append(x[..., arg])
            elif op == 6:
                st[-1] = -st[-1]
            elif op == 7:
                b, k, n = consts[arg: arg + 3]
                tn = np.where(st[-1] > 0, np.abs(st[-1]) ** n, 0.0)
                st[-1] = b * tn / (k + tn)
            else:
                r = st.pop()
                st[-1] = st[-1] + r if op == 2 else st[-1] - r if op == 3 else st[-1] * r if op == 4 else st[-1] / r
        if st:
            out[..., g] = st[0]
    return out
