"""Row f4 of SURVEY.md section 8: the ground-truth Hill-kinetics simulator of the reference's in-silico data.

The reference generates its 350 / 690-gene training data in R: `GraphGRN_core.R:425-486` turns the regulatory graph
into one rate expression per gene (shipped as `ground_truth_simulator/clean_data/ode_system_functions_*.csv`, e.g.
`((1 * fAct(GAL11, 0.408, 1.797)) * 1 - 1 * A1_ALPHA2) / 1`; AND = product, OR = a + b - ab, NOT = 1 - a are already
spelled out in the text; `"input gene"` rows are held constant) and `SimulationGRN_core_init_var.R:203-247` draws
initial states and integrates every sample with `deSolve::ode`.  R is not available in this image, so this module
restates the simulator from those files: the expressions are compiled on the host into postfix programs and
evaluated / integrated on the GPU (`phx_hill_rhs`, `phx_hill_simulate` in `csrc/phx_hill.inc`), and the result is
written in the reference's CSV wire format (`csvreader.py:58-73`) so that `DataHandler.fromcsv` can train on it.
Parity: the RHS is checked against Python's own evaluation of the shipped expression strings (fp64) and the
trajectories against a tight-tolerance scipy solve of the same strings (golden G11); the R run itself cannot be
reproduced here (its RNG draws are not recorded), which DESIGN.md states.
"""
import ast
import csv
import ctypes as C

import numpy as np
import torch

from . import _lib, engine

OPS = {"PUSHC": 0, "PUSHX": 1, "ADD": 2, "SUB": 3, "MUL": 4, "DIV": 5, "NEG": 6, "FACT": 7}
MAX_STACK = 24   # HILL_STACK in csrc/phx_hill.inc


def fact_constants(ec50, n):
    """GraphGRN_core.R:431-434: B = (EC50^n - 1) / (2 EC50^n - 1), K_n = B - 1."""
    b = (ec50 ** n - 1.0) / (2.0 * ec50 ** n - 1.0)
    return b, b - 1.0, n


class _Compiler(ast.NodeVisitor):
    def __init__(self, index):
        self.index, self.code, self.consts, self.depth, self.max_depth = index, [], [], 0, 0

    def _push(self, op, arg=0):
        self.code.append((OPS[op], arg))
        self.depth += 1
        self.max_depth = max(self.max_depth, self.depth)

    def _const(self, *vals):
        k = len(self.consts)
        self.consts.extend(float(v) for v in vals)
        return k

    def visit_Expression(self, node):
        self.visit(node.body)

    def visit_Constant(self, node):
        self._push("PUSHC", self._const(node.value))

    def visit_Name(self, node):
        if node.id not in self.index:
            raise ValueError("unknown gene '%s' in a rate expression" % node.id)
        self._push("PUSHX", self.index[node.id])

    def visit_UnaryOp(self, node):
        self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            self.code.append((OPS["NEG"], 0))
        elif not isinstance(node.op, ast.UAdd):
            raise ValueError("unsupported unary operator in a rate expression")

    def visit_BinOp(self, node):
        ops = {ast.Add: "ADD", ast.Sub: "SUB", ast.Mult: "MUL", ast.Div: "DIV"}
        if type(node.op) not in ops:
            raise ValueError("unsupported operator in a rate expression")
        self.visit(node.left)
        self.visit(node.right)
        self.code.append((OPS[ops[type(node.op)]], 0))
        self.depth -= 1

    def visit_Call(self, node):
        if not (isinstance(node.func, ast.Name) and node.func.id == "fAct" and len(node.args) == 3):
            raise ValueError("only fAct(TF, EC50, n) calls are supported in rate expressions")
        self.visit(node.args[0])
        ec50, n = (float(ast.literal_eval(a)) for a in node.args[1:])
        self.code.append((OPS["FACT"], self._const(*fact_constants(ec50, n))))

    def generic_visit(self, node):
        raise ValueError("unsupported syntax in a rate expression: %s" % type(node).__name__)


# ---------------------------------------------------------------------------------------------------------------------
# The equation builder (GraphGRN_core.R:163-190, 221-236, 312-330): how the reference turns a node's logic equation and the
# properties of its incoming edges into the rate-expression TEXT that it ships in ode_system_functions_*.csv.
#   NODE(tf)  ->  "<weight> * fAct(<tf>, <EC50>, <n>)"            (NODE: R:163-169, generateActivationEqnReg: R:312-330)
#   NOT(e)    ->  "(1 - e)"                                       (R:171-174)
#   AND(a, b) ->  "(a * b)"                                       (R:176-179)
#   OR(a, b)  ->  "(a + b - a * b)"                               (R:182-185)
#   rate      ->  "((act) * spmax - spdeg * NAME) / tau"          (generateRateEqn_modified: R:221-236)
# A logic tree is ("node", tf) | ("not", t) | ("and", a, b) | ("or", a, b).  The logic equations themselves are drawn at
# random when the reference builds its graph (genericLogicEqn, R:727-800: `runif(...) < propor`) and are NOT shipped, so
# they are recovered from the shipped expression text (`logic_of_expression`); what the pin (golden G16,
# tests/test_simulator_cpu.py) then checks is that `rate_expression(logic, edge properties)` reproduces every shipped
# expression text for text from edge_properties_G*.csv -- weights, EC50 and Hill constants as R printed them -- and that an
# edge is negated exactly where its `activation` flag is FALSE.
# ---------------------------------------------------------------------------------------------------------------------
def activation_text(edge):
    """`weight * fAct(from, EC50, n)` of one edge; `edge`: dict with the CSV's text fields from, weight, EC50, n"""
    return "%s * fAct(%s, %s, %s)" % (edge["weight"], edge["from"], edge["EC50"], edge["n"])


def logic_text(tree, node, edges):
    kind = tree[0]
    if kind == "node":
        return activation_text(edges[(tree[1], node)])
    if kind == "not":
        return "(1 - %s)" % logic_text(tree[1], node, edges)
    a, b = logic_text(tree[1], node, edges), logic_text(tree[2], node, edges)
    if kind == "and":
        return "(%s * %s)" % (a, b)
    if kind == "or":
        return "(%s + %s - %s * %s)" % (a, b, a, b)
    raise ValueError("unknown logic operator %r" % (kind,))


def rate_expression(node, tree, edges, spmax="1", spdeg="1", tau="1"):
    """generateRateEqn_modified (R:221-236) for a non-input node"""
    return "((%s) * %s - %s * %s) / %s" % (logic_text(tree, node, edges), spmax, spdeg, node, tau)


def logic_of_expression(expr, node):
    """inverse of `rate_expression`: (logic tree, spmax, spdeg, tau) recovered from a shipped expression string"""
    import re
    m = re.fullmatch(r"\(\((.*)\) \* (\S+) - (\S+) \* %s\) / (\S+)" % re.escape(node), expr)
    if not m:
        raise ValueError("not a rate expression of node %s: %s" % (node, expr[:80]))
    body, spmax, spdeg, tau = m.groups()
    pos = [0]

    def peek(txt):
        return body.startswith(txt, pos[0])

    def eat(txt):
        if not peek(txt):
            raise ValueError("expected %r at %d in %s" % (txt, pos[0], body[:120]))
        pos[0] += len(txt)

    def term():
        if peek("(1 - "):
            eat("(1 - ")
            t = term()
            eat(")")
            return ("not", t)
        if peek("("):
            eat("(")
            a = term()
            if peek(" * "):
                eat(" * ")
                b = term()
                eat(")")
                return ("and", a, b)
            eat(" + ")
            b = term()
            eat(" - ")
            a2 = term()
            eat(" * ")
            b2 = term()
            eat(")")
            if a2 != a or b2 != b:
                raise ValueError("malformed OR in %s" % body[:120])
            return ("or", a, b)
        m2 = re.compile(r"([0-9.eE+-]+) \* fAct\((\w+), ([0-9.eE+-]+), ([0-9.eE+-]+)\)").match(body, pos[0])
        if not m2:
            raise ValueError("expected an activation term at %d in %s" % (pos[0], body[:120]))
        pos[0] = m2.end()
        return ("node", m2.group(2), m2.group(1), m2.group(3), m2.group(4))

    t = term()
    if pos[0] != len(body):
        raise ValueError("trailing text in %s" % body[:120])

    def strip(x):   # leaves keep only the regulator's name: weight / EC50 / n come from the edge table when rebuilding
        return ("node", x[1]) if x[0] == "node" else (x[0],) + tuple(strip(y) for y in x[1:])

    def leaves(x, neg=False, out=None):
        out = [] if out is None else out
        if x[0] == "node":
            out.append((x[1], x[2], x[3], x[4], neg))
        elif x[0] == "not":
            leaves(x[1], not neg, out)
        else:
            for y in x[1:]:
                leaves(y, neg, out)
        return out

    return strip(t), spmax, spdeg, tau, leaves(t)


# ---------------------------------------------------------------------------------------------------------------------
# External inputs of a simulated data set (SimulationGRN_core_init_var.R:40-160).  Host-side sampling, as in the reference
# (R); the draws come from numpy's generator, so a data set is statistically, not bitwise, the reference's.
# ---------------------------------------------------------------------------------------------------------------------
def create_input_models(input_names, prop_bimodal=0.0, rng=None):
    """`createInputModels` (R:40-68): every input gene gets a one- or two-component normal mixture.
    Bimodal (probability `prop_bimodal`): weights (p, 1-p), p ~ U(0.2, 0.8); means Beta(10,100) and Beta(10,10).
    Unimodal: mean Beta(10,10).  sd = max(Beta(15,15) * min(mean, 1-mean)/3, 0.01) per component."""
    rng = np.random.default_rng() if rng is None else rng
    models = {}
    for name in input_names:
        if rng.random() < prop_bimodal:
            p = rng.uniform(0.2, 0.8)
            prop = np.array([p, 1.0 - p])
            mean = np.array([rng.beta(10, 100), rng.beta(10, 10)])
        else:
            prop = np.array([1.0])
            mean = np.array([rng.beta(10, 10)])
        maxsd = np.minimum(mean, 1.0 - mean) / 3.0
        sd = np.array([max(rng.beta(15, 15) * x, 0.01) for x in maxsd])
        models[name] = {"prop": prop, "mean": mean, "sd": sd}
    return models


def vine_correlation(d, betaparam=5.0, rng=None):
    """`vineS` (R:76-98): random correlation matrix by the C-vine construction of Lewandowski, Kurowicka and Joe (2009):
    partial correlations 2 Beta(b, b) - 1 (smaller `betaparam` = stronger correlations) converted to correlations, then a
    random simultaneous permutation of rows and columns.  The R loop starts at k = 2, so the first variable stays
    uncorrelated with the others before the permutation; kept.  d < 3: identity (the R loop bounds are not meaningful)."""
    rng = np.random.default_rng() if rng is None else rng
    S = np.eye(d)
    if d < 3:
        return S
    P = np.zeros((d, d))
    for k in range(1, d - 1):             # R: k in 2:(d-1), 1-based
        for i in range(k + 1, d):         # R: i in (k+1):d
            P[k, i] = (rng.beta(betaparam, betaparam) - 0.5) * 2.0
            p = P[k, i]
            for l in range(k - 1, -1, -1):    # R: l in (k-1):1
                p = p * np.sqrt((1.0 - P[l, i] ** 2) * (1.0 - P[l, k] ** 2)) + P[l, i] * P[l, k]
            S[k, i] = S[i, k] = p
    perm = rng.permutation(d)
    return S[np.ix_(perm, perm)]


def generate_input_data(models, numsamples, cor_strength=5.0, input_gene_var=1.0, rng=None):
    """`generateInputData` (R:101-160): [numsamples, n_inputs] external inputs in [0, 1] and the mixture classes.
    Every input is drawn from its mixture (component sd scaled by `input_gene_var`), out-of-range values are redrawn
    until none is left; with `cor_strength > 0` the unimodal inputs are then made rank-correlated: a multivariate normal
    sample with a `vine_correlation` matrix supplies the ranks, the sorted marginal draws supply the values (marginals
    unchanged); bimodal inputs keep their own draws ("avoid correlated bimodal inputs")."""
    rng = np.random.default_rng() if rng is None else rng
    names = list(models)
    X = np.full((numsamples, len(names)), -1.0)
    classf = {}
    for c, name in enumerate(names):
        m = models[name]
        mix = rng.choice(len(m["prop"]), size=numsamples, p=m["prop"])
        while True:
            out = (X[:, c] < 0) | (X[:, c] > 1)
            if not out.any():
                break
            for comp in range(len(m["prop"])):
                sel = out & (mix == comp)
                X[sel, c] = rng.normal(m["mean"][comp], m["sd"][comp] * input_gene_var, size=int(sel.sum()))
        if len(m["prop"]) > 1:
            classf[name] = mix + 1            # R's components are numbered from 1
    if cor_strength > 0 and numsamples > 1 and len(names) > 0:
        dm = np.sort(X, axis=0)
        cov = vine_correlation(len(names), cor_strength, rng)
        cordata = rng.multivariate_normal(np.zeros(len(names)), cov, size=numsamples, method="eigh")
        for c, name in enumerate(names):
            if name in classf:
                cordata[:, c] = X[:, c]
            else:
                ranks = np.argsort(np.argsort(cordata[:, c], kind="stable"), kind="stable")
                cordata[:, c] = dm[ranks, c]
        X = cordata
    return X, classf


def read_ode_system(path):
    """`ode_system_functions_*.csv`: columns node, eqn -> (names, expressions) in file order."""
    with open(path, newline="") as fh:
        rows = list(csv.DictReader(fh))
    return [r["node"] for r in rows], [r["eqn"] for r in rows]


class HillSystem:
    """The compiled rate expressions of one regulatory network, resident on the device."""

    def __init__(self, names, expressions, device="cuda"):
        self.names, self.expressions = list(names), list(expressions)
        self.N = len(self.names)
        index = {n: i for i, n in enumerate(self.names)}
        code, consts, off, length = [], [], [], []
        self.is_input = np.zeros(self.N, bool)
        for g, expr in enumerate(self.expressions):
            off.append(len(code))
            if expr.strip() == "input gene":            # GraphGRN_core.R:477-479
                self.is_input[g] = True
                length.append(0)
                continue
            c = _Compiler(index)
            c.visit(ast.parse(expr, mode="eval"))
            if c.max_depth > MAX_STACK:
                raise ValueError("rate expression of %s needs a deeper evaluation stack (%d)" % (self.names[g], c.max_depth))
            base = len(consts)
            code.extend((op, arg + base if op in (OPS["PUSHC"], OPS["FACT"]) else arg) for op, arg in c.code)
            consts.extend(c.consts)
            length.append(len(c.code))
        self.code_host = np.asarray(code, np.int32).reshape(-1, 2)
        self.consts_host = np.asarray(consts, np.float64)
        self.off_host, self.len_host = np.asarray(off, np.int32), np.asarray(length, np.int32)
        self.device = torch.device(device)
        if self.device.type == "cuda":
            self.code = torch.from_numpy(self.code_host).to(self.device)
            self.consts = torch.from_numpy(self.consts_host.astype(np.float32)).to(self.device)
            self.off = torch.from_numpy(self.off_host).to(self.device)
            self.len = torch.from_numpy(self.len_host).to(self.device)

    @classmethod
    def from_csv(cls, path, device="cuda"):
        return cls(*read_ode_system(path), device=device)

    # ---- device
    def _args(self):
        p = engine._p
        return p(self.code), p(self.off), p(self.len), p(self.consts)

    def rhs(self, x):
        """rates [B,N] at states x [B,N] (what `my_ode` returns, GraphGRN_core.R:441-470)."""
        engine._require_gpu(x, "x")
        x2 = x.detach().reshape(-1, self.N).contiguous().float()
        out = torch.empty_like(x2)
        engine._check_call(_lib.load().phx_hill_rhs(*self._args(), engine._p(x2), engine._p(out), x2.shape[0], self.N,
                                                    engine._stream_ptr()))
        return out.reshape(x.shape)

    def simulate(self, x0, times, dt_max=0.01):
        """states [T,B,N] at `times` (first row = x0), SimulationGRN_core_init_var.R:226-229,243-244."""
        engine._require_gpu(x0, "x0")
        x2 = x0.detach().reshape(-1, self.N).contiguous().float()
        t64 = torch.as_tensor(np.asarray(times, np.float64), device=x2.device)
        out = torch.empty((t64.numel(), x2.shape[0], self.N), dtype=torch.float32, device=x2.device)
        engine._check_call(_lib.load().phx_hill_simulate(*self._args(), engine._p(x2), engine._p(t64), t64.numel(),
                                                         C.c_double(dt_max), engine._p(out), x2.shape[0], self.N,
                                                         engine._stream_ptr()))
        return out

    def sample_initial(self, numsamples, output_gene_var=1.0, rng=None, input_models=None, cor_strength=5.0,
                       input_gene_var=1.0, prop_bimodal=0.0):
        """Initial states of `simDataset` (SimulationGRN_core_init_var.R:165-213).  Regulated genes: Beta(2/var, 2/var)
        shifted by the gene's own U(-.25, .25), clipped to [0, 1] (R:206-213).  Input genes ("input gene" rows of the
        expression file, rate 0): drawn from their input models with rank-correlated unimodal inputs (R:177-184,
        `generate_input_data`); `input_models=None` creates the models first (`create_input_models`, R:32-34)."""
        rng = np.random.default_rng() if rng is None else rng
        a = 2.0 / output_gene_var
        x = rng.beta(a, a, size=(numsamples, self.N)) + rng.uniform(-0.25, 0.25, size=(1, self.N))
        x = np.clip(x, 0.0, 1.0)
        if self.is_input.any():
            in_names = [n for n, f in zip(self.names, self.is_input) if f]
            if input_models is None:
                input_models = create_input_models(in_names, prop_bimodal, rng)
            xin, self.last_input_classes = generate_input_data({n: input_models[n] for n in in_names}, numsamples,
                                                                cor_strength, input_gene_var, rng)
            x[:, self.is_input] = xin
        return x.astype(np.float32)


def generate_dataset(system, numsamples, time_stamps=(0.0, 2.0, 3.0, 7.0, 9.0), expnoise=0.0, dt_max=0.01, rng=None,
                     path=None, derivative=False, **initial_kw):
    """One in-silico data set like the reference's `example_creator...R` run: `numsamples` trajectories at
    `time_stamps` (example_creator_for_chalmers_codebase_0noise.R:134-139), optional Gaussian measurement noise
    (`addNormNoise`, SimulationGRN_core_init_var.R:1-3, 260-265); written with `writecsv` if `path` is given.
    `derivative=True` is the reference's `get_derivative_instead` branch (R:222-240): the table holds the rates at the
    solution's states instead of the states, and zeros in the input columns.  `initial_kw` goes to
    `HillSystem.sample_initial` (input models, `cor_strength`, `input_gene_var`, `output_gene_var`, `prop_bimodal`).
    Returns (data_np, t_np) in `readcsv`'s shapes."""
    from .data import writecsv
    rng = np.random.default_rng() if rng is None else rng
    x0 = torch.from_numpy(system.sample_initial(numsamples, rng=rng, **initial_kw)).to(system.device)
    sol = system.simulate(x0, time_stamps, dt_max)                          # [T, S, N]
    if derivative:
        sol = system.rhs(sol.reshape(-1, system.N)).reshape(sol.shape)      # rate 0 for input genes = the zero columns
    sol = sol.cpu().numpy()
    if expnoise > 0:
        sol = sol + rng.normal(0.0, expnoise, size=sol.shape).astype(np.float32)
    data_np = [sol[:, s].reshape(len(time_stamps), 1, system.N) for s in range(numsamples)]
    t_np = [np.asarray(time_stamps, np.float64) for _ in range(numsamples)]
    if path is not None:
        writecsv(path, system.N, numsamples, data_np, t_np)
    return data_np, t_np
