#!/usr/bin/env python3
"""bench.py -- throughput of the PHOENIX NeuralODE hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic trajectories:
    forward  odeint(ODENet, y0[B,1,N], t[B,2])       (one persistent launch, all samples)
  + backward OdeintAdjointMethod.backward            (one persistent launch, all samples), entered through autograd
             with the loss cotangent dL/dsol handed to `torch.autograd.backward` (a loss head is not part of the path;
             the reference's own two losses are timed in the separate `training_step` object)
  + the per-step parameter re-layout a real optimizer step forces (version bump) + the gradient reduction kernel
  + (N GPUs > 1) one grouped RCCL all-reduce of the P = 4HN+2H+N gradient floats.
Default workload = BASELINE.json config C4 (the one the north-star target is quoted on):
breast-cancer scale N=11165 genes, H=40, 256 trajectory intervals per GPU, dopri5 rtol 1e-7 / atol 1e-9.

`--gpus N` with N > 1 and no RANK in the environment: this process starts N ranks itself (torch.distributed.run as a
child, one rank per device, before anything touches a GPU) and exits with the child's code; rank 0 prints the line.
Scaling: N = 1 and `--scaling weak` (default): every rank integrates its own 256 trajectories (no data-path collective).
With N > 1 the default also measures BASELINE.json config 4 as written -- "256 trajectory batch sharded 1/2/4/8" --
in the same run: `extra.strong` holds the step time / rate of the ONE 256-trajectory batch sharded over the ranks (loss
normalised by the GLOBAL batch); `--scaling strong` makes that the headline instead.

Metric: gene x trajectory RHS evaluations per second = sum over samples of (forward NFE + augmented NFE) * N / time.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, H, B per GPU, method, t-row, y0 sampler, description)
    "breast": dict(N=11165, H=40, B=256, method="dopri5", t=[0.0, 0.0051],
                   desc="C4 breast_cancer ~11k-gene pseudotime, 256 trajectory intervals/GPU, dopri5 + adjoint"),
    "insilico": dict(N=350, H=40, B=1024, method="rk4", t=[0.0, 2.0, 3.0, 7.0, 9.0],
                     desc="C2 in-silico 350-gene sim, 1024 trajectories x 4 intervals, fused 3/8-rule rk4 + adjoint"),
    "yeast": dict(N=2000, H=120, B=23, method="dopri5", t=[0.0, 5.0], init="reference",
                  desc="C3 yeast ~2000-gene oscillatory, 23 pairs, dopri5 + adjoint (reference 95%-sparse init)"),
    "bcell": dict(N=14691, H=200, B=256, method="dopri5", t=[0.0, 1.0], init="reference",
                  desc="C5 B-cell ~15k-gene, H=200, 256 trajectories/GPU, dopri5 + adjoint (reference 95%-sparse init)"),
}


def make_problem(wl, device, seed):
    import phoenix_amd
    N, H, B = wl["N"], wl["H"], wl["B"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    net = phoenix_amd.ODENet("cpu", N, neurons=H)
    with torch.no_grad():
        if wl.get("init", "dense") == "dense":
            # trained-like dense factors (Adam densifies the 95%-sparse init), SURVEY 8d
            for lin in (net.net_sums.linear_out, net.net_prods.linear_out, net.net_alpha_combine.linear_out):
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.05)
        # else: keep the reference's nn.init.sparse_(0.95, std 0.05) (dense N(0,0.05) weights blow the
        # yeast-range dynamics up over dt=5: both the oracle and the engine then report 'underflow in dt')
        net.gene_multipliers.copy_(torch.rand(1, N, generator=g))
    net = net.to(device)
    if wl is WORKLOADS["yeast"]:
        y0 = (torch.randn(B, 1, N, generator=g) * 0.4).clamp_(-2.5, 4.0)
    else:
        y0 = (torch.randn(B, 1, N, generator=g) * 0.15 + 0.5).clamp_(0.03, 1.07)
    t = torch.tensor(wl["t"], dtype=torch.float32).repeat(B, 1)
    return net, y0.to(device), t.to(device)


def one_step(net, y0, t, G, method, world):
    import phoenix_amd
    from phoenix_amd import parallel
    for p in net.parameters():
        p.grad = None
        # A training step changes the parameters (opt.step()), so the engine re-lays them out every step (transposed
        # W_alpha copy + packed LDS weight images).  The version bump makes the timed step pay for that like a real one
        # does -- without it the layout cache hits on every step and that work sits outside `value`.
        torch.autograd.graph.increment_version(p)
    y = y0.detach().requires_grad_(True)
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=method)
    torch.autograd.backward(sol, G)          # G = dL/dsol [T,B,1,N]: the backward solve is entered with it as it is
    if world > 1:
        parallel.allreduce_grads(net)
    return sol


def cpu_baseline_batched(wl, net, y0, t, G):
    """BASELINE.md section 3, shape (ii): the PyTorch-CPU formulation as ONE batched call over y0[B,1,N] (shared step
    control, dense GEMMs on every host core), forward solve + adjoint backward -- oracle/torch_baseline.py, a
    restatement of the reference pinned against the C oracle.  Whole workload, one warm-up on a quarter of it."""
    from oracle import torch_baseline as tb
    P = lambda x: x.detach().cpu()
    tnet = tb.Net(P(net.net_sums.linear_out.weight), P(net.net_sums.linear_out.bias),
                  P(net.net_prods.linear_out.weight), P(net.net_prods.linear_out.bias),
                  P(net.net_alpha_combine.linear_out.weight), P(net.gene_multipliers))
    N = wl["N"]
    yc, tc, Gc = P(y0), P(t)[0].double(), P(G)
    B = yc.shape[0]

    def run(nb):
        t0 = time.perf_counter()
        sol, nf = tb.odeint(tnet, yc[:nb], tc)
        t1 = time.perf_counter()
        _, _, nb_aug = tb.adjoint_backward(tnet, tc, sol, Gc[:, :nb])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, nf, nb_aug

    # warm-up 1 (at full size), best of 3 -- for every thread count of a sweep, because torch's
    # default (one thread per hardware thread of a 256-core host) oversubscribes the ~17 small ops of an evaluation
    cores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    sweep = sorted({n for n in (8, 16, 32, 64, 128, default_threads) if n <= max(cores, 8)})
    budget_s, t_begin = 150.0, time.perf_counter()
    per_threads, best = {}, None
    try:
        for nth in sweep:
            torch.set_num_threads(nth)
            tw = run(B)                               # warm-up at full size
            reps = []
            for _ in range(3):
                reps.append(run(B))
                if time.perf_counter() - t_begin > budget_s:
                    break
            tf, tbw, nf, nba = min(reps, key=lambda r: r[0] + r[1])
            rate = (nf + nba) * B * N / (tf + tbw)
            per_threads[str(nth)] = {"evals_per_s": rate, "forward_s": tf, "adjoint_s": tbw, "warmup_s": tw[0] + tw[1],
                                     "times_s": [round(r[0] + r[1], 4) for r in reps]}
            if best is None or rate > best[0]:
                best = (rate, nth, tf, tbw, nf, nba, len(reps))
            if time.perf_counter() - t_begin > budget_s:
                break
    finally:
        torch.set_num_threads(default_threads)
    rate, nth, tf, tbw, nf, nba, nrep = best
    out = {"value": rate, "unit": "gene*trajectory RHS evals/s",
           "cores": nth, "host_cores": cores, "threads": nth, "repeats": nrep, "kind": "port",
           "formulation": "one batched odeint_adjoint call (PyTorch CPU, shared step control)",
           "sample": "%d of %d trajectories, best of %d after a full-size warm-up at %d threads: forward %.3f s + adjoint %.3f s"
                     % (B, B, nrep, nth, tf, tbw),
           "forward_evals_per_s": nf * B * N / tf, "augmented_evals_per_s": nba * B * N / tbw,
           "nfe_forward": int(nf) * B, "nfe_augmented": int(nba) * B, "thread_sweep": per_threads}
    # SURVEY / BASELINE.md section 2 probe: the same formulation reached 1.4e8 forward evals/s at C4 on 8 cores
    if wl is WORKLOADS["breast"]:
        out["survey_probe_forward_evals_per_s"] = 1.4e8
        if out["forward_evals_per_s"] < 1.4e8:
            out["note"] = ("forward rate below the survey's 8-core probe (1.4e8) at every thread count tried (%s): "
                           "this host's cores are slower per thread / the run was budget-capped" % ",".join(per_threads))
    return out


def cpu_baseline(wl, net, y0, t, G, seconds_budget=20.0):
    """The CPU oracle (a port of the reference's algorithm; the Python reference cannot travel to the GPU
    box) on a bounded sample of the same workload, reference-shaped: per-sample loop, theta block in the
    adjoint norm, OpenMP across samples on all host cores."""
    from oracle import oracle as orc
    orc.build()
    P = lambda x: x.detach().cpu().numpy()
    onet = orc.Net(P(net.net_sums.linear_out.weight), P(net.net_sums.linear_out.bias),
                   P(net.net_prods.linear_out.weight), P(net.net_prods.linear_out.bias),
                   P(net.net_alpha_combine.linear_out.weight), P(net.gene_multipliers))
    N = wl["N"]
    cores = os.cpu_count() or 1
    y = P(y0).reshape(-1, N)
    tt = P(t)
    T = tt.shape[1]
    Gn = P(G).reshape(T, -1, N).transpose(1, 0, 2)

    def run(nsamp):
        t0 = time.perf_counter()
        sol, nfe_f, _ = orc.odeint_per_sample(onet, y[:nsamp], tt[:nsamp], method=wl["method"], nthreads=cores,
                                              return_stats=True)
        _, _, nfe_b, _ = orc.adjoint_backward_per_sample(onet, tt[:nsamp], sol, np.ascontiguousarray(Gn[:nsamp]),
                                                         method=wl["method"], theta_in_norm=True, nthreads=cores,
                                                         return_stats=True)
        return time.perf_counter() - t0, nfe_f, nfe_b

    nsamp = min(cores, y.shape[0])
    dt, nf, nb = run(nsamp)                     # calibration pass (also the warm-up)
    scale = max(1, int(seconds_budget / max(dt, 1e-3)))
    nsamp2 = min(y.shape[0], nsamp * scale)
    if nsamp2 > nsamp:
        dt, nf, nb = run(nsamp2)
        nsamp = nsamp2
    return {"value": (nf + nb) * N / dt, "unit": "gene*trajectory RHS evals/s", "cores": cores, "kind": "port",
            "formulation": "per-sample loop (reference semantics), C port, OpenMP over samples",
            "sample": "%d of %d trajectories, forward + adjoint, per-sample loop, %.1f s" % (nsamp, y.shape[0], dt),
            "nfe_forward": int(nf), "nfe_augmented": int(nb)}


def full_training_step_ms(wl, net, y0, t, device, K=10000, reps=5):
    import phoenix_amd

    class Handler:
        def __init__(self, b, tt, y):
            self.b, self.device = (b, tt, y), b.device

        def get_batch(self, bs):
            return self.b

    N = wl["N"]
    g = torch.Generator(device="cpu").manual_seed(7)
    target = (y0 + 0.01 * torch.randn(y0.shape, generator=g).to(device))
    X = (torch.rand(K, 1, N, generator=g) - 0.5).to(device)
    prior_grad = (torch.randn(K, 1, N, generator=g) * 0.05).to(device)
    opt = torch.optim.Adam(net.parameters(), lr=1e-6)
    h = Handler(y0, t, target)
    times = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        phoenix_amd.training_step(net, h, opt, wl["method"], y0.shape[0], False, False, X, prior_grad, 0.99)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    # ... and back to back, as a training loop issues them (deferred status: no host round trip inside a step); the figure
    # above is the latency of ONE synchronised step, host time included
    from phoenix_amd import engine
    prev = engine.status_mode()
    engine.set_status_mode("deferred")
    try:
        n = 4 * reps
        for _ in range(2):
            phoenix_amd.training_step(net, h, opt, wl["method"], y0.shape[0], False, False, X, prior_grad, 0.99)
        engine.check_pending_status(wait=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            phoenix_amd.training_step(net, h, opt, wl["method"], y0.shape[0], False, False, X, prior_grad, 0.99)
        engine.check_pending_status(wait=True)
        torch.cuda.synchronize()
        full_training_step_ms.back_to_back = (time.perf_counter() - t0) / n * 1e3
    finally:
        engine.set_status_mode(prev)
    return float(np.median(times[2:]))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="breast", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="profiling runs: only the timed loop and the per-kernel timing (no training step, no "
                         "strong-scaling shards: their launches would be averaged into the per-kernel profile)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default: weak is the headline; with more than one rank the sharded (strong) problem is "
                         "measured in the same run and reported under extra.strong")
    ap.add_argument("--status", default="deferred", choices=["deferred", "immediate"],
                    help="solver status read-back: behind the backward kernel (checked at a later engine call; the "
                         "library default) or a blocking round trip at the end of every backward()")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="diagnostic, single process only: with --scaling strong, integrate rank 0's shard of a run with "
                         "this many ranks (what ONE GPU of that run would do per step)")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="untimed device pre-warm before the W warm-up steps, as a number of steps fixed per workload "
                         "from this many seconds (rank-uniform); 0 disables it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = PLUMBING TEST ONLY (CPU, tests/test_bench_cpu.py): rank spawning, barriers, the gradient "
                         "all-reduce and the output line with a stub in place of the engine; prints value null")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`--gpus N` without a launcher: start the N ranks as a child torch.distributed.run BEFORE this process touches a
    GPU (device_count() does not initialise one) and hand its exit code on.  The child ranks inherit stdout, rank 0
    prints the one JSON line."""
    import socket
    import subprocess
    if args.backend == "nccl":
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (args.gpus, ndev))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def stub_main(args, rank, world):
    """--backend gloo: everything around the engine (process group, barriers, max-over-ranks timing, grouped gradient
    all-reduce, one line from rank 0) with a CPU stub as the step.  Not a measurement: value is null."""
    import torch.distributed as dist
    from phoenix_amd import parallel
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8)
    x = torch.ones(4, 8) * (rank + 1)

    def step():
        for p in lin.parameters():
            p.grad = None
        lin(x).sum().backward()
        parallel.allreduce_grads(lin)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    te = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    gsum = float(lin.bias.grad.sum())          # 8 outputs x 4 rows x world ranks
    if rank == 0:
        out = {"metric": "ODE-RHS evals/sec (genes x trajectories)", "value": None, "unit": "gene*trajectory RHS evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(te) / args.steps * 1e3,
               "higher_is_better": True, "scaling": args.scaling or "weak", "vs_baseline": None, "dtype": "f32",
               "data": "stub (gloo plumbing test: no engine, no GPU)", "stub": True,
               "config": {"workload": "stub", "parallelism": "trajectory-sharded x%d, grouped gradient all-reduce" % world},
               "extra": {"allreduce_check": gsum == 8.0 * 4 * world}}
        os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def timed_region(step, sync_all, steps, engine, use_dist, device):
    """barrier + synchronize, exactly `steps` steps (every solve's status checked inside), barrier + synchronize;
    returns the MAX over ranks of the elapsed seconds"""
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    engine.check_pending_status(wait=True)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed


def main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # under torchrun (RANK set) the collective path is always taken, also with a single rank, so that a 1-GPU
    # box exercises exactly the code the 8-GPU run uses (RCCL init, barrier, flat gradient all-reduce)
    use_dist = "RANK" in os.environ
    if args.backend == "gloo":
        return stub_main(args, rank, world)
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import phoenix_amd  # noqa: F401  (fails loudly if the HIP library is missing)
    from phoenix_amd import engine, parallel

    engine.set_status_mode(args.status)
    wl = WORKLOADS[args.workload]
    N, H, Bw = wl["N"], wl["H"], wl["B"]
    headline = args.scaling or "weak"

    def problem(scaling):
        """(net, y0, t, G, B_local, B_global) of one scaling mode"""
        if scaling == "strong":
            # the SAME 256-trajectory problem at every world size, sharded over the ranks (contiguous chunks)
            net, y0, t = make_problem(wl, device, seed=0)
            lo, hi = parallel.shard_range(Bw, rank, world if world > 1 or not args.shard_of else args.shard_of)
            y0, t = y0[lo:hi].contiguous(), t[lo:hi].contiguous()
            B_global, B = Bw, hi - lo
        else:
            net, y0, t = make_problem(wl, device, seed=rank)      # every rank its own trajectories
            B_global, B = Bw * world, Bw
        if use_dist:   # replicas share parameters
            import torch.distributed as dist
            for p in net.parameters():
                dist.broadcast(p.data, 0)
        T = t.shape[1]
        gg = torch.Generator(device="cpu").manual_seed(100 + rank)
        # cotangent of a mean-type loss over the GLOBAL batch (reference torch.mean over the whole batch,
        # train_insilico.py:132)
        G = (torch.randn(T, B, 1, N, generator=gg) / (B_global * N)).to(device)
        G[0].zero_()
        return net, y0, t, G, B, B_global

    net, y0, t, G, B, B_global = problem(headline)
    T = t.shape[1]
    wdist = 2 if use_dist else 1

    def sync_all():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        one_step(net, y0, t, G, wl["method"], wdist)

    # Untimed device pre-warm.  A fresh process on an idle GPU starts at low clocks with cold caches, lazily created RCCL
    # communicators and a growing caching allocator, and a 20-step timed region is ~20 ms: the pre-warm runs the same
    # step for `--prewarm-seconds` (a step count fixed per workload => the same on every rank; not counted in `warmup`
    # or `steps`), in windows whose per-step times go into the line (extra.prewarm_windows_ms) so that a ramp is visible.
    nominal_ms = {"breast": 1.0, "insilico": 1.0, "yeast": 10.0, "bcell": 50.0}[args.workload]
    n_pre = int(args.prewarm_seconds * 1e3 / nominal_ms)
    pre_windows = []
    win = max(1, min(100, n_pre // 10 if n_pre >= 10 else 1))
    done = 0
    while done < n_pre:
        k = min(win, n_pre - done)
        pre_windows.append(timed_region(step, sync_all, k, engine, use_dist, device) / k * 1e3)
        done += k
    for _ in range(args.warmup):
        step()
    elapsed = timed_region(step, sync_all, args.steps, engine, use_dist, device)
    # the same region five more times: how far single windows of K steps scatter on this box (p50 / p90)
    windows = [elapsed / args.steps * 1e3]
    for _ in range(5):
        windows.append(timed_region(step, sync_all, args.steps, engine, use_dist, device) / args.steps * 1e3)

    # ---- per-kernel timing with HIP events on the launch stream + NFE accounting (rank-local).
    # The library records one event pair immediately around each solve kernel, on the stream it is launched on
    # (phx_debug_queue_kernel_events: a FIFO, one pair per solve launch, forward and backward alternating), while the
    # SAME steps as in the timed region run back to back -- the launch durations are those of the steady state the
    # headline is measured in (an isolated launch behind a synchronize runs ~4 % longer: cold caches), and they are what
    # `rocprofv3 --kernel-trace --stats` of this command averages.
    from phoenix_amd import _lib
    import ctypes as C
    nrep = max(5, min(args.steps, 20))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nrep)]
    for quad in ev:
        for e in quad:
            e.record()   # instantiates the underlying hipEvent_t
    torch.cuda.synchronize()
    for _ in range(3):
        step()
    for quad in ev:
        _lib.load().phx_debug_queue_kernel_events(C.c_void_p(quad[0].cuda_event), C.c_void_p(quad[1].cuda_event))
        _lib.load().phx_debug_queue_kernel_events(C.c_void_p(quad[2].cuda_event), C.c_void_p(quad[3].cuda_event))
    for _ in range(nrep):
        step()
    engine.check_pending_status(wait=True)
    torch.cuda.synchronize()
    _lib.load().phx_debug_queue_kernel_events(None, None)
    fwd_ms = [q[0].elapsed_time(q[1]) for q in ev]
    adj_ms = [q[2].elapsed_time(q[3]) for q in ev]
    fwd_ms_avg, adj_ms_avg = float(np.mean(fwd_ms)), float(np.mean(adj_ms))
    # NFE accounting: one more solve through the engine entry points (statistics outputs)
    engine.set_status_mode("immediate")
    p = engine.Params(net.net_sums.linear_out.weight, net.net_sums.linear_out.bias, net.net_prods.linear_out.weight,
                      net.net_prods.linear_out.bias, net.net_alpha_combine.linear_out.weight, net.gene_multipliers)
    y2 = y0.reshape(B, N).contiguous()
    t64 = t.double().contiguous()
    Gc = G.reshape(T, B, N).contiguous()
    sol, status, nfe_f, nsteps_f = engine.solve_forward(p, y2, t64, wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9,
                                                        True, True)
    adj, grads, st2, nfe_b, nsteps_b = engine.solve_adjoint(p, t64, sol, Gc, wl["method"], _lib.CTRL_PER_TRAJECTORY,
                                                            1e-7, 1e-9, True, True)
    torch.cuda.synchronize()
    assert int(status.max()) == 0 and int(st2.max()) == 0
    nfe_fwd = int(nfe_f.sum().item())
    nfe_aug = int(nfe_b.sum().item())

    def whole_job(evals_rank):
        if not use_dist:
            return float(evals_rank)
        import torch.distributed as dist
        te = torch.tensor([float(evals_rank)], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        return float(te.item())

    # whole-job value
    total_evals_per_step = whole_job((nfe_fwd + nfe_aug) * N)      # gene x trajectory evaluations, all ranks
    value = total_evals_per_step * args.steps / elapsed

    # ---- the other scaling mode in the same run (more than one rank, no explicit --scaling): config 4 as written
    other = None
    if world > 1 and args.scaling is None and not args.no_extras:
        engine.set_status_mode(args.status)
        net2, y02, t2, G2, B2, _ = problem("strong")

        def step2():
            one_step(net2, y02, t2, G2, wl["method"], wdist)

        for _ in range(max(args.warmup, 20)):
            step2()
        el2 = timed_region(step2, sync_all, args.steps, engine, use_dist, device)
        win2 = [el2 / args.steps * 1e3]
        for _ in range(3):
            win2.append(timed_region(step2, sync_all, args.steps, engine, use_dist, device) / args.steps * 1e3)
        engine.set_status_mode("immediate")
        # the sharded problem's OWN evaluation counts: one more solve of this rank's shard through the engine entry
        # points (statistics outputs), summed over the ranks
        p2 = engine.Params(net2.net_sums.linear_out.weight, net2.net_sums.linear_out.bias, net2.net_prods.linear_out.weight,
                           net2.net_prods.linear_out.bias, net2.net_alpha_combine.linear_out.weight, net2.gene_multipliers)
        t64_2 = t2.double().contiguous()
        sol2, st_f2, nfe_f2, _ = engine.solve_forward(p2, y02.reshape(B2, N).contiguous(), t64_2, wl["method"],
                                                      _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
        _, _, st_b2, nfe_b2, _ = engine.solve_adjoint(p2, t64_2, sol2, G2.reshape(T, B2, N).contiguous(), wl["method"],
                                                      _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
        torch.cuda.synchronize()
        assert int(st_f2.max()) == 0 and int(st_b2.max()) == 0
        evals2 = whole_job((int(nfe_f2.sum().item()) + int(nfe_b2.sum().item())) * N)
        other = {"scaling": "strong", "trajectories_total": Bw, "trajectories_per_gpu": B2,
                 "ms_per_step": el2 / args.steps * 1e3, "value": evals2 * args.steps / el2,
                 "window_ms_per_step": win2,
                 "note": "the ONE %d-trajectory batch of BASELINE.json config 4 sharded over the %d ranks, loss "
                         "normalised by the global batch, grouped gradient all-reduce; evaluations counted from the "
                         "sharded problem's own NFE outputs, summed over the ranks" % (Bw, world)}

    if rank == 0:
        P = 4 * H * N + 2 * H + N
        # ALGORITHMIC bytes: per RHS eval of a batch of B: 4P + 8BN; per augmented eval: 2*4P + 16BN.
        # Units per launch = batch-evaluations = (sum_b nfe_b) / B.
        alg_fwd = (nfe_fwd / B) * (4 * P + 8 * B * N)
        alg_adj = (nfe_aug / B) * (8 * P + 16 * B * N)
        # ALGORITHMIC flops: 8BNH per RHS eval of the batch, 24BNH per augmented eval (forward + VJP
        # w.r.t. y + VJP w.r.t. the parameters).
        flop_fwd = (nfe_fwd / B) * 8.0 * B * N * H
        flop_adj = (nfe_aug / B) * 24.0 * B * N * H
        mid = _lib.METHODS[wl["method"]]
        adj_kernel = {0: "k_solve_adj", 1: "k1_solve_adj", 2: "k1_solve_adj2", 3: "k1_solve_adj3", 4: "k1_solve_adj3c"}[
            _lib.load().phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, mid)]
        fwd_kernel = {0: "k_solve_fwd", 1: "k1_solve_fwd", 3: "k1_solve_fwd3", 4: "k1_solve_fwd3c"}[
            _lib.load().phx_debug_forward_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, mid)]
        dom = adj_kernel if adj_ms_avg >= fwd_ms_avg else fwd_kernel   # key into the PMC summary
        dom_launches = max(1, _lib.load().phx_debug_solve_launches(
            _lib.OP_ADJOINT if dom == adj_kernel else _lib.OP_ODEINT, N, H, B, T, _lib.CTRL_PER_TRAJECTORY, mid))
        alg, flop, ms = (alg_adj, flop_adj, adj_ms_avg) if dom == adj_kernel else (alg_fwd, flop_fwd, fwd_ms_avg)
        achieved = alg / (ms * 1e-3) / 1e9
        # Which roof binds: arithmetic intensity of the algorithmic figures against the fp32 ridge of the part
        # (157.3 TFLOP/s f32-input MFMA / 8 TB/s HBM = 19.7 flop/B, MI355X_MICROARCH.md); SURVEY 8d: C4 at B = 256
        # sits above it, the small-batch reference shapes below.
        PEAK_HBM, PEAK_MFMA_F32 = 8000.0, 157.3
        tflops = flop / (ms * 1e-3) / 1e12
        mfma_bound = (flop / alg) > (PEAK_MFMA_F32 * 1e12) / (PEAK_HBM * 1e9)
        # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # --pmc WRITE_SIZE, separate runs; FETCH_SIZE doubled per MI355X_MICROARCH.md for 16-B/lane streams)
        # (the counters cannot be read while the bench runs un-profiled: the figure is the committed summary of the
        # profiled run of this same command, regenerated by tools/collect_profiles.sh)
        # Provenance: the summary names the library it was collected on (sha256 of libphoenix_hip.so, written by
        # tools/summarize_profiles.py); a summary of ANOTHER build is not this run's traffic -> `traffic: null` + a note.
        traffic, pmc_file, pmc_note = None, None, None
        import hashlib
        lib_sha = hashlib.sha256(open(_lib.lib_path(), "rb").read()).hexdigest()
        for tag in ("r5", "r4", "r3", "r2"):
            cand = os.path.join(ROOT, "profiles", "%s_%s_pmc_hbm.json" % (tag, args.workload))
            if os.path.exists(cand):
                try:
                    pmc = json.load(open(cand))
                    if pmc.get("library_sha256") != lib_sha:
                        pmc_note = ("%s was collected on another build of libphoenix_hip.so (sha256 %s, running %s): traffic "
                                    "not reported" % (os.path.basename(cand), str(pmc.get("library_sha256"))[:12], lib_sha[:12]))
                        break
                    # (per-launch counters x the launches one solve of this batch makes: 256 B-cell trajectories = two)
                    traffic = (2.0 * pmc["FETCH_SIZE_KB_per_launch"][dom] + pmc["WRITE_SIZE_KB_per_launch"][dom]) * 1024.0 * dom_launches
                    pmc_file = os.path.relpath(cand, ROOT)
                    break
                except Exception as exc:   # noqa: BLE001
                    traffic, pmc_note = None, "%s: %r" % (os.path.basename(cand), exc)
                    break
        roofline = {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom,
                    "achieved": tflops if mfma_bound else achieved,
                    "peak": PEAK_MFMA_F32 if mfma_bound else PEAK_HBM,
                    "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": tflops / PEAK_MFMA_F32 if mfma_bound else achieved / PEAK_HBM,
                    "arithmetic_intensity_flop_per_byte": flop / alg,
                    "hbm": {"achieved_GBps": achieved, "peak_GBps": PEAK_HBM, "frac": achieved / PEAK_HBM},
                    "mfma_f32": {"achieved_TFLOPs": tflops, "peak_TFLOPs": PEAK_MFMA_F32,
                                 "frac": tflops / PEAK_MFMA_F32, "algorithmic_flop_per_launch": flop},
                    "traffic": traffic,
                    "measured_hbm_GBps": (traffic / (ms * 1e-3) / 1e9) if traffic else None,
                    "traffic_source": ("%s (2*FETCH_SIZE + WRITE_SIZE of a separate profiled run of this command on this "
                                       "same library build, sha256 %s; not measured in this run)"
                                       % (pmc_file, lib_sha[:12])) if traffic else pmc_note,
                    "library_sha256": lib_sha,
                    "algorithmic_bytes_per_launch": alg, "launch_ms": ms, "launches_per_solve": dom_launches,
                    "forward": {"kernel": fwd_kernel, "launch_ms": fwd_ms_avg, "GBps": alg_fwd / (fwd_ms_avg * 1e-3) / 1e9,
                                "TFLOPs": flop_fwd / (fwd_ms_avg * 1e-3) / 1e12, "batch_evals": nfe_fwd / B},
                    "adjoint": {"kernel": adj_kernel, "launch_ms": adj_ms_avg, "GBps": alg_adj / (adj_ms_avg * 1e-3) / 1e9,
                                "TFLOPs": flop_adj / (adj_ms_avg * 1e-3) / 1e12, "batch_evals": nfe_aug / B}}
        ws = sorted(windows)
        out = {
            "metric": "ODE-RHS evals/sec (genes x trajectories)", "value": value,
            "unit": "gene*trajectory RHS evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": headline,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "step_definition": "odeint_adjoint forward solve + torch.autograd.backward(sol, dL/dsol) (the backward solve is "
                               "entered with the loss cotangent; no loss-head kernels) + parameter re-layout + gradient "
                               "reduction (+ all-reduce); since round 4 (rounds 1-3 timed (sol * G).sum().backward())",
            "config": {"workload": wl["desc"], "genes": N, "hidden": H, "trajectories_per_gpu": B,
                       "method": wl["method"], "rtol": 1e-7, "atol": 1e-9, "time_points": wl["t"],
                       "parallelism": "trajectory-sharded x%d, grouped gradient all-reduce" % world,
                       "nfe_forward_per_step": nfe_fwd, "nfe_augmented_per_step": nfe_aug,
                       "status_readback": args.status},
            "roofline": roofline,
            # forward and augmented evaluations are different units of work: their own rates, from the
            # launch durations of the two solve kernels on this rank
            "rates": {"forward_evals_per_s": nfe_fwd * N / (fwd_ms_avg * 1e-3),
                      "augmented_evals_per_s": nfe_aug * N / (adj_ms_avg * 1e-3),
                      "kernel_ms_per_step": fwd_ms_avg + adj_ms_avg,
                      "step_over_kernels": (elapsed / args.steps * 1e3) / (fwd_ms_avg + adj_ms_avg)},
            "extra": {"step_windows_ms": {"windows": windows, "p50": float(np.percentile(windows, 50)),
                                          "p90": float(np.percentile(windows, 90)), "min": ws[0], "max": ws[-1],
                                          "note": "ms per step of six consecutive timed regions of `steps` steps each; the "
                                                  "first is the one `value` and `ms_per_step` report"},
                      "prewarm_steps": n_pre, "prewarm_windows_ms": [round(x, 4) for x in pre_windows]},
        }
        if other is not None:
            out["extra"]["strong"] = other
        if world == 1 and not args.no_extras:
            # informational: the reference's full training_step (train_insilico.py:124-140) with its K = 10 000-row
            # prior branch and an Adam step; NOT part of `value` (the metric counts ODE RHS evaluations only)
            try:
                K = 10000
                ts_ms = full_training_step_ms(wl, net, y0, t, device, K=K)
                out["training_step_ms"] = ts_ms
                out["training_step_back_to_back_ms"] = getattr(full_training_step_ms, "back_to_back", None)
                # algorithmic flops of the whole step: the two solves + the prior branch (forward 8KNH, backward with
                # parameter gradients 16KNH); bytes: the solves' + 4P + 8KN (forward) + 8P + 12KN (backward)
                ts_flop = flop_fwd + flop_adj + 24.0 * K * N * H
                ts_bytes = alg_fwd + alg_adj + (12 * P + 20.0 * K * N)
                out["training_step"] = {
                    "ms": ts_ms, "prior_rows": K,
                    "roofline": {"bound": "mfma", "achieved": ts_flop / (ts_ms * 1e-3) / 1e12, "peak": PEAK_MFMA_F32,
                                 "unit": "TFLOP/s", "frac": ts_flop / (ts_ms * 1e-3) / 1e12 / PEAK_MFMA_F32,
                                 "algorithmic_flop": ts_flop, "algorithmic_bytes": ts_bytes,
                                 "hbm_view_frac": ts_bytes / (ts_ms * 1e-3) / 1e9 / PEAK_HBM, "traffic": None}}
            except Exception as exc:   # noqa: BLE001  (diagnostic extra must never break the bench line)
                out["training_step_ms"] = None
                out["training_step_error"] = repr(exc)[:200]
            # BASELINE.json config 4 is a STRONG-scaling problem (one 256-trajectory batch sharded 1/2/4/8): what one GPU
            # of such a run does per step (its shard of the batch, no all-reduce), measured here on this single device
            try:
                if headline == "weak" and not use_dist:
                    strong = {}
                    engine.set_status_mode(args.status)
                    try:
                        for W in (1, 2, 4, 8):
                            lo, hi = parallel.shard_range(wl["B"], 0, W)
                            ys, ts_, Gs = y0[lo:hi].contiguous(), t[lo:hi].contiguous(), G[:, lo:hi].contiguous()
                            for _ in range(10):
                                one_step(net, ys, ts_, Gs, wl["method"], 1)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(20):
                                one_step(net, ys, ts_, Gs, wl["method"], 1)
                            engine.check_pending_status(wait=True)
                            torch.cuda.synchronize()
                            strong[str(W)] = (time.perf_counter() - t0) / 20 * 1e3
                    finally:
                        engine.set_status_mode("immediate")
                    # the same step with a blocking status read at the end of every backward()
                    for _ in range(3):
                        one_step(net, y0, t, G, wl["method"], 1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        one_step(net, y0, t, G, wl["method"], 1)
                    torch.cuda.synchronize()
                    out["extra"].update({
                        "strong_scaling_ms": strong,
                        "strong_scaling_note": "per-rank step time of a W-rank run of the SHARDED %d-trajectory "
                                               "batch (rank 0's shard, this one device, no all-reduce)" % wl["B"],
                        "ms_per_step_status_immediate": (time.perf_counter() - t0) / 20 * 1e3})
            except Exception as exc:   # noqa: BLE001
                out["extra"]["error"] = repr(exc)[:200]
            # The reference's OWN batch size (config_breast.cfg:6 `batch_size = 17`, config_yeast.cfg:6 4, config_BCell.cfg:6 2;
            # loop train_insilico.py:128-130): the same step on that many trajectories, per-sample control.  HBM / latency
            # bound there (AI = 8BNH / (4P + 8BN) is ~7 flop/B forward at B = 17, below the 19.7 ridge).
            try:
                Bref = {"breast": 17, "yeast": 4, "bcell": 2, "insilico": 4}[args.workload]
                if not use_dist and Bref < B:
                    engine.set_status_mode(args.status)
                    try:
                        ys, ts_, Gs = y0[:Bref].contiguous(), t[:Bref].contiguous(), (G[:, :Bref] * (B / Bref)).contiguous()
                        for _ in range(20):
                            one_step(net, ys, ts_, Gs, wl["method"], 1)
                        engine.check_pending_status(wait=True)
                        torch.cuda.synchronize()
                        reps = 50
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            one_step(net, ys, ts_, Gs, wl["method"], 1)
                        engine.check_pending_status(wait=True)
                        torch.cuda.synchronize()
                        ms_ref = (time.perf_counter() - t0) / reps * 1e3
                    finally:
                        engine.set_status_mode("immediate")
                    # the same step recorded into ONE HIP graph (phoenix_amd.GraphedStep): what it costs once the host's
                    # Python / launch work is out of the way (at this size the eager step is host bound on slower hosts)
                    ms_graph = None
                    try:
                        gstep = phoenix_amd.GraphedStep(lambda: one_step(net, ys, ts_, Gs, wl["method"], 1))
                        for _ in range(10):
                            gstep()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            gstep()
                        torch.cuda.synchronize()
                        ms_graph = (time.perf_counter() - t0) / reps * 1e3
                        gstep.check_status()
                    except Exception as exc:   # noqa: BLE001
                        ms_graph = repr(exc)[:160]
                    solr, str_, nfr, _ = engine.solve_forward(p, ys.reshape(Bref, N).contiguous(), ts_.double().contiguous(),
                                                              wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
                    _, _, stb_, nbr, _ = engine.solve_adjoint(p, ts_.double().contiguous(), solr,
                                                              Gs.reshape(T, Bref, N).contiguous(), wl["method"],
                                                              _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
                    torch.cuda.synchronize()
                    nf_r, nb_r = int(nfr.sum().item()), int(nbr.sum().item())
                    bytes_r = (nf_r / Bref) * (4 * P + 8 * Bref * N) + (nb_r / Bref) * (8 * P + 16 * Bref * N)
                    flop_r = (nf_r / Bref) * 8.0 * Bref * N * H + (nb_r / Bref) * 24.0 * Bref * N * H
                    out["extra"]["reference_batch"] = {
                        "trajectories": Bref, "control": "per-sample", "ms_per_step": ms_ref, "graph_ms_per_step": ms_graph,
                        "value": (nf_r + nb_r) * N / (ms_ref * 1e-3), "unit": "gene*trajectory RHS evals/s",
                        "nfe_forward": nf_r, "nfe_augmented": nb_r,
                        "roofline": {"bound": "hbm", "achieved": bytes_r / (ms_ref * 1e-3) / 1e9, "peak": PEAK_HBM,
                                     "unit": "GB/s", "frac": bytes_r / (ms_ref * 1e-3) / 1e9 / PEAK_HBM,
                                     "arithmetic_intensity_flop_per_byte": flop_r / bytes_r,
                                     "algorithmic_bytes_per_step": bytes_r, "traffic": None},
                        "note": "the reference's own batch_size for this data set (its config file), whole step wall time "
                                "(two solve launches + layout + gradient reduction), deferred status"}
            except Exception as exc:   # noqa: BLE001
                out["extra"]["reference_batch_error"] = repr(exc)[:200]
        if world == 1 and not args.no_cpu_baseline:
            # both shapes BASELINE.md section 3 names; speed-ups are quoted against the stronger one
            loop = cpu_baseline(wl, net, y0, t, G)
            try:
                batched = cpu_baseline_batched(wl, net, y0, t, G)
            except Exception as exc:   # noqa: BLE001
                batched = {"value": 0.0, "error": repr(exc)[:200]}
            cb = dict(batched if batched["value"] >= loop["value"] else loop)
            cb["per_sample_loop"] = loop
            cb["batched"] = batched
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
        os.write(_REAL_STDOUT, (json.dumps(out) + "\n").encode())
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


_REAL_STDOUT = 1   # importing callers write the result line to fd 1; the script entry below redirects native chatter


if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(_args))         # nothing has touched a GPU yet
    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a version banner at the first collective)
    # and anything else that writes to fd 1 go to stderr; the result line is written to the real stdout at the end
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main(_args)
