"""world_size-2 gloo tests of the data-parallel path (trajectory sharding + flat gradient all-reduce)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phoenix_amd import parallel
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 3)
    X = torch.arange(10 * 7, dtype=torch.float32).reshape(10, 7) / 10.0
    lo, hi = parallel.shard_range(10, rank, world)
    # loss normalised by the GLOBAL batch (reference torch.mean over the whole batch, train_insilico.py:132)
    loss = (lin(X[lo:hi]) ** 2).sum() / (10 * 3)
    loss.backward()
    parallel.allreduce_grads(lin)
    (tot,) = parallel.allreduce_scalars(loss)
    # plain lists: tensors in an mp.Queue travel as shared-memory fds that die with the worker
    q.put((rank, lin.weight.grad.tolist(), lin.bias.grad.tolist(), float(tot), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 3)
    X = torch.arange(10 * 7, dtype=torch.float32).reshape(10, 7) / 10.0
    loss = (lin(X) ** 2).mean()
    loss.backward()
    ranges = sorted(r[4] for r in res)
    assert ranges == [(0, 5), (5, 10)]
    for _, gw, gb, tot, _ in res:
        assert torch.allclose(torch.tensor(gw), lin.weight.grad, rtol=1e-5, atol=1e-6)
        assert torch.allclose(torch.tensor(gb), lin.bias.grad, rtol=1e-5, atol=1e-6)
        assert abs(tot - float(loss)) < 1e-5 * abs(float(loss))


def test_shard_range_covers_everything():
    from phoenix_amd.parallel import shard_range
    for n in (1, 7, 256, 1023):
        for w in (1, 2, 3, 8):
            pieces = [shard_range(n, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            for a, b in zip(pieces, pieces[1:]):
                assert a[1] == b[0]
