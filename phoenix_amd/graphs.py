"""A whole training step as ONE HIP graph.

The host side of a step -- Python, autograd, six to eight kernel launches -- costs 0.35-0.5 ms depending on the host, which
is what a small-batch step takes when its kernels need less (the reference's own batch of 17 trajectories at the
breast-cancer shape: 0.32 ms of kernels, train_insilico.py:128-130).  Everything the engine issues for a step is
stream-ordered and allocation-stable (plain launches behind the occupancy check, cached workspaces, the parameter layout
kernel), so `torch.cuda.graph` can record it; the only host round trip of a step, the solver status, stays on the device
("captured" status mode) and is read when the caller asks.

    step = GraphedStep(lambda: train_one_batch())     # forward solve, loss, backward[, optimizer.step()]
    for it in range(n):
        fill_static_inputs(...)                          # copy_ into the tensors the step reads
        step()                                           # one graph launch
        if it % 50 == 0:
            step.check_status()                          # the reference's AssertionErrors, a few steps late

The step function must be replay-safe in PyTorch's sense (static input tensors, no host reads, no data-dependent Python
control flow).  Gradient all-reduces (phoenix_amd.parallel) are not recorded: one process, one device.
"""
import torch

from . import engine


class GraphedStep:
    def __init__(self, fn, warmup=3):
        """runs `fn` `warmup` times on a private stream (plans, workspaces, allocations), then records one call"""
        self.fn = fn
        self.stream = torch.cuda.Stream()
        self.graph = torch.cuda.CUDAGraph()
        prev = engine.status_mode()
        engine.set_status_mode("captured")
        try:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for _ in range(max(1, warmup)):
                    del engine.captured_status[:]
                    fn()
            torch.cuda.current_stream().wait_stream(self.stream)
            torch.cuda.synchronize()
            engine.check_captured_status()               # a step that fails eagerly is not worth recording
            del engine.captured_status[:]
            # the warm-up ran on the stream the capture uses: its workspaces are the step's own, and the recorded solves
            # vouch for them (no exchange-buffer fills in the graph) -- nothing else ever runs on this stream
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.result = fn()
            self.status_blocks = list(engine.captured_status)
            del engine.captured_status[:]
        finally:
            engine.set_status_mode(prev)
        engine.forget_params()

    def __call__(self):
        self.graph.replay()
        # parameters the graph updates change behind PyTorch's version counters: eager calls lay them out again
        engine.forget_params()
        return self.result

    def check_status(self):
        """the solver status of the LAST replay (one device->host read per solve pair): raises like the eager step"""
        engine.check_captured_status(self.status_blocks)
