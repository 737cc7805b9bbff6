"""HIP-graph capture of a whole training step.

Small graphs (Cora: |V| = 2708) make every kernel on the path a few microseconds long, so an eager
epoch is pure launch latency (~60 launches, ~0.8 ms).  ``CapturedTrainStep`` records one
forward + loss + backward + optimizer step into a HIP graph once and replays it: the same kernels
in the same order with the same numerics, issued from device-side descriptors.

Requirements (checked where possible): every tensor the step reads is static (same storage each
replay -- copy new data INTO the inputs), the optimizer is constructed with ``capturable=True``,
and nothing inside synchronises (the C ABI never does; see include/stgraph_hip.h).
"""
from __future__ import annotations

import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", **kw):
    """``torch.cuda.graph(graph)`` with Python's cyclic garbage collector kept out of the capture.  A dead reference cycle that
    holds another ``CUDAGraph`` (an object that stored its own bound method, a closure that names its owner ...) is freed
    whenever the collector happens to run -- and destroying a HIP graph while a stream of the process is capturing is an error
    the runtime reports from a destructor: the process aborts (torch 2.10 no longer collects before a capture).  So: collect
    once before, no collection during."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()



class CapturedTrainStep:
    def __init__(self, step_fn, optimizer, params, warmup: int = 3):
        """``step_fn()`` must run forward, compute the loss, call ``optimizer.zero_grad(set_to_none=False)``
        (or otherwise zero the grads in place), ``loss.backward()`` and ``optimizer.step()``, and return
        the loss tensor."""
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("CapturedTrainStep needs an optimizer constructed with capturable=True")
        params = [p for p in params if p.requires_grad]
        dev = params[0].device
        # capturing must not advance the training state: snapshot, warm up + capture, restore in place
        saved = [p.detach().clone() for p in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph):
            self.loss = step_fn()
        with torch.no_grad():
            for p, s in zip(params, saved):
                p.copy_(s)
            for st in optimizer.state.values():          # fresh optimizer state, in place
                for v in st.values():
                    if isinstance(v, torch.Tensor):
                        v.zero_()

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss
