"""Deferred weight gradients: one launch per parameter per backward pass.

A recurrent model applies the same weights at every timestep, so reverse-mode autodiff produces one
weight-gradient contribution per timestep and adds them up one by one: for a 25-step TGCN window that
is 25 small tall-skinny GEMMs and 24 accumulation kernels PER PARAMETER.  The contributions all have
the form ``A_t^T B_t``; their sum over t is a single tall-skinny GEMM over the concatenated rows.

The custom autograd nodes of this package therefore do not return weight gradients.  During backward
they register ``(param, A_t, B_t)`` with the active :class:`WeightGradAccumulator`; an autograd-engine
callback queued on the first registration fires once the backward pass has finished and computes
every parameter's ``sum_t A_t^T B_t`` (and bias ``sum_t colsum(A_t)``) with ONE split-K MFMA launch
over all segments (``kernels.gemm_tn_multi``), then adds the result into ``param.grad``.

After ``loss.backward()`` returns, ``param.grad`` holds exactly what autograd would have produced
(summation order differs: fp32 rounding).  Not supported while deferral is on: ``torch.autograd.grad``
for these parameters, parameter-level backward hooks, double backward.
``stgraph_amd.nn.functional.set_deferred_weight_grads(False)`` restores plain autograd behaviour.
"""
from __future__ import annotations

import threading

import torch
from torch.autograd import Variable

from .. import kernels

# One backward pass at a time per process.  The autograd engine may run custom nodes on a device worker
# thread and the end-of-pass callback on the thread that called backward(), so this is a plain global
# guarded by a lock, not a thread-local.  The accumulator is tied to the engine's graph task: a backward()
# that raised never runs its queued callbacks, so whatever it registered is dropped -- not flushed into, or
# withheld from, the next pass -- as soon as a node of another task asks for the accumulator.
_lock = threading.Lock()
_active = None
_active_task = None


def _graph_task_id() -> int:
    return int(torch._C._current_graph_task_id())


class WeightGradAccumulator:
    def __init__(self):
        self.entries = {}          # key -> dict(sink, As, Bs, colsum_sink)
        self.order = []

    def add(self, key, A, B, sink, colsum_sink=None):
        """Register one contribution ``A^T B`` for ``sink(dW)`` (and ``colsum(A)`` for ``colsum_sink``)."""
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = {"As": [], "Bs": [], "sink": sink, "colsum_sink": colsum_sink}
            self.order.append(key)
        e["As"].append(A)
        e["Bs"].append(B)

    def flush(self):
        try:
            for key in self.order:
                e = self.entries[key]
                groups = {}
                for a, b in zip(e["As"], e["Bs"]):           # segments of one launch share one shape
                    groups.setdefault((a.shape, b.shape), ([], []))
                    groups[(a.shape, b.shape)][0].append(a)
                    groups[(a.shape, b.shape)][1].append(b)
                dW = db = None
                for As, Bs in groups.values():
                    if e["colsum_sink"] is not None:
                        c, cs = kernels.gemm_tn_multi(As, Bs, colsum=True)
                        db = cs if db is None else db + cs
                    else:
                        c = kernels.gemm_tn_multi(As, Bs)
                    dW = c if dW is None else dW + c
                e["sink"](dW)
                if e["colsum_sink"] is not None:
                    e["colsum_sink"](db)
        finally:
            self.entries, self.order = {}, []


def add_to_grad(param: torch.Tensor, value: torch.Tensor) -> None:
    """``param.grad += value`` (creating it if needed), in place so that gradient buckets stay views."""
    value = value.view_as(param) if value.shape != param.shape else value
    if param.grad is None:
        param.grad = value.detach().clone()
    else:
        param.grad.add_(value)


def current() -> WeightGradAccumulator:
    """The accumulator of the backward pass that is running on this thread (created on demand; its
    flush is queued to run when the autograd engine finishes the pass)."""
    global _active, _active_task
    task = _graph_task_id()
    with _lock:
        acc = _active
        if acc is None or _active_task != task:          # None, or left behind by a backward() that raised
            acc = _active = WeightGradAccumulator()
            _active_task = task

            def _finish(acc=acc):
                global _active, _active_task
                with _lock:
                    if _active is acc:
                        _active, _active_task = None, None
                acc.flush()

            Variable._execution_engine.queue_callback(_finish)
    return acc


def reset() -> None:
    """Drop whatever an interrupted backward pass left registered (``current()`` does the same on its own when the
    next pass starts; this frees the tensors right away)."""
    global _active, _active_task
    with _lock:
        _active, _active_task = None, None
