"""Run-time half of the ``@compile`` API: launches the plan's kernels and keeps
the reference's execution state (compiler/executor.py:29-96, :236-259, :328-426).

State kept across calls, exactly as the reference's ``ExeState``:
  * ``tensor_map_stack``    -- one entry per forward call: the tensors its backward
                               units will read (pushed in ``forward_cb``, popped in
                               ``backward_cb``; LIFO order == BPTT order);
  * ``graph_timestamp_stack`` -- for ``DynamicGraph``: the snapshot each forward call
                               saw; ``backward_cb`` asks the graph to walk back to it
                               (``get_backward_graph``) before launching.
"""
from __future__ import annotations

from collections import deque

import torch

from ..graph.dynamic.dynamic_graph import DynamicGraph


class Stack:
    def __init__(self):
        self.content = deque()

    def push(self, val):
        self.content.append(val)

    def pop(self):
        return self.content.pop()

    def top(self):
        return self.content[-1]

    def __len__(self):
        return len(self.content)


class ExeState:
    def __init__(self):
        self.tensor_map_stack = Stack()
        self.graph_timestamp_stack = Stack()
        self.current_tensor_map = {}


class Executor:
    def __init__(self, graph, plan, signature):
        self.graph = graph
        self.plan = plan
        self.signature = signature
        self.ts = ExeState()
        self.new_zeros = None
        self.raw_ptr = None
        self.num_nodes = graph.get_num_nodes()
        self.num_edges = graph.get_num_edges()
        self._inputs = plan.input_names()
        self._diff = plan.differentiable()
        self._serial = 0

    # -- reference API (executor.py:236-265) ----------------------------------------------------
    def restart(self, input_map, graph=None):
        self.ts.current_tensor_map = dict(input_map)
        if graph is not None:
            self.graph = graph
            self.num_nodes = graph.get_num_nodes()
            self.num_edges = graph.get_num_edges()

    def set_raw_ptr_cb(self, cb):
        self.raw_ptr = cb

    def set_new_zeros_cb(self, cb):
        self.new_zeros = cb

    def execute(self, FuncWrapper):
        """Forward pass: one autograd node around the fused forward unit(s)."""
        tensors = [self.ts.current_tensor_map[k] for k in self._inputs]
        # A call that autograd will never walk back through (torch.no_grad() evaluation, no differentiable input)
        # keeps nothing: its entry would otherwise stay on the stack for good (only backward_cb pops).
        self._track = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)
        rets = FuncWrapper.apply(self, 0, self._inputs, None, *tensors)
        return rets if isinstance(rets, tuple) else (rets,)

    # -- called by KernelWrapper ------------------------------------------------------------------
    def forward_cb(self, uid, kernel_args, rets, tensor_list):
        n_feats = {name: t for (kind, name), t in zip(kernel_args, tensor_list) if kind == "n"}
        e_feats = {name: t for (kind, name), t in zip(kernel_args, tensor_list) if kind == "e"}
        outs, saved = self.plan.forward(self.graph, n_feats, e_feats)
        self.ts.current_tensor_map = {}
        if not getattr(self, "_track", True):
            return outs, None
        self._serial += 1
        saved["__serial__"] = self._serial
        self.ts.tensor_map_stack.push(saved)
        if isinstance(self.graph, DynamicGraph):
            self.ts.graph_timestamp_stack.push(self.graph.current_timestamp)
        return outs, self._serial

    def backward_cb(self, kid, grad_list, serial=None):
        # Normally the entry is the top of the stack (autograd replays calls in reverse, which is
        # what the reference relies on, executor.py:383); look it up by serial so that an unusual
        # replay order can never pair a backward launch with another call's tensors.
        stack = self.ts.tensor_map_stack.content
        pos = len(stack) - 1
        if serial is not None:
            while pos >= 0 and stack[pos].get("__serial__") != serial:
                pos -= 1
            if pos < 0:
                raise RuntimeError("backward called twice for the same compiled vertex-function call "
                                   "(its saved tensors were already released)")
        saved = stack[pos]
        if isinstance(self.graph, DynamicGraph):
            self.graph.get_backward_graph(self.ts.graph_timestamp_stack.content[pos])
        grads = []
        for g, ref in zip(grad_list, saved.get("__outs__", [None] * len(grad_list))):
            if g is None:                       # set_materialize_grads(False) can hand over None
                g = torch.zeros_like(ref)
            grads.append(g.contiguous())
        result = self.plan.backward(self.graph, saved, grads)
        del stack[pos]
        if isinstance(self.graph, DynamicGraph):
            del self.ts.graph_timestamp_stack.content[pos]
        return tuple(result.get(k) for k in self._inputs)
