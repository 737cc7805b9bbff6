"""Autograd boundary around a fused unit (reference compiler/backend/kernel_wrapper.py:1-19)."""
from __future__ import annotations


class KernelWrapper:
    @staticmethod
    def forward(executor, kid, kernel_args, rets, *args):
        outs, serial = executor.forward_cb(kid, kernel_args, rets, args)
        executor.ts.tensor_map_stack.top()["__outs__"] = [o.detach() for o in outs]
        KernelWrapper._last_serial = serial
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def setup_context(ctx, inputs, output):
        executor, kid = inputs[0], inputs[1]
        ctx.backward_cache = executor, kid, executor.ts.tensor_map_stack.top()["__serial__"]
        ctx.set_materialize_grads(False)

    @staticmethod
    def backward(ctx, *gradout):
        executor, kid, serial = ctx.backward_cache
        return (None, None, None, None) + executor.backward_cb(kid, gradout, serial)
