"""Autograd boundary around a fused unit (reference compiler/backend/kernel_wrapper.py:1-19)."""
from __future__ import annotations


class KernelWrapper:
    @staticmethod
    def forward(executor, kid, kernel_args, rets, *args):
        outs, serial = executor.forward_cb(kid, kernel_args, rets, args)
        if serial is not None:                  # None: nothing was kept (no backward can follow this call)
            executor.ts.tensor_map_stack.top()["__outs__"] = [o.detach() for o in outs]
        executor._last_serial = serial
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def setup_context(ctx, inputs, output):
        executor, kid = inputs[0], inputs[1]
        ctx.backward_cache = executor, kid, executor._last_serial
        ctx.set_materialize_grads(False)

    @staticmethod
    def backward(ctx, *gradout):
        executor, kid, serial = ctx.backward_cache
        if serial is None:
            raise RuntimeError("backward through a compiled vertex-function call that kept no state "
                               "(it ran under torch.no_grad() or without a differentiable input)")
        return (None, None, None, None) + executor.backward_cb(kid, gradout, serial)
