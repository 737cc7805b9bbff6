"""Symbolic values recorded while a vertex function runs once.

Same protocol as the reference's ``TorchVal`` (compiler/val/pytorch/torch_val.py):
``*``, ``+``, ``-``, ``/`` between values or value and scalar; Python's builtin
``sum([...])`` starts with ``0 + val`` and therefore lands in ``__radd__``, which
records the neighbour aggregation ``AggSum`` and yields a DEST value
(torch_val.py:117-127).  Where the reference monkey-patches every function of the
``torch`` namespace and the module's ``_parameters/_buffers/_modules`` for the
duration of the trace (compiler/stgraph.py:126-173), this tracer relies on
``__torch_function__``: ``torch.exp(val)`` or ``self.leaky_relu(val)`` dispatch to
the symbolic value without touching global state, so tracing is re-entrant.
"""
from __future__ import annotations

import torch

from .gir import Node, Program, ValType, infer_val_type

_ELEMENTWISE = {
    "exp": "Exp",
    "relu": "Relu",
    "leaky_relu": "LeakyRelu",
}


def _broadcast_shape(a: tuple, b: tuple) -> tuple:
    try:
        return tuple(torch.broadcast_shapes(a, b))
    except RuntimeError as e:       # same failure mode as executing the op on the traced tensors
        raise ValueError(f"feature shapes {a} and {b} do not broadcast") from e


class Val:
    """A traced per-vertex / per-edge value."""

    __array_priority__ = 1000

    def __init__(self, node: Node, prog: Program):
        self.node = node
        self.prog = prog

    # -- construction -----------------------------------------------------------------------------
    @classmethod
    def leaf(cls, prog: Program, name: str, val_type: ValType, tensor: torch.Tensor, reduce_dim: bool = True):
        shape = tuple(tensor.shape[1:]) if reduce_dim else tuple(tensor.shape)
        node = Node("Leaf", val_type, shape, name=name, requires_grad=bool(tensor.requires_grad),
                    value=None if reduce_dim else tensor)
        return cls(prog.intern(node), prog)

    def _lift(self, other) -> Node:
        if isinstance(other, Val):
            return other.node
        if isinstance(other, (int, float, bool)):
            return self.prog.intern(Node("Const", None, (), value=other))
        if isinstance(other, torch.Tensor):      # a parameter / buffer of the gnn module used in the function
            name = f"param{id(other):x}"
            return self.prog.intern(Node("Leaf", ValType.PARAM, tuple(other.shape), name=name, value=other,
                                         requires_grad=bool(other.requires_grad)))
        raise TypeError(f"unsupported operand in a vertex function: {type(other).__name__}")

    def _binary(self, op: str, lhs: Node, rhs: Node) -> "Val":
        shape = _broadcast_shape(lhs.shape, rhs.shape)
        node = Node(op, infer_val_type((lhs, rhs)), shape, args=(lhs, rhs),
                    requires_grad=lhs.requires_grad or rhs.requires_grad)
        return Val(self.prog.intern(node), self.prog)

    def _unary(self, op: str, params: tuple = ()) -> "Val":
        node = Node(op, self.node.val_type, self.node.shape, args=(self.node,), params=params,
                    requires_grad=self.node.requires_grad)
        return Val(self.prog.intern(node), self.prog)

    # -- properties the reference's Val exposes ---------------------------------------------------
    @property
    def val_type(self) -> ValType:
        return self.node.val_type

    @property
    def size(self) -> list:
        return list(self.node.shape)

    @property
    def var(self) -> Node:
        return self.node

    # -- arithmetic -------------------------------------------------------------------------------
    def __mul__(self, other):
        return self._binary("Mul", self.node, self._lift(other))

    def __rmul__(self, other):
        return self._binary("Mul", self.node, self._lift(other))      # torch_val.py:96-97: delegates to __mul__

    def __add__(self, other):
        return self._binary("Add", self.node, self._lift(other))

    def __radd__(self, other):
        # builtin sum([...]) == 0 + val : neighbour aggregation  (torch_val.py:117-127)
        if not (isinstance(other, int) and not isinstance(other, bool)):
            raise TypeError("only Python's builtin sum() over neighbours may appear on the left of '+'")
        if self.val_type not in (ValType.SRC, ValType.EDGE):
            raise TypeError("sum([...]) aggregates per-neighbour (SRC) or per-edge (EDGE) values")
        node = Node("AggSum", ValType.DEST, self.node.shape, args=(self.node,),
                    requires_grad=self.node.requires_grad)
        return Val(self.prog.intern(node), self.prog)

    def __sub__(self, other):
        return self._binary("Sub", self.node, self._lift(other))

    def __truediv__(self, other):
        return self._binary("TrueDiv", self.node, self._lift(other))

    def __floordiv__(self, other):
        raise NotImplementedError("__floordiv__ Op not supported")

    def sum(self, *args, **kwargs):
        raise NotImplementedError("Tensor.sum inside a vertex function is not supported by the MI355X kernels yet")

    def view(self, *args, **kwargs):
        raise NotImplementedError("Tensor.view inside a vertex function is not supported by the MI355X kernels yet")

    # -- torch.<fn>(val) and module(val) ----------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        vals = [a for a in args if isinstance(a, Val)]
        if not vals:
            return NotImplemented
        self = vals[0]
        if name in ("mul", "__mul__", "__rmul__", "multiply"):
            a, b = args[0], args[1]
            return (a if isinstance(a, Val) else b).__mul__(b if isinstance(a, Val) else a)
        if name in ("add", "__add__", "__radd__") and isinstance(args[0], Val) and isinstance(args[1], Val):
            return args[0].__add__(args[1])
        if name in ("sub", "__sub__") and isinstance(args[0], Val):
            return args[0].__sub__(args[1])
        if name in ("div", "true_divide", "__truediv__") and isinstance(args[0], Val):
            return args[0].__truediv__(args[1])
        if name in _ELEMENTWISE and isinstance(args[0], Val):
            if name == "leaky_relu":
                slope = args[1] if len(args) > 1 else kwargs.get("negative_slope", 0.01)
                return self._unary("LeakyRelu", (("negative_slope", float(slope)),))
            return self._unary(_ELEMENTWISE[name])
        raise NotImplementedError(
            f"torch function '{name}' is not in the op set of the MI355X Seastar kernels "
            "(supported: *, +, -, /, sum-over-neighbours, torch.exp, relu, leaky_relu)")

    def __repr__(self) -> str:
        return f"Val({self.node.key})"
