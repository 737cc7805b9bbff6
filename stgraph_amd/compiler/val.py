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

    def sum(self, dim=None, keepdim: bool = False, **kwargs):
        """``Tensor.sum`` over FEATURE dimensions (torch_val.py:172-197): ``dim`` counts the feature dimensions as the
        traced value has them (the vertex / edge dimension is not one of them; negative values from the end), None =
        all of them.  Recorded as a ``Sum`` statement; the generated kernels evaluate it in registers."""
        if kwargs:
            raise NotImplementedError(f"Tensor.sum({', '.join(kwargs)}=...) inside a vertex function")
        shape = self.node.shape
        nd = len(shape)
        dims = tuple(range(nd)) if dim is None else tuple(sorted({(d + nd) % nd for d in ((dim,) if isinstance(dim, int) else dim)}))
        if not dims and nd:
            raise ValueError("empty dim list")
        if any(d < 0 or d >= nd for d in dims):
            raise IndexError(f"sum over dimension(s) {dim} of a value with feature shape {shape}")
        out = tuple(1 if i in dims else s for i, s in enumerate(shape)) if keepdim else \
            tuple(s for i, s in enumerate(shape) if i not in dims)
        node = Node("Sum", self.node.val_type, out, args=(self.node,), params=(("dims", dims), ("keepdim", bool(keepdim))),
                    requires_grad=self.node.requires_grad)
        return Val(self.prog.intern(node), self.prog)

    def view(self, *shape):
        """``Tensor.view`` of the FEATURE shape (torch_val.py:199-227: the run-time call is ``t.view(-1, *shape)``)."""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        total = 1
        for s_ in self.node.shape:
            total *= int(s_)
        shape = [int(x) for x in shape]
        if shape.count(-1) > 1:
            raise ValueError("only one dimension of a view can be inferred")
        known = 1
        for x in shape:
            if x != -1:
                known *= x
        if -1 in shape:
            if known == 0 or total % known:
                raise ValueError(f"shape {tuple(shape)} is invalid for a value of feature shape {self.node.shape}")
            shape[shape.index(-1)] = total // known
            known = total
        if known != total:
            raise ValueError(f"shape {tuple(shape)} is invalid for a value of feature shape {self.node.shape}")
        node = Node("View", self.node.val_type, tuple(shape), args=(self.node,), params=(("shape", tuple(shape)),),
                    requires_grad=self.node.requires_grad)
        return Val(self.prog.intern(node), self.prog)

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


def agg_max(values) -> Val:
    """``AggMax``: the maximum over a vertex's in-neighbours / in-edges of a per-neighbour or per-edge value
    (reference registry.py:295-337; its front end never exposed it -- ``compiler/stgraph.py:6`` has the import commented
    out -- and Python's builtin ``max`` over the one-element neighbour list is the identity).  Use inside
    a vertex function as ``agg_max([nb.h for nb in v.innbs])``.  A vertex without in-edges gets -inf; ties share the
    gradient (``BackwardAMax``: 1 for every edge attaining the maximum)."""
    vals = list(values)
    if len(vals) != 1 or not isinstance(vals[0], Val):
        raise TypeError("agg_max takes the neighbour comprehension of a vertex function")
    v = vals[0]
    if v.val_type not in (ValType.SRC, ValType.EDGE):
        raise TypeError("agg_max([...]) aggregates per-neighbour (SRC) or per-edge (EDGE) values")
    node = Node("AggMax", ValType.DEST, v.node.shape, args=(v.node,), requires_grad=v.node.requires_grad)
    return Val(v.prog.intern(node), v.prog)
