"""Graph IR of a traced vertex function.

Plays the role of the reference's ``Var`` / ``Stmt`` / ``Program`` (compiler/
program.py) and of its value/op type enums (compiler/utils.py:15-45), reduced to
what the kernel dispatcher needs: an expression DAG with hash-consing (which is
what the reference's CSE pass, passes/cse.py, achieves on the statement list) and
a canonical string per node.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum


class ValType(Enum):          # compiler/utils.py:15-19
    SRC = 0
    DEST = 1
    EDGE = 2
    PARAM = 3


_TAG = {ValType.SRC: "S", ValType.DEST: "D", ValType.EDGE: "E", ValType.PARAM: "P"}

COMMUTATIVE = {"Mul", "Add"}


@dataclass(eq=False)
class Node:
    """One GIR value: a leaf (graph feature / parameter) or the result of an op."""

    op: str                                   # 'Leaf' | 'Const' | Mul | Add | Sub | TrueDiv | Exp | LeakyRelu | Relu | AggSum
    val_type: ValType | None
    shape: tuple                              # feature shape (node/edge dimension dropped), e.g. (F,), (H, D), (H, 1)
    args: tuple = ()
    name: str | None = None                   # leaf: feature name ('h', 'norm', 'edge_weight', ...)
    params: tuple = ()                        # op attributes, e.g. (('negative_slope', 0.2),)
    value: object = None                      # Const: python scalar; PARAM leaf: the tensor
    requires_grad: bool = False
    key: str = field(default="", init=False)

    def __post_init__(self):
        self.key = self._make_key()

    def _make_key(self) -> str:
        shp = "x".join(str(s) for s in self.shape)
        if self.op == "Leaf":
            return f"{_TAG[self.val_type]}:{self.name}[{shp}]"
        if self.op == "Const":
            return f"C:{self.value!r}"
        keys = [a.key for a in self.args]
        if self.op in COMMUTATIVE:
            keys = sorted(keys)
        par = "" if not self.params else "{" + ",".join(f"{k}={v!r}" for k, v in self.params) + "}"
        return f"{self.op}{par}({','.join(keys)})"

    def __repr__(self) -> str:
        return self.key


class Program:
    """Statement list in trace order plus a hash-consing table."""

    def __init__(self):
        self.nodes: list[Node] = []
        self._table: dict[str, Node] = {}

    def intern(self, node: Node) -> Node:
        hit = self._table.get(node.key)
        if hit is not None and hit.op != "Leaf":
            return hit
        if hit is not None and hit.op == "Leaf":
            return hit
        self._table[node.key] = node
        self.nodes.append(node)
        return node

    def __str__(self) -> str:
        return "\n".join(f"  %{i}: {n.key}" for i, n in enumerate(self.nodes) if n.op not in ("Leaf", "Const"))


def infer_val_type(args) -> ValType:
    """compiler/utils.py:51-66: mixed SRC/DEST/EDGE operands make an EDGE value; PARAMs are neutral."""
    kinds = [a.val_type for a in args if a.op != "Const" and a.val_type != ValType.PARAM]
    if not kinds:
        return ValType.PARAM
    first = kinds[-1]
    return ValType.EDGE if any(k != first for k in kinds) else first
