"""Generic code generation for traced vertex functions: GIR -> HIP kernels for gfx950.

The reference turns ANY traced vertex function into CUDA: it fuses the GIR into execution units
(compiler/passes/fusion.py:13-59,183-306), differentiates it (compiler/autodiff.py:16-142 with the
per-op rules of compiler/registry.py:195-406), renders one kernel per unit from a jinja template
(compiler/code_gen/templates/fa/tpl_fa_csr*.jinja) and builds it at run time with nvcc.  The three
vertex functions of ``stgraph.nn`` are served by hand-written kernels here (``dispatch.py``); this
module restores the general case with the same division of labour, designed for the MI355X:

* **values** are per source vertex (SRC), per destination vertex (DEST) or per edge (EDGE); an
  aggregation ``sum([...])`` turns per-edge / per-neighbour values into a per-vertex one;
* a **unit** is one row-parallel kernel over a CSR: rows = destinations over the forward CSR, or rows =
  sources over the backward CSR (the reference's Dst-/SrcParallel modes, execution_unit.py:271-282).
  Inside: a sequential loop over the row's edges accumulating every aggregation of the unit (one fp32
  accumulator per (row, feature), edges in CSR order: bit-identical to the reference's loops), then the
  per-vertex statements that consume the sums.  Per-edge and per-neighbour expressions are never
  materialised: every unit re-evaluates them in registers from the leaf tensors; only per-vertex values
  that cross units (or are needed by the backward pass) are written to HBM;
* **reverse mode** runs over the same IR with the reference's local derivatives (registry.py: Mul, Add,
  TrueDiv, Exp, LeakyRelu/BackwardLeakyRelu, Relu/BackwardRelu, AggSum; ``Sub`` with the correct sign,
  SURVEY.md D3).  Gradients of per-source inputs are aggregations over OUT-edges (backward CSR, no
  atomics); gradients of per-destination inputs are aggregations over in-edges; per-edge inputs get
  per-edge writes.  A broadcast operand's adjoint is carried at the broadcast shape down to the leaf and
  reduced there (``sum_to_size``), which needs no cross-lane reduction inside the kernels;
* the source is compiled with hiprtc (``stg_jit_*``, csrc/jit.hip) -- ``--offload-arch=gfx950 -O3
  -ffp-contract=off`` -- once per vertex function and input signature.

Lane mapping: ``G = min(256, pow2 >= F)`` lanes per row (four features per lane from 64 features on), ``256 / G``
rows per workgroup, lanes stride the feature index, so neighbouring lanes read neighbouring floats of one gathered
row.  Unlike the reference (defect D1) every feature column is computed.

Edge loop (round 4; G >= 4): a row's edges are taken ``min(G, 64)`` at a time -- lane ``l`` of the row's group fetches
the column index (edge id, and every scalar per-neighbour input) of edge ``base + l`` once, the values reach the
other lanes by ``__shfl``, and the next chunk's fetch is issued before the current chunk's gathers are consumed.
Per-neighbour inputs of one value per vertex (``norm``: ``KernelSpec.pregather``) are gathered per EDGE once per
graph (``stg_edge_gather_f32``, cached on the CSR) and read with the column index: the dependent second round trip
per edge disappears.  A lane past the end of its row adds the gathered term masked to +0 (its bits AND-ed with
``-(int)ok``) instead of skipping it: a guard or a select around the add is a branch to this compiler, which then
sinks the loads under it and serialises them.  Measured at 1M / 16M (profiles/r04_codegen_vs_handwritten.jsonl):
0.49 / 0.93 / 0.95 of the HBM roofline at F = 16 / 64 / 128, the hand-written kernel 0.41 / 0.89 / 0.94.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import torch

from .. import _C, kernels
from .gir import Node, Program, ValType, infer_val_type

_NODE_TYPES = (ValType.SRC, ValType.DEST)
_BINARY = {"Mul": "*", "Add": "+", "Sub": "-", "TrueDiv": "/"}
_UNARY = ("Exp", "LeakyRelu", "Relu", "BwdLeakyRelu")
_AGGS = ("AggSum", "AggMax")
_RESHAPES = ("Sum", "View")          # evaluated in their operand's own index space (nested block in the kernel)


# ----------------------------------------------------------------------------------------- IR helpers
def _kind(n: Node) -> str:
    return "e" if n.val_type == ValType.EDGE else "p" if n.val_type == ValType.PARAM else "n"


def _broadcast(a: tuple, b: tuple) -> tuple:
    return tuple(torch.broadcast_shapes(a, b))


class Builder:
    """Creates hash-consed nodes (same CSE the tracer gets from ``Program.intern``)."""

    def __init__(self):
        self.prog = Program()

    def const(self, v) -> Node:
        return self.prog.intern(Node("Const", None, (), value=v))

    def leaf(self, name: str, vt: ValType, shape: tuple) -> Node:
        return self.prog.intern(Node("Leaf", vt, tuple(shape), name=name))

    def op(self, op: str, *args: Node, params: tuple = ()) -> Node:
        shape = ()
        for a in args:
            shape = _broadcast(shape, a.shape)
        return self.prog.intern(Node(op, infer_val_type(args), shape, args=tuple(args), params=params))

    def agg(self, arg: Node, vt: ValType) -> Node:
        # over in-edges into a DEST value, over out-edges into a SRC value (the key keeps them apart)
        return self.prog.intern(Node("AggSum", vt, arg.shape, args=(arg,), params=(("to", vt.name),)))

    def reshape(self, op: str, arg: Node, shape: tuple, params: tuple) -> Node:
        """``Sum`` / ``View``: same row type as the operand, feature shape given."""
        return self.prog.intern(Node(op, arg.val_type, tuple(shape), args=(arg,), params=params))

    def sum_to(self, g: Node, shape: tuple) -> Node:
        """``g`` (carried at a broadcast shape) reduced to ``shape`` -- torch's ``sum_to_size`` as a ``Sum`` node."""
        gs = (1,) * (len(shape) - len(g.shape)) + tuple(g.shape)
        if len(gs) > len(shape):
            lead = len(gs) - len(shape)
            g = self.reshape("Sum", g, gs[lead:], (("dims", tuple(range(lead))), ("keepdim", False)))
            gs = gs[lead:]
        elif tuple(g.shape) != gs:
            g = self.reshape("View", g, gs, (("shape", gs),))
        dims = tuple(i for i, (a, b) in enumerate(zip(gs, shape)) if b == 1 and a != 1)
        if dims:
            g = self.reshape("Sum", g, tuple(1 if i in dims else a for i, a in enumerate(gs)), (("dims", dims), ("keepdim", True)))
        return g


def topo(roots) -> list:
    seen, out = set(), []

    def visit(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for a in n.args:
            visit(a)
        out.append(n)
    for r in roots:
        visit(r)
    return out


# ------------------------------------------------------------------------------------------ placement
class Analysis:
    """Which values live in registers and which become tensors, and in which unit.

    inline  : leaves, constants, per-edge expressions and per-vertex expressions of leaves only --
              re-evaluated wherever they are used;
    stage   : for an aggregation, 1 + the latest stage of any materialised per-vertex value its argument
              reads (0 if it reads leaves only); a per-vertex statement that consumes materialised values
              runs in the latest of their stages (after that unit's edge loop);
    unit    : the non-inline per-vertex nodes with the same (row type, stage).
    """

    def __init__(self, roots: list):
        self.roots = list(roots)
        self.order = topo(roots)
        self.inline: dict[int, bool] = {}
        self.stage: dict[int, int] = {}       # non-inline nodes
        self.avail: dict[int, int] = {}       # inline nodes: latest stage of a materialised value they read (-1: none)
        for n in self.order:
            self._place(n)
        self.units: dict[tuple, list] = {}
        for n in self.order:
            if not self.inline[id(n)]:
                self.units.setdefault((n.val_type, self.stage[id(n)]), []).append(n)

    def _ready(self, n: Node) -> int:
        return self.avail[id(n)] if self.inline[id(n)] else self.stage[id(n)]

    def _place(self, n: Node) -> None:
        i = id(n)
        if n.op in ("Leaf", "Const"):
            self.inline[i], self.avail[i] = True, -1
        elif n.op in _AGGS:
            if n.val_type not in _NODE_TYPES:
                raise NotImplementedError("aggregation into a non-vertex value")
            self.inline[i] = False
            self.stage[i] = self._ready(n.args[0]) + 1
        elif n.val_type in _NODE_TYPES and any(not self.inline[id(a)] for a in n.args):
            self.inline[i] = False
            self.stage[i] = max(self.stage[id(a)] for a in n.args if not self.inline[id(a)])
            if n.op in _RESHAPES:
                self.stage[i] += 1          # reads OTHER feature columns of a materialised value: a later kernel
        else:
            self.inline[i] = True
            self.avail[i] = max([self._ready(a) for a in n.args], default=-1)

    def is_tensor(self, n: Node) -> bool:
        return n.op == "Leaf" or not self.inline[id(n)]


# ------------------------------------------------------------------------------------------- autodiff
def differentiate(fwd: Analysis, rets: list, builder: Builder):
    """Reverse mode over the traced DAG.  Returns ``(grad_roots, saved)``: for every leaf node that
    requires grad the node holding its adjoint (built with ``builder``), and the forward non-inline nodes
    the backward expressions read (they must be materialised by the forward kernels)."""
    saved: dict[int, Node] = {}          # id(forward node) -> leaf standing for its tensor in backward space
    saved_nodes: list = []
    cut_memo: dict[int, Node] = {}

    def cut(x: Node) -> Node:
        """The forward value ``x`` as seen from the backward graph."""
        i = id(x)
        if i in cut_memo:
            return cut_memo[i]
        if x.op == "Const":
            r = builder.const(x.value)
        elif x.op == "Leaf":
            r = builder.leaf(x.name, x.val_type, x.shape)
        elif not fwd.inline[i]:
            r = builder.leaf(f"__saved{len(saved_nodes)}", x.val_type, x.shape)
            saved[i] = r
            saved_nodes.append(x)
        elif x.op in _RESHAPES:
            r = builder.reshape(x.op, cut(x.args[0]), x.shape, x.params)
        else:
            r = builder.op(x.op, *[cut(a) for a in x.args], params=x.params)
        cut_memo[i] = r
        return r

    adj: dict[int, list] = {}
    for k, r in enumerate(rets):
        adj.setdefault(id(r), []).append(builder.leaf(f"__gout{k}", r.val_type, r.shape))
    grad_roots: dict[int, tuple] = {}
    for n in reversed(fwd.order):
        parts = adj.get(id(n))
        if not parts or not n.requires_grad:
            continue
        g = parts[0]
        for p in parts[1:]:
            g = builder.op("Add", g, p)
        if n.op == "Leaf":
            if n.val_type == ValType.PARAM and g.val_type == ValType.PARAM:
                raise NotImplementedError("gradient of a module parameter that meets no graph feature")
            grad_roots[id(n)] = (n, g)                    # PARAM: per-row / per-edge contributions, summed by the plan
            continue
        if n.op in _RESHAPES and tuple(g.shape) != tuple(n.shape):
            g = builder.sum_to(g, n.shape)                # the adjoint arrives at a broadcast shape
        for pos, a in enumerate(n.args):
            if a.op == "Const" or not a.requires_grad:
                continue
            c = _local_derivative(builder, cut, n, pos, g)
            if a.val_type in _NODE_TYPES and c.val_type != a.val_type:
                c = builder.agg(c, a.val_type)            # registry.py:180-188: per-edge adjoint of a per-vertex value
            adj.setdefault(id(a), []).append(c)
    return list(grad_roots.values()), saved_nodes, saved


def _local_derivative(b: Builder, cut, n: Node, pos: int, g: Node) -> Node:
    if n.op == "Mul":                                   # registry.py:255-259
        return b.op("Mul", g, cut(n.args[1 - pos]))
    if n.op == "Add":                                   # registry.py:195-198
        return g
    if n.op == "Sub":                                   # registry.py:210-213 uses +1 for both operands; -1 here
        return g if pos == 0 else b.op("Mul", b.const(-1.0), g)
    if n.op == "TrueDiv":                               # registry.py:339-352
        x0, x1 = cut(n.args[0]), cut(n.args[1])
        if pos == 0:
            return b.op("Mul", g, b.op("TrueDiv", b.const(1.0), x1))
        return b.op("Mul", g, b.op("TrueDiv", b.op("Mul", b.const(-1.0), x0), b.op("Mul", x1, x1)))
    if n.op == "Exp":                                   # registry.py:242-245
        return b.op("Mul", g, cut(n))
    if n.op == "LeakyRelu":                             # registry.py:225-232, 397-406
        return b.op("Mul", g, b.op("BwdLeakyRelu", cut(n.args[0]), params=n.params))
    if n.op == "Relu":                                  # registry.py:363-367, 378-388
        return b.op("BwdRelu", cut(n.args[0]), g)
    if n.op == "AggSum":                                # registry.py:269-276
        return g
    if n.op == "AggMax":                                # registry.py:295-306, 326-337: 1 where the edge attains the maximum
        return b.op("Mul", g, b.op("BwdAMax", cut(n.args[0]), cut(n)))
    if n.op == "Sum":                                   # broadcast back over the reduced dimensions
        dims, keep = dict(n.params)["dims"], dict(n.params)["keepdim"]
        if keep:
            return g
        ks = tuple(1 if i in dims else s_ for i, s_ in enumerate(n.args[0].shape))
        return b.reshape("View", g, ks, (("shape", ks),))
    if n.op == "View":
        a = n.args[0]
        return b.reshape("View", g, a.shape, (("shape", tuple(a.shape)),))
    raise NotImplementedError(f"no gradient rule for {n.op}")


# ------------------------------------------------------------------------------------------- emission
def _pow2_at_least(x: int) -> int:
    p = 1
    while p < x:
        p *= 2
    return p


def _numel(shape: tuple) -> int:
    out = 1
    for s in shape:
        out *= int(s)
    return out


def _index_expr(shape: tuple, full: tuple, tx: str = "tx") -> str:
    """Flat index of the element of an operand of ``shape`` that index ``tx`` (flat in ``full``) reads."""
    if _numel(shape) == 1:
        return "0"
    if tuple(shape) == tuple(full):
        return tx
    if len(shape) > len(full) or any(a not in (1, b) for a, b in zip((1,) * (len(full) - len(shape)) + tuple(shape), full)):
        raise NotImplementedError(f"operand of shape {shape} in an index space of shape {full}")
    sh = (1,) * (len(full) - len(shape)) + tuple(shape)
    terms, stride_full, stride_op = [], 1, 1
    for j in range(len(full) - 1, -1, -1):
        if sh[j] != 1:
            coord = f"(({tx} / {stride_full}) % {full[j]})" if stride_full > 1 else f"({tx} % {full[j]})"
            terms.append(coord if stride_op == 1 else f"{coord} * {stride_op}")
            stride_op *= sh[j]
        stride_full *= full[j]
    return " + ".join(terms) if terms else "0"


def _canonical_cond(shape: tuple, full: tuple) -> str | None:
    """Condition under which lane ``tx`` is the one lane that writes an output of ``shape``."""
    if tuple(shape) == tuple(full):
        return None
    sh = (1,) * (len(full) - len(shape)) + tuple(shape)
    conds, stride = [], 1
    for j in range(len(full) - 1, -1, -1):
        if sh[j] == 1 and full[j] != 1:
            conds.append(f"((tx / {stride}) % {full[j]}) == 0" if stride > 1 else f"(tx % {full[j]}) == 0")
        stride *= full[j]
    return " && ".join(conds) if conds else None


def _lit(v) -> str:
    """Constants as the reference's emitter prints them (registry.py: the Python value is formatted into the
    CUDA text): integral values are int literals (float arithmetic), anything else is a DOUBLE literal, so
    ``x * 0.2`` is a double multiply rounded once on assignment -- kept for bit parity."""
    f = float(v)
    if f == int(f) and abs(f) < 1e9:
        return f"{int(f)}"
    return repr(f)


@dataclass
class KernelSpec:
    name: str
    row_type: ValType | None            # DEST: rows = dst over the forward CSR; SRC: rows = src over the backward CSR
    has_loop: bool
    full: tuple                         # feature shape the lanes enumerate
    tensors: list = field(default_factory=list)     # ordered tensor keys (inputs then outputs)
    outputs: list = field(default_factory=list)     # (key, rows 'N'|'E', shape)
    uses_eids: bool = False
    source: str = ""
    stage: int = 0
    pregather: list = field(default_factory=list)   # indices into `tensors`: per-neighbour SCALAR inputs the kernel takes gathered into CSR order

    @property
    def csr_side(self) -> str:
        return "bwd" if self.row_type == ValType.SRC else "fwd"

    @property
    def vec(self) -> int:
        """Consecutive feature indices per lane: 4 when the innermost enumerated dimension is a multiple of 4
        (a lane's four elements are then contiguous in every tensor that is indexed by the feature at all, and
        16-byte aligned: the compiler fuses their loads / stores into dwordx4 accesses)."""
        inner = self.full[-1] if self.full else 1
        # measured at |V| = 1M, |E| = 16M (profiles/r01_codegen_vs_handwritten.jsonl): F = 128: 0.60 -> 0.78 of the
        # HBM roofline, F = 64: 0.66 -> 0.67, F = 16: 0.35 -> 0.32 (four lanes per row leave too few rows in flight)
        return 4 if inner % 4 == 0 and _numel(self.full) >= 64 else 1

    @property
    def lanes_per_row(self) -> int:
        return min(256, _pow2_at_least(max(1, _numel(self.full) // self.vec)))


class _Emitter:
    """Emits one unit.  ``key_of(node)`` names the tensor a leaf / materialised node lives in."""

    def __init__(self, an: Analysis, key_of, name: str, row_type, stage: int):
        self.an, self.key_of, self.name, self.row_type, self.stage = an, key_of, name, row_type, stage
        self.tensors: list = []
        self.outputs: list = []
        self.uses_eids = False
        self.full: tuple = ()
        self.ctx: list = []                    # nested index spaces: (full, tx variable, id) of Sum / View operands
        self._nvar = 0

    def space(self):
        return self.ctx[-1] if self.ctx else (self.full, "tx", 0)

    def arg(self, key) -> str:
        if key not in self.tensors:
            self.tensors.append(key)
        return f"T{self.tensors.index(key)}"

    def widen(self, shape: tuple) -> None:
        self.full = _broadcast(self.full, tuple(shape))

    # -- expression generation --------------------------------------------------------------------
    def _load(self, n: Node, level: str) -> str:
        """Read the tensor behind ``n`` (a leaf or a materialised node) at ``level`` ('row' | 'edge')."""
        full, tx, _ = self.space()
        t, size, idx = self.arg(self.key_of(n)), _numel(n.shape), _index_expr(n.shape, full, tx)
        if n.val_type == ValType.PARAM:
            return f"{t}[{idx}]"
        if n.val_type == ValType.EDGE:
            if level != "edge":
                raise NotImplementedError("per-edge value used in a per-vertex statement")
            self.uses_eids = True
            where = "eid"
        elif n.val_type == self.row_type or self.row_type is None:
            where = "row"
        else:
            if level != "edge":
                raise NotImplementedError("neighbour value used outside the edge loop")
            where = "c"
        return f"{t}[(long)({where}) * {size} + {idx}]" if size > 1 else f"{t}[{where}]"

    def expr(self, n: Node, level: str, lines: list, memo: dict, local: dict) -> str:
        """C expression (a variable name or literal) for ``n`` at ``level``; statements go to ``lines``."""
        i = id(n)
        full, tx, cid = self.space()
        if i in local and cid == 0:                      # a per-vertex statement of this unit, already computed
            return local[i]
        if (i, level, cid) in memo:
            return memo[(i, level, cid)]
        if n.op == "Const":
            return _lit(n.value)
        self._nvar += 1
        var = f"v{self._nvar}_{level[0]}"
        if self.an.is_tensor(n):
            lines.append(f"const float {var} = {self._load(n, level)};")
        elif n.op in _RESHAPES:
            self._reshape(n, var, level, lines, memo, local)
        else:
            a = [self.expr(x, level, lines, memo, local) for x in n.args]
            lines.append(f"const float {var} = {_op_code(n, a)};")
        memo[(i, level, cid)] = var
        return var

    def _reshape(self, n: Node, var: str, level: str, lines: list, memo: dict, local: dict) -> None:
        """``Sum`` / ``View``: the operand lives in its own index space.  The lane's output element (flat index ``o`` in
        n's shape) selects, for View, the operand element with the same flat index; for Sum, the operand elements
        whose non-reduced coordinates are o's, visited in index order (one sequential fp32 sum)."""
        full, tx, _ = self.space()
        arg = n.args[0]
        A = tuple(arg.shape)
        self._nvar += 1
        k = self._nvar
        o = f"o{k}"
        if n.op == "View":
            lines.append(f"float {var};")
            lines.append(f"{{ const int {o} = {_index_expr(n.shape, full, tx)};")
            self.ctx.append((A, o, k))
            inner: list = []
            v = self.expr(arg, level, inner, memo, local)
            self.ctx.pop()
            lines += ["  " + ln for ln in inner] + [f"  {var} = {v}; }}"]
            return
        dims, keep = dict(n.params)["dims"], dict(n.params)["keepdim"]
        K = tuple(1 if i in dims else s_ for i, s_ in enumerate(A))          # keepdim shape
        out_shape = K if keep else tuple(s_ for i, s_ in enumerate(A) if i not in dims)
        assert tuple(out_shape) == tuple(n.shape), (out_shape, n.shape)
        R = 1
        for d in dims:
            R *= A[d]
        # flat operand index from o (flat over the kept dims, in order) and r (flat over the reduced dims)
        terms, astride, kstride, rstride = [], 1, 1, 1
        for i in range(len(A) - 1, -1, -1):
            if A[i] != 1:
                if i in dims:
                    coord = f"((r{k} / {rstride}) % {A[i]})" if rstride > 1 else f"(r{k} % {A[i]})"
                else:
                    coord = f"(({o} / {kstride}) % {A[i]})" if kstride > 1 else f"({o} % {A[i]})"
                terms.append(coord if astride == 1 else f"{coord} * {astride}")
            if i in dims:
                rstride *= A[i]
            else:
                kstride *= A[i]
            astride *= A[i]
        lines.append(f"float {var} = 0.0f;")
        lines.append(f"{{ const int {o} = {_index_expr(n.shape, full, tx)};")
        lines.append(f"  for (int r{k} = 0; r{k} < {R}; ++r{k}) {{")
        lines.append(f"    const int t{k} = {' + '.join(terms) if terms else '0'};")
        self.ctx.append((A, f"t{k}", k))
        inner = []
        v = self.expr(arg, level, inner, memo, local)
        self.ctx.pop()
        lines += ["    " + ln for ln in inner] + [f"    {var} = {var} + {v};", "  } }"]


def _op_code(n: Node, a: list) -> str:
    if n.op in _BINARY:
        return f"{a[0]} {_BINARY[n.op]} {a[1]}"
    if n.op == "Exp":
        return f"expf({a[0]})"
    slope = dict(n.params).get("negative_slope", 0.01)
    if n.op == "LeakyRelu":
        return f"{a[0]} > 0 ? {a[0]} : {_lit(slope)} * {a[0]}"
    if n.op == "Relu":
        return f"{a[0]} > 0 ? {a[0]} : 0"
    if n.op == "BwdLeakyRelu":
        return f"{a[0]} > 0 ? 1 : {_lit(slope)}"
    if n.op == "BwdRelu":
        return f"{a[0]} > 0 ? {a[1]} : 0"
    if n.op == "BwdAMax":
        return f"{a[0]} == {a[1]} ? 1 : 0"
    raise NotImplementedError(f"no code for op {n.op}")


def _collect_shapes(an: Analysis, n: Node, em: _Emitter, seen: set) -> None:
    """Widen the unit's lane space over everything evaluated inline under ``n``."""
    if id(n) in seen:
        return
    seen.add(id(n))
    em.widen(n.shape)
    if not an.is_tensor(n) and n.op not in _RESHAPES:    # a Sum / View operand is enumerated by its own nested loop
        for a in n.args:
            _collect_shapes(an, a, em, seen)


def emit_unit(an: Analysis, key_of, name: str, row_type, stage: int, nodes: list, needed: set) -> KernelSpec:
    """One kernel for the non-inline per-vertex ``nodes`` of a unit; ``needed`` = ids of nodes to write out."""
    em = _Emitter(an, key_of, name, row_type, stage)
    seen: set = set()
    for n in nodes:
        em.widen(n.shape)
        if n.op in _RESHAPES:
            continue
        for a in n.args:
            if n.op in _AGGS or an.inline[id(a)]:
                _collect_shapes(an, a, em, seen)
    aggs = [n for n in nodes if n.op in _AGGS]
    edge_lines, memo = [], {}
    acc = {}
    for k, n in enumerate(aggs):
        v = em.expr(n.args[0], "edge", edge_lines, memo, {})
        acc[id(n)] = f"acc{k}"
        edge_lines.append(f"acc{k} += {v};" if n.op == "AggSum" else f"acc{k} = fmaxf({v}, acc{k});")
    post, local = [], {}
    for n in nodes:
        if n.op in _AGGS:
            local[id(n)] = acc[id(n)]
        elif n.op in _RESHAPES:
            var = f"r{len(local)}"
            tmp: list = []
            em._reshape(n, var, "row", tmp, memo, local)
            post += tmp
            local[id(n)] = var
        else:
            a = [em.expr(x, "row", post, memo, local) for x in n.args]
            var = f"r{len(local)}"
            post.append(f"const float {var} = {_op_code(n, a)};")
            local[id(n)] = var
        if id(n) in needed:
            key = key_of(n)
            t = em.arg(key)
            em.outputs.append((key, "N", n.shape))
            size, idx, cond = _numel(n.shape), _index_expr(n.shape, em.full), _canonical_cond(n.shape, em.full)
            store = f"{t}[(long)row * {size} + {idx}] = {local[id(n)]};"
            post.append(f"if ({cond}) {store}" if cond else store)
    return _finish(em, ["0.0f" if n.op == "AggSum" else "-__builtin_inff()" for n in aggs], edge_lines, post, has_loop=True)


def emit_row_only(an: Analysis, key_of, name: str, row_type, roots: list) -> KernelSpec:
    """Per-vertex outputs that need no edge loop (expressions of leaves / materialised values only)."""
    em = _Emitter(an, key_of, name, row_type, 1 << 30)
    seen: set = set()
    for n in roots:
        _collect_shapes(an, n, em, seen)
    post, memo = [], {}
    for n in roots:
        v = em.expr(n, "row", post, memo, {})
        key = ("out", id(n))
        t = em.arg(key)
        em.outputs.append((key, "N", n.shape))
        size, idx, cond = _numel(n.shape), _index_expr(n.shape, em.full), _canonical_cond(n.shape, em.full)
        store = f"{t}[(long)row * {size} + {idx}] = {v};"
        post.append(f"if ({cond}) {store}" if cond else store)
    return _finish(em, [], [], post, has_loop=False)


def emit_edge_outputs(an: Analysis, key_of, name: str, roots: list) -> KernelSpec:
    """Per-edge outputs (gradients of per-edge inputs): one pass over the forward CSR writing ``out[eid]``."""
    em = _Emitter(an, key_of, name, ValType.DEST, 1 << 30)
    seen: set = set()
    for n in roots:
        _collect_shapes(an, n, em, seen)
    lines, memo = [], {}
    em.uses_eids = True
    for n in roots:
        v = em.expr(n, "edge", lines, memo, {})
        key = ("out", id(n))
        t = em.arg(key)
        em.outputs.append((key, "E", n.shape))
        size, idx, cond = _numel(n.shape), _index_expr(n.shape, em.full), _canonical_cond(n.shape, em.full)
        store = f"{t}[(long)eid * {size} + {idx}] = {v};"
        lines.append(f"if ({cond}) {store}" if cond else store)
    return _finish(em, [], lines, [], has_loop=True)


def _finish(em: _Emitter, init: list, edge_lines: list, post: list, has_loop: bool) -> KernelSpec:
    spec = KernelSpec(em.name, em.row_type, has_loop, em.full or (1,), em.tensors, em.outputs, em.uses_eids,
                      stage=em.stage)
    G, V = spec.lanes_per_row, spec.vec
    fmax = _numel(spec.full)
    naccs = len(init)
    params = "".join(f"float *__restrict__ T{i}_, " for i in range(len(em.tensors)))
    # per-neighbour scalar INPUTS (norm[c] ...) read in the edge loop arrive gathered into CSR order (one coalesced stream instead of
    # a dependent 4-byte gather that costs a whole sector per edge: csrc/gcn_agg.hip "PRE"); the launcher caches the gathered array
    # per (CSR, tensor version).  Decided below, once the edge statements are known: the parameter list is patched in at the end.
    body = [f'extern "C" __global__ void __launch_bounds__(256) {em.name}(', f"    {params}@PG@",
            "    const int *__restrict__ row_offset, const int *__restrict__ col_idx, const int *__restrict__ eids,",
            "    const int *__restrict__ node_ids, int N)", "{"]
    # torch allocations are 256-byte aligned and row sizes that reach the vector path are multiples of 4 floats
    body += [f"    float *__restrict__ T{i} = (float *)__builtin_assume_aligned(T{i}_, 16);" for i in range(len(em.tensors))]
    body += [f"    const int r_idx = blockIdx.x * {256 // G} + (int)(threadIdx.x / {G});",
             "    if (r_idx >= N) return;",
             "    const int row = node_ids ? node_ids[r_idx] : r_idx;"]
    if has_loop:
        body.append("    const int beg = row_offset[row], end = row_offset[row + 1];")
    body.append(f"    for (int tx0 = (int)(threadIdx.x % {G}) * {V}; tx0 < {fmax}; tx0 += {G * V}) {{")
    ind1, ind2, ind3 = " " * 8, " " * 12, " " * 16
    for k in range(naccs):
        body.append(ind1 + f"float acc{k}_[{V}];")
    if naccs:
        body.append(ind1 + f"for (int q = 0; q < {V}; ++q) {{ " + " ".join(f"acc{k}_[q] = {init[k]};" for k in range(naccs)) + " }")
    refs = [f"float &acc{k} = acc{k}_[q];" for k in range(naccs)]
    if has_loop and G >= 4:
        # The hand-written mapping (csrc/gcn_agg.hip) as a template: edges in chunks of U; the G lanes of a row fetch the chunk's
        # column indices (and edge ids) as ONE coalesced request -- lane l takes edge l mod U -- and every per-edge SCALAR operand
        # (a [.., 1]-shaped neighbour value such as norm[c], an edge weight w[eid]) the same way, one dependent round trip per
        # chunk instead of one per edge; the values reach the other lanes by __shfl inside the row's lane group.  Then the U
        # feature gathers are issued together (no load sits behind a guard: lanes past the row's end re-read its last edge) and
        # the sums are taken in CSR order, guarded -- one fp32 accumulator per (row, feature): the reference's sums, bit for bit.
        import re
        GW = min(G, 64)                                   # the lanes of a row inside one wave: the scope of __shfl
        U = min(8, GW)
        scal, edge_body = [], []
        for ln in edge_lines:
            m = re.match(r"^const float (v\d+_e) = (T\d+)\[(c|eid)\];$", ln)
            if m:
                var, t, where = m.group(1), m.group(2), m.group(3)
                key = em.tensors[int(t[1:])]
                if where == "c" and isinstance(key, tuple) and key and key[0] == "leaf":
                    spec.pregather.append(int(t[1:]))
                    t, where = f"PG{len(spec.pregather) - 1}", "e"
                scal.append((var, t, where))
            else:
                edge_body.append(ln)
        guarded = []
        for ln in edge_body:
            m = re.match(r"^(acc\d+) \+= (.*);$", ln)
            m2 = re.match(r"^(acc\d+) = fmaxf\((.*), (acc\d+)\);$", ln)
            if m:
                # Not a branch and not a select either: the term of a lane past its row's end is masked to +0.0 and ADDED (x + 0
                # is x).  A guarded add -- or a select, which the optimiser turns back into a branch when one side hangs on a
                # load -- pulls the gathers behind it under the condition: one guarded dword load + s_waitcnt vmcnt(0) per
                # element (measured: 0.45 of the roofline against 0.79 for the plain loop).
                guarded.append(f"{m.group(1)} = {m.group(1)} + __int_as_float(__float_as_int({m.group(2)}) & okm);")
            elif m2:
                guarded.append(f"{m2.group(1)} = fmaxf(__int_as_float((__float_as_int({m2.group(2)}) & okm) | (~okm & (int)0xff800000u)), {m2.group(3)});")
            else:
                guarded.append(("if (ok) " + ln) if re.match(r"^T\d+\[", ln) else ln)     # per-edge stores: real edges only
        body.append(ind1 + f"const int l_ = (int)(threadIdx.x % {GW}) % {U};")
        # two chunks deep: the records of chunk k + 1 (index, then the scalars behind it) are loaded while chunk k is summed
        body.append(ind1 + "int cl_ = 0; (void)cl_;" + (" int eidl_ = 0; (void)eidl_;" if em.uses_eids else "") +
                    "".join(f" float sl{k}_ = 0.0f;" for k in range(len(scal))))
        body.append(ind1 + "if (beg < end) {")
        body.append(ind2 + "const int el_ = min(beg + l_, end - 1);")
        body.append(ind2 + "cl_ = col_idx[el_];")
        if em.uses_eids:
            body.append(ind2 + "eidl_ = eids[el_];")
        for k, (_, t, where) in enumerate(scal):
            body.append(ind2 + f"sl{k}_ = {t}[{ {'c': 'cl_', 'eid': 'eidl_', 'e': 'el_'}[where] }];")
        body.append(ind1 + "}")
        body.append(ind1 + f"for (int e0 = beg; e0 < end; e0 += {U}) {{")
        body.append(ind2 + f"const int eln_ = min(e0 + {U} + l_, end - 1);")
        body.append(ind2 + "const int cln_ = col_idx[eln_]; (void)cln_;")
        if em.uses_eids:
            body.append(ind2 + "const int eidln_ = eids[eln_]; (void)eidln_;")
        body.append(ind2 + f"int cs_[{U}];" + (f" int eids_[{U}];" if em.uses_eids else "") +
                    "".join(f" float ss{k}_[{U}];" for k in range(len(scal))))
        body.append("#pragma unroll")
        body.append(ind2 + f"for (int u = 0; u < {U}; ++u) {{")
        body.append(ind3 + f"cs_[u] = __shfl(cl_, u, {GW});")
        if em.uses_eids:
            body.append(ind3 + f"eids_[u] = __shfl(eidl_, u, {GW});")
        for k in range(len(scal)):
            body.append(ind3 + f"ss{k}_[u] = __shfl(sl{k}_, u, {GW});")
        body.append(ind2 + "}")
        body.append("#pragma unroll")
        body.append(ind2 + f"for (int u = 0; u < {U}; ++u) {{")
        body.append(ind3 + "const int c = cs_[u]; (void)c;")
        if em.uses_eids:
            body.append(ind3 + "const int eid = eids_[u]; (void)eid;")
        body.append(ind3 + "const bool ok = e0 + u < end; (void)ok;")
        body.append(ind3 + "const int okm = -(int)ok; (void)okm;")
        for k, (var, _, _) in enumerate(scal):
            body.append(ind3 + f"const float {var} = ss{k}_[u];")
        body.append("#pragma unroll")
        body.append(ind3 + f"for (int q = 0; q < {V}; ++q) {{")
        ind4 = " " * 20
        body.append(ind4 + "const int tx = tx0 + q; (void)tx;")
        body += [ind4 + r for r in refs]
        body += [ind4 + s for s in guarded]
        body.append(ind3 + "}")
        body.append(ind2 + "}")
        for k, (_, t, where) in enumerate(scal):
            body.append(ind2 + f"sl{k}_ = {t}[{ {'c': 'cln_', 'eid': 'eidln_', 'e': 'eln_'}[where] }];")
        body.append(ind2 + "cl_ = cln_;" + (" eidl_ = eidln_;" if em.uses_eids else ""))
        body.append(ind1 + "}")
    elif has_loop:
        # fewer than four lanes per row: the plain loop, unrolled so that the loads of several edges are in flight together;
        # the adds stay in CSR order per (row, feature) (no reassociation without fast-math), so the sums are unchanged
        body.append("#pragma unroll 4")
        body.append(ind1 + "for (int e = beg; e < end; ++e) {")
        body.append(ind2 + "const int c = col_idx[e]; (void)c;")
        if em.uses_eids:
            body.append(ind2 + "const int eid = eids[e];")
        body.append("#pragma unroll")
        body.append(ind2 + f"for (int q = 0; q < {V}; ++q) {{")
        body.append(ind3 + "const int tx = tx0 + q; (void)tx;")
        body += [ind3 + r for r in refs]
        body += [ind3 + s for s in edge_lines]
        body.append(ind2 + "}")
        body.append(ind1 + "}")
    body.append("#pragma unroll")
    body.append(ind1 + f"for (int q = 0; q < {V}; ++q) {{")
    body.append(ind2 + "const int tx = tx0 + q; (void)tx;")
    body += [ind2 + r for r in refs]
    body += [ind2 + s for s in post]
    body.append(ind1 + "}")
    body += ["    }", "}", ""]
    pg = "".join(f"const float *__restrict__ PG{j}, " for j in range(len(spec.pregather)))
    spec.source = "\n".join(body).replace("@PG@", pg)
    return spec


def _fit(g: torch.Tensor, shape: tuple) -> torch.Tensor:
    """An adjoint carried at a broadcast shape brought to the leaf's shape: summed over the dimensions the leaf has
    with size 1 (it was broadcast in the forward pass), expanded over those a ``Sum`` reduced away."""
    shape = tuple(shape)
    gs = tuple(g.shape)
    if len(gs) < len(shape):
        g = g.reshape((1,) * (len(shape) - len(gs)) + gs)
        gs = tuple(g.shape)
    lead = len(gs) - len(shape)
    tgt = (1,) * lead + shape
    inter = tuple(1 if t == 1 else a for a, t in zip(gs, tgt))
    if inter != gs:
        g = g.sum_to_size(inter)
    if inter != tgt:
        g = g.expand(tgt)
    return g.reshape(shape).contiguous() if lead else g.contiguous()


# ------------------------------------------------------------------------------------------- the plan
class _Module:
    """hiprtc-compiled code object for all kernels of one plan; loaded on the device on first launch."""

    def __init__(self, source: str, name: str):
        self.source = source
        code, size, log = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p()
        _C.check(_C.lib.stg_jit_compile(source.encode(), name.encode(), ctypes.byref(code), ctypes.byref(size),
                                        ctypes.byref(log)))
        if log.value:
            _C.lib.stg_jit_free(log)
        self._code = ctypes.string_at(code.value, size.value)
        _C.lib.stg_jit_free(code)
        self._modules: dict[int, ctypes.c_void_p] = {}
        self._functions: dict[tuple, ctypes.c_void_p] = {}

    def function(self, device: torch.device, name: str) -> ctypes.c_void_p:
        dev = device.index or 0
        f = self._functions.get((dev, name))
        if f is None:
            with torch.cuda.device(device):
                m = self._modules.get(dev)
                if m is None:
                    m = ctypes.c_void_p()
                    _C.check(_C.lib.stg_jit_load(self._code, ctypes.byref(m)))
                    self._modules[dev] = m
                f = ctypes.c_void_p()
                _C.check(_C.lib.stg_jit_get_function(m, name.encode(), ctypes.byref(f)))
            self._functions[(dev, name)] = f
        return f


class GenericPlan:
    """Kernel plan generated from the GIR of an arbitrary vertex function (same interface as the
    hand-written plans of ``dispatch.py``)."""

    name = "generated"
    _count = 0

    def __init__(self, rets: list, program: Program):
        GenericPlan._count += 1
        self.uid = GenericPlan._count
        self.rets = list(rets)
        for r in self.rets:
            if r.val_type not in _NODE_TYPES and r.val_type != ValType.EDGE:
                raise NotImplementedError("a vertex function must return per-vertex or per-edge values")
        leaves = [n for n in topo(self.rets) if n.op == "Leaf"]
        self._inputs, seen = [], set()
        self._params: dict = {}              # name -> the module's parameter / buffer read inside the vertex function
        for l in leaves:
            k = (_kind(l), l.name)
            if l.val_type == ValType.PARAM:
                self._params[l.name] = l.value
            if k not in seen:
                seen.add(k)
                self._inputs.append(k)
        self._diff = sorted({(_kind(l), l.name) for l in leaves if l.requires_grad})
        self._input_shape = {(_kind(l), l.name): l.shape for l in leaves}

        # forward placement, reverse mode, then forward placement again with the saved values as extra roots
        fwd0 = Analysis(self.rets)
        self._bwd_builder = Builder()
        self.grad_roots, self.saved_nodes, saved_leaf = [], [], {}
        if self._diff:
            self.grad_roots, self.saved_nodes, saved_leaf = differentiate(fwd0, self.rets, self._bwd_builder)
        self.fwd = Analysis(self.rets + self.saved_nodes)
        self._saved_name = {id(x): saved_leaf[id(x)].name for x in self.saved_nodes}
        fwd_roots = self.rets + self.saved_nodes
        self.fwd_kernels, self._fwd_out_key = self._build_pass(self.fwd, fwd_roots, [r.val_type for r in fwd_roots],
                                                               f"stg_f{self.uid}")
        self.bwd_kernels, self._bwd_out_key = [], {}
        if self.grad_roots:
            roots = [g for _, g in self.grad_roots]
            self.bwd = Analysis(roots)
            # a parameter's adjoint is built per row (or per edge) of whatever it was combined with, then summed
            targets = [g.val_type if l.val_type == ValType.PARAM else l.val_type for l, g in self.grad_roots]
            self.bwd_kernels, self._bwd_out_key = self._build_pass(self.bwd, roots, targets, f"stg_b{self.uid}")
        self.source = "\n".join(k.source for k in self.fwd_kernels + self.bwd_kernels)
        self.module = _Module(self.source, f"stg_generated_{self.uid}.hip")

    # -- construction -------------------------------------------------------------------------------
    @staticmethod
    def _leaf_key(n: Node):
        return ("leaf", _kind(n), n.name)

    def params(self) -> dict:
        """{name: tensor} of the module parameters / buffers the vertex function reads (autograd inputs of the call)."""
        return dict(self._params)

    def _build_pass(self, an: Analysis, roots: list, targets: list, prefix: str):
        """Kernels (in launch order) computing ``roots``; ``targets[i]`` is the kind of tensor root i has to
        become (per-vertex [N, ...] or per-edge [E, ...]).  Returns the kernels and {id(root): tensor key}."""
        def key_of(n: Node):
            return self._leaf_key(n) if n.op == "Leaf" else ("tmp", id(n))
        # which non-inline nodes must be written: roots, and anything read from another unit
        needed = {id(r) for r in roots if not an.inline[id(r)]}
        unit_of = {id(n): u for u, ns in an.units.items() for n in ns}

        def scan(n, unit, seen):
            if id(n) in seen:
                return
            seen.add(id(n))
            if n.op in ("Leaf", "Const"):
                return
            if not an.inline[id(n)]:
                if unit_of[id(n)] != unit:
                    needed.add(id(n))
                return
            for a in n.args:
                scan(a, unit, seen)
        for u, ns in an.units.items():
            seen: set = set()
            for n in ns:
                for a in n.args:
                    scan(a, u, seen)
        edge_roots = [r for r, t in zip(roots, targets) if t == ValType.EDGE]
        for r, t in zip(roots, targets):
            if t != ValType.EDGE and r.val_type != t and r.op != "Const":
                raise NotImplementedError(f"a {r.val_type.name} value cannot become a {t.name} tensor")
            if an.inline[id(r)] or t == ValType.EDGE:
                scan(r, None, set())
        specs = []
        for k, (u, ns) in enumerate(sorted(an.units.items(), key=lambda kv: (kv[0][1], kv[0][0].value))):
            specs.append(emit_unit(an, key_of, f"{prefix}_u{k}", u[0], u[1], ns, needed))
        out_key = {id(r): key_of(r) for r, t in zip(roots, targets) if not an.inline[id(r)] and t != ValType.EDGE}
        inline_roots = [r for r, t in zip(roots, targets) if an.inline[id(r)] and t != ValType.EDGE]
        for vt, tag in ((ValType.DEST, "d"), (ValType.SRC, "s")):
            group = [r for r in inline_roots if r.val_type == vt or (r.op == "Const" and vt == ValType.DEST)]
            if group:
                specs.append(emit_row_only(an, key_of, f"{prefix}_row{tag}", vt, group))
                out_key.update({id(r): ("out", id(r)) for r in group})
        if edge_roots:
            specs.append(emit_edge_outputs(an, key_of, f"{prefix}_edge", edge_roots))
            out_key.update({id(r): ("out", id(r)) for r in edge_roots})
        specs.sort(key=lambda s: s.stage)
        return specs, out_key

    # -- plan interface -----------------------------------------------------------------------------
    def input_names(self):
        return list(self._inputs)

    def differentiable(self):
        return list(self._diff)

    def _launch(self, spec: KernelSpec, graph, env: dict, N: int, E: int, device) -> None:
        use_nid = kernels.rows_by_node_ids(graph.graph_type())
        csr = graph.csr(spec.csr_side)
        for key, rows, shape in spec.outputs:
            env[key] = torch.empty((N if rows == "N" else E,) + tuple(shape), dtype=torch.float32, device=device)
        ptrs = []
        for key in spec.tensors:
            t = env[key]
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("generated kernels take contiguous fp32 HIP tensors (no CPU fallback)")
            ptrs.append(t.data_ptr())
        for j, ti in enumerate(spec.pregather):       # table[col[e]] in CSR order, cached on the CSR per (storage, version)
            t = env[spec.tensors[ti]]
            cache = csr.__dict__.setdefault("_gen_edge_cache", {})
            tag = (spec.tensors[ti], t.data_ptr(), t._version, t.numel())
            hit = cache.get(spec.tensors[ti])
            if hit is None or hit[0] != tag:
                with torch.cuda.device(device):
                    g = torch.empty(csr.num_edges, dtype=torch.float32, device=device)
                    _C.check(_C.lib.stg_edge_gather_f32(g.data_ptr(), t.data_ptr(), csr.column_indices.data_ptr(), csr.num_edges,
                                                        ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
                hit = cache[spec.tensors[ti]] = (tag, g, t)          # (t kept alive: its address cannot be handed to another tensor)
            ptrs.append(hit[1].data_ptr())
        eids = csr.eids if spec.uses_eids else None
        ptrs += [csr.row_offset.data_ptr(), csr.column_indices.data_ptr(), eids.data_ptr() if eids is not None else 0,
                 (csr.node_ids_if_ready.data_ptr() if (use_nid and csr.node_ids_if_ready is not None) else 0)]
        G = spec.lanes_per_row
        rows_per_block = 256 // G
        grid = (N + rows_per_block - 1) // rows_per_block
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        ints = (ctypes.c_int32 * 1)(N)
        fn = self.module.function(device, spec.name)
        with torch.cuda.device(device):
            _C.check(_C.lib.stg_jit_launch(fn, grid, 256, arr, len(ptrs), ints, 1,
                                           ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))

    def _bind_inputs(self, n_feats: dict, e_feats: dict) -> dict:
        env = {}
        for kind, name in self._inputs:
            t = self._params[name] if kind == "p" else (e_feats if kind == "e" else n_feats)[name]
            if not t.is_cuda:
                raise RuntimeError("stgraph_amd has no CPU fallback: vertex-function inputs must be HIP tensors")
            env[("leaf", kind, name)] = t.detach().contiguous().float()
        return env

    def forward(self, graph, n_feats, e_feats):
        env = self._bind_inputs(n_feats, e_feats)
        device = next(iter(env.values())).device
        N, E = graph.get_num_nodes(), graph.csr("fwd").column_indices.shape[0]
        for spec in self.fwd_kernels:
            self._launch(spec, graph, env, N, E, device)
        outs = tuple(env[self._fwd_out_key[id(r)]] if id(r) in self._fwd_out_key else env[self._leaf_key(r)]
                     for r in self.rets)
        saved = {"env": {k: v for k, v in env.items() if k[0] == "leaf"}}
        saved["env"].update({("leaf", "n", self._saved_name[id(x)]): env[self._fwd_out_key[id(x)]]
                             for x in self.saved_nodes})
        return outs, saved

    def backward(self, graph, saved, grads):
        env = dict(saved["env"])
        for k, g in enumerate(grads):
            env[("leaf", "n" if self.rets[k].val_type != ValType.EDGE else "e", f"__gout{k}")] = g.contiguous().float()
        device = grads[0].device
        N, E = graph.get_num_nodes(), graph.csr("fwd").column_indices.shape[0]
        for spec in self.bwd_kernels:
            self._launch(spec, graph, env, N, E, device)
        result = {}
        for leaf, root in self.grad_roots:
            key = (_kind(leaf), leaf.name)
            if id(root) in self._bwd_out_key:
                g = env[self._bwd_out_key[id(root)]]
            elif root.op == "Leaf":
                g = env[self._leaf_key(root)]
            else:
                raise RuntimeError("gradient root was not emitted")
            rows = g.shape[0]
            if leaf.val_type == ValType.PARAM:          # contributions of every row / edge, then down to the parameter's shape
                shp = tuple(self._input_shape[key])
                g = _fit(g.sum(0), shp) if shp else g.sum()
                result[key] = g if key not in result else result[key] + g
                continue
            g = _fit(g, (rows,) + tuple(self._input_shape[key]))
            result[key] = g if key not in result else result[key] + g
        return result

    def describe(self) -> str:
        lines = [f"generated plan {self.uid}: {len(self.fwd_kernels)} forward + {len(self.bwd_kernels)} backward kernel(s)"]
        for s in self.fwd_kernels + self.bwd_kernels:
            lines.append(f"  {s.name}: rows={'src' if s.row_type == ValType.SRC else 'dst'} loop={s.has_loop} "
                         f"lanes={s.lanes_per_row} features={_numel(s.full)} tensors={len(s.tensors)} eids={s.uses_eids}")
        return "\n".join(lines)
