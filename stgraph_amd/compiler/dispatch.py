"""Map a traced vertex function (GIR) onto the hand-written gfx950 kernels.

The reference fuses the GIR into execution units with an FSM (passes/fusion.py),
differentiates it (autodiff.py) and emits one CUDA kernel per unit.  For the three
vertex functions on the hot path the resulting units are fixed (SURVEY.md
Appendix B); this module recognises those GIRs structurally and returns a *plan*
that launches the corresponding hand-written HIP kernels, forward and backward.
Multiplication order matters for bit-level parity, so patterns match the exact
association the reference would emit (operands of one ``Mul`` may be swapped --
that does not change the rounding -- but ``(a*b)*c`` is not ``a*(b*c)``).

Any other vertex function goes to the code generator (``codegen.py``: GIR -> HIP source ->
hiprtc), which plays the role of the reference's fusion + autodiff + template pipeline.
"""
from __future__ import annotations

from dataclasses import dataclass

from .. import kernels
from .gir import Node, ValType


def _leaf(n: Node, vt: ValType) -> bool:
    return n.op == "Leaf" and n.val_type == vt


def _is_scalar(n: Node) -> bool:
    """[1]-shaped (or [H,1]-free) feature: one value per vertex/edge."""
    return all(s == 1 for s in n.shape)


def _split_mul(n: Node, pred_a, pred_b):
    """If n == Mul(x, y) with pred_a(x) and pred_b(y) in either operand order, return (x, y)."""
    if n.op != "Mul" or len(n.args) != 2:
        return None
    x, y = n.args
    if pred_a(x) and pred_b(y):
        return x, y
    if pred_a(y) and pred_b(x):
        return y, x
    return None


# ----------------------------------------------------------------------------------- plans
@dataclass
class GcnPlan:
    """out = D:norm * AggSum((S:norm * S:x) [* E:w])   --   gcn_conv.py:162-182."""

    x: str                  # n_feats key of the gathered feature
    norm_src: str           # n_feats key read at the neighbour
    norm_dst: str           # n_feats key read at the centre
    ew: str | None          # e_feats key
    name = "gcn_agg"

    def input_names(self):
        return [("n", self.x), ("n", self.norm_src), ("n", self.norm_dst)] + ([("e", self.ew)] if self.ew else [])

    def differentiable(self):
        return [("n", self.x)]

    def forward(self, graph, n_feats, e_feats):
        x = n_feats[self.x]
        F = int(x[0].numel()) if x.shape[0] else 1
        ew = e_feats[self.ew] if self.ew else None
        out = kernels.gcn_agg(x, n_feats[self.norm_dst], n_feats[self.norm_src], graph.csr("fwd"),
                              ew=ew, use_node_ids=(kernels.rows_by_node_ids(graph.graph_type())),
                              f_active=kernels.active_columns(F))
        saved = {"norm_src": n_feats[self.norm_src], "norm_dst": n_feats[self.norm_dst], "ew": ew}
        return (out,), saved

    def backward(self, graph, saved, grads):
        (g,) = grads
        F = int(g[0].numel()) if g.shape[0] else 1
        # acc += grad[dst]*norm_cen[dst]; grad_h[src] = acc*norm_inb[src]
        gx = kernels.gcn_agg(g, saved["norm_src"], saved["norm_dst"], graph.csr("bwd"), ew=saved["ew"],
                             use_node_ids=(kernels.rows_by_node_ids(graph.graph_type())),
                             f_active=kernels.active_columns(F))
        return {("n", self.x): gx}


@dataclass
class GatPlan:
    """out = AggSum((c / AggSum(c)) * S:feat),  c = exp(leaky_relu(emb - emb)),  emb = S:el + D:er
    --  gat_conv.py:48-56 (``max(embs)`` over a one-element list returns ``emb``: SURVEY D2)."""

    el: str
    er: str
    feat: str
    slope: float
    name = "gat"

    def input_names(self):
        return [("n", self.el), ("n", self.er), ("n", self.feat)]

    def differentiable(self):
        return [("n", self.el), ("n", self.er), ("n", self.feat)]

    def forward(self, graph, n_feats, e_feats):
        el, er, feat = n_feats[self.el], n_feats[self.er], n_feats[self.feat]
        use_nid = kernels.rows_by_node_ids(graph.graph_type())
        out, A, S = kernels.gat_fwd(el, er, feat, graph.csr("fwd"), self.slope, use_nid, ones_shortcut=True)
        return (out,), {"A": A, "S": S, "out": out, "el": el, "er": er, "feat": feat}

    def backward(self, graph, saved, grads):
        (g,) = grads
        use_nid = kernels.rows_by_node_ids(graph.graph_type())
        gf, gel, ger = kernels.gat_bwd(saved["A"], saved["S"], saved["out"], g, saved["el"], saved["er"],
                                       saved["feat"], graph.csr("fwd"), graph.csr("bwd"), self.slope, use_nid)
        # el is read at the neighbour (grad_el), er at the centre (grad_er); when both keys name the
        # same tensor autograd adds the two contributions, as the reference's grad accumulation does
        return {("n", self.feat): gf, ("n", self.el): gel.view_as(saved["el"]), ("n", self.er): ger.view_as(saved["er"])}


# --------------------------------------------------------------------------------- matching
def _orient(norm_src: Node, x: Node):
    """F == 1: both factors are [1]-shaped; the differentiable one is the gathered feature."""
    if _is_scalar(x) and norm_src.requires_grad and not x.requires_grad:
        return x, norm_src
    return norm_src, x


def _match_gcn(ret: Node):
    top = _split_mul(ret, lambda n: n.op == "AggSum", lambda n: _leaf(n, ValType.DEST) and _is_scalar(n))
    if top is None:
        return None
    agg, norm_dst = top
    inner = agg.args[0]
    src_scalar = lambda n: _leaf(n, ValType.SRC) and _is_scalar(n)          # noqa: E731
    src_vec = lambda n: _leaf(n, ValType.SRC)                               # noqa: E731
    # no edge weight:  Mul(S:x, S:norm)
    m = _split_mul(inner, src_scalar, src_vec)
    if m is not None and m[0] is not m[1]:
        norm_src, x = _orient(*m)
        return GcnPlan(x.name, norm_src.name, norm_dst.name, None), [x, norm_src, norm_dst]
    # edge weight:  Mul(Mul(S:norm, S:x), E:w)
    m = _split_mul(inner, lambda n: n.op == "Mul", lambda n: _leaf(n, ValType.EDGE) and _is_scalar(n))
    if m is not None:
        prod, w = m
        m2 = _split_mul(prod, src_scalar, src_vec)
        if m2 is not None and m2[0] is not m2[1]:
            norm_src, x = _orient(*m2)
            return GcnPlan(x.name, norm_src.name, norm_dst.name, w.name), [x, norm_src, norm_dst, w]
    return None


def _match_gat(ret: Node):
    if ret.op != "AggSum":
        return None
    m = _split_mul(ret.args[0], lambda n: n.op == "TrueDiv", lambda n: _leaf(n, ValType.SRC))
    if m is None:
        return None
    div, feat = m
    c, s = div.args
    if s.op != "AggSum" or s.args[0] is not c or c.op != "Exp":
        return None
    lr = c.args[0]
    if lr.op != "LeakyRelu":
        return None
    sub = lr.args[0]
    if sub.op != "Sub" or sub.args[0] is not sub.args[1]:
        return None
    emb = sub.args[0]
    m = None
    if emb.op == "Add":
        a, b = emb.args
        if _leaf(a, ValType.SRC) and _leaf(b, ValType.DEST):
            m = (a, b)
        elif _leaf(b, ValType.SRC) and _leaf(a, ValType.DEST):
            m = (b, a)
    if m is None:
        return None
    el, er = m
    if len(feat.shape) != 2 or el.shape != (feat.shape[0], 1) or er.shape != el.shape:
        return None
    slope = dict(lr.params).get("negative_slope", 0.01)
    return GatPlan(el.name, er.name, feat.name, float(slope)), [el, er, feat]


_FORCE_GENERATED = False


def set_force_generated(enabled: bool) -> None:
    """Testing aid: route EVERY vertex function (also the GCN / GAT ones) through the code generator."""
    global _FORCE_GENERATED
    _FORCE_GENERATED = bool(enabled)


def make_plan(rets: list, program) -> object:
    """The kernel plan of a traced vertex function: the hand-written units where the GIR is one of theirs,
    generated kernels (``codegen.GenericPlan``) for everything else."""
    if len(rets) == 1 and not _FORCE_GENERATED:
        for matcher in (_match_gcn, _match_gat):
            hit = matcher(rets[0])
            if hit is not None:
                plan, leaves = hit
                diff = {(("e" if l.val_type == ValType.EDGE else "n"), l.name) for l in leaves if l.requires_grad}
                if diff - set(plan.differentiable()):
                    break          # e.g. d/d(norm): not one of the hand-written backward units -> generate
                return plan
    from .codegen import GenericPlan
    try:
        return GenericPlan(rets, program)
    except NotImplementedError as e:
        raise NotImplementedError(
            f"this vertex function cannot be compiled for the MI355X yet: {e}\nTraced program:\n" + str(program) +
            "\nreturn: " + ", ".join(r.key for r in rets)) from e
