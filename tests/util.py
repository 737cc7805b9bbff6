"""Shared helpers for the test-suite (oracle access, golden fixtures, random graphs)."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

GCN_WIDTHS = (1, 4, 7, 16, 32, 64, 100, 128, 256, 300)
GAT_SHAPES = ((1, 7), (2, 4), (8, 8), (8, 64))


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


def random_graph(seed: int, n: int, e: int, duplicates: bool = False, hub: bool = True):
    """Seeded directed multigraph-free (unless ``duplicates``) edge arrays in random order."""
    rng = np.random.default_rng(seed)
    if duplicates:
        src = rng.integers(0, n, e, dtype=np.int64)
        dst = rng.integers(0, n, e, dtype=np.int64)
    else:
        total = n * n
        if e > total:
            raise ValueError("too many edges")
        if total <= 4 * e or total < (1 << 22):
            keys = rng.choice(total, size=e, replace=False)
        else:
            keys = np.unique(rng.integers(0, total, int(e * 1.2) + 16, dtype=np.int64))
            rng.shuffle(keys)
            keys = keys[:e]
            assert keys.shape[0] == e
        src, dst = keys // n, keys % n
    if hub and n > 4 and e > 8:           # a high-degree destination to exercise the ragged path
        k = max(1, e // 8)
        dst[:k] = 1
        if not duplicates:
            pair = np.unique(np.stack([src, dst], 1), axis=0)
            rng.shuffle(pair)
            src, dst = pair[:, 0], pair[:, 1]
    return src.astype(np.int32), dst.astype(np.int32)


def gcn_norm(in_degrees: np.ndarray) -> np.ndarray:
    """norm = in_deg^-0.5, inf -> 0  (benchmarking/gcn/seastar/train.py:53-57)."""
    with np.errstate(divide="ignore"):
        norm = np.power(in_degrees.astype(np.float32), np.float32(-0.5))
    norm[np.isinf(norm)] = 0
    return norm.astype(np.float32).reshape(-1, 1)


# ----------------------------------------------------------------------- vertex-function reference
def trace_vertex_function(fn, n_feats: dict, e_feats: dict):
    """Run the product tracer on ``fn``; returns (ret nodes, program)."""
    from stgraph_amd.compiler.gir import Program, ValType
    from stgraph_amd.compiler.node import CentralNode
    from stgraph_amd.compiler.val import Val
    prog, cen = Program(), CentralNode()
    for k, v in n_feats.items():
        setattr(cen, k, Val.leaf(prog, k, ValType.DEST, v))
        for nb in cen.innbs:
            setattr(nb, k, Val.leaf(prog, k, ValType.SRC, v))
    for k, v in e_feats.items():
        for e in cen.inedges:
            setattr(e, k, Val.leaf(prog, k, ValType.EDGE, v))
    r = fn(cen)
    return [x.node for x in (r if isinstance(r, (tuple, list)) else [r])], prog


def eval_vertex_function(fn, src_by_eid, dst_by_eid, num_nodes, n_feats: dict, e_feats: dict, p_feats: dict | None = None):
    """Plain-torch reference of a vertex function: every traced node is evaluated with torch ops on
    whole tensors (per-vertex values [N, ...], per-edge values [E, ...] in eid order, ``sum([...])`` =
    ``index_add_`` over the destination).  Differentiable through torch autograd; works in any dtype."""
    import torch
    from stgraph_amd.compiler.gir import ValType
    rets, _ = trace_vertex_function(fn, n_feats, e_feats)
    memo = {}

    def to_edge(n, v):
        if n.op == "Const" or n.val_type == ValType.PARAM or n.val_type == ValType.EDGE:
            return v
        return v[src_by_eid] if n.val_type == ValType.SRC else v[dst_by_eid]

    def ev(n):
        if id(n) in memo:
            return memo[id(n)]
        if n.op == "Const":
            r = n.value
        elif n.op == "Leaf" and n.val_type == ValType.PARAM:
            r = (p_feats or {}).get(n.name, n.value)         # a module parameter read inside the vertex function
        elif n.op == "Leaf":
            r = (e_feats if n.val_type == ValType.EDGE else n_feats)[n.name]
        elif n.op == "AggSum":
            a = to_edge(n.args[0], ev(n.args[0]))
            r = torch.zeros((num_nodes,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device).index_add_(0, dst_by_eid, a)
        elif n.op == "AggMax":
            a = to_edge(n.args[0], ev(n.args[0]))
            idx = dst_by_eid.view((-1,) + (1,) * (a.dim() - 1)).expand_as(a)
            r = torch.full((num_nodes,) + tuple(a.shape[1:]), float("-inf"), dtype=a.dtype, device=a.device)
            r = r.scatter_reduce(0, idx, a, reduce="amax", include_self=True)
        elif n.op == "Sum":
            v = ev(n.args[0])
            pr = dict(n.params)
            r = v.sum(dim=tuple(d + 1 for d in pr["dims"]), keepdim=pr["keepdim"])
        elif n.op == "View":
            v = ev(n.args[0])
            r = v.reshape((v.shape[0],) + tuple(n.shape))
        else:
            vals = [ev(a) for a in n.args]
            if n.val_type == ValType.EDGE:
                vals = [to_edge(a, v) for a, v in zip(n.args, vals)]
            if n.op == "Mul":
                r = vals[0] * vals[1]
            elif n.op == "Add":
                r = vals[0] + vals[1]
            elif n.op == "Sub":
                r = vals[0] - vals[1]
            elif n.op == "TrueDiv":
                r = vals[0] / vals[1]
            elif n.op == "Exp":
                r = torch.exp(vals[0])
            elif n.op == "Relu":
                r = torch.relu(vals[0])
            elif n.op == "LeakyRelu":
                r = torch.nn.functional.leaky_relu(vals[0], dict(n.params)["negative_slope"])
            else:
                raise NotImplementedError(n.op)
        memo[id(n)] = r
        return r
    return [ev(r) for r in rets]


def edges_by_eid(csr):
    """(src, dst) of every edge indexed by eid, from a forward (dst-major) CSR."""
    import torch
    ro = csr.row_offset.long()
    n = ro.shape[0] - 1
    dst_pos = torch.repeat_interleave(torch.arange(n, device=ro.device), ro[1:] - ro[:-1])
    src_pos = csr.column_indices.long()
    eid = csr.eids.long()
    src, dst = torch.empty_like(src_pos), torch.empty_like(dst_pos)
    src[eid], dst[eid] = src_pos, dst_pos
    return src, dst
