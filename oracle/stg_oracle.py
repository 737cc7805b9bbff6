"""ctypes/numpy front end of the CPU oracle (oracle/stg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (stgraph_amd/) never imports
this module; it fails loudly when its HIP library is missing instead.

Each wrapper restates a reference entry point (paths relative to
/root/reference/stgraph):

* ``csr_ctor``            -- graph/static/csr.cu:68-157 (``CSR::CSR``)
* ``prepare_edge_lists``  -- graph/static/static_graph.py:65-78
* ``build_graph``         -- graph/static/static_graph.py:40-62 (fwd + bwd CSR)
* ``gcn_agg``             -- emitted GCN units, SURVEY.md Appendix B.1/B.2
* ``gat_k0/gat_k1/gat_bwd`` -- emitted GAT units, SURVEY.md Appendix B.3
* ``ref_active_columns``  -- compiler/execution_unit.py:92-116 (defect D1)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: dict[bool, ctypes.CDLL] = {}

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (make -C oracle)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _lib(omp: bool = False) -> ctypes.CDLL:
    if omp not in _LIBS:
        name = "libstg_oracle_omp.so" if omp else "libstg_oracle.so"
        path = os.path.join(_HERE, name)
        srcs = [os.path.join(_HERE, f) for f in ("stg_oracle.c", "stg_pcsr_oracle.c", "stg_gpma_oracle.c")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
            build()
        lib = ctypes.CDLL(path)
        lib.orc_csr_ctor.restype = ctypes.c_int
        lib.orc_prepare_edge_lists.restype = ctypes.c_int
        lib.orc_ref_active_columns.restype = ctypes.c_int
        lib.orc_gcn_agg.restype = None
        lib.orc_gat_k0.restype = None
        lib.orc_gat_k1.restype = None
        lib.orc_gat_bwd.restype = None
        _LIBS[omp] = lib
    return _LIBS[omp]


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray | None, typ):
    if a is None:
        return typ()
    return a.ctypes.data_as(typ)


def ref_active_columns(feat_size: int) -> int:
    return int(_lib().orc_ref_active_columns(ctypes.c_int(int(feat_size))))


@dataclass
class OracleCSR:
    """Host arrays of one reference ``CSR`` object (csr.cu:35-59)."""

    row_offset: np.ndarray
    column_indices: np.ndarray
    eids: np.ndarray
    node_ids: np.ndarray
    in_degrees: np.ndarray
    out_degrees: np.ndarray
    weighted_out_degrees: np.ndarray


def csr_ctor(a, b, eid, edge_weight, num_nodes: int, is_edge_reverse: bool = False) -> OracleCSR:
    a, b, eid = _i32(a), _i32(b), _i32(eid)
    E, N = int(a.shape[0]), int(num_nodes)
    ew = None if edge_weight is None else _f32(edge_weight)
    out = OracleCSR(
        np.empty(N + 1, np.int32), np.empty(E, np.int32), np.empty(E, np.int32),
        np.empty(N, np.int32), np.empty(N, np.int32), np.empty(N, np.int32),
        np.empty(N, np.float32),
    )
    rc = _lib().orc_csr_ctor(
        _p(a, _i32p), _p(b, _i32p), _p(eid, _i32p), _p(ew, _f32p),
        ctypes.c_int64(E), ctypes.c_int(N), ctypes.c_int(int(bool(is_edge_reverse))),
        _p(out.row_offset, _i32p), _p(out.column_indices, _i32p), _p(out.eids, _i32p),
        _p(out.node_ids, _i32p), _p(out.in_degrees, _i32p), _p(out.out_degrees, _i32p),
        _p(out.weighted_out_degrees, _f32p),
    )
    if rc != 0:
        raise ValueError(f"orc_csr_ctor failed ({rc}): vertex id outside [0, num_nodes)")
    return out


def prepare_edge_lists(src, dst):
    """Return (perm_fwd, fwd(src,dst,eid), bwd(src,dst,eid)) -- static_graph.py:65-78."""
    src, dst = _i32(src), _i32(dst)
    E = int(src.shape[0])
    perm = np.empty(E, np.int64)
    f = [np.empty(E, np.int32) for _ in range(3)]
    b = [np.empty(E, np.int32) for _ in range(3)]
    rc = _lib().orc_prepare_edge_lists(
        _p(src, _i32p), _p(dst, _i32p), ctypes.c_int64(E), _p(perm, _i64p),
        *[_p(x, _i32p) for x in f], *[_p(x, _i32p) for x in b])
    if rc != 0:
        raise MemoryError("orc_prepare_edge_lists")
    return perm, tuple(f), tuple(b)


@dataclass
class OracleGraph:
    """Forward (dst-major) + backward (src-major) CSR, as ``StaticGraph`` builds them."""

    num_nodes: int
    num_edges: int
    perm_fwd: np.ndarray       # caller position of the edge that became eid j
    fwd: OracleCSR
    bwd: OracleCSR

    def in_degrees(self) -> np.ndarray:       # static_graph.py:115-117
        return self.fwd.out_degrees.astype(np.int32)

    def out_degrees(self) -> np.ndarray:      # static_graph.py:119-121
        return self.fwd.in_degrees.astype(np.int32)


def build_graph(src, dst, num_nodes: int, edge_weights=None) -> OracleGraph:
    """``StaticGraph.__init__`` (static_graph.py:40-62) on (src, dst) arrays.

    ``edge_weights`` is indexed by eid (= position in (dst,src)-sorted order),
    exactly as the reference indexes it (csr.cu:126).
    """
    perm, f, b = prepare_edge_lists(src, dst)
    fwd = csr_ctor(f[0], f[1], f[2], edge_weights, num_nodes, is_edge_reverse=True)
    bwd = csr_ctor(b[0], b[1], b[2], edge_weights, num_nodes, is_edge_reverse=False)
    return OracleGraph(int(num_nodes), int(len(perm)), perm, fwd, bwd)


def gcn_agg(x, norm_row, norm_col, csr: OracleCSR, ew=None, use_node_ids: bool = False,
            f_active: int | None = None, omp: bool = False) -> np.ndarray:
    x = _f32(x)
    N = csr.row_offset.shape[0] - 1
    F = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    assert x.shape[0] == N
    norm_row, norm_col = _f32(norm_row).reshape(-1), _f32(norm_col).reshape(-1)
    assert norm_row.shape[0] == N and norm_col.shape[0] == N
    ew_ = None if ew is None else _f32(ew).reshape(-1)
    out = np.zeros((N, F), np.float32)
    fa = F if f_active is None else int(f_active)
    _lib(omp).orc_gcn_agg(
        _p(x, _f32p), _p(norm_row, _f32p), _p(norm_col, _f32p), _p(ew_, _f32p), _p(out, _f32p),
        _p(csr.row_offset, _i32p), _p(csr.column_indices, _i32p), _p(csr.eids, _i32p),
        _p(csr.node_ids if use_node_ids else None, _i32p),
        ctypes.c_int(N), ctypes.c_int(F), ctypes.c_int(fa))
    return out.reshape(x.shape)


def gat_k0(el, er, csr: OracleCSR, num_edges: int, slope: float = 0.2, use_node_ids=False,
           h_active: int | None = None, omp: bool = False):
    el, er = _f32(el), _f32(er)
    N = csr.row_offset.shape[0] - 1
    H = int(np.prod(el.shape[1:]))
    A = np.zeros((num_edges, H, 1), np.float32)
    S = np.zeros((N, H, 1), np.float32)
    _lib(omp).orc_gat_k0(
        _p(el, _f32p), _p(er, _f32p), _p(A, _f32p), _p(S, _f32p),
        _p(csr.row_offset, _i32p), _p(csr.column_indices, _i32p), _p(csr.eids, _i32p),
        _p(csr.node_ids if use_node_ids else None, _i32p),
        ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(H if h_active is None else h_active),
        ctypes.c_float(slope))
    return A, S


def gat_k1(A, S, feat, csr: OracleCSR, use_node_ids=False, hd_active: int | None = None,
           omp: bool = False):
    A, S, feat = _f32(A), _f32(S), _f32(feat)
    N, H, D = feat.shape
    out = np.zeros((N, H, D), np.float32)
    _lib(omp).orc_gat_k1(
        _p(A, _f32p), _p(S, _f32p), _p(feat, _f32p), _p(out, _f32p),
        _p(csr.row_offset, _i32p), _p(csr.column_indices, _i32p), _p(csr.eids, _i32p),
        _p(csr.node_ids if use_node_ids else None, _i32p),
        ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(D),
        ctypes.c_int(H * D if hd_active is None else hd_active))
    return out


def gat_bwd(A, S, out, g, el, er, feat, bwd_csr: OracleCSR, slope: float = 0.2,
            use_node_ids=False, hd_active: int | None = None):
    A, S, out, g, el, er, feat = map(_f32, (A, S, out, g, el, er, feat))
    N, H, D = feat.shape
    grad_feat = np.zeros((N, H, D), np.float32)
    grad_el = np.zeros((N, H, 1), np.float32)
    grad_er = np.zeros((N, H, 1), np.float32)
    _lib(False).orc_gat_bwd(
        _p(A, _f32p), _p(S, _f32p), _p(out, _f32p), _p(g, _f32p), _p(el, _f32p), _p(er, _f32p),
        _p(feat, _f32p), _p(grad_feat, _f32p), _p(grad_el, _f32p), _p(grad_er, _f32p),
        _p(bwd_csr.row_offset, _i32p), _p(bwd_csr.column_indices, _i32p), _p(bwd_csr.eids, _i32p),
        _p(bwd_csr.node_ids if use_node_ids else None, _i32p),
        ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(D),
        ctypes.c_int(H * D if hd_active is None else hd_active), ctypes.c_float(slope))
    return grad_feat, grad_el, grad_er
