"""ctypes front end of oracle/_ref/libref_shim.so: the reference's OWN compiled CSR / PCSR classes
(prebuilt ``csr.so`` / ``pcsr.so`` under /root/reference, see ref_shim.cpp).

TEST INFRASTRUCTURE ONLY, and build-container only: ``available()`` is False wherever
/root/reference is absent (the GPU box), and every user skips then.  Used to pin oracle/*.c against
the reference's real native code and to generate tests/golden/*.npz.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_shim.so")
_LIB = None
_vp = ctypes.c_void_p


def _load():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(_PATH)
        lib.ref_csr_ctor.restype = ctypes.c_int
        lib.ref_pcsr_new.restype = _vp
        lib.ref_pcsr_copy.restype = _vp
        lib.ref_pcsr_build.restype = ctypes.c_int64
        lib.ref_pcsr_edge_count.restype = ctypes.c_int64
        lib.ref_pcsr_capacity.restype = ctypes.c_int64
        lib.ref_pcsr_get_edges.restype = ctypes.c_int64
        for f in ("ref_pcsr_update", "ref_pcsr_label", "ref_pcsr_degrees", "ref_pcsr_state"):
            getattr(lib, f).restype = None
        _LIB = lib
    return _LIB


def available() -> bool:
    if not os.path.exists(_PATH):
        return False
    try:
        _load()
        return True
    except OSError:
        return False


def _ptr(a):
    return _vp(a.ctypes.data)


def csr_ctor(a, b, eid, edge_weight, num_nodes: int, is_edge_reverse: bool = False) -> dict:
    """The reference's ``CSR(edge_list, edge_weight, num_nodes, is_edge_reverse)`` (csr.cu:68-157), run from csr.so."""
    a, b, eid = (np.ascontiguousarray(x, np.int32) for x in (a, b, eid))
    E, N = len(a), int(num_nodes)
    w = np.ascontiguousarray(edge_weight if edge_weight is not None else np.ones(E), np.float32)
    out = {"row_offset": np.empty(N + 1, np.int32), "column_indices": np.empty(E, np.int32),
           "eids": np.empty(E, np.int32), "node_ids": np.empty(N, np.int32), "in_degrees": np.empty(N, np.int32),
           "out_degrees": np.empty(N, np.int32), "weighted_out_degrees": np.empty(N, np.float32)}
    rc = _load().ref_csr_ctor(_ptr(a), _ptr(b), _ptr(eid), _ptr(w), ctypes.c_int64(E), ctypes.c_int32(N),
                              int(bool(is_edge_reverse)), *[_ptr(v) for v in out.values()])
    if rc:
        raise RuntimeError("reference CSR produced arrays of unexpected size")
    return out


class RefPCSR:
    """The reference's ``PCSR`` object (pcsr.cu:273-318), driven through its exported C++ methods."""

    def __init__(self, num_nodes: int, max_edges: int, _h=None):
        self.n, self.max_edges = int(num_nodes), int(max_edges)
        self._h = _h if _h is not None else _vp(_load().ref_pcsr_new(ctypes.c_uint32(self.n),
                                                                    ctypes.c_uint32(self.max_edges)))

    def copy(self) -> "RefPCSR":
        return RefPCSR(self.n, self.max_edges, _vp(_load().ref_pcsr_copy(self._h)))

    def edge_update_list(self, edges, is_delete=False, is_reverse_edge=False) -> None:
        e = np.asarray(edges, np.uint32).reshape(-1, 2)
        a, b = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
        _load().ref_pcsr_update(self._h, _ptr(a), _ptr(b), ctypes.c_int64(len(a)), int(is_delete), int(is_reverse_edge))

    def label_edges(self) -> None:
        _load().ref_pcsr_label(self._h)

    @property
    def edge_count(self) -> int:
        return int(_load().ref_pcsr_edge_count(self._h))

    def _build(self, reverse: bool) -> dict:
        E = self.edge_count
        if E > self.max_edges:
            raise RuntimeError("edge_count exceeds max_num_edges (the reference would overflow its arrays)")
        ro, nid = np.empty(self.n + 1, np.uint32), np.empty(self.n, np.uint32)
        col, eid = np.empty(E, np.uint32), np.empty(E, np.uint32)
        got = _load().ref_pcsr_build(self._h, int(reverse), _ptr(ro), _ptr(col), _ptr(eid), _ptr(nid))
        assert got == E
        return {"row_offset": ro, "column_indices": col, "eids": eid, "node_ids": nid}

    def build_csr(self) -> dict:
        return self._build(False)

    def build_reverse_csr(self) -> dict:
        return self._build(True)

    def degrees(self):
        i, o = np.empty(self.n, np.uint32), np.empty(self.n, np.uint32)
        _load().ref_pcsr_degrees(self._h, _ptr(i), _ptr(o))
        return i, o

    def state(self) -> dict:
        cap = int(_load().ref_pcsr_capacity(self._h))
        dims, items, nodes = np.empty(3, np.int32), np.empty((cap, 2), np.uint32), np.empty((self.n, 4), np.uint32)
        _load().ref_pcsr_state(self._h, _ptr(dims), _ptr(items), _ptr(nodes))
        return {"N": int(dims[0]), "H": int(dims[1]), "logN": int(dims[2]), "items": items, "nodes": nodes}

    def get_edges(self) -> np.ndarray:
        out = np.empty((max(self.edge_count, 1), 3), np.uint32)
        k = int(_load().ref_pcsr_get_edges(self._h, _ptr(out)))
        return out[:k]
