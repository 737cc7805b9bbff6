"""ctypes front end of oracle/stg_pcsr_oracle.c -- the reference's PCSR store restated on the CPU.

TEST INFRASTRUCTURE ONLY (see stg_pcsr_oracle.c).  ``OraclePCSR`` has the methods of the reference's
pybind class (pcsr.cu:916-939): ``edge_update_list``, ``label_edges``, ``build_csr``,
``build_reverse_csr``, ``get_edges``, ``in_degrees`` / ``out_degrees`` / ``edge_count``, copy.
``OraclePCSRGraph`` replays the ``PCSRGraph`` protocol (graph/dynamic/pcsr/pcsr_graph.py:46-166) on it.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import stg_oracle as _so

_vp = ctypes.c_void_p
_READY = False


def _lib():
    global _READY
    lib = _so._lib()
    if not _READY:
        lib.orc_pcsr_new.restype = _vp
        lib.orc_pcsr_copy.restype = _vp
        for f in ("orc_pcsr_build_csr", "orc_pcsr_build_reverse_csr", "orc_pcsr_edge_count", "orc_pcsr_capacity",
                  "orc_pcsr_get_edges"):
            getattr(lib, f).restype = ctypes.c_int64
        for f in ("orc_pcsr_free", "orc_pcsr_edge_update_list", "orc_pcsr_label_edges", "orc_pcsr_degrees",
                  "orc_pcsr_state"):
            getattr(lib, f).restype = None
        _READY = True
    return lib


def _ptr(a):
    return _vp(a.ctypes.data)


class OraclePCSR:
    def __init__(self, num_nodes: int, max_edges: int, _h=None):
        self.n, self.max_edges = int(num_nodes), int(max_edges)
        self._h = _h if _h is not None else _vp(_lib().orc_pcsr_new(ctypes.c_uint32(self.n),
                                                                   ctypes.c_uint32(self.max_edges)))

    def __del__(self):
        try:
            _lib().orc_pcsr_free(self._h)
        except Exception:
            pass

    def copy(self) -> "OraclePCSR":
        return OraclePCSR(self.n, self.max_edges, _vp(_lib().orc_pcsr_copy(self._h)))

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()

    def edge_update_list(self, edges, is_delete=False, is_reverse_edge=False) -> None:
        e = np.asarray(edges, np.uint32).reshape(-1, 2)
        a, b = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
        _lib().orc_pcsr_edge_update_list(self._h, _ptr(a), _ptr(b), ctypes.c_int64(len(a)), int(is_delete),
                                         int(is_reverse_edge))

    def label_edges(self) -> None:
        _lib().orc_pcsr_label_edges(self._h)

    @property
    def edge_count(self) -> int:
        return int(_lib().orc_pcsr_edge_count(self._h))

    def _build(self, fn) -> dict:
        E = self.edge_count
        ro, nid = np.empty(self.n + 1, np.uint32), np.empty(self.n, np.uint32)
        col, eid = np.empty(E, np.uint32), np.empty(E, np.uint32)
        fn(self._h, _ptr(ro), _ptr(col), _ptr(eid), _ptr(nid))
        return {"row_offset": ro, "column_indices": col, "eids": eid, "node_ids": nid}

    def build_csr(self) -> dict:
        return self._build(_lib().orc_pcsr_build_csr)

    def build_reverse_csr(self) -> dict:
        return self._build(_lib().orc_pcsr_build_reverse_csr)

    def degrees(self):
        i, o = np.empty(self.n, np.uint32), np.empty(self.n, np.uint32)
        _lib().orc_pcsr_degrees(self._h, _ptr(i), _ptr(o))
        return i, o

    @property
    def in_degrees(self):
        return self.degrees()[0]

    @property
    def out_degrees(self):
        return self.degrees()[1]

    def state(self) -> dict:
        cap = int(_lib().orc_pcsr_capacity(self._h))
        dims, items, nodes = np.empty(3, np.int32), np.empty((cap, 2), np.uint32), np.empty((self.n, 4), np.uint32)
        _lib().orc_pcsr_state(self._h, _ptr(dims), _ptr(items), _ptr(nodes))
        return {"N": int(dims[0]), "H": int(dims[1]), "logN": int(dims[2]), "items": items, "nodes": nodes}

    def get_edges(self) -> np.ndarray:
        out = np.empty((max(self.edge_count, 1) + 8, 3), np.uint32)
        k = int(_lib().orc_pcsr_get_edges(self._h, _ptr(out)))
        return out[:k]
