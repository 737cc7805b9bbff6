"""ctypes front end of oracle/stg_gpma_oracle.c -- what the reference's GPMA graph hands to its kernels,
restated on the CPU.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no reference binary, see the C header).

``OracleGPMA`` holds a gapped image of an edge set (holes + lazily deleted entries at seeded positions)
and exposes the reference's pipeline on it: ``label_edges`` (gpma.cu:1121-1163), ``build_backward_csr``
(:1165-1231), ``node_ids`` (:1239-1270), ``gcn_agg`` (tpl_fa_gpma.jinja with the GCN statements).
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
        lib.orc_gpma_image.restype = ctypes.c_int64
        for f in ("orc_gpma_degrees", "orc_gpma_label_edges", "orc_gpma_build_backward_csr", "orc_gpma_node_ids",
                  "orc_gpma_gcn_agg"):
            getattr(lib, f).restype = None
        _READY = True
    return lib


def _p(a):
    return _vp(0 if a is None else a.ctypes.data)


class OracleGPMA:
    """rows = the high half of a key.  ``edges``: iterable of (row, col); ``dead``: (row, col) pairs that were
    deleted lazily and still sit in the array with label 0."""

    def __init__(self, num_rows: int, edges, dead=(), hole_pct: int = 30, seed: int = 1):
        self.n = int(num_rows)
        pack = lambda pairs: np.unique(np.array([(int(r) << 32) | int(c) for r, c in pairs], dtype=np.uint64))  # noqa: E731
        self.live, self.dead = pack(edges), pack(dead)
        assert not np.intersect1d(self.live, self.dead).size
        E, D = len(self.live), len(self.dead)
        cap = 4 * (E + D + self.n) + 64
        for _ in range(8):
            self.keys = np.empty(cap, np.uint64)
            self.values = np.empty(cap, np.uint32)
            self.row_offset = np.empty(self.n + 1, np.uint32)
            size = int(_lib().orc_gpma_image(_p(self.live), ctypes.c_int64(E), _p(self.dead), ctypes.c_int64(D),
                                             ctypes.c_int(self.n), ctypes.c_int(int(hole_pct)), ctypes.c_uint64(seed),
                                             _p(self.keys), _p(self.values), _p(self.row_offset), ctypes.c_int64(cap)))
            if size >= 0:
                break
            cap *= 2
        assert size >= 0
        self.keys, self.values = self.keys[:size].copy(), self.values[:size].copy()
        self.in_degree, self.out_degree = np.empty(self.n, np.uint32), np.empty(self.n, np.uint32)
        _lib().orc_gpma_degrees(_p(self.live), ctypes.c_int64(E), ctypes.c_int(self.n), _p(self.in_degree),
                                _p(self.out_degree))
        self.edge_count = E
        self.bwd = None

    def label_edges(self) -> None:
        _lib().orc_gpma_label_edges(_p(self.row_offset), _p(self.keys), _p(self.values), _p(self.out_degree),
                                    ctypes.c_int(self.n))

    def build_backward_csr(self) -> dict:
        ro = np.empty(self.n + 1, np.uint32)
        k, v = np.empty(self.edge_count, np.uint64), np.empty(self.edge_count, np.uint32)
        _lib().orc_gpma_build_backward_csr(_p(self.row_offset), _p(self.keys), _p(self.values), _p(self.in_degree),
                                           ctypes.c_int(self.n), ctypes.c_uint32(self.edge_count), _p(ro), _p(k), _p(v))
        self.bwd = {"row_offset": ro, "keys": k, "values": v}
        return self.bwd

    def node_ids(self, backward: bool = False) -> np.ndarray:
        out = np.empty(self.n, np.uint32)
        _lib().orc_gpma_node_ids(_p(self.in_degree if backward else self.out_degree), ctypes.c_int(self.n), _p(out))
        return out

    def live_edges(self):
        """[(row, col, label)] of the forward array in slot order, holes / walls / tombstones skipped."""
        m = (self.keys != np.uint64(0xFFFFFFFFFFFFFFFE)) & ((self.keys & np.uint64(0xFFFFFFFF)) != np.uint64(0xFFFFFFFF)) \
            & (self.values != 0)
        k, v = self.keys[m], self.values[m]
        return [(int(a >> np.uint64(32)), int(a & np.uint64(0xFFFFFFFF)), int(b)) for a, b in zip(k, v)]

    def gcn_agg(self, x, norm_row, norm_col, ew=None, backward: bool = False, use_node_ids: bool = True,
                f_active: int | None = None, literal: bool = False) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        N, F = x.shape
        nr = np.ascontiguousarray(norm_row, np.float32).reshape(-1)
        nc = np.ascontiguousarray(norm_col, np.float32).reshape(-1)
        w = None if ew is None else np.ascontiguousarray(ew, np.float32).reshape(-1)
        out = np.zeros((N, F), np.float32)
        if backward:
            ro, eids, keys = self.bwd["row_offset"], self.bwd["values"], self.bwd["keys"]
        else:
            ro, eids, keys = self.row_offset, self.values, self.keys
        nid = self.node_ids(backward) if use_node_ids else None
        _lib().orc_gpma_gcn_agg(_p(x), _p(nr), _p(nc), _p(w), _p(out), _p(ro), _p(eids), _p(keys), _p(nid),
                                ctypes.c_int(N), ctypes.c_int(F), ctypes.c_int(F if f_active is None else f_active),
                                ctypes.c_int(int(literal)))
        return out
