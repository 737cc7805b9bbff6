"""``gpma`` -- drop-in for the reference's pybind module ``stgraph.graph.dynamic.gpma.gpma``
(graph/dynamic/gpma/gpma.cu:1441-1467): the ``GPMA`` class plus the module-level functions
``init_gpma, init_graph_updates, edge_update_t, label_edges, build_backward_csr, free_backward_csr,
get_csr_ptrs, get_in_degrees, get_out_degrees`` and the logging helpers ``get_graph_attr,
get_gpma_edge_list, get_reverse_csr_edge_list, get_node_ids``.

The reference keeps a GPU packed-memory array (Sha et al.): a gapped key array with a segment tree,
batched inserts that rebalance level by level (gpma.cu:412-452, 552-836), deletes that only zero the
value (lazy, :215-222), a relabelling walk and a counting sort with atomics for the reverse CSR
(:1121-1231).  All of that serves one observable contract -- what its kernels read (tpl_fa_gpma.jinja):
per row, the live keys in ascending order with 1-based labels that count live edges in key order.
Here that contract is met by the dynamic edge store (csrc/edge_store.hip): the edge SET as two dense
sorted key arrays in HBM, an update = one merge pass per orientation, the CSR emitted in key order
(``STG_EMIT_KEY_ORDER``).  The forward ``column_indices`` array handed out by ``get_csr_ptrs`` is the
packed 64-bit key array itself (``row << 32 | column``), i.e. a GPMA with every hole squeezed out, so the
reference's own kernel template would walk it unchanged.

Differences, all where the reference is undefined or wrong:
  * reverse-CSR rows are ascending (reference: whatever order ``atomicSub`` hands out, :1165-1188);
  * the reference's kernel predicate ``eid != 0`` is applied AFTER ``eid = label - 1``
    (tpl_fa_gpma.jinja:34,43): it drops the edge labelled 1 and lets lazily deleted entries
    (label 0 -> eid 0xFFFFFFFF) through.  This store has neither holes nor tombstones.
No oracle binary exists for this component (``gpma.so`` is absent from the reference tree and the
source needs nvcc + thrust + cub + ``-rdc``): the tests check this module against a CPU restatement of
the cited functions run over a gapped array, and against ``NaiveGraph`` snapshots -- "parity unpinned" in
the task's terms.
"""
from __future__ import annotations

import time

from .... import kernels
from ...static.csr import _LIVE
from ..pcsr.pcsr import PCSR, _pairs


class GPMA(PCSR):
    """State object of the module functions (reference class: gpma.cu:57-105; attributes ``row_num``,
    ``edge_count``; ``copy``/``deepcopy`` supported like the pybind class :1459-1466)."""

    _key_order = True

    def __init__(self, device=None):
        super().__init__(0, 0, device)
        self.row_num = 0
        self._updates = {}            # t -> {"add": (keys_fwd, keys_bwd), "delete": (...)}  packed + sorted
        self._backward = None         # StoreCSR published by build_backward_csr

    def get_size(self) -> int:        # gpma.cu:99-102: slots of the key array (dense here)
        return self.edge_count


def init_gpma(gpma: GPMA, num_nodes: int) -> None:
    """gpma.cu:947-980: an empty graph with ``num_nodes`` rows."""
    gpma.row_num = gpma._n = int(num_nodes)
    gpma._pending = {"add": [], "delete": []}
    gpma._replace_set(kernels.edgeset_empty(gpma._n, gpma._device))      # drops the emitted CSRs, the cached norm, a pending emission
    gpma._published, gpma._backward = None, None


def init_graph_updates(gpma: GPMA, updates, reverse_edges: bool = False) -> None:
    """gpma.cu:984-1032: move every timestamp's add/delete list to the device once.  ``updates`` is the
    reference's ``{str(t): {"add": [(a, b), ...], "delete": [...]}}`` (``DynamicGraph.graph_updates``);
    with ``reverse_edges`` the row of an edge is its second element (every caller passes True, so rows are
    graph destinations).  Lists may also be ``(a, b)`` tensor/array pairs."""
    gpma._updates = {}
    for t in range(len(updates)):
        u = updates[str(t)] if str(t) in updates else updates[t]
        entry = {}
        for kind in ("add", "delete"):
            a, b = _pairs(u[kind], gpma._device)
            row, col = (b, a) if reverse_edges else (a, b)
            if gpma._device.type == "cuda":
                entry[kind] = kernels.edgeset_pack_sorted(col, row, gpma._device)     # (src=col, dst=row)
            else:
                entry[kind] = (row, col)
        gpma._updates[t] = entry


def edge_update_t(gpma: GPMA, timestamp: int, revert_update: bool = False):
    """gpma.cu:1064-1119: apply (or, with ``revert_update``, undo) the updates of ``timestamp``.
    Returns ``[update seconds, degree-update seconds]`` like the reference (degrees are row lengths of the
    emitted CSR here, so the second figure is 0)."""
    u = gpma._updates[int(timestamp)]
    add, dele = (u["delete"], u["add"]) if revert_update else (u["add"], u["delete"])
    t0 = time.perf_counter()
    if gpma._device.type == "cuda":
        gpma.merge_sorted(add, dele)
    else:
        gpma.edge_update_list(add, False, False)
        gpma.edge_update_list(dele, True, False)
        gpma._flush()
    gpma._backward = None
    return [time.perf_counter() - t0, 0.0]


def label_edges(gpma: GPMA) -> None:
    """gpma.cu:1148-1163 relabels every live edge 1..E in key order after each update; in a dense sorted
    key array the label IS the position + 1, so there is nothing to do until something asks for it."""
    gpma._flush()


def build_backward_csr(gpma: GPMA):
    """gpma.cu:1190-1231: the reverse CSR (rows = key columns) with the forward labels."""
    t0 = time.perf_counter()
    gpma._backward = gpma._emit(True)
    return [0.0, 0.0, time.perf_counter() - t0]


def free_backward_csr(gpma: GPMA) -> None:
    """gpma.cu:1233-1237."""
    gpma._backward = None


def _addresses(arrays):
    for t in arrays:
        _LIVE[t.data_ptr()] = t
    return tuple(int(t.data_ptr()) for t in arrays)


def get_csr_ptrs(gpma: GPMA, is_backward: bool = False):
    """gpma.cu:1239-1270: ``(row_offset, column_indices, eids, node_ids)`` device addresses in the types the
    reference's kernels declare (tpl_fa_gpma.jinja:3-6): uint32 row offsets, 1-based uint32 labels,
    **uint64 packed keys** as column_indices, node ids by non-increasing row length."""
    if is_backward:
        if gpma._backward is None:
            raise RuntimeError("get_csr_ptrs(is_backward=True) before build_backward_csr()")
        c = gpma._backward
    else:
        c = gpma._emit(False)
    return _addresses((c.row_offset, c.keys, c.eids1, c.node_ids))


def get_out_degrees(gpma: GPMA):
    """Edges per row (``out_degree[key >> 32]``, gpma.cu:1034-1049)."""
    return gpma.row_lengths(False).cpu().tolist()


def get_in_degrees(gpma: GPMA):
    """Edges per column (``in_degree[(uint32)key]``)."""
    return gpma.row_lengths(True).cpu().tolist()


# ---- logging helpers (gpma.cu:1288-1439) ------------------------------------------------------------------
def get_graph_attr(gpma: GPMA):
    return (gpma.row_num, gpma.edge_count)


def get_gpma_edge_list(gpma: GPMA):
    """{(row, column, label)} of the forward array (gpma.cu:1355-1397)."""
    gpma._flush()
    k = gpma._set.keys_fwd.cpu().numpy()
    return {(int(x >> 32), int(x & 0xFFFFFFFF), i + 1) for i, x in enumerate(k)}


def get_reverse_csr_edge_list(gpma: GPMA):
    """{(row, column, label)} of the reverse CSR (gpma.cu:1399-1439)."""
    if gpma._backward is None:
        raise RuntimeError("get_reverse_csr_edge_list before build_backward_csr()")
    c = gpma._backward
    k, lab = c.keys.cpu().numpy(), c.eids1.cpu().numpy()
    return {(int(x >> 32), int(x & 0xFFFFFFFF), int(l)) for x, l in zip(k, lab)}


def get_node_ids(gpma: GPMA):
    """Node ids as last published by ``get_csr_ptrs`` (forward order unless a reverse CSR is built)."""
    c = gpma._backward if gpma._backward is not None else gpma._emit(False)
    return c.node_ids.cpu().tolist()


def print_gpma_info(gpma: GPMA, node: int) -> None:
    c = gpma._emit(False)
    ro = c.row_offset.cpu().numpy()
    beg, end = int(ro[node]), int(ro[node + 1])
    k = gpma._set.keys_fwd[beg:end].cpu().numpy()
    print(f"node {node} ({beg}, {end}): " + "  ".join(f"{int(x & 0xFFFFFFFF)}({beg + i + 1})" for i, x in enumerate(k)))


__all__ = ["GPMA", "init_gpma", "init_graph_updates", "edge_update_t", "label_edges", "build_backward_csr",
           "free_backward_csr", "get_csr_ptrs", "get_in_degrees", "get_out_degrees", "get_graph_attr",
           "get_gpma_edge_list", "get_reverse_csr_edge_list", "get_node_ids", "print_gpma_info"]
