"""``PCSR`` -- drop-in for the reference's pybind class ``stgraph.graph.dynamic.pcsr.pcsr.PCSR``
(graph/dynamic/pcsr/pcsr.cu:273-318, bindings :916-939), native on the MI355X.

Same methods and attributes: ``edge_update_list(edge_list, is_delete, is_reverse_edge)``,
``label_edges()``, ``build_csr()``, ``build_reverse_csr()``, ``get_csr_ptrs()``, ``get_edges()``,
``get_n()``, ``in_degrees`` / ``out_degrees`` / ``edge_count``, ``copy`` / ``deepcopy``.  What differs is
the mechanism (see csrc/edge_store.hip): the edge set lives in HBM as two sorted key arrays, an update
is one merge pass on the GPU, and the CSR is emitted on the GPU -- no host packed-memory array, no
pinned staging copy, no H2D transfer per timestamp.  The emitted arrays are the reference's, bit for
bit (rows back to front, 1-based labels), for every update stream the reference itself handles
consistently (tests/test_gpu_pcsr.py, tests/golden/pcsr_*.npz).

Orientation: like the reference, the class speaks in the store's own (src, dst); ``PCSRGraph`` passes
``is_reverse_edge=True`` everywhere, which makes the store's src the graph's dst.
"""
from __future__ import annotations

import numpy as np
import torch

from .... import kernels
from ...static.csr import _LIVE, default_device


def _pairs(edge_list, device):
    """(first, second) int32 device tensors from a list of 2-tuples / [E,2] array / (a, b) arrays."""
    if isinstance(edge_list, tuple) and len(edge_list) == 2 and getattr(edge_list[0], "ndim", 0) == 1:
        a, b = edge_list
    else:
        if isinstance(edge_list, torch.Tensor):
            arr = edge_list.reshape(-1, 2)
        else:
            arr = np.asarray(edge_list, dtype=np.int64).reshape(-1, 2)
        a, b = arr[:, 0], arr[:, 1]
    to = lambda x: (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(  # noqa: E731
        device=device, dtype=torch.int32)
    return to(a), to(b)


FUSED_STEP = True       # merge_sorted as ONE fused device step (kernels.edgeset_step); False: merge, then emit on demand


class PCSR:
    _key_order = False      # emission layout: PCSR rows come out back to front (the GPMA subclass sets True)

    def __init__(self, init_n: int, max_edge_count: int, device=None):
        self._device = torch.device(device) if device is not None else default_device()
        self._n = int(init_n)
        self.max_edge_count = int(max_edge_count)
        self._set = kernels.edgeset_empty(self._n, self._device)
        self._pending = {"add": [], "delete": []}       # (store src, store dst) tensor pairs, not yet merged
        self._emitted = {}                              # reverse(bool) -> kernels.StoreCSR of self._set
        self._emission = kernels.EmissionQueue()        # shared by the copies of this store (deferred_emission)
        self._published = None                          # the arrays the last build_* call handed out
        # sticky status word of the fused steps, shared by every copy of this store (made here, outside any HIP-graph capture)
        self._status = torch.zeros(1, dtype=torch.int32, device=self._device) if self._device.type == "cuda" else None
        self._norm_in = None                            # in_deg ** -0.5 [N, 1] of the current set, when a fused step made it
        self.update_count = 0                           # merge passes issued (two orientations each)

    def _replace_set(self, new_set) -> None:
        """Every replacement of the edge set goes through here: what was derived from the old set -- the emitted CSRs, the
        cached ``norm``, an emission still pending in the queue (written now: someone may hold its arrays) -- goes with it."""
        if self._emission.pending is not None:
            self._emission.flush()
        self._set = new_set
        self._emitted = {}
        self._norm_in = None

    # -- copies share everything immutable (the reference's copies share the device arrays too) --------
    def __copy__(self) -> "PCSR":
        self._flush()
        c = type(self).__new__(type(self))
        c.__dict__.update(self.__dict__)
        c._pending = {"add": [], "delete": []}
        c._emitted = dict(self._emitted)
        return c

    def __deepcopy__(self, memo) -> "PCSR":
        return self.__copy__()

    # -- updates -----------------------------------------------------------------------------------------
    def edge_update_list(self, edge_list, is_delete: bool = False, is_reverse_edge: bool = False) -> None:
        a, b = _pairs(edge_list, self._device)
        src, dst = (b, a) if is_reverse_edge else (a, b)
        if a.numel() == 0:
            return
        # additions followed by deletions are merged in ONE pass; a deletion followed by an addition may
        # touch the same edge, so it closes the batch
        if not is_delete and self._pending["delete"]:
            self._flush()
        self._pending["delete" if is_delete else "add"].append((src, dst))

    def _flush(self) -> None:
        if not (self._pending["add"] or self._pending["delete"]):
            return
        cat = lambda lst, i: (torch.cat([p[i] for p in lst]) if len(lst) > 1 else lst[0][i]) if lst else None  # noqa: E731
        add, dele = self._pending["add"], self._pending["delete"]
        empty = torch.empty(0, dtype=torch.int32, device=self._device)
        a_src, a_dst = (cat(add, 0), cat(add, 1)) if add else (empty, empty)
        d_src, d_dst = (cat(dele, 0), cat(dele, 1)) if dele else (empty, empty)
        # EdgeSet speaks graph orientation (forward rows = graph dst = store src)
        self._replace_set(kernels.edgeset_update(self._set, a_dst, a_src, d_dst, d_src))
        self._pending = {"add": [], "delete": []}
        self.update_count += 1

    def merge_sorted(self, add_keys, del_keys) -> None:
        """Fast path for update batches that were packed + sorted up front (``kernels.edgeset_pack_sorted``,
        graph orientation): adds and deletes of one timestamp in two scatter passes."""
        self._flush()
        if add_keys[0].numel() == 0 and del_keys[0].numel() == 0:
            return
        if FUSED_STEP and self._device.type == "cuda":
            # merge + both CSRs + in-degree norm (+ its per-edge gathers) in two launches: what every consumer of a
            # timestamp asks for next anyway (build_csr / build_reverse_csr, the loop's norm)
            # (the row offsets the previous step emitted for the current set: the new ones are derived from them)
            f0, b0 = self._emitted.get(False), self._emitted.get(True)
            hints = (f0.row_offset, b0.row_offset) if f0 is not None and b0 is not None else None
            self._set, fwd, bwd, self._norm_in = kernels.edgeset_step(self._set, add_keys, del_keys, self._key_order,
                                                                      self._status, hints, self._emission)
            self._emitted = {False: fwd, True: bwd}
        else:
            self._replace_set(kernels.edgeset_merge(self._set, add_keys, del_keys))
        self.update_count += 1

    def label_edges(self) -> None:
        """Labels are positions in the sorted key array: nothing to compute (pcsr.cu:745-757 walks the PMA)."""
        self._flush()

    def check(self) -> None:
        """Raise ValueError if an update violated the stream contract (added a present edge, deleted an
        absent one, id out of range).  Synchronises; the reference has no such check (its counters and
        arrays silently diverge instead)."""
        self._flush()
        kernels.edgeset_check(self._set)

    # -- emitted CSR ---------------------------------------------------------------------------------------
    def _emit(self, reverse: bool):
        self._flush()
        hit = self._emitted.get(reverse)
        if hit is None:
            hit = self._emitted[reverse] = kernels.edgeset_emit_csr(self._set, reverse, self._key_order)
        return hit

    def _publish(self, reverse: bool) -> float:
        self._published = self._emit(reverse)
        return 0.0                                       # "move to GPU" time: there is no transfer

    def build_csr(self) -> float:
        return self._publish(False)

    def build_reverse_csr(self) -> float:
        return self._publish(True)

    def get_csr_ptrs(self):
        """(row_offset, column_indices, eids, node_ids) device addresses of the last build, eids 1-based as in
        the reference (pcsr.cu:888-895).  Asking for them emits the labels."""
        if self._published is None:
            self._publish(False)
        c = self._published
        arrays = (c.row_offset, c.column_indices, c.eids1, c.node_ids)
        for t in arrays:
            _LIVE[t.data_ptr()] = t
        return tuple(int(t.data_ptr()) for t in arrays)

    def csr(self, reverse: bool = False) -> kernels.StoreCSR:
        """The CSR as the launch wrappers take it (0-based eids, emitted lazily); one emission per edge set."""
        return self._emit(reverse)

    def labels(self, reverse: bool = False) -> torch.Tensor:
        """1-based edge labels in CSR order (what the reference's ``eids`` array holds)."""
        return self._emit(reverse).eids1

    def row_lengths(self, reverse: bool = False) -> torch.Tensor:
        return self._emit(reverse).degrees

    # -- the pybind attributes -----------------------------------------------------------------------------
    @property
    def edge_count(self) -> int:
        self._flush()
        return self._set.num_edges

    @property
    def out_degrees(self):
        """Edges per store-source (``out_degrees[src] += 1``, pcsr.cu:773-774) = forward row lengths."""
        return self.row_lengths(False).cpu().numpy()

    @property
    def in_degrees(self):
        return self.row_lengths(True).cpu().numpy()

    def get_n(self) -> int:
        return self._n

    def get_edges(self):
        """[(src, dst, label)] in label order (pcsr.cu:723-743)."""
        self._flush()
        k = self._set.keys_fwd.cpu().numpy()
        return [(int(x >> 32), int(x & 0xFFFFFFFF), i + 1) for i, x in enumerate(k)]

    @property
    def edge_set(self) -> kernels.EdgeSet:
        self._flush()
        return self._set
