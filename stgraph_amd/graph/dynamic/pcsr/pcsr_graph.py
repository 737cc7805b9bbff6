"""``PCSRGraph`` -- drop-in for ``stgraph.graph.PCSRGraph`` (reference
graph/dynamic/pcsr/pcsr_graph.py:14-166): ONE resident graph plus per-timestamp add/delete lists
instead of T snapshots; ``get_graph(t)`` applies updates forward, the backward pass walks them back.

Same protocol and the same emitted arrays as the reference; the update lists are moved to the device
once at construction and every step is a GPU merge + emit (csrc/edge_store.hip).  Deviations, all in
the direction of correctness (defects D13-D15):
  * a restored graph (``reset_graph`` -> "base", or the window boundary) publishes ITS OWN arrays; the
    reference's copies share one set of device arrays, so after ``reset_graph`` its first forward step
    of every epoch but the first reads whatever the last backward step left there;
  * update streams are validated on the device (``check()``).
"""
from __future__ import annotations

import contextlib
import copy
import time

import numpy as np
import torch

from .... import kernels
from ...static.csr import _LIVE, default_device
from ...static.static_graph import edge_arrays
from ..dynamic_graph import DynamicGraph
from .pcsr import PCSR


def _keys(edges) -> np.ndarray:
    s, d = edge_arrays(edges)
    s = s.cpu().numpy() if isinstance(s, torch.Tensor) else np.asarray(s)
    d = d.cpu().numpy() if isinstance(d, torch.Tensor) else np.asarray(d)
    return np.unique((d.astype(np.int64) << 32) | s.astype(np.int64))       # (dst, src) order, distinct


class PCSRGraph(DynamicGraph):
    def __init__(self, edge_list, max_num_nodes: int, device=None) -> None:
        self._ptr_src = {}
        super().__init__(edge_list, max_num_nodes)
        self._device = torch.device(device) if device is not None else default_device()
        t0 = time.time()
        # per-timestamp updates as device tensors, sorted by (dst, src) like dynamic_graph.py:56-79
        self._updates = []
        prev = np.empty(0, np.int64)
        ever = prev
        for t in range(self._num_timestamps):
            cur = _keys(edge_list[t])
            add = np.setdiff1d(cur, prev, assume_unique=True)
            dele = np.setdiff1d(prev, cur, assume_unique=True)
            u = {"add": self._to_device(add), "delete": self._to_device(dele)}
            if self._device.type == "cuda":       # packed + sorted once, here; a step is then two scatter passes
                u["add_keys"] = kernels.edgeset_pack_sorted(*u["add"], self._device)
                u["delete_keys"] = kernels.edgeset_pack_sorted(*u["delete"], self._device)
            self._updates.append(u)
            self._distinct_edges[t] = int(cur.shape[0])
            ever = np.union1d(ever, add)
            prev = cur
        self.max_num_edges = int(ever.shape[0])                              # pcsr_graph.py:64-71
        self.move_to_gpu_time += time.time() - t0

        self._forward_graph = self._new_store()
        if self._num_timestamps:
            self._forward_graph.edge_update_list(self._updates[0]["add"], is_reverse_edge=True)
        self._forward_graph.label_edges()
        self._forward_graph.build_csr()
        self._get_graph_csr_ptrs()
        self.graph_cache = {"base": copy.deepcopy(self._forward_graph)}

    def _new_store(self):
        return PCSR(self.max_num_nodes, self.max_num_edges, self._device)

    @staticmethod
    def _published_arrays(c):
        """The four arrays behind ``*_row_offset_ptr, *_column_indices_ptr, *_eids_ptr, *_node_ids_ptr``."""
        return (c.row_offset, c.column_indices, c.eids1, c.node_ids)

    def _to_device(self, keys: np.ndarray):
        src = torch.from_numpy((keys & 0xFFFFFFFF).astype(np.int32)).to(self._device)
        dst = torch.from_numpy((keys >> 32).astype(np.int32)).to(self._device)
        return (src, dst)

    # -- DynamicGraph protocol ---------------------------------------------------------------------------
    def graph_type(self) -> str:
        return "pcsr"

    @property
    def device(self) -> torch.device:
        return self._device

    def _num_edges_at(self, timestamp: int) -> int:
        return self._distinct_edges[timestamp]

    def _cache_graph(self) -> None:
        self.graph_cache[str(self.current_timestamp)] = copy.deepcopy(self._forward_graph)

    def _get_cached_graph(self, timestamp) -> bool:
        if timestamp == "base":
            self._forward_graph = copy.deepcopy(self.graph_cache["base"])
            self._forward_graph.build_csr()
            self._is_backprop_state = False
            self._get_graph_csr_ptrs()
            return True
        if str(timestamp) in self.graph_cache:
            self._forward_graph = self.graph_cache.pop(str(timestamp))
            self._forward_graph.build_csr()
            self._get_graph_csr_ptrs()
            return True
        return False

    def in_degrees(self) -> np.ndarray:
        return np.array(self._forward_graph.out_degrees, dtype="int32")      # pcsr_graph.py:101-103

    def out_degrees(self) -> np.ndarray:
        return np.array(self._forward_graph.in_degrees, dtype="int32")

    def in_degrees_tensor(self) -> torch.Tensor:
        return self._forward_graph.row_lengths(False)

    def in_degree_norm_tensor(self):
        """``in_deg ** -0.5`` [N, 1] of the current timestamp if the store's fused step already produced it (else None)."""
        return self._forward_graph._norm_in

    def _get_graph_csr_ptrs(self, *_):
        """Remember which build the eight ``fwd_*/bwd_*_ptr`` attributes refer to; the addresses themselves
        (and with them the 1-based label array) are produced when an attribute is read."""
        self._ptr_src["bwd" if self._is_backprop_state else "fwd"] = self._forward_graph._published

    def _ptrs(self, side: str):
        c = self._ptr_src.get(side)
        if c is None:
            return (None, None, None, None)
        arrays = self._published_arrays(c)
        for t in arrays:
            _LIVE[t.data_ptr()] = t
        return tuple(int(t.data_ptr()) for t in arrays)

    def _on_timestamp_change(self) -> None:
        if self._is_backprop_state:
            self._forward_graph.build_reverse_csr()
        else:
            self._forward_graph.build_csr()
        self._get_graph_csr_ptrs()

    def _apply(self, u: dict, inverse: bool = False) -> None:
        g = self._forward_graph
        a, d = ("delete", "add") if inverse else ("add", "delete")
        if "add_keys" in u:
            g.merge_sorted(u[a + "_keys"], u[d + "_keys"])
        else:
            g.edge_update_list(u[a], is_reverse_edge=True)
            g.edge_update_list(u[d], is_delete=True, is_reverse_edge=True)
        g.label_edges()

    def _update_graph_forward(self) -> None:
        t = self.current_timestamp + 1
        if t >= self._num_timestamps:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_forward()")
        self._apply(self._updates[t])
        self.move_to_gpu_time += self._forward_graph.build_csr()
        self._get_graph_csr_ptrs()

    def _init_reverse_graph(self) -> None:
        self.move_to_gpu_time += self._forward_graph.build_reverse_csr()
        self._get_graph_csr_ptrs()

    def _update_graph_backward(self) -> None:
        t = self.current_timestamp
        if t <= 0:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_backward()")
        self._apply(self._updates[t], inverse=True)                          # undo the step t-1 -> t
        self.move_to_gpu_time += self._forward_graph.build_reverse_csr()
        self._get_graph_csr_ptrs()

    # -- tensors for the launch wrappers -------------------------------------------------------------------
    def csr(self, direction: str, timestamp=None) -> kernels.DeviceCSR:
        if timestamp is not None and timestamp != self.current_timestamp:
            raise ValueError("a PCSRGraph holds one timestamp at a time; move it with get_graph / get_backward_graph")
        return self._forward_graph.csr(direction == "bwd")

    def check(self) -> None:
        self._forward_graph.check()

    @contextlib.contextmanager
    def deferred_emission(self):
        """For a caller that moves the graph through several timestamps BEFORE anything reads the CSRs' columns (a captured
        BPTT window collects all its snapshots' CSR handles first): inside the block a step's emission -- columns + per-edge
        norm of both CSRs -- is not launched on its own but rides in the NEXT step's merge launch
        (stg_edgeset_step_deferred_device: one launch per timestamp instead of two); the last one is issued on leaving the
        block.  Row offsets, degrees and norm are complete after every step; column arrays only after the block."""
        q = self._forward_graph._emission
        q.defer = True
        try:
            yield self
        finally:
            q.defer = False
            q.flush()


def _ptr_property(side: str, index: int):
    def get(self):
        return self._ptrs(side)[index]

    def set_(self, value):          # STGraphBase.__init__ assigns None to all eight
        if value is not None:
            raise AttributeError("PCSRGraph publishes its own CSR pointers")
    return property(get, set_)


for _side in ("fwd", "bwd"):          # inherited by GPMAGraph
    for _i, _name in enumerate(("row_offset", "column_indices", "eids", "node_ids")):
        setattr(PCSRGraph, f"{_side}_{_name}_ptr", _ptr_property(_side, _i))
