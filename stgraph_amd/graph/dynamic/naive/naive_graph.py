"""``NaiveGraph`` -- per-snapshot CSR dynamic graph, drop-in for
``stgraph.graph.NaiveGraph`` (reference graph/dynamic/naive/naive_graph.py:12-151).

The reference builds T forward + T backward CSRs on the host up front and keeps
all 2T resident on the GPU; moving between timestamps only swaps pointers.  Here a
snapshot's CSR pair is built ON THE DEVICE (``stg_graph_build_direct_device``: histogram,
scan, scatter and rank-by-counting in 6 launches; ``stg_graph_build_device`` -- radix sort
+ binary search -- above 2M edges or when a row is longer than 2048 entries) -- either all up front (``resident=True``, the reference's memory
behaviour) or on first use (``resident=False``: "per-snapshot CSR rebuild", with
``max_cached`` most-recent snapshots kept so that a BPTT window can walk back
through the snapshots its forward pass just used).  ``graph_type()`` is ``'csr'``:
kernels visit rows in ``node_ids`` (degree-descending) order, as tpl_fa_csr.jinja does.

``_get_cached_graph`` implements the intended semantics (SURVEY.md Appendix A, D5:
the reference's override lacks the ``timestamp`` parameter and raises TypeError).
"""
from __future__ import annotations

import time
from collections import OrderedDict

import numpy as np
import torch

from .... import kernels
from ...static.csr import _LIVE, default_device
from ...static.static_graph import count_distinct_edges, edge_arrays, reorder_inplace
from ..dynamic_graph import DynamicGraph


class NaiveGraph(DynamicGraph):
    def __init__(self, edge_list, max_num_nodes: int, device=None, resident: bool = True,
                 max_cached: int | None = None, sort_inplace: bool = True):
        self._ptr_src = {}
        super().__init__(edge_list, max_num_nodes)
        self._device = torch.device(device) if device is not None else default_device()
        self._sort_inplace = sort_inplace
        self._resident = bool(resident)
        self._max_cached = max_cached
        self._snapshots: "OrderedDict[int, kernels.GraphCSR]" = OrderedDict()
        self._built_by = {}                    # t -> builder that produced snapshot t last time ('direct' | 'sort')
        self._pending_status = []              # status words of rebuilds whose read was skipped; checked in bulk
        self._edges = []                       # per-t (src, dst) device tensors in caller order
        t0 = time.time()
        for t in range(self._num_timestamps):
            s, d = edge_arrays(edge_list[t])
            to = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(a)).to(  # noqa: E731
                device=self._device, dtype=torch.int32)
            self._edges.append((to(s), to(d)))
        self.move_to_gpu_time += time.time() - t0
        self.build_count = 0
        self.build_time = 0.0
        if self._resident:
            for t in range(self._num_timestamps):
                self._snapshot(t)
        self._update_count = 0
        self._total_update_time = 0
        self._gpu_move_time = 0
        if self._num_timestamps:
            self._get_graph_csr_ptrs(0)

    # -- snapshot store -------------------------------------------------------------------------
    def _snapshot(self, t: int, counters_slot: int = 0) -> kernels.GraphCSR:
        g = self._snapshots.get(t)
        if g is None:
            t0 = time.time()
            s, d = self._edges[t]
            # snapshots built on demand live for one training step: their degree sorts (node_ids) wait for a reader
            g = kernels.build_graph_csr(s, d, self.max_num_nodes, self._device, lazy_node_ids=not self._resident,
                                        known_path=self._built_by.get(t),       # validated once: no status sync on rebuilds
                                        counters_slot=counters_slot)
            self._built_by[t] = g.built_by
            if g.unchecked_status is not None and not any(g.unchecked_status is p for p in self._pending_status[-1:]):
                self._pending_status.append(g.unchecked_status)      # (the fused rebuild's word is one per device: once)
                if len(self._pending_status) >= 4096:
                    self.verify_builds()
            self.build_count += 1
            if t not in self._distinct_edges:
                self._distinct_edges[t] = count_distinct_edges(g)
                if self._sort_inplace and g.num_edges:
                    reorder_inplace(self._edge_list_ref[t], g.perm_fwd.cpu().numpy())
                    self._edges[t] = (s[g.perm_fwd], d[g.perm_fwd])     # keep in step with the caller's list
                    g.perm_fwd = torch.arange(g.num_edges, device=self._device)
            self._snapshots[t] = g
            self.build_time += time.time() - t0
            if not self._resident and self._max_cached is not None:
                while len(self._snapshots) > max(1, self._max_cached):
                    self._snapshots.popitem(last=False)
        else:
            self._snapshots.move_to_end(t)
        return g

    def prebuild(self, timestamps, counters_base: int = 0) -> int:
        """Rebuild the not-yet-cached snapshots among ``timestamps`` (each validated by an earlier build on the counting
        path, non-empty) as one batched device build -- the launches of ONE snapshot's rebuild, bit-identical CSRs.  What
        does not qualify is left to ``_snapshot``.  Returns the number of snapshots built.  ``counters_base``: see
        ``kernels.build_graph_csr_batch`` (builds issued on a second stream)."""
        cap = kernels._C.BUILD_BATCH_MAX
        if not self._resident and self._max_cached is not None:          # never more snapshots alive than the cache allows
            cap = min(cap, max(1, self._max_cached) - len(self._snapshots))
        ts = [t for t in timestamps if t not in self._snapshots and self._built_by.get(t) == "direct"
              and t in self._distinct_edges and self._edges[t][0].numel() > 0]
        batches = -(-len(ts) // kernels._C.BUILD_BATCH_MAX)              # a window of 20: two batches of 10, not 16 + 4
        ts = ts[:max(min(cap, -(-len(ts) // max(batches, 1))), 0)]
        if (len(ts) < 2 or self._device.type != "cuda" or not kernels.FUSED_REBUILD or not kernels._DIRECT_BUILD
                or not 0 < self.max_num_nodes <= kernels.BUILD_BATCH_MAX_NODES or any(self._edges[t][0].numel() > kernels.DIRECT_BUILD_MAX_EDGES for t in ts)):
            return 0
        t0 = time.time()
        built = kernels.build_graph_csr_batch([self._edges[t] for t in ts], self.max_num_nodes, self._device, counters_base,
                                              ids=[t + 1 for t in ts])
        for t, g in zip(ts, built):
            self._snapshots[t] = g
        if not any(built[0].unchecked_status is p for p in self._pending_status[-1:]):
            self._pending_status.append(built[0].unchecked_status)
        self.build_count += len(ts)
        self.build_time += time.time() - t0
        return len(ts)

    def verify_builds(self) -> None:
        """One host sync for all rebuilds since the last call whose per-build status read was skipped (a snapshot that
        was validated when first built).  ``reset_graph`` -- the start of every epoch of the reference's loops -- calls it."""
        pending, self._pending_status = self._pending_status, []
        kernels.check_build_statuses(pending)

    def reset_graph(self) -> None:
        self.verify_builds()
        super().reset_graph()

    def _num_edges_at(self, timestamp: int) -> int:
        if timestamp not in self._distinct_edges:
            self._snapshot(timestamp)
        return self._distinct_edges[timestamp]

    def csr(self, direction: str, timestamp=None) -> kernels.DeviceCSR:
        g = self._snapshot(self.current_timestamp if timestamp is None else timestamp)
        return g.fwd if direction == "fwd" else g.bwd

    @property
    def device(self) -> torch.device:
        return self._device

    def graph_type(self) -> str:
        return "csr"

    def _cache_graph(self) -> None:
        pass

    def _get_cached_graph(self, timestamp=None) -> bool:
        return False

    def in_degrees(self) -> np.ndarray:
        return self._snapshot(self.current_timestamp).in_degrees.cpu().numpy().astype("int32")

    def out_degrees(self) -> np.ndarray:
        return self._snapshot(self.current_timestamp).out_degrees.cpu().numpy().astype("int32")

    def in_degrees_tensor(self) -> torch.Tensor:
        """Device-resident in-degrees of the current snapshot (avoids the D2H copy of ``in_degrees``)."""
        return self._snapshot(self.current_timestamp).in_degrees

    def in_degree_norm_tensor(self):
        """``in_deg ** -0.5`` [N, 1] of the current snapshot if its build already produced it (fused rebuild), else None."""
        return getattr(self._snapshot(self.current_timestamp), "norm_in", None)

    def _get_graph_csr_ptrs(self, timestamp: int) -> None:
        """Remember which snapshot the ``fwd_*/bwd_*_ptr`` attributes refer to; the addresses (and with them a
        lazily built snapshot's ``node_ids``) are produced when an attribute is read."""
        g = self._snapshot(timestamp)
        self._ptr_src["bwd" if self._is_backprop_state else "fwd"] = g.bwd if self._is_backprop_state else g.fwd

    def _ptrs(self, side: str):
        c = self._ptr_src.get(side)
        if c is None:
            return (None, None, None, None)
        arrays = (c.row_offset, c.column_indices, c.eids, c.node_ids)
        for t in arrays:
            _LIVE[t.data_ptr()] = t
        return tuple(int(t.data_ptr()) for t in arrays)

    # Moving between timestamps is a pointer swap here, so jump straight to the target instead of
    # stepping through (and, when not resident, building) every snapshot in between.
    def get_graph(self, timestamp: int) -> None:
        t0 = time.time()
        self._is_backprop_state = False
        if timestamp < self.current_timestamp:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_forward()")
        if timestamp >= self._num_timestamps:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_forward()")
        self.current_timestamp = timestamp
        self._get_graph_csr_ptrs(timestamp)
        self.get_fwd_graph_time += time.time() - t0

    def get_backward_graph(self, timestamp: int) -> None:
        t0 = time.time()
        if timestamp > self.current_timestamp:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_backward()")
        if timestamp < 0:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_backward()")
        self._is_backprop_state = True
        self.current_timestamp = timestamp
        self._get_graph_csr_ptrs(timestamp)
        self.get_bwd_graph_time += time.time() - t0

    def _on_timestamp_change(self) -> None:
        if self._num_timestamps:
            self._get_graph_csr_ptrs(self.current_timestamp)

    def _update_graph_forward(self) -> None:
        if self.current_timestamp + 1 >= self._num_timestamps:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_forward()")
        self._get_graph_csr_ptrs(self.current_timestamp + 1)

    def _init_reverse_graph(self) -> None:
        self._get_graph_csr_ptrs(self.current_timestamp)

    def _update_graph_backward(self) -> None:
        if self.current_timestamp < 0:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_backward()")
        self._get_graph_csr_ptrs(self.current_timestamp - 1)


def _ptr_property(side: str, index: int):
    def get(self):
        return self._ptrs(side)[index]

    def set_(self, value):          # STGraphBase.__init__ assigns None to all eight
        if value is not None:
            raise AttributeError("NaiveGraph publishes its own CSR pointers")
    return property(get, set_)


for _side in ("fwd", "bwd"):
    for _i, _name in enumerate(("row_offset", "column_indices", "eids", "node_ids")):
        setattr(NaiveGraph, f"{_side}_{_name}_ptr", _ptr_property(_side, _i))
