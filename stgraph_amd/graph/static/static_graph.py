"""``StaticGraph`` -- drop-in for ``stgraph.graph.StaticGraph``
(reference graph/static/static_graph.py:16-126).

Same constructor ``StaticGraph(edge_list, edge_weights, num_nodes)`` and methods.
What differs underneath: the reference sorts a Python list of tuples with a lambda
key, copies it through pybind and walks it in single-threaded C++ (seconds at 16 M
edges); here both CSRs are produced on the MI355X by ``stg_graph_build_device``
(radix sort + binary search), bit-identical arrays, with the edge list accepted
either as the reference's list of ``(src, dst)`` tuples or as an ``[E, 2]`` /
``(src, dst)`` integer array.

Reference semantics kept on purpose:
  * the caller's edge list is re-ordered IN PLACE into (dst, src) order
    (static_graph.py:66-67) -- pass ``sort_inplace=False`` to opt out;
  * edge-weight / edge-feature tensors are indexed by eid = position in that order;
  * ``get_num_edges()`` counts distinct (src, dst) pairs (static_graph.py:48)
    while the CSR keeps duplicates.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import kernels
from ..stgraph_base import STGraphBase
from .csr import _LIVE, default_device


def edge_arrays(edge_list):
    """(src, dst) int32 numpy arrays from a list of tuples, an [E,2] array or a (src, dst) pair."""
    if isinstance(edge_list, tuple) and len(edge_list) == 2 and not np.isscalar(edge_list[0]) \
            and len(np.shape(edge_list[0])) == 1:
        src, dst = edge_list
        if isinstance(src, torch.Tensor):
            return src, dst
        return np.ascontiguousarray(src, dtype=np.int32), np.ascontiguousarray(dst, dtype=np.int32)
    if isinstance(edge_list, torch.Tensor):
        if edge_list.dim() != 2 or edge_list.shape[1] != 2:
            raise ValueError("edge tensor must be [E, 2]")
        return edge_list[:, 0].contiguous(), edge_list[:, 1].contiguous()
    arr = np.asarray(edge_list)
    if arr.size == 0:
        return np.empty(0, np.int32), np.empty(0, np.int32)
    if arr.ndim != 2 or arr.shape[1] < 2:
        raise ValueError("edge_list must be a list of (src, dst) tuples or an [E, 2] array")
    if arr.min() < 0 or arr.max() > np.iinfo(np.int32).max:
        raise ValueError("vertex ids must be non-negative int32")
    return np.ascontiguousarray(arr[:, 0], dtype=np.int32), np.ascontiguousarray(arr[:, 1], dtype=np.int32)


def reorder_inplace(edge_list, perm: np.ndarray) -> None:
    """Apply the forward (dst, src) ordering to the caller's container (static_graph.py:66-67)."""
    if isinstance(edge_list, list):
        edge_list[:] = [edge_list[i] for i in perm.tolist()]
    elif isinstance(edge_list, np.ndarray):
        edge_list[:] = edge_list[perm]
    elif isinstance(edge_list, torch.Tensor):
        edge_list.copy_(edge_list[torch.as_tensor(perm, device=edge_list.device)])


def count_distinct_edges(g: kernels.GraphCSR) -> int:
    """len(set(edge_list)) from the sorted forward CSR: equal (row, col) pairs are adjacent."""
    E = g.num_edges
    if E <= 1:
        return E
    col = g.fwd.column_indices
    rows = torch.repeat_interleave(
        torch.arange(g.num_nodes, device=col.device, dtype=torch.int32),
        (g.fwd.row_offset[1:] - g.fwd.row_offset[:-1]).to(torch.int64), output_size=E)
    dup = (col[1:] == col[:-1]) & (rows[1:] == rows[:-1])
    return E - int(dup.sum().item())


class StaticGraph(STGraphBase):
    def __init__(self, edge_list, edge_weights, num_nodes: int, device=None, sort_inplace: bool = True):
        super().__init__()
        self._num_nodes = int(num_nodes)
        self._device = torch.device(device) if device is not None else default_device()
        src, dst = edge_arrays(edge_list)
        self._graph = kernels.build_graph_csr(src, dst, self._num_nodes, self._device)
        self._num_edges = count_distinct_edges(self._graph)
        self._edge_weights = edge_weights
        if sort_inplace and self._graph.num_edges > 0:
            reorder_inplace(edge_list, self._graph.perm_fwd.cpu().numpy())
        self._forward_graph = self._graph.fwd
        self._backward_graph = self._graph.bwd
        self._get_graph_csr_ptrs()

    def _get_graph_csr_ptrs(self) -> None:
        f, b = self._graph.fwd, self._graph.bwd
        self.fwd_row_offset_ptr, self.fwd_column_indices_ptr = f.row_offset_ptr, f.column_indices_ptr
        self.fwd_eids_ptr, self.fwd_node_ids_ptr = f.eids_ptr, f.node_ids_ptr
        self.bwd_row_offset_ptr, self.bwd_column_indices_ptr = b.row_offset_ptr, b.column_indices_ptr
        self.bwd_eids_ptr, self.bwd_node_ids_ptr = b.eids_ptr, b.node_ids_ptr
        for c in (f, b):
            for t in (c.row_offset, c.column_indices, c.eids, c.node_ids):
                _LIVE[t.data_ptr()] = t

    @property
    def device(self) -> torch.device:
        return self._device

    def csr(self, direction: str, timestamp=None) -> kernels.DeviceCSR:
        return self._graph.fwd if direction == "fwd" else self._graph.bwd

    def get_num_nodes(self) -> int:
        return self._num_nodes

    def get_num_edges(self) -> int:
        return self._num_edges

    def get_ndata(self, field):
        return self._ndata.get(field, None)

    def set_ndata(self, field, val) -> None:
        self._ndata[field] = val

    def graph_type(self) -> str:
        return "csr_unsorted"

    def in_degrees(self) -> np.ndarray:
        return self._graph.in_degrees.cpu().numpy().astype("int32")

    def out_degrees(self) -> np.ndarray:
        return self._graph.out_degrees.cpu().numpy().astype("int32")

    def weighted_in_degrees(self) -> np.ndarray:
        """Sum of in-edge weights per vertex, truncated to int32 as the reference does
        (static_graph.py:124-126; csr.cu:126 accumulates sequentially in fp32)."""
        g = self._graph
        E = g.num_edges
        src = g.fwd.column_indices.cpu().numpy()
        ro = g.fwd.row_offset.cpu().numpy()
        dst = np.repeat(np.arange(self._num_nodes, dtype=np.int32), np.diff(ro))
        w = None if self._edge_weights is None else np.asarray(self._edge_weights, dtype=np.float32).reshape(-1)
        h = kernels.csr_ctor_host(src, dst, np.arange(E, dtype=np.int32), w, self._num_nodes, True)
        return np.array(h["weighted_out_degrees"], dtype="int32")
