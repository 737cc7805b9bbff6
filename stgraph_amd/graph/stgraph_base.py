"""Abstract graph surface the Seastar kernels read.

Mirrors ``stgraph.graph.STGraphBase`` (reference graph/stgraph_base.py:46-90):
eight integer device addresses ``fwd_/bwd_{row_offset,column_indices,eids,
node_ids}_ptr`` plus node-data and size accessors.  The new build additionally
exposes the arrays as tensors through :meth:`csr` so that the launch wrappers can
validate shapes and devices before a kernel sees a raw pointer.
"""
from __future__ import annotations

from abc import ABC, abstractmethod


class STGraphBase(ABC):
    def __init__(self) -> None:
        self._ndata = {}

        self._forward_graph = None
        self._backward_graph = None

        self.fwd_row_offset_ptr = None
        self.fwd_column_indices_ptr = None
        self.fwd_eids_ptr = None
        self.fwd_node_ids_ptr = None

        self.bwd_row_offset_ptr = None
        self.bwd_column_indices_ptr = None
        self.bwd_eids_ptr = None
        self.bwd_node_ids_ptr = None

    @abstractmethod
    def _get_graph_csr_ptrs(self) -> None:
        ...

    @abstractmethod
    def get_num_nodes(self) -> int:
        ...

    @abstractmethod
    def get_num_edges(self) -> int:
        ...

    @abstractmethod
    def get_ndata(self, field: str):
        ...

    @abstractmethod
    def set_ndata(self, field: str, val) -> None:
        ...

    @abstractmethod
    def graph_type(self) -> str:
        ...

    @abstractmethod
    def csr(self, direction: str, timestamp: int | None = None):
        """Return the :class:`stgraph_amd.kernels.DeviceCSR` for 'fwd' or 'bwd'."""
