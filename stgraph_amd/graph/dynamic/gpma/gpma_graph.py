"""``GPMAGraph`` -- drop-in for ``stgraph.graph.GPMAGraph`` (reference
graph/dynamic/gpma/gpma_graph.py:24-152): one device-resident graph plus per-timestamp add/delete
batches; ``get_graph(t)`` applies batches forward, the backward pass reverts them and rebuilds the reverse
CSR each step.  ``graph_type()`` is ``'gpma'``: kernels visit rows through ``node_ids`` and see, per row,
the live neighbours in ascending order with labels counting live edges in key order
(tpl_fa_gpma.jinja, gpma.cu:1121-1188) -- which for a duplicate-free stream is exactly the CSR a
``NaiveGraph`` builds for the same snapshot (forward eids = positions, reverse eids = forward positions).

Protocol, caching and timers are ``PCSRGraph``'s (the two reference classes are line-for-line parallel:
pcsr_graph.py:73-166 vs gpma_graph.py:75-152); the store is the ``gpma`` module's (see its docstring
for what replaces the packed-memory array and why).  The eight ``fwd_*/bwd_*_ptr`` attributes publish the
reference's array types: uint64 packed keys as ``column_indices``, 1-based uint32 labels as ``eids``.
"""
from __future__ import annotations

from ..pcsr.pcsr_graph import PCSRGraph
from .gpma import GPMA, init_gpma


class GPMAGraph(PCSRGraph):
    def _new_store(self):
        g = GPMA(self._device)
        init_gpma(g, self.max_num_nodes)
        return g

    @staticmethod
    def _published_arrays(c):
        return (c.row_offset, c.keys, c.eids1, c.node_ids)

    def graph_type(self) -> str:
        return "gpma"
