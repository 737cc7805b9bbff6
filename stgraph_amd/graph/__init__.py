"""Graph representations read by the Seastar kernels (reference: stgraph/graph/__init__.py).
"""
from .dynamic.dynamic_graph import DynamicGraph
from .dynamic.gpma.gpma_graph import GPMAGraph
from .dynamic.naive.naive_graph import NaiveGraph
from .dynamic.pcsr.pcsr_graph import PCSRGraph
from .static.static_graph import StaticGraph
from .stgraph_base import STGraphBase

__all__ = ["DynamicGraph", "GPMAGraph", "NaiveGraph", "PCSRGraph", "StaticGraph", "STGraphBase"]
