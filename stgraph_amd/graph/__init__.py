"""Graph representations read by the Seastar kernels (reference: stgraph/graph/__init__.py).

``GPMAGraph`` and ``PCSRGraph`` are not part of this build ("next").
"""
from .dynamic.dynamic_graph import DynamicGraph
from .dynamic.naive.naive_graph import NaiveGraph
from .static.static_graph import StaticGraph
from .stgraph_base import STGraphBase

__all__ = ["DynamicGraph", "NaiveGraph", "StaticGraph", "STGraphBase"]
