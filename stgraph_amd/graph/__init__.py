"""Graph representations read by the Seastar kernels (reference: stgraph/graph/__init__.py).

``GPMAGraph`` is not part of this build ("next").
"""
from .dynamic.dynamic_graph import DynamicGraph
from .dynamic.naive.naive_graph import NaiveGraph
from .dynamic.pcsr.pcsr_graph import PCSRGraph
from .static.static_graph import StaticGraph
from .stgraph_base import STGraphBase

__all__ = ["DynamicGraph", "NaiveGraph", "PCSRGraph", "StaticGraph", "STGraphBase"]
