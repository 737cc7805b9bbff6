"""Seastar vertex-centric front end (reference: stgraph/compiler/__init__.py)."""
from .stgraph import Context, STGraph
from .val import agg_max

__all__ = ["Context", "STGraph", "agg_max"]
