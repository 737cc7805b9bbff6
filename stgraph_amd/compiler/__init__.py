"""Seastar vertex-centric front end (reference: stgraph/compiler/__init__.py)."""
from .stgraph import Context, STGraph

__all__ = ["Context", "STGraph"]
