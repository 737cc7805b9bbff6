"""Expose stgraph_amd under the reference's import names (``stgraph.graph`` ...)."""
from __future__ import annotations

import importlib
import sys

_ALIASES = {
    "stgraph": "stgraph_amd",
    "stgraph.graph": "stgraph_amd.graph",
    "stgraph.graph.stgraph_base": "stgraph_amd.graph.stgraph_base",
    "stgraph.graph.static": "stgraph_amd.graph.static",
    "stgraph.graph.static.static_graph": "stgraph_amd.graph.static.static_graph",
    "stgraph.graph.static.csr": "stgraph_amd.graph.static.csr",
    "stgraph.graph.dynamic": "stgraph_amd.graph.dynamic",
    "stgraph.graph.dynamic.dynamic_graph": "stgraph_amd.graph.dynamic.dynamic_graph",
    "stgraph.graph.dynamic.naive": "stgraph_amd.graph.dynamic.naive",
    "stgraph.graph.dynamic.naive.naive_graph": "stgraph_amd.graph.dynamic.naive.naive_graph",
    "stgraph.graph.dynamic.pcsr": "stgraph_amd.graph.dynamic.pcsr",
    "stgraph.graph.dynamic.pcsr.pcsr": "stgraph_amd.graph.dynamic.pcsr.pcsr",
    "stgraph.graph.dynamic.pcsr.pcsr_graph": "stgraph_amd.graph.dynamic.pcsr.pcsr_graph",
    "stgraph.graph.dynamic.gpma": "stgraph_amd.graph.dynamic.gpma",
    "stgraph.graph.dynamic.gpma.gpma": "stgraph_amd.graph.dynamic.gpma.gpma",
    "stgraph.graph.dynamic.gpma.gpma_graph": "stgraph_amd.graph.dynamic.gpma.gpma_graph",
    "stgraph.compiler": "stgraph_amd.compiler",
    "stgraph.compiler.node": "stgraph_amd.compiler.node",
    "stgraph.compiler.backend": "stgraph_amd.compiler.backend",
    "stgraph.compiler.backend.callback": "stgraph_amd.compiler.backend.callback",
    "stgraph.compiler.backend.pytorch": "stgraph_amd.compiler.backend.pytorch",
    "stgraph.compiler.backend.pytorch.torch_callback": "stgraph_amd.compiler.backend.pytorch.torch_callback",
    "stgraph.nn": "stgraph_amd.nn",
    "stgraph.nn.pytorch": "stgraph_amd.nn.pytorch",
    "stgraph.nn.pytorch.static": "stgraph_amd.nn.pytorch.static",
    "stgraph.nn.pytorch.static.gcn_conv": "stgraph_amd.nn.pytorch.static.gcn_conv",
    "stgraph.nn.pytorch.static.gat_conv": "stgraph_amd.nn.pytorch.static.gat_conv",
    "stgraph.nn.pytorch.temporal": "stgraph_amd.nn.pytorch.temporal",
    "stgraph.nn.pytorch.temporal.tgcn": "stgraph_amd.nn.pytorch.temporal.tgcn",
    "stgraph.utils": "stgraph_amd.utils",
    "stgraph.utils.constants": "stgraph_amd.utils.constants",
}


def install_as_stgraph() -> None:
    """Make ``import stgraph...`` resolve to this package (refuses to shadow a real stgraph)."""
    existing = sys.modules.get("stgraph")
    if existing is not None and not getattr(existing, "__name__", "").startswith("stgraph_amd"):
        raise RuntimeError("a different 'stgraph' package is already imported")
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
