"""stgraph_amd -- MI355X (gfx950) native implementation of STGraph's Seastar hot path.

Public surface mirrors the reference package layout:
    stgraph_amd.graph            StaticGraph, NaiveGraph, DynamicGraph, STGraphBase
    stgraph_amd.compiler         STGraph (``@compile`` vertex-centric operator API)
    stgraph_amd.nn.pytorch       static.gcn_conv.GCNConv, static.gat_conv.GATConv, temporal.tgcn.TGCN
``stgraph_amd.compat.install_as_stgraph()`` registers the same modules under the
``stgraph.*`` names so unmodified reference model code imports them.

Importing this package loads ``stgraph_amd/lib/libstgraph_hip.so`` and fails if it
has not been built: there is no CPU fallback.
"""
from . import _C  # noqa: F401  (loads the HIP library; raises ImportError when it is missing)
from .kernels import reference_compat, set_reference_compat

__version__ = "0.1.0"
__all__ = ["reference_compat", "set_reference_compat", "__version__"]
