"""The @compile operator API without a GPU: tracing, GIR, kernel dispatch, error behaviour."""
import pytest
import torch
from torch import nn

from stgraph_amd.compiler import STGraph
from stgraph_amd.compiler.backend.pytorch.torch_callback import STGraphBackendTorch
from stgraph_amd.compiler.dispatch import GatPlan, GcnPlan
from stgraph_amd.graph import StaticGraph


def _graph(n=4):
    return StaticGraph([(0, 1), (1, 0), (2, 1), (0, 2), (3, 2)], None, n, device="cpu")


class Probe(nn.Module):
    def __init__(self):
        super().__init__()
        self.leaky_relu = nn.LeakyReLU(0.2)
        self.stgraph = STGraph(STGraphBackendTorch())


def _plan(ctx, **kw):
    return ctx._setup_executor(**kw).plan


def test_gcn_vertex_function_maps_to_gcn_agg():
    m, g = Probe(), _graph()

    @m.stgraph.compile(gnn_module=m)
    def nb_compute(v):
        return sum([nb.h * nb.norm for nb in v.innbs]) * v.norm

    h = torch.randn(4, 16, requires_grad=True)
    plan = _plan(nb_compute, g=g, n_feats={"norm": torch.ones(4, 1), "h": h})
    assert isinstance(plan, GcnPlan) and (plan.x, plan.norm_src, plan.norm_dst, plan.ew) == ("h", "norm", "norm", None)
    prog = str(nb_compute._executor_cache.program)
    assert "AggSum(Mul(S:h[16],S:norm[1]))" in prog and "Mul(AggSum" in prog


def test_gcn_edge_weight_variant_and_cache_key():
    m, g = Probe(), _graph()
    h = torch.randn(4, 8, requires_grad=True)
    w = torch.rand(5, 1)

    @m.stgraph.compile(gnn_module=m)
    def nb_compute(v):       # noqa: F811  same name as the unweighted variant on purpose
        return sum([e.src.norm * e.src.h * e.edge_weight for e in v.inedges]) * v.norm

    p1 = _plan(nb_compute, g=g, n_feats={"norm": torch.ones(4, 1), "h": h}, e_feats={"edge_weight": w})
    assert isinstance(p1, GcnPlan) and p1.ew == "edge_weight"

    @m.stgraph.compile(gnn_module=m)
    def nb_compute(v):       # noqa: F811
        return sum([nb.h * nb.norm for nb in v.innbs]) * v.norm

    p2 = _plan(nb_compute, g=g, n_feats={"norm": torch.ones(4, 1), "h": h})
    assert p2.ew is None                                   # not aliased to the first-traced variant
    assert len(m.stgraph._ctx_map) == 1 and len(m.stgraph._ctx_map["nb_compute"]._executors) == 2


def test_gat_vertex_function_maps_to_gat_kernels():
    m, g = Probe(), _graph()

    @m.stgraph.compile(gnn_module=m)
    def nb_forward(v):
        embs = [nb.el + v.er for nb in v.innbs]
        coeff = [torch.exp(m.leaky_relu(emb - max(embs))) for emb in embs]
        s = sum(coeff)
        alpha = [c / s for c in coeff]
        feat_src = [nb.feat_src for nb in v.innbs]
        return sum([alpha[i] * feat_src[i] for i in range(len(feat_src))])

    feats = {"el": torch.randn(4, 2, 1), "er": torch.randn(4, 2, 1), "feat_src": torch.randn(4, 2, 8)}
    plan = _plan(nb_forward, g=g, n_feats=feats)
    assert isinstance(plan, GatPlan) and plan.slope == pytest.approx(0.2)
    assert (plan.el, plan.er, plan.feat) == ("el", "er", "feat_src")
    # the "softmax shift" is emb - emb: python's max() over a one-element list
    assert "Sub(Add(D:er[2x1],S:el[2x1]),Add(D:er[2x1],S:el[2x1]))" in str(nb_forward._executor_cache.program)


def test_other_vertex_functions_are_generated_and_api_errors_fail_loudly():
    m, g = Probe(), _graph()

    @m.stgraph.compile(gnn_module=m)
    def weird(v):
        return sum([nb.h * nb.h for nb in v.innbs])

    with pytest.raises(RuntimeError) as ei:                 # compiles (hiprtc, no GPU needed) but CPU tensors: no fallback
        weird(g=g, n_feats={"h": torch.randn(4, 3)})
    assert "no CPU fallback" in str(ei.value)
    assert weird._executors and next(iter(weird._executors.values())).plan.name == "generated"

    @m.stgraph.compile(gnn_module=m)
    def wrong_assoc(v):      # h * (norm * w) rounds differently from (norm * h) * w: never mapped onto the GCN kernel
        return sum([e.src.h * (e.src.norm * e.w) for e in v.inedges]) * v.norm

    with pytest.raises(RuntimeError):
        wrong_assoc(g=g, n_feats={"h": torch.randn(4, 3), "norm": torch.ones(4, 1)}, e_feats={"w": torch.ones(5, 1)})
    assert next(iter(wrong_assoc._executors.values())).plan.name == "generated"

    @m.stgraph.compile(gnn_module=m)
    def unsupported_op(v):
        return sum([torch.tanh(nb.h) for nb in v.innbs])

    with pytest.raises(NotImplementedError) as ei:          # outside the reference's op set
        unsupported_op(g=g, n_feats={"h": torch.randn(4, 3)})
    assert "tanh" in str(ei.value)

    @m.stgraph.compile(gnn_module=m)
    def nb_compute(v):
        return sum([nb.h * nb.norm for nb in v.innbs]) * v.norm

    with pytest.raises(NameError):                          # reference: compiler/stgraph.py:50-51
        nb_compute(n_feats={"h": torch.randn(4, 3), "norm": torch.ones(4, 1)})
    with pytest.raises(RuntimeError) as ei:                 # CPU tensors: no fallback
        nb_compute(g=g, n_feats={"h": torch.randn(4, 3), "norm": torch.ones(4, 1)})
    assert "no CPU fallback" in str(ei.value)
    assert next(iter(nb_compute._executors.values())).plan.name == "gcn_agg"
    # d/d(norm) is not one of the hand-written backward units: that signature gets generated kernels
    with pytest.raises(RuntimeError):
        nb_compute(g=g, n_feats={"h": torch.randn(4, 3), "norm": torch.ones(4, 1, requires_grad=True)})
    assert sorted(e.plan.name for e in nb_compute._executors.values()) == ["gcn_agg", "generated"]

    @m.stgraph.compile(gnn_module=m)
    def returns_nothing(v):
        return None

    with pytest.raises(NameError):
        returns_nothing(g=g, n_feats={"h": torch.randn(4, 3)})


def test_gcnconv_argument_checks_match_the_reference():
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    g = _graph()
    conv = GCNConv(3, 5)
    assert conv.weight.shape == (3, 5) and conv.bias.shape == (5,) and not conv.bias.any()
    with pytest.raises(KeyError):
        conv(g, torch.randn(4, 3))
    g.set_ndata("norm", torch.ones(4))
    with pytest.raises(ValueError):
        conv(g, torch.randn(4, 3))
    g.set_ndata("norm", torch.ones(3, 1))
    with pytest.raises(ValueError):
        conv(g, torch.randn(4, 3))


def test_state_dict_names_match_the_reference_layers():
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
    assert set(GATConv(4, 3, 2).state_dict()) == {"fc.weight", "attn_l", "attn_r"}
    names = set(TGCN(4, 8).state_dict())
    for gate in "zrh":
        assert {f"conv_{gate}.weight", f"conv_{gate}.bias", f"linear_{gate}.weight", f"linear_{gate}.bias"} <= names


def test_install_as_stgraph_aliases_the_reference_import_paths():
    import importlib
    import sys
    from stgraph_amd import compat
    compat.install_as_stgraph()
    try:
        from stgraph.compiler import STGraph as S2
        from stgraph.graph import StaticGraph as SG2
        from stgraph.nn.pytorch.static.gcn_conv import GCNConv as G2
        from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
        assert S2 is STGraph and SG2 is StaticGraph and G2 is GCNConv
        assert importlib.import_module("stgraph.compiler.backend.pytorch.torch_callback").STGraphBackendTorch
    finally:
        for k in [k for k in sys.modules if k == "stgraph" or k.startswith("stgraph.")]:
            del sys.modules[k]
