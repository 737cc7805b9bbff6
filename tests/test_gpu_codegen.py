"""Generated kernels (stgraph_amd/compiler/codegen.py: GIR -> HIP -> hiprtc) against a plain torch
restatement of the same vertex function (fp64 for values, autograd for gradients), through the
``@compile`` API on a StaticGraph and a NaiveGraph.  Tolerance 1e-4 (north star); the forward of
sum-only functions is additionally bit-exact against the sequential oracle."""
import numpy as np
import pytest
import torch

import stgraph_amd
from stgraph_amd.compiler import dispatch
from stgraph_amd.compiler.backend.pytorch.torch_callback import STGraphBackendTorch
from stgraph_amd.compiler.stgraph import STGraph
from tests.util import edges_by_eid, eval_vertex_function, random_graph

pytestmark = pytest.mark.gpu

H, D = 4, 8
FUNCTIONS = {
    # name: (vertex function, node features {name: shape}, edge features {name: shape}, differentiable names)
    "gin": (lambda v: sum([nb.h for nb in v.innbs]) + v.h, {"h": (16,)}, {}, ["h"]),
    "mean": (lambda v: sum([nb.h for nb in v.innbs]) / v.deg, {"h": (12,), "deg": (1,)}, {}, ["h"]),
    "edge_affine": (lambda v: sum([e.src.h * e.w + e.b for e in v.inedges]), {"h": (8,)}, {"w": (1,), "b": (8,)},
                    ["h", "w", "b"]),
    "exp_diff": (lambda v: sum([torch.exp(nb.h - v.h) for nb in v.innbs]), {"h": (8,)}, {}, ["h"]),
    "relu_gcn": (lambda v: torch.relu(sum([nb.h * nb.norm for nb in v.innbs]) * v.norm), {"h": (32,), "norm": (1,)}, {}, ["h"]),
    "two_level": (lambda v: sum([nb.h * sum([n2.g for n2 in v.innbs]) for nb in v.innbs]), {"h": (8,), "g": (8,)}, {}, ["h", "g"]),
    "softmax_like": (lambda v: sum([(torch.exp(e.src.a + e.dst.a) / sum([torch.exp(e2.src.a + e2.dst.a) for e2 in v.inedges]))
                                    * e.src.f for e in v.inedges]), {"a": (H, 1), "f": (H, D)}, {}, ["a", "f"]),
    "leaky_edge": (lambda v: sum([torch.nn.functional.leaky_relu(e.src.h * e.w - e.dst.h, 0.2) for e in v.inedges]),
                   {"h": (8,)}, {"w": (8,)}, ["h", "w"]),
    "two_outputs": (lambda v: (sum([nb.h for nb in v.innbs]), sum([nb.h * nb.h for nb in v.innbs]) * v.norm),
                    {"h": (8,), "norm": (1,)}, {}, ["h"]),
    "gcn_all_grads": (lambda v: sum([nb.h * nb.norm for nb in v.innbs]) * v.norm, {"h": (16,), "norm": (1,)}, {}, ["h", "norm"]),
    "wide": (lambda v: sum([nb.h * nb.norm for nb in v.innbs]) * v.norm + v.h, {"h": (300,), "norm": (1,)}, {}, ["h"]),
}


class _Mod(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.stgraph = STGraph(STGraphBackendTorch())


def _graph(cuda, kind, n, e, seed):
    from stgraph_amd.graph import NaiveGraph, StaticGraph
    src, dst = random_graph(seed, n, e)
    if kind == "static":
        return StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g = NaiveGraph([(src, dst)], n, device=cuda, sort_inplace=False)
    g.get_graph(0)
    return g


@pytest.mark.parametrize("kind", ["static", "naive"])
@pytest.mark.parametrize("name", sorted(FUNCTIONS))
def test_generated_kernels_match_torch(cuda, name, kind):
    fn, nshapes, eshapes, diff = FUNCTIONS[name]
    n, e = 400, 3000
    g = _graph(cuda, kind, n, e, 3)
    E = g.csr("fwd").column_indices.shape[0]
    gen = torch.Generator().manual_seed(7)
    mk = lambda rows, shape, key: (torch.rand((rows,) + shape, generator=gen) + 0.5 if key in ("deg", "norm")  # noqa: E731
                                   else torch.randn((rows,) + shape, generator=gen) * 0.5).to(cuda)
    nf = {k: mk(n, s, k).requires_grad_(k in diff) for k, s in nshapes.items()}
    ef = {k: mk(E, s, k).requires_grad_(k in diff) for k, s in eshapes.items()}
    mod = _Mod()
    fn_c = mod.stgraph.compile(gnn_module=mod)(fn)
    outs = fn_c(g=g, n_feats=nf, e_feats=ef)
    outs = outs if isinstance(outs, tuple) else (outs,)
    plan = fn_c._executor_cache.plan
    assert plan.name == "generated", plan
    # fp64 torch reference on the same graph
    src, dst = edges_by_eid(g.csr("fwd"))
    nf64 = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in nf.items()}
    ef64 = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in ef.items()}
    ref = eval_vertex_function(fn, src, dst, n, nf64, ef64)
    assert len(ref) == len(outs)
    for o, r in zip(outs, ref):
        assert o.shape == r.shape
        torch.testing.assert_close(o.double(), r, rtol=1e-4, atol=1e-4)
    Rs = [torch.randn(o.shape, generator=gen).to(cuda) for o in outs]
    sum((o * R).sum() for o, R in zip(outs, Rs)).backward()
    sum((r * R.double()).sum() for r, R in zip(ref, Rs)).backward()
    for k in diff:
        got = (nf.get(k) if k in nf else ef[k]).grad
        want = (nf64.get(k) if k in nf64 else ef64[k]).grad
        assert got is not None, k
        torch.testing.assert_close(got.double(), want, rtol=1e-4, atol=1e-4, msg=lambda m: f"grad {k}: {m}")
    st = fn_c._executor_cache.ts
    assert len(st.tensor_map_stack) == 0


def test_forced_generation_of_gcn_is_bit_exact_with_the_hand_written_kernel(cuda):
    """Same vertex function through both routes: identical sums in identical order."""
    from oracle import stg_oracle as orc
    from stgraph_amd.graph import StaticGraph
    n, e, F = 2000, 30000, 64
    src, dst = random_graph(5, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    og = orc.build_graph(src, dst, n)
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(n, F, generator=gen).to(cuda)
    norm = (torch.rand(n, 1, generator=gen) + 0.5).to(cuda)
    w = (torch.rand(e, 1, generator=gen) + 0.5).to(cuda)
    fn = lambda v: sum([nb.h * nb.norm for nb in v.innbs]) * v.norm                   # noqa: E731
    fn_w = lambda v: sum([e.src.norm * e.src.h * e.w for e in v.inedges]) * v.norm    # noqa: E731
    res = {}
    for forced in (False, True):
        dispatch.set_force_generated(forced)
        try:
            for tag, f, ef in (("plain", fn, {}), ("ew", fn_w, {"w": w})):
                mod = _Mod()
                x = h.clone().requires_grad_(True)
                fc = mod.stgraph.compile(gnn_module=mod)(f)
                out = fc(g=g, n_feats={"h": x, "norm": norm}, e_feats=ef)
                assert fc._executor_cache.plan.name == ("generated" if forced else "gcn_agg")
                (out * h).sum().backward()
                res[(forced, tag)] = (out.detach().cpu().numpy(), x.grad.cpu().numpy())
        finally:
            dispatch.set_force_generated(False)
    for tag, ew in (("plain", None), ("ew", w.cpu().numpy())):
        assert np.array_equal(res[(True, tag)][0], res[(False, tag)][0]), tag
        want = orc.gcn_agg(h.cpu().numpy(), norm.cpu().numpy(), norm.cpu().numpy(), og.fwd, ew=ew)
        assert np.array_equal(res[(True, tag)][0], want), tag
        np.testing.assert_allclose(res[(True, tag)][1], res[(False, tag)][1], rtol=1e-5, atol=1e-5)


def test_generated_kernels_on_a_pcsr_graph(cuda):
    """Generated kernels read the same graph surface: rows back to front, edge tensors by label-1."""
    from stgraph_amd.graph import PCSRGraph
    n, e = 300, 2500
    src, dst = random_graph(9, n, e)
    G = PCSRGraph([(src, dst)], n, device=cuda)
    G.get_graph(0)
    e = G.csr("fwd").column_indices.shape[0]
    fn, nshapes, eshapes, diff = FUNCTIONS["edge_affine"]
    gen = torch.Generator().manual_seed(3)
    nf = {"h": torch.randn(n, 8, generator=gen).to(cuda).requires_grad_(True)}
    ef = {"w": torch.randn(e, 1, generator=gen).to(cuda).requires_grad_(True),
          "b": torch.randn(e, 8, generator=gen).to(cuda).requires_grad_(True)}
    mod = _Mod()
    fc = mod.stgraph.compile(gnn_module=mod)(fn)
    out = fc(g=G, n_feats=nf, e_feats=ef)
    s, d = edges_by_eid(G.csr("fwd"))
    nf64 = {k: v.detach().double().requires_grad_(True) for k, v in nf.items()}
    ef64 = {k: v.detach().double().requires_grad_(True) for k, v in ef.items()}
    (ref,) = eval_vertex_function(fn, s, d, n, nf64, ef64)
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)
    R = torch.randn(out.shape, generator=gen).to(cuda)
    (out * R).sum().backward()
    (ref * R.double()).sum().backward()
    for k in ("h",):
        torch.testing.assert_close(nf[k].grad.double(), nf64[k].grad, rtol=1e-4, atol=1e-4)
    for k in ("w", "b"):
        torch.testing.assert_close(ef[k].grad.double(), ef64[k].grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["mean", "relu_gcn", "leaky_edge", "two_level"])
def test_generated_kernels_match_the_reference_compiler(cuda, name):
    """Against what the REFERENCE's compiler (fusion + autodiff + emitted kernels) computed for the same
    functions (tests/golden/codegen.npz): forward bit-identical, gradients to 1e-5."""
    from stgraph_amd.graph import StaticGraph
    from tests.test_codegen_cpu import GOLDEN_FUNCTIONS, golden_case
    d, n, _, _, nf, ef = golden_case(name, device=cuda, dtype=torch.float32)
    el = [(int(a), int(b)) for a, b in zip(d["src"], d["dst"])]
    g = StaticGraph(el, None, n, device=cuda)
    mod = _Mod()
    fc = mod.stgraph.compile(gnn_module=mod)(GOLDEN_FUNCTIONS[name])
    out = fc(g=g, n_feats=nf, e_feats=ef)
    assert fc._executor_cache.plan.name == "generated"
    assert np.array_equal(out.detach().cpu().numpy(), d[f"{name}_out"]), name
    (out * torch.from_numpy(d[f"{name}_R"]).to(cuda)).sum().backward()
    for k, t in {**nf, **ef}.items():
        if t.requires_grad and (name, k) != ("two_level", "g"):       # reference defect D16
            np.testing.assert_allclose(t.grad.cpu().numpy(), d[f"{name}_grad_{k}"], rtol=1e-5, atol=1e-5, err_msg=k)


def test_forced_generation_of_gat_forward_matches_the_hand_written_units(cuda):
    """GATConv's vertex function (gat_conv.py:48-56, incl. the emb - max([emb]) quirk) through the generator: same
    forward as the hand-written K0/K1 units.  The BACKWARD differs on purpose: the hand-written K2 reproduces the
    reference's emitted gradient, in which Sub passes +g to both operands, so attn_l / attn_r receive a
    gradient although d(emb - emb) = 0; the generator differentiates correctly (grad_el = grad_er = 0), which is
    what torch autograd gives for the same function ("softmax_like" above covers a non-degenerate softmax)."""
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    n, e = 500, 6000
    src, dst = random_graph(21, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    x0 = torch.randn(n, 12, device=cuda)
    res = {}
    for forced in (False, True):
        dispatch.set_force_generated(forced)
        try:
            torch.manual_seed(1)
            conv = GATConv(12, 8, 4).to(cuda)
            x = x0.clone().requires_grad_(True)
            out = conv(g, x)
            out.square().sum().backward()
            res[forced] = (out.detach(), conv.attn_l.grad.clone(), conv.fc.weight.grad.clone())
        finally:
            dispatch.set_force_generated(False)
    torch.testing.assert_close(res[False][0], res[True][0], rtol=1e-5, atol=1e-5)
    assert float(res[True][1].abs().max()) == 0.0 and float(res[False][1].abs().max()) > 0.0
    assert torch.isfinite(res[True][2]).all()


# ---------------------------------------------------------------- AggMax, Tensor.sum / .view, module parameters
from stgraph_amd.compiler import agg_max  # noqa: E402

EXTRA = {
    # name: (vertex function factory(mod), node features, edge features, differentiable names)
    "agg_max": (lambda mod: (lambda v: agg_max([nb.h for nb in v.innbs]) + v.h), {"h": (12,)}, {}, ["h"]),
    "agg_max_edge": (lambda mod: (lambda v: agg_max([e.src.h * e.w for e in v.inedges]) * v.norm), {"h": (8,), "norm": (1,)},
                     {"w": (8,)}, ["h", "w"]),
    "row_sum_keep": (lambda mod: (lambda v: sum([nb.f * (nb.f.sum(-1, keepdim=True)) for nb in v.innbs])),
                     {"f": (H, D)}, {}, ["f"]),
    "attn_like": (lambda mod: (lambda v: sum([(e.src.f * e.dst.f).sum(-1, keepdim=True) * e.src.f for e in v.inedges])),
                  {"f": (H, D)}, {}, ["f"]),
    "sum_of_aggregate": (lambda mod: (lambda v: sum([nb.f for nb in v.innbs]).sum(-1, keepdim=True) * v.f), {"f": (H, D)}, {}, ["f"]),
    "view_heads": (lambda mod: (lambda v: sum([nb.x.view(H, D) * nb.a for nb in v.innbs])), {"x": (H * D,), "a": (H, 1)}, {},
                   ["x", "a"]),
    "sum_nokeep": (lambda mod: (lambda v: sum([nb.f.sum(1) * nb.s for nb in v.innbs])), {"f": (H, D), "s": (H,)}, {}, ["f", "s"]),
    "param_scale": (lambda mod: (lambda v: sum([nb.h * mod.scale for nb in v.innbs]) + mod.shift), {"h": (16,)}, {}, ["h"]),
    "param_edge": (lambda mod: (lambda v: sum([torch.exp(e.src.h * mod.scale) * e.w for e in v.inedges])), {"h": (16,)},
                   {"w": (1,)}, ["h", "w"]),
}


class _PMod(_Mod):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.scale = torch.nn.Parameter(torch.rand(16, generator=g) + 0.5)
        self.shift = torch.nn.Parameter(torch.randn(16, generator=g))


@pytest.mark.parametrize("kind", ["static", "naive"])
@pytest.mark.parametrize("name", sorted(EXTRA))
def test_aggmax_sum_view_and_parameters_match_torch(cuda, name, kind):
    """The ops of the reference's Python surface / registry that its own pipeline cannot reach (AggMax: front end
    commented out, compiler/stgraph.py:6; Tensor.sum / .view: traced by torch_val.py:172-227 but without a code generator
    entry; module parameters inside the function: stgraph.py:126-173) against torch in fp64, values and gradients --
    including the gradients of the parameters."""
    make, nshapes, eshapes, diff = EXTRA[name]
    n, e = 400, 3000
    g = _graph(cuda, kind, n, e, 3)
    E = g.csr("fwd").column_indices.shape[0]
    gen = torch.Generator().manual_seed(7)
    mk = lambda rows, shape, key: (torch.rand((rows,) + shape, generator=gen) + 0.5 if key in ("deg", "norm")  # noqa: E731
                                   else torch.randn((rows,) + shape, generator=gen) * 0.5).to(cuda)
    nf = {k: mk(n, s, k).requires_grad_(k in diff) for k, s in nshapes.items()}
    ef = {k: mk(E, s, k).requires_grad_(k in diff) for k, s in eshapes.items()}
    mod = _PMod().to(cuda)
    fn = make(mod)
    fn_c = mod.stgraph.compile(gnn_module=mod)(fn)
    outs = fn_c(g=g, n_feats=nf, e_feats=ef)
    outs = outs if isinstance(outs, tuple) else (outs,)
    assert fn_c._executor_cache.plan.name == "generated"
    src, dst = edges_by_eid(g.csr("fwd"))
    nf64 = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in nf.items()}
    ef64 = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in ef.items()}
    pf64 = {f"param{id(p):x}": p.detach().double().requires_grad_(True) for p in (mod.scale, mod.shift)}
    ref = eval_vertex_function(fn, src, dst, n, nf64, ef64, pf64)
    for o, r in zip(outs, ref):
        assert o.shape == r.shape, (o.shape, r.shape)
        fin = torch.isfinite(r)                        # agg_max: -inf on vertices without in-edges, on both sides
        assert torch.equal(torch.isfinite(o), fin)
        torch.testing.assert_close(o.double()[fin], r[fin], rtol=1e-4, atol=1e-4)
    Rs = [torch.randn(o.shape, generator=gen).to(cuda) for o in outs]
    fin = [torch.isfinite(r) for r in ref]
    sum((torch.where(f, o, torch.zeros_like(o)) * R).sum() for o, R, f in zip(outs, Rs, fin)).backward()
    sum((torch.where(f, r, torch.zeros_like(r)) * R.double()).sum() for r, R, f in zip(ref, Rs, fin)).backward()
    for k in diff:
        got = (nf.get(k) if k in nf else ef[k]).grad
        want = (nf64.get(k) if k in nf64 else ef64[k]).grad
        assert got is not None, k
        torch.testing.assert_close(got.double(), want, rtol=1e-4, atol=1e-4, msg=lambda m: f"grad {k}: {m}")
    if name.startswith("param"):
        used = [(mod.scale, pf64[f"param{id(mod.scale):x}"])] + ([(mod.shift, pf64[f"param{id(mod.shift):x}"])] if name == "param_scale" else [])
        for p, p64 in used:
            assert p.grad is not None
            torch.testing.assert_close(p.grad.double(), p64.grad, rtol=1e-4, atol=1e-3)
    assert len(fn_c._executor_cache.ts.tensor_map_stack) == 0
