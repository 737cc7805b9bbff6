"""GPU parity of the drop-in layers against golden vectors produced by the REFERENCE layers
(GCNConv, GATConv, TGCN on StaticGraph / NaiveGraph), through the @compile operator API."""
import numpy as np
import pytest
import torch

import stgraph_amd
from tests.util import GAT_SHAPES, GCN_WIDTHS, golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _edges(d, prefix=""):
    return [(int(a), int(b)) for a, b in zip(d[prefix + "src"], d[prefix + "dst"])]


@pytest.fixture(autouse=True)
def _compat_on():
    # the golden vectors carry reference defect D1 (zero tail for F < 64 not a power of two)
    stgraph_amd.set_reference_compat(True)
    yield
    stgraph_amd.set_reference_compat(False)


@pytest.mark.parametrize("gname", ["static", "naive"])
def test_gcnconv_matches_reference_layer(cuda, gname):
    from stgraph_amd.graph import NaiveGraph, StaticGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    d = golden("gcn.npz")
    n = int(d["num_nodes"])
    el = _edges(d)
    g = StaticGraph(el, [1.0] * len(el), n, device=cuda) if gname == "static" else NaiveGraph([el], n, device=cuda)
    assert g.graph_type() == ("csr_unsorted" if gname == "static" else "csr")
    g.set_ndata("norm", _t(d[f"{gname}_norm"], cuda))
    w = _t(d["edge_weight_by_eid"], cuda)
    for F in GCN_WIDTHS:
        for use_ew in (False, True):
            tag = f"{gname}_F{F}_{'ew' if use_ew else 'now'}"
            conv = GCNConv(F, F, bias=False).to(cuda)
            with torch.no_grad():
                conv.weight.copy_(torch.eye(F))
            x = _t(d[tag + "_x"], cuda).requires_grad_(True)
            if gname == "naive":
                g.get_graph(0)
            out = conv(g, x, edge_weight=w if use_ew else None)
            (out * _t(d[tag + "_R"], cuda)).sum().backward()
            assert np.array_equal(out.detach().cpu().numpy(), d[tag + "_out"]), tag
            assert np.array_equal(x.grad.cpu().numpy(), d[tag + "_grad_x"]), tag
            st = conv.stgraph._ctx_map["nb_compute"]._executor_cache.ts
            assert len(st.tensor_map_stack) == 0 and len(st.graph_timestamp_stack) == 0


@pytest.mark.parametrize("H,D", GAT_SHAPES)
def test_gatconv_matches_reference_layer(cuda, H, D):
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    d = golden("gat.npz")
    n = int(d["num_nodes"])
    el = _edges(d)
    g = StaticGraph(el, [1.0] * len(el), n, device=cuda)
    tag = f"H{H}_D{D}"
    conv = GATConv(d[tag + "_x"].shape[1], D, H).to(cuda)
    with torch.no_grad():
        conv.fc.weight.copy_(_t(d[tag + "_fc_weight"], cuda))
        conv.attn_l.copy_(_t(d[tag + "_attn_l"], cuda))
        conv.attn_r.copy_(_t(d[tag + "_attn_r"], cuda))
    x = _t(d[tag + "_x"], cuda).requires_grad_(True)
    out = conv(g, x)
    (out * _t(d[tag + "_R"], cuda)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), d[tag + "_out"], rtol=TOL, atol=TOL)
    for name, got in (("grad_x", x.grad), ("grad_fc_weight", conv.fc.weight.grad),
                      ("grad_attn_l", conv.attn_l.grad), ("grad_attn_r", conv.attn_r.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), d[f"{tag}_{name}"], rtol=TOL, atol=TOL, err_msg=name)


class TGCNModel(torch.nn.Module):
    """tests/scripts/v1_1_0/temporal_tgcn_dataloaders model shape: TGCN -> ReLU -> Linear."""

    def __init__(self, fin, hid, out):
        super().__init__()
        from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
        self.temporal = TGCN(fin, hid)
        self.linear = torch.nn.Linear(hid, out)

    def forward(self, g, x, edge_weight, hidden):
        h = self.temporal(g, x, edge_weight, hidden)
        return self.linear(torch.relu(h)), h


def _load_params(model, d, prefix, dev):
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(_t(d[prefix + k], dev))


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("B", [3, 6])
def test_tgcn_bptt_matches_reference(cuda, B, fuse):
    from stgraph_amd.graph import StaticGraph
    d = golden("tgcn.npz")
    n, T = int(d["num_nodes"]), d["feats"].shape[0]
    el = _edges(d)
    w = _t(d["edge_weight_by_eid"], cuda)
    g = StaticGraph(el, d["edge_weight_by_eid"].reshape(-1).tolist(), n, device=cuda)
    g.set_ndata("norm", _t(d["norm"], cuda))
    feats, targets = _t(d["feats"], cuda), _t(d["targets"], cuda)
    stgraph_amd.set_reference_compat(not fuse)     # widths here are powers of two: D1 cannot trigger
    model = TGCNModel(feats.shape[2], 16, 1).to(cuda)
    model.temporal.fuse_gates = fuse
    _load_params(model, d, f"B{B}_param_", cuda)
    hs, costs = [], []
    for w0 in range(0, T, B):
        model.zero_grad()
        hidden, cost = None, 0
        for t in range(w0, w0 + B):
            y, hidden = model(g, feats[t], w, hidden)
            cost = cost + torch.mean((y - targets[t]) ** 2)
            hs.append(hidden.detach())
        cost = cost / (B + 1)
        cost.backward()
        costs.append(cost.detach())
        for k, p in model.named_parameters():
            np.testing.assert_allclose(p.grad.cpu().numpy(), d[f"B{B}_w{w0}_grad_{k}"], rtol=TOL, atol=TOL,
                                       err_msg=f"window {w0} {k}")
    np.testing.assert_allclose(torch.stack(hs).cpu().numpy(), d[f"B{B}_hidden"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(torch.stack(costs).cpu().numpy(), d[f"B{B}_cost"], rtol=TOL, atol=TOL)
    for conv in (model.temporal.conv_z, model.temporal.conv_r, model.temporal.conv_h):
        if "nb_compute" in conv.stgraph._ctx_map:
            for ex in conv.stgraph._ctx_map["nb_compute"]._executors.values():
                assert len(ex.ts.tensor_map_stack) == 0
    used = [c for c in (model.temporal.conv_z, model.temporal.conv_r, model.temporal.conv_h)
            if "nb_compute" in c.stgraph._ctx_map]
    assert len(used) == (1 if fuse else 3)        # fused gates: ONE aggregation launch per step


def test_tgcn_adam_training_loop_matches_reference(cuda):
    from stgraph_amd.graph import StaticGraph
    d = golden("tgcn.npz")
    n, T, B = int(d["num_nodes"]), d["feats"].shape[0], 3
    el = _edges(d)
    w = _t(d["edge_weight_by_eid"], cuda)
    g = StaticGraph(el, None, n, device=cuda)
    g.set_ndata("norm", _t(d["norm"], cuda))
    feats, targets = _t(d["feats"], cuda), _t(d["targets"], cuda)
    model = TGCNModel(feats.shape[2], 16, 1).to(cuda)
    _load_params(model, d, "train_param0_", cuda)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses, sums = [], []
    for _ in range(2):
        for w0 in range(0, T, B):
            opt.zero_grad()
            hidden, cost = None, 0
            for t in range(w0, w0 + B):
                y, hidden = model(g, feats[t], w, hidden)
                cost = cost + torch.mean((y - targets[t]) ** 2)
            cost = cost / (B + 1)
            cost.backward()
            opt.step()
            losses.append(cost.detach())
            sums.append(torch.stack([p.detach().double().sum() for p in model.parameters()]))
    np.testing.assert_allclose(torch.stack(losses).cpu().numpy(), d["train_losses"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(torch.stack(sums).cpu().numpy(), d["train_param_sums"], rtol=1e-3, atol=1e-3)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), d["train_paramT_" + k], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("resident", [True, False])
def test_naive_graph_tgcn_bptt_matches_reference(cuda, resident):
    from stgraph_amd.graph import NaiveGraph
    d = golden("naive_tgcn.npz")
    n, T = int(d["num_nodes"]), int(d["T"])
    snaps = [_edges(d, f"t{t}_") for t in range(T)]
    G = NaiveGraph(snaps, n, device=cuda, resident=resident, max_cached=None if resident else 2)
    for t in range(T):      # per-snapshot CSRs are bit-exact (node_ids: permutation with monotone degrees)
        for side in ("fwd", "bwd"):
            c = G.csr(side, t)
            for k in ("row_offset", "column_indices", "eids"):
                assert np.array_equal(getattr(c, k).cpu().numpy(), d[f"t{t}_{side}_{k}"]), (t, side, k)
    feats, targets = _t(d["feats"], cuda), _t(d["targets"], cuda)
    model = TGCNModel(feats.shape[2], 16, 1).to(cuda)
    _load_params(model, d, "param_", cuda)
    G.reset_graph()
    hidden, cost, hs = None, 0, []
    for t in range(T):
        G.get_graph(t)
        if G.get_ndata("norm") is None:
            deg = torch.from_numpy(G.in_degrees()).float()
            norm = torch.pow(deg, -0.5)
            norm[torch.isinf(norm)] = 0
            G.set_ndata("norm", norm.unsqueeze(1).to(cuda))
        np.testing.assert_array_equal(G.get_ndata("norm").cpu().numpy(), d[f"t{t}_norm"])
        y, hidden = model(G, feats[t], None, hidden)
        cost = cost + torch.mean((y - targets[t]) ** 2)
        hs.append(hidden.detach())
    cost = cost / (T + 1)
    cost.backward()
    assert G.current_timestamp == 0
    np.testing.assert_allclose(torch.stack(hs).cpu().numpy(), d["hidden"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cost.item(), float(d["cost"]), rtol=TOL, atol=TOL)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), d["grad_" + k], rtol=TOL, atol=TOL, err_msg=k)
    # a second window can start again from an earlier snapshot after reset
    G.reset_graph()
    G.get_graph(1)
    assert G.current_timestamp == 1


def test_fused_cell_matches_unfused_tgcn(cuda):
    """cell.TGCNCellFn (fused row-local stages) == the torch formulation: outputs and every gradient."""
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
    from tests.util import random_graph
    n, e = 5000, 60000
    src, dst = random_graph(78, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    g.set_ndata("norm", torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1))
    torch.manual_seed(3)
    m = TGCN(16, 32).to(cuda)
    with torch.no_grad():                                  # make the clamp bite on a few entries
        m.conv_z.bias[:4] = 2e6
        m.conv_h.bias[-3:] = -3e6
    x = torch.randn(n, 16, device=cuda, requires_grad=True)
    w = torch.rand(len(src), 1, device=cuda) + 0.5
    res = []
    for fuse in (True, False):
        m.fuse_cell = fuse
        m.zero_grad()
        x.grad = None
        H = None
        for _ in range(3):
            H = m(g, x, w, H)
        (H * torch.linspace(-1, 1, 32, device=cuda)).sum().backward()
        res.append((H.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5)
    for k in res[0][2]:
        torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=1e-4, atol=1e-4, msg=k)


def test_fused_gates_equal_separate_gates(cuda):
    """One width-3H aggregation == three width-H aggregations, column for column."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.temporal.tgcn import TGCN
    from tests.util import random_graph
    n, e = 4000, 50000
    src, dst = random_graph(77, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    f = g.csr("fwd")
    deg = (f.row_offset[1:] - f.row_offset[:-1]).float()
    norm = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1)
    g.set_ndata("norm", norm)
    h3 = torch.randn(n, 192, device=cuda)
    fused = kernels.gcn_agg(h3, norm, norm, f)
    for k in range(3):
        part = kernels.gcn_agg(h3[:, 64 * k:64 * (k + 1)].contiguous(), norm, norm, f)
        assert torch.equal(fused[:, 64 * k:64 * (k + 1)], part)
    torch.manual_seed(0)
    m = TGCN(32, 64).to(cuda)
    x = torch.randn(n, 32, device=cuda)
    w = torch.rand(len(src), 1, device=cuda) + 0.5
    outs = []
    for fuse in (True, False):
        m.fuse_gates = fuse
        m.zero_grad()
        H = m(g, x, w, None)
        H = m(g, x, w, H)
        (H ** 2).sum().backward()
        outs.append((H.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    for a, b in zip(outs[0][1], outs[1][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


def test_captured_window_replays_the_eager_loop(cuda):
    """HIP-graph replay of a BPTT window == the eager loop (losses and parameters after 2 epochs)."""
    from stgraph_amd import temporal
    from stgraph_amd.graph import StaticGraph
    from tests.util import random_graph
    # n >= 4096 and feat 16 -> 48: the native path (fused step node, deferred weight gradients) is what gets captured
    n, e, feat, hid, T, B = 5000, 50000, 16, 16, 12, 4
    src, dst = random_graph(5, n, e)
    e = len(src)
    results = []
    for captured in (False, True):
        g = StaticGraph((src.copy(), dst.copy()), None, n, device=cuda, sort_inplace=False)
        g.set_ndata("norm", temporal.in_degree_norm(g))
        gen = torch.Generator(device=cuda).manual_seed(9)
        ew = torch.rand(e, 1, device=cuda, generator=gen) + 0.5
        targets = torch.randn(T, n, 1, device=cuda, generator=gen)
        torch.manual_seed(1)
        model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        bucket = temporal.GradBucket(model.parameters())
        losses = []
        if captured:
            cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, feat)
        for ep in range(2):
            if captured:
                losses += temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, feat, epoch=ep)
            else:
                losses += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=ep)
        results.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    torch.testing.assert_close(results[0][0], results[1][0], rtol=1e-5, atol=1e-7)
    for a, b in zip(results[0][1], results[1][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_deferred_weight_grads_equal_per_step_grads(cuda):
    """nn.deferred: one launch per parameter per backward pass == per-step gradients through autograd
    (TGCN over a 5-step window with the fused step node, head Linears included)."""
    from stgraph_amd import temporal
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from tests.util import random_graph
    n, e, feat, hid, B = 6000, 70000, 32, 64, 5
    src, dst = random_graph(8, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", temporal.in_degree_norm(g))
    gen = torch.Generator(device=cuda).manual_seed(2)
    ew = torch.rand(len(src), 1, device=cuda, generator=gen) + 0.5
    targets = torch.randn(B, n, 1, device=cuda, generator=gen)
    x0 = torch.randn(n, feat, device=cuda, generator=gen)
    torch.manual_seed(4)
    model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
    grads = []
    try:
        for defer in (True, False):
            SF.set_deferred_weight_grads(defer)
            model.zero_grad()
            cost, hidden, y = 0, None, x0
            for t in range(B):
                y_out, y, hidden = model(g, y, ew, hidden)
                cost = cost + torch.mean((y_out - targets[t]) ** 2)
            (cost / (B + 1)).backward()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
            assert all(p.grad is not None for p in model.parameters())
    finally:
        SF.set_deferred_weight_grads(True)
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        scale = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, k


def test_strided_and_missing_gradients_are_handled(cuda):
    """Reference defect D6: the reference reads grad tensors through raw data_ptr, so an expanded
    (stride-0) gradient such as the one ``out.sum().backward()`` produces is read as garbage.  Here it
    is made contiguous; result must equal the dense formula."""
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    from tests.util import random_graph
    stgraph_amd.set_reference_compat(False)                  # F = 12: all columns (D1 would stop at 8)
    n, e, F = 500, 4000, 12
    src, dst = random_graph(31, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    norm = torch.rand(n, 1, device=cuda) + 0.5
    g.set_ndata("norm", norm)
    conv = GCNConv(F, F, bias=False).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(F))
    x = torch.randn(n, F, device=cuda, requires_grad=True)
    conv(g, x).sum().backward()                               # grad_out is an expanded scalar (stride 0)
    want = kernels.gcn_agg(torch.ones(n, F, device=cuda), norm, norm, g.csr("bwd"))
    torch.testing.assert_close(x.grad, want, rtol=1e-6, atol=1e-6)


def test_gcnconv_with_duplicate_edges_and_self_loops(cuda):
    """Multigraph input: the CSR keeps duplicates and the layer sums over all of them."""
    from oracle import stg_oracle as orc
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    stgraph_amd.set_reference_compat(False)
    src = np.array([0, 0, 1, 2, 2, 2, 3, 3], np.int32)
    dst = np.array([1, 1, 1, 2, 0, 0, 3, 0], np.int32)
    g = StaticGraph(np.stack([src, dst], 1), None, 4, device=cuda, sort_inplace=False)
    assert g.get_num_edges() == 6 and g.csr("fwd").num_edges == 8
    norm_np = np.array([[0.5], [1.0], [2.0], [0.25]], np.float32)
    g.set_ndata("norm", torch.from_numpy(norm_np).to(cuda))
    conv = GCNConv(5, 5, bias=False).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(5))
    x_np = np.arange(20, dtype=np.float32).reshape(4, 5) / 7
    out = conv(g, torch.from_numpy(x_np).to(cuda)).detach().cpu().numpy()
    og = orc.build_graph(src, dst, 4)
    assert np.array_equal(out, orc.gcn_agg(x_np, norm_np, norm_np, og.fwd))


def test_gatconv_on_naive_graph_snapshots(cuda):
    """GAT through the per-snapshot CSR path ('csr' type, node_ids order, timestamp stack)."""
    from oracle import stg_oracle as orc
    from stgraph_amd.graph import NaiveGraph
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    d = golden("naive_tgcn.npz")
    n, T = int(d["num_nodes"]), int(d["T"])
    snaps = [_edges(d, f"t{t}_") for t in range(T)]
    G = NaiveGraph(snaps, n, device=cuda)
    stgraph_amd.set_reference_compat(False)
    torch.manual_seed(0)
    conv = GATConv(6, 8, 2).to(cuda)
    xs = [torch.randn(n, 6, device=cuda, requires_grad=True) for _ in range(T)]
    G.reset_graph()
    total = 0
    outs = []
    for t in range(T):
        G.get_graph(t)
        o = conv(G, xs[t])
        outs.append(o)
        total = total + (o * (t + 1)).sum()
    total.backward()
    assert G.current_timestamp == 0
    for t in range(T):                                       # forward of every snapshot == oracle on that snapshot
        og = orc.build_graph(d[f"t{t}_src"], d[f"t{t}_dst"], n)
        feat = conv.fc(xs[t]).view(-1, 2, 8)
        el = (feat * conv.attn_l).sum(-1).unsqueeze(-1)
        er = (feat * conv.attn_r).sum(-1).unsqueeze(-1)
        A0, S0 = orc.gat_k0(el.detach().cpu().numpy(), er.detach().cpu().numpy(), og.fwd, og.num_edges, use_node_ids=True)
        o0 = orc.gat_k1(A0, S0, feat.detach().cpu().numpy(), og.fwd, use_node_ids=True)
        np.testing.assert_allclose(outs[t].detach().cpu().numpy(), o0, rtol=1e-5, atol=1e-6)
        assert xs[t].grad is not None and bool(torch.isfinite(xs[t].grad).all())


@pytest.mark.parametrize("N,K,M", [(2708, 16, 7), (1, 1, 1), (3072, 8, 16), (777, 16, 8), (100, 3, 5), (20000, 12, 12), (65536, 16, 16)])
def test_small_dense_layer_backward_in_one_launch(cuda, N, K, M):
    """stg_mm_bwd_small (gx = g W^T and gw = x^T g of y = x W for a layer one workgroup holds) against fp64, no worse than the two
    library GEMMs it replaces; run twice: identical; SF.mm takes it (launch record) and can be switched off."""
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    gen = torch.Generator(device=cuda).manual_seed(N + K + M)
    x = torch.randn(N, K, device=cuda, generator=gen)
    w = torch.randn(K, M, device=cuda, generator=gen)
    g = torch.randn(N, M, device=cuda, generator=gen)
    assert kernels.mm_bwd_small_usable(g, x, w)
    gx, gw = kernels.mm_bwd_small(g, x, w)
    gx2, gw2 = kernels.mm_bwd_small(g, x, w)
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2)
    for got, lib, ref in ((gx, g @ w.t(), g.double() @ w.double().t()), (gw, x.t() @ g, x.double().t() @ g.double())):
        scale = float(ref.abs().max()) + 1e-30
        e_new, e_lib = float((got.double() - ref).abs().max()) / scale, float((lib.double() - ref).abs().max()) / scale
        assert e_new <= max(2 * e_lib, 2e-6), (e_new, e_lib)
    out = []
    for on in (True, False):
        kernels.set_mm_bwd_small(on)
        try:
            recs = []
            kernels.enable_launch_timing(recs)
            xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            SF.mm(xa, wa).backward(g)
            kernels.enable_launch_timing(None)
            assert ("mm_bwd_small" in [r[0] for r in recs]) == on
            out.append((xa.grad, wa.grad))
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_mm_bwd_small(True)
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-4, atol=1e-4 * float(out[1][1].abs().max()))
    assert not kernels.mm_bwd_small_usable(torch.randn(70000, 7, device=cuda), torch.randn(70000, 16, device=cuda), torch.randn(16, 7, device=cuda))
    assert not kernels.mm_bwd_small_usable(torch.randn(100, 17, device=cuda), torch.randn(100, 16, device=cuda), torch.randn(16, 17, device=cuda))


def test_small_layer_below_a_relu_layer_masks_and_sums_in_its_backward_launch(cuda):
    """Two GCNConv layers of Cora's widths on a small graph: the second layer's backward launch (stg_mm_bwd_small) applies the first
    layer's ReLU mask and leaves its bias gradient on the tensor, so no bias_act_bwd launch runs -- every parameter gradient and the
    input gradient equal the unfused path's; with a hook on the hidden activation the fusion steps aside and the hook sees torch's value."""
    import torch.nn.functional as F
    from stgraph_amd import kernels
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    from tests.util import gcn_norm, random_graph
    stgraph_amd.set_reference_compat(False)                  # (the file's fixture turns it on: the layer tail is off then)
    n, e = 2708, 10556
    src, dst = random_graph(5, n, e)
    g = StaticGraph((src, dst), None, n, device=cuda, sort_inplace=False)
    g.set_ndata("norm", torch.from_numpy(gcn_norm(np.bincount(dst, minlength=n))).to(cuda))
    torch.manual_seed(3)
    l1, l2 = GCNConv(40, 16, activation=F.relu).to(cuda), GCNConv(16, 7, activation=None).to(cuda)
    x0 = torch.randn(n, 40, device=cuda)
    R = torch.randn(n, 7, device=cuda)
    res = []
    for mode in ("fused", "unfused", "hooked"):
        kernels.set_mm_bwd_small(mode != "unfused")
        try:
            for p in list(l1.parameters()) + list(l2.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_(True)
            recs, seen = [], []
            kernels.enable_launch_timing(recs)
            h = l1(g, x)
            if mode == "hooked":
                h.register_hook(lambda t: seen.append(t.clone()))
            l2(g, h).backward(R)
            kernels.enable_launch_timing(None)
            names = [r[0] for r in recs]
            assert ("mm_bwd_small" in names) == (mode != "unfused"), names
            # (one bias_act_bwd is the second layer's own bias gradient; the other the first layer's ReLU mask + bias gradient)
            assert names.count("bias_act_bwd") == (1 if mode == "fused" else 2), names
            res.append([x.grad.clone()] + [p.grad.clone() for p in list(l1.parameters()) + list(l2.parameters())] + seen)
        finally:
            kernels.enable_launch_timing(None)
            kernels.set_mm_bwd_small(True)
    for a, b in zip(res[0], res[1]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max() + 1))
    for a, b in zip(res[2][:-1], res[1]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max() + 1))
    assert len(res[2]) == len(res[1]) + 1                       # the hook ran once and saw the UNMASKED gradient of h
    kernels.set_mm_bwd_small(False)
    try:
        x = x0.clone().requires_grad_(True)
        h = l1(g, x)
        h.retain_grad()
        l2(g, h).backward(R)
        torch.testing.assert_close(res[2][-1], h.grad, rtol=2e-4, atol=1e-5)
    finally:
        kernels.set_mm_bwd_small(True)


@pytest.mark.parametrize("N,Ka,Mb", [(2708, 1433, 16), (1, 1, 1), (3000, 17, 7), (65536, 40, 16), (5, 300, 3)])
def test_small_graph_weight_gradient_in_one_launch(cuda, N, Ka, Mb):
    """stg_gemm_tn_small_f32 (a^T b, a of any width, b <= 16 columns, N <= 65536) against fp64, no worse than the library GEMM; twice
    the same; taken by SF.mm's backward for a layer whose input carries no gradient."""
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF
    gen = torch.Generator(device=cuda).manual_seed(N + Ka)
    a = torch.randn(N, Ka, device=cuda, generator=gen)
    b = torch.randn(N, Mb, device=cuda, generator=gen)
    assert kernels.gemm_tn_small_usable(a, b)
    c, c2 = kernels.gemm_tn_small(a, b), kernels.gemm_tn_small(a, b)
    assert torch.equal(c, c2)
    ref = a.double().t() @ b.double()
    scale = float(ref.abs().max()) + 1e-30
    e_new, e_lib = float((c.double() - ref).abs().max()) / scale, float(((a.t() @ b).double() - ref).abs().max()) / scale
    assert e_new <= max(2 * e_lib, 2e-6), (e_new, e_lib)
    w = torch.randn(Ka, Mb, device=cuda, generator=gen).requires_grad_(True)
    recs = []
    kernels.enable_launch_timing(recs)
    try:
        SF.mm(a, w).backward(b)
    finally:
        kernels.enable_launch_timing(None)
    # (from 4096 rows the split-K contraction of the large graphs takes the weight gradient: functional.MIN_K)
    assert ("gemm_tn_small" in [r[0] for r in recs]) == (N < 4096), [r[0] for r in recs]
    if N < 4096:
        torch.testing.assert_close(w.grad.double(), ref, rtol=1e-4, atol=1e-5 * scale)
    assert not kernels.gemm_tn_small_usable(torch.randn(10, 5, device=cuda), torch.randn(10, 17, device=cuda))
