"""GPU parity of the GPMA counterpart (``gpma`` module + ``GPMAGraph`` on csrc/edge_store.hip).
PARITY UNPINNED upstream (no gpma.so): checked against oracle/stg_gpma_oracle.c's restatement of the
reference's kernel contract on a gapped array, against the reference-recorded NaiveGraph run of the same
snapshots (the quantity a GPMA graph represents), and -- at bench scale -- through properties."""
import numpy as np
import pytest
import torch

from oracle.stg_gpma_oracle import OracleGPMA
from tests.test_gpu_layers import TGCNModel, _edges, _load_params, _t
from tests.test_host_gpma import (check_emit_equals_static_builder, check_graph_protocol,
                                  check_module_against_oracle, snapshots)
from tests.util import gcn_norm, golden, random_graph

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_gpma_module_device(cuda):
    check_module_against_oracle(cuda)


def test_gpma_graph_protocol_device(cuda):
    check_graph_protocol(cuda)


@pytest.mark.parametrize("n,e", [(1, 1), (64, 400), (2708, 10556), (1 << 20, 1 << 24)])
def test_key_order_emission_equals_static_csr_device(cuda, n, e):
    """Up to BASELINE's |V| = 1M, |E| = 16M: the emitted GPMA view is the static builder's CSR, bit for bit."""
    check_emit_equals_static_builder(cuda, n, e)


@pytest.mark.parametrize("F", [7, 16, 64, 128])
@pytest.mark.parametrize("use_ew", [False, True])
def test_gcnconv_on_gpma_graph_matches_the_gapped_array_kernel(cuda, F, use_ew):
    """GCNConv on a GPMAGraph == the reference's 'gpma' kernel loop run by the oracle over a GAPPED image of
    the same snapshot (holes, row walls, tombstones): forward bit-identical (same in-row order), backward
    within 1e-5 (the reference leaves the reverse rows' order to atomics)."""
    from stgraph_amd.graph import GPMAGraph
    from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
    n = 300
    snaps = snapshots(31, n, 4000, 500, 3)
    G = GPMAGraph(snaps, n, device=cuda)
    G.get_graph(2)                                              # two update steps away from the base graph
    edges = snaps[2]
    rng = np.random.default_rng(F)
    present = set(edges)
    dead = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, 300), rng.integers(0, n, 300))
            if (int(b), int(a)) not in present]
    o = OracleGPMA(n, [(d, s) for s, d in edges], dead=dead, hole_pct=30, seed=F)
    o.label_edges()
    o.build_backward_csr()
    norm = gcn_norm(G.in_degrees())
    assert np.array_equal(G.in_degrees(), o.out_degree.astype(np.int32))
    G.set_ndata("norm", _t(norm, cuda))
    w = (rng.random(len(edges)) + 0.5).astype(np.float32)      # indexed by label - 1
    x_np = rng.standard_normal((n, F)).astype(np.float32)
    R_np = rng.standard_normal((n, F)).astype(np.float32)
    conv = GCNConv(F, F, bias=False).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(F))
    x = _t(x_np, cuda).requires_grad_(True)
    out = conv(G, x, edge_weight=_t(w.reshape(-1, 1), cuda) if use_ew else None)
    (out * _t(R_np, cuda)).sum().backward()
    ew = w if use_ew else None
    assert np.array_equal(out.detach().cpu().numpy(), o.gcn_agg(x_np, norm, norm, ew))
    np.testing.assert_allclose(x.grad.cpu().numpy(), o.gcn_agg(R_np, norm, norm, ew, backward=True),
                               rtol=1e-5, atol=1e-5)


def test_gpma_graph_tgcn_bptt_matches_reference_run_of_the_same_snapshots(cuda):
    """tests/golden/naive_tgcn.npz was recorded from the reference's own NaiveGraph + TGCN run; a GPMA graph
    of the same snapshots represents the same per-timestamp graphs, so hidden states, loss and every gradient
    must agree (the reference's GPMA kernels would differ only through defect D17)."""
    from stgraph_amd.graph import GPMAGraph
    d = golden("naive_tgcn.npz")
    n, T = int(d["num_nodes"]), int(d["T"])
    G = GPMAGraph([_edges(d, f"t{t}_") for t in range(T)], n, device=cuda)
    feats, targets = _t(d["feats"], cuda), _t(d["targets"], cuda)
    model = TGCNModel(feats.shape[2], 16, 1).to(cuda)
    _load_params(model, d, "param_", cuda)
    for epoch in range(2):
        model.zero_grad()
        G.reset_graph()
        hidden, cost, hs = None, 0, []
        for t in range(T):
            G.get_graph(t)
            for side in ("fwd",):
                c = G.csr(side)
                for k in ("row_offset", "column_indices", "eids"):
                    assert np.array_equal(getattr(c, k).cpu().numpy(), d[f"t{t}_{side}_{k}"]), (t, side, k)
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
    G.check()


def test_gpma_graph_equals_naive_graph_bit_for_bit_on_tgcn(cuda):
    """Same snapshots through GPMAGraph and NaiveGraph: identical CSRs -> identical kernels -> identical bits."""
    from stgraph_amd.graph import GPMAGraph, NaiveGraph
    n, T = 2000, 6
    snaps = snapshots(77, n, 30000, 2000, T)
    res = []
    for cls in (GPMAGraph, NaiveGraph):
        G = cls([list(s) for s in snaps], n, device=cuda)
        torch.manual_seed(1)
        model = TGCNModel(8, 16, 1).to(cuda)
        feats = torch.randn(T, n, 8, device=cuda)
        G.reset_graph()
        hidden, cost = None, 0
        for t in range(T):
            G.get_graph(t)
            deg = torch.from_numpy(G.in_degrees()).float()
            G.set_ndata("norm", torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg)).unsqueeze(1).to(cuda))
            y, hidden = model(G, feats[t], None, hidden)
            cost = cost + (y ** 2).mean()
        cost.backward()
        res.append([cost.detach()] + [p.grad.clone() for p in model.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_update_is_a_pure_function_of_the_edge_set_at_scale(cuda):
    """|V| = 1M, |E| = 16M, +-5 %: forward then reverted updates restore the key arrays; the view after the
    update equals the static builder's CSR of the new edge list."""
    from stgraph_amd import kernels
    from stgraph_amd.graph.dynamic.gpma import gpma as M
    n, e = 1 << 20, 1 << 24
    src, dst = random_graph(12, n, e, hub=False)
    k = e // 20
    s, d = torch.from_numpy(src).to(cuda), torch.from_numpy(dst).to(cuda)
    g = M.GPMA(device=cuda)
    M.init_gpma(g, n)
    M.init_graph_updates(g, {"0": {"add": (s[k:], d[k:]), "delete": (s[:0], d[:0])},
                             "1": {"add": (s[:k], d[:k]), "delete": (s[-k:], d[-k:])}}, reverse_edges=True)
    M.edge_update_t(g, 0)
    base = g.edge_set
    M.edge_update_t(g, 1)
    M.label_edges(g)
    g.check()
    ref = kernels.build_graph_csr(s[:-k], d[:-k], n, cuda)
    M.build_backward_csr(g)
    for rev, side in ((False, ref.fwd), (True, ref.bwd)):
        c = g.csr(rev)
        assert torch.equal(c.row_offset, side.row_offset) and torch.equal(c.column_indices, side.column_indices)
        assert torch.equal(c.eids, side.eids)
    assert M.get_graph_attr(g) == (n, e - k)
    M.edge_update_t(g, 1, revert_update=True)
    assert torch.equal(g.edge_set.keys_fwd, base.keys_fwd) and torch.equal(g.edge_set.keys_bwd, base.keys_bwd)
    g.check()
