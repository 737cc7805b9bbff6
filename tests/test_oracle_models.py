"""Pin the oracle-backed CPU layers (tests/oracle_layers.py) against fixtures produced by the
REFERENCE layers: TGCN BPTT on a StaticGraph, incl. every parameter gradient."""
import numpy as np
import pytest
import torch

from tests.oracle_layers import OracleGraphView, make_oracle_tgcn
from tests.util import golden


class Model(torch.nn.Module):
    def __init__(self, fin, hid, out):
        super().__init__()
        self.temporal = make_oracle_tgcn()(fin, hid)
        self.linear = torch.nn.Linear(hid, out)

    def forward(self, g, x, ew, hidden):
        h = self.temporal(g, x, ew, hidden)
        return self.linear(torch.relu(h)), h


@pytest.mark.parametrize("B", [3, 6])
def test_oracle_tgcn_matches_reference_bptt(B):
    torch.set_num_threads(1)
    d = golden("tgcn.npz")
    n, T = int(d["num_nodes"]), d["feats"].shape[0]
    g = OracleGraphView(d["src"], d["dst"], n)
    g.set_ndata("norm", torch.from_numpy(d["norm"]))
    w = torch.from_numpy(d["edge_weight_by_eid"])
    feats, targets = torch.from_numpy(d["feats"]), torch.from_numpy(d["targets"])
    model = Model(feats.shape[2], 16, 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"B{B}_param_{k}"]))
    hs = []
    for i, w0 in enumerate(range(0, T, B)):
        model.zero_grad()
        hidden, cost = None, 0
        for t in range(w0, w0 + B):
            y, hidden = model(g, feats[t], w, hidden)
            cost = cost + torch.mean((y - targets[t]) ** 2)
            hs.append(hidden.detach())
        cost = cost / (B + 1)
        cost.backward()
        np.testing.assert_allclose(cost.item(), d[f"B{B}_cost"][i], rtol=1e-5, atol=1e-6)
        for k, p in model.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), d[f"B{B}_w{w0}_grad_{k}"], rtol=1e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(torch.stack(hs).numpy(), d[f"B{B}_hidden"], rtol=1e-5, atol=1e-6)


class StaticTemporalModel(torch.nn.Module):
    """benchmarking/static-temporal-tgcn/seastar/model.py:6-18 on the oracle-backed TGCN."""

    def __init__(self, feat, hid, out):
        super().__init__()
        self.temporal = make_oracle_tgcn()(feat, hid)
        self.linear = torch.nn.Linear(hid, feat)
        self.linear2 = torch.nn.Linear(feat, out)

    def forward(self, g, x, ew, hidden):
        h = self.temporal(g, x, ew, hidden)
        y = self.linear(torch.relu(h))
        return self.linear2(y), y, h


@pytest.mark.parametrize("use_ew", [False, True])
def test_oracle_static_temporal_loop_at_native_widths(use_ew):
    """The oracle's aggregation inside the reference's window loop at the benchmark's widths (32 -> 64, N = 4096) against
    tgcn_native.npz, recorded from the reference stack (tests/golden/make_golden_models.py)."""
    torch.set_num_threads(4)
    d = golden("tgcn_native.npz")
    n, T, feat, hid, B = int(d["num_nodes"]), int(d["T"]), int(d["feat"]), int(d["hidden"]), 3
    g = OracleGraphView(d["src"], d["dst"], n)
    g.set_ndata("norm", torch.from_numpy(d["norm"]))
    w = torch.from_numpy(d["edge_weight_by_eid"]) if use_ew else None
    targets = torch.from_numpy(np.random.default_rng(int(d["targets_seed"])).standard_normal((T, n, 1), dtype=np.float32))
    tag = f"{'ew' if use_ew else 'now'}_B{B}"
    model = StaticTemporalModel(feat, hid, 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"{tag}_param_{k}"]))
    rows, hs = d["rows"], []
    for index in range(T // B):
        model.zero_grad()
        cost, hidden = 0, None
        y_hat = torch.from_numpy(np.random.default_rng(int(d[f"{tag}_x0_seeds"][index])).standard_normal((n, feat), dtype=np.float32))
        for k in range(B):
            y_out, y_hat, hidden = model(g, y_hat, w, hidden)
            cost = cost + torch.mean((y_out - targets[index * B + k]) ** 2)
            hs.append(hidden.detach())
        cost = cost / (B + 1)
        cost.backward()
        np.testing.assert_allclose(cost.item(), d[f"{tag}_cost"][index], rtol=1e-5, atol=1e-6)
        for k, p in model.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), d[f"{tag}_w{index}_grad_{k}"], rtol=1e-4, atol=1e-6, err_msg=k)
    H = torch.stack(hs)
    np.testing.assert_allclose(H[:, rows].numpy(), d[f"{tag}_hidden_rows"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H.double().sum(1).numpy(), d[f"{tag}_hidden_colsum"], rtol=0,
                               atol=float(1e-6 * d[f"{tag}_hidden_abs_colsum"].max()))


def test_oracle_gcn_model_step():
    """2-layer GCN 128 -> 128 -> 128 on the Cora-shaped graph (reference order, the oracle's aggregation) against
    gcn_model.npz: logits, loss and every gradient of the first step."""
    from tests.oracle_layers import OracleGCNConv
    torch.set_num_threads(4)
    d = golden("gcn_model.npz")
    tag, n, ntrain = "w128_128_128", int(d["num_nodes"]), int(d["ntrain"])
    g = OracleGraphView(d["src"], d["dst"], n)
    g.set_ndata("norm", torch.from_numpy(d["norm"]))
    rng = np.random.default_rng(int(d[tag + "_seed"]))
    x = torch.from_numpy(rng.standard_normal((n, 128), dtype=np.float32))
    labels = torch.from_numpy(rng.integers(0, 128, n).astype(np.int64))
    layers = torch.nn.ModuleList([OracleGCNConv(128, 128, torch.relu), OracleGCNConv(128, 128, None)])
    model = torch.nn.Module()
    model.layers = layers
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"{tag}_param0_{k}"]))
    logits = layers[1](g, layers[0](g, x))
    loss = torch.nn.functional.cross_entropy(logits[:ntrain], labels[:ntrain])
    loss.backward()
    np.testing.assert_allclose(logits.detach()[d["rows"]].numpy(), d[tag + "_logits_rows"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss.item(), d[tag + "_losses"][0], rtol=1e-6)
    for k, p in model.named_parameters():
        w = d[f"{tag}_grad0_{k}"]
        assert np.abs(p.grad.numpy() - w).max() <= 1e-5 * np.abs(w).max(), k


class DynamicTemporalModel(torch.nn.Module):
    """benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21 on the oracle-backed TGCN."""

    def __init__(self, feat, hid):
        super().__init__()
        self.temporal = make_oracle_tgcn()(feat, hid)
        self.linear = torch.nn.Linear(hid, feat)

    def forward(self, g, x, ew, hidden):
        h = self.temporal(g, x, ew, hidden)
        return self.linear(torch.relu(h)), h


def _dyn_window(d, model, graphs, B, index, n, feat, M):
    edges = [torch.from_numpy(d[f"t{t}_label_edges"]) for t in range(len(graphs) - 1)]
    tg = torch.cat([torch.ones(M), torch.zeros(M)])
    crit = torch.nn.BCEWithLogitsLoss()
    x = torch.from_numpy(np.random.default_rng(int(d[f"B{B}_x0_seeds"][index])).standard_normal((n, feat), dtype=np.float32))
    cost, h = 0, None
    for t in range(index * B, min((index + 1) * B, len(graphs) - 1)):
        x, h = model(graphs[t], x, None, h)
        cost = cost + crit((x[edges[t][0]] * x[edges[t][1]]).sum(-1), tg)
    return cost / (B + 1)


def _dyn_graphs(d, n, T):
    from tests.oracle_layers import gcn_norm_tensor
    graphs = []
    for t in range(T):
        g = OracleGraphView(d[f"t{t}_src"], d[f"t{t}_dst"], n, "csr")
        g.set_ndata("norm", gcn_norm_tensor(g.in_degrees()))
        graphs.append(g)
    return graphs


@pytest.mark.parametrize("B", [3, 6])
def test_oracle_dynamic_temporal_loop_at_native_widths(B):
    torch.set_num_threads(4)
    d = golden("dyn_tgcn.npz")
    n, T, feat, hid, M = (int(d[k]) for k in ("num_nodes", "T", "feat", "hidden", "M"))
    graphs = _dyn_graphs(d, n, T)
    model = DynamicTemporalModel(feat, hid)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"B{B}_param_{k}"]))
    for index in range(len(d[f"B{B}_cost"])):
        model.zero_grad()
        cost = _dyn_window(d, model, graphs, B, index, n, feat, M)
        cost.backward()
        np.testing.assert_allclose(cost.item(), d[f"B{B}_cost"][index], rtol=1e-6)
        for k, p in model.named_parameters():
            w = d[f"B{B}_w{index}_grad_{k}"]
            assert np.abs(p.grad.numpy() - w).max() <= 1e-5 * np.abs(w).max(), k


def test_reference_second_epoch_on_a_naive_graph_runs_snapshot_0_on_a_stale_forward_csr():
    """Reference defect D12, pinned so that the deviation is a known quantity: after reset_graph() the
    NaiveGraph keeps the forward CSR pointers of the previous epoch's last forward snapshot (dynamic_graph.py:81-107,
    naive_graph.py:103-139), so from the second epoch on snapshot 0's FORWARD aggregation runs over the last snapshot the
    previous epoch's get_graph calls reached (here T - 1) while its
    backward runs over snapshot 0.  The recorded second-epoch numbers are reproduced exactly by that substitution; the
    product implements the intended semantics (every epoch == the reference's first epoch)."""
    import copy
    torch.set_num_threads(4)
    d = golden("dyn_tgcn.npz")
    n, T, feat, hid, M = (int(d[k]) for k in ("num_nodes", "T", "feat", "hidden", "M"))
    graphs = _dyn_graphs(d, n, T)
    stale = copy.copy(graphs[0])
    stale.g = copy.copy(graphs[0].g)
    stale.g.fwd = graphs[int(d["stale_forward_snapshot"])].g.fwd          # forward: stale pointers; backward + norm: snapshot 0
    model = DynamicTemporalModel(feat, hid)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"B3_param_{k}"]))
    cost = _dyn_window(d, model, [stale] + graphs[1:], 3, 0, n, feat, M)
    cost.backward()
    np.testing.assert_allclose(cost.item(), d["stale_B3_cost"][0], rtol=1e-6)
    assert abs(cost.item() - d["B3_cost"][0]) > 1e-5                       # and it is NOT the first epoch's value
    for k, p in model.named_parameters():
        w = d[f"stale_B3_w0_grad_{k}"]
        assert np.abs(p.grad.numpy() - w).max() <= 1e-5 * np.abs(w).max(), k
