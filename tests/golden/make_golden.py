"""Generate the golden vectors under tests/golden/ from the reference itself.

BUILD CONTAINER ONLY (reads /root/reference through tests/golden/ref_harness.py;
see that module's docstring for exactly what runs unmodified and what is
substituted).  Usage:   python tests/golden/make_golden.py

Fixture matrix follows SURVEY.md 8(c):
  csr_*.npz      StaticGraph(edge_list, w, N): fwd/bwd row_offset, column_indices,
                 eids, degrees (bit-exact targets), for N in {1,5,64,2708}
  gcn.npz        GCNConv (identity weight => the bare aggregation kernel) fwd + grad,
                 F in {1,4,7,16,32,64,100,128,256,300} x {no-ew, ew} x {StaticGraph ('csr_unsorted'),
                 NaiveGraph ('csr', node_ids indirection)}
  gat.npz        GATConv fwd + grads (layer level and kernel level: A, S, grad_el, grad_er, grad_feat)
                 (H,D) in {(1,7),(2,4),(8,8),(8,64)}, one vertex with in-degree 0
  tgcn.npz       TGCN, N=50 E=300 F_in=8 hidden=16 T=6, B in {3,6}: per-step hidden states, loss,
                 every parameter gradient; plus a 2-epoch Adam loop (per-window loss, param checksums)
  naive_tgcn.npz NaiveGraph with 4 snapshots (+-10 % churn): per-t CSRs and a TGCN BPTT pass
  gcn_n200.npz   the GCN matrix again on N = 200, E = 1500 with an isolated vertex, self-loops and a hub of in-degree 90
                 (rows longer than a wave, rows spanning several index rounds), all widths, StaticGraph
  gcn_cora.npz   the Cora-SHAPED graph of the benchmark (N = 2708, E = 10556, mirrored pairs, max degree 168), widths
                 {7, 16, 64, 300} x {no-ew, ew}: inputs are re-drawn from the stored seeds (torch CPU generator); of the
                 outputs the file keeps 296 rows in full (the 40 highest-degree rows + 256 sampled) and the fp64 column
                 sums over ALL rows, which pin every row of a bit-exact result
  gat_cora.npz   GATConv on the same graph, (H,D) in {(8,8),(8,64)}: A, S, grad_el, grad_er in full, out / grad_feat on
                 the sampled rows + fp64 column sums
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.load_reference()
from stgraph.compiler.executor import Executor  # noqa: E402
from stgraph.graph import NaiveGraph, StaticGraph  # noqa: E402
from stgraph.nn.pytorch.static.gat_conv import GATConv  # noqa: E402
from stgraph.nn.pytorch.static.gcn_conv import GCNConv  # noqa: E402
from stgraph.nn.pytorch.temporal.tgcn import TGCN  # noqa: E402
from stgraph.graph.static.csr import get_array  # noqa: E402

torch.set_num_threads(1)


# ----------------------------------------------------------------------------- helpers
def random_edges(rng, n, e, self_loops=True, hub=None, isolated=()):
    """Duplicate-free directed edge list in random order."""
    pairs = set()
    iso = set(isolated)
    if hub is not None:
        for u in range(n):
            if u != hub and u not in iso and len(pairs) < e // 3:
                pairs.add((u, hub))
    tries = 0
    while len(pairs) < e:
        s, d = int(rng.integers(n)), int(rng.integers(n))
        tries += 1
        if tries > 100 * e + 1000:
            raise RuntimeError("cannot place edges")
        if s in iso or d in iso:
            continue
        if not self_loops and s == d:
            continue
        pairs.add((s, d))
    lst = list(pairs)
    rng.shuffle(lst)
    return [(int(a), int(b)) for a, b in lst]


def csr_arrays(csr, n, e):
    return dict(
        row_offset=np.array(get_array(csr.row_offset_ptr, n + 1), np.int32),
        column_indices=np.array(get_array(csr.column_indices_ptr, e), np.int32),
        eids=np.array(get_array(csr.eids_ptr, e), np.int32),
        node_ids=np.array(get_array(csr.node_ids_ptr, n), np.int32),
    )


def norm_of(g):
    deg = torch.from_numpy(np.asarray(g.in_degrees())).float()
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0          # benchmarking/gcn/seastar/train.py:53-57
    return norm.unsqueeze(1)


class Capture:
    """Record kernel-level tensors crossing Executor.forward_cb / backward_cb."""

    def __init__(self):
        self.fwd, self.bwd = [], []
        self._f, self._b = Executor.forward_cb, Executor.backward_cb

    def __enter__(self):
        cap = self

        def fcb(ex, uid, kernel_args, rets, tensor_list):
            units = ex.forward_exec_units[uid]
            args = {v.id: t.detach().clone() for v, t in zip(units.joint_args(), tensor_list)}
            saved_ids = list(ex.ts.bwd_common_tensor_list)
            cur = ex.ts.current_tensor_map
            out = cap._f(ex, uid, kernel_args, rets, tensor_list)
            saved = {k: ex.ts.tensor_map_stack.top()[k].detach().clone() for k in saved_ids
                     if k in ex.ts.tensor_map_stack.top()}
            cap.fwd.append(dict(args=args, rets={r.id: t.detach().clone() for r, t in zip(rets, out)},
                                saved=saved))
            del cur
            return out

        def bcb(ex, kid, grad_list):
            funits = ex.forward_exec_units[kid]
            out = cap._b(ex, kid, grad_list)
            names = [a.id for a in funits.joint_args()]
            cap.bwd.append(dict(grad_out=[g.detach().clone() for g in grad_list],
                                grads={n: (t.detach().clone() if t is not None else None)
                                       for n, t in zip(names, out)}))
            return out

        Executor.forward_cb, Executor.backward_cb = fcb, bcb
        return self

    def __exit__(self, *a):
        Executor.forward_cb, Executor.backward_cb = self._f, self._b


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in d.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


# ----------------------------------------------------------------------------- CSR
def gen_csr():
    cases = [
        ("n1", 1, 1, dict()),                                   # single self-loop
        ("n5", 5, 12, dict(isolated=(3,))),
        ("n64", 64, 400, dict(hub=7, isolated=(0, 63, 31))),
        ("n2708", 2708, 10556, dict(hub=100, isolated=(5, 2707))),
    ]
    for tag, n, e, kw in cases:
        rng = np.random.default_rng(100 + n)
        el = random_edges(rng, n, e, **kw)
        given = np.array(el, np.int32)
        w = rng.uniform(0.5, 1.5, e).astype(np.float32)
        g = StaticGraph(el, w.tolist(), n)
        d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1], weights_by_eid=w,
                 sorted_inplace=np.array(el, np.int32),          # static_graph.py:66-67 sorts the caller's list
                 in_degrees=g.in_degrees(), out_degrees=g.out_degrees(),
                 weighted_in_degrees=g.weighted_in_degrees(),
                 num_edges=g.get_num_edges())
        for side, csr in (("fwd", g._forward_graph), ("bwd", g._backward_graph)):
            for k, v in csr_arrays(csr, n, e).items():
                d[f"{side}_{k}"] = v
        save(f"csr_{tag}.npz", d)
    # the 5-edge example recorded in SURVEY.md 8(c) (observed on the reference during the survey)
    el = [(0, 1), (1, 0), (2, 1), (0, 2), (3, 2)]
    g = StaticGraph(list(el), [1.0] * 5, 4)
    f, b = csr_arrays(g._forward_graph, 4, 5), csr_arrays(g._backward_graph, 4, 5)
    assert f["row_offset"].tolist() == [0, 1, 3, 5, 5] and f["column_indices"].tolist() == [1, 0, 2, 0, 3]
    assert b["row_offset"].tolist() == [0, 2, 3, 4, 5] and b["column_indices"].tolist() == [1, 2, 0, 1, 2]
    assert b["eids"].tolist() == [1, 3, 0, 2, 4]


# ----------------------------------------------------------------------------- GCN
def gen_gcn():
    n, e = 24, 120
    rng = np.random.default_rng(11)
    el = random_edges(rng, n, e, hub=3, isolated=(9,))
    given = np.array(el, np.int32)
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1])
    static = StaticGraph(list(el), [1.0] * e, n)
    naive = NaiveGraph([list(el)], n)
    w_eid = torch.from_numpy(rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32))
    d["edge_weight_by_eid"] = w_eid
    for gname, g in (("static", static), ("naive", naive)):
        norm = norm_of(g)
        g.set_ndata("norm", norm)
        d[f"{gname}_norm"] = norm
        fw = g._forward_graph if gname == "static" else g._forward_graph[0]
        bw = g._backward_graph if gname == "static" else g._backward_graph[0]
        for side, csr in (("fwd", fw), ("bwd", bw)):
            for k, v in csr_arrays(csr, n, e).items():
                d[f"{gname}_{side}_{k}"] = v
        for F in (1, 4, 7, 16, 32, 64, 100, 128, 256, 300):
            for use_ew in (False, True):
                torch.manual_seed(1000 + F)
                conv = GCNConv(F, F, bias=False)
                with torch.no_grad():
                    conv.weight.copy_(torch.eye(F))      # h = x @ I  (exact) => bare aggregation kernel
                x = torch.randn(n, F, requires_grad=True)
                R = torch.randn(n, F)
                if gname == "naive":
                    g.get_graph(0)
                out = conv(g, x, edge_weight=w_eid if use_ew else None)
                (out * R).sum().backward()
                tag = f"{gname}_F{F}_{'ew' if use_ew else 'now'}"
                d[tag + "_x"], d[tag + "_R"] = x.detach(), R
                d[tag + "_out"], d[tag + "_grad_x"] = out.detach(), x.grad.detach()
    save("gcn.npz", d)


# ----------------------------------------------------------------------------- GAT
def gen_gat():
    n, e = 24, 110
    rng = np.random.default_rng(12)
    el = random_edges(rng, n, e, hub=2, isolated=(5,))
    # vertex 7: out-edges only (in-degree 0 but present in the graph)
    el = [(s, t) for (s, t) in el if t != 7]
    if not any(s == 7 for s, _ in el):
        el.append((7, 1))
    e = len(el)
    given = np.array(el, np.int32)
    g = StaticGraph(list(el), [1.0] * e, n)
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1])
    for side, csr in (("fwd", g._forward_graph), ("bwd", g._backward_graph)):
        for k, v in csr_arrays(csr, n, e).items():
            d[f"{side}_{k}"] = v
    fin = 6
    for (H, D) in ((1, 7), (2, 4), (8, 8), (8, 64)):
        torch.manual_seed(2000 + H * 100 + D)
        conv = GATConv(fin, D, H)
        x = torch.randn(n, fin, requires_grad=True)
        R = torch.randn(n, H, D)
        with Capture() as cap:
            out = conv(g, x)
            (out * R).sum().backward()
        tag = f"H{H}_D{D}"
        d[tag + "_x"], d[tag + "_R"] = x.detach(), R
        d[tag + "_fc_weight"] = conv.fc.weight.detach()
        d[tag + "_attn_l"], d[tag + "_attn_r"] = conv.attn_l.detach(), conv.attn_r.detach()
        d[tag + "_out"] = out.detach()
        d[tag + "_grad_x"] = x.grad.detach()
        d[tag + "_grad_fc_weight"] = conv.fc.weight.grad.detach()
        d[tag + "_grad_attn_l"] = conv.attn_l.grad.detach()
        d[tag + "_grad_attn_r"] = conv.attn_r.grad.detach()
        f0, b0 = cap.fwd[0], cap.bwd[0]
        d[tag + "_k_el"] = f0["args"]["Velinb"]
        d[tag + "_k_er"] = f0["args"]["Vercen"]
        d[tag + "_k_feat"] = f0["args"]["Vfeat_srcinb"]
        # saved-for-backward tensors: the two [*,H,1] intermediates are A (edge) and S (dest)
        for k, t in f0["saved"].items():
            if t.dim() == 3 and t.shape[-1] == 1 and t.shape[0] == e and not k.endswith("inb") and not k.endswith("cen"):
                d[tag + "_k_A"] = t
            if t.dim() == 3 and t.shape[-1] == 1 and t.shape[0] == n and not k.endswith("inb") and not k.endswith("cen"):
                d[tag + "_k_S"] = t
        d[tag + "_k_grad_el"] = b0["grads"]["Velinb"]
        d[tag + "_k_grad_er"] = b0["grads"]["Vercen"]
        d[tag + "_k_grad_feat"] = b0["grads"]["Vfeat_srcinb"]
        assert tag + "_k_A" in d and tag + "_k_S" in d, list(f0["saved"].keys())
    save("gat.npz", d)


# ------------------------------------------------------------- larger graphs (round 2)
def cora_shaped(seed=0, n=2708, pairs=5278, max_deg=168):
    """The benchmark's Cora-shaped generator (bench.py::cora_shaped, restated so that this script stands alone)."""
    rng = np.random.default_rng(seed)
    w = (np.arange(1, n + 1, dtype=np.float64)) ** -0.6
    w = np.minimum(w / w.sum() * 2 * pairs, max_deg)
    p = w / w.sum()
    got = set()
    while len(got) < pairs:
        a = rng.choice(n, size=2 * pairs, p=p)
        b = rng.choice(n, size=2 * pairs, p=p)
        for u, v in zip(a, b):
            if u != v and (min(u, v), max(u, v)) not in got and len(got) < pairs:
                got.add((min(u, v), max(u, v)))
    und = np.array(sorted(got), np.int32)
    return np.concatenate([und[:, 0], und[:, 1]]), np.concatenate([und[:, 1], und[:, 0]])


def sample_rows(deg, count=256, top=40, seed=0):
    rng = np.random.default_rng(seed)
    hubs = np.argsort(-deg, kind="stable")[:top]
    rest = rng.choice(len(deg), size=min(count, len(deg)), replace=False)
    return np.unique(np.concatenate([hubs, rest])).astype(np.int64)


def gen_gcn_n200():
    n, e = 200, 1500
    rng = np.random.default_rng(21)
    el = random_edges(rng, n, e, self_loops=True, hub=17, isolated=(3, 199))
    for v in (0, 5, 50):
        if (v, v) not in el:
            el.append((v, v))                                 # explicit self-loops
    hubs = [(u, 17) for u in range(20, 111) if (u, 17) not in el]
    el += hubs[: max(0, 90 - sum(1 for _, t in el if t == 17))]
    e = len(el)
    given = np.array(el, np.int32)
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1])
    g = StaticGraph(list(el), [1.0] * e, n)
    w_eid = torch.from_numpy(rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32))
    d["edge_weight_by_eid"] = w_eid
    norm = norm_of(g)
    g.set_ndata("norm", norm)
    d["norm"] = norm
    d["in_degrees"] = g.in_degrees()
    for side, csr in (("fwd", g._forward_graph), ("bwd", g._backward_graph)):
        for k, v in csr_arrays(csr, n, e).items():
            d[f"{side}_{k}"] = v
    for F in (1, 4, 7, 16, 32, 64, 100, 128, 256, 300):
        for use_ew in (False, True):
            torch.manual_seed(3000 + F)
            conv = GCNConv(F, F, bias=False)
            with torch.no_grad():
                conv.weight.copy_(torch.eye(F))
            x = torch.randn(n, F, requires_grad=True)
            R = torch.randn(n, F)
            out = conv(g, x, edge_weight=w_eid if use_ew else None)
            (out * R).sum().backward()
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            d[tag + "_x"], d[tag + "_R"] = x.detach(), R
            d[tag + "_out"], d[tag + "_grad_x"] = out.detach(), x.grad.detach()
    save("gcn_n200.npz", d)


def gen_gcn_cora():
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    el = [(int(a), int(b)) for a, b in zip(src, dst)]
    g = StaticGraph(list(el), [1.0] * e, n)
    rng = np.random.default_rng(22)
    w_eid = torch.from_numpy(rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32))
    norm = norm_of(g)
    g.set_ndata("norm", norm)
    deg = np.asarray(g.in_degrees())
    rows = sample_rows(deg)
    d = dict(num_nodes=n, src=src, dst=dst, edge_weight_by_eid=w_eid, norm=norm, in_degrees=deg, rows=rows,
             max_in_degree=int(deg.max()))
    for side, csr in (("fwd", g._forward_graph), ("bwd", g._backward_graph)):
        for k, v in csr_arrays(csr, n, e).items():
            d[f"{side}_{k}"] = v
    for F in (7, 16, 64, 300):
        for use_ew in (False, True):
            seed = 4000 + F
            conv = GCNConv(F, F, bias=False)
            with torch.no_grad():
                conv.weight.copy_(torch.eye(F))
            xr = np.random.default_rng(seed)                   # numpy's PCG64 + ziggurat: the same bits on every host, so
            x = torch.from_numpy(xr.standard_normal((n, F), dtype=np.float32)).requires_grad_(True)   # the tests re-draw
            R = torch.from_numpy(xr.standard_normal((n, F), dtype=np.float32))                        # x, R from the seed
            out = conv(g, x, edge_weight=w_eid if use_ew else None)
            (out * R).sum().backward()
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            d[tag + "_seed"] = seed
            d[tag + "_x_rows"], d[tag + "_R_rows"] = x.detach()[rows], R[rows]   # spot check of the re-drawn inputs
            d[tag + "_out_rows"], d[tag + "_grad_x_rows"] = out.detach()[rows], x.grad.detach()[rows]
            d[tag + "_out_colsum"] = out.detach().double().sum(0)
            d[tag + "_grad_x_colsum"] = x.grad.detach().double().sum(0)
            d[tag + "_out_abs_colsum"] = out.detach().double().abs().sum(0)
            d[tag + "_grad_x_abs_colsum"] = x.grad.detach().double().abs().sum(0)
    save("gcn_cora.npz", d)


def gen_gat_cora():
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    el = [(int(a), int(b)) for a, b in zip(src, dst)]
    g = StaticGraph(list(el), [1.0] * e, n)
    deg = np.asarray(g.in_degrees())
    rows = sample_rows(deg)
    d = dict(num_nodes=n, src=src, dst=dst, rows=rows, in_degrees=deg)
    fin = 6
    for (H, D) in ((8, 8), (8, 64)):
        seed = 5000 + H * 100 + D
        torch.manual_seed(seed)
        conv = GATConv(fin, D, H)
        xr = np.random.default_rng(seed)
        # x in multiples of 1/8, fc weight in multiples of 1/32: every partial sum of feat = fc(x) is exact in fp32, so
        # the tests recompute feat with any BLAS and get the reference's bits; R re-drawn from the seed (numpy)
        x = torch.from_numpy((xr.integers(-8, 9, (n, fin)) / 8.0).astype(np.float32)).requires_grad_(True)
        with torch.no_grad():
            conv.fc.weight.copy_(torch.from_numpy((xr.integers(-16, 17, (H * D, fin)) / 32.0).astype(np.float32)))
        R = torch.from_numpy(xr.standard_normal((n, H, D), dtype=np.float32))
        with Capture() as cap:
            out = conv(g, x)
            (out * R).sum().backward()
        tag = f"H{H}_D{D}"
        f0, b0 = cap.fwd[0], cap.bwd[0]
        d[tag + "_seed"] = seed
        d[tag + "_k_el"], d[tag + "_k_er"] = f0["args"]["Velinb"], f0["args"]["Vercen"]
        feat = f0["args"]["Vfeat_srcinb"]
        d[tag + "_x"] = x.detach()
        d[tag + "_fc_weight"] = conv.fc.weight.detach()
        d[tag + "_R_rows"] = R[rows]
        d[tag + "_k_feat_rows"] = feat[rows]
        for k, t in f0["saved"].items():
            if t.dim() == 3 and t.shape[-1] == 1 and t.shape[0] == e and not k.endswith("inb") and not k.endswith("cen"):
                d[tag + "_k_A"] = t
            if t.dim() == 3 and t.shape[-1] == 1 and t.shape[0] == n and not k.endswith("inb") and not k.endswith("cen"):
                d[tag + "_k_S"] = t
        d[tag + "_out_rows"] = out.detach()[rows]
        d[tag + "_out_colsum"] = out.detach().double().sum(0)
        d[tag + "_k_grad_el"], d[tag + "_k_grad_er"] = b0["grads"]["Velinb"], b0["grads"]["Vercen"]
        gf = b0["grads"]["Vfeat_srcinb"]
        d[tag + "_k_grad_feat_rows"] = gf[rows]
        d[tag + "_k_grad_feat_colsum"] = gf.double().sum(0)
        d[tag + "_grad_attn_l"], d[tag + "_grad_attn_r"] = conv.attn_l.grad.detach(), conv.attn_r.grad.detach()
        d[tag + "_attn_l"], d[tag + "_attn_r"] = conv.attn_l.detach(), conv.attn_r.detach()
    save("gat_cora.npz", d)


# ----------------------------------------------------------------------------- TGCN
class RefTGCNModel(torch.nn.Module):
    """Model of tests/scripts/v1_1_0/temporal_tgcn_dataloaders (TGCN + ReLU + Linear head)."""

    def __init__(self, node_features, num_hidden_units, out_features):
        super().__init__()
        self.temporal = TGCN(node_features, num_hidden_units)
        self.linear = torch.nn.Linear(num_hidden_units, out_features)

    def forward(self, g, node_feat, edge_weight, hidden_state):
        h = self.temporal(g, node_feat, edge_weight, hidden_state)
        y = torch.relu(h)
        y = self.linear(y)
        return y, h


def params_dict(model, prefix, grads=False):
    out = {}
    for k, p in model.named_parameters():
        out[f"{prefix}{k}"] = (p.grad if grads else p).detach().clone()
    return out


def gen_tgcn():
    n, e, fin, hid, T = 50, 300, 8, 16, 6
    rng = np.random.default_rng(13)
    el = random_edges(rng, n, e, hub=4, isolated=(17,))
    given = np.array(el, np.int32)
    w_eid = torch.from_numpy(rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32))
    g = StaticGraph(list(el), w_eid.reshape(-1).tolist(), n)
    g.set_ndata("norm", norm_of(g))
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1], edge_weight_by_eid=w_eid, norm=g.get_ndata("norm"))
    feats = torch.from_numpy(rng.standard_normal((T, n, fin)).astype(np.float32))
    targets = torch.from_numpy(rng.standard_normal((T, n, 1)).astype(np.float32))
    d["feats"], d["targets"] = feats, targets
    for B in (3, 6):
        torch.manual_seed(3000 + B)
        model = RefTGCNModel(fin, hid, 1)
        d.update(params_dict(model, f"B{B}_param_"))
        hs, cost_hist = [], []
        for w0 in range(0, T, B):
            model.zero_grad()
            hidden, cost = None, 0
            for t in range(w0, w0 + B):
                y, hidden = model(g, feats[t], w_eid, hidden)
                cost = cost + torch.mean((y - targets[t]) ** 2)
                hs.append(hidden.detach().clone())
            cost = cost / (B + 1)        # static-temporal-tgcn/seastar/train.py:183
            cost.backward()
            cost_hist.append(cost.detach().clone())
            d.update({f"B{B}_w{w0}_grad_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
        d[f"B{B}_hidden"] = torch.stack(hs)
        d[f"B{B}_cost"] = torch.stack(cost_hist)
        st = model.temporal.conv_z.stgraph._ctx_map["nb_compute"]._executor_cache.ts
        assert len(st.tensor_map_stack.content) == 0 and len(st.graph_timestamp_stack.content) == 0

    # short Adam training loop (2 epochs, B=3) : per-window loss + parameter checksums after each step
    torch.manual_seed(3100)
    model = RefTGCNModel(fin, hid, 1)
    d.update(params_dict(model, "train_param0_"))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses, sums = [], []
    B = 3
    for epoch in range(2):
        for w0 in range(0, T, B):
            opt.zero_grad()
            hidden, cost = None, 0
            for t in range(w0, w0 + B):
                y, hidden = model(g, feats[t], w_eid, hidden)
                cost = cost + torch.mean((y - targets[t]) ** 2)
            cost = cost / (B + 1)
            cost.backward()
            opt.step()
            losses.append(cost.detach().clone())
            sums.append(torch.stack([p.detach().double().sum() for p in model.parameters()]))
    d["train_losses"] = torch.stack(losses)
    d["train_param_sums"] = torch.stack(sums)
    d.update(params_dict(model, "train_paramT_"))
    save("tgcn.npz", d)


def gen_naive_tgcn():
    n, e0, fin, hid, T = 40, 200, 8, 16, 4
    rng = np.random.default_rng(14)
    cur = set(random_edges(rng, n, e0, hub=1, isolated=(11,)))
    snaps = []
    for t in range(T):
        if t > 0:
            lst = sorted(cur)
            rng.shuffle(lst)
            for p in lst[: e0 // 10]:
                cur.discard(tuple(p))
            while len(cur) < e0:
                s, dd = int(rng.integers(n)), int(rng.integers(n))
                if s != 11 and dd != 11:
                    cur.add((s, dd))
        lst = [tuple(map(int, p)) for p in cur]
        rng.shuffle(lst)
        snaps.append([(int(a), int(b)) for a, b in lst])
    d = dict(num_nodes=n, T=T)
    for t in range(T):
        arr = np.array(snaps[t], np.int32)
        d[f"t{t}_src"], d[f"t{t}_dst"] = arr[:, 0], arr[:, 1]
    G = NaiveGraph([list(s) for s in snaps], n)
    for t in range(T):
        for side, csr in (("fwd", G._forward_graph[t]), ("bwd", G._backward_graph[t])):
            for k, v in csr_arrays(csr, n, len(snaps[t])).items():
                d[f"t{t}_{side}_{k}"] = v
    feats = torch.from_numpy(rng.standard_normal((T, n, fin)).astype(np.float32))
    targets = torch.from_numpy(rng.standard_normal((T, n, 1)).astype(np.float32))
    d["feats"], d["targets"] = feats, targets
    torch.manual_seed(4000)
    model = RefTGCNModel(fin, hid, 1)
    d.update(params_dict(model, "param_"))
    G.reset_graph()
    hidden, cost, hs = None, 0, []
    for t in range(T):
        G.get_graph(t)
        if G.get_ndata("norm") is None:      # dynamic-temporal-tgcn/seastar/train.py:213-218
            G.set_ndata("norm", norm_of(G))
        d[f"t{t}_norm"] = G.get_ndata("norm")
        y, hidden = model(G, feats[t], None, hidden)
        cost = cost + torch.mean((y - targets[t]) ** 2)
        hs.append(hidden.detach().clone())
    cost = cost / (T + 1)
    cost.backward()
    d["hidden"], d["cost"] = torch.stack(hs), cost.detach()
    d.update({f"grad_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
    assert G.current_timestamp == 0          # backward walked the snapshots back to t=0
    save("naive_tgcn.npz", d)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, fn in (("csr", gen_csr), ("gcn", gen_gcn), ("gat", gen_gat), ("tgcn", gen_tgcn), ("naive_tgcn", gen_naive_tgcn),
                     ("gcn_n200", gen_gcn_n200), ("gcn_cora", gen_gcn_cora), ("gat_cora", gen_gat_cora)):
        if not only or name in only:
            fn()
    print("emitted CUDA translation units compiled through the SIMT header:", len(rh.EMITTED_SOURCES))
