"""Golden vectors for the MODEL-level paths the benchmark times, from the reference itself (round 3).

BUILD CONTAINER ONLY (reads /root/reference through tests/golden/ref_harness.py).
Usage:   python tests/golden/make_golden_models.py [tgcn_native] [gcn_model] [gat_model] [dyn_tgcn]

The model classes are the reference's own (benchmarking/*/seastar/model.py, imported from their files); the training
loops restate benchmarking/*/seastar/train.py (whose imports do not resolve in this snapshot) line for line.

  tgcn_native.npz  static-temporal loop (static-temporal-tgcn/seastar/train.py:160-187, model.py:6-18) at the
                   benchmark's NATIVE widths: N = 4096, E = 40960, feat 32 -> hidden 64, T = 6, B in {3, 6}, with and
                   without edge weights: per-snapshot hidden state / y_hat / y_out, the window costs, all 16 parameter
                   gradients of every window, and a 2-epoch Adam loop (B = 3): per-window cost + parameters at the end.
                   [T, N, *] tensors are kept on 296 sampled rows (the 40 highest in-degree + 256 drawn) together with
                   fp64 column sums and abs column sums over ALL rows.
  gcn_model.npz    benchmarking/gcn/seastar/model.py GCN (1 hidden layer, ReLU) on the Cora-shaped graph of the bench
                   (N = 2708, E = 10556), loop of gcn/seastar/train.py:63-101 (CrossEntropyLoss on the first 60 % rows,
                   Adam lr 1e-2 wd 5e-4): widths 128-128-128 (the cfg2 model), 1433-16-8 and 1433-16-7 (cfg1; the last
                   one carries reference defect D1): logits, loss and every gradient at step 0, loss of steps 0..3,
                   parameters after 3 steps; (round 4) the parameters and gradients of steps 1 and 2.
  gat_model.npz    benchmarking/gat/seastar/model.py GAT (1 hidden layer, heads [8, 1], ELU) on the same graph, loop of
                   gat/seastar/train.py:95-125: shapes in 32 -> 8 x 8 -> 16 and in 64 -> 8 x 64 -> 16 (the cfg3 layer
                   widths): logits, loss, every gradient at step 0, losses of steps 0..3, parameters after 3 steps;
                   (round 4) the parameters and gradients of steps 1 and 2.
  dyn_tgcn.npz     dynamic-temporal loop (dynamic-temporal-tgcn/seastar/train.py:179-231, model.py:5-21) on a
                   NaiveGraph: N = 4096, E_t = 32768 with 5 % churn per step, T = 7, B in {3, 6}, feat 32 -> hidden 64,
                   link-prediction head on 2048 positive + 2048 negative label edges per snapshot: window costs, every
                   parameter gradient per window, sampled rows + column sums of per-snapshot hidden states.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.load_reference()
from stgraph.graph import NaiveGraph, StaticGraph  # noqa: E402

torch.set_num_threads(1)
REF_BENCH = os.path.join(rh.REFERENCE_ROOT, "benchmarking")


def ref_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_BENCH, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in d.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


def norm_of(g, weighted=False):
    deg = torch.from_numpy(np.asarray(g.weighted_in_degrees() if weighted else g.in_degrees()))
    norm = torch.pow(deg.type(torch.int32) if weighted else deg.float(), -0.5)   # gcn/seastar/train.py:53-57 (int32 there)
    norm[torch.isinf(norm)] = 0
    return norm.float().unsqueeze(1)


# No ReLU input (hidden state) of a recorded run lies closer to zero than this.  Hidden states here are ~0.07 in size, the two
# products they are summed from <= 0.1, so fp32 evaluation-order noise on them is ~1e-8; ~7 values per 3-snapshot window fall
# below 5e-7 whatever the input, ~1 below 1e-7 -- the margin is 10x the noise and still reachable by re-drawing the input.
TIE_MARGIN = 1e-7


def uniform_edges(seed, n, e):
    """Duplicate-free directed edges in random order (numpy PCG64: same bits on every host)."""
    rng = np.random.default_rng(seed)
    keys = rng.choice(n * n, size=e, replace=False)
    return (keys // n).astype(np.int32), (keys % n).astype(np.int32)


def sample_rows(deg, count=256, top=40, seed=0):
    rng = np.random.default_rng(seed)
    hubs = np.argsort(-deg, kind="stable")[:top]
    rest = rng.choice(len(deg), size=min(count, len(deg)), replace=False)
    return np.unique(np.concatenate([hubs, rest])).astype(np.int64)


def put_sampled(d, key, t, rows):
    """``t``: [..., N, C] -> rows in full + fp64 column sums / abs column sums over all N."""
    t = t.detach()
    d[key + "_rows"] = t[..., rows, :]
    d[key + "_colsum"] = t.double().sum(-2)
    d[key + "_abs_colsum"] = t.double().abs().sum(-2)


def cora_shaped(seed=0, n=2708, pairs=5278, max_deg=168):
    """bench.py::cora_shaped restated (as in make_golden.py)."""
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


# ----------------------------------------------------------------------------- static-temporal TGCN, native widths
def gen_tgcn_native():
    STGraphTGCN = ref_module("static-temporal-tgcn/seastar/model.py", "ref_static_tgcn_model").STGraphTGCN
    n, e, feat, hid, T = 4096, 40960, 32, 64, 6
    src, dst = uniform_edges(31, n, e)
    el = [(int(a), int(b)) for a, b in zip(src, dst)]
    rng = np.random.default_rng(32)
    w_np = rng.uniform(0.5, 1.5, (e, 1)).astype(np.float32)          # indexed by eid = position in (dst, src) order
    w_eid = torch.from_numpy(w_np)
    g = StaticGraph(list(el), w_np.reshape(-1).tolist(), n)
    norm = norm_of(g)
    g.set_ndata("norm", norm)
    deg = np.asarray(g.in_degrees())
    rows = sample_rows(deg)
    targets = torch.from_numpy(np.random.default_rng(33).standard_normal((T, n, 1), dtype=np.float32))
    d = dict(num_nodes=n, src=src, dst=dst, edge_weight_by_eid=w_eid, norm=norm, rows=rows, in_degrees=deg,
             targets_seed=33, feat=feat, hidden=hid, T=T)

    def x0_of(seed):                                                   # the loop's torch.randn, re-drawable anywhere
        return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, feat), dtype=np.float32))

    def run_window(model, ew, B, index, seed):
        """One BPTT window of the reference loop (train.py:160-187) from ``torch.randn`` stand-in ``x0_of(seed)``."""
        model.zero_grad()
        cost, hidden = 0, None
        y_hat = x0_of(seed)
        hs, ys, youts = [], [], []
        for k in range(B):
            t = index * B + k
            y_out, y_hat, hidden = model(g, y_hat, ew, hidden)
            cost = cost + torch.mean((y_out - targets[t]) ** 2)
            hs.append(hidden.detach().clone())
            ys.append(y_hat.detach().clone())
            youts.append(y_out.detach().clone())
        cost = cost / (B + 1)                                   # train.py:183
        cost.backward()
        return cost.detach().clone(), hs, ys, youts

    for use_ew in (False, True):
        ew = w_eid if use_ew else None
        for B in (3, 6):
            tag = f"{'ew' if use_ew else 'now'}_B{B}"
            torch.manual_seed(7000 + B + (100 if use_ew else 0))
            model = STGraphTGCN(feat, hid, 1)
            d.update({f"{tag}_param_{k}": p.detach().clone() for k, p in model.named_parameters()})
            hs, ys, youts, costs, seeds, margins = [], [], [], [], [], []
            for index in range(T // B):
                # The head applies ReLU to the hidden state (model.py:13).  A hidden value within fp32 rounding of zero
                # falls on either side of the kink depending on summation order (the reference's own FMA build and this
                # no-FMA emulation would disagree there), which moves gradients by one row's share (~1e-3 of their
                # size at N = 4096).  So the window's input is re-drawn until no hidden value lies within TIE_MARGIN of
                # zero: the comparison is then independent of rounding.  The seed used is recorded.
                seed = 7100 + 1000 * index
                while True:
                    cost, h_w, y_w, yo_w = run_window(model, ew, B, index, seed)
                    margin = float(torch.stack(h_w).abs().min())
                    if margin >= TIE_MARGIN:
                        break
                    seed += 1
                seeds.append(seed), margins.append(margin), costs.append(cost)
                hs += h_w; ys += y_w; youts += yo_w
                d.update({f"{tag}_w{index}_grad_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
            d[f"{tag}_cost"] = torch.stack(costs)
            d[f"{tag}_x0_seeds"], d[f"{tag}_min_abs_hidden"] = np.array(seeds), np.array(margins)
            put_sampled(d, f"{tag}_hidden", torch.stack(hs), rows)
            put_sampled(d, f"{tag}_y", torch.stack(ys), rows)
            put_sampled(d, f"{tag}_yout", torch.stack(youts), rows)
            print(tag, "costs", [float(c) for c in costs], "seeds", seeds, "min|h|", margins, flush=True)

    # 2 epochs of the training loop (B = 3, edge weights, Adam lr 1e-2): window costs + final parameters
    torch.manual_seed(7500)
    model = STGraphTGCN(feat, hid, 1)
    d.update({f"train_param0_{k}": p.detach().clone() for k, p in model.named_parameters()})
    B = 3
    while True:                                             # re-drawn as a whole until all 4 windows are tie-free
        base = d.get("train_x0_seed_base", 7590) + 10
        d["train_x0_seed_base"] = base
        torch.manual_seed(7500)
        model = STGraphTGCN(feat, hid, 1)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        costs, margin, step_grads = [], 1.0, {}
        for epoch in range(2):
            for index in range(T // B):
                opt.zero_grad()
                cost, hidden = 0, None
                y_hat = x0_of(base + epoch * 2 + index)
                for k in range(B):
                    y_out, y_hat, hidden = model(g, y_hat, w_eid, hidden)
                    cost = cost + torch.mean((y_out - targets[index * B + k]) ** 2)
                    margin = min(margin, float(hidden.detach().abs().min()))
                cost = cost / (B + 1)
                cost.backward()
                # round 4: the gradient every optimizer step consumed (the GPU test names the entries whose gradient is
                # rounding noise -- where Adam's g / (|g| + eps) is ill-conditioned -- by THESE values)
                step_grads.update({f"train_grad{len(costs)}_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
                opt.step()
                costs.append(cost.detach().clone())
        print("train loop base", base, "min|h|", margin, flush=True)
        if margin >= TIE_MARGIN:
            break
    d["train_min_abs_hidden"] = margin
    d.update(step_grads)
    d["train_costs"] = torch.stack(costs)
    d.update({f"train_paramT_{k}": p.detach().clone() for k, p in model.named_parameters()})
    save("tgcn_native.npz", d)


# ----------------------------------------------------------------------------- GCN model (cfg1 / cfg2 model shapes)
def gen_gcn_model():
    import torch.nn.functional as F
    GCN = ref_module("gcn/seastar/model.py", "ref_gcn_model").GCN
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    el = [(int(a), int(b)) for a, b in zip(src, dst)]
    g = StaticGraph(list(el), [1] * e, n)
    norm = norm_of(g, weighted=True)
    g.set_ndata("norm", norm)
    deg = np.asarray(g.in_degrees())
    rows = sample_rows(deg)
    ntrain = int(0.6 * n)
    d = dict(num_nodes=n, src=src, dst=dst, norm=norm, rows=rows, ntrain=ntrain)
    for (fin, hid, out) in ((128, 128, 128), (1433, 16, 8), (1433, 16, 7)):
        tag = f"w{fin}_{hid}_{out}"
        seed = 8000 + fin + out
        rng = np.random.default_rng(seed)
        if fin == 1433:
            x = torch.from_numpy((rng.random((n, fin)) < 0.0127).astype(np.float32))      # Cora-like bag of words
        else:
            x = torch.from_numpy(rng.standard_normal((n, fin), dtype=np.float32))
        labels = torch.from_numpy(rng.integers(0, out, n).astype(np.int64))
        d[tag + "_seed"] = seed
        torch.manual_seed(seed)
        model = GCN(g, fin, hid, out, 1, F.relu)
        d.update({f"{tag}_param0_{k}": p.detach().clone() for k, p in model.named_parameters()})
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4)
        loss_fcn = torch.nn.CrossEntropyLoss()
        losses = []
        for step in range(4):
            logits = model(g, x)
            loss = loss_fcn(logits[:ntrain], labels[:ntrain])
            opt.zero_grad()
            loss.backward()
            if step == 0:
                d[tag + "_logits_rows"] = logits.detach()[rows]
                d[tag + "_logits_colsum"] = logits.detach().double().sum(0)
                d[tag + "_logits_abs_colsum"] = logits.detach().double().abs().sum(0)
                d.update({f"{tag}_grad0_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
            elif step < 3:
                # round 4: the parameters every later step starts from and the gradients it produces, so that the GPU test
                # checks GRADIENTS at the reference's own parameters step by step instead of parameters after 3 Adam steps
                # (whose entries with a rounding-noise gradient move by lr in either direction)
                d.update({f"{tag}_param{step}_{k}": p.detach().clone() for k, p in model.named_parameters()})
                d.update({f"{tag}_grad{step}_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
            losses.append(loss.detach().clone())
            if step < 3:
                opt.step()
        d[tag + "_losses"] = torch.stack(losses)
        d.update({f"{tag}_param3_{k}": p.detach().clone() for k, p in model.named_parameters()})
        print(tag, "losses", [float(v) for v in losses], flush=True)
    save("gcn_model.npz", d)


# ----------------------------------------------------------------------------- GAT model (cfg3 model shape)
def gen_gat_model():
    import torch.nn.functional as F
    GAT = ref_module("gat/seastar/model.py", "ref_gat_model").GAT
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    el = [(int(a), int(b)) for a, b in zip(src, dst)]
    g = StaticGraph(list(el), [1] * e, n)
    deg = np.asarray(g.in_degrees())
    rows = sample_rows(deg)
    ntrain = int(0.6 * n)
    d = dict(num_nodes=n, src=src, dst=dst, rows=rows, ntrain=ntrain)
    for (fin, D, H, classes) in ((32, 8, 8, 16), (64, 64, 8, 16)):
        tag = f"in{fin}_H{H}_D{D}"
        seed = 9000 + fin + D
        rng = np.random.default_rng(seed)
        x = torch.from_numpy(rng.standard_normal((n, fin), dtype=np.float32))
        labels = torch.from_numpy(rng.integers(0, classes, n).astype(np.int64))
        d[tag + "_seed"] = seed
        torch.manual_seed(seed)
        model = GAT(g, 1, fin, D, classes, [H, 1], F.elu, 0.0, 0.0, 0.2, False)
        d.update({f"{tag}_param0_{k}": p.detach().clone() for k, p in model.named_parameters()})
        opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
        loss_fcn = torch.nn.CrossEntropyLoss()
        losses = []
        for step in range(4):
            logits = model(x)
            loss = loss_fcn(logits[:ntrain], labels[:ntrain])
            opt.zero_grad()
            loss.backward()
            if step == 0:
                d[tag + "_logits_rows"] = logits.detach()[rows]
                d[tag + "_logits_colsum"] = logits.detach().double().sum(0)
                d[tag + "_logits_abs_colsum"] = logits.detach().double().abs().sum(0)
                d.update({f"{tag}_grad0_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
            elif step < 3:
                # round 4: the parameters every later step starts from and the gradients it produces, so that the GPU test
                # checks GRADIENTS at the reference's own parameters step by step instead of parameters after 3 Adam steps
                # (whose entries with a rounding-noise gradient move by lr in either direction)
                d.update({f"{tag}_param{step}_{k}": p.detach().clone() for k, p in model.named_parameters()})
                d.update({f"{tag}_grad{step}_{k}": p.grad.detach().clone() for k, p in model.named_parameters()})
            losses.append(loss.detach().clone())
            if step < 3:
                opt.step()
        d[tag + "_losses"] = torch.stack(losses)
        d.update({f"{tag}_param3_{k}": p.detach().clone() for k, p in model.named_parameters()})
        print(tag, "losses", [float(v) for v in losses], flush=True)
    save("gat_model.npz", d)


# ----------------------------------------------------------------------------- dynamic-temporal TGCN, native widths
def gen_dyn_tgcn():
    DynModel = ref_module("dynamic-temporal-tgcn/seastar/model.py", "ref_dyn_tgcn_model").STGraphTGCN
    n, e0, feat, hid, T, M = 4096, 32768, 32, 64, 7, 2048
    rng = np.random.default_rng(41)
    stream = rng.choice(n * n, size=e0 + T * (e0 // 20), replace=False)
    churn = e0 // 20
    snaps = []
    for t in range(T):                                     # sliding window over the stream (preprocess_temporal_data.py:46-126)
        keys = stream[t * churn: t * churn + e0]
        keys = keys[rng.permutation(e0)]
        snaps.append(np.stack([keys // n, keys % n], 1).astype(np.int32))
    d = dict(num_nodes=n, T=T, feat=feat, hidden=hid, M=M)
    for t in range(T):
        d[f"t{t}_src"], d[f"t{t}_dst"] = snaps[t][:, 0], snaps[t][:, 1]
    G = NaiveGraph([[(int(a), int(b)) for a, b in s] for s in snaps], n)
    # label edges: M edges of snapshot t + 1 as positives, M uniformly drawn pairs as negatives (train.py:120-150 shape)
    edges, targets = [], []
    for t in range(T - 1):
        pos = snaps[t + 1][rng.choice(e0, M, replace=False)].T.astype(np.int64)
        neg = rng.integers(0, n, (2, M)).astype(np.int64)
        ei = np.concatenate([pos, neg], 1)
        edges.append(torch.from_numpy(np.ascontiguousarray(ei)))
        targets.append(torch.cat([torch.ones(M), torch.zeros(M)]))
        d[f"t{t}_label_edges"] = ei
    edges.append(edges[-1]); targets.append(targets[-1])
    criterion = torch.nn.BCEWithLogitsLoss()
    deg0 = np.bincount(snaps[0][:, 1], minlength=n)
    rows = sample_rows(deg0)
    d["rows"] = rows

    def x0_of(seed):
        return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, feat), dtype=np.float32))

    def run_epoch(G, model, B, seeds):
        """One epoch of the reference loop (train.py:179-231) -> per-window (cost, grads, hidden states)."""
        G.reset_graph()
        out = []
        for index in range((T + B - 1) // B):
            model.zero_grad()
            cost, hidden = 0, None
            y_hat = x0_of(seeds[index])
            G.get_graph(index * B)
            hs = []
            for k in range(B):
                t = index * B + k
                if t >= T - 1:
                    break
                G.get_graph(t)
                if G.get_ndata("norm") is None:
                    G.set_ndata("norm", norm_of(G))
                y_hat, hidden = model(G, y_hat, None, hidden)
                outp = model.decode(y_hat, edges[t]).view(-1)
                cost = cost + criterion(outp, targets[t])
                hs.append(hidden.detach().clone())
            if isinstance(cost, int):
                break
            cost = cost / (B + 1)
            cost.backward()
            out.append((cost.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, hs))
        return out

    for B in (3, 6):
        tag = f"B{B}"
        nwin = (T + B - 1) // B
        seeds = [7800 + 1000 * i for i in range(nwin)]
        while True:
            # tie-free hidden states (see gen_tgcn_native).  A FRESH graph AND model per try: a layer's executor keeps the
            # graph object of its first call (compiler/stgraph.py:221-226 caches the Context, executor.py holds .graph), so a
            # model cannot move to a second NaiveGraph, and a second epoch on the same graph hits defect D12 below.
            torch.manual_seed(7700 + B)
            model = DynModel(feat, hid)
            d.update({f"{tag}_param_{k}": p.detach().clone() for k, p in model.named_parameters()})
            G = NaiveGraph([[(int(a), int(b)) for a, b in sn] for sn in snaps], n)
            res = run_epoch(G, model, B, seeds)
            bad = [i for i, (_, _, hs) in enumerate(res) if float(torch.stack(hs).abs().min()) < TIE_MARGIN]
            if not bad:
                break
            for i in bad:
                seeds[i] += 1
        hs_all = []
        for index, (cost, grads, hs) in enumerate(res):
            d.update({f"{tag}_w{index}_grad_{k}": v for k, v in grads.items()})
            hs_all += hs
        d[f"{tag}_cost"] = torch.stack([c for c, _, _ in res])
        d[f"{tag}_x0_seeds"] = np.array(seeds[: len(res)])
        d[f"{tag}_min_abs_hidden"] = float(torch.stack(hs_all).abs().min())
        put_sampled(d, f"{tag}_hidden", torch.stack(hs_all), rows)
        print(tag, "costs", [float(c) for c, _, _ in res], "seeds", seeds, flush=True)
        if B == 3:
            # Reference defect (recorded, NOT a parity target; DESIGN.md "D12"): a SECOND epoch on the same NaiveGraph.
            # reset_graph() sets current_timestamp = 0 but does not reload the forward CSR pointers, and get_graph(0)
            # then has nothing to step (dynamic_graph.py:81-107, naive_graph.py:103-139), so snapshot 0's forward pass
            # runs over the LAST snapshot the previous epoch's forward walk reached (t = 6 = T - 1 here: the loop calls
            # get_graph(index * backprop_every) before it finds the window empty) while its backward pass uses snapshot
            # 0's own reverse CSR.  Only window 0 of the epoch is affected.
            res2 = run_epoch(G, model, B, seeds)
            d["stale_B3_cost"] = torch.stack([c for c, _, _ in res2])
            d.update({f"stale_B3_w0_grad_{k}": v for k, v in res2[0][1].items()})
            d["stale_forward_snapshot"] = ((T + B - 1) // B - 1) * B
            print("second epoch on the same graph: costs", [float(c) for c, _, _ in res2], flush=True)
    save("dyn_tgcn.npz", d)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, fn in (("tgcn_native", gen_tgcn_native), ("gcn_model", gen_gcn_model), ("gat_model", gen_gat_model),
                     ("dyn_tgcn", gen_dyn_tgcn)):
        if not only or name in only:
            fn()
    print("emitted CUDA translation units compiled through the SIMT header:", len(rh.EMITTED_SOURCES))
