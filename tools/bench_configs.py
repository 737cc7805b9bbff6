#!/usr/bin/env python3
"""Measure every BASELINE.json configuration on one MI355X and print one JSON line each.

  cfg1  2-layer GCN 1433->16->7 on a Cora-SHAPED synthetic graph (|V|=2708, |E|=10556), epochs/s,
        plus the roofline variant "Cora x K": K disjoint replicas so the kernel is bandwidth- not
        launch-bound
  cfg2  -> bench.py (main line)
  cfg3  GAT 8 heads, |V|=256K |E|=8M, in=64, D=64: per-kernel time + roofline fraction, layer fwd+bwd
  cfg4  -> bench.py ("tgcn" object)
  cfg5  dynamic-temporal TGCN, |V|=25K, E0=250K, +-6250 edges/step, T=40, B=20, per-snapshot device
        CSR rebuild; epochs/s and the CSR-build share
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from bench import GCN, HBM_PEAK_GBS, cora_run, cora_shaped, degree_norm, synthetic_graph
from stgraph_amd import kernels, temporal
from stgraph_amd.graph import NaiveGraph, PCSRGraph, StaticGraph


def kernel_table(records):
    out = {}
    for name, a, b, nbytes, units in records:
        d = out.setdefault(name, {"ms": [], "bytes": nbytes, "units": units})
        d["ms"].append(a.elapsed_time(b))
    return {k: {"launches": len(v["ms"]), "mean_ms": float(np.mean(v["ms"])), "algorithmic_bytes": v["bytes"],
                "GBps": v["bytes"] / np.mean(v["ms"]) / 1e6, "frac_of_hbm_peak": v["bytes"] / np.mean(v["ms"]) / 1e6 / HBM_PEAK_GBS}
            for k, v in out.items()}


def cfg1(dev, epochs=200, K=1024):
    line = {"config": "cfg1", **cora_run(dev, epochs)}
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    # roofline variant: K disjoint replicas
    big_src = np.concatenate([src + k * n for k in range(K)]).astype(np.int32)
    big_dst = np.concatenate([dst + k * n for k in range(K)]).astype(np.int32)
    gk = kernels.build_graph_csr(big_src, big_dst, n * K, dev)
    norm = torch.rand(n * K, 1, device=dev) + 0.5
    tab = {}
    for Fw in (16, 7, 64, 128):
        xk = torch.randn(n * K, Fw, device=dev)
        rec = []
        for _ in range(3):
            kernels.gcn_agg(xk, norm, norm, gk.fwd)
        kernels.enable_launch_timing(rec)
        for _ in range(10):
            kernels.gcn_agg(xk, norm, norm, gk.fwd)
            kernels.gcn_agg(xk, norm, norm, gk.bwd)
        torch.cuda.synchronize()
        kernels.enable_launch_timing(None)
        tab[f"F{Fw}"] = kernel_table(rec)["gcn_agg"]
    line[f"cora_x{K}"] = {"nodes": n * K, "edges": e * K, "gcn_agg_fwd_bwd": tab}
    return line


def cfg3(dev):
    n, e, fin, H, D = 256_000, 8_000_000, 64, 8, 64
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    src, dst = synthetic_graph(n, e, 2, dev)
    g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
    torch.manual_seed(2)
    conv = GATConv(fin, D, H).to(dev)
    x = torch.randn(n, fin, device=dev, requires_grad=True)
    R = torch.randn(n, H, D, device=dev)
    for _ in range(3):
        (conv(g, x) * R).sum().backward()
    rec = []
    kernels.enable_launch_timing(rec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 10
    for _ in range(iters):
        conv.zero_grad()
        x.grad = None
        (conv(g, x) * R).sum().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    kernels.enable_launch_timing(None)
    return {"config": f"cfg3 GATConv in={fin} H={H} D={D} on |V|={n} |E|={e}: layer fwd+bwd",
            "ms_per_fwd_bwd": dt * 1e3, "edges_feat_per_s": 2 * e * H * D / dt, "kernels": kernel_table(rec)}


def cfg5(dev, epochs=6):
    n, e0, churn, T, B, feat, hid = 25_000, 250_000, 6250, 40, 20, 32, 64
    rng = np.random.default_rng(4)
    stream = rng.choice(n * n, size=e0 + churn * T, replace=False)       # sliding window over an edge stream
    snaps, pn_edges, pn_targets = [], [], []
    for t in range(T):
        keys = stream[t * churn: t * churn + e0]
        s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
        snaps.append((torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)))
        m = 10_000
        pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(dev)
        neg = torch.randint(0, n, (2, m), device=dev)
        pn_edges.append(torch.cat([pos, neg], 1))
        pn_targets.append(torch.cat([torch.ones(m, device=dev), torch.zeros(m, device=dev)]))
    out = {}
    for mode, kw in (("rebuild_per_snapshot", dict(resident=False, max_cached=B + 1)), ("resident", dict(resident=True)),
                     ("pcsr_store", None)):
        G = PCSRGraph(snaps, n, device=dev) if kw is None else NaiveGraph(snaps, n, device=dev, sort_inplace=False, **kw)
        torch.manual_seed(4)
        model = temporal.DynamicSTGraphTGCN(feat, hid).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        bucket = temporal.GradBucket(model.parameters())
        dur = []
        for ep in range(epochs):
            if mode == "rebuild_per_snapshot":
                G._snapshots.clear()
                G._ndata.clear()
            if mode == "pcsr_store":
                G._ndata.clear()
                G.build_count, G.build_time = G._forward_graph.update_count, 0.0
            b0, bt0 = G.build_count, G.build_time
            torch.cuda.synchronize()
            t0 = time.time()
            temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
            torch.cuda.synchronize()
            if mode == "pcsr_store":
                G.build_count = G._forward_graph.update_count
                G.check()
            if ep >= 3:
                dur.append((time.time() - t0, G.build_count - b0, G.build_time - bt0))
        out[mode] = {"epochs_per_s": 1.0 / float(np.mean([d[0] for d in dur])),
                     "s_per_epoch": float(np.mean([d[0] for d in dur])),
                     "csr_builds_per_epoch": float(np.mean([d[1] for d in dur])),
                     "csr_build_host_seconds_per_epoch": float(np.mean([d[2] for d in dur]))}
    # device CSR build alone (fwd + bwd + degrees + node_ids), stream-ordered
    s, d = snaps[0]
    for _ in range(3):
        kernels.build_graph_csr(s, d, n, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        kernels.build_graph_csr(s, d, n, dev)
    torch.cuda.synchronize()
    out["device_csr_build_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    return {"config": f"cfg5 dynamic-temporal TGCN |V|={n} E0={e0} +-{churn}/step T={T} B={B} feat={feat} hidden={hid}", **out}


def csr_build_cfg2(dev):
    n, e = 1_000_000, 16_000_000
    src, dst = synthetic_graph(n, e, 1, dev)
    kernels.build_graph_csr(src, dst, n, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        kernels.build_graph_csr(src, dst, n, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    return {"config": "device CSR build (fwd+bwd+degrees+node_ids) |V|=1M |E|=16M", "ms": dt * 1e3,
            "algorithmic_bytes": 2 * 16 * e, "GBps": 2 * 16 * e / dt / 1e9}


def edge_store_cfg2(dev):
    """Dynamic edge store at |V|=1M, |E|=16M with 5 % churn per step: one update (both orientations),
    forward emit, reverse emit (structure: 8 B key read + 4 B column written per edge), labels on demand.
    Algorithmic bytes of an update = 2 orientations x (8 B read + 8 B written) per stored edge."""
    n, e, k = 1_000_000, 16_000_000, 800_000
    src, dst = synthetic_graph(n, e + k, 1, dev)
    base = kernels.edgeset_update(kernels.edgeset_empty(n, dev), src[:e], dst[:e])
    kernels.edgeset_check(base)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def timed(fn, iters=10):
        fn()
        a, b = ev(), ev()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters
    upd = timed(lambda: kernels.edgeset_update(base, src[e:], dst[e:], src[:k], dst[:k]))
    new = kernels.edgeset_update(base, src[e:], dst[e:], src[:k], dst[:k])
    kernels.edgeset_check(new)
    ak, dk = kernels.edgeset_pack_sorted(src[e:], dst[e:], dev), kernels.edgeset_pack_sorted(src[:k], dst[:k], dev)
    mrg = timed(lambda: kernels.edgeset_merge(base, ak, dk))
    m = kernels.edgeset_merge(base, ak, dk)
    kernels.edgeset_check(m)
    assert torch.equal(m.keys_fwd, new.keys_fwd) and torch.equal(m.keys_bwd, new.keys_bwd)
    fwd = timed(lambda: kernels.edgeset_emit_csr(new, False))
    bwd = timed(lambda: kernels.edgeset_emit_csr(new, True))
    lab_f = timed(lambda: kernels.edgeset_emit_csr(new, False).eids) - fwd
    lab_b = timed(lambda: kernels.edgeset_emit_csr(new, True).eids) - bwd
    full = timed(lambda: kernels.build_graph_csr(src[k:], dst[k:], n, dev), 5)
    return {"config": f"dynamic edge store |V|={n} |E|={e} +-{k} edges per step",
            "update_unsorted_lists_ms": upd, "merge_presorted_ms": mrg, "merge_GBps": 2 * 16 * e / mrg / 1e6,
            "emit_fwd_ms": fwd, "emit_fwd_GBps": 12 * e / fwd / 1e6,
            "emit_bwd_ms": bwd, "emit_bwd_GBps": 12 * e / bwd / 1e6,
            "labels_fwd_ms (on demand)": lab_f, "labels_bwd_ms (on demand)": lab_b,
            "step_forward_ms": mrg + fwd, "step_backward_ms": mrg + bwd, "full_rebuild_ms (NaiveGraph path: 2 radix sorts)": full}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="cfg1,cfg3,cfg5,csr,store")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fns = {"cfg1": cfg1, "cfg3": cfg3, "cfg5": cfg5, "csr": csr_build_cfg2, "store": edge_store_cfg2}
    for k in args.only.split(","):
        print(json.dumps(fns[k](dev)), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
