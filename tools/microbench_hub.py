#!/usr/bin/env python3
"""gcn_agg on skewed graphs: one Cora-shaped graph (degree-168 hub) and low-degree graphs with ONE hub of degree D.
Device time per launch from a HIP-graph replay of 20 back-to-back launches (includes the ~1.5 us launch boundary)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import _C, kernels
from bench import cora_shaped


def per_launch_us(fn, reps=20, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3 / reps


def hub_graph(n, hub_deg, seed=0):
    rng = np.random.default_rng(seed)
    keys = rng.choice(n * n, size=4 * n, replace=False)
    s, d = keys // n, keys % n
    keep = d != 0
    s, d = s[keep], d[keep]
    hs = rng.choice(n - 1, size=hub_deg, replace=False) + 1
    return np.concatenate([s, hs]).astype(np.int32), np.concatenate([d, np.zeros(hub_deg, np.int64)]).astype(np.int32)


def main():
    dev = torch.device("cuda", 0)
    cases = [("cora", *cora_shaped(), 2708)]
    for D in (8, 17, 64, 128, 168, 1000):
        s, d = hub_graph(4096, D)
        cases.append((f"hub{D}", s, d, 4096))
    for name, src, dst, n in cases:
        g = kernels.build_graph_csr(src.astype(np.int32), dst.astype(np.int32), n, dev)
        norm = torch.rand(n, 1, device=dev) + 0.5
        for F in (16, 7, 64):
            x = torch.randn(n, F, device=dev)
            row = {"graph": name, "F": F}
            ref = None
            for label, thr in (("off", 1 << 30), ("t8", 8), ("t16", 16), ("t32", 32)):
                _C.set_tuning("gcn_long_threshold", thr)
                o = kernels.gcn_agg(x, norm, norm, g.fwd)
                ref = o.clone() if ref is None else ref
                assert torch.equal(o, ref)
                row[label + "_us"] = round(per_launch_us(lambda: kernels.gcn_agg(x, norm, norm, g.fwd)), 2)
            _C.set_tuning("gcn_long_threshold", 0)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
