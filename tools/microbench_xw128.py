#!/usr/bin/env python3
"""cfg2's layer at 128 -> 128: aggregate-then-transform in ONE kernel ((A_hat x) W, W = 64 KB in LDS, h never written)
against what the layer runs today (rocBLAS x W, then gcn_agg at width 128).  VERDICT r01 item 7."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_graph
from stgraph_amd import _C, kernels


def med(fn, iters=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    dev = torch.device("cuda", 0)
    n, e, f = 1_000_000, 16_000_000, 128
    src, dst = synthetic_graph(n, e, 1, dev)
    g = kernels.build_graph_csr(src, dst, n, dev)
    x = torch.randn(n, f, device=dev)
    W = torch.randn(f, f, device=dev) * 0.1
    norm = torch.rand(n, 1, device=dev) + 0.5
    res = {"N": n, "E": e, "F": f}
    res["gemm_ms"] = med(lambda: torch.mm(x, W))
    h = torch.mm(x, W)
    res["gcn_agg_ms"] = med(lambda: kernels.gcn_agg(h, norm, norm, g.fwd))
    for rows in (64, 32):
        _C.set_tuning("xw_rows", rows)
        res[f"agg_transform_rows{rows}_ms"] = med(lambda: kernels.gcn_agg_transform(x, W, norm, norm, g.fwd, want_p=False))
    _C.set_tuning("xw_rows", 0)
    out, _ = kernels.gcn_agg_transform(x, W, norm, norm, g.fwd, want_p=False)
    ref = kernels.gcn_agg(h, norm, norm, g.fwd)
    res["max_rel_diff"] = float((out - ref).abs().max() / ref.abs().max())
    res["two_kernels_ms"] = res["gemm_ms"] + res["gcn_agg_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
