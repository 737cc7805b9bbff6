#!/usr/bin/env python3
"""Kernel-only timing of stg_gcn_agg over the tuning knobs (HIP events, median of N launches)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_graph
from stgraph_amd import _C, kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ew", action="store_true")
    ap.add_argument("--configs", default="0:0:1,0:0:0,64:4:1,64:4:0,64:2:1,32:4:1")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    src, dst = synthetic_graph(args.nodes, args.edges, 1, dev)
    g = kernels.build_graph_csr(src, dst, args.nodes, dev)
    x = torch.randn(args.nodes, args.feat, device=dev)
    norm = torch.rand(args.nodes, 1, device=dev) + 0.5
    ew = (torch.rand(args.edges, 1, device=dev) + 0.5) if args.ew else None
    nbytes = kernels.gcn_agg_algorithmic_bytes(args.nodes, args.edges, args.feat, args.ew)
    for cfg in args.configs.split(","):
        parts = list(map(int, cfg.split(":")))
        lanes, unroll = parts[0], parts[1]
        kernels.set_edge_cache(bool(parts[2]) if len(parts) > 2 else True)
        _C.set_tuning("gcn_lanes_per_row", lanes)
        _C.set_tuning("gcn_unroll", unroll)
        for csr_name, csr in (("fwd", g.fwd), ("bwd", g.bwd)):
            for _ in range(2):
                kernels.gcn_agg(x, norm, norm, csr, ew=ew)
            ts = []
            for _ in range(args.iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                kernels.gcn_agg(x, norm, norm, csr, ew=ew)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            med = float(np.median(ts))
            print(json.dumps({"edge_cache": kernels._EDGE_CACHE, "lanes": lanes, "unroll": unroll, "csr": csr_name, "F": args.feat, "ms": round(med, 4),
                              "min_ms": round(min(ts), 4), "GBps_alg": round(nbytes / med / 1e6, 1),
                              "frac_of_8TBps": round(nbytes / med / 1e6 / 8000, 3)}), flush=True)


if __name__ == "__main__":
    main()
