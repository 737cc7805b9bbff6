#!/usr/bin/env python3
"""gcn_agg on the Cora x K roofline graph (narrow features, low degree): knob sweep."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import _C, kernels
from bench import cora_shaped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--feats", default="16,7,32,64")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-long", action="store_true", help="switch the long-row workgroups off")
    args = ap.parse_args()
    kernels.set_long_row_path(not args.no_long)
    dev = torch.device("cuda", 0)
    src, dst = cora_shaped()
    n, K = 2708, args.K
    big_src = np.concatenate([src + k * n for k in range(K)]).astype(np.int32)
    big_dst = np.concatenate([dst + k * n for k in range(K)]).astype(np.int32)
    g = kernels.build_graph_csr(big_src, big_dst, n * K, dev)
    N, E = n * K, len(big_src)
    norm = torch.rand(N, 1, device=dev) + 0.5
    for F in map(int, args.feats.split(",")):
        x = torch.randn(N, F, device=dev)
        nbytes = kernels.gcn_agg_algorithmic_bytes(N, E, F, False)
        ref = None
        for window in (1, 0):                 # 1 = per-edge scalars pre-gathered (stg_gcn_agg_edge)
            for nid in (False, True):
                for unroll in (8, 4):
                    kernels.set_edge_cache(bool(window))
                    _C.set_tuning("gcn_unroll", unroll)
                    for _ in range(2):
                        o = kernels.gcn_agg(x, norm, norm, g.fwd, use_node_ids=nid)
                    if ref is None:
                        ref = o.clone()
                    assert torch.equal(o, ref)
                    ts = []
                    for _ in range(args.iters):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        kernels.gcn_agg(x, norm, norm, g.fwd, use_node_ids=nid)
                        b.record()
                        torch.cuda.synchronize()
                        ts.append(a.elapsed_time(b))
                    med = float(np.median(ts))
                    print(json.dumps({"F": F, "edge_cache": window, "node_ids": nid, "unroll": unroll, "ms": round(med, 4),
                                      "GBps_alg": round(nbytes / med / 1e6, 1), "frac": round(nbytes / med / 8e9, 3)}),
                          flush=True)
    kernels.set_edge_cache(True)
    _C.set_tuning("gcn_unroll", 0)


if __name__ == "__main__":
    main()
