#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes on the narrow-row case: gcn_agg forward on Cora x 1024,
F = 16 and F = 7, --iters launches each (find them by dispatch order / kernel name in the counter CSV)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import cora_shaped
from stgraph_amd import kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--feats", default="16,7")
    ap.add_argument("--pipe", action="store_true", help="narrow rows on the persistent software-pipelined tile kernel")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.pipe:
        from stgraph_amd import _C
        _C.set_tuning("gcn_tile_pipe", 2)
    src, dst = cora_shaped()
    n, K = 2708, args.K
    big_src = np.concatenate([src + k * n for k in range(K)]).astype(np.int32)
    big_dst = np.concatenate([dst + k * n for k in range(K)]).astype(np.int32)
    g = kernels.build_graph_csr(big_src, big_dst, n * K, dev)
    N, E = n * K, len(big_src)
    norm = torch.rand(N, 1, device=dev) + 0.5
    out = {}
    for F in map(int, args.feats.split(",")):
        x = torch.randn(N, F, device=dev)
        for _ in range(args.iters):
            kernels.gcn_agg(x, norm, norm, g.fwd)
        torch.cuda.synchronize()
        out[f"F{F}"] = {"N": N, "E": E, "algorithmic_bytes": kernels.gcn_agg_algorithmic_bytes(N, E, F, False)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
