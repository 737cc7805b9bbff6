#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss).

Launches, in this order (find them by dispatch order in the counter CSV):
  A. calibration: gcn_agg on an IDENTITY graph (row i has the single neighbour i), |V| = 4M,
     F = 128: every x row is read exactly once in order, so the kernel's HBM reads are known
     (4*N*F + index arrays) with the SAME access shape as the real gather (one 512-B row per
     32 lanes, 16 B per lane).  FETCH_SIZE / known bytes = the gfx950 correction for this shape.
  B. the BASELINE configs[1] launch: |V| = 1M, |E| = 16M, F = 128, forward CSR.
  C. the same on the backward CSR.
Each is launched --iters times.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_graph
from stgraph_amd import kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    F = 128
    # A: identity graph
    n_id = 4_000_000
    idx = torch.arange(n_id, device=dev, dtype=torch.int32)
    gi = kernels.build_graph_csr(idx, idx, n_id, dev)
    x = torch.randn(n_id, F, device=dev)
    ones = torch.ones(n_id, 1, device=dev)
    for _ in range(args.iters):
        kernels.gcn_agg(x, ones, ones, gi.fwd)
    torch.cuda.synchronize()
    known_read = 4 * n_id * F + 4 * (n_id + 1) + 4 * n_id + 4 * n_id + 4 * n_id
    known_write = 4 * n_id * F
    del gi, x, ones
    torch.cuda.empty_cache()
    # B/C: cfg2
    n, e = 1_000_000, 16_000_000
    src, dst = synthetic_graph(n, e, 1, dev)
    g = kernels.build_graph_csr(src, dst, n, dev)
    x = torch.randn(n, F, device=dev)
    norm = torch.rand(n, 1, device=dev) + 0.5
    for csr in (g.fwd, g.bwd):
        for _ in range(args.iters):
            kernels.gcn_agg(x, norm, norm, csr)
    torch.cuda.synchronize()
    print(json.dumps({"calibration": {"N": n_id, "F": F, "known_read_bytes": known_read, "known_write_bytes": known_write},
                      "cfg2": {"N": n, "E": e, "F": F,
                               "algorithmic_bytes": kernels.gcn_agg_algorithmic_bytes(n, e, F, False),
                               "compulsory_bytes": 8 * n * F + 4 * (n + 1) + 4 * e + 8 * n},
                      "iters": args.iters}))


if __name__ == "__main__":
    main()
