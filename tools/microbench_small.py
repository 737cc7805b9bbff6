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
    ap.add_argument("--xcd", default="0", help="comma list of gcn_xcd_tile values (0 = auto, 1 = round robin)")
    ap.add_argument("--block", default="0", help="comma list of gcn_block values (0 = auto, 64, 128, 256)")
    ap.add_argument("--tile", default="0", help="comma list of gcn_tile values (0 = auto, 1 = off, 2 = forced)")
    ap.add_argument("--tile-rows", default="0", help="comma list of gcn_tile_rows values (0 = auto)")
    ap.add_argument("--quick", action="store_true", help="only the default configuration (edge cache, vertex order)")
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
        for window, nid, unroll, xcd, blk, tl, tr in [(w_, n_, u_, x_, b_, t_, r_) for w_ in ((1,) if args.quick else (1, 0))
                                          for n_ in ((False,) if args.quick else (False, True))
                                          for u_ in ((8,) if args.quick else (8, 4))
                                          for x_ in map(int, args.xcd.split(","))
                                          for b_ in map(int, args.block.split(","))
                                          for t_ in map(int, args.tile.split(","))
                                          for r_ in map(int, args.tile_rows.split(","))]:
                    kernels.set_edge_cache(bool(window))
                    _C.set_tuning("gcn_unroll", unroll)
                    _C.set_tuning("gcn_xcd_tile", xcd)
                    _C.set_tuning("gcn_block", blk)
                    _C.set_tuning("gcn_tile", tl)
                    _C.set_tuning("gcn_tile_rows", tr)
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
                    print(json.dumps({"F": F, "edge_cache": window, "node_ids": nid, "unroll": unroll, "xcd_tile": xcd, "block": blk, "tile": tl, "tile_rows": tr, "ms": round(med, 4),
                                      "GBps_alg": round(nbytes / med / 1e6, 1), "frac": round(nbytes / med / 8e9, 3)}),
                          flush=True)
    kernels.set_edge_cache(True)
    _C.set_tuning("gcn_unroll", 0)
    _C.set_tuning("gcn_xcd_tile", 0)
    _C.set_tuning("gcn_block", 0)
    _C.set_tuning("gcn_tile", 0)
    _C.set_tuning("gcn_tile_rows", 0)


if __name__ == "__main__":
    main()
