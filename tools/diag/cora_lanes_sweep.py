#!/usr/bin/env python3
"""North-star shape (Cora x 1024, F = 16 and 7): the hand-written aggregation under each lanes-per-row setting, and the GENERATED
kernel of the same vertex function (compiler/codegen.py) beside it.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from bench import cora_shaped
from stgraph_amd import _C, kernels
from stgraph_amd.compiler import dispatch
from stgraph_amd.compiler.backend.pytorch.torch_callback import STGraphBackendTorch
from stgraph_amd.compiler.stgraph import STGraph
from stgraph_amd.graph import StaticGraph

dev = torch.device("cuda", 0)
src, dst = cora_shaped()
n, K = 2708, 1024
big_src = np.concatenate([src + k * n for k in range(K)]).astype(np.int32)
big_dst = np.concatenate([dst + k * n for k in range(K)]).astype(np.int32)
N, E = n * K, len(big_src)
g = kernels.build_graph_csr(big_src, big_dst, N, dev)
norm = torch.rand(N, 1, device=dev) + 0.5


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


res = {}
for F in (16, 7):
    x = torch.randn(N, F, device=dev)
    nbytes = kernels.gcn_agg_algorithmic_bytes(N, E, F, False)
    row = {}
    for lanes in (0, 2, 4, 8, 16):
        try:
            _C.set_tuning("gcn_lanes_per_row", lanes)
            ms = [timed(lambda c=c: kernels.gcn_agg(x, norm, norm, c)) for c in (g.fwd, g.bwd)]
            row[f"lanes{lanes}"] = 2 * nbytes / (sum(ms) * 1e-3) / 8e12
        except Exception as ex:  # noqa: BLE001
            row[f"lanes{lanes}"] = str(ex)[:80]
    _C.set_tuning("gcn_lanes_per_row", 0)
    res[f"F{F}"] = row
# the generated kernel of the GCN vertex function at the same widths (forward launch only: its unit is the same both ways)
G = StaticGraph((big_src, big_dst), None, N, device=dev, sort_inplace=False)


class Mod(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.stgraph = STGraph(STGraphBackendTorch())


dispatch.set_force_generated(True)
for F in (16, 7):
    x = torch.randn(N, F, device=dev)
    mod = Mod()
    fc = mod.stgraph.compile(gnn_module=mod)(lambda v: sum([nb.h * nb.norm for nb in v.innbs]) * v.norm)
    with torch.no_grad():
        ms = timed(lambda: fc(g=G, n_feats={"h": x, "norm": norm}, e_feats={}))
    res[f"F{F}"]["generated_fwd"] = kernels.gcn_agg_algorithmic_bytes(N, E, F, False) / (ms * 1e-3) / 8e12
dispatch.set_force_generated(False)
print(json.dumps(res))
