#!/usr/bin/env python3
"""Per-snapshot CSR rebuild (stg_graph_build_direct2_device) at the cfg5 snapshot size with the histogram pass in LDS / in global atomics."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synthetic_graph
from stgraph_amd import _C, kernels
dev = torch.device("cuda", 0)
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000, int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
src, dst = synthetic_graph(n, e, 5, dev)
kernels.build_graph_csr(src, dst, n, dev, lazy_node_ids=True)
res = {"N": n, "E": e}
for mode, name in ((2, "global_atomics"), (1, "lds_histograms")):
    _C.set_tuning("build_lds_count", mode)
    for _ in range(5):
        kernels.build_graph_csr(src, dst, n, dev, lazy_node_ids=True, known_path="direct")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            kernels.build_graph_csr(src, dst, n, dev, lazy_node_ids=True, known_path="direct")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        g.replay()
    b.record(); torch.cuda.synchronize()
    res[name + "_us_per_build"] = round(a.elapsed_time(b) * 1e3 / 200, 2)
print(json.dumps(res))
