#!/usr/bin/env python3
"""gcn_agg_transform (aggregate-then-transform, TGCN cfg4 shape) with 4 / 8 waves per workgroup, same process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_graph
from stgraph_amd import _C, kernels


def main():
    dev = torch.device("cuda", 0)
    for n, e, fin, fout in ((50_000, 500_000, 32, 192), (1_000_000, 16_000_000, 32, 96)):
        src, dst = synthetic_graph(n, e, 3, dev)
        g = kernels.build_graph_csr(src, dst, n, dev)
        x = torch.randn(n, fin, device=dev)
        W = torch.randn(fin, fout, device=dev)
        norm = torch.rand(n, 1, device=dev) + 0.5
        ew = torch.rand(e, 1, device=dev) + 0.5
        ref = None
        for waves, rows in ((4, 64), (4, 32), (8, 32), (4, 64), (4, 32), (8, 32)):
            _C.set_tuning("xw_waves", waves)
            _C.set_tuning("xw_rows", rows)
            for _ in range(3):
                out, P = kernels.gcn_agg_transform(x, W, norm, norm, g.fwd, ew=ew)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref)
            ts = []
            for _ in range(20):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                kernels.gcn_agg_transform(x, W, norm, norm, g.fwd, ew=ew)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print(json.dumps({"N": n, "E": e, "Fin": fin, "Fout": fout, "waves": waves, "rows": rows, "us": round(float(np.median(ts)) * 1e3, 1)}),
                  flush=True)
    _C.set_tuning("xw_waves", 0)
    _C.set_tuning("xw_rows", 0)


if __name__ == "__main__":
    main()
