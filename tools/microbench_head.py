#!/usr/bin/env python3
"""Fused TGCN head forward / backward: 32-row tiles vs 16-row tiles ("cell_rows"), same process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import _C, kernels


def main():
    dev = torch.device("cuda", 0)
    for N, C in ((50_000, 64), (25_000, 64), (10_000, 64), (400_000, 64)):
        g = torch.Generator(device=dev).manual_seed(1)
        r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
        h, W1, b1, W2, b2, t, gy = r(N, C), r(32, C) * 0.3, r(32), r(1, 32) * 0.3, r(1), r(N, 1), r(N, 32)
        gl = torch.ones(1, device=dev)
        for rows in (32, 16, 32, 16):
            _C.set_tuning("cell_rows", rows)
            for _ in range(3):
                rr, y, yo, loss = kernels.tgcn_head_fwd(h, W1, b1, W2, b2, t)
                kernels.tgcn_head_bwd(gl, gy, None, h, yo, t, W1, W2)
            tf, tb = [], []
            for _ in range(20):
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                a.record()
                kernels.tgcn_head_fwd(h, W1, b1, W2, b2, t)
                b.record()
                kernels.tgcn_head_bwd(gl, gy, None, h, yo, t, W1, W2)
                c.record()
                torch.cuda.synchronize()
                tf.append(a.elapsed_time(b))
                tb.append(b.elapsed_time(c))
            print(json.dumps({"N": N, "C": C, "tile_rows": rows, "fwd_us": round(float(np.median(tf)) * 1e3, 1),
                              "bwd_us": round(float(np.median(tb)) * 1e3, 1)}), flush=True)
    _C.set_tuning("cell_rows", 0)


if __name__ == "__main__":
    main()
