#!/usr/bin/env python3
"""Fused TGCN forward cell: 32-row tiles (v_mfma_f32_32x32x2_f32, 8-wave workgroups) vs 16-row tiles
(v_mfma_f32_16x16x4_f32, 16-wave workgroups), same process, cfg4 shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import _C
from stgraph_amd.nn.pytorch.temporal import cell


def main():
    dev = torch.device("cuda", 0)
    for N, C in ((50_000, 64), (25_000, 64), (50_000, 32), (400_000, 64)):
        g = torch.Generator(device=dev).manual_seed(1)
        r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
        a3, b3, H = r(N, 3 * C), r(3 * C), r(N, C)
        Ws = [r(C, 2 * C) * 0.2 for _ in range(3)]
        bs = [r(C) for _ in range(3)]
        dHn = r(N, C)
        for rows in (32, 16, 32, 16):
            _C.set_tuning("cell_rows", rows)
            for _ in range(3):
                Hn, (CZ, CR, CH, Z, R, Ht) = cell._cell_forward(a3, b3, H, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])
                cell._cell_backward(dHn, a3, b3, H, Ws[0], Ws[1], Ws[2], CZ, CR, CH, Z, R, Ht)
            tf, tb = [], []
            for _ in range(20):
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                a.record()
                cell._cell_forward(a3, b3, H, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])
                b.record()
                cell._cell_backward(dHn, a3, b3, H, Ws[0], Ws[1], Ws[2], CZ, CR, CH, Z, R, Ht)
                c.record()
                torch.cuda.synchronize()
                tf.append(a.elapsed_time(b))
                tb.append(b.elapsed_time(c))
            print(json.dumps({"N": N, "C": C, "tile_rows": rows, "fwd_us": round(float(np.median(tf)) * 1e3, 1),
                              "bwd_us": round(float(np.median(tb)) * 1e3, 1)}), flush=True)
    _C.set_tuning("cell_rows", 0)


if __name__ == "__main__":
    main()
