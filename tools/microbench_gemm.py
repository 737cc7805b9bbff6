#!/usr/bin/env python3
"""stg_gemm_tn_f32 vs torch (rocBLAS/hipBLASLt) on the weight-gradient shapes of the BASELINE configs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import kernels


def t_ms(fn, iters=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    dev = torch.device("cuda", 0)
    for K, M, N in ((50_000, 32, 192), (50_000, 64, 128), (50_000, 32, 64), (50_000, 1, 32), (25_000, 64, 128),
                    (1_000_000, 128, 128), (2708, 1433, 16), (256_000, 64, 512)):
        a = torch.randn(K, M, device=dev)
        b = torch.randn(K, N, device=dev)
        mine = t_ms(lambda: kernels.gemm_tn(a, b))
        ref = t_ms(lambda: torch.mm(a.t(), b))
        nbytes = 4 * K * (M + N)
        print(json.dumps({"K": K, "M": M, "N": N, "stg_ms": round(mine, 4), "torch_ms": round(ref, 4),
                          "speedup": round(ref / mine, 2), "stg_GBps": round(nbytes / mine / 1e6, 1),
                          "stg_TFLOPs": round(2 * K * M * N / mine / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
