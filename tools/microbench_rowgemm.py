#!/usr/bin/env python3
"""stg_rowgemm_f32 vs torch (rocBLAS/hipBLASLt) on the skinny forward / input-gradient GEMM shapes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stgraph_amd import kernels
from tools.microbench_gemm import t_ms


def main():
    dev = torch.device("cuda", 0)
    only = int(sys.argv[1]) if len(sys.argv) > 1 else None      # one shape per run: per-shape device times under rocprofv3
    for idx, (N, K, M, tw) in enumerate(((50_000, 128, 64, True), (50_000, 64, 128, False), (50_000, 192, 32, True),
                        (50_000, 64, 32, True), (25_000, 128, 64, True), (1_000_000, 64, 128, False),
                                            (1_000_000, 128, 128, False), (1_000_000, 128, 128, True), (256_000, 64, 128, True))):
        if only is not None and idx != only:
            continue
        x = torch.randn(N, K, device=dev)
        w = torch.randn((M, K) if tw else (K, M), device=dev)
        b = torch.randn(M, device=dev)
        mine = t_ms(lambda: kernels.rowgemm(x, w, b if tw else None, trans_w=tw))
        ref = t_ms((lambda: torch.addmm(b, x, w.t())) if tw else (lambda: torch.mm(x, w)))
        nbytes = 4 * N * (K + M)
        print(json.dumps({"N": N, "K": K, "M": M, "trans_w": tw, "stg_ms": round(mine, 4), "torch_ms": round(ref, 4),
                          "speedup": round(ref / mine, 2), "stg_GBps": round(nbytes / mine / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
