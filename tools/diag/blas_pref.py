#!/usr/bin/env python3
"""[1M,128] x [128,128] fp32 on torch's two BLAS back ends (rocBLAS / hipBLASLt): which one cfg2's three dense GEMMs should use."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.microbench_gemm import t_ms
dev = torch.device("cuda", 0)
x = torch.randn(1_000_000, 128, device=dev); w = torch.randn(128, 128, device=dev); b = torch.randn(128, device=dev)
res = {}
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        res[lib] = {"mm": round(t_ms(lambda: torch.mm(x, w)), 4), "mm_t": round(t_ms(lambda: torch.mm(x, w.t())), 4),
                    "addmm_t": round(t_ms(lambda: torch.addmm(b, x, w.t())), 4)}
    except Exception as e:  # noqa: BLE001
        res[lib] = repr(e)
print(json.dumps(res))
