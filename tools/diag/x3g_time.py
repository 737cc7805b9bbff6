"""Device time of the split-form contractions only (no error check: for the diagnosis builds of tools/diag/build_x3g_ablate.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stgraph_amd import _C, kernels
from tools.microbench_gemm_x3 import med
dev = torch.device("cuda", 0)
_C.set_tuning("gemm_x3", 2)
A, B = torch.randn(1_000_000, 128, device=dev), torch.randn(1_000_000, 128, device=dev)
res = {"lib": os.path.basename(_C.LIB_PATH), "cfg2_128x128_us": med(lambda: kernels.gemm_tn(A, B))}
del A, B
Kw, T, C, Fin = 50_000, 25, 64, 32
D3 = [torch.randn(Kw, 3 * C, device=dev) for _ in range(T)]
H, P = [torch.randn(Kw, C, device=dev) for _ in range(T)], [torch.randn(Kw, Fin, device=dev) for _ in range(T)]
res["zr_128x96_us"] = med(lambda: kernels.gemm_tn_form(As=[d[:, :2 * C] for d in D3], Bs=H, B2s=P, M=2 * C, N=C + Fin, nsplit=C))
res["h_64x96_us"] = med(lambda: kernels.gemm_tn_form(As=[d[:, 2 * C:] for d in D3], Bs=H, B2s=P, M=C, N=C + Fin, nsplit=C))
print(json.dumps(res))
