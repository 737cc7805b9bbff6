import sys, os
sys.path.insert(0, os.getcwd())
import torch
from stgraph_amd import kernels, _C
dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "narrow":
    _C.set_tuning("gemm_wide", 1)
K, C, T = 50_000, 64, 25
d = [torch.randn(K, C, device=dev) for _ in range(T)]
x3 = [torch.randn(K, 3 * C, device=dev) for _ in range(T)]
H = [torch.randn(K, C, device=dev) for _ in range(T)]
xs = [x[:, :C] for x in x3]
for _ in range(6):
    kernels.gemm_tn_form(d, xs, C, 2 * C, B2s=H, nsplit=C, b_op=kernels.GEMM_B_CLAMP, lo=-1e6, hi=1e6, colsum=True)
torch.cuda.synchronize()
