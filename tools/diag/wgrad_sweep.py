import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from stgraph_amd import kernels, _C
dev = torch.device("cuda", 0)
K, C, T = 50_000, 64, 25
d = [torch.randn(K, C, device=dev) for _ in range(T)]
x3 = [torch.randn(K, 3 * C, device=dev) for _ in range(T)]
H = [torch.randn(K, C, device=dev) for _ in range(T)]
xs = [x[:, :C] for x in x3]
def med(fn, iters=12):
    for _ in range(3): fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3
fn = lambda: kernels.gemm_tn_form(d, xs, C, 2 * C, B2s=H, nsplit=C, b_op=kernels.GEMM_B_CLAMP, lo=-1e6, hi=1e6, colsum=True)
ref = None
for rep in range(2):
  for wide, cyc in ((1, 0), (0, 0), (0, 1), (0, 2)):
    _C.set_tuning("gemm_wide", wide); _C.set_tuning("gemm_cyclic", cyc)
    out = fn()
    if ref is None: ref = out
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref))
    print("narrow" if wide else "wide", "cyclic%d" % cyc, "us", round(med(fn), 1), "relerr vs narrow", err, flush=True)
