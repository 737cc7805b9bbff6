"""The dense products of cfg3's second GAT layer (GATConv(512, 16, 1 head) at |V| = 256 K) on the library GEMM and on the kernels here:
fc forward [N, 512] x [512, 16] (torch 101 us alone, 145 in the epoch), its input gradient [N, 16] x [16, 512] (119 us; floor 85: the 524 MB
it writes), the weight gradient (torch.mm 377 us, stg_gemm_tn_f32 114).  Neither row-product kernel covers K = 512 -> 16 / 16 -> 512.
python tools/diag/mb_l2fc.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stgraph_amd import _C, kernels
dev = torch.device("cuda", 0)
N = 256_000
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
x = torch.randn(N, 512, device=dev); W = torch.randn(16, 512, device=dev) / 20; g = torch.randn(N, 16, device=dev)
res = {"supported_512_16": bool(_C.lib.stg_rowgemm_supported(512, 16)), "supported_16_512": bool(_C.lib.stg_rowgemm_supported(16, 512))}
res["torch_fwd_us"] = timed(lambda: torch.mm(x, W.t()))
res["torch_bwd_gx_us"] = timed(lambda: torch.mm(g, W))
res["torch_bwd_gw_us"] = timed(lambda: torch.mm(g.t(), x))
if res["supported_512_16"]:
    y = kernels.rowgemm(x, W, None, trans_w=True)
    res["rowgemm_fwd_us"] = timed(lambda: kernels.rowgemm(x, W, None, trans_w=True))
    res["fwd_err"] = float((y - x @ W.t()).abs().max())
if res["supported_16_512"]:
    y = kernels.rowgemm(g, W, None, trans_w=False)
    res["rowgemm_bwd_gx_us"] = timed(lambda: kernels.rowgemm(g, W, None, trans_w=False))
    res["gx_err"] = float((y - g @ W).abs().max())
res["gemm_tn_gw_us"] = timed(lambda: kernels.gemm_tn(g, x))
print(json.dumps(res))
