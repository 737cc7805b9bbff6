import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from stgraph_amd import kernels
device = torch.device("cuda", 0)
src, dst = bench.cora_shaped()
n, K = 2708, 1024
N, E = n * K, len(src) * K
off = (torch.arange(K, device=device, dtype=torch.int64) * n).repeat_interleave(len(src))
s = torch.from_numpy(src).to(device).long().repeat(K) + off
d = torch.from_numpy(dst).to(device).long().repeat(K) + off
g = kernels.build_graph_csr(s.int(), d.int(), N, device)
norm = torch.rand(N, 1, device=device) + 0.5
def t(fn, iters=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for F_, ld in ((7, 7), (7, 8), (5, 5), (5, 8), (3, 3), (3, 4), (6, 6), (6, 8), (12, 12), (12, 16), (16, 16)):
    x = torch.randn(N, ld, device=device)
    nbytes = kernels.gcn_agg_algorithmic_bytes(N, E, F_, False)
    ms = [t(lambda: kernels.gcn_agg(x, norm, norm, csr, f_active=F_)) for csr in (g.fwd, g.bwd)]
    ref = kernels.gcn_agg(x[:, :F_].contiguous(), norm, norm, g.fwd)
    got = kernels.gcn_agg(x, norm, norm, g.fwd, f_active=F_)[:, :F_]
    print(f"F={F_} ld={ld}: fwd {ms[0]:.3f} bwd {ms[1]:.3f} ms  frac {2*nbytes/(sum(ms)*1e-3)/1e9/8000:.3f}  bit-equal {bool(torch.equal(ref, got))}", flush=True)
