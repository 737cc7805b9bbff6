import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tools.bench_powerlaw import chung_lu
from stgraph_amd import kernels
from stgraph_amd.graph import StaticGraph
dev = torch.device("cuda", 0)
n, e, F = 1_000_000, 16_000_000, 128
src, dst = chung_lu(n, e, 0.75, 5, dev)
g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
norm = torch.rand(n, 1, device=dev) + 0.5
x = torch.randn(n, F, device=dev)
for _ in range(5):
    kernels.gcn_agg(x, norm, norm, g.csr("fwd"))
torch.cuda.synchronize()
deg = (g.csr("fwd").row_offset[1:] - g.csr("fwd").row_offset[:-1])
for lo, hi in ((1024, 4096), (4096, 16384), (16384, 10**9)):
    m = (deg > lo) & (deg <= hi)
    print(lo, hi, "rows", int(m.sum()), "edges", int(deg[m].sum()))
print("rows 256..1024", int(((deg > 256) & (deg <= 1024)).sum()), "edges", int(deg[(deg > 256) & (deg <= 1024)].sum()))
