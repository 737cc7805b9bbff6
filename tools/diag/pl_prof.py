import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tools.bench_powerlaw import chung_lu
from stgraph_amd import kernels, _C
from stgraph_amd.graph import StaticGraph
dev = torch.device("cuda", 0)
n, e, F = 1_000_000, 16_000_000, 128
src, dst = chung_lu(n, e, 0.75, 5, dev)
g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
norm = torch.rand(n, 1, device=dev) + 0.5
x = torch.randn(n, F, device=dev)
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_C.set_tuning("gcn_long_threshold", thr)
if len(sys.argv) > 2: _C.set_tuning("gcn_unroll", int(sys.argv[2]))
for _ in range(5):
    kernels.gcn_agg(x, norm, norm, g.csr("fwd"))
torch.cuda.synchronize()
