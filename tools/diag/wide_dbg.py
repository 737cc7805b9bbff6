import sys, os
sys.path.insert(0, os.getcwd())
import torch
from stgraph_amd import kernels
dev = torch.device("cuda", 0)
K, M, N = 4096, 64, 128
for (k0, m0) in ((0, 0), (0, 5), (1, 5), (3, 17), (70, 63), (4095, 1)):
    a = torch.zeros(K, M, device=dev); b = torch.zeros(K, N, device=dev)
    a[k0, m0] = 1.0
    b[k0] = torch.arange(1, N + 1, device=dev).float()
    c = kernels.gemm_tn(a, b)
    nz = c.nonzero()
    print((k0, m0), "nonzero rows", sorted(set(nz[:, 0].tolist())), "count", len(nz), "row vals", c[nz[0, 0]][:8].tolist() if len(nz) else None)
