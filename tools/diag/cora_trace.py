#!/usr/bin/env python3
"""One captured Cora epoch (bench.cora_run's step) replayed 200 times: run under `rocprofv3 --kernel-trace --output-format csv`
and print the kernels of one replay in order with their durations and the gaps between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import bench
from stgraph_amd.capture import CapturedTrainStep
from stgraph_amd.graph import StaticGraph
from stgraph_amd.nn import functional as SF
dev = torch.device("cuda", 0)
src, dst = bench.cora_shaped()
n = 2708
g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
g.set_ndata("norm", bench.degree_norm(g))
gen = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(n, 1433, device=dev, generator=gen) < 0.0127).float()
labels = torch.randint(0, 7, (n,), device=dev, generator=gen)
ntrain = int(0.6 * n)
torch.manual_seed(0)
model = bench.GCN(1433, 16, 7, 1, F.relu).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4, capturable=True, fused=True)
one = torch.ones((), device=dev)
def step():
    logits = model(g, x)
    loss = SF.cross_entropy(logits, labels, ntrain)
    opt.zero_grad()
    loss.backward(one)
    opt.step()
    return loss.detach()
run = CapturedTrainStep(step, opt, list(model.parameters()))
for _ in range(200):
    run()
torch.cuda.synchronize()
