"""cfg3's captured model epoch alone (uniform form), for a kernel trace: python tools/diag/gat_only.py [epochs]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
import bench
from stgraph_amd.capture import CapturedTrainStep
from stgraph_amd.graph import StaticGraph
from stgraph_amd.nn import functional as SF

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n, e, fin, H, D, classes = 256_000, 8_000_000, 64, 8, 64, 16
src, dst = bench.synthetic_graph(n, e, 2, dev)
g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
gen = torch.Generator(device=dev).manual_seed(2)
feats = torch.randn(n, fin, device=dev, generator=gen)
labels = torch.randint(0, classes, (n,), device=dev, generator=gen)
ntrain = int(0.6 * n)
torch.manual_seed(2)
model = bench.GAT(g, 1, fin, D, classes, [H, 1], F.elu).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True, fused=True)


def step():
    model.train()
    logits = model(feats)
    loss = SF.cross_entropy(logits, labels, ntrain)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach()


run = CapturedTrainStep(step, opt, list(model.parameters()))
dur = []
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    torch.cuda.synchronize()
    time.sleep(0.002)                    # a gap the trace reader (cora_kernels.py) splits the replays at
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dur.append(time.perf_counter() - t0)
print(json.dumps({"ms_per_epoch": 1e3 * float(np.mean(dur[3:]))}))
