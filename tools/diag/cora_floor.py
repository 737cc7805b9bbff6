"""Where the wall time of a captured Cora epoch goes: replay + sync of the epoch graph, the same replays back to back (one sync at the
end), and the floor -- a captured graph of ONE empty kernel, replay + sync.  python tools/diag/cora_floor.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
import bench
from stgraph_amd import kernels
from stgraph_amd.capture import CapturedTrainStep
from stgraph_amd.graph import StaticGraph
from stgraph_amd.nn import functional as SF

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
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


out = {}
for on in (False, True, False, True):
    kernels.set_xent_small(on)
    r2 = CapturedTrainStep(step, opt, list(model.parameters()))
    for _ in range(20):
        r2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        r2.graph.replay()
    torch.cuda.synchronize()
    out.setdefault("back_to_back_us_xent_small_%s" % on, []).append(1e6 * (time.perf_counter() - t0) / 500)
    del r2
kernels.set_xent_small(True)
run = CapturedTrainStep(step, opt, list(model.parameters()))
for _ in range(20):
    run()
torch.cuda.synchronize()
d = []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    d.append(time.perf_counter() - t0)
out["epoch_replay_plus_sync_us"] = 1e6 * float(np.mean(d))
out["epoch_replay_plus_sync_us_median"] = 1e6 * float(np.median(d))
d = []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    d.append((t1 - t0, time.perf_counter() - t1))
out["replay_call_us"] = 1e6 * float(np.mean([a for a, _ in d]))
out["sync_after_us"] = 1e6 * float(np.mean([b for _, b in d]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    run.graph.replay()
torch.cuda.synchronize()
out["epoch_back_to_back_us"] = 1e6 * (time.perf_counter() - t0) / 300
# the floor: one empty kernel
z = torch.zeros(1, device=dev)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    z.add_(1.0)
for _ in range(20):
    gr.replay()
d = []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gr.replay()
    torch.cuda.synchronize()
    d.append(time.perf_counter() - t0)
out["one_kernel_graph_replay_plus_sync_us"] = 1e6 * float(np.mean(d))
d = []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z.add_(1.0)
    torch.cuda.synchronize()
    d.append(time.perf_counter() - t0)
out["one_kernel_eager_plus_sync_us"] = 1e6 * float(np.mean(d))
print(json.dumps(out))
