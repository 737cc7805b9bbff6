"""One mode of bench.dynamic_run (argv[1]: rebuild_per_snapshot | pcsr_store | resident_snapshots), T = 40, for rocprofv3."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, time
import bench
from stgraph_amd import temporal
from stgraph_amd.graph import GPMAGraph, NaiveGraph, PCSRGraph
mode = sys.argv[1]
device = torch.device("cuda", 0)
n, e0, churn, T, B, feat, hidden = 25_000, 250_000, 6_250, 40, 20, 32, 64
rng = np.random.default_rng(4)
stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
snaps, pn_edges, pn_targets = [], [], []
gen = torch.Generator(device=device).manual_seed(4)
m = 10_000
for t in range(T):
    keys = stream[t * churn: t * churn + e0]
    s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
    snaps.append((torch.from_numpy(s).to(device), torch.from_numpy(d).to(device)))
    pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(device)
    neg = torch.randint(0, n, (2, m), device=device, generator=gen)
    pn_edges.append(torch.cat([pos, neg], 1))
    pn_targets.append(torch.cat([torch.ones(m, device=device), torch.zeros(m, device=device)]))
if mode == "resident_snapshots":
    G = NaiveGraph(snaps, n, device=device, sort_inplace=False)
elif mode == "rebuild_per_snapshot":
    G = NaiveGraph(snaps, n, device=device, sort_inplace=False, resident=False, max_cached=B + 1)
else:
    G = (PCSRGraph if mode == "pcsr_store" else GPMAGraph)(snaps, n, device=device)
torch.manual_seed(4)
model = temporal.DynamicSTGraphTGCN(feat, hidden).to(device)
opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
bucket = temporal.GradBucket(model.parameters())
cd = None
for ep in range(13):
    if mode == "rebuild_per_snapshot":
        G._snapshots.clear()
    G._ndata.clear()
    if ep >= 1:
        cd = cd or temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat)
        temporal.train_epoch_dynamic_captured(cd, epoch=ep)
    else:
        temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
torch.cuda.synchronize()
