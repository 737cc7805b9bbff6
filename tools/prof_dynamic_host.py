#!/usr/bin/env python3
"""cProfile of one dynamic-temporal training epoch (BASELINE configs[4] shape, per-snapshot rebuild): where the host time of the eager loop goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from stgraph_amd import temporal
from stgraph_amd.graph import NaiveGraph
dev = torch.device("cuda", 0)
n, e0, churn, T, B, feat, hidden = 25_000, 250_000, 6_250, 160, 20, 32, 64
rng = np.random.default_rng(4)
stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
snaps, pn_edges, pn_targets = [], [], []
gen = torch.Generator(device=dev).manual_seed(4)
m = 10_000
for t in range(T):
    keys = stream[t * churn: t * churn + e0]
    s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
    snaps.append((torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)))
    pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(dev)
    neg = torch.randint(0, n, (2, m), device=dev, generator=gen)
    pn_edges.append(torch.cat([pos, neg], 1))
    pn_targets.append(torch.cat([torch.ones(m, device=dev), torch.zeros(m, device=dev)]))
G = NaiveGraph(snaps, n, device=dev, sort_inplace=False, resident=False, max_cached=B + 1)
torch.manual_seed(4)
model = temporal.DynamicSTGraphTGCN(feat, hidden).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-2)
bucket = temporal.GradBucket(model.parameters())
def epoch(ep):
    G._snapshots.clear(); G._ndata.clear()
    temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep)
epoch(0); epoch(1); torch.cuda.synchronize()
import time
t0=time.perf_counter(); epoch(2); torch.cuda.synchronize(); print("epoch s", time.perf_counter()-t0)
pr = cProfile.Profile(); pr.enable(); epoch(3); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
