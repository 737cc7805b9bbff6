"""RCCL on the one GPU a test box has.  Two ranks cannot share a device (RCCL 2.26: "Duplicate GPU detected", measured
with tools/diag/rccl_one_gpu.py), so what CAN be exercised on hardware is a ONE-rank ``nccl`` process group: communicator
creation with ``device_id=``, the flat-bucket all-reduce, and -- the part that matters for ``allreduce_in_graph`` -- the
collective captured INTO the optimizer-tail HIP graph and replayed.  Runs in a child process (its own default group)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["STG_ROOT"])
import numpy as np, torch, torch.distributed as dist
from stgraph_amd import temporal
from stgraph_amd.graph import StaticGraph
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, e, feat, hid, T, B = 4096, 40000, 32, 64, 8, 4
rng = np.random.default_rng(0)
keys = rng.choice(n * n, size=e, replace=False)
g = StaticGraph(((keys // n).astype(np.int32), (keys % n).astype(np.int32)), None, n, device=dev, sort_inplace=False)
deg = torch.bincount(torch.from_numpy((keys % n)).to(dev), minlength=n).float()
norm = deg.pow(-0.5); norm[torch.isinf(norm)] = 0
g.set_ndata("norm", norm.unsqueeze(1))
targets = torch.randn(T, n, 1, device=dev)
res = []
for in_graph in (True, False):
    torch.manual_seed(1)
    model = temporal.STGraphTGCN(feat, hid, 1).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
    bucket = temporal.GradBucket(model.parameters())
    cw = temporal.CapturedStaticWindow(model, g, None, targets, B, opt, bucket, feat, world=1, rank=0,
                                       group=dist.group.WORLD if in_graph else None, allreduce_in_graph=in_graph)
    assert cw.allreduce_in_graph == in_graph, "RCCL refused to capture the all-reduce"
    costs = []
    for ep in range(3):
        costs += temporal.train_epoch_static_captured(cw, model, g, None, targets, opt, bucket, feat, epoch=ep,
                                                      group=dist.group.WORLD if in_graph else None)
    res.append((torch.stack(costs), [p.detach().clone() for p in model.parameters()]))
torch.testing.assert_close(res[0][0], res[1][0], rtol=0, atol=0)
for a, b in zip(res[0][1], res[1][1]):
    torch.testing.assert_close(a, b, rtol=0, atol=0)
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t)
assert torch.equal(t, torch.arange(8, device=dev, dtype=torch.float32))
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


def test_rccl_communicator_and_captured_allreduce_on_one_rank(cuda):
    env = dict(os.environ, STG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


CHILD_DYNAMIC = r'''
import os, sys
sys.path.insert(0, os.environ["STG_ROOT"])
import numpy as np, torch, torch.distributed as dist
from stgraph_amd import temporal
from stgraph_amd.graph import NaiveGraph
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, e0, churn, T, B, m, feat, hid = 3000, 25000, 600, 9, 4, 1000, 32, 64
rng = np.random.default_rng(11)
stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
snaps, pn_edges, pn_targets = [], [], []
gen = torch.Generator(device=dev).manual_seed(4)
for t in range(T):
    keys = stream[t * churn: t * churn + e0]
    s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
    snaps.append((torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)))
    pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(dev)
    neg = torch.randint(0, n, (2, m), device=dev, generator=gen)
    pn_edges.append(torch.cat([pos, neg], 1))
    pn_targets.append(torch.cat([torch.ones(m, device=dev), torch.zeros(m, device=dev)]))
res = []
for in_graph in (True, False):
    G = NaiveGraph(snaps, n, device=dev, sort_inplace=False, resident=False, max_cached=B + 1)
    torch.manual_seed(1)
    model = temporal.DynamicSTGraphTGCN(feat, hid).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
    bucket = temporal.GradBucket(model.parameters())
    group = dist.group.WORLD if in_graph else None
    costs = [c.clone() for c in temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=0, group=group)]
    cd = temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat, world=1, rank=0, group=group,
                                         allreduce_in_graph=in_graph)
    for ep in range(1, 4):
        G._snapshots.clear()
        G._ndata.clear()
        costs += [c.clone() for c in temporal.train_epoch_dynamic_captured(cd, epoch=ep)]
    assert cd.step_graph is not None and cd.allreduce_in_graph == in_graph, "RCCL refused to capture the all-reduce"
    res.append((torch.stack(costs), [p.detach().clone() for p in model.parameters()]))
torch.testing.assert_close(res[0][0], res[1][0], rtol=0, atol=0)
for a, b in zip(res[0][1], res[1][1]):
    torch.testing.assert_close(a, b, rtol=0, atol=0)
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_DYNAMIC_OK")
'''


def test_rccl_captured_allreduce_in_the_dynamic_windows_tail_on_one_rank(cuda):
    """CapturedDynamicWindows(allreduce_in_graph=True) with a one-rank RCCL group: the collective is captured into the
    optimizer-tail graph and the run equals the eager-collective one bit for bit (the static twin is the test above)."""
    env = dict(os.environ, STG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD_DYNAMIC], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_DYNAMIC_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
