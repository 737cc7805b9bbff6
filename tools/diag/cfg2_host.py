"""cfg2 step: host issue time against device time (is the eager step launch-bound on this host?)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from stgraph_amd import kernels  # noqa: E402

dev = torch.device("cuda", 0)
step, meta = bench.gcn_setup(dev, seed=1, n=1_000_000, e=16_000_000, feat=128)
for _ in range(5):
    step()
torch.cuda.synchronize()
out = {}
for name, rec in (("plain", None), ("launch_timing", [])):
    kernels.enable_launch_timing(rec)
    host, wall = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0)
        wall.append(t2 - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    back = (time.perf_counter() - t0) / 20
    kernels.enable_launch_timing(None)
    out[name] = {"host_issue_ms_mean": 1e3 * sum(host) / len(host), "host_issue_ms_min": 1e3 * min(host),
                 "wall_ms_mean_synced_each": 1e3 * sum(wall) / len(wall), "back_to_back_ms": back * 1e3}
from stgraph_amd.capture import CapturedTrainStep  # noqa: E402
print(json.dumps(out))
