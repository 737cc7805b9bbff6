"""cfg2's training step alone (eager, as bench.py times it), for a kernel trace: python tools/diag/cfg2_only.py [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
step, meta = bench.gcn_setup(dev, seed=1, n=1_000_000, e=16_000_000, feat=128)
for _ in range(3):
    step()
torch.cuda.synchronize()
dur = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.cuda.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dur.append(time.perf_counter() - t0)
print(json.dumps({"ms_per_step": 1e3 * sum(dur[2:]) / len(dur[2:])}))
