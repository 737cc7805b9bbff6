"""cfg1 alone (for profiling): python tools/diag/cora_only.py [epochs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bench.cora_roofline = lambda *a, **k: {"frac": 0.0}
d = bench.cora_run(dev, epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 60)
print(json.dumps({k: d[k] for k in ("eager", "hip_graph")}))
