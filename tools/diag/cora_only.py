"""Cora configuration of bench.py alone (configs[0]): prints its object.  python tools/diag/cora_only.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
d = bench.cora_run(dev, cpu_baseline=False)
print(json.dumps({"eager": d["eager"], "hip_graph": d["hip_graph"]}))
