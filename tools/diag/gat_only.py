"""GAT configuration of bench.py alone (configs[2]): prints its object.  python tools/diag/gat_only.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print(json.dumps(bench.gat_run(dev, cpu_baseline=False)))
