"""cfg4 alone (for a kernel trace): python tools/diag/tgcn_only.py [epochs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = bench.tgcn_run(dev, 0, 1, epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 2, warmup_epochs=3, n=50_000, e=500_000, T=200, feat=32,
                     hidden=64, B=25, allreduce_in_graph=False, cpu_baseline=False, share_device=False)
print(json.dumps({"value": out["value"], "us_per_snapshot": out.get("us_per_snapshot")}))
