"""cfg5 (dynamic-temporal TGCN, T = 40) in ONE mode, for profiling: python tools/diag/dyn_only.py [rebuild_per_snapshot|resident_snapshots|pcsr_store|gpma_store] [epochs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "rebuild_per_snapshot"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = bench.dynamic_run(dev, 0, 1, epochs=epochs, T=40, only_modes=[mode])
print(json.dumps({mode: out[mode]}))
