"""cfg1 captured epoch with / without the small-matrix launches (same process, alternating): python tools/diag/cora_ab.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from stgraph_amd import kernels
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bench.cora_roofline = lambda *a, **k: {"frac": 0.0}
for on in (True, False, True, False):
    kernels.set_xent_small(on)
    d = bench.cora_run(dev, epochs=300)
    print(json.dumps({"xent_small": on, "hip_graph_epochs_per_s": d["hip_graph"]["epochs_per_s"], "us": 1e3 * d["hip_graph"]["ms_per_epoch"],
                      "eager_us": 1e3 * d["eager"]["ms_per_epoch"]}), flush=True)
