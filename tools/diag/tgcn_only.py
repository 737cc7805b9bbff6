"""TGCN (cfg4) and dynamic (cfg5, T = 40) objects of bench.py alone.  python tools/diag/tgcn_only.py [--from-p 0|1] [--folded 0|1]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from stgraph_amd import kernels
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = dict(zip(sys.argv[1::2], sys.argv[2::2]))
kernels.set_step_wgrad_from_p(bool(int(args.get("--from-p", 1))))
kernels.set_step_folded(bool(int(args.get("--folded", 1))))
if "--waves" in args:
    from stgraph_amd import _C
    _C.set_tuning("step_waves", int(args["--waves"]))
t = bench.tgcn_run(dev, 0, 1, epochs=8, warmup_epochs=3, n=50_000, e=500_000, T=1000, feat=32, hidden=64, B=25, cpu_baseline=False)
d = bench.dynamic_run(dev, 0, 1, epochs=8, cpu_baseline=False)
kernels.check_step_fold_status(dev)
print(json.dumps({"from_p": kernels.STEP_WGRAD_FROM_P, "folded": kernels.STEP_FOLDED, "tgcn_epochs_per_s": t["value"],
                  "us_per_snapshot": t["roofline"]["seconds_per_snapshot"] * 1e6,
                  "dynamic": {k: (v.get("epochs_per_s") if isinstance(v, dict) else v) for k, v in d.items()
                              if k in ("value", "resident_snapshots", "rebuild_per_snapshot", "pcsr_store", "gpma_store")}}))
