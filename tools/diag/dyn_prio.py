"""cfg5 rebuild mode, builds one window ahead on a second stream (CapturedDynamicWindows.prefetch_builds), with the build stream at the lowest / the default priority: python tools/diag/dyn_prio.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from stgraph_amd import temporal
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print("priority_range", torch.cuda.Stream.priority_range())
orig = temporal.CapturedDynamicWindows.__init__
for low in (True, False, True, False):
    def init(self, *a, _low=low, **k):
        orig(self, *a, **k)
        self.build_stream_low_priority = _low
        self.prefetch_builds = True
    temporal.CapturedDynamicWindows.__init__ = init
    out = bench.dynamic_run(dev, 0, 1, epochs=40, T=40, only_modes=["rebuild_per_snapshot"])
    print(json.dumps({"low_priority": low, "epochs_per_s": out["rebuild_per_snapshot"]["epochs_per_s"]}))
