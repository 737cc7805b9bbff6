import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util import golden
from tests.test_gpu_models_ref import _static_setup, _load, _x0
from stgraph_amd import temporal
cuda = torch.device("cuda", 0)
d = golden("tgcn_native.npz")
for use_ew in (False, True):
  g, targets, ew, n, T = _static_setup(d, cuda, use_ew)
  feat, hid, B = int(d["feat"]), int(d["hidden"]), 3
  tag = f"{'ew' if use_ew else 'now'}_B{B}"
  for fused in (True, False):
    temporal.set_fused_window(fused)
    model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
    _load(model, d, f"{tag}_param_", cuda)
    for index in range(2):
        model.zero_grad()
        x0 = _x0(int(d["x0_seed_base"]) + index, n, feat, cuda)
        cost = temporal.window_cost_of(model, g, x0, ew, targets[index * B:(index + 1) * B]) / (B + 1)
        cost.backward()
        errs = {k: float(np.abs(p.grad.cpu().numpy() - d[f"{tag}_w{index}_grad_{k}"]).max() / np.abs(d[f"{tag}_w{index}_grad_{k}"]).max()) for k, p in model.named_parameters()}
        print(tag, "fused" if fused else "unfused", index, float(cost), d[f"{tag}_cost"][index], {k.replace("temporal.", ""): f"{v:.1e}" for k, v in errs.items() if v > 2e-5})
