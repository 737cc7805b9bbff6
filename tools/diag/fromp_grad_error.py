"""Per-parameter gradient agreement of one static TGCN window between the weight-gradient formulations
(kernels.STEP_WGRAD_FROM_P / STEP_FOLDED on and off) and an fp64 torch evaluation of the same window."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from stgraph_amd import kernels, temporal
from stgraph_amd.graph import StaticGraph

dev = torch.device("cuda", 0)
n, e, feat, hid, B = 4096, 40000, 32, 64, 6
rng = np.random.default_rng(0)
keys = rng.choice(n * n, size=e, replace=False)
g = StaticGraph(((keys // n).astype(np.int32), (keys % n).astype(np.int32)), None, n, device=dev, sort_inplace=False)
deg = torch.bincount(torch.from_numpy(keys % n).to(dev), minlength=n).float()
norm = deg.pow(-0.5); norm[torch.isinf(norm)] = 0
g.set_ndata("norm", norm.unsqueeze(1))
torch.manual_seed(1)
model = temporal.STGraphTGCN(feat, hid, 1).to(dev)
x0 = torch.randn(n, feat, device=dev)
targets = torch.randn(B, n, 1, device=dev)
res = {}
for name, fp, fo in (("old", False, False), ("from_p", True, False), ("from_p_folded", True, True)):
    kernels.set_step_wgrad_from_p(fp); kernels.set_step_folded(fo)
    model.zero_grad()
    cost = temporal.window_cost(model, g, x0, None, targets) / (B + 1)
    cost.backward()
    res[name] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}
kernels.check_step_fold_status(dev)
out = {}
for k in res["old"]:
    a = res["old"][k]
    row = {"max_abs": float(a.abs().max())}
    for name in ("from_p", "from_p_folded"):
        d = (res[name][k] - a).abs()
        row[name + "_max_err_over_max"] = float(d.max() / a.abs().max())
        row[name + "_frac_rel_gt_1e-3"] = float((d > 1e-3 * a.abs() + 1e-12).double().mean())
    out[k] = row
print(json.dumps(out, indent=1))
