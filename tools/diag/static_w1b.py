import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util import golden
from tests.test_gpu_models_ref import _static_setup, _load, _x0
from stgraph_amd import temporal
cuda = torch.device("cuda", 0)
d = golden("tgcn_native.npz")
g, targets, ew, n, T = _static_setup(d, cuda, False)
feat, hid, B = int(d["feat"]), int(d["hidden"]), 3
tag = "now_B3"
model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
_load(model, d, f"{tag}_param_", cuda)
for index in range(2):
    x0 = _x0(int(d["x0_seed_base"]) + index, n, feat, cuda)
    cost = temporal.window_cost_of(model, g, x0, ew, targets[index * B:(index + 1) * B])
    s = cost.grad_fn.saved_tensors
    Hn, X3 = s[16], s[12]
    v, i = Hn.abs().flatten().topk(5, largest=False)
    print(index, "smallest |h|:", v.tolist(), [(int(k) // (n * hid), (int(k) // hid) % n, int(k) % hid) for k in i])
    v, i = X3.abs().flatten().topk(3, largest=False)
    print(index, "smallest |x3|:", v.tolist())
