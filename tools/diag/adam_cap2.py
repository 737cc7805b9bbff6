import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util import golden
from tests.test_gpu_models_ref import _static_setup, _load, _x0
from stgraph_amd import temporal
cuda = torch.device("cuda", 0)
d = golden("tgcn_native.npz")
g, targets, ew, n, T = _static_setup(d, cuda, True)
feat, hid, B = int(d["feat"]), int(d["hidden"]), 3
model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
_load(model, d, "train_param0_", cuda)
x0 = _x0(int(d["train_x0_seed_base"]), n, feat, cuda)
cost = temporal.window_cost_of(model, g, x0, ew, targets[:B]) / (B + 1)
cost.backward()
W = model.temporal.linear_z.weight
gr = W.grad.detach().clone()
print("grad absmax", gr.abs().max().item(), "min nonzero", gr[gr != 0].abs().min().item(), "zeros", (gr == 0).sum().item())
outs = {}
for name, kw in (("plain", {}), ("cap", dict(capturable=True))):
    p = torch.nn.Parameter(W.detach().clone())
    opt = torch.optim.Adam([p], lr=1e-2, **kw)
    p.grad = gr.clone(); opt.step()
    outs[name] = p.detach().clone()
    st = opt.state[p]
    print(name, {k: (v.dtype, v.device, float(v) if v.numel() == 1 else None) for k, v in st.items()})
diff = (outs["plain"] - outs["cap"]).abs()
idx = diff.flatten().topk(8).indices
print("max diff", diff.max().item())
for i in idx.tolist():
    print(i, "g=%.3e" % gr.flatten()[i].item(), "plain d=%.3e" % (outs["plain"].flatten()[i] - W.detach().flatten()[i]).item(),
          "cap d=%.3e" % (outs["cap"].flatten()[i] - W.detach().flatten()[i]).item())
