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
base = int(d["train_x0_seed_base"])
def chunk(num_nodes, f, epoch, c, device, seed=0, out=None):
    cw = temporal.chunk_windows(num_nodes, f)
    buf = out if out is not None else torch.zeros(cw, num_nodes, f, device=device)
    for w in range(min(cw, T // B)):
        buf[w].copy_(_x0(base + epoch * 10 + c * cw + w, num_nodes, f, device))
    return buf
temporal.window_input_chunk = chunk
res = {}
for captured, capturable in ((False, False), (False, True), (True, False), (True, True)):
    temporal._LAST_CHUNK.clear()
    model = temporal.STGraphTGCN(feat, hid, 1).to(cuda)
    _load(model, d, "train_param0_", cuda)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=capturable)
    bucket = temporal.GradBucket(model.parameters())
    cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, feat) if captured else None
    costs = []
    for epoch in range(2):
        if captured:
            costs += temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, feat, epoch=epoch)
        else:
            costs += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=epoch)
    res[(captured, capturable)] = ({k: p.detach().cpu().numpy().copy() for k, p in model.named_parameters()}, [float(c) for c in costs])

ref = res[(False, False)][0]
for key in res:
    print(key, res[key][1])
    for k in ("temporal.linear_z.weight", "temporal.conv_z.weight", "linear.weight"):
        w = d["train_paramT_" + k]; c = res[key][0][k]
        print(f"   {k:28s} vs ref {np.abs(c-w).max():.2e}  vs eager {np.abs(c-ref[k]).max():.2e}")
