#!/usr/bin/env python3
"""Where do parameters of the static-temporal training loop leave the reference run after 4 Adam steps, by optimizer form?
(diagnosis for tests/test_gpu_models_ref.py::test_static_training_loop_matches_the_reference_adam_run: per tensor the largest
errors with the reference's recorded gradients of the four steps beside them)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from stgraph_amd import kernels, temporal
from tests.test_gpu_models_ref import _load, _static_setup, _x0
from tests.util import golden

dev = torch.device("cuda", 0)
d = golden("tgcn_native.npz")
out = {}
for folded in (True, False):
    kernels.set_step_folded(folded)
    for mode in ("eager", "hip_graph", "capturable", "capturable_fused"):
        g, targets, ew, n, T = _static_setup(d, dev, True)
        feat, hid, B = int(d["feat"]), int(d["hidden"]), 3
        model = temporal.STGraphTGCN(feat, hid, 1).to(dev)
        _load(model, d, "train_param0_", dev)
        cap = mode.startswith("capturable")
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=cap, fused=True if mode.endswith("fused") else None)
        bucket = temporal.GradBucket(model.parameters())
        base = int(d["train_x0_seed_base"])
        temporal.window_input = lambda num_nodes, f, epoch, w, device, seed=0, out=None, base=base: (
            _x0(base + epoch * 2 + w, num_nodes, f, device) if out is None else out.copy_(_x0(base + epoch * 2 + w, num_nodes, f, device)))
        cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, feat) if mode != "eager" else None
        grads = []
        for epoch in range(2):
            if cw is not None:
                temporal.train_epoch_static_captured(cw, model, g, ew, targets, opt, bucket, feat, epoch=epoch)
            else:
                temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=epoch)
        rep = {}
        for k, p in model.named_parameters():
            err = np.abs(p.detach().cpu().numpy() - d["train_paramT_" + k]).reshape(-1)
            gs = np.stack([np.abs(d[f"train_grad{s}_{k}"]).reshape(-1) for s in range(4)])
            top = np.argsort(err)[-3:][::-1]
            rep[k] = {"max_err": float(err.max()), "gmax": float(gs.max()), "frac_gt_2e-5": float((err > 2e-5).mean()),
                      "top": [{"err": float(err[i]), "g": [float(v) for v in gs[:, i]]} for i in top]}
        out[f"{'folded' if folded else 'reference_form'}:{mode}"] = rep
print(json.dumps(out))
