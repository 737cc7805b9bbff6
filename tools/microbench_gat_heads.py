#!/usr/bin/env python3
"""Device time of the GAT per-head products at cfg3 (|V| = 256 K, 8 heads of 64 over 64 inputs): csrc/gat_heads_x3.hip (knob rowgemm_x3 = 0)
against the launches it replaces (= 1): fc scores, fc out + elu, backward per-vertex pass + g W_h."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stgraph_amd import _C, kernels

N, fin, H, D = 256_000, 64, 8, 64
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
x, W, al, ar = r(N, fin), r(H * D, fin) / 8, r(H, D), r(H, D)
S = torch.rand(N, H, 1, device=dev, generator=gen) * 5 + 0.5
out, g = r(N, H, D), r(N, H, D)
P, st = kernels._ptr, kernels._stream_ptr(dev)
new = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
o, a, gp, pack, ger, gW = new(N, H, D), new(N, H, D), new(N, H, D), new(N, 16), new(N, H, 1), new(H, N, fin)


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bwd_split():
    _C.check(_C.lib.stg_gat_bwd_prepass(P(S), P(out), P(g), P(gp), P(pack), N, H, D, 0.2, P(ger), st))
    _C.check(_C.lib.stg_rowgemm_heads_f32(P(gp), P(W), P(gW), N, D, fin, H, st))


res = {}
for knob in (0, 1):
    _C.set_tuning("rowgemm_x3", knob)
    k = "x3" if knob == 0 else "fp32"
    res[f"fc_scores_{k}_us"] = timed(lambda: kernels.gat_fc_fwd(x, W, al, ar, H, D, store_feat=False))
    res[f"fc_feat_scores_{k}_us"] = timed(lambda: kernels.gat_fc_fwd(x, W, al, ar, H, D))
    res[f"fc_out_elu_{k}_us"] = timed(lambda: _C.check(_C.lib.stg_gat_fc_out(P(x), P(W), P(o), P(a), N, fin, H, D, st)))
_C.set_tuning("rowgemm_x3", 0)
res["bwd_prepass_then_heads_us"] = timed(bwd_split)
res["bwd_prepass_heads_one_pass_us"] = timed(lambda: _C.check(_C.lib.stg_gat_bwd_prepass_heads(
    P(S), P(out), P(g), P(gp), P(pack), P(ger), P(W), P(gW), N, H, D, fin, 0.2, st)))
res["bytes"] = {"fc_out_elu": 4 * N * (fin + 2 * H * D), "bwd_one_pass": 4 * N * H * (3 * D + fin)}
print(json.dumps(res))
