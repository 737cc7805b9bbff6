#!/usr/bin/env python3
"""The tall-skinny weight-gradient contractions (C = sum_t A_t^T B_t over K rows) on the fp32 matrix instruction (knob "gemm_x3" 1)
and as 3-term bf16 splits (2): device time and worst error against fp64 in ulps of sum |a| |b| -- at the shapes the bench meets:
cfg2 [1M,128]^T [1M,128] (+ column sums); a TGCN window's [d_z | d_r]^T [H | P] (25 x 50 K rows, 128 x 96, lda = 192) and
d_h^T [H R | P] (64 x 96); the head's dyt^T relu(Hn) (32 x 64)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import _C, kernels


def med(fn, iters=30):
    """Back-to-back launches between two events (the steady state a training step sees), best of three rounds."""
    for _ in range(5):
        fn()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / iters * 1e3
        best = t if best is None else min(best, t)
    return best


def ulps(got, As, Bs):
    want = sum(a.double().t() @ b.double() for a, b in zip(As, Bs))
    scale = sum(a.double().abs().t() @ b.double().abs() for a, b in zip(As, Bs))
    return float(((got.double() - want).abs() / (scale * 2.0 ** -24)).max())


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
    res = {}
    # cfg2
    K = 1_000_000
    A, B = r(K, 128), r(K, 128)
    for name, knob in (("f32", 1), ("x3", 2)):
        _C.set_tuning("gemm_x3", knob)
        out = kernels.gemm_tn(A, B, colsum=True)
        res[f"cfg2_128x128_{name}"] = {"us": med(lambda: kernels.gemm_tn(A, B, colsum=True)), "ulps": ulps(out[0], [A], [B]),
                                       "colsum_err": float((out[1].double() - A.double().sum(0)).abs().max())}
    del A, B
    # a TGCN window
    Kw, T, C, Fin = 50_000, 25, 64, 32
    D3 = [r(Kw, 3 * C) for _ in range(T)]
    H, HR, P, Hn, dyt = ([r(Kw, w) for _ in range(T)] for w in (C, C, Fin, C, Fin))
    cases = {"zr_128x96": dict(As=[d[:, :2 * C] for d in D3], Bs=H, B2s=P, M=2 * C, N=C + Fin, nsplit=C, colsum=True),
             "h_64x96": dict(As=[d[:, 2 * C:] for d in D3], Bs=HR, B2s=P, M=C, N=C + Fin, nsplit=C, colsum=True),
             "head_32x64_relu": dict(As=dyt, Bs=Hn, M=Fin, N=C, b_op=kernels.GEMM_B_RELU, colsum=True)}
    for cname, kw in cases.items():
        Bfull = ([torch.cat([b, b2], 1) for b, b2 in zip(kw["Bs"], kw["B2s"])] if "B2s" in kw else
                 [torch.relu(b) for b in kw["Bs"]])
        for name, knob in (("f32", 1), ("x3", 2)):
            _C.set_tuning("gemm_x3", knob)
            out = kernels.gemm_tn_form(**kw)
            res[f"tgcn_{cname}_{name}"] = {"us": med(lambda: kernels.gemm_tn_form(**kw)), "ulps": ulps(out[0], kw["As"], Bfull),
                                           "colsum_err": float((out[1].double() - sum(a.double().sum(0) for a in kw["As"])).abs().max())}
    _C.set_tuning("gemm_x3", 0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
