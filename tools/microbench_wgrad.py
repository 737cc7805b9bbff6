#!/usr/bin/env python3
"""Weight-gradient contraction of a TGCN gate Linear over a 25-snapshot window (K = 50 K rows, M = 64, N = 128) in the
operand forms stg_gemm_tn_form_f32 takes: what do the split B operand and the clamp-on-load cost?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stgraph_amd import kernels


def med(fn, iters=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)) * 1e3


def main():
    dev = torch.device("cuda", 0)
    K, C, T = 50_000, 64, 25
    d = [torch.randn(K, C, device=dev) for _ in range(T)]
    dense = [torch.randn(K, 2 * C, device=dev) for _ in range(T)]
    x3 = [torch.randn(K, 3 * C, device=dev) for _ in range(T)]
    H = [torch.randn(K, C, device=dev) for _ in range(T)]
    res = {"K": K, "T": T}
    res["multi_dense_us"] = med(lambda: kernels.gemm_tn_multi(d, dense, colsum=True))
    res["form_dense_us"] = med(lambda: kernels.gemm_tn_form(d, dense, C, 2 * C, colsum=True))
    res["form_dense_clamp_us"] = med(lambda: kernels.gemm_tn_form(d, dense, C, 2 * C, b_op=kernels.GEMM_B_CLAMP, lo=-1e6, hi=1e6, colsum=True))
    hl = [h[:, :C] for h in dense]
    res["form_split_dense_halves_us"] = med(lambda: kernels.gemm_tn_form(d, hl, C, 2 * C, B2s=H, nsplit=C, colsum=True))
    xs = [x[:, :C] for x in x3]
    res["form_split_strided_us"] = med(lambda: kernels.gemm_tn_form(d, xs, C, 2 * C, B2s=H, nsplit=C, colsum=True))
    res["form_split_strided_clamp_us"] = med(lambda: kernels.gemm_tn_form(d, xs, C, 2 * C, B2s=H, nsplit=C, b_op=kernels.GEMM_B_CLAMP,
                                                                          lo=-1e6, hi=1e6, colsum=True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
