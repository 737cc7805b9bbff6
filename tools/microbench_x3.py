#!/usr/bin/env python3
"""Row products at K, M in {64, 128}: the fp32 row-piece kernel against the 3-term bf16 split on the matrix cores
(stg_set_tuning("rowgemm_x3", 1 / 2)) and torch; time and worst error against fp64 in units of 2^-24 * (|x| . |w|)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stgraph_amd import _C, kernels
from tools.microbench_gemm import t_ms


def main():
    dev = torch.device("cuda", 0)
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"        # diagnosis builds (tools/diag/x3_ablate.sh): times only
    if quick:
        x = torch.randn(1_000_000, 128, device=dev)
        w = torch.randn(128, 128, device=dev)
        rec = {}
        for name, knob in (("bf16_x3", 2), ("bf16_x3_ring", 3)):
            _C.set_tuning("rowgemm_x3", knob)
            rec[name + "_ms"] = round(t_ms(lambda: kernels.rowgemm_act(x, w, None, True)), 4)
        print(json.dumps(rec), flush=True)
        return
    for N, K, M, tw in ((1_000_000, 128, 128, False), (1_000_000, 128, 128, True), (256_000, 64, 128, True),
                        (1_000_000, 64, 64, True), (50_000, 128, 64, True)):
        x = torch.randn(N, K, device=dev)
        w = torch.randn((M, K) if tw else (K, M), device=dev)
        rec = {"N": N, "K": K, "M": M, "trans_w": tw}
        wd = w.double().t() if tw else w.double()
        rows = slice(0, 50_000)
        want = x[rows].double() @ wd
        scale = x[rows].double().abs() @ wd.abs()
        for name, knob in (("fp32_mfma", 1), ("bf16_x3", 2), ("bf16_x3_ring", 3)):
            _C.set_tuning("rowgemm_x3", knob)
            rec[name + "_ms"] = round(t_ms(lambda: kernels.rowgemm_act(x, w, None, tw)), 4)
            got = kernels.rowgemm_act(x, w, None, tw)
            rec[name + "_err_ulps_of_scale"] = round(float(((got[rows].double() - want).abs() / scale).max()) * 2 ** 24, 3)
        _C.set_tuning("rowgemm_x3", 0)
        rec["torch_ms"] = round(t_ms((lambda: torch.mm(x, w.t())) if tw else (lambda: torch.mm(x, w))), 4)
        rec["hbm_floor_ms"] = round(4 * N * (K + M) / 8e9, 4)
        rec["x3_GBps"] = round(4 * N * (K + M) / rec["bf16_x3_ms"] / 1e6, 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
