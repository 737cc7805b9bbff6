#!/usr/bin/env python3
"""gcn_agg at the cfg2 width on a POWER-LAW graph (Chung-Lu, |V| = 1M, |E| ~ 16M, max degree >= 10^5) next to the
uniform cfg2 graph: fraction of the HBM roofline (SURVEY.md 8(d) bytes) forward + backward, with the feature-sliced
long-row workgroups of round 3 on and off.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import HBM_PEAK_GBS, synthetic_graph
from stgraph_amd import _C, kernels
from stgraph_amd.graph import StaticGraph


def chung_lu(n, e, alpha, seed, device):
    gen = torch.Generator(device=device).manual_seed(seed)
    w = torch.arange(1, n + 1, device=device, dtype=torch.float64).pow(-alpha)
    p = (w / w.sum()).float()
    m = int(e * 1.15)
    s = torch.multinomial(p, m, replacement=True, generator=gen)
    d = torch.multinomial(p, m, replacement=True, generator=gen)
    perm = torch.randperm(n, device=device, generator=gen)           # hubs scattered over the id range
    key = torch.unique(perm[s] * n + perm[d])
    key = key[torch.randperm(key.shape[0], device=device, generator=gen)][:e]
    return (key // n).to(torch.int32), (key % n).to(torch.int32)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda", 0)
    n, e, F = 1_000_000, 16_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 128       # width: any (round 4: 300, 131 ...)
    out = {"N": n, "F": F}
    x = torch.randn(n, F, device=dev)
    for name in ("uniform", "power_law"):
        src, dst = synthetic_graph(n, e, 1, dev) if name == "uniform" else chung_lu(n, e, 0.75, 5, dev)
        E = int(src.shape[0])
        g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
        norm = torch.rand(n, 1, device=dev) + 0.5
        f, b = g.csr("fwd"), g.csr("bwd")
        nbytes = kernels.gcn_agg_algorithmic_bytes(n, E, F, False)
        res = {"E": E, "max_in_degree": int((f.row_offset[1:] - f.row_offset[:-1]).max()),
               "max_out_degree": int((b.row_offset[1:] - b.row_offset[:-1]).max()),
               "rows_over_1024": int(((f.row_offset[1:] - f.row_offset[:-1]) > 1024).sum())}
        for mode in ("wide_long_on", "wide_long_off"):
            _C.set_tuning("gcn_wide_long", 0 if mode == "wide_long_on" else 1)
            iters = 10 if (mode == "wide_long_on" or name == "uniform") else 2
            ms = [timed(lambda: kernels.gcn_agg(x, norm, norm, csr), iters) for csr in (f, b)]
            res[mode] = {"fwd_ms": ms[0], "bwd_ms": ms[1], "frac_of_hbm_peak": 2 * nbytes / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        _C.set_tuning("gcn_wide_long", 0)
        out[name] = res
        del g, f, b
        torch.cuda.empty_cache()
    out["power_law_over_uniform"] = out["power_law"]["wide_long_on"]["frac_of_hbm_peak"] / out["uniform"]["wide_long_on"]["frac_of_hbm_peak"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
