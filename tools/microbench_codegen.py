"""Generated (codegen.py) vs hand-written aggregation kernels on the cfg2 graph: the same GCN vertex function
through both routes.  One JSON line per feature width."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_graph  # noqa: E402
from stgraph_amd import kernels  # noqa: E402
from stgraph_amd.compiler import dispatch  # noqa: E402
from stgraph_amd.compiler.backend.pytorch.torch_callback import STGraphBackendTorch  # noqa: E402
from stgraph_amd.compiler.stgraph import STGraph  # noqa: E402
from stgraph_amd.graph import StaticGraph  # noqa: E402


class Mod(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.stgraph = STGraph(STGraphBackendTorch())


def timed(fn, iters=10):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda", 0)
    n, e = 1_000_000, 16_000_000
    src, dst = synthetic_graph(n, e, 1, dev)
    g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
    norm = torch.rand(n, 1, device=dev) + 0.5
    fn = lambda v: sum([nb.h * nb.norm for nb in v.innbs]) * v.norm  # noqa: E731
    for F in (16, 64, 128):
        x = torch.randn(n, F, device=dev, requires_grad=True)
        R = torch.randn(n, F, device=dev)
        out = {}
        for forced in (False, True):
            dispatch.set_force_generated(forced)
            mod = Mod()
            fc = mod.stgraph.compile(gnn_module=mod)(fn)
            with torch.no_grad():
                fwd = timed(lambda: fc(g=g, n_feats={"h": x, "norm": norm}, e_feats={}))

            def both():
                x.grad = None
                (fc(g=g, n_feats={"h": x, "norm": norm}, e_feats={}) * R).sum().backward()
            fb = timed(both, 5)
            out["generated" if forced else "hand_written"] = {"fwd_ms": fwd, "fwd_bwd_ms_incl_torch_loss": fb}
        dispatch.set_force_generated(False)
        by = kernels.gcn_agg_algorithmic_bytes(n, e, F, False)
        print(json.dumps({"F": F, **out, "algorithmic_GB": by / 1e9,
                          "generated_fwd_frac_of_8TBs": by / (out["generated"]["fwd_ms"] * 1e-3) / 8e12,
                          "hand_written_fwd_frac_of_8TBs": by / (out["hand_written"]["fwd_ms"] * 1e-3) / 8e12}), flush=True)


if __name__ == "__main__":
    main()
