#!/usr/bin/env python3
"""Interval timeline of the matrix-core TGCN step kernels (library built by tools/diag/build_stepx_trace.sh; run with
STG_LIB=build/tracex/libstgraph_hip.so copied over stgraph_amd/lib/libstgraph_hip.so on the GPU box).  Per interval: how long the
waves of team 0 / team 1 WORK (enter -> barrier arrival, median over workgroups, max over the team's waves) and how long the
interval lasts (enter -> next enter)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from bench import degree_norm, synthetic_graph
from stgraph_amd import _C, kernels
from stgraph_amd.graph import StaticGraph

C, FIN, FH, SLOTS = 64, 32, 32, 128


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    e = n * 10
    dev = torch.device("cuda", 0)
    src, dst = synthetic_graph(n, e, 3, dev)
    g = StaticGraph((src, dst), None, n, device=dev, sort_inplace=False)
    norm = degree_norm(g)
    ew = torch.rand(e, 1, device=dev) + 0.5
    f, b = g.csr("fwd"), g.csr("bwd")
    r = lambda *s: torch.randn(*s, device=dev) * 0.2  # noqa: E731
    p = dict(Wcat=r(FIN, 3 * C), b3=r(3 * C), Wz=r(C, 2 * C), bz=r(C), Wr=r(C, 2 * C), br=r(C), Wh=r(C, 2 * C), bh=r(C),
             W1=r(FH, C), b1=r(FH), W2=r(FH), b2=r(1))
    x, H, tgt = r(n, FIN), r(n, C), r(n)
    new = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
    out = dict(P=new(n, FIN), x3=new(n, 3 * C), Z=new(n, C), R=new(n, C), Ht=new(n, C), Hn=new(n, C), HR=new(n, C),
               y=new(n, FH), y_out=new(n), loss_partial=new(-(-n // 16)), clamp_mask=torch.empty(n, 12, dtype=torch.int32, device=dev))
    ncf, ewf = kernels._edge_gathered(f, "norm", norm, f.column_indices), kernels._edge_gathered(f, "ew", ew, f.eids)
    ncb, ewb = kernels._edge_gathered(b, "norm", norm, b.column_indices), kernels._edge_gathered(b, "ew", ew, b.eids)
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    img_f, img_b = kernels.tgcn_pack_weights_x3(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"], p["W1"], p["b1"],
                                                p["W2"], p["b2"])
    WcatT = p["Wcat"].t().contiguous()

    def fwd():
        kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices,
                              node_ids=None, norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT,
                              b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"],
                              W1=p["W1"], b1=p["b1"], W2=p["W2"], b2=p["b2"], w_image=img_f, **out)
    bo = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=new(n, 3 * C), dH=new(n, C), dyt=new(n, FH), dyo=new(n), z=new(n, FIN))
    zn, dHn, gc = r(n, FIN), r(n, C), torch.ones(1, device=dev)
    T = {k: p[k].t().contiguous() for k in ("Wz", "Wr", "Wh", "W1")}

    def bwd():
        kernels.tgcn_step_bwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=b.row_offset, column_indices=b.column_indices,
                              node_ids=None, norm_col_edge=ncb, ew_edge=ewb, norm=norm.view(-1), zn=zn, dHn=dHn, g_cost=gc, Z=out["Z"],
                              R=out["R"], Ht=out["Ht"], H=H, Hn=out["Hn"], x3=None, clamp_mask=out["clamp_mask"], y_out=out["y_out"],
                              target=tgt, WzT=T["Wz"], WrT=T["Wr"], WhT=T["Wh"], Wcat=p["Wcat"], W1T=T["W1"], W2=p["W2"], w_image=img_b, **bo)

    tiles = -(-n // 16)
    grid = max(1, min(256, (tiles + 1) // 2))
    res = {"N": n, "tiles": tiles, "grid": grid}
    for name, fn, setter in (("fwd", fwd, "stg_debug_set_stepx_trace_fwd"), ("bwd", bwd, "stg_debug_set_stepx_trace_bwd")):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        buf = torch.zeros(grid * 8 * SLOTS, dtype=torch.int64, device=dev)
        set_fn = getattr(_C.lib, setter)
        set_fn.argtypes, set_fn.restype = [ctypes.c_void_p], ctypes.c_int
        assert set_fn(buf.data_ptr()) == 0
        fn()
        torch.cuda.synchronize()
        assert set_fn(None) == 0
        t = buf.cpu().numpy().reshape(grid, 8, SLOTS // 2, 2).astype(np.int64)
        t0 = t[t > 0].min()
        nit = int((t[0, 0, :, 0] > 0).sum())
        rows = []
        for it in range(nit):
            enter, arrive = t[:, :, it, 0], t[:, :, it, 1]
            nxt = t[:, :, it + 1, 0] if it + 1 < nit else arrive
            work = (arrive - enter) / 100.0                   # us per wave
            rows.append({"it": it, "enter_us_median": float(np.median((enter - t0) / 100.0)),
                         "team0_work_us": float(np.median(work[:, :4].max(1))), "team1_work_us": float(np.median(work[:, 4:].max(1))),
                         "team0_work_by_wave": [float(np.median(work[:, w])) for w in range(4)],
                         "team1_work_by_wave": [float(np.median(work[:, 4 + w])) for w in range(4)],
                         "interval_us": float(np.median((nxt - enter).max(1) / 100.0))})
        res[name] = {"intervals": nit, "launch_us_by_trace": float((t.max() - t0) / 100.0), "rows": rows}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
