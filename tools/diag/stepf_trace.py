#!/usr/bin/env python3
"""Phase timeline of the folded TGCN forward step kernel (library: tools/diag/build_stepf_trace.sh).
    STGRAPH_AMD_LIB=stgraph_amd/lib/diag/stepf_trace.so python tools/diag/stepf_trace.py [N]
Marks per wave (100 MHz wall clock): 0 start, 1 weights staged, 2 gather done, 3 P stored + split, 5 x3 + gates r, z issued, 6 gate h issued,
7 Ht / Hn done, 8 head -- of the wave's LAST tile.  Printed: per mark, the median / max over waves of the time since the launch's first stamp."""
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
    f = g.csr("fwd")
    r = lambda *s: torch.randn(*s, device=dev) * 0.2  # noqa: E731
    p = dict(Wcat=r(FIN, 3 * C), b3=r(3 * C), Wz=r(C, 2 * C), bz=r(C), Wr=r(C, 2 * C), br=r(C), Wh=r(C, 2 * C), bh=r(C),
             W1=r(FH, C), b1=r(FH), W2=r(FH), b2=r(1))
    x, H, tgt = r(n, FIN), r(n, C), r(n)
    new = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
    out = dict(P=new(n, FIN), x3=new(n, 3 * C), Z=new(n, C), R=new(n, C), Ht=new(n, C), Hn=new(n, C), HR=new(n, C),
               y=new(n, FH), y_out=new(n), loss_partial=new(-(-n // 16)), clamp_mask=torch.empty(n, 12, dtype=torch.int32, device=dev))
    ncf, ewf = kernels._edge_gathered(f, "norm", norm, f.column_indices), kernels._edge_gathered(f, "ew", ew, f.eids)
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    w_fold, b_fold = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"])
    WcatT = p["Wcat"].t().contiguous()

    def fwd():
        kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices,
                              node_ids=None, norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT,
                              b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"],
                              W1=p["W1"], b1=p["b1"], W2=p["W2"], b2=p["b2"], w_fold=w_fold, b_fold=b_fold, **out)

    tiles = -(-n // 16)
    waves = 12
    grid = max(1, min(256, tiles))
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    buf = torch.zeros(grid * waves * SLOTS, dtype=torch.int64, device=dev)
    set_fn = _C.lib.stg_debug_set_stepf_trace_fwd
    set_fn.argtypes, set_fn.restype = [ctypes.c_void_p], ctypes.c_int
    assert set_fn(buf.data_ptr()) == 0
    fwd()
    torch.cuda.synchronize()
    assert set_fn(None) == 0
    t = buf.cpu().numpy().reshape(grid * waves, SLOTS).astype(np.int64)[:, :9]
    t0 = t[t > 0].min()
    res = {"N": n, "tiles": tiles, "grid": grid, "waves_per_workgroup": waves, "unit": "us since the first stamp of the launch"}
    names = ["start", "weights_staged", "gather_done", "p_split", "unused", "x3_r_z_issued", "gate_h_issued", "hn_done", "head"]
    for k, name in enumerate(names):
        col = t[:, k]
        col = col[col > 0]
        if col.size == 0:
            continue
        res[name] = {"median": round(float(np.median(col - t0)) / 100, 2), "max": round(float((col - t0).max()) / 100, 2),
                     "min": round(float((col - t0).min()) / 100, 2), "waves": int(col.size)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
