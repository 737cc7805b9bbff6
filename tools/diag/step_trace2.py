#!/usr/bin/env python3
"""Where a step launch ends (trace library of tools/diag/build_step_trace.sh, loaded through STGRAPH_AMD_LIB): per wave and per SIMD
(workgroup, wave % 4) the time of the last mark, split by workgroups WITH a tile of the partial round and without; the shared-tile
section's start / end (marks 10 / 11).  argv: N [coop 0|1]."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from tools.diag.step_coop_ab import C, FIN, FH  # noqa: F401
from bench import degree_norm, synthetic_graph
from stgraph_amd import _C, kernels
from stgraph_amd.graph import StaticGraph

SLOTS = 16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    coop_off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    _C.set_tuning("step_coop", coop_off)
    dev = torch.device("cuda", 0)
    e = n * 10
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
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    w_fold, b_fold, bound, _ = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"], with_bound=True)
    out = dict(P=new(n, FIN), x3=None, Z=new(n, C), R=new(n, C), Ht=new(n, C), Hn=new(n, C), HR=new(n, C), y=new(n, FH), y_out=new(n),
               loss_partial=new(-(-n // 16)), clamp_mask=kernels.step_ones_mask(n, dev), w_fold=w_fold, b_fold=b_fold, fold_bound=bound)
    ncf, ewf = kernels._edge_gathered(f, "norm", norm, f.column_indices), kernels._edge_gathered(f, "ew", ew, f.eids)
    WcatT = p["Wcat"].t().contiguous()

    def fwd():
        kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices, node_ids=None,
                              norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT, b3=p["b3"],
                              Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"], W1=p["W1"], b1=p["b1"], W2=p["W2"],
                              b2=p["b2"], **out)
    tiles = -(-n // 16)
    grid, waves = 256, 12
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    buf = torch.zeros(grid * waves * SLOTS, dtype=torch.int64, device=dev)
    set_fn = _C.lib.stg_debug_set_step_trace_fwd
    set_fn.argtypes, set_fn.restype = [ctypes.c_void_p], ctypes.c_int
    assert set_fn(buf.data_ptr()) == 0
    fwd()
    torch.cuda.synchronize()
    assert set_fn(None) == 0
    t = buf.cpu().numpy().reshape(grid, waves, SLOTS).astype(np.int64)
    t0 = t[:, :, 0][t[:, :, 0] > 0].min()
    us = (t - t0) / 100.0
    end = np.maximum(us[:, :, 6], np.where(t[:, :, 11] > 0, us[:, :, 11], 0))
    rem = tiles - (tiles // (grid * waves)) * grid * waves
    has = (np.arange(grid) < rem % grid) | (rem >= grid)
    q = lambda a: [round(float(v), 2) for v in np.percentile(a, [0, 10, 50, 90, 99, 100])]  # noqa: E731
    simd_end = np.stack([end[:, s::4].max(1) for s in range(4)], 1)
    res = {"N": n, "tiles": tiles, "coop": not coop_off, "blocks_with_partial_round_tiles": int(has.sum()),
           "staging_done": q(us[:, :, 1]), "first_gather_done": q(us[:, :, 2]),
           "wave_end_all": q(end), "simd_end_blocks_with": q(simd_end[has]) if has.any() else None,
           "simd_end_blocks_without": q(simd_end[~has]) if (~has).any() else None,
           "wave_end_by_class_median": {f"waves {4 * c}-{4 * c + 3}": round(float(np.median(end[:, 4 * c:4 * c + 4])), 2) for c in range(3)}}
    m = t[:, :, 10] > 0
    if m.any():
        res["shared_tile_start"] = q(us[:, :, 10][m])
        res["shared_tile_end"] = q(us[:, :, 11][m])
        res["shared_tile_duration"] = q((us[:, :, 11] - us[:, :, 10])[m])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
