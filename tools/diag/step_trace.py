#!/usr/bin/env python3
"""Per-wave phase timestamps of the TGCN step kernels (library built by tools/diag/build_step_trace.sh with -DSTG_STEP_TRACE;
copy it over stgraph_amd/lib/libstgraph_hip.so on the GPU box before running this).  Prints, per kernel, when (us after the
first wave started) the waves passed each mark: min / median / max over the one-tile waves, and over the waves that took a second
tile, plus the shader clock the launch ran at."""
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

C, FIN, FH, SLOTS = 64, 32, 32, 16


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
    WcatT = p["Wcat"].t().contiguous()
    folded = "--folded" in sys.argv            # the folded form (the window nodes' default): forward from P, backward without da3
    bo_extra = {}
    if folded:
        Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
        bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
        w_fold, b_fold, bound, w_fold_t = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"],
                                                                    with_bound=True)
        out.update(w_fold=w_fold, b_fold=b_fold, fold_bound=bound, x3=None)
        bo_extra = dict(w_fold_t=w_fold_t)

    def fwd():
        kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices,
                              node_ids=None, norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT,
                              b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"],
                              W1=p["W1"], b1=p["b1"], W2=p["W2"], b2=p["b2"], **out)
    bo = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=None if folded else new(n, 3 * C), dH=new(n, C), dyt=new(n, FH), dyo=new(n),
              z=new(n, FIN), **bo_extra)
    zn, dHn, gc = r(n, FIN), r(n, C), torch.ones(1, device=dev)
    T = {k: p[k].t().contiguous() for k in ("Wz", "Wr", "Wh", "W1")}

    def bwd():
        kernels.tgcn_step_bwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=b.row_offset, column_indices=b.column_indices,
                              node_ids=None, norm_col_edge=ncb, ew_edge=ewb, norm=norm.view(-1), zn=zn, dHn=dHn, g_cost=gc, Z=out["Z"],
                              R=out["R"], Ht=out["Ht"], H=H, Hn=out["Hn"], x3=None, clamp_mask=out["clamp_mask"], y_out=out["y_out"],
                              target=tgt, WzT=T["Wz"], WrT=T["Wr"], WhT=T["Wh"], Wcat=p["Wcat"], W1T=T["W1"], W2=p["W2"], **bo)

    tiles = -(-n // 16)
    grid, waves = min(256, -(-tiles // 12) if tiles >= 256 * 12 else min(tiles, 256)), 12
    grid = 256 if tiles >= 256 else tiles
    res = {"N": n, "tiles": tiles}
    for name, fn, setter, last in (("fwd", fwd, "stg_debug_set_step_trace_fwd", 6), ("bwd", bwd, "stg_debug_set_step_trace_bwd", 7)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        buf = torch.zeros(grid * waves * SLOTS, dtype=torch.int64, device=dev)
        set_fn = getattr(_C.lib, setter)
        set_fn.argtypes, set_fn.restype = [ctypes.c_void_p], ctypes.c_int
        assert set_fn(buf.data_ptr()) == 0
        fn()
        torch.cuda.synchronize()
        assert set_fn(None) == 0
        t = buf.cpu().numpy().reshape(grid * waves, SLOTS).astype(np.int64)
        wave_id = np.arange(grid * waves)
        first_tile = (wave_id % waves) * grid + wave_id // waves
        ran = t[:, 0] > 0
        two = ran & (first_tile + grid * waves < tiles)
        one = ran & ~two & (first_tile < tiles)
        t0 = t[ran, 0].min()
        us = lambda col, m: ((t[m, col] - t0) / 100.0)  # noqa: E731   100 MHz wall clock
        rows = {}
        for k in range(last + 1):
            a = us(k, one)
            rows[f"mark{k}_one_tile"] = [round(float(v), 2) for v in np.percentile(a, [0, 10, 50, 90, 99, 100])]
            if two.any():
                bb = us(k, two)
                rows[f"mark{k}_two_tiles"] = [round(float(v), 2) for v in (bb.min(), np.median(bb), bb.max())]
        dur = (t[one, last] - t[one, 0]) / 100.0
        clk = (t[one, 15] - t[one, 14]) / np.maximum(dur, 1e-9) / 1e3
        rows["wave_duration_us"] = [round(float(v), 2) for v in np.percentile(dur, [0, 10, 50, 90, 99, 100])]
        # per SIMD (block, wave % 4): when its first wave left the gather, when its last wave finished, and the phase lengths per wave
        tt = ((t - t0) / 100.0).reshape(grid, waves, SLOTS)
        simd_first = np.stack([tt[:, s::4, 2].min(1) for s in range(4)], 1).reshape(-1)
        simd_last_g = np.stack([tt[:, s::4, 2].max(1) for s in range(4)], 1).reshape(-1)
        simd_end = np.stack([tt[:, s::4, last].max(1) for s in range(4)], 1).reshape(-1)
        rows["simd_first_gather_done"] = [round(float(v), 2) for v in np.percentile(simd_first, [0, 50, 100])]
        rows["simd_last_gather_done"] = [round(float(v), 2) for v in np.percentile(simd_last_g, [0, 50, 100])]
        rows["simd_end"] = [round(float(v), 2) for v in np.percentile(simd_end, [0, 10, 50, 90, 100])]
        rows["phase_us_per_wave_p10_p50_p90"] = {f"{k}->{k + 1}": [round(float(v), 2) for v in np.percentile((t[one, k + 1] - t[one, k]) / 100.0, [10, 50, 90])]
                                                 for k in range(last)}
        cls = (wave_id % waves) // 4
        rows["phase_us_median_by_wave_class"] = {f"waves {4 * c}-{4 * c + 3}": {f"{k}->{k + 1}": round(float(np.median((t[one & (cls == c), k + 1] - t[one & (cls == c), k]) / 100.0)), 2)
                                                                                for k in range(last)} for c in range(3)}
        if name == "fwd":
            rows["fwd_gate0_us_p10_p50_p90"] = {k: [round(float(v), 2) for v in np.percentile((t[one, b] - t[one, a]) / 100.0, [10, 50, 90])]
                                                for k, (a, b) in {"x3 product + stores + clamp (32 MFMA)": (2, 8), "gate product (128 MFMA)": (8, 9),
                                                                  "sigmoid + Z stores": (9, 3)}.items()}
        rows["shader_clock_GHz_median"] = round(float(np.median(clk)), 3)
        rows["waves_one_two"] = [int(one.sum()), int(two.sum())]
        res[name] = rows
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
