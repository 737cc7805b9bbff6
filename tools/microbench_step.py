#!/usr/bin/env python3
"""Device time of the one-launch TGCN step kernels (csrc/tgcn_step.hip) at the cfg4 shape, next to the launches they
replace (aggregate-then-transform + fused cell + head forward; head + cell backward + backward aggregation)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import degree_norm, synthetic_graph
from stgraph_amd import kernels
from stgraph_amd.graph import StaticGraph

C, FIN, FH = 64, 32, 32


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=50_000)
    ap.add_argument("--edges", type=int, default=500_000)
    ap.add_argument("--waves", type=str, default="")
    ap.add_argument("--packed-grid", action="store_true", help="step_spread = 1: fill CUs with 12 tiles each instead of one workgroup per CU")
    ap.add_argument("--nid", action="store_true", help="visit rows in degree order (node_ids), as the layers do on a StaticGraph")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, e = args.nodes, args.edges
    from stgraph_amd import _C
    if args.waves:
        _C.set_tuning("step_waves", int(args.waves))
    if args.packed_grid:
        _C.set_tuning("step_spread", 1)
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
               y=new(n, FH), y_out=new(n), loss_partial=new(-(-n // 16)),
               clamp_mask=torch.empty(n, 12, dtype=torch.int32, device=dev))     # as temporal.window_cost passes it
    ncf, ewf = kernels._edge_gathered(f, "norm", norm, f.column_indices), kernels._edge_gathered(f, "ew", ew, f.eids)
    ncb, ewb = kernels._edge_gathered(b, "norm", norm, b.column_indices), kernels._edge_gathered(b, "ew", ew, b.eids)
    WcatT = p["Wcat"].t().contiguous()

    def fwd():
        kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices,
                              node_ids=f.node_ids if args.nid else None,
                              norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT,
                              b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"],
                              W1=p["W1"], b1=p["b1"], W2=p["W2"], b2=p["b2"], **out)
    bo = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=new(n, 3 * C), dH=new(n, C), dyt=new(n, FH), dyo=new(n), z=new(n, FIN))
    zn, dHn, gc = r(n, FIN), r(n, C), torch.ones(1, device=dev)
    T = {k: p[k].t().contiguous() for k in ("Wz", "Wr", "Wh", "W1")}

    def bwd():
        kernels.tgcn_step_bwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=b.row_offset, column_indices=b.column_indices,
                              node_ids=b.node_ids if args.nid else None,
                              norm_col_edge=ncb, ew_edge=ewb, norm=norm.view(-1), zn=zn, dHn=dHn, g_cost=gc, Z=out["Z"],
                              R=out["R"], Ht=out["Ht"], H=H, Hn=out["Hn"], x3=None, clamp_mask=out["clamp_mask"], y_out=out["y_out"], target=tgt,
                              WzT=T["Wz"], WrT=T["Wr"], WhT=T["Wh"], Wcat=p["Wcat"], W1T=T["W1"], W2=p["W2"], **bo)
    res = {"N": n, "E": e, "waves": args.waves or "auto", "step_fwd_us": timed(fwd), "step_bwd_us": timed(bwd)}
    # the same two launches without the x3 / da3 stores (both optional: the weight gradients can be formed from P^T d_g)
    keep = out["x3"], bo["da3"]
    out["x3"], bo["da3"] = None, None
    res["step_fwd_no_x3_us"], res["step_bwd_no_da3_us"] = timed(fwd), timed(bwd)
    out["x3"], bo["da3"] = keep
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    ref = {k: v.clone() for k, v in out.items()}
    # the folded form on the fp32 instruction (the default of the window nodes): forward from P on the folded weights, no x3;
    # backward with z from d_g and the folded weights' transposed P part, no da3
    w_fold2, b_fold2, bound, w_fold_t = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"],
                                                                 with_bound=True)
    res["fold_weights_with_bound_us"] = timed(lambda: kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"],
                                                                                p["bh"], with_bound=True))
    keep2 = out["x3"], bo["da3"]
    out["w_fold"], out["b_fold"], out["fold_bound"], out["x3"], bo["w_fold_t"], bo["da3"] = w_fold2, b_fold2, bound, None, w_fold_t, None
    fwd()
    res["step_fwd_folded_fp32_us"], res["step_bwd_folded_fp32_us"] = timed(fwd), timed(bwd)
    res["step_folded_fp32_max_err_vs_fp32_form"] = {k: float((out[k] - ref[k]).abs().max()) for k in ("Z", "R", "Ht", "Hn", "HR", "y", "y_out")}
    del out["fold_bound"], bo["w_fold_t"], out["w_fold"], out["b_fold"]
    out["x3"], bo["da3"] = keep2
    res["gcn_agg_F32_us"] = timed(lambda: kernels.gcn_agg(x, norm, norm, f, ew=ew))
    fwd()

    # ablations: which phase costs what
    def fwd_variant(head, gather):
        kw = dict(H=H, b3=p["b3"], Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"], x3=out["x3"],
                  Z=out["Z"], R=out["R"], Ht=out["Ht"], Hn=out["Hn"], HR=out["HR"])
        if gather:
            kw.update(row_offsets=f.row_offset, column_indices=f.column_indices, norm_col_edge=ncf, ew_edge=ewf,
                      norm=norm.view(-1), x=x, WcatT=WcatT, P=out["P"])
        else:
            kw.update(a3=a3c)
        if head:
            kw.update(W1=p["W1"], b1=p["b1"], W2=p["W2"], b2=p["b2"], target=tgt, y=out["y"], y_out=out["y_out"],
                      loss_partial=out["loss_partial"])
        return lambda: kernels.tgcn_step_fwd(n, C, FIN, FH, head, -1e6, 1e6, dev, **kw)
    a3c = torch.randn(n, 3 * C, device=dev) * 0.2
    res["fwd_cell_only_us"] = timed(fwd_variant(0, False))
    res["fwd_cell_head_us"] = timed(fwd_variant(2, False))
    res["fwd_gather_cell_us"] = timed(fwd_variant(0, True))

    def bwd_variant(head, gather, want_z=True):
        kw = dict(dHn=dHn, Z=out["Z"], R=out["R"], Ht=out["Ht"], H=H, x3=out["x3"], WzT=T["Wz"], WrT=T["Wr"], WhT=T["Wh"],
                  Wcat=p["Wcat"], dzl=bo["dzl"], drl=bo["drl"], dhl=bo["dhl"], da3=bo["da3"], dH=bo["dH"],
                  z=bo["z"] if want_z else None)
        if head:
            kw.update(Hn=out["Hn"], y_out=out["y_out"], target=tgt, g_cost=gc, W1T=T["W1"], W2=p["W2"], dyt=bo["dyt"], dyo=bo["dyo"])
        if gather:
            kw.update(row_offsets=b.row_offset, column_indices=b.column_indices, norm_col_edge=ncb, ew_edge=ewb,
                      norm=norm.view(-1), zn=zn)
        return lambda: kernels.tgcn_step_bwd(n, C, FIN, FH, head, -1e6, 1e6, dev, **kw)
    res["bwd_cell_only_us"] = timed(bwd_variant(0, False))
    res["bwd_cell_only_noz_us"] = timed(bwd_variant(0, False, False))
    res["bwd_cell_head_us"] = timed(bwd_variant(2, False))

    # the launches they replace
    def old_fwd():
        a3, P = kernels.gcn_agg_transform(x, p["Wcat"], norm, norm, f, ew=ew)
        Hn, extra = kernels.tgcn_cell_fused_fwd(a3, p["b3"], H, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"], -1e6, 1e6)
        kernels.tgcn_head_fwd(Hn, p["W1"], p["b1"], p["W2"].view(1, -1), p["b2"], tgt.view(-1, 1))
        return a3, Hn, extra
    res["old_fwd_us"] = timed(old_fwd)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
