"""A/B of the one-launch TGCN step kernels (folded forms, cfg4 shape): tiles of the last partial round shared by four waves
("step_coop" 0) against one wave each (1): device time per launch and bit-identity of every output."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import degree_norm, synthetic_graph  # noqa: E402
from stgraph_amd import _C, kernels  # noqa: E402
from stgraph_amd.graph import StaticGraph  # noqa: E402

C, FIN, FH = 64, 32, 32


def timed(fn, iters=int(os.environ.get('STEP_ITERS', '200')), warm=int(os.environ.get('STEP_WARM', '20'))):
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
    new = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    Wc = [p["Wcat"][:, k * C:(k + 1) * C].contiguous() for k in range(3)]
    bc = [p["b3"][k * C:(k + 1) * C].contiguous() for k in range(3)]
    w_fold, b_fold, bound, w_fold_t = kernels.tgcn_fold_weights(*Wc, *bc, p["Wz"], p["bz"], p["Wr"], p["br"], p["Wh"], p["bh"], with_bound=True)
    ncf, ewf = kernels._edge_gathered(f, "norm", norm, f.column_indices), kernels._edge_gathered(f, "ew", ew, f.eids)
    ncb, ewb = kernels._edge_gathered(b, "norm", norm, b.column_indices), kernels._edge_gathered(b, "ew", ew, b.eids)
    WcatT = p["Wcat"].t().contiguous()
    T = {k: p[k].t().contiguous() for k in ("Wz", "Wr", "Wh", "W1")}
    zn, dHn, gc = r(n, FIN), r(n, C), torch.ones(1, device=dev)
    res, outs = {"N": n, "tiles": -(-n // 16)}, {}
    variants = [int(v) for v in os.environ.get("STEP_VARIANTS", "1,0,1,0").split(",")]
    for coop in variants:
        _C.set_tuning("step_coop", coop)
        out = dict(P=new(n, FIN), x3=None, Z=new(n, C), R=new(n, C), Ht=new(n, C), Hn=new(n, C), HR=new(n, C), y=new(n, FH), y_out=new(n),
                   loss_partial=new(-(-n // 16)), clamp_mask=kernels.step_ones_mask(n, dev), w_fold=w_fold, b_fold=b_fold, fold_bound=bound)
        bo = dict(dzl=new(n, C), drl=new(n, C), dhl=new(n, C), da3=None, dH=new(n, C), dyt=new(n, FH), dyo=new(n), z=new(n, FIN), w_fold_t=w_fold_t)

        def fwd():
            kernels.tgcn_step_fwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=f.row_offset, column_indices=f.column_indices, node_ids=None,
                                  norm_col_edge=ncf, ew_edge=ewf, norm=norm.view(-1), x=x, H=H, target=tgt, WcatT=WcatT, b3=p["b3"],
                                  Wz=p["Wz"], bz=p["bz"], Wr=p["Wr"], br=p["br"], Wh=p["Wh"], bh=p["bh"], W1=p["W1"], b1=p["b1"], W2=p["W2"],
                                  b2=p["b2"], **out)

        def bwd():
            kernels.tgcn_step_bwd(n, C, FIN, FH, 2, -1e6, 1e6, dev, row_offsets=b.row_offset, column_indices=b.column_indices, node_ids=None,
                                  norm_col_edge=ncb, ew_edge=ewb, norm=norm.view(-1), zn=zn, dHn=dHn, g_cost=gc, Z=out["Z"], R=out["R"],
                                  Ht=out["Ht"], H=H, Hn=out["Hn"], x3=None, clamp_mask=out["clamp_mask"], y_out=out["y_out"], target=tgt,
                                  WzT=T["Wz"], WrT=T["Wr"], WhT=T["Wh"], Wcat=p["Wcat"], W1T=T["W1"], W2=p["W2"], **bo)
        fwd(), bwd()
        torch.cuda.synchronize()
        key = {0: "coop", 1: "one_wave"}.get(coop, f"flags{coop}")
        res.setdefault(key, []).append({"fwd_us": timed(fwd), "bwd_us": timed(bwd)})
        outs[key] = {k: v.clone() for k, v in list(out.items()) + list(bo.items()) if torch.is_tensor(v) and v.dtype == torch.float32 and k not in ("w_fold", "b_fold", "fold_bound", "w_fold_t")}
    _C.set_tuning("step_coop", 0)
    base = outs[sorted(outs)[0]]
    res["bit_identical"] = {name: all(bool(torch.equal(o[k], base[k])) for k in base) for name, o in outs.items()}
    res["no_nan_left"] = all(not bool(torch.isnan(v).any()) for o in outs.values() for v in o.values())
    res["fold_status"] = int(kernels.step_fold_status_word(dev).item())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
