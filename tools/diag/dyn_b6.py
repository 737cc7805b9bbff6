import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.util import golden
from tests.test_gpu_models_ref import _t, _load, _x0
from stgraph_amd import temporal
from stgraph_amd.graph import NaiveGraph
cuda = torch.device("cuda", 0)
d = golden("dyn_tgcn.npz")
n, T, feat, hid, M = (int(d[k]) for k in ("num_nodes", "T", "feat", "hidden", "M"))
snaps = [[(int(a), int(b)) for a, b in zip(d[f"t{t}_src"], d[f"t{t}_dst"])] for t in range(T)]
edges = [_t(d[f"t{t}_label_edges"], cuda) for t in range(T - 1)]; edges.append(edges[-1])
targets = [torch.cat([torch.ones(M), torch.zeros(M)]).to(cuda) for _ in range(T)]
for B in (3, 4, 5, 6):
  for fused in (True, False):
    temporal.set_fused_window(fused)
    G = NaiveGraph(snaps, n, device=cuda, resident=True)
    model = temporal.DynamicSTGraphTGCN(feat, hid).to(cuda)
    _load(model, d, f"B6_param_", cuda)
    G.reset_graph()
    model.zero_grad()
    x0 = _x0(int(d["x0_seed_base"]), n, feat, cuda)
    G.get_graph(0)
    steps = []; cost = 0; hidden = None; y_hat = x0
    for t in range(B):
        G.get_graph(t)
        if G.get_ndata("norm") is None:
            G.set_ndata("norm", temporal.in_degree_norm(G))
        if fused:
            steps.append(dict(fwd=G.csr("fwd"), bwd=G.csr("bwd"), norm=G.get_ndata("norm"), edges=edges[t],
                              targets=targets[t], incidence=temporal.SF._incidence_of(edges[t], n)))
        else:
            cost, y_hat, hidden = model.step_loss(G, y_hat, None, hidden, edges[t], targets[t], cost)
    if fused:
        cost = temporal.dyn_window_cost(model, G, x0, steps)
    cost = cost / (6 + 1)
    cost.backward()
    gr = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
    if fused: gf = gr; cf = float(cost)
    else:
        print("B", B, "cost fused", cf, "unfused", float(cost), "ref(B6)", d["B6_cost"])
        for k in gr:
            w = d[f"B6_w0_grad_{k}"]
            print(f"   {k:28s} fused-unfused {np.abs(gf[k]-gr[k]).max()/np.abs(gr[k]).max():.2e}" + (f"  fused-ref {np.abs(gf[k]-w).max()/np.abs(w).max():.2e} unfused-ref {np.abs(gr[k]-w).max()/np.abs(w).max():.2e}" if B == 6 else ""))
