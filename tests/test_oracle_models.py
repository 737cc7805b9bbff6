"""Pin the oracle-backed CPU layers (tests/oracle_layers.py) against fixtures produced by the
REFERENCE layers: TGCN BPTT on a StaticGraph, incl. every parameter gradient."""
import numpy as np
import pytest
import torch

from tests.oracle_layers import OracleGraphView, make_oracle_tgcn
from tests.util import golden


class Model(torch.nn.Module):
    def __init__(self, fin, hid, out):
        super().__init__()
        self.temporal = make_oracle_tgcn()(fin, hid)
        self.linear = torch.nn.Linear(hid, out)

    def forward(self, g, x, ew, hidden):
        h = self.temporal(g, x, ew, hidden)
        return self.linear(torch.relu(h)), h


@pytest.mark.parametrize("B", [3, 6])
def test_oracle_tgcn_matches_reference_bptt(B):
    torch.set_num_threads(1)
    d = golden("tgcn.npz")
    n, T = int(d["num_nodes"]), d["feats"].shape[0]
    g = OracleGraphView(d["src"], d["dst"], n)
    g.set_ndata("norm", torch.from_numpy(d["norm"]))
    w = torch.from_numpy(d["edge_weight_by_eid"])
    feats, targets = torch.from_numpy(d["feats"]), torch.from_numpy(d["targets"])
    model = Model(feats.shape[2], 16, 1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d[f"B{B}_param_{k}"]))
    hs = []
    for i, w0 in enumerate(range(0, T, B)):
        model.zero_grad()
        hidden, cost = None, 0
        for t in range(w0, w0 + B):
            y, hidden = model(g, feats[t], w, hidden)
            cost = cost + torch.mean((y - targets[t]) ** 2)
            hs.append(hidden.detach())
        cost = cost / (B + 1)
        cost.backward()
        np.testing.assert_allclose(cost.item(), d[f"B{B}_cost"][i], rtol=1e-5, atol=1e-6)
        for k, p in model.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), d[f"B{B}_w{w0}_grad_{k}"], rtol=1e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(torch.stack(hs).numpy(), d[f"B{B}_hidden"], rtol=1e-5, atol=1e-6)
