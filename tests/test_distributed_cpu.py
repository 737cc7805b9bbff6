"""The N > 1 path on CPU: 2 ranks over gloo run the window-sharded TGCN loop
(stgraph_amd.temporal) with the oracle-backed layers; the result must equal a single process
that consumes the same two windows per optimizer step and averages their gradients."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stgraph_amd import temporal
from tests.oracle_layers import OracleGraphView, gcn_norm_tensor, make_oracle_tgcn
from tests.util import random_graph

N, E, FEAT, HID, T, B, EPOCHS, SEED = 60, 500, 4, 8, 10, 2, 2, 5


def _problem():
    src, dst = random_graph(21, N, E)
    g = OracleGraphView(src, dst, N)
    g.set_ndata("norm", gcn_norm_tensor(g.in_degrees()))
    rng = np.random.default_rng(3)
    ew = torch.from_numpy(rng.uniform(0.5, 1.5, (E, 1)).astype(np.float32))
    targets = torch.from_numpy(rng.standard_normal((T, N, 1)).astype(np.float32))
    torch.manual_seed(SEED)
    model = temporal.STGraphTGCN(FEAT, HID, 1, tgcn_cls=make_oracle_tgcn())
    return g, ew, targets, model


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, ew, targets, model = _problem()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        bucket = temporal.GradBucket(model.parameters())
        losses = []
        for ep in range(EPOCHS):
            losses += temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, FEAT, epoch=ep,
                                                  rank=rank, world=world, seed=SEED)
            bucket.check_views()
        torch.save({"params": [p.detach().clone() for p in model.parameters()],
                    "losses": torch.stack(losses), "calls": bucket.comm_calls},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_process_equivalent(world):
    """Same global batch without any collective: per step, average the gradients of `world` windows."""
    g, ew, targets, model = _problem()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    nwin = temporal.num_windows(T, B)
    losses = {r: [] for r in range(world)}
    for ep in range(EPOCHS):
        for s in range((nwin + world - 1) // world):
            grads = [torch.zeros_like(p) for p in model.parameters()]
            for r in range(world):
                w = s * world + r
                if w >= nwin:
                    continue
                model.zero_grad()
                hidden, cost = None, 0
                y_hat = temporal.window_input(N, FEAT, ep, w, targets.device, SEED)
                for k in range(B):
                    t = w * B + k
                    if t >= T:
                        break
                    y_out, y_hat, hidden = model(g, y_hat, ew, hidden)
                    cost = cost + torch.mean((y_out - targets[t]) ** 2)
                cost = cost / (B + 1)
                cost.backward()
                losses[r].append(cost.detach())
                for acc, p in zip(grads, model.parameters()):
                    acc += p.grad
            for acc, p in zip(grads, model.parameters()):
                p.grad = acc / world
            opt.step()
    return [p.detach() for p in model.parameters()], losses


def test_window_schedule():
    assert temporal.num_windows(1000, 25) == 40 and temporal.num_windows(10, 3) == 4 and temporal.num_windows(7, 0) == 1
    seen = []
    for r in range(4):
        sched = temporal.windows_of_rank(10, 3, r, 4)
        assert [s for s, _ in sched] == [0]
        seen += [w for _, w in sched if w is not None]
    assert sorted(seen) == [0, 1, 2, 3]
    sched = temporal.windows_of_rank(10, 2, 1, 2)        # 5 windows on 2 ranks: rank 1 pads the last step
    assert sched == [(0, 1), (1, 3), (2, None)]
    a = temporal.window_input(5, 3, 1, 2, "cpu", 7)
    assert torch.equal(a, temporal.window_input(5, 3, 1, 2, "cpu", 7))
    assert not torch.equal(a, temporal.window_input(5, 3, 1, 3, "cpu", 7))


def test_grad_bucket_aliases_parameter_grads():
    lin = torch.nn.Linear(3, 2)
    b = temporal.GradBucket(lin.parameters())
    assert b.nbytes == (6 + 2) * 4
    lin(torch.ones(4, 3)).sum().backward()
    assert b.flat.abs().sum() > 0
    b.check_views()
    b.zero()
    assert not lin.weight.grad.any()
    lin.zero_grad()                                    # set_to_none drops the views: must be detected
    with pytest.raises(RuntimeError):
        b.check_views()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_equal_single_process_with_same_global_batch():
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    want_params, want_losses = _single_process_equivalent(world)
    steps = (temporal.num_windows(T, B) + world - 1) // world * EPOCHS
    for r in range(world):
        assert res[r]["calls"] == steps                    # ONE all-reduce per optimizer step
        np.testing.assert_allclose(res[r]["losses"].numpy(), torch.stack(want_losses[r]).numpy(), rtol=1e-5, atol=1e-7)
        for got, want in zip(res[r]["params"], want_params):
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    for a, b in zip(res[0]["params"], res[1]["params"]):   # replicas stay in lock step
        assert torch.equal(a, b)
