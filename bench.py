#!/usr/bin/env python3
"""Benchmark of the Seastar hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W [--workload gcn|tgcn]

N = 1 (default workload "gcn"): BASELINE.json configs[1] -- 2-layer GCN
128->128->128 on a synthetic CSR with |V| = 1M, |E| = 16M (uniform, seed 1).  One
step = one training epoch (forward, cross-entropy on the train mask, backward,
Adam), inputs resident in HBM.  `value` = edges*feat/s = (aggregation launches per
step x E x F) / wall time, whole job.  N > 1 runs N independent replicas of that
workload (single-graph GCN does not shard: SURVEY.md 8(e) "replicas only").

Every line also carries a "tgcn" object: BASELINE.json configs[3] (static-temporal
TGCN, |V| = 50K, |E| = 500K, T = 1000, feat 32, hidden 64, backprop_every 25) with
its BPTT windows sharded over the N ranks and ONE RCCL all-reduce of the flattened
gradient bucket per optimizer step -- the path the north star scales to 8 GPUs.

"roofline": dominant kernel gcn_agg, algorithmic bytes per launch
over its mean launch time measured with HIP events on the launch stream inside the
timed region.  "cpu_baseline": the C oracle (OpenMP port) timed on this host's
cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


class GCN(nn.Module):
    """benchmarking/gcn/seastar/model.py:4-27 (input layer, n_layers-1 hidden, output layer)."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation):
        super().__init__()
        from stgraph_amd.nn.pytorch.static.gcn_conv import GCNConv
        self.layers = nn.ModuleList()
        self.layers.append(GCNConv(in_feats, n_hidden, activation))
        for _ in range(n_layers - 1):
            self.layers.append(GCNConv(n_hidden, n_hidden, activation))
        self.layers.append(GCNConv(n_hidden, n_classes, None))

    def forward(self, g, features):
        h = features
        for layer in self.layers:
            h = layer(g, h)
        return h


def synthetic_graph(n, e, seed, device):
    """Uniform random directed edges (duplicates removed on the device), as (src, dst) int32."""
    gen = torch.Generator(device=device).manual_seed(seed)
    m = int(e * 1.02) + 1024
    key = torch.randint(0, n * n, (m,), generator=gen, device=device, dtype=torch.int64)
    key = torch.unique(key)
    key = key[torch.randperm(key.shape[0], generator=gen, device=device)][:e]
    assert key.shape[0] == e, "increase the oversampling factor"
    return (key // n).to(torch.int32), (key % n).to(torch.int32)


def gcn_setup(device, seed, n=1_000_000, e=16_000_000, feat=128):
    from stgraph_amd.graph import StaticGraph
    src, dst = synthetic_graph(n, e, seed, device)
    g = StaticGraph((src, dst), None, n, device=device, sort_inplace=False)
    deg = g.csr("fwd").row_offset[1:] - g.csr("fwd").row_offset[:-1]
    norm = torch.pow(deg.float(), -0.5)
    norm[torch.isinf(norm)] = 0
    g.set_ndata("norm", norm.unsqueeze(1))
    gen = torch.Generator(device=device).manual_seed(seed + 100)
    x = torch.randn(n, feat, device=device, generator=gen)
    labels = torch.randint(0, feat, (n,), device=device, generator=gen)
    train_mask = torch.zeros(n, dtype=torch.bool, device=device)
    train_mask[: int(0.6 * n)] = True                      # benchmarking/gcn/seastar/utils.py:25-27
    torch.manual_seed(seed)
    model = GCN(feat, feat, feat, 1, F.relu).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4)
    loss_fn = nn.CrossEntropyLoss()
    train_idx = train_mask.nonzero().squeeze(1)

    def step():
        model.train()
        logits = model(g, x)
        loss = loss_fn(logits[train_idx], labels[train_idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    meta = dict(n=n, e=e, feat=feat, agg_launches_per_step=4, graph=g, norm=norm, x=x)
    return step, meta


def cpu_baseline_gcn(meta, budget_s=20.0):
    """Oracle (C/OpenMP port of the emitted kernel) on this host: full-size aggregation launches
    of the same graph/features until ~budget_s of CPU time has been spent (at least one)."""
    from oracle import stg_oracle as orc
    g = meta["graph"]
    f = g.csr("fwd")
    csr = orc.OracleCSR(f.row_offset.cpu().numpy(), f.column_indices.cpu().numpy(), f.eids.cpu().numpy(),
                        f.node_ids.cpu().numpy(), None, None, None)
    x = meta["x"].cpu().numpy()
    norm = meta["norm"].cpu().numpy().reshape(-1, 1)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    t0 = time.time()
    launches = 0
    while True:
        orc.gcn_agg(x, norm, norm, csr, omp=True)
        launches += 1
        if time.time() - t0 > budget_s or launches >= 8:
            break
    dt = time.time() - t0
    return {"value": launches * meta["e"] * meta["feat"] / dt, "unit": "edges*feat/s", "cores": cores,
            "kind": "port",
            "sample": f"{launches} full-size gcn_agg launches (|V|={meta['n']}, |E|={meta['e']}, F={meta['feat']}) "
                      f"by oracle/stg_oracle.c with OpenMP over rows, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gcn", choices=["gcn"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--feat", type=int, default=128)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from stgraph_amd import kernels

    step, meta = gcn_setup(device, seed=1 + rank, n=args.nodes, e=args.edges, feat=args.feat)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    records = []
    kernels.enable_launch_timing(records)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kernels.enable_launch_timing(None)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ef_per_step = meta["agg_launches_per_step"] * meta["e"] * meta["feat"]
    value = world * args.steps * ef_per_step / dt
    ms = [a.elapsed_time(b) for (_, a, b, _, _) in records]
    bytes_alg = records[0][3]
    mean_ms = float(np.mean(ms))
    achieved = bytes_alg / (mean_ms * 1e-3) / 1e9
    line = {
        "metric": "edges*feat/s (GCN epoch throughput)", "value": value, "unit": "edges*feat/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "epochs_per_s": args.steps / dt,
        "config": {"workload": f"2-layer GCN {args.feat}->{args.feat}->{args.feat}, synthetic CSR "
                               f"|V|={meta['n']} |E|={meta['e']} (BASELINE configs[1]), fwd+CE+bwd+Adam per step",
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas",
                   "agg_launches_per_step": meta["agg_launches_per_step"],
                   "reference_compat_D1": kernels.reference_compat()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "gcn_agg_kernel",
                     "algorithmic_bytes_per_launch": bytes_alg, "mean_launch_ms": mean_ms,
                     "launches_timed": len(ms)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_gcn(meta)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
