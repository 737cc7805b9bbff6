#!/usr/bin/env python3
"""Benchmark of the Seastar hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W

Output: the LAST stdout line is one JSON object of <= 6 KB (``final_line``): the contract's keys + ``roofline`` + ``cpu_baseline`` of
the headline and a short summary per BASELINE config.  Everything long (per-kernel tables, byte models, sample descriptions) is the
full record in ``bench_detail.json`` next to this script, also printed on an EARLIER stdout line prefixed ``# bench_detail ``.

N = 1 headline: BASELINE.json configs[1] -- 2-layer GCN 128->128->128 on a synthetic CSR with |V| = 1M, |E| = 16M (uniform,
duplicate-free, seed 1).  One step = one training epoch (forward, cross-entropy on the train mask, backward, Adam), inputs
resident in HBM.  `value` = edges*feat/s = sum of E x F over the aggregation launches EXECUTED in the timed region / wall time
(SURVEY.md 8(d)); the step executes 3 aggregations (the first layer's input carries no gradient, nn/functional._InputLayer),
`value_reference_formulation` counts the 4 of the reference's order.  "roofline": dominant kernel gcn_agg -- algorithmic bytes per
launch over its mean launch time (HIP events on the launch stream inside the timed region); "traffic" = HBM bytes per launch from
rocprofv3 --pmc child passes (tools/pmc_gcn.py; the committed passes under profiles/ if a child run fails);
"roofline.north_star" = the aggregation at the Cora widths on 1024 replicas of the Cora-shaped graph (the >= 60 % target).
"cpu_baseline": the same epoch in plain torch on the host (SURVEY.md 8(d) variant (i)), bounded sample, rank 0 at N = 1 only.
The other configs ride along as objects: "cora" (configs[0]), "gat" (configs[2]: the default uniform-attention form and
`general_form` = the emitted K0/K1/K2 at full width), "tgcn" (configs[3]), "dynamic" (configs[4]); each has its own achieved
`roofline.frac`, `edges_feat_per_s` counted on executed launches, and `cpu_baseline`.

N > 1 headline: BASELINE.json configs[3], the path that shards: one step = one training epoch of the static-temporal
TGCN (T = 1000, 40 BPTT windows dealt round-robin to the ranks, ONE RCCL all-reduce of the 133 KB gradient bucket per optimizer
step); `value` = whole-job epochs/s, "scaling": "strong".  cfg5 (windows sharded) and N independent cfg2 replicas ("gcn_replicas",
weak) are sub-objects.  `python3 bench.py --gpus N` without a launcher starts its own N ranks as child processes
(torch.distributed.run) before touching the GPU; under the driver's torch.distributed.run form it reads RANK / WORLD_SIZE.
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
FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (f32 in, exact f32): same guide, Matrix cores table

FINAL_LINE_MAX_BYTES = 6144     # the driver reads the LAST stdout line; everything long goes to bench_detail.json
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
CPU_THREADS_RULE = ("cfg2/cfg3 (few large ops): every usable logical CPU; cfg4/cfg5 (thousands of small ops): min(32, usable); "
                    "cfg1 (2708 vertices): faster of {16, all}")


def _sig(x, digits=6):
    """Floats to ``digits`` significant digits (a machine-read line carries no 17-digit noise); NaN / inf -> None."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _clip(x, limit, by_key=None, key=None):
    """Every string of a nested record cut to ``limit`` characters (``by_key``: per-key limits): a machine-read line is no place for prose."""
    if isinstance(x, str):
        n = (by_key or {}).get(key, limit)
        return x if len(x) <= n else x[:n - 3] + "..."
    if isinstance(x, dict):
        return {k: _clip(v, limit, by_key, k) for k, v in x.items()}
    if isinstance(x, list):
        return [_clip(v, limit, by_key, key) for v in x]
    return x


def _pick(d, *keys):
    """Sub-dict of the keys present (and not None) in ``d``; {} for a missing object."""
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _config_summary(obj, extra=()):
    """<= 400 bytes per config: value, unit, achieved roofline fraction + its kernel, the CPU baseline's value."""
    if not isinstance(obj, dict):
        return None
    roof, cpu = obj.get("roofline") or {}, obj.get("cpu_baseline") or {}
    out = {"metric": obj.get("metric"), "value": obj.get("value"),
           "edges_feat_per_s": obj.get("edges_feat_per_s"),
           "roofline": _pick(roof, "bound", "frac", "kernel"),
           "cpu_baseline": _pick(cpu, "value", "cores")}
    if roof.get("x1024_frac") is not None:          # cfg1: the single 2708-vertex graph is launch bound; its bandwidth-bound variant
        out["roofline"]["x1024_frac"] = roof["x1024_frac"]
    if isinstance(out["roofline"].get("kernel"), str):
        out["roofline"]["kernel"] = out["roofline"]["kernel"][:64]
    for k in extra:
        if obj.get(k) is not None:
            out[k] = obj[k]
    return {k: v for k, v in out.items() if v not in (None, {})}


def final_line(detail):
    """The ONE machine-read stdout line (contract: task statement; <= FINAL_LINE_MAX_BYTES) from the full ``detail``
    record: the contract's keys, ``roofline`` and ``cpu_baseline`` of the headline, and a short summary per config.
    Paragraph-length strings and per-kernel tables stay in bench_detail.json."""
    roof = detail.get("roofline") or {}
    cpu = detail.get("cpu_baseline")
    line = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data")}
    cfg = detail.get("config") or {}
    line["config"] = _pick(cfg, "workload", "parallelism", "aggregations_executed_per_step", "edges_feat_per_step",
                           "windows_per_epoch", "optimizer_steps_per_epoch", "backprop_every")
    line["roofline"] = _pick(roof, "bound", "achieved", "peak", "unit", "frac", "kernel", "algorithmic_bytes_per_launch",
                             "flops_per_launch", "mean_launch_ms", "launches_timed", "gcn_agg_share_of_step", "hbm_frac")
    line["roofline"]["traffic"] = roof.get("traffic")
    if roof.get("north_star"):
        line["roofline"]["north_star"] = _pick(roof["north_star"], "frac", "frac_F16", "frac_F7")
    if cpu:
        host = cpu.get("host") or {}
        line["cpu_baseline"] = _pick(cpu, "value", "unit", "cores", "kind", "seconds_per_epoch")
        line["cpu_baseline"]["sample"] = (cpu.get("sample_short") or cpu.get("sample") or "")[:160]
        line["cpu_baseline"]["host"] = (f"{host.get('sockets')} x {host.get('cpu_model')}, {host.get('physical_cores')} cores / "
                                        f"{host.get('logical_cores')} threads")[:96]
    else:
        line["cpu_baseline"] = None
    for k in ("value_reference_formulation", "epochs_per_s"):
        if detail.get(k) is not None:
            line[k] = detail[k]
    if detail.get("allreduce"):
        line["allreduce"] = _pick(detail["allreduce"], "bytes", "calls_per_epoch", "share_of_epoch", "in_graph")
    if detail.get("process_group"):
        pg = detail["process_group"]
        line["process_group"] = {"world_size": pg.get("world_size"), "backend": pg.get("backend"),
                                 "distinct_devices": pg.get("distinct_devices"),
                                 "devices": [str(r.get("name"))[:24] for r in pg.get("ranks", [])][:8]}
    configs = {}
    extras = {"cora": ("eager_epochs_per_s",), "gat": ("ms_per_epoch", "general_form"),
              "tgcn": ("us_per_snapshot", "n_gpus"), "dynamic": ("csr_build_share", "resident_epochs_per_s", "T"),
              "gcn_replicas": ("ms_per_step", "n_gpus")}
    for name, ex in extras.items():
        summ = _config_summary(detail.get(name), ex)
        if summ:
            configs[name] = summ
    line["configs"] = configs
    line["detail_file"] = os.path.basename(DETAIL_FILE)
    line = _clip(_sig(line), 96, {"workload": 240, "metric": 120, "parallelism": 160, "sample": 160})
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > FINAL_LINE_MAX_BYTES:        # never let a summary cost the measurement: values only, then fail loudly
        line["configs"] = {k: _pick(v, "value") for k, v in line["configs"].items()}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > FINAL_LINE_MAX_BYTES:
        raise RuntimeError(f"final bench line is {len(text)} bytes (limit {FINAL_LINE_MAX_BYTES})")
    return text


def emit(detail):
    """bench_detail.json next to this script, the same record on an EARLIER stdout line, then the final line."""
    full = json.dumps(_sig(detail, 9), allow_nan=False)
    try:
        with open(DETAIL_FILE, "w") as f:
            f.write(full + "\n")
        for extra_dir in (os.path.join(ROOT, "gpurun_out"),):
            if os.path.isdir(extra_dir):
                with open(os.path.join(extra_dir, "bench_detail.json"), "w") as f:
                    f.write(full + "\n")
    except OSError as exc:
        progress(f"could not write {DETAIL_FILE}: {exc}")
    text = final_line(detail)
    print("# bench_detail " + full, flush=True)
    progress(f"line sizes: detail {len(full)} bytes, final {len(text)} bytes")
    print(text, flush=True)


def launch_ranks(gpus, argv):
    """``python3 bench.py --gpus N`` without a launcher (WORLD_SIZE unset, N > 1): start the N ranks as CHILD processes through
    torch.distributed.run -- before this process has touched the GPU, never an exec -- relay their output and return their
    status.  The driver's own ``python -m torch.distributed.run ... bench.py --gpus N`` form sets WORLD_SIZE and never comes here."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    progress(f"--gpus {gpus} without WORLD_SIZE: starting {gpus} ranks: {' '.join(cmd[1:8])} ...")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


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
    """Uniform random duplicate-free directed edges as (src, dst) int32 device tensors."""
    gen = torch.Generator(device=device).manual_seed(seed)
    m = int(e * 1.02) + 1024
    key = torch.randint(0, n * n, (m,), generator=gen, device=device, dtype=torch.int64)
    key = torch.unique(key)
    key = key[torch.randperm(key.shape[0], generator=gen, device=device)][:e]
    assert key.shape[0] == e, "increase the oversampling factor"
    return (key // n).to(torch.int32), (key % n).to(torch.int32)


def degree_norm(g):
    f = g.csr("fwd")
    norm = torch.pow((f.row_offset[1:] - f.row_offset[:-1]).float(), -0.5)
    norm[torch.isinf(norm)] = 0                          # benchmarking/gcn/seastar/train.py:53-57
    return norm.unsqueeze(1)


# ------------------------------------------------------------------------------ GCN (cfg 2)
def gcn_setup(device, seed, n, e, feat):
    from stgraph_amd.graph import StaticGraph
    src, dst = synthetic_graph(n, e, seed, device)
    g = StaticGraph((src, dst), None, n, device=device, sort_inplace=False)
    norm = degree_norm(g)
    g.set_ndata("norm", norm)
    gen = torch.Generator(device=device).manual_seed(seed + 100)
    x = torch.randn(n, feat, device=device, generator=gen)
    labels = torch.randint(0, feat, (n,), device=device, generator=gen)
    ntrain = int(0.6 * n)                                      # train mask = first 60 % (gcn/seastar/utils.py:25-27)
    torch.manual_seed(seed)
    model = GCN(feat, feat, feat, 1, F.relu).to(device)
    # torch.optim.Adam as the reference script constructs it (benchmarking/gcn/seastar/train.py:63-66), in torch's single-launch
    # implementation on the GPU (fused=True: the same update rule; the default foreach form is eight multi-tensor launches a step)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4, fused=device.type == "cuda")
    # Same loss as the reference's nn.CrossEntropyLoss() on logits[train_mask] (mean over the masked rows).
    # Written as a slice (the mask is a prefix: a view, no gather/scatter); the loss itself is the library's fused
    # softmax cross-entropy (csrc/xent.hip; torch's fused 'mean' reduction runs single-block nll_loss kernels on
    # ROCm: 1.3 + 1.0 ms for 600K rows, profiles/r01_bench_gcn_cfg2_kernel_stats.csv).
    from stgraph_amd.nn import functional as SF
    loss_fn = SF.cross_entropy

    def step():
        model.train()
        logits = model(g, x)
        loss = loss_fn(logits, labels, ntrain)            # = nn.CrossEntropyLoss()(logits[:ntrain], labels[:ntrain])
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    return step, dict(n=n, e=e, feat=feat, agg_launches_per_step=4, graph=g, norm=norm, x=x, labels=labels)


def cpu_baseline_gcn(meta, budget_s=20.0):
    """Oracle (C/OpenMP port of the emitted kernel) on this host: full-size aggregation launches
    of the same graph/features until ~budget_s of wall time has been spent (at least one)."""
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    from oracle import stg_oracle as orc
    f = meta["graph"].csr("fwd")
    csr = orc.OracleCSR(f.row_offset.cpu().numpy(), f.column_indices.cpu().numpy(), f.eids.cpu().numpy(),
                        f.node_ids.cpu().numpy(), None, None, None)
    x = meta["x"].cpu().numpy()
    norm = meta["norm"].cpu().numpy().reshape(-1, 1)
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
            "sample": f"{launches} full-size gcn_agg launch(es) (|V|={meta['n']}, |E|={meta['e']}, F={meta['feat']}) "
                      f"by oracle/stg_oracle.c, OpenMP over rows, {dt:.1f} s wall"}


def torch_cpu_gcn_epochs(row_off, col, norm, x, labels, ntrain, widths, epochs, warmup, budget_s, seed=0, threads=None):
    """SURVEY.md 8(d) CPU variant (i): the reference model's epoch in plain torch on the host -- A_hat as a
    torch.sparse_csr_tensor (values norm[row] * norm[col]), each layer ``act(A_hat (h W) + b)``, cross-entropy on the
    first ``ntrain`` rows, backward, Adam(1e-2, wd 5e-4).  Returns (mean seconds per epoch, epochs timed, threads)."""
    threads = threads or cpu_threads()
    torch.set_num_threads(threads)
    import warnings
    row_off, col = row_off.cpu().long(), col.cpu().long()
    nrm = norm.cpu().reshape(-1)
    n = nrm.shape[0]
    rows = torch.repeat_interleave(torch.arange(n), row_off[1:] - row_off[:-1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        A = torch.sparse_csr_tensor(row_off, col, nrm[rows] * nrm[col], size=(n, n))
    x, labels = x.cpu(), labels.cpu()
    torch.manual_seed(seed)
    Ws, bs = [], []
    for fi, fo in zip(widths[:-1], widths[1:]):
        w = torch.empty(fi, fo)
        torch.nn.init.xavier_uniform_(w)
        Ws.append(w.requires_grad_(True))
        bs.append(torch.zeros(fo, requires_grad=True))
    opt = torch.optim.Adam(Ws + bs, lr=1e-2, weight_decay=5e-4)
    dur = []
    t_start = time.time()
    for ep in range(warmup + epochs):
        t0 = time.perf_counter()
        h = x
        for li, (w, b) in enumerate(zip(Ws, bs)):
            h = torch.sparse.mm(A, h @ w) + b
            if li + 1 < len(Ws):
                h = F.relu(h)
        loss = F.cross_entropy(h[:ntrain], labels[:ntrain])
        opt.zero_grad()
        loss.backward()
        opt.step()
        if ep >= warmup:
            dur.append(time.perf_counter() - t0)
        if time.time() - t_start > budget_s and len(dur) >= 1:
            break
    return float(np.mean(dur)), len(dur), threads


def host_description():
    """CPU model, sockets, physical and logical cores of this host (BASELINE.md section 3 asks for all of them)."""
    model, phys, sockets = "unknown", set(), set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                    sockets.add(pid)
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
            if pid is not None and cid is not None:
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"cpu_model": model, "sockets": len(sockets) or None, "physical_cores": len(phys) or None,
            "logical_cores": os.cpu_count(), "logical_cores_usable_by_this_process": usable, "torch": torch.__version__}


def cpu_baseline_epoch_gcn(meta, ntrain, labels, executed_per_step, budget_s=20.0):
    """The main line's workload (cfg2 training epoch) on the host in plain torch; bounded: at most 2 epochs /
    ``budget_s`` (no warm-up epoch: one epoch takes ~20 s on a 256-thread host).  ``value`` uses the SAME aggregation
    count per step as the GPU line's ``value`` (``executed_per_step``: the launches the GPU step executes), so the two
    values are in one unit and their ratio is the ratio of the step times."""
    f = meta["graph"].csr("fwd")
    sec, timed, threads = torch_cpu_gcn_epochs(f.row_offset, f.column_indices, meta["norm"], meta["x"], labels, ntrain,
                                               [meta["feat"]] * 3, epochs=2, warmup=0, budget_s=budget_s)
    return {"value": executed_per_step * meta["e"] * meta["feat"] / sec, "unit": "edges*feat/s",
            "value_is": f"{executed_per_step:g} x E x F per step / CPU seconds per step: the aggregation count of the GPU "
                        "line's `value` (the CPU epoch itself performs the reference formulation's 4 sparse products: "
                        "value_reference_formulation)",
            "value_reference_formulation": meta["agg_launches_per_step"] * meta["e"] * meta["feat"] / sec,
            "cores": threads, "threads_rule": CPU_THREADS_RULE, "kind": "port", "seconds_per_epoch": sec,
            "sample": f"{timed} full training epoch(s) of the same model and graph (|V|={meta['n']}, |E|={meta['e']}, "
                      f"{meta['feat']}->{meta['feat']}->{meta['feat']}) in plain torch on the host: "
                      "torch.sparse_csr_tensor(A_hat) @ dense per layer, cross-entropy, backward, Adam "
                      "(SURVEY.md 8(d) CPU variant (i); the reference has no CPU path of its own)",
            "host": host_description()}


def progress(msg):
    """One line on stderr per section: a long run must keep writing (the GPU box kills a command that is silent for minutes)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_threads(cap=None):
    """Threads for a CPU baseline: the logical CPUs this process may run on (affinity mask), optionally capped -- the temporal
    baselines are thousands of SMALL ops, where more threads than ~32 only add fork/join cost."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(n, cap) if cap else n)


def _cpu_sparse(row_off, col, values, n):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.sparse_csr_tensor(row_off.cpu().long(), col.cpu().long(), values, size=(n, n))


def _cpu_rows(row_off, n):
    ro = row_off.cpu().long()
    return torch.repeat_interleave(torch.arange(n), ro[1:] - ro[:-1])


def _cpu_tgcn_params(feat, hidden, head_out, seed):
    """Random-init parameters of TGCN(feat -> hidden) + Linear(hidden, feat) (+ Linear(feat, head_out)) as plain CPU
    tensors (nn/pytorch/temporal/tgcn.py:9-14, static-temporal-tgcn/seastar/model.py:6-11)."""
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).requires_grad_(True)  # noqa: E731
    p = {"Wc": [mk(feat, hidden) for _ in range(3)], "bc": [mk(hidden) for _ in range(3)],
         "Wg": [mk(hidden, 2 * hidden) for _ in range(3)], "bg": [mk(hidden) for _ in range(3)],
         "W1": mk(feat, hidden), "b1": mk(feat)}
    if head_out:
        p["W2"], p["b2"] = mk(head_out, feat), mk(head_out)
    flat = [t for v in p.values() for t in (v if isinstance(v, list) else [v])]
    return p, flat


def _cpu_tgcn_cell(A, x, H, p):
    """nn/pytorch/temporal/tgcn.py:21-55 in plain torch: three GCNConv gates (A_hat (x W) + b, clamp), the gate Linears, GRU."""
    conv = lambda k: torch.clamp(torch.sparse.mm(A, x @ p["Wc"][k]) + p["bc"][k], -1e6, 1e6)  # noqa: E731
    Z = torch.sigmoid(torch.cat((conv(0), H), 1) @ p["Wg"][0].t() + p["bg"][0])
    R = torch.sigmoid(torch.cat((conv(1), H), 1) @ p["Wg"][1].t() + p["bg"][1])
    Ht = torch.tanh(torch.cat((conv(2), H * R), 1) @ p["Wg"][2].t() + p["bg"][2])
    return Z * H + (1 - Z) * Ht


def _cpu_window_sample(run_window, B, budget_s):
    """Time ``run_window(steps)`` (forward, loss, backward, Adam over ``steps`` snapshots) within ``budget_s``: a 2-snapshot
    probe first, then ONE window of as many snapshots (<= B) as the budget allows.  Returns (seconds per snapshot, snapshots timed)."""
    t0 = time.perf_counter()
    run_window(2)
    probe = (time.perf_counter() - t0) / 2
    steps = int(max(2, min(B, budget_s / max(probe, 1e-6))))
    t0 = time.perf_counter()
    run_window(steps)
    return (time.perf_counter() - t0) / steps, steps


def cpu_baseline_tgcn(g, ew, targets, n, e, T, feat, hidden, B, budget_s=12.0):
    """cfg4's training loop (static-temporal-tgcn/seastar/train.py:160-187) in plain torch on the host: A_hat as ONE
    torch.sparse_csr_tensor with values norm[row] * w[eid] * norm[col]; per window hidden = None, y_hat = randn, the
    snapshots of cell + relu + two Linears + MSE, cost / (B + 1), backward through time, Adam.  Bounded sample (see
    ``_cpu_window_sample``); epochs/s = 1 / (seconds per snapshot x T)."""
    threads = cpu_threads(32)
    torch.set_num_threads(threads)
    f = g.csr("fwd")
    nrm = g.get_ndata("norm").detach().cpu().reshape(-1)
    rows, col = _cpu_rows(f.row_offset, n), f.column_indices.cpu().long()
    w = ew.detach().cpu().reshape(-1)[f.eids.cpu().long()]
    A = _cpu_sparse(f.row_offset, f.column_indices, nrm[rows] * w * nrm[col], n)
    tg = targets.detach()[:B].cpu().reshape(B, n, 1)
    p, flat = _cpu_tgcn_params(feat, hidden, 1, seed=3)
    opt = torch.optim.Adam(flat, lr=1e-2)

    def run_window(steps):
        opt.zero_grad()
        cost, H, y = 0, torch.zeros(n, hidden), torch.randn(n, feat)
        for k in range(steps):
            H = _cpu_tgcn_cell(A, y, H, p)
            y = torch.relu(H) @ p["W1"].t() + p["b1"]
            y_out = y @ p["W2"].t() + p["b2"]
            cost = cost + torch.mean((y_out - tg[k]) ** 2)
        cost = cost / (B + 1)
        cost.backward()
        opt.step()
    sps, steps = _cpu_window_sample(run_window, B, budget_s)
    return {"value": 1.0 / (sps * T), "unit": "epochs/s", "cores": threads, "threads_rule": CPU_THREADS_RULE, "kind": "port",
            "seconds_per_snapshot": sps,
            "sample": f"one BPTT window of {steps} snapshot(s) (of the epoch's {T}: forward, loss, backward through time, Adam) "
                      f"on the same graph (|V|={n}, |E|={e}, edge weights) and model shape in plain torch on the host: "
                      "torch.sparse_csr_tensor(A_hat) @ dense for the three gate convolutions of every snapshot; epochs/s = "
                      "1 / (seconds per snapshot x T) (the reference has no CPU path of its own)",
            "host": host_description()}


def cpu_baseline_dynamic(snaps, pn_edges, pn_targets, n, T, feat, hidden, B, budget_s=12.0):
    """cfg5's loop (dynamic-temporal-tgcn/seastar/train.py:192-254) in plain torch on the host: one un-weighted A_hat per
    snapshot (built up front, outside the timed region, as the reference's NaiveGraph builds its CSRs at construction),
    TGCN cell, relu + Linear, dot-product decoder on the label edges, BCE-with-logits, cost / (B + 1), backward, Adam."""
    threads = cpu_threads(32)
    torch.set_num_threads(threads)
    use = min(B, T - 1)
    As = []
    for t in range(use):
        s_, d_ = snaps[t][0].cpu().long(), snaps[t][1].cpu().long()
        order = torch.argsort(d_ * n + s_)
        s_, d_ = s_[order], d_[order]
        deg = torch.bincount(d_, minlength=n)
        ro = torch.zeros(n + 1, dtype=torch.long)
        ro[1:] = torch.cumsum(deg, 0)
        nrm = deg.float().pow(-0.5)
        nrm[torch.isinf(nrm)] = 0
        As.append(_cpu_sparse(ro, s_, nrm[d_] * nrm[s_], n))
    p, flat = _cpu_tgcn_params(feat, hidden, 0, seed=4)
    opt = torch.optim.Adam(flat, lr=1e-2)
    edges = [x.cpu() for x in pn_edges[:use]]
    tgts = [x.cpu() for x in pn_targets[:use]]

    def run_window(steps):
        opt.zero_grad()
        cost, H, y = 0, torch.zeros(n, hidden), torch.randn(n, feat)
        for t in range(min(steps, use)):
            H = _cpu_tgcn_cell(As[t], y, H, p)
            y = torch.relu(H) @ p["W1"].t() + p["b1"]
            logits = (y[edges[t][0]] * y[edges[t][1]]).sum(-1)
            cost = cost + F.binary_cross_entropy_with_logits(logits, tgts[t])
        cost = cost / (B + 1)
        cost.backward()
        opt.step()
    sps, steps = _cpu_window_sample(run_window, use, budget_s)
    return {"value": 1.0 / (sps * (T - 1)), "unit": "epochs/s", "cores": threads, "threads_rule": CPU_THREADS_RULE, "kind": "port",
            "seconds_per_snapshot": sps,
            "sample": f"one BPTT window of {steps} snapshot(s) of the T = {T} stream (|V|={n}; one un-weighted A_hat per snapshot, "
                      "built outside the timed region) in plain torch on the host: sparse_csr @ dense TGCN cell, link head "
                      "(dot-product decoder + BCE-with-logits), backward, Adam; epochs/s = 1 / (seconds per snapshot x (T - 1))",
            "host": host_description()}


def cpu_baseline_gat(g, feats, labels, ntrain, n, e, fin, H, D, classes, budget_s=15.0, max_epochs=2):
    """cfg3's model epoch (benchmarking/gat/seastar/model.py:4-42, train.py) in plain torch on the host.  The vertex
    function's `emb - max([emb])` is +0, so each layer is fc -> uniform mean over the in-neighbours
    (one sparse_csr @ dense with values 1 / in-degree) with el / er still formed; ELU between the layers, mean over the
    output heads, cross-entropy on the first 60 %, Adam(5e-3, wd 5e-4)."""
    threads = cpu_threads()
    torch.set_num_threads(threads)
    f = g.csr("fwd")
    rows = _cpu_rows(f.row_offset, n)
    deg = (f.row_offset[1:] - f.row_offset[:-1]).cpu().float()
    inv = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg))
    A = _cpu_sparse(f.row_offset, f.column_indices, inv[rows], n)
    x, y = feats.detach().cpu(), labels.cpu()
    gen = torch.Generator().manual_seed(2)
    mk = lambda *s: (torch.randn(*s, generator=gen) * 0.1).requires_grad_(True)  # noqa: E731
    layers = [(mk(H * D, fin), mk(H, D), mk(H, D), H, D), (mk(classes, H * D), mk(1, classes), mk(1, classes), 1, classes)]
    params = [t for l in layers for t in l[:3]]
    opt = torch.optim.Adam(params, lr=5e-3, weight_decay=5e-4)
    dur, t_start = [], time.time()
    for ep in range(max_epochs):
        t0 = time.perf_counter()
        h = x
        for li, (W, al, ar, hh, dd) in enumerate(layers):
            feat = (h @ W.t()).view(n, hh, dd)
            el, er = (feat * al).sum(-1), (feat * ar).sum(-1)                      # formed as the layer forms them
            out = torch.sparse.mm(A, feat.view(n, hh * dd)).view(n, hh, dd) + 0.0 * (el + er).unsqueeze(-1)
            h = F.elu(out).flatten(1) if li == 0 else out.mean(1)
        loss = F.cross_entropy(h[:ntrain], y[:ntrain])
        opt.zero_grad()
        loss.backward()
        opt.step()
        dur.append(time.perf_counter() - t0)
        if time.time() - t_start > budget_s:
            break
    sec = float(np.mean(dur))
    return {"value": 1.0 / sec, "unit": "epochs/s", "cores": threads, "threads_rule": CPU_THREADS_RULE, "kind": "port", "seconds_per_epoch": sec,
            "sample": f"{len(dur)} full training epoch(s) of the same 2-layer model and graph (|V|={n}, |E|={e}, "
                      f"{fin} -> {H} x {D} -> {classes}) in plain torch on the host: fc, mean aggregation as "
                      "torch.sparse_csr_tensor @ dense, ELU, cross-entropy, backward, Adam",
            "host": host_description()}


# ------------------------------------------------------------------------- GCN-Cora (cfg 1)
def cora_shaped(seed=0, n=2708, pairs=5278, max_deg=168):
    """Cora-SHAPED synthetic graph (the real dataset needs the network): Chung-Lu style power-law
    expected degrees capped at max_deg, 5278 undirected pairs mirrored => |E| = 10556, no self loops."""
    rng = np.random.default_rng(seed)
    w = (np.arange(1, n + 1, dtype=np.float64)) ** -0.6
    w = np.minimum(w / w.sum() * 2 * pairs, max_deg)
    p = w / w.sum()
    got = set()
    while len(got) < pairs:
        a = rng.choice(n, size=2 * pairs, p=p)
        b = rng.choice(n, size=2 * pairs, p=p)
        for u, v in zip(a, b):
            if u != v and (min(u, v), max(u, v)) not in got and len(got) < pairs:
                got.add((min(u, v), max(u, v)))
    und = np.array(sorted(got), np.int32)
    return np.concatenate([und[:, 0], und[:, 1]]), np.concatenate([und[:, 1], und[:, 0]])


def cora_run(device, epochs=200, cpu_baseline=False):
    """BASELINE configs[0] shape: 2-layer GCN 1433 -> 16 -> 7, Adam(1e-2, wd 5e-4), cross-entropy on
    the first 60 % (benchmarking/gcn/seastar/train.py:63-101).  Timing rule of the reference: wall
    clock between device syncs around each epoch, epochs 0-2 discarded.  Reported eagerly and with the
    whole epoch (fwd + loss + bwd + Adam) replayed from one HIP graph."""
    from stgraph_amd.capture import CapturedTrainStep
    from stgraph_amd.graph import StaticGraph
    src, dst = cora_shaped()
    n, e = 2708, len(src)
    g = StaticGraph((src, dst), None, n, device=device, sort_inplace=False)
    g.set_ndata("norm", degree_norm(g))
    gen = torch.Generator(device=device).manual_seed(0)
    x = (torch.rand(n, 1433, device=device, generator=gen) < 0.0127).float()
    labels = torch.randint(0, 7, (n,), device=device, generator=gen)
    ntrain = int(0.6 * n)
    from stgraph_amd.nn import functional as SF
    loss_fn = SF.cross_entropy                  # nn.CrossEntropyLoss() as one launch each way (csrc/xent.hip)
    out = {"workload": f"2-layer GCN 1433->16->7 on a Cora-shaped synthetic graph |V|={n} |E|={e} "
                       "(BASELINE configs[0]), Adam, cross-entropy, 200 epochs"}
    for mode in ("eager", "hip_graph"):
        torch.manual_seed(0)
        model = GCN(1433, 16, 7, 1, F.relu).to(device)
        # captured mode: torch's single-kernel Adam (fused=True; same update rule as the default foreach form,
        # which takes 5 multi-tensor launches per step)
        opt = (torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4, capturable=True, fused=True)
               if mode == "hip_graph" else torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4))

        one = torch.ones((), device=device)     # d loss / d loss, kept instead of filled anew by every backward()

        def step():
            logits = model(g, x)
            loss = loss_fn(logits, labels, ntrain)        # = nn.CrossEntropyLoss()(logits[:ntrain], labels[:ntrain])
            opt.zero_grad()
            loss.backward(one)
            opt.step()
            return loss.detach()

        run = step if mode == "eager" else CapturedTrainStep(step, opt, list(model.parameters()))
        dur = []
        for ep in range(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = run()
            torch.cuda.synchronize()
            if ep >= 3:
                dur.append(time.perf_counter() - t0)
        out[mode] = {"epochs_per_s": 1.0 / float(np.mean(dur)), "ms_per_epoch": float(np.mean(dur)) * 1e3,
                     "edges_feat_per_s": 2 * e * (16 + 7) / float(np.mean(dur)), "final_loss": float(loss)}
    out["metric"], out["value"] = "epochs/s", out["hip_graph"]["epochs_per_s"]
    out["roofline_x1024"] = cora_roofline(device, src, dst, n)
    # the single 2708-vertex graph is launch-latency bound: its own fraction, for the record
    per_epoch_bytes = 2 * (kernels_bytes(n, e, 16) + kernels_bytes(n, e, 7))
    out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                       "bytes_model": "4 gcn_agg launches per epoch (F = 16 and F = 7, forward + backward), SURVEY.md 8(d) bytes",
                       "achieved": per_epoch_bytes / (out["hip_graph"]["ms_per_epoch"] * 1e-3) / 1e9,
                       "frac": per_epoch_bytes / (out["hip_graph"]["ms_per_epoch"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "whole captured epoch (GEMMs, loss, Adam included) against the aggregation bytes only: "
                               "launch bound; the bandwidth-bound variant of the same shape is roofline_x1024",
                       "x1024_frac": out["roofline_x1024"]["frac"]}
    if cpu_baseline:
        f = g.csr("fwd")
        # a 2708-vertex epoch is a few ms of work: with every logical core in the pool torch spends its time waking
        # threads, so the pool size is part of the baseline -- both settings are timed, the faster one is the value
        tried = {}
        for th in sorted({min(16, os.cpu_count() or 1), os.cpu_count() or 1}):
            sec, timed, _ = torch_cpu_gcn_epochs(f.row_offset, f.column_indices, g.get_ndata("norm"), x, labels, ntrain,
                                                 [1433, 16, 7], epochs=100, warmup=3, budget_s=7.0, threads=th)
            tried[th] = (sec, timed)
        threads = min(tried, key=lambda k: tried[k][0])
        sec, timed = tried[threads]
        out["cpu_baseline"] = {"value": 1.0 / sec, "unit": "epochs/s", "cores": threads, "threads_rule": CPU_THREADS_RULE, "kind": "port",
                               "edges_feat_per_s": 2 * e * (16 + 7) / sec, "ms_per_epoch": sec * 1e3,
                               "ms_per_epoch_by_threads": {str(k): v[0] * 1e3 for k, v in tried.items()},
                               "sample": f"{timed} epochs of the same model on the same graph in plain torch on the host "
                                         "(torch.sparse_csr_tensor(A_hat) @ dense, cross-entropy, backward, Adam): BASELINE "
                                         "configs[0] 'CPU PyTorch reference path'; baseline, not target",
                               "host": host_description()}
    return out


def kernels_bytes(n, e, f, ew=False):
    from stgraph_amd import kernels
    return kernels.gcn_agg_algorithmic_bytes(n, e, f, ew)


def cora_roofline(device, src, dst, n, K=1024, iters=40):
    """SURVEY.md 8(d) "Cora x K": K disjoint replicas of the Cora-shaped graph (block diagonal, |V| = 2.77M,
    |E| = 10.8M), so that the aggregation at the layer widths of the Cora model (16, 7) is bandwidth- instead of
    launch-bound.  Forward + backward launch per width, HIP events, algorithmic bytes of SURVEY.md 8(d)."""
    from stgraph_amd import kernels
    big_src = np.concatenate([src + k * n for k in range(K)]).astype(np.int32)
    big_dst = np.concatenate([dst + k * n for k in range(K)]).astype(np.int32)
    N, E = n * K, len(big_src)
    g = kernels.build_graph_csr(big_src, big_dst, N, device)
    norm = torch.rand(N, 1, device=device) + 0.5
    res = {"workload": f"gcn_agg forward + backward launch on {K} disjoint replicas of the Cora-shaped graph "
                       f"(|V|={N}, |E|={E})"}
    tot_b, tot_ms = 0, 0.0
    for F_ in (16, 7):
        x = torch.randn(N, F_, device=device)
        nbytes = kernels.gcn_agg_algorithmic_bytes(N, E, F_, False)
        ms = []
        for csr in (g.fwd, g.bwd):
            for _ in range(5):
                kernels.gcn_agg(x, norm, norm, csr)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                kernels.gcn_agg(x, norm, norm, csr)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b) / iters)
        gbps = 2 * nbytes / (sum(ms) * 1e-3) / 1e9
        res[f"F{F_}"] = {"fwd_ms": ms[0], "bwd_ms": ms[1], "algorithmic_bytes_per_launch": nbytes,
                         "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS}
        tot_b += 2 * nbytes
        tot_ms += sum(ms)
    # the Cora model's four aggregation launches per epoch (F = 16 and F = 7, forward and backward) together
    res["achieved"] = tot_b / (tot_ms * 1e-3) / 1e9
    res["frac"] = res["achieved"] / HBM_PEAK_GBS
    res["bound"], res["unit"], res["peak"] = "hbm", "GB/s", HBM_PEAK_GBS
    del g, x, norm
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------ GAT (cfg 3)
class GAT(nn.Module):
    """benchmarking/gat/seastar/model.py:4-42 (hidden GATConv layers flattened over heads, output layer averaged)."""

    def __init__(self, g, num_layers, in_dim, num_hidden, num_classes, heads, activation, feat_drop=0., attn_drop=0.,
                 negative_slope=0.2):
        super().__init__()
        from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
        self.g = g
        self.num_layers = num_layers
        self.gat_layers = nn.ModuleList()
        self.gat_layers.append(GATConv(in_dim, num_hidden, heads[0], feat_drop, attn_drop, negative_slope, activation))
        for l in range(1, num_layers):
            self.gat_layers.append(GATConv(num_hidden * heads[l - 1], num_hidden, heads[l], feat_drop, attn_drop,
                                           negative_slope, activation))
        self.gat_layers.append(GATConv(num_hidden * heads[-2], num_classes, heads[-1], feat_drop, attn_drop,
                                       negative_slope, None))

    def forward(self, inputs):
        h = inputs
        for l in range(self.num_layers):
            h = self.gat_layers[l](self.g, h).flatten(1)
        return self.gat_layers[-1](self.g, h).mean(1)


# launches whose record's `units` field is the edges*feat the launch actually gathers (sum of E x F_launch)
# launches priced against the fp32 matrix rate (they run v_mfma_f32_*_f32).  The row products and the weight-gradient contractions run
# as 3-term bf16 splits on the bf16 instruction since rounds 4 / 5 and are bound by their operand streams: priced on bytes.
TGCN_DENSE = ("tgcn_step_fwd", "tgcn_step_bwd")
AGG_LAUNCHES = ("gcn_agg", "gcn_agg_transform", "gat_k1", "gat_k1_uniform", "gat_bwd", "gat_bwd_uniform")
GAT_DENSE = ("gat_bwd_gw",)            # (gat_fc / gat_fc_out run in the bf16 split form since round 5: bound by their stores)


def executed_edges_feat(records):
    """Sum of E x F over the aggregation launches in ``records`` (kernels.enable_launch_timing tuples)."""
    return float(sum(r[4] for r in records if r[0] in AGG_LAUNCHES))


def kernel_table(records, iters, dense):
    """Per launch name: launches per iteration, mean time, the bytes it moves / its flops (summed over the launches, so that
    one name covering two shapes is priced on what all of them did), the fraction of the roofline that bounds it."""
    tab = {}
    for name, a, b, nbytes, units in records:
        d = tab.setdefault(name, {"ms": 0.0, "bytes": 0.0, "units": 0.0, "n": 0})
        d["ms"] += a.elapsed_time(b)
        d["bytes"] += nbytes
        d["units"] += units
        d["n"] += 1
    out = {}
    for k, v in tab.items():
        ent = {"launches_per_iter": v["n"] / iters, "mean_ms": v["ms"] / v["n"], "bytes": v["bytes"] / v["n"]}
        if k in dense:
            tf = v["units"] / v["ms"] / 1e9
            ent.update({"bound": "mfma", "flops": v["units"] / v["n"], "achieved_TFLOPs": tf, "peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS,
                        "frac": tf / FP32_MFMA_PEAK_TFLOPS, "hbm_frac": v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS})
        else:
            gb = v["bytes"] / v["ms"] / 1e6
            ent.update({"bound": "hbm", "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS})
        out[k] = ent
    return out


def gat_run(device, n=256_000, e=8_000_000, fin=64, H=8, D=64, classes=16, layer_iters=10, epochs=23, cpu_baseline=False):
    """BASELINE configs[2]: (a) one GATConv(64 -> 8 x 64) forward + backward with per-kernel HIP-event times, each
    against the bytes it moves; (b) the 2-layer model of benchmarking/gat/seastar/model.py
    (GATConv(64, 64, 8 heads, elu) -> GATConv(512, classes, 1 head), mean over heads), cross-entropy on the first 60 %,
    Adam(5e-3, wd 5e-4) as gat/seastar/train.py runs it: epochs/s by the reference's timing rule.  Both are measured twice:
    in the default form (uniform attention: the vertex function's scores are identically +0, SURVEY.md D2, so K1 / K2 run at
    the input width) and as ``general_form`` -- the emitted K0 / K1 / K2 as hand-written units at width H x D, what any
    non-degenerate edge softmax runs (kernels.set_gat_uniform_form(False), set_gat_uniform_backward(False))."""
    from stgraph_amd import kernels
    from stgraph_amd.capture import CapturedTrainStep
    from stgraph_amd.graph import StaticGraph
    from stgraph_amd.nn import functional as SF
    from stgraph_amd.nn.pytorch.static.gat_conv import GATConv
    src, dst = synthetic_graph(n, e, 2, device)
    g = StaticGraph((src, dst), None, n, device=device, sort_inplace=False)
    gen = torch.Generator(device=device).manual_seed(2)
    xl = torch.randn(n, fin, device=device, generator=gen)
    R = torch.randn(n, H, D, device=device, generator=gen)
    feats = torch.randn(n, fin, device=device, generator=gen)
    labels = torch.randint(0, classes, (n,), device=device, generator=gen)
    ntrain = int(0.6 * n)
    emitted = kernels.gat_algorithmic_bytes(n, e, H, D)

    def measure_layer():
        """forward + backward of the LAYER: the upstream gradient R is handed to backward() as it is (a `(out * R).sum()`
        loss costs three more passes over [N, H, D] each way that are not the layer's)."""
        torch.manual_seed(2)
        conv = GATConv(fin, D, H).to(device)
        x = xl.clone().requires_grad_(True)
        for _ in range(3):
            conv(g, x).backward(R)
        rec = []
        kernels.enable_launch_timing(rec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(layer_iters):
            conv.zero_grad()
            x.grad = None
            conv(g, x).backward(R)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / layer_iters
        kernels.enable_launch_timing(None)
        ktab = kernel_table(rec, layer_iters, GAT_DENSE)
        if "gat_k0" in ktab and ktab["gat_k0"]["frac"] > 1.0:
            # finite scores: A == 1.0f, S == in-degree, so K0 writes S from the row offsets and visits no edge: priced on what it moves
            moved = 4 * n * H + 8 * n
            ktab["gat_k0"].update({"bytes": moved, "achieved_GBps": moved / ktab["gat_k0"]["mean_ms"] / 1e6,
                                   "frac": moved / ktab["gat_k0"]["mean_ms"] / 1e6 / HBM_PEAK_GBS})
        if "gat_bwd" in ktab:   # the factored backward does not perform K2's second E*H*D gather: priced on what it moves
            moved = 4 * e * H * D + 12 * e * H + 16 * n * H * D
            ktab["gat_bwd"].update({"bytes": moved, "achieved_GBps": moved / ktab["gat_bwd"]["mean_ms"] / 1e6,
                                    "frac": moved / ktab["gat_bwd"]["mean_ms"] / 1e6 / HBM_PEAK_GBS,
                                    "emitted_unit_bytes_SURVEY_8d": emitted["gat_bwd"]})
        if "gat_k1_uniform" in ktab:
            ktab["gat_k1_uniform"]["emitted_unit_bytes_SURVEY_8d"] = emitted["gat_k1"]
        hbm = {k: v for k, v in ktab.items() if v["bound"] == "hbm"}
        dom = max(hbm, key=lambda k: hbm[k]["mean_ms"] * hbm[k]["launches_per_iter"])
        ef = executed_edges_feat(rec) / layer_iters
        del conv, x
        torch.cuda.empty_cache()
        return {"ms_per_fwd_bwd": dt * 1e3, "edges_feat_per_s": ef / dt, "edges_feat_executed_per_fwd_bwd": ef,
                "edges_feat_per_s_reference_formulation": 2 * e * H * D / dt, "dominant_kernel": dom,
                "moved_bytes_frac_of_hbm_peak": sum(v["bytes"] * v["launches_per_iter"] for v in ktab.values()) / dt / 1e9 / HBM_PEAK_GBS,
                "kernels": ktab}

    def measure_model(modes_wanted):
        modes = {}
        ef_epoch = None
        for mode in modes_wanted:
            torch.manual_seed(2)
            model = GAT(g, 1, fin, D, classes, [H, 1], F.elu).to(device)
            # gat/seastar/train.py defaults; captured mode: the same rule as torch's single-kernel capturable Adam
            opt = (torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True, fused=True)
                   if mode == "hip_graph" else torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4))

            def step():
                model.train()
                logits = model(feats)
                loss = SF.cross_entropy(logits, labels, ntrain)    # = CrossEntropyLoss()(logits[train_mask], labels[train_mask])
                opt.zero_grad()                                    # (gat/seastar/train.py: optimizer.zero_grad(); today's default drops the grads)
                loss.backward()
                opt.step()
                return loss.detach()

            if ef_epoch is None:            # the aggregation launches of ONE epoch, from an eager pass with launch records
                rec = []
                kernels.enable_launch_timing(rec)
                step()
                torch.cuda.synchronize()
                kernels.enable_launch_timing(None)
                ef_epoch = executed_edges_feat(rec)
                del rec
            run = step if mode == "eager" else CapturedTrainStep(step, opt, list(model.parameters()))
            dur = []
            for ep in range(epochs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss = run()
                torch.cuda.synchronize()
                if ep >= 3:
                    dur.append(time.perf_counter() - t0)
            modes[mode] = {"epochs_per_s": 1.0 / float(np.mean(dur)), "ms_per_epoch": float(np.mean(dur)) * 1e3,
                           "final_loss": float(loss), "epochs_timed": len(dur)}
            del model, opt, run
            torch.cuda.empty_cache()
        return modes, ef_epoch

    layer = measure_layer()
    modes, ef_epoch = measure_model(("eager", "hip_graph"))
    sec = modes["hip_graph"]["ms_per_epoch"] * 1e-3
    # the same layer and model with the emitted units at full width (edge-softmax fused aggregation as BASELINE configs[2] names it)
    progress("cfg3: GAT, general form (K0 / K1 / K2 at width H x D)")
    kernels.set_gat_uniform_form(False)
    kernels.set_gat_uniform_backward(False)
    try:
        g_layer = measure_layer()
        g_modes, g_ef_epoch = measure_model(("hip_graph",))
    finally:
        kernels.set_gat_uniform_form(True)
        kernels.set_gat_uniform_backward(True)
    g_sec = g_modes["hip_graph"]["ms_per_epoch"] * 1e-3
    g_dom = g_layer["kernels"][g_layer["dominant_kernel"]]
    general = {"what": "set_gat_uniform_form(False) + set_gat_uniform_backward(False): K0 / K1 / K2 as hand-written units at width H x D",
               "metric": "epochs/s", "value": 1.0 / g_sec, "ms_per_epoch": g_sec * 1e3,
               "edges_feat_per_s": g_ef_epoch / g_sec, "layer_ms_per_fwd_bwd": g_layer["ms_per_fwd_bwd"],
               "dominant_kernel": g_layer["dominant_kernel"], "dominant_kernel_frac": g_dom["frac"],
               "dominant_kernel_mean_ms": g_dom["mean_ms"], "layer": g_layer}
    if cpu_baseline:
        progress("cfg3: CPU baseline")
    cpu = cpu_baseline_gat(g, feats, labels, ntrain, n, e, fin, H, D, classes) if cpu_baseline else None
    k1 = layer["kernels"][layer["dominant_kernel"]]
    return {"cpu_baseline": cpu,
            "workload": f"GAT |V|={n} |E|={e} in={fin} heads={H} D={D} negative_slope=0.2 (BASELINE configs[2]); "
                        f"layer = GATConv({fin}, {D}, {H}) forward + backward; model = GATConv({fin},{D},{H},elu) -> "
                        f"GATConv({H * D},{classes},1), cross-entropy, Adam (benchmarking/gat/seastar)",
            "metric": "epochs/s", "value": 1.0 / sec, "ms_per_epoch": sec * 1e3, "epochs_timed": modes["hip_graph"]["epochs_timed"],
            "value_is": "the whole epoch (forward + loss + backward + Adam) replayed from one HIP graph; uniform-attention form "
                        "(identical results: scores are +0, SURVEY.md D2); general_form = the emitted units at full width",
            "eager": modes["eager"], "hip_graph": modes["hip_graph"],
            "edges_feat_per_s": ef_epoch / sec, "edges_feat_executed_per_epoch": ef_epoch,
            "edges_feat_per_s_reference_formulation": 2 * e * (H * D + classes) / sec,
            "final_loss": modes["hip_graph"]["final_loss"],
            "general_form": {k: v for k, v in general.items() if k != "layer"}, "general_form_layer": g_layer,
            "layer": layer,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "kernel": layer["dominant_kernel"],
                         "kernel_is": "the layer's dominant launch by time, priced on the bytes it moves (layer.kernels has every one)",
                         "achieved": k1.get("achieved_GBps"), "frac": k1.get("frac"),
                         "algorithmic_bytes_per_launch": k1.get("bytes"), "mean_launch_ms": k1.get("mean_ms"),
                         "layer_frac_on_moved_bytes": layer["moved_bytes_frac_of_hbm_peak"],
                         "general_form": {"kernel": g_layer["dominant_kernel"], "frac": g_dom["frac"], "mean_launch_ms": g_dom["mean_ms"]}}}


# ----------------------------------------------------------------------------- TGCN (cfg 4)
def process_group_info(device, rank, world, require_distinct=True):
    """What the process group actually consists of (for auditing a SCALE record): backend, size, and per rank the
    device it computes on.  With N > 1 ranks the group must have exactly N members, and (``require_distinct``: always,
    except under the testing-only --share-device) N distinct devices -- a SCALE line of N ranks on fewer GPUs is refused."""
    info = {"world_size": world, "backend": None, "ranks": [{"rank": rank, "device": str(device),
                                                              "name": torch.cuda.get_device_name(device)}]}
    if world > 1:
        info["backend"] = dist.get_backend()
        info["world_size"] = dist.get_world_size()
        mine = {"rank": rank, "device": str(device), "name": torch.cuda.get_device_name(device),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None),
                "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", ""))}
        got = [None] * world
        dist.all_gather_object(got, mine)
        info["ranks"] = got
        info["distinct_devices"] = len({(r["uuid"] or r["device"]) for r in got})
        if info["world_size"] != world:
            raise SystemExit(f"process group has {info['world_size']} ranks, --gpus says {world}")
        if require_distinct and info["distinct_devices"] != world:
            raise SystemExit(f"{world} ranks on {info['distinct_devices']} distinct device(s): every rank needs its own GPU "
                             "(--share-device is the testing-only exception)")
    return info


def tgcn_run(device, rank, world, epochs, warmup_epochs, n, e, T, feat, hidden, B, allreduce_in_graph=False,
             cpu_baseline=False, share_device=False):
    from stgraph_amd import kernels, temporal
    from stgraph_amd.graph import StaticGraph
    src, dst = synthetic_graph(n, e, 3, device)                 # same graph on every rank
    g = StaticGraph((src, dst), None, n, device=device, sort_inplace=False)
    g.set_ndata("norm", degree_norm(g))
    gen = torch.Generator(device=device).manual_seed(3)
    ew = torch.rand(e, 1, device=device, generator=gen) + 0.5    # U(0.5, 1.5), indexed by eid
    targets = torch.randn(T, n, 1, device=device, generator=gen)
    torch.manual_seed(3)                                         # identical replicas
    model = temporal.STGraphTGCN(feat, hidden, 1).to(device)
    # capturable + fused: the update rule of the default Adam as ONE kernel that a HIP graph can hold (the captured
    # windows replay it together with the gradient scaling; the eager epoch below uses the same optimizer object)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
    bucket = temporal.GradBucket(model.parameters())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, count):
        barrier()
        t0 = time.perf_counter()
        for i in range(count):
            fn(i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # (a) eager loop, as the reference scripts run it: one warm-up epoch, one timed epoch with
    #     per-launch HIP events (aggregation share, all-reduce share)
    temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=0, rank=rank, world=world)
    records = []
    kernels.enable_launch_timing(records)
    dt_eager = timed(lambda i: temporal.train_epoch_static(model, g, ew, targets, B, opt, bucket, feat, epoch=1 + i,
                                                           rank=rank, world=world, timed_comm=True), 1)
    kernels.enable_launch_timing(None)
    comm = bucket.collect_comm_time()
    # per-kernel HIP-event times of the eager epoch, each against the roofline that bounds it: the step launches and the
    # weight-gradient contractions run on the fp32-input matrix instruction (157.3 TFLOP/s), everything else moves bytes
    native_bytes_epoch = float(sum(r[3] for r in records))       # byte models of this build's own kernels, one epoch
    ktab = kernel_table(records, 1, TGCN_DENSE)
    step_launches = sum(1 for r in records if r[0] in ("tgcn_step_fwd", "tgcn_step_bwd"))
    # edges*feat EXECUTED by rank 0 in one epoch: each step launch gathers P = A_hat x at the INPUT width (E x feat), plus whatever
    # separate aggregation launches ran (none in the fused form)
    ef_rank0_epoch = step_launches * e * feat + executed_edges_feat(records)
    records = [r for r in records if r[0] in ("gcn_agg", "gcn_agg_transform")]
    agg_s = float(np.sum([a.elapsed_time(b) for (_, a, b, _, _) in records])) * 1e-3
    agg_launches = len(records)
    if world > 1:
        t = torch.tensor([comm], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm = float(t.item())
    # (b) the same windows replayed from a captured HIP graph (one capture, 40/N replays per epoch)
    cw = temporal.CapturedStaticWindow(model, g, ew, targets, B, opt, bucket, feat, world=world, rank=rank,
                                       allreduce_in_graph=allreduce_in_graph)
    targets_c = targets if world == 1 else None           # N > 1: the window object holds this rank's share
    for ep in range(warmup_epochs):
        temporal.train_epoch_static_captured(cw, model, g, ew, targets_c, opt, bucket, feat, epoch=2 + ep, rank=rank,
                                             world=world)
    calls0 = bucket.comm_calls
    comm0 = bucket.collect_comm_time()
    dt = timed(lambda i: temporal.train_epoch_static_captured(cw, model, g, ew, targets_c, opt, bucket, feat,
                                                              epoch=10 + i, rank=rank, world=world,
                                                              timed_comm=True), epochs)
    comm_g = bucket.collect_comm_time() - comm0
    if world > 1:
        t = torch.tensor([comm_g], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm_g = float(t.item())
    calls_timed = bucket.comm_calls - calls0
    # Per optimizer step: CPU time the rank spends issuing it (thread CPU time: a replay of a graph that is still in
    # flight blocks in the runtime, which wall time would count) and device idle time between steps (epoch wall time
    # minus the device time of the steps, from HIP events around each step)
    steps_epoch = (temporal.num_windows(T, B) + world - 1) // world
    ev = []
    run_plain = cw.run

    def run_timed(w, timed_comm=False):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = run_plain(w, timed_comm)
        b.record()
        ev.append((a, b))
        return r
    barrier()
    tc0, tw0 = time.thread_time(), time.perf_counter()
    temporal.train_epoch_static_captured(cw, model, g, ew, targets_c, opt, bucket, feat, epoch=98, rank=rank, world=world)
    host_cpu_s = time.thread_time() - tc0
    cw.run = run_timed
    barrier()
    tw0 = time.perf_counter()
    temporal.train_epoch_static_captured(cw, model, g, ew, targets_c, opt, bucket, feat, epoch=99, rank=rank, world=world)
    barrier()
    wall_ev = time.perf_counter() - tw0
    del cw.run                  # back to the class's method (an instance attribute holding the object's own bound method is a cycle)
    dev_busy_s = sum(a.elapsed_time(b) for a, b in ev) * 1e-3
    bucket.check_views()
    fused = bool(model.temporal.fuse_gates)
    agg_per_step = 2 if fused else 6                             # (fwd + bwd) x (1 fused | 3 separate) gates
    width = (3 if fused else 1) * hidden                         # width of the layer's aggregation A_hat (X W)
    steps_rank0 = sum(min(B, T - w * B) for _, w in temporal.windows_of_rank(T, B, rank, world) if w is not None)
    sec_per_snapshot = dt / epochs / max(steps_rank0, 1)         # this rank's snapshots run back to back
    # SURVEY.md 8(d) byte model of the reference's formulation: per snapshot three edge-weighted width-`hidden`
    # aggregations forward + three backward (nn/pytorch/temporal/tgcn.py:21-43 through gcn_conv.py:169-182)
    ref_bytes = 6 * kernels.gcn_agg_algorithmic_bytes(n, e, hidden, True)
    own_bytes = native_bytes_epoch / max(steps_rank0, 1)
    roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "seconds_per_snapshot": sec_per_snapshot,
            "bytes_model": "bytes MOVED per snapshot by this build: the sum of the byte models of every native launch of one "
                           "snapshot as it runs here (one width-in aggregate-then-transform + gates + GRU + head in the forward "
                           "step launch, their backward in the backward step launch, amortised weight-gradient contractions) "
                           "over the WHOLE captured step time (Adam and the loss included)",
            "algorithmic_bytes_per_snapshot": own_bytes,
            "achieved": own_bytes / sec_per_snapshot / 1e9, "frac": own_bytes / sec_per_snapshot / 1e9 / HBM_PEAK_GBS,
            "note": "the step launches are bound by the f32-input matrix instruction, not by HBM (profiles/r04_coexec_f32mfma.jsonl; "
                    "DESIGN.md section 0): this fraction says how far the snapshot is from its own HBM floor",
            "reference_formulation": {
                "bytes_model": "6 edge-weighted gcn_agg launches of width hidden per snapshot (3 gates, forward + backward), "
                               "SURVEY.md 8(d) bytes each: what the reference's formulation would have to move, over this "
                               "build's time -- a speed-up figure, NOT an achieved fraction (it can exceed 1)",
                "bytes_per_snapshot": ref_bytes,
                "equivalent_GBps": ref_bytes / sec_per_snapshot / 1e9,
                "equivalent_frac_of_hbm_peak": ref_bytes / sec_per_snapshot / 1e9 / HBM_PEAK_GBS}}
    step_k = {k: ktab[k] for k in ("tgcn_step_fwd", "tgcn_step_bwd") if k in ktab}
    if step_k:          # the dominant launch by time, against the roofline that bounds it (fp32-input matrix instruction)
        dom = max(step_k, key=lambda k: step_k[k]["mean_ms"] * step_k[k]["launches_per_iter"])
        roof["dominant_kernel"] = {"kernel": "stg::" + dom + "_kernel", "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MFMA_PEAK_TFLOPS,
                                   "achieved": step_k[dom]["achieved_TFLOPs"], "frac": step_k[dom]["frac"],
                                   "flops_per_launch": step_k[dom]["flops"], "mean_launch_ms": step_k[dom]["mean_ms"],
                                   "hbm_frac": step_k[dom]["hbm_frac"], "timed": "HIP events, eager epoch of rank 0"}
        roof["kernel"] = "whole snapshot on moved bytes; dominant launch: " + dom
    roof["kernels"] = ktab
    if cpu_baseline:
        progress("cfg4: CPU baseline")
    cpu = cpu_baseline_tgcn(g, ew, targets, n, e, T, feat, hidden, B) if cpu_baseline else None
    idle_us = max(0.0, wall_ev - dev_busy_s) / max(steps_epoch, 1) * 1e6
    per_rank = [{"rank": rank, "windows_run_per_epoch": len(cw.my_windows), "targets_resident_windows": int(cw.targets_w.shape[0]),
                 "device_idle_us_per_optimizer_step": idle_us, "allreduce_in_graph": bool(cw.allreduce_in_graph)}]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, per_rank[0])
        per_rank = got
        idle_us = max(r["device_idle_us_per_optimizer_step"] for r in per_rank)
    return {
        "roofline": roof, "cpu_baseline": cpu,
        "workload": f"static-temporal TGCN |V|={n} |E|={e} T={T} feat={feat} hidden={hidden} backprop_every={B} "
                    f"(BASELINE configs[3]), windows sharded over {world} rank(s), Adam; "
                    f"{'fused 3-gate aggregation (one launch; aggregate-then-transform on the matrix cores)' if fused else 'three width-64 aggregations'} per snapshot, fused row-local GRU cell, "
                    "the compute of each BPTT window (fwd + bwd through time) replayed from a HIP graph, then one eager "
                    "all-reduce of the gradient bucket and one Adam step",
        "metric": "epochs/s", "value": epochs / dt, "epochs": epochs, "seconds_per_epoch": dt / epochs,
        "edges_feat_per_s": ef_rank0_epoch * (T / max(steps_rank0, 1)) * epochs / dt,
        "edges_feat_per_s_is": "sum of E x F over the gathers EXECUTED (each step launch aggregates at the input width: 2 x T x E x feat "
                               "per epoch), whole job / wall time; edges_feat_per_s_reference_formulation counts the reference's six width-hidden aggregations per snapshot",
        "edges_feat_per_s_reference_formulation": agg_per_step * T * e * width * epochs / dt,
        "us_per_snapshot": sec_per_snapshot * 1e6,
        "scaling": "strong", "n_gpus": world,
        "windows_per_epoch": temporal.num_windows(T, B),
        "optimizer_steps_per_epoch": (temporal.num_windows(T, B) + world - 1) // world,
        "host_cpu_us_per_optimizer_step": host_cpu_s / max(steps_epoch, 1) * 1e6,
        "device_idle_us_per_optimizer_step": idle_us,
        "device_idle_us_per_optimizer_step_is": "max over ranks" if world > 1 else "this rank",
        "per_rank": per_rank,
        "host_ops_per_optimizer_step": ("graph replay (window), all-reduce (eager, N > 1 only), graph replay (grad / N, "
                                        "Adam, window index)") if cw.step_graph is not None else
                                       "graph replay, all-reduce + div, eager optimizer step, window index add",
        "eager": {"seconds_per_epoch": dt_eager, "epochs_per_s": 1.0 / dt_eager,
                  "rank0_gcn_agg_kernel_seconds": agg_s, "rank0_gcn_agg_launches": agg_launches,
                  "rank0_gcn_agg_share": agg_s / dt_eager, "rank0_native_kernels": ktab,
                  "allreduce_seconds_max_rank": comm, "allreduce_share": comm / dt_eager},
        "process_group": process_group_info(device, rank, world, require_distinct=not share_device),
        "allreduce": {"bytes": bucket.nbytes, "calls_per_epoch": calls_timed / max(epochs, 1),
                      "seconds_max_rank": comm_g, "share_of_epoch": comm_g / dt if dt else None,
                      "in_graph": bool(cw.allreduce_in_graph),
                      "collective": "one all-reduce(sum)/N of the flattened gradient bucket per optimizer step"},
    }


def dynamic_run(device, rank, world, epochs, n=25_000, e0=250_000, churn=6_250, T=None, B=20, feat=32, hidden=64,
                cpu_baseline=False, only_modes=None):
    """BASELINE.json configs[4]: dynamic-temporal TGCN (benchmarking/dynamic-temporal-tgcn/seastar/train.py loop:
    link prediction on a sliding window over an edge stream, un-weighted GCN gates), once with the per-snapshot
    device CSR rebuild (NaiveGraph(resident=False)), with all snapshots resident as the reference's NaiveGraph keeps
    them, and on the dynamic edge store behind both of the reference's delta-based graph classes (PCSRGraph,
    GPMAGraph: one resident graph + per-timestamp deltas).  Every mode replays one HIP graph per BPTT window after an
    eager epoch; BPTT windows are sharded over the ranks like the static configuration.  At ONE rank the workload is
    BASELINE.md's own T = 40 (2 windows of 20) and the T = 160 stream is the "T160" sub-object; with N > 1 ranks T = 160
    (8 windows) is the sharded workload (40 snapshots are 2 windows, which 8 ranks cannot share)."""
    from stgraph_amd import kernels, temporal
    from stgraph_amd.graph import GPMAGraph, NaiveGraph, PCSRGraph
    if T is None:
        T = 40 if world == 1 else 160

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_modes(T, modes, epochs, cpu=False):
        rng = np.random.default_rng(4)
        stream = rng.choice(n * n, size=e0 + churn * T, replace=False)
        snaps, pn_edges, pn_targets = [], [], []
        gen = torch.Generator(device=device).manual_seed(4)
        m = 10_000
        for t in range(T):
            keys = stream[t * churn: t * churn + e0]
            s, d = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
            snaps.append((torch.from_numpy(s).to(device), torch.from_numpy(d).to(device)))
            pos = torch.from_numpy(np.stack([s[:m], d[:m]]).astype(np.int64)).to(device)
            neg = torch.randint(0, n, (2, m), device=device, generator=gen)
            pn_edges.append(torch.cat([pos, neg], 1))
            pn_targets.append(torch.cat([torch.ones(m, device=device), torch.zeros(m, device=device)]))
        out = {}
        if cpu:
            progress(f"cfg5 T={T}: CPU baseline")
            out["cpu_baseline"] = cpu_baseline_dynamic(snaps, pn_edges, pn_targets, n, T, feat, hidden, B)
        for mode in modes:
            if rank == 0:
                progress(f"cfg5 T={T}: {mode}")
            if mode == "resident_snapshots":         # NaiveGraph as the reference defines it: all 2T CSRs built up front
                G = NaiveGraph(snaps, n, device=device, sort_inplace=False)
            elif mode in ("rebuild_per_snapshot", "rebuild_prefetch"):
                # rebuild_prefetch: the window's builds as a graph of their own, replayed one window ahead on a second stream
                # (CapturedDynamicWindows.prefetch_builds; an option) instead of at the head of the window's training graph
                G = NaiveGraph(snaps, n, device=device, sort_inplace=False, resident=False, max_cached=B + 1)
            else:
                G = (PCSRGraph if mode == "pcsr_store" else GPMAGraph)(snaps, n, device=device)
            torch.manual_seed(4)
            model = temporal.DynamicSTGraphTGCN(feat, hidden).to(device)
            # capturable + fused: the default Adam's update rule as one kernel that a HIP graph can hold
            opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True, fused=True)
            bucket = temporal.GradBucket(model.parameters())
            # every window replayed from its own HIP graph after one eager epoch: in rebuild mode the graph contains the
            # snapshot builds, on the stores the merges and CSR emissions of get_graph(t) -- both still run every epoch
            cd = None

            def epoch(ep):
                nonlocal cd
                if mode in ("rebuild_per_snapshot", "rebuild_prefetch"):
                    G._snapshots.clear()                 # every epoch rebuilds every snapshot it touches
                G._ndata.clear()
                if ep >= 1:
                    if cd is None:
                        cd = temporal.CapturedDynamicWindows(model, G, pn_edges, pn_targets, B, opt, bucket, feat, world=world,
                                                             rank=rank)
                        cd.prefetch_builds = mode == "rebuild_prefetch"
                    temporal.train_epoch_dynamic_captured(cd, epoch=ep)
                else:
                    temporal.train_epoch_dynamic(model, G, pn_edges, pn_targets, B, opt, bucket, feat, epoch=ep, rank=rank,
                                                 world=world)
            recs = []
            kernels.enable_launch_timing(recs)
            epoch(0)
            kernels.enable_launch_timing(None)
            own_bytes = float(sum(r[3] for r in recs))       # byte models of the native launches of one eager epoch
            # edges*feat executed by this rank in one epoch: every step launch gathers A_hat x at the input width over e0 edges
            ef_rank = sum(1 for r in recs if r[0] in ("tgcn_step_fwd", "tgcn_step_bwd")) * e0 * feat + executed_edges_feat(recs)
            del recs
            epoch(1)                                     # captures
            epoch(2)                                     # first pure replay (warm-up; epochs 0-2 discarded as the reference does)
            barrier()
            t0 = time.perf_counter()
            for ep in range(epochs):
                epoch(3 + ep)
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if mode in ("pcsr_store", "gpma_store"):
                G.check()
            out[mode] = {"epochs_per_s": epochs / dt, "seconds_per_epoch": dt / epochs, "hip_graph_per_window": True,
                         "epochs_timed": epochs, "epochs_discarded": 3, "native_launch_bytes_per_epoch": own_bytes,
                         "edges_feat_executed_per_epoch_this_rank": ef_rank}
            del G, model, opt, bucket, cd
            torch.cuda.empty_cache()
        return out

    all_modes = ("resident_snapshots", "rebuild_per_snapshot", "rebuild_prefetch", "pcsr_store", "gpma_store")
    if only_modes:                                               # (profiling runs: tools/diag/dyn_only.py)
        return run_modes(T, tuple(only_modes), epochs)
    if T == 40:
        epochs = max(epochs, 20)                                 # the reference's rule: >= 20 epochs, the first three discarded
    out = run_modes(T, all_modes, epochs, cpu=cpu_baseline)
    cpu = out.pop("cpu_baseline", None)
    dt_e = out["rebuild_per_snapshot"]["seconds_per_epoch"]
    steps_rank0 = sum(max(0, min(B, T - 1 - w * B)) for _, w in temporal.windows_of_rank(T, B, rank, world) if w is not None)
    sps = dt_e / max(steps_rank0, 1)
    ref_bytes = 6 * kernels.gcn_agg_algorithmic_bytes(n, e0, hidden, False) + 2 * 16 * e0
    # bytes this build MOVES per snapshot: the byte models of its native launches (step kernels, link head, amortised weight
    # gradients: recorded over the eager epoch) + the CSR build's 16 B/edge per direction
    own_bytes = out["rebuild_per_snapshot"]["native_launch_bytes_per_epoch"] / max(steps_rank0, 1) + 2 * 16 * e0
    roofline = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "seconds_per_snapshot": sps,
                "bytes_model": "bytes MOVED per snapshot by this build (rebuild_per_snapshot mode): the byte models of its native "
                               "launches (forward / backward step launches, link head, amortised weight gradients) + the CSR "
                               "build's 16 B/edge per direction (SURVEY.md 8(d)), over the WHOLE step time incl. Adam",
                "algorithmic_bytes_per_snapshot": own_bytes, "achieved": own_bytes / sps / 1e9,
                "frac": own_bytes / sps / 1e9 / HBM_PEAK_GBS,
                "note": "|V| = 25K snapshots: launch- and latency-bound, not bandwidth-bound",
                "reference_formulation": {
                    "bytes_model": "6 un-weighted gcn_agg launches of width hidden per snapshot (3 gates, forward + backward) "
                                   "+ the CSR build's 16 B/edge per direction: what the reference's formulation would move, "
                                   "over this build's time -- a speed-up figure, NOT an achieved fraction",
                    "bytes_per_snapshot": ref_bytes, "equivalent_GBps": ref_bytes / sps / 1e9,
                    "equivalent_frac_of_hbm_peak": ref_bytes / sps / 1e9 / HBM_PEAK_GBS}}
    res = {"workload": f"dynamic-temporal TGCN |V|={n} E0={e0} +-{churn} edges/step T={T} backprop_every={B} feat={feat} "
                       f"hidden={hidden} (BASELINE configs[4]), link-prediction loss, windows sharded over {world} rank(s)",
           "metric": "epochs/s", "value": out["rebuild_per_snapshot"]["epochs_per_s"], "T": T,
           "resident_epochs_per_s": out["resident_snapshots"]["epochs_per_s"],
           "edges_feat_per_s": out["rebuild_per_snapshot"]["edges_feat_executed_per_epoch_this_rank"] * ((T - 1) / max(steps_rank0, 1))
                               * out["rebuild_per_snapshot"]["epochs_per_s"],
           "edges_feat_per_s_reference_formulation": 6 * (T - 1) * e0 * hidden * out["rebuild_per_snapshot"]["epochs_per_s"],
           "value_is": "rebuild_per_snapshot -- the configuration BASELINE.md names (a fresh device CSR build per snapshot "
                       "and epoch, O(window) memory).  resident_snapshots is NaiveGraph as the reference keeps it (T forward "
                       "+ T backward CSRs built once at construction, graph/dynamic/naive/naive_graph.py); the two "
                       "delta-based stores follow.  " +
                       ("T = 40 is BASELINE.md's cfg5 exactly (2 BPTT windows of 20 per epoch); the longer T = 160 stream "
                        "(8 windows, the multi-rank workload) is in 'T160'" if T == 40 else
                        "T = 160 instead of BASELINE.md's 40: 40 snapshots are 2 windows of 20, which 8 ranks cannot share"),
           "csr_build_share": 1.0 - out["resident_snapshots"]["seconds_per_epoch"] / out["rebuild_per_snapshot"]["seconds_per_epoch"],
           "csr_build_share_is": "1 - seconds_per_epoch(resident_snapshots) / seconds_per_epoch(rebuild_per_snapshot): what the "
                                 "per-snapshot builds (and the per-edge coefficient gathers that follow a new CSR) add to the epoch; "
                                 "_prefetch = the same builds as a HIP graph of their own on a second stream, one window ahead of the "
                                 "training graph that reads them (an option: the step launches slow down beside them)",
           "csr_build_share_prefetch": 1.0 - out["resident_snapshots"]["seconds_per_epoch"] / out["rebuild_prefetch"]["seconds_per_epoch"],
           "scaling": "strong", "cpu_baseline": cpu,
           "n_gpus": world, "epochs": epochs, "windows_per_epoch": temporal.num_windows(T, B), "roofline": roofline,
           **out}
    if world == 1 and T == 40:
        t160 = run_modes(160, all_modes, epochs)
        res["T160"] = {"workload": "the same stream continued to T = 160 (8 BPTT windows of 20 per epoch: the workload the "
                                   "multi-rank runs shard)",
                       "metric": "epochs/s", "value": t160["rebuild_per_snapshot"]["epochs_per_s"], **t160}
    return res


def live_pmc_traffic(iters=2, timeout_s=170):
    """HBM bytes per cfg2 gcn_agg launch measured NOW: two child runs of tools/pmc_gcn.py under
    ``rocprofv3 --pmc`` (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters only, no trace domains), reduced as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: KiB units, WRITE_SIZE exact for 16-B-per-lane stores,
    FETCH_SIZE of a wide coalesced read x2 on gfx950 -- the factor is re-measured on the script's calibration launch
    (known bytes, same access shape) and both corrections are returned.  None if rocprofv3 is unavailable or a
    pass fails (the caller then falls back to the committed passes under profiles/)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="stg_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    rows, meta = {}, None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_gcn.py"), "--iters", str(iters)],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            meta = json.loads(r.stdout.strip().splitlines()[-1])
            files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
            if not files:
                return None
            sel = [x for x in csv.DictReader(open(files[0]))
                   if "gcn_agg_kernel" in x["Kernel_Name"] and x["Counter_Name"] == counter]
            sel.sort(key=lambda x: int(x["Dispatch_Id"]))
            rows[counter] = [float(x["Counter_Value"]) * 1024 for x in sel]
        seg = lambda v, k: float(np.mean(v[k * iters:(k + 1) * iters]))  # noqa: E731
        if len(rows["FETCH_SIZE"]) < 3 * iters or len(rows["WRITE_SIZE"]) < 3 * iters:
            return None
        corr = meta["calibration"]["known_read_bytes"] / seg(rows["FETCH_SIZE"], 0)
        per = {}
        for k, name in ((1, "forward_csr"), (2, "backward_csr")):
            fr, wr = seg(rows["FETCH_SIZE"], k), seg(rows["WRITE_SIZE"], k)
            per[name] = {"FETCH_SIZE_bytes_raw": fr, "WRITE_SIZE_bytes": wr, "traffic_bytes": fr * corr + wr,
                         "traffic_bytes_x2_rule": fr * 2 + wr}
        return {"traffic": 0.5 * (per["forward_csr"]["traffic_bytes"] + per["backward_csr"]["traffic_bytes"]),
                "traffic_guide_x2_rule": 0.5 * (per["forward_csr"]["traffic_bytes_x2_rule"] +
                                                per["backward_csr"]["traffic_bytes_x2_rule"]),
                "fetch_correction_measured": corr,
                "write_calibration_ratio": seg(rows["WRITE_SIZE"], 0) / meta["calibration"]["known_write_bytes"],
                "per_csr": per}
    except Exception:                                            # noqa: BLE001  (a profiler hiccup must not cost the bench line)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cfg2_run(args, device, rank, world, want_cpu):
    """BASELINE configs[1] (the N = 1 headline; N > 1: one independent replica per rank): K timed training steps of the 2-layer GCN
    between barrier + device sync, MAX over ranks; the dominant kernel's launches timed with HIP events inside the region."""
    from stgraph_amd import kernels
    from stgraph_amd.nn import functional as SF

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step, meta = gcn_setup(device, seed=1 + rank, n=args.nodes, e=args.edges, feat=args.feat)
    for _ in range(args.warmup):
        step()
    probe = []
    kernels.enable_launch_timing(probe)
    step()                                                     # one more untimed step: how many launches a step records
    torch.cuda.synchronize()
    records = []
    kernels.enable_launch_timing(records, prepare=(len(probe) + 2) * (args.steps + 1))      # events made before the clock starts
    del probe
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

    # the same step with every layer in the reference's order (x W first): 4 aggregation launches
    SF.set_input_layer_reorder(False)
    try:
        step()
        barrier()
        k_ref = max(2, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(k_ref):
            step()
        barrier()
        dt_ref = (time.perf_counter() - t0) / k_ref
    finally:
        SF.set_input_layer_reorder(True)

    ef_per_step = meta["agg_launches_per_step"] * meta["e"] * meta["feat"]
    # (the row products run as bf16 triples on the bf16 matrix instruction and are bound by their access pattern: priced on bytes)
    other = kernel_table([r for r in records if r[0] != "gcn_agg"], args.steps, ())
    gemm_ms = [a.elapsed_time(b) for (name, a, b, _, _) in records if name == "gemm_tn"]
    records = [r for r in records if r[0] == "gcn_agg"]           # the dominant kernel
    ms = [a.elapsed_time(b) for (_, a, b, _, _) in records]
    bytes_alg = records[0][3]
    mean_ms = float(np.mean(ms))
    achieved = bytes_alg / (mean_ms * 1e-3) / 1e9
    # edges*feat/s = sum over the aggregation launches IN THE TIMED REGION of E * F_launch / wall time
    executed = len(ms) / max(args.steps, 1)
    ef_executed = executed * meta["e"] * meta["feat"]
    line = {
        "metric": "edges*feat/s (2-layer GCN training epoch, BASELINE configs[1])",
        "value": world * args.steps * ef_executed / dt, "unit": "edges*feat/s",
        "value_is": "sum of E x F over the aggregation launches executed in the timed region / wall time (SURVEY.md 8(d)); "
                    "value_reference_formulation counts the 4 aggregations the same step has in the reference's order",
        "value_reference_formulation": world * args.steps * ef_per_step / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "epochs_per_s": world * args.steps / dt,
        "config": {"workload": f"2-layer GCN {args.feat}->{args.feat}->{args.feat}, synthetic CSR |V|={meta['n']} "
                               f"|E|={meta['e']} (BASELINE configs[1]); step = fwd + cross-entropy + bwd + Adam",
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas (no collective)",
                   "agg_launches_per_step_reference_formulation": meta["agg_launches_per_step"],
                   "aggregations_executed_per_step": executed,
                   "edges_feat_per_step": ef_executed,
                   "edges_feat_per_step_reference_formulation": ef_per_step,
                   "reference_compat_D1": kernels.reference_compat()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "stg::gcn_agg_kernel",
                     "algorithmic_bytes_per_launch": bytes_alg, "mean_launch_ms": mean_ms,
                     "launches_timed": len(ms),
                     "gcn_agg_share_of_step": float(np.sum(ms)) * 1e-3 / dt,
                     "weight_grad_gemm_tn_mean_ms": float(np.mean(gemm_ms)) if gemm_ms else None,
                     "other_kernels_of_the_step": other},
    }
    line["reference_order"] = {
        "ms_per_step": dt_ref * 1e3, "value": world * ef_per_step / dt_ref, "steps": k_ref, "aggregations_executed_per_step": 4,
        "what": "the same step with every GCNConv in the reference's order (x W, then aggregate: 4 aggregation launches); the default runs "
                "the first layer aggregate-first (its input carries no gradient): tests/test_gpu_input_layer.py, profiles/r03_input_layer_error.json"}
    # HBM-side traffic per launch: PMC counters cannot be read inside this process; the committed rocprofv3 --pmc passes over the
    # SAME kernel / shape first, replaced by this run's own child passes at the end (live_pmc_traffic)
    if (meta["n"], meta["e"], meta["feat"]) == (1_000_000, 16_000_000, 128):
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gcn_agg_edge*.json")))
        if pmc_files:
            pmc = json.load(open(pmc_files[-1]))
            line["roofline"]["traffic"] = 0.5 * (pmc["cfg2_forward_csr"]["traffic_bytes"] + pmc["cfg2_backward_csr"]["traffic_bytes"])
            line["roofline"]["traffic_source"] = os.path.relpath(pmc_files[-1], ROOT)
            line["roofline"]["traffic_guide_x2_rule"] = 0.5 * (pmc["cfg2_forward_csr"]["traffic_bytes_x2_rule"] +
                                                               pmc["cfg2_backward_csr"]["traffic_bytes_x2_rule"])
    cpu = None
    if want_cpu:
        progress("cfg2: CPU baseline (torch epoch, then the OpenMP aggregation)")
        cpu = cpu_baseline_epoch_gcn(meta, int(0.6 * meta["n"]), meta["labels"], executed)
        cpu["aggregation_kernel_openmp"] = cpu_baseline_gcn(meta, budget_s=8.0)
    line["cpu_baseline"] = cpu
    del step, meta
    torch.cuda.empty_cache()
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tgcn", action="store_true")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--tgcn-epochs", type=int, default=20, help="N = 1: timed epochs of the tgcn object (the reference: >= 20, the first three "
                                                                "discarded); at N > 1 the TGCN epoch IS the step: --steps / --warmup")
    ap.add_argument("--tgcn-timestamps", type=int, default=1000)
    ap.add_argument("--no-cora", action="store_true")
    ap.add_argument("--no-dynamic", action="store_true")
    ap.add_argument("--no-gat", action="store_true")
    ap.add_argument("--no-gcn", action="store_true", help="N > 1: skip the cfg2 replicas sub-object")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from the committed PMC passes")
    ap.add_argument("--dynamic-epochs", type=int, default=20)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device: exercise the multi-rank logic on a single GPU (testing only)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (testing only)")
    ap.add_argument("--allreduce-in-graph", action="store_true",
                    help="N > 1: capture the gradient all-reduce into the optimizer-tail HIP graph (default: eager between the replays)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python3 bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU and never will)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not args.share_device and torch.cuda.device_count() < world:          # device_count() does not initialise the GPU
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} HIP device(s) visible: every rank needs its own GPU "
                         "(--backend gloo --share-device rehearses the multi-rank logic on one)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible (there is no CPU fallback)")
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    from stgraph_amd import kernels
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    if world == 1:
        progress("cfg2: building the graph and the model")
        line = cfg2_run(args, device, rank, world, want_cpu)
        if not args.no_cora:
            progress("cfg1: Cora-shaped GCN")
            line["cora"] = cora_run(device, cpu_baseline=want_cpu)
            x1024 = line["cora"]["roofline_x1024"]
            line["roofline"]["north_star"] = {
                "workload": "fused GCN aggregation forward + backward at the Cora model's widths (F = 16 and F = 7) on 1024 "
                            "disjoint replicas of the Cora-shaped graph (SURVEY.md 8(d) 'Cora x K'); target >= 0.60",
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": x1024["achieved"], "frac": x1024["frac"],
                "frac_F16": x1024["F16"]["frac_of_hbm_peak"], "frac_F7": x1024["F7"]["frac_of_hbm_peak"]}
        if not args.no_gat:
            progress("cfg3: GAT")
            line["gat"] = gat_run(device, cpu_baseline=want_cpu)
            torch.cuda.empty_cache()
        if not args.no_tgcn:
            progress("cfg4: static-temporal TGCN")
            line["tgcn"] = tgcn_run(device, rank, world, epochs=args.tgcn_epochs, warmup_epochs=3, n=50_000, e=500_000,
                                    T=args.tgcn_timestamps, feat=32, hidden=64, B=25, allreduce_in_graph=args.allreduce_in_graph,
                                    cpu_baseline=want_cpu, share_device=args.share_device)
        if not args.no_dynamic:
            progress("cfg5: dynamic-temporal TGCN")
            line["dynamic"] = dynamic_run(device, rank, world, epochs=args.dynamic_epochs, cpu_baseline=want_cpu)
        if not args.no_live_pmc and line["roofline"].get("traffic") is not None:
            torch.cuda.empty_cache()
            progress("cfg2: rocprofv3 --pmc child passes (HBM traffic of gcn_agg)")
            live = live_pmc_traffic()
            if live is not None:
                line["roofline"]["traffic_committed_passes"] = line["roofline"]["traffic"]
                line["roofline"]["traffic"] = live["traffic"]
                line["roofline"]["traffic_guide_x2_rule"] = live["traffic_guide_x2_rule"]
                line["roofline"]["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child "
                                                      "passes over tools/pmc_gcn.py (same kernel, same shape)")
                line["roofline"]["traffic_detail"] = {k: live[k] for k in ("fetch_correction_measured",
                                                                           "write_calibration_ratio", "per_csr")}
    else:
        # N > 1: the path that SHARDS is the headline (BASELINE.json: "TGCN 1/2/4/8 MI355X"): one step = one training epoch of the
        # static-temporal TGCN (40 BPTT windows dealt to the ranks, one RCCL all-reduce of the gradient bucket per optimizer
        # step), `value` = whole-job epochs/s, scaling "strong" (the epoch is fixed, ranks split it)
        if rank == 0:
            progress(f"cfg4: static-temporal TGCN on {world} ranks (the headline at N > 1)")
        tg = tgcn_run(device, rank, world, epochs=args.steps, warmup_epochs=max(args.warmup, 1), n=50_000, e=500_000,
                      T=args.tgcn_timestamps, feat=32, hidden=64, B=25, allreduce_in_graph=args.allreduce_in_graph,
                      cpu_baseline=False, share_device=args.share_device)
        dk = tg["roofline"].get("dominant_kernel") or {}
        line = {"metric": "epochs/s (static-temporal TGCN, BASELINE configs[3], BPTT windows sharded over the ranks)",
                "value": tg["value"], "unit": "epochs/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
                "ms_per_step": tg["seconds_per_epoch"] * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"static-temporal TGCN |V|=50000 |E|=500000 T={args.tgcn_timestamps} feat=32 hidden=64 backprop_every=25; "
                                       "step = one training epoch",
                           "parallelism": f"dp{world}: window w on rank w mod {world}, one all-reduce(sum)/N of the 133 KB gradient bucket per optimizer step",
                           "windows_per_epoch": tg["windows_per_epoch"], "optimizer_steps_per_epoch": tg["optimizer_steps_per_epoch"],
                           "backprop_every": 25},
                "roofline": {"bound": dk.get("bound", "mfma"), "achieved": dk.get("achieved"), "peak": dk.get("peak"), "unit": dk.get("unit"),
                             "frac": dk.get("frac"), "traffic": None, "kernel": dk.get("kernel"), "flops_per_launch": dk.get("flops_per_launch"),
                             "mean_launch_ms": dk.get("mean_launch_ms"), "hbm_frac": dk.get("hbm_frac"),
                             "snapshot_on_moved_bytes_frac_of_hbm": tg["roofline"]["frac"]},
                "cpu_baseline": None,
                "allreduce": tg["allreduce"], "process_group": tg["process_group"], "edges_feat_per_s": tg["edges_feat_per_s"],
                "tgcn": tg}
        if not args.no_dynamic:
            if rank == 0:
                progress("cfg5: dynamic-temporal TGCN (windows sharded)")
            line["dynamic"] = dynamic_run(device, rank, world, epochs=args.dynamic_epochs, cpu_baseline=False)
        if not args.no_gcn:
            if rank == 0:
                progress("cfg2: one independent replica per rank (a single-graph GCN does not shard: SURVEY.md 8(e))")
            rep = cfg2_run(args, device, rank, world, False)
            line["gcn_replicas"] = {"metric": rep["metric"], "value": rep["value"], "unit": rep["unit"], "scaling": "weak", "n_gpus": world,
                                    "ms_per_step": rep["ms_per_step"], "roofline": rep["roofline"], "config": rep["config"]}
    line["knobs"] = kernels.knobs() if hasattr(kernels, "knobs") else None
    line["cpu_threads_rule"] = CPU_THREADS_RULE
    if rank == 0:
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
