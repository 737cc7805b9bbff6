"""Per-kernel cost of a window's batched snapshot rebuild against one build per snapshot (|V| = 25 K, |E| = 250 K, cfg5).
Run under rocprofv3 --kernel-trace --stats for the per-kernel split; prints wall time per snapshot of both forms."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stgraph_amd import kernels  # noqa: E402

n, e, jobs = 25_000, 250_000, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
lists = []
for _ in range(jobs):
    keys = rng.choice(n * n, size=e, replace=False)
    lists.append((torch.from_numpy((keys // n).astype(np.int32)).to(dev), torch.from_numpy((keys % n).astype(np.int32)).to(dev)))
    kernels.build_graph_csr(*lists[-1], n, dev, lazy_node_ids=True)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / jobs * 1e6


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep


g1, k1 = graphed(lambda: [kernels.build_graph_csr(s, d, n, dev, lazy_node_ids=True, known_path="direct") for s, d in lists])
g2, k2 = graphed(lambda: kernels.build_graph_csr_batch(lists, n, dev))
print(f"one build per snapshot : {timed(g1.replay):7.2f} us / snapshot (HIP graph of {jobs})")
print(f"batched                : {timed(g2.replay):7.2f} us / snapshot (HIP graph of {jobs})")
