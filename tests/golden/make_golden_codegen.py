"""Golden vectors for vertex functions OUTSIDE stgraph.nn, produced by the reference's own compiler
(tracer -> fusion -> autodiff -> emitted CUDA, run through tests/golden/ref_harness.py).

BUILD CONTAINER ONLY.  Usage:  python tests/golden/make_golden_codegen.py
Writes tests/golden/codegen.npz: for each function its inputs, output(s) and input gradients on a
StaticGraph (N=40, E=260).  Functions avoid the reference's known defects (feature widths are powers
of two: D1; no subtraction: D3; differentiable inputs are never broadcast inside the function, because
the emitted backward writes a broadcast operand's gradient from every lane without reducing).
"""
from __future__ import annotations

import numpy as np
import torch

import make_golden as mg
from stgraph.compiler import STGraph  # noqa: E402
from stgraph.compiler.backend.pytorch.torch_callback import STGraphBackendTorch  # noqa: E402
from stgraph.graph import StaticGraph  # noqa: E402


class Host(torch.nn.Module):
    """A gnn_module for ``compile``: holds the STGraph context and the activation sub-modules."""

    def __init__(self):
        super().__init__()
        self.stgraph = STGraph(STGraphBackendTorch())
        self.leaky_relu = torch.nn.LeakyReLU(0.2)

    def run(self, name, g, nf, ef):
        if name == "gin":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([nb.h for nb in v.innbs]) + v.h
        elif name == "mean":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([nb.h for nb in v.innbs]) / v.deg
        elif name == "edge_affine":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([e.src.h * e.w + e.b for e in v.inedges])
        elif name == "relu_gcn":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return torch.relu(sum([nb.h * nb.norm for nb in v.innbs]) * v.norm)
        elif name == "exp_edge":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([torch.exp(e.src.h) * e.w for e in v.inedges])
        elif name == "leaky_edge":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([self.leaky_relu(e.src.h * e.w) for e in v.inedges])
        elif name == "two_level":
            @self.stgraph.compile(gnn_module=self)
            def f(v):
                return sum([nb.h * sum([n2.g for n2 in v.innbs]) for nb in v.innbs])
        else:
            raise KeyError(name)
        return f(g=g, n_feats=nf, e_feats=ef)


SPECS = {   # name: (node feats {name: (shape, differentiable)}, edge feats)
    "gin": ({"h": ((16,), True)}, {}),
    "mean": ({"h": ((16,), True), "deg": ((1,), False)}, {}),
    "edge_affine": ({"h": ((8,), True)}, {"w": ((8,), True), "b": ((8,), True)}),
    "relu_gcn": ({"h": ((32,), True), "norm": ((1,), False)}, {}),
    "exp_edge": ({"h": ((8,), True)}, {"w": ((8,), True)}),
    "leaky_edge": ({"h": ((8,), True)}, {"w": ((8,), True)}),
    "two_level": ({"h": ((8,), True), "g": ((8,), True)}, {}),
}


def main():
    n, e = 40, 260
    rng = np.random.default_rng(31)
    el = mg.random_edges(rng, n, e, hub=2, isolated=(7,))
    given = np.array(el, np.int32)
    d = dict(num_nodes=n, src=given[:, 0], dst=given[:, 1])
    ok = []
    for name, (nspec, espec) in SPECS.items():
        g = StaticGraph(list(el), [1.0] * e, n)
        gen = torch.Generator().manual_seed(hash_name(name))
        mk = lambda rows, shape, key: (torch.rand((rows,) + shape, generator=gen) + 0.5 if key in ("deg", "norm")  # noqa: E731
                                       else torch.randn((rows,) + shape, generator=gen) * 0.5)
        nf = {k: mk(n, s, k).requires_grad_(dg) for k, (s, dg) in nspec.items()}
        ef = {k: mk(e, s, k).requires_grad_(dg) for k, (s, dg) in espec.items()}
        try:
            out = Host().run(name, g, nf, ef)
            R = torch.randn(out.shape, generator=gen)
            (out * R).sum().backward()
        except Exception as ex:                                   # noqa: BLE001
            print(f"[reference cannot run '{name}': {type(ex).__name__}: {str(ex)[:200]}]")
            continue
        for k, v in {**nf, **ef}.items():
            d[f"{name}_in_{k}"] = v.detach()
            if v.requires_grad:
                d[f"{name}_grad_{k}"] = v.grad.detach().clone()
        d[f"{name}_out"], d[f"{name}_R"] = out.detach(), R
        ok.append(name)
    d["functions"] = np.array(ok)
    mg.save("codegen.npz", d)
    print("functions recorded:", ok)


def hash_name(name):
    return sum(ord(c) * (i + 1) for i, c in enumerate(name)) + 9000


if __name__ == "__main__":
    main()
