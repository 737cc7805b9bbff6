"""``STGraph(backend).compile(gnn_module=...)`` -- the vertex-centric operator API
(reference compiler/stgraph.py:24-226).

``compile`` returns a decorator; the decorated vertex function becomes a
``Context`` that is called as ``fn(g=graph, n_feats={...}, e_feats={...})`` and
returns a tensor (or a tuple).  First call per input signature: run the function
once on symbolic values (``val.Val``), obtain the GIR, pick the kernel plan
(``dispatch.make_plan``).  Every call: bind tensors, run the plan's forward
kernels inside one autograd node.

Deviation from the reference, on purpose: the
reference caches ONE executor per function *name*, so the two ``nb_compute``
variants of GCNConv alias each other; here the cache key also contains the input
signature (feature names, feature shapes, which inputs require grad).
"""
from __future__ import annotations

import functools
from collections.abc import Iterable

from .backend.callback import STGraphBackend
from .dispatch import make_plan
from .executor import Executor
from .gir import Program, ValType
from .node import CentralNode
from .val import Val


class Context:
    def __init__(self, func, nspace, run_cb):
        functools.update_wrapper(self, func)
        self._f = func
        self._nspace = nspace
        self._entry_count = 0
        self._run_cb = run_cb
        self._executors = {}
        self._executor_cache = None          # most recent executor (reference attribute name)

    def __call__(self, **kwargs):
        executor = self._setup_executor(**kwargs)
        ret = self._run_cb(executor)
        if len(ret) == 1:
            return ret[0]
        return ret

    @staticmethod
    def _signature(node_feats, edge_feats):
        sig = []
        for kind, feats in (("n", node_feats), ("e", edge_feats)):
            for k, v in feats.items():
                sig.append((kind, k, tuple(v.shape[1:]), bool(v.requires_grad)))
        return tuple(sig)

    def _setup_executor(self, **kwargs):
        graph = kwargs.get("g", None)
        node_feats = kwargs.get("n_feats", {}) or {}
        edge_feats = kwargs.get("e_feats", {}) or {}
        if not graph:
            raise NameError("Need to provide the graph as one of keyward arguments")
        sig = self._signature(node_feats, edge_feats)
        executor = self._executors.get(sig)
        if executor is None:
            prog = Program()
            rets = self._trace(node_feats, edge_feats, prog)
            plan = make_plan(rets, prog)
            executor = Executor(graph, plan, sig)
            executor.program = prog
            self._executors[sig] = executor
        input_map = {("n", k): v for k, v in node_feats.items()}
        input_map.update({("e", k): v for k, v in edge_feats.items()})
        if hasattr(executor.plan, "params"):          # module parameters read inside the vertex function: autograd inputs too
            input_map.update({("p", k): v for k, v in executor.plan.params().items()})
        executor.restart(input_map, graph)
        self._executor_cache = executor
        self._entry_count += 1
        return executor

    def _trace(self, nfeats, efeats, prog):
        cen = CentralNode()
        for k, v in nfeats.items():
            setattr(cen, k, Val.leaf(prog, k, ValType.DEST, v))
            for nb in cen.innbs:
                setattr(nb, k, Val.leaf(prog, k, ValType.SRC, v))
        for k, v in efeats.items():
            for e in cen.inedges:
                setattr(e, k, Val.leaf(prog, k, ValType.EDGE, v))
        ret = self._f(cen)
        if ret is None:
            raise NameError("Ret is none. Execution is aborted")
        rets = list(ret) if isinstance(ret, Iterable) and not isinstance(ret, Val) else [ret]
        for r in rets:
            if not isinstance(r, Val):
                raise TypeError("a vertex function must return traced values (got %r)" % type(r).__name__)
        return [r.node for r in rets]


class STGraph:
    def __init__(self, backend_framework: STGraphBackend):
        self._ctx_map = {}
        self._backend_framework = backend_framework
        self._run_cb = backend_framework.backend_cb

    def compile(self, gnn_module, hetero_graph=False):
        namespace = [gnn_module, self._backend_framework.backend_module]

        def wrapper(func):
            if func.__name__ not in self._ctx_map:
                if hetero_graph:
                    raise NotImplementedError("Heterogeneous graph is not supported yet")
                self._ctx_map[func.__name__] = Context(func, namespace, self._run_cb)
            else:
                # the closure is re-created on every layer call (gcn_conv.py:162): keep the newest body
                self._ctx_map[func.__name__]._f = func
            return self._ctx_map[func.__name__]

        return wrapper
