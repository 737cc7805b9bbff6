"""Run the reference's Python compiler stack on CPU tensors (BUILD CONTAINER ONLY).

This module exists to GENERATE golden vectors (tests/golden/make_golden.py).
It reads /root/reference, which does not exist on the GPU box, so nothing in
the test-suite imports it at run time; the committed ``*.npz`` files are what
travels.

What runs unmodified from /root/reference/stgraph:
  the tracer (compiler/stgraph.py), GIR + passes + autodiff, the registry's
  per-op code snippets, the jinja kernel templates (code_gen/), the launch
  geometry (execution_unit.py:92-106), the Executor with its state/timestamp
  stacks (executor.py), the autograd wrapper (backend/), the nn layers
  (nn/pytorch/...), and the Python halves of StaticGraph / NaiveGraph /
  DynamicGraph (edge-list sorting, eid assignment, per-timestamp pointer swap).

What is substituted, and why:
  1. import-only stubs for pip modules the image lacks (snoop, prettytable,
     termcolor, cuda-python, pynvrtc) and for the pcsr/gpma extension modules
     that stgraph/graph/__init__.py imports eagerly.  None of them computes
     anything on this path.
  2. ``stgraph.graph.static.csr`` / ``stgraph.graph.dynamic.pcsr.pcsr`` (csr.cu and
     pcsr.cu need cuda_runtime/thrust/cub/nvcc: unbuildable here, and the shipped
     csr.so / pcsr.so are CPython-3.8 modules this interpreter refuses to import).
     The ``CSR`` / ``PCSR`` classes installed here call the reference's OWN compiled
     C++ classes inside those .so files through oracle/_ref/libref_shim.so
     (oracle/ref_shim.cpp) and hand out HOST addresses instead of device ones; only
     the pybind glue is replaced.  (Without the shim -- it needs /root/reference --
     ``CSR`` falls back to oracle.orc_csr_ctor, which
     tests/test_oracle_pcsr.py::test_csr_oracle_equals_reference_binary
     shows to be identical.)
  3. the nvcc -> PTX -> cuModuleLoad step (code_gen/compiler.py:36-44): the
     CUDA source the reference EMITS is compiled verbatim with g++ behind a
     10-line header that serialises the SIMT launch (blockIdx/threadIdx loops,
     ``__ldg`` = load, ``atomicAdd`` = add).  x86-64 without -mfma and
     -ffp-contract=off, so every a*b+c is two roundings, in program order.
"""
from __future__ import annotations

import ctypes
import enum
import hashlib
import importlib
import os
import re
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch  # noqa: F401  (must be imported BEFORE the stubs are installed)

REFERENCE_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)

from oracle import ref_shim  # noqa: E402
from oracle import stg_oracle as orc  # noqa: E402

_WORKDIR = tempfile.mkdtemp(prefix="stg_ref_harness_")
EMITTED_SOURCES: list[str] = []      # every CUDA translation unit the reference emitted


# --------------------------------------------------------------------------- stubs
def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _CUresult(enum.Enum):
    CUDA_SUCCESS = 0


class _NvrtcResult(enum.Enum):
    NVRTC_SUCCESS = 0


class _AnyAttr:
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return 0


class _CudaDriverStub(types.ModuleType):
    """cuda.cuda / cuda.cudart: every function 'succeeds' and returns nothing."""

    CUresult = _CUresult
    CUdevice_attribute = _AnyAttr()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: (_CUresult.CUDA_SUCCESS, None)


class _RefCSR:
    """The pybind class of csr.cu:181-201: the reference's compiled ``CSR::CSR`` (csr.so) when the shim
    is available, else the oracle's restatement."""

    def __init__(self, edge_list, edge_weight, num_nodes, is_edge_reverse=False):
        arr = np.asarray(edge_list, dtype=np.int64).reshape(-1, 3)
        w = np.asarray(edge_weight, dtype=np.float32)
        if ref_shim.available():
            c = types.SimpleNamespace(**ref_shim.csr_ctor(arr[:, 0], arr[:, 1], arr[:, 2], w, num_nodes,
                                                          is_edge_reverse))
        else:
            c = orc.csr_ctor(arr[:, 0], arr[:, 1], arr[:, 2], w, num_nodes, is_edge_reverse)
        self._c = c                      # keeps the host arrays alive
        self.row_offset_ptr = c.row_offset.ctypes.data
        self.column_indices_ptr = c.column_indices.ctypes.data
        self.eids_ptr = c.eids.ctypes.data
        self.node_ids_ptr = c.node_ids.ctypes.data
        self.out_degrees = c.out_degrees.tolist()
        self.in_degrees = c.in_degrees.tolist()
        self.weighted_out_degrees = c.weighted_out_degrees.tolist()


class _RefPCSR:
    """The pybind class of pcsr.cu:916-939 on top of the reference's compiled ``PCSR`` (pcsr.so).
    Like the original, an object owns ONE set of output arrays ("device" arrays, here host memory)
    that every build overwrites and that copies share (the copy constructor copies the pointers)."""

    BUILD_LOG: list = []                 # (kind, arrays) of every build, for the fixture generator

    def __init__(self, init_n, max_edge_count, _p=None, _dev=None):
        self._p = _p if _p is not None else ref_shim.RefPCSR(init_n, max_edge_count)
        self._dev = _dev if _dev is not None else [np.zeros(init_n + 1, np.int32), np.zeros(max_edge_count, np.int32),
                                                   np.zeros(max_edge_count, np.int32), np.zeros(init_n, np.int32)]

    def __deepcopy__(self, memo):
        return _RefPCSR(self._p.n, self._p.max_edges, self._p.copy(), self._dev)

    __copy__ = lambda self: self.__deepcopy__({})  # noqa: E731

    @property
    def in_degrees(self):
        return self._p.degrees()[0].tolist()

    @property
    def out_degrees(self):
        return self._p.degrees()[1].tolist()

    @property
    def edge_count(self):
        return self._p.edge_count

    def get_n(self):
        return self._p.n

    def edge_update_list(self, edge_list, is_delete=False, is_reverse_edge=False):
        self._p.edge_update_list(edge_list, is_delete, is_reverse_edge)

    def label_edges(self):
        self._p.label_edges()

    def get_edges(self):
        return [tuple(map(int, r)) for r in self._p.get_edges()]

    def _publish(self, kind, out):
        for dst, k in zip(self._dev, ("row_offset", "column_indices", "eids", "node_ids")):
            dst[: len(out[k])] = out[k].astype(np.int32)
        _RefPCSR.BUILD_LOG.append((kind, {k: v.copy() for k, v in out.items()}))
        return 0.0

    def build_csr(self):
        return self._publish("fwd", self._p.build_csr())

    def build_reverse_csr(self):
        return self._publish("bwd", self._p.build_reverse_csr())

    def get_csr_ptrs(self):
        return tuple(a.ctypes.data for a in self._dev)


def _get_array(ptr, size):
    return list((ctypes.c_int32 * size).from_address(ptr))


def install_stubs() -> None:
    if "stgraph" in sys.modules:
        return
    ident = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)  # noqa: E731
    _module("snoop", install=lambda **k: None, snoop=ident)
    sys.modules["snoop"].__call__ = ident

    class _Snoop(types.ModuleType):
        def __call__(self, f=None, *a, **k):
            return f if callable(f) else (lambda g: g)
    sn = _Snoop("snoop")
    sn.install = lambda **k: None
    sn.snoop = ident
    sys.modules["snoop"] = sn

    _module("prettytable", PrettyTable=type("PrettyTable", (), {}))
    _module("termcolor", colored=lambda s, *a, **k: s)
    cuda_pkg = _module("cuda")
    cuda_pkg.__path__ = []
    drv = _CudaDriverStub("cuda.cuda")
    rt = _CudaDriverStub("cuda.cudart")
    nv = _module("cuda.nvrtc", nvrtcResult=_NvrtcResult)
    sys.modules["cuda.cuda"], sys.modules["cuda.cudart"] = drv, rt
    cuda_pkg.cuda, cuda_pkg.cudart, cuda_pkg.nvrtc = drv, rt, nv
    p = _module("pynvrtc")
    p.__path__ = []
    p.compiler = _module("pynvrtc.compiler", Program=object, ProgramException=Exception)
    _module("pynvml")

    sys.path.insert(0, REFERENCE_ROOT)
    _module("stgraph.graph.static.csr", CSR=_RefCSR, get_array=_get_array)
    _module("stgraph.graph.dynamic.pcsr.pcsr", PCSR=_RefPCSR if ref_shim.available() else object)
    names = ["GPMA", "build_backward_csr", "edge_update_t", "free_backward_csr", "get_csr_ptrs",
             "get_in_degrees", "get_out_degrees", "init_gpma", "init_graph_updates", "label_edges"]
    _module("stgraph.graph.dynamic.gpma.gpma", **{n: object for n in names})


# --------------------------------------------------------------- emitted CUDA -> g++
_SIMT_HEADER = r"""
#include <cmath>
struct dim3_ { int x, y, z; };
static thread_local dim3_ blockIdx, threadIdx, blockDim, gridDim;
#define __global__
template <class T> static inline T __ldg(const T *p) { return *p; }
static inline float atomicAdd(float *p, float v) { float o = *p; *p += v; return o; }
using std::exp;
"""


def _compile_emitted(cuda_text: str):
    """Replacement for code_gen/compiler.py:36-44 (``compile_cuda``)."""
    EMITTED_SOURCES.append(cuda_text)
    launchers = []
    for m in re.finditer(r'extern "C" __global__ void (K\d+)\s*\((.*?)\)\s*\{', cuda_text, re.S):
        name, params = m.group(1), m.group(2)
        plist = [p.strip() for p in params.split(",") if p.strip()]
        pnames = [re.split(r"[\s\*]+", p)[-1] for p in plist]
        launchers.append(
            'extern "C" void launch_%s(%s, int nblk_, int nthr_) {\n'
            "  blockDim.x = nthr_; gridDim.x = nblk_;\n"
            "  for (int b_ = 0; b_ < nblk_; ++b_) for (int t_ = 0; t_ < nthr_; ++t_) {\n"
            "    blockIdx.x = b_; threadIdx.x = t_; %s(%s); }\n}\n"
            % (name, ", ".join(plist), name, ", ".join(pnames)))
    src = _SIMT_HEADER + cuda_text + "\n" + "\n".join(launchers)
    tag = hashlib.sha1(src.encode()).hexdigest()[:16]
    cpp, so = os.path.join(_WORKDIR, tag + ".cpp"), os.path.join(_WORKDIR, tag + ".so")
    if not os.path.exists(so):
        with open(cpp, "w") as f:
            f.write(src)
        subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", cpp, "-o", so])
    return ctypes.CDLL(so)


def load_reference():
    """Import the reference with the substitutions above; returns the ``stgraph`` package."""
    install_stubs()
    import stgraph  # noqa: F401
    cg = importlib.import_module("stgraph.compiler.code_gen.code_gen")
    cg.compile_cuda = _compile_emitted
    eu = importlib.import_module("stgraph.compiler.execution_unit")
    eu.cuModuleGetFunction = lambda mod, name: (_CUresult.CUDA_SUCCESS, getattr(mod, "launch_" + name.decode()))

    def _run(self, tensor_list):
        # execution_unit.py:359-372 / :407-415: grid = launch_config[0], block = launch_config[3]
        self.K(*tensor_list, *self.const_kernel_args,
               ctypes.c_int(self.launch_config[0]), ctypes.c_int(self.launch_config[3]))
    eu.Kernel.run = _run

    from stgraph.graph.dynamic.naive.naive_graph import NaiveGraph
    NaiveGraph._get_cached_graph = lambda self, ts=None: False
    return sys.modules["stgraph"]
