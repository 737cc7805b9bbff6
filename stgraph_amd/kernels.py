"""Tensor-level launch wrappers over the C ABI (include/stgraph_hip.h).

PyTorch is used for device memory and the current HIP stream only; all compute
on this path happens in libstgraph_hip.so.  Every wrapper validates shapes,
dtypes, contiguity and device placement on the host BEFORE launching (a faulting
kernel can take the whole node down), and raises instead of falling back.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _C

_REF_COMPAT = False


def set_reference_compat(enabled: bool) -> None:
    """Reproduce reference defect D1.

    The reference's launch geometry (compiler/execution_unit.py:92-106) computes
    only the first ``2**floor(log2 F)`` feature columns when F < 64 is not a power
    of two and leaves the rest zero.  Off (default): all F columns are computed.
    """
    global _REF_COMPAT
    _REF_COMPAT = bool(enabled)


def reference_compat() -> bool:
    return _REF_COMPAT


def ref_active_columns(feat_size: int) -> int:
    """Columns the reference computes (execution_unit.py:92-116)."""
    if feat_size >= 64:
        return feat_size
    ub = 64
    while ub > feat_size:
        ub //= 2
    return max(ub, 1)


def active_columns(feat_size: int) -> int:
    return ref_active_columns(feat_size) if _REF_COMPAT else feat_size


# ----------------------------------------------------------------------- launch timing
_TIMING: list | None = None


_EVENTS: list = []


def enable_launch_timing(records: list | None, prepare: int = 0) -> None:
    """Bracket every aggregation launch with HIP events on its own stream (bench.py uses this
    to measure the dominant kernel inside the timed region).  ``None`` switches it off.
    ``prepare``: that many event pairs are created AND recorded once now, so that the timed region finds them made (a first
    hipEventCreate / first record of a fresh event is a runtime allocation: measured up to 4 ms of host time per 30 events on
    a cold process, which made an otherwise device-bound step host-bound)."""
    global _TIMING
    _TIMING = records
    del _EVENTS[:]
    if records is not None and prepare > 0 and torch.cuda.is_available():
        for _ in range(2 * int(prepare)):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            _EVENTS.append(ev)
        torch.cuda.synchronize()


def _event():
    return _EVENTS.pop() if _EVENTS else torch.cuda.Event(enable_timing=True)


class _Timed:
    """Context manager recording (name, start event, stop event, algorithmic bytes, units)."""

    def __init__(self, name: str, nbytes: int, units: int):
        self.rec = (name, nbytes, units)

    def __enter__(self):
        if _TIMING is not None:
            self.e0, self.e1 = _event(), _event()
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _TIMING is not None and exc[0] is None:
            self.e1.record()
            _TIMING.append((self.rec[0], self.e0, self.e1, self.rec[1], self.rec[2]))
        return False


def gat_algorithmic_bytes(N: int, E: int, H: int, D: int) -> dict:
    """SURVEY.md 8(d), per launch."""
    idx = 4 * (N + 1) + 4 * E
    return {
        "gat_k0": 4 * E * H + 4 * E * H + 8 * N * H + idx + 4 * E,
        "gat_k1": 4 * E * H * D + 4 * E * H + 4 * N * H + 4 * N * H * D + idx + 4 * E,
        "gat_bwd": 2 * 4 * E * H * D + 2 * 4 * E * H + 2 * 4 * N * H * D + 8 * N * H + idx + 4 * E + 4 * E * H,
        "gat_bwd_er": 4 * E * H + 4 * N * H + idx,
    }


def gcn_agg_algorithmic_bytes(N: int, E: int, F: int, edge_weighted: bool) -> int:
    """neighbour-row gather + output + row_offsets + column_indices +
    norm[col] per edge + norm[row] (+ eids and weights)."""
    return 4 * E * F + 4 * N * F + 4 * (N + 1) + 4 * E + 4 * E + 4 * N + (8 * E if edge_weighted else 0)


# --------------------------------------------------------------------------- helpers
def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} lives on {t.device}; the Seastar kernels run only on an MI355X (HIP) device "
            "and stgraph_amd has no CPU fallback")


def _f32(t: torch.Tensor, name: str, device: torch.device | None = None) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    _require_gpu(t, name)
    if device is not None and t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    return t if t.is_contiguous() else t.contiguous()       # reference defect D6: raw data_ptr of strided grads


def _ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------- CSR
class DeviceCSR:
    """One CSR (the four arrays of the reference's ``CSR`` object, csr.cu:35-59).

    ``node_ids`` (rows by non-increasing degree) only fixes a processing ORDER -- results never depend on it --
    and sorting |V| degrees costs more launches than the rest of a small graph's build, so a builder may leave it
    to first use (``degrees`` given, ``node_ids=None``).  ``node_ids_if_ready`` is what the launch wrappers ask
    for: a snapshot that is built, used for one training step and dropped never sorts its degrees."""

    def __init__(self, row_offset, column_indices, eids, node_ids=None, degree_sorted: bool = False, degrees=None):
        self.row_offset = row_offset           # int32 [N+1]
        self.column_indices = column_indices   # int32 [E]
        self.eids = eids                       # int32 [E]
        self._node_ids = node_ids              # int32 [N] or None (lazy)
        self._degrees = degrees                # int32 [N] row lengths, needed only for the lazy case
        # True when node_ids is KNOWN to be ordered by non-increasing degree (set by this package's builders).  Only
        # then may the aggregation find its long rows through it (stg_gcn_agg_edge's rows_by_degree); a CSR assembled
        # by hand keeps every row on the row-group path.
        self.degree_sorted = bool(degree_sorted)

    @property
    def node_ids(self) -> torch.Tensor:
        if self._node_ids is None:
            self._node_ids = rows_by_degree(self._degrees if self._degrees is not None
                                            else (self.row_offset[1:] - self.row_offset[:-1]).contiguous())
            self.degree_sorted = True
        return self._node_ids

    @node_ids.setter
    def node_ids(self, value) -> None:
        self._node_ids = value

    @property
    def node_ids_if_ready(self):
        return self._node_ids

    @property
    def num_nodes(self) -> int:
        return self.row_offset.shape[0] - 1

    @property
    def num_edges(self) -> int:
        return self.column_indices.shape[0]

    # attribute names of the pybind class (csr.cu:186-189)
    @property
    def row_offset_ptr(self) -> int:
        return self.row_offset.data_ptr()

    @property
    def column_indices_ptr(self) -> int:
        return self.column_indices.data_ptr()

    @property
    def eids_ptr(self) -> int:
        return self.eids.data_ptr()

    @property
    def node_ids_ptr(self) -> int:
        return self.node_ids.data_ptr()


@dataclass
class GraphCSR:
    """Forward (dst-major) + backward (src-major) CSR of one graph / snapshot."""

    num_nodes: int
    fwd: DeviceCSR
    bwd: DeviceCSR
    in_degrees: torch.Tensor       # int32 [N]
    out_degrees: torch.Tensor      # int32 [N]
    perm_fwd: torch.Tensor         # int64 [E]: caller position of the edge that became eid j
    built_by: str = ""             # 'direct' | 'sort' | 'host': which builder produced (and validated) it
    unchecked_status: "torch.Tensor | None" = None    # the build's status word when its read was skipped (known_path)
    norm_in: "torch.Tensor | None" = None             # in_deg ** -0.5 [N, 1] when the (fused re-)build produced it

    @property
    def num_edges(self) -> int:
        return self.fwd.num_edges


def _as_i32(a, device: torch.device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
    return t.to(device=device, dtype=torch.int32).contiguous()


def rows_by_degree(degrees: torch.Tensor) -> torch.Tensor:
    """Rows by non-increasing degree, ties by ascending id (stg_rows_by_degree_device / a stable host sort)."""
    N = int(degrees.shape[0])
    out = torch.empty(N, dtype=torch.int32, device=degrees.device)
    if N == 0:
        return out
    if degrees.device.type != "cuda":
        return torch.sort(degrees.to(torch.int64), descending=True, stable=True).indices.to(torch.int32)
    ws_bytes = int(_C.lib.stg_rows_by_degree_workspace_bytes(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=degrees.device)
    with torch.cuda.device(degrees.device):
        _C.check(_C.lib.stg_rows_by_degree_device(_ptr(degrees.contiguous()), N, _ptr(out), _ptr(ws), ws_bytes,
                                                  _stream_ptr(degrees.device)))
    return out


BUILD_NEEDS_SORT = 32                  # include/stgraph_hip.h STG_BUILD_NEEDS_SORT
# measured (MI355X): 250K edges 0.22 vs 0.35 ms, 500K 0.33 vs 0.44, 16M 6.9 vs 1.9 (one-workgroup scan, atomics)
DIRECT_BUILD_MAX_EDGES = 2_000_000
_DIRECT_BUILD = True


def check_build_statuses(statuses) -> None:
    """Verify, with ONE host sync, the status words of builds whose read was skipped (``known_path``): a snapshot
    whose edge list changed since it was validated (an endpoint out of range, a row too long for the counting
    build) is reported here instead of never."""
    seen, uniq = set(), []
    for s in statuses:                     # the fused rebuilds of a device share ONE sticky word: read and report it once
        if s is not None and s.data_ptr() not in seen:
            seen.add(s.data_ptr())
            uniq.append(s)
    statuses = uniq
    if not statuses:
        return
    codes = torch.stack([s.reshape(()) for s in statuses]).cpu().tolist()
    bad = [c for c in codes if c != 0]
    if bad:
        for s_, c in zip(statuses, codes):
            if c != 0:
                s_.zero_()                 # a sticky word (fused rebuild) is shared by later builds: reported once, then clean
        # a batched build leaves the id of the FIRST list that failed in bits 8 .. 30 (stg_build_job::id; NaiveGraph: timestamp + 1)
        code, who = bad[0] & 0xff, bad[0] >> 8
        where = f" (first: build id {who}" + (f" = timestamp {who - 1} of a NaiveGraph" if who else "") + ")" if who else ""
        raise ValueError(f"{len(bad)} per-snapshot CSR build(s) whose validation was deferred failed "
                         f"(libstgraph_hip status {code}){where}: the edge list changed after it was first built")


def set_direct_build(enabled: bool) -> None:
    """True (default): device graphs are built by counting (stg_graph_build_direct_device), falling back to the
    sort-based builder when a row is longer than 2048 entries.  False: always the sort-based builder."""
    global _DIRECT_BUILD
    _DIRECT_BUILD = bool(enabled)


FUSED_REBUILD = True     # re-builds of a validated edge list go through stg_graph_build_direct2_device
BUILD_SLOTS = 16         # counter buffers per (device, |V|): that many rebuilds may be in flight (side streams, or the jobs of a batch)
BUILD_BATCH_MAX_NODES = 1 << 18     # batched rebuilds (and their 16 counter buffers of 8 |V| bytes) are for small snapshots; above: 4 slots
_BUILD_COUNTERS = {}
_BUILD_COUNTER_PINS = {}        # (device, N) -> number of live owners of captured graphs that hold raw pointers into its buffers
BUILD_COUNTER_SIZES_KEPT = 8    # distinct (device, |V|) whose counter buffers stay cached when nobody pins them


def pin_build_counters(owner, device, N: int) -> None:
    """``owner`` (an object whose captured HIP graphs hold RAW POINTERS into the counter buffers of (device, N): a replay adds into
    them assuming they are zero) keeps them alive: they are not evicted until the last such owner is garbage."""
    import weakref
    key = (str(device), int(N))
    _BUILD_COUNTER_PINS[key] = _BUILD_COUNTER_PINS.get(key, 0) + 1

    def _unpin(k=key):
        n = _BUILD_COUNTER_PINS.get(k, 0) - 1
        if n <= 0:
            _BUILD_COUNTER_PINS.pop(k, None)
        else:
            _BUILD_COUNTER_PINS[k] = n
    weakref.finalize(owner, _unpin)


def _build_counters(device, N: int, slot: int = 0):
    """(2 N zeroed counters, sticky status word) of ``device`` for stg_graph_build_direct2_device: the counters are zero
    between builds by that function's contract, so one buffer per (device, N) serves every rebuild on a stream; builds
    issued on CONCURRENT streams take different ``slot``s.  The slots of a (device, N) are made one at a time, on first use
    (8 N bytes each).  Buffers of the ``BUILD_COUNTER_SIZES_KEPT`` most recently used sizes stay cached; older sizes are dropped
    unless an owner of captured graphs pinned them (``pin_build_counters``) -- a process that builds graphs of many distinct |V|
    no longer grows without bound."""
    key = (str(device), int(N), int(slot))
    hit = _BUILD_COUNTERS.pop(key, None)
    if hit is None:
        hit = (torch.zeros(2 * ((max(N, 1) + 3) & ~3), dtype=torch.int32, device=device), _build_status_word(device))
        sizes = []
        for k in _BUILD_COUNTERS:                                  # insertion order = least recently used first
            if (k[0], k[1]) not in sizes:
                sizes.append((k[0], k[1]))
        unpinned = [sz for sz in sizes if sz not in _BUILD_COUNTER_PINS and sz != (key[0], key[1])]
        for sz in unpinned[:max(0, len(unpinned) - (BUILD_COUNTER_SIZES_KEPT - 1))]:
            for k in [k for k in _BUILD_COUNTERS if (k[0], k[1]) == sz]:
                del _BUILD_COUNTERS[k]
    _BUILD_COUNTERS[key] = hit                                     # (re-inserted: most recently used last)
    return hit


_BUILD_STATUS = {}


def _build_status_word(device) -> torch.Tensor:
    """ONE sticky status word per device for every direct build (whatever |V| or slot): `check_build_status` reads it."""
    key = str(device)
    w = _BUILD_STATUS.get(key)
    if w is None:
        w = _BUILD_STATUS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def build_graph_csr_batch(edge_lists, num_nodes: int, device: torch.device | str, counters_base: int = 0, ids=None) -> list:
    """The re-builds of SEVERAL validated edge lists over the same vertex set -- the snapshots of a BPTT window -- in the
    launches of one (stg_graph_build_direct2_batch_device; <= _C.BUILD_BATCH_MAX lists).  Every GraphCSR is bit-identical
    to ``build_graph_csr(s, d, N, device, lazy_node_ids=True, known_path='direct')`` of its list.
    ``counters_base``: first counter slot of the batch (``_build_counters``); a build issued on a second stream beside builds of the
    same |V| on another takes a disjoint range (temporal.CapturedDynamicWindows: ``_C.BUILD_BATCH_MAX``).  ``ids``: a name per list
    (1 .. 2^23 - 1; NaiveGraph: timestamp + 1) that the shared sticky status word carries when THAT list fails its validation
    (:func:`check_build_statuses` reports it)."""
    device = torch.device(device)
    N = int(num_nodes)
    n = len(edge_lists)
    if not 0 < n <= _C.BUILD_BATCH_MAX or not 0 < N <= BUILD_BATCH_MAX_NODES or device.type != "cuda":
        raise ValueError("build_graph_csr_batch: 1 .. %d edge lists over 1 .. %d vertices on a GPU" % (_C.BUILD_BATCH_MAX, BUILD_BATCH_MAX_NODES))
    i32 = dict(dtype=torch.int32, device=device)
    jobs = (_C.BuildJob * n)()
    out, keep = [], []
    sticky = _build_counters(device, N, 0)[1]
    for k, (src, dst) in enumerate(edge_lists):
        s, d = _as_i32(src, device), _as_i32(dst, device)
        if s.dim() != 1 or s.shape != d.shape:
            raise ValueError("src and dst must be 1-D arrays of equal length")
        E = int(s.shape[0])
        if E == 0 or E > DIRECT_BUILD_MAX_EDGES:
            raise ValueError("build_graph_csr_batch: every list needs 1 .. DIRECT_BUILD_MAX_EDGES edges")
        perm = torch.empty(E, dtype=torch.int64, device=device)
        indeg, outdeg = torch.empty(N, **i32), torch.empty(N, **i32)
        fwd = DeviceCSR(torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32), None, False, indeg)
        bwd = DeviceCSR(torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32), None, False, outdeg)
        norm = torch.empty(N, 1, dtype=torch.float32, device=device)
        nc_f = torch.empty(E, dtype=torch.float32, device=device)
        nc_b = torch.empty(E, dtype=torch.float32, device=device)
        ws_bytes = int(_C.lib.stg_graph_build_direct_workspace_bytes(E, N))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        counters = _build_counters(device, N, int(counters_base) + k)[0]
        j = jobs[k]
        j.src, j.dst, j.E = _ptr(s), _ptr(d), E
        j.perm_fwd = _ptr(perm)
        j.fwd_row_offset, j.fwd_column_indices, j.fwd_eids = _ptr(fwd.row_offset), _ptr(fwd.column_indices), _ptr(fwd.eids)
        j.bwd_row_offset, j.bwd_column_indices, j.bwd_eids = _ptr(bwd.row_offset), _ptr(bwd.column_indices), _ptr(bwd.eids)
        j.in_degrees, j.out_degrees = _ptr(indeg), _ptr(outdeg)
        j.norm, j.norm_col_fwd, j.norm_col_bwd = _ptr(norm), _ptr(nc_f), _ptr(nc_b)
        j.zero_counters, j.workspace, j.workspace_bytes = _ptr(counters), _ptr(ws), ws_bytes
        j.id = 0 if ids is None else int(ids[k]) & 0x7fffff
        keep.append((s, d, ws))
        g = GraphCSR(N, fwd, bwd, indeg, outdeg, perm)
        g.built_by = "direct"
        g.unchecked_status = sticky
        g.norm_in = norm
        fwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_f)}
        bwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_b)}
        out.append(g)
    with torch.cuda.device(device):
        _C.check(_C.lib.stg_graph_build_direct2_batch_device(jobs, n, N, _ptr(sticky), _stream_ptr(device)))
    return out


def build_graph_csr(src, dst, num_nodes: int, device: torch.device | str, lazy_node_ids: bool = False,
                    known_path: str | None = None, counters_slot: int = 0) -> GraphCSR:
    """Build both CSRs of a graph from (src, dst) arrays (static_graph.py:40-78).

    ``device`` cuda -> the direct (counting) build for small graphs, the sort-based build otherwise or when a
    row is too long (stream ordered); ``device`` cpu -> stg_graph_build_host (host arrays; used for host logic
    tests and as the upload source the reference itself uses).

    The device builds report through a status word whose read is the one host sync of a build (endpoint
    validation, long-row verdict).  ``known_path`` ('direct' | 'sort', from ``GraphCSR.built_by`` of an earlier,
    validated build of the SAME edge list -- a dynamic graph rebuilding a snapshot every epoch) skips that read.
    """
    device = torch.device(device)
    N = int(num_nodes)
    s, d = _as_i32(src, device), _as_i32(dst, device)
    if s.dim() != 1 or s.shape != d.shape:
        raise ValueError("src and dst must be 1-D arrays of equal length")
    E = int(s.shape[0])
    if N < 0:
        raise ValueError("num_nodes must be >= 0")
    i32 = dict(dtype=torch.int32, device=device)
    perm = torch.empty(E, dtype=torch.int64, device=device)
    indeg, outdeg = torch.empty(N, **i32), torch.empty(N, **i32)
    # ``lazy_node_ids`` (honoured by the direct device build only): leave the two degree sorts to first use
    lazy = bool(lazy_node_ids) and device.type == "cuda" and _DIRECT_BUILD and E <= DIRECT_BUILD_MAX_EDGES
    nid_f, nid_b = (None, None) if lazy else (torch.empty(N, **i32), torch.empty(N, **i32))
    fwd = DeviceCSR(torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32), nid_f, not lazy, indeg)
    bwd = DeviceCSR(torch.empty(N + 1, **i32), torch.empty(E, **i32), torch.empty(E, **i32), nid_b, not lazy, outdeg)
    arrays = [perm, fwd.row_offset, fwd.column_indices, fwd.eids, nid_f,
              bwd.row_offset, bwd.column_indices, bwd.eids, nid_b, indeg, outdeg]
    built_by = "host"
    if device.type == "cuda":
        status = torch.empty(1, **i32)
        code = BUILD_NEEDS_SORT
        built_by = "sort"
        direct_ok = _DIRECT_BUILD and E <= DIRECT_BUILD_MAX_EDGES
        if direct_ok and not torch.cuda.is_current_stream_capturing():
            for slot in range(BUILD_SLOTS if N <= BUILD_BATCH_MAX_NODES else 4):     # made outside any capture: a later captured rebuild finds them
                _build_counters(device, N, slot)
        if known_path == "direct" and lazy == bool(lazy_node_ids) and direct_ok:
            ws_bytes = int(_C.lib.stg_graph_build_direct_workspace_bytes(E, N))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            if FUSED_REBUILD and N > 0 and E > 0:
                # five launches, one atomic pass; norm = in_deg ** -0.5 and its per-edge gathers ride along (what the
                # dynamic loop computes from every new snapshot); status: the device's sticky word
                counters, sticky = _build_counters(device, N, counters_slot)
                norm = torch.empty(N, 1, dtype=torch.float32, device=device)
                nc_f = torch.empty(E, dtype=torch.float32, device=device)
                nc_b = torch.empty(E, dtype=torch.float32, device=device)
                with torch.cuda.device(device):
                    _C.check(_C.lib.stg_graph_build_direct2_device(
                        _ptr(s), _ptr(d), E, N, *[_ptr(a) for a in arrays], _ptr(norm), _ptr(nc_f), _ptr(nc_b),
                        _ptr(counters), _ptr(sticky), _ptr(ws), ws_bytes, _stream_ptr(device)))
                g = GraphCSR(N, fwd, bwd, indeg, outdeg, perm)
                g.built_by = "direct"
                g.unchecked_status = sticky
                g.norm_in = norm
                fwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_f)}
                bwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_b)}
                return g
            with torch.cuda.device(device):
                _C.check(_C.lib.stg_graph_build_direct_device(
                    _ptr(s), _ptr(d), E, N, *[_ptr(a) for a in arrays],
                    _ptr(status), _ptr(ws), ws_bytes, _stream_ptr(device)))
            g = GraphCSR(N, fwd, bwd, indeg, outdeg, perm)
            g.built_by = "direct"
            g.unchecked_status = status            # the caller verifies it later, in bulk (check_build_statuses)
            return g
        if known_path != "sort" and _DIRECT_BUILD and E <= DIRECT_BUILD_MAX_EDGES:      # counting build: 6 launches (+ node_ids)
            ws_bytes = int(_C.lib.stg_graph_build_direct_workspace_bytes(E, N))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _C.check(_C.lib.stg_graph_build_direct_device(
                    _ptr(s), _ptr(d), E, N, *[_ptr(a) for a in arrays],
                    _ptr(status), _ptr(ws), ws_bytes, _stream_ptr(device)))
            code = int(status.item())      # one 4-byte sync per graph build: endpoint validation / long-row verdict
            built_by = "direct"
        if code & BUILD_NEEDS_SORT:
            built_by = "sort"        # a row longer than 2048 entries (or the direct path is off): sort-based build
            if lazy:
                fwd.node_ids, bwd.node_ids = torch.empty(N, **i32), torch.empty(N, **i32)
                fwd.degree_sorted = bwd.degree_sorted = True
                arrays[4], arrays[8] = fwd.node_ids, bwd.node_ids
            ws_bytes = int(_C.lib.stg_graph_build_device_workspace_bytes(E, N))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _C.check(_C.lib.stg_graph_build_device(
                    _ptr(s), _ptr(d), E, N, *[_ptr(a) for a in arrays],
                    _ptr(status), _ptr(ws), ws_bytes, _stream_ptr(device)))
            code = 0 if known_path == "sort" else int(status.item())
        if code != 0:
            raise ValueError(f"edge endpoint outside [0, {N}) (libstgraph_hip status {code})")
    else:
        _C.check(_C.lib.stg_graph_build_host(_ptr(s), _ptr(d), E, N, *[_ptr(a) for a in arrays]))
    g = GraphCSR(N, fwd, bwd, indeg, outdeg, perm)
    g.built_by = built_by
    return g


def csr_ctor_host(a, b, eid, edge_weight, num_nodes: int, is_edge_reverse: bool = False):
    """Host counterpart of the pybind ``CSR`` constructor (csr.cu:68-157); numpy in/out."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    eid = np.ascontiguousarray(eid, dtype=np.int32)
    E, N = int(a.shape[0]), int(num_nodes)
    ew = None if edge_weight is None else np.ascontiguousarray(edge_weight, dtype=np.float32)
    if ew is not None and ew.shape[0] < E:
        raise ValueError("edge_weight shorter than the edge list")
    out = dict(
        row_offset=np.empty(N + 1, np.int32), column_indices=np.empty(E, np.int32),
        eids=np.empty(E, np.int32), node_ids=np.empty(N, np.int32),
        in_degrees=np.empty(N, np.int32), out_degrees=np.empty(N, np.int32),
        weighted_out_degrees=np.empty(N, np.float32))
    p = lambda x: ctypes.c_void_p(0 if x is None else x.ctypes.data)  # noqa: E731
    _C.check(_C.lib.stg_csr_ctor_host(p(a), p(b), p(eid), p(ew), E, N, int(bool(is_edge_reverse)),
                                      *[p(v) for v in out.values()]))
    return out


# ---------------------------------------------------------------------- dynamic edge store
@dataclass
class EdgeSet:
    """The edge set of one timestamp of a dynamic graph: two ascending arrays of packed keys
    (``dst << 32 | src`` and ``src << 32 | dst``; int64 tensors).  The state behind
    :class:`stgraph_amd.graph.dynamic.pcsr.pcsr.PCSR` (reference pcsr.cu:273-318).  Immutable:
    an update returns a new set, so caching a timestamp is keeping a reference."""

    num_nodes: int
    keys_fwd: torch.Tensor
    keys_bwd: torch.Tensor
    status: torch.Tensor | None = None      # int32[1] written by the update that produced this set

    @property
    def num_edges(self) -> int:
        return int(self.keys_fwd.shape[0])

    @property
    def device(self) -> torch.device:
        return self.keys_fwd.device


_STATUS_TEXT = {1: "vertex id out of range", 2: "added edge already present (or repeated in the batch)",
                4: "deleted edge absent (or repeated in the batch)", 8: "edge added and deleted in the same update",
                16: "pre-sorted batch is not ascending"}


def edgeset_empty(num_nodes: int, device) -> EdgeSet:
    device = torch.device(device)
    z = torch.empty(0, dtype=torch.int64, device=device)
    return EdgeSet(int(num_nodes), z, z.clone())


def edgeset_update(es: EdgeSet, add_src, add_dst, del_src=None, del_dst=None) -> EdgeSet:
    """(es \\ del) U add, both orientations (stg_edgeset_update_device / _host).  The stream contract
    (added edges absent, deleted edges present) is checked on the device; call
    :func:`edgeset_check` to fetch the verdict (one 4-byte sync)."""
    device, N = es.device, es.num_nodes
    a_s, a_d = _as_i32(add_src, device), _as_i32(add_dst, device)
    empty = torch.empty(0, dtype=torch.int32, device=device)
    d_s = _as_i32(del_src, device) if del_src is not None else empty
    d_d = _as_i32(del_dst, device) if del_dst is not None else empty
    if a_s.shape != a_d.shape or d_s.shape != d_d.shape or a_s.dim() != 1 or d_s.dim() != 1:
        raise ValueError("update lists must be 1-D (src, dst) arrays of equal length")
    E, na, nd = es.num_edges, int(a_s.shape[0]), int(d_s.shape[0])
    if E + na - nd < 0:
        raise ValueError("more deletions than edges")
    kf = torch.empty(E + na - nd, dtype=torch.int64, device=device)
    kb = torch.empty(E + na - nd, dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    if device.type == "cuda":
        ws_bytes = int(_C.lib.stg_edgeset_update_workspace_bytes(na, nd))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _C.check(_C.lib.stg_edgeset_update_device(
                _ptr(es.keys_fwd), _ptr(es.keys_bwd), E, _ptr(a_s), _ptr(a_d), na, _ptr(d_s), _ptr(d_d), nd, N,
                _ptr(kf), _ptr(kb), _ptr(status), _ptr(ws), ws_bytes, _stream_ptr(device)))
    else:
        _C.check(_C.lib.stg_edgeset_update_host(
            _ptr(es.keys_fwd), _ptr(es.keys_bwd), E, _ptr(a_s), _ptr(a_d), na, _ptr(d_s), _ptr(d_d), nd, N,
            _ptr(kf), _ptr(kb), _ptr(status)))
    return EdgeSet(N, kf, kb, status)


def edgeset_pack_sorted(src, dst, device):
    """An update batch packed and sorted in both orientations (int64 key tensors), ready for
    :func:`edgeset_merge`.  Done once per batch (a PCSRGraph does it at construction)."""
    device = torch.device(device)
    s, d = _as_i32(src, device).long(), _as_i32(dst, device).long()
    return torch.sort((d << 32) | s).values, torch.sort((s << 32) | d).values


def edgeset_merge(es: EdgeSet, add_keys, del_keys) -> EdgeSet:
    """(es \\ del) U add from batches already packed + sorted (``edgeset_pack_sorted``): two scatter
    passes (stg_edgeset_merge_device), no sort, no workspace."""
    device, N, E = es.device, es.num_nodes, es.num_edges
    if device.type != "cuda":
        raise ValueError("edgeset_merge is the device fast path; use edgeset_update on host arrays")
    na, nd = int(add_keys[0].shape[0]), int(del_keys[0].shape[0])
    if E + na - nd < 0:
        raise ValueError("more deletions than edges")
    kf = torch.empty(E + na - nd, dtype=torch.int64, device=device)
    kb = torch.empty(E + na - nd, dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        st = _stream_ptr(device)
        _C.check(_C.lib.stg_edgeset_merge_device(_ptr(es.keys_fwd), E, _ptr(add_keys[0]), na, _ptr(del_keys[0]), nd,
                                                 _ptr(kf), _ptr(status), st))
        _C.check(_C.lib.stg_edgeset_merge_device(_ptr(es.keys_bwd), E, _ptr(add_keys[1]), na, _ptr(del_keys[1]), nd,
                                                 _ptr(kb), _ptr(status), st))
    return EdgeSet(N, kf, kb, status)


class EmissionQueue:
    """At most one store-step emission left pending (stg_store_emission + the tensors it points into), shared by the copies of
    a store.  While ``defer`` is set :func:`edgeset_step` leaves its emission here and carries the one it finds in its merge
    launch -- one launch per step; whoever set ``defer`` calls :meth:`flush` before anything reads the emitted columns."""

    def __init__(self):
        self.defer = False
        self.pending = None             # (StoreEmission, device, tensors kept alive)

    def flush(self) -> None:
        p, self.pending = self.pending, None
        if p is not None:
            with torch.cuda.device(p[1]):
                _C.check(_C.lib.stg_edgeset_emit_pending_device(ctypes.byref(p[0]), _stream_ptr(p[1])))


def edgeset_step(es: EdgeSet, add_keys, del_keys, key_order: bool, status: torch.Tensor | None = None, old_row_offsets=None,
                 queue: "EmissionQueue | None" = None):
    """One timestamp of a delta store in two launches (three without ``old_row_offsets``; stg_edgeset_step_device): returns ``(new set, forward StoreCSR,
    backward StoreCSR, norm [N, 1])`` -- the merge of :func:`edgeset_merge`, both emissions of :func:`edgeset_emit_csr`, the
    in-degrees, ``norm = in_deg ** -0.5`` (:func:`degree_norm`'s values) and ``norm`` gathered per edge of either CSR,
    already filed in the CSRs' per-edge caches under the returned ``norm`` tensor.  ``status``: the store's sticky status
    word (OR-ed into; a fresh zero word if None).  ``old_row_offsets``: ``(forward, backward)`` row offsets of ``es`` as an earlier
    step emitted them (EXACTLY those of ``es``: the new row offsets, degrees and norm are derived from them and the batches), or None."""
    device, N, E = es.device, es.num_nodes, es.num_edges
    if device.type != "cuda":
        raise ValueError("edgeset_step is the device fast path")
    na, nd = int(add_keys[0].shape[0]), int(del_keys[0].shape[0])
    E_out = E + na - nd
    if E_out < 0:
        raise ValueError("more deletions than edges")
    i32 = dict(dtype=torch.int32, device=device)
    kf = torch.empty(E_out, dtype=torch.int64, device=device)
    kb = torch.empty(E_out, dtype=torch.int64, device=device)
    if status is None:
        status = torch.zeros(1, **i32)
    ro_f, ro_b = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
    col_f, col_b = torch.empty(E_out, **i32), torch.empty(E_out, **i32)
    deg = torch.empty(N, **i32)
    norm = torch.empty(N, 1, dtype=torch.float32, device=device)
    nc_f = torch.empty(E_out, dtype=torch.float32, device=device)
    nc_b = torch.empty(E_out, dtype=torch.float32, device=device)
    args = (_ptr(es.keys_fwd), _ptr(es.keys_bwd), E, _ptr(add_keys[0]), _ptr(add_keys[1]), na, _ptr(del_keys[0]),
            _ptr(del_keys[1]), nd, N, EMIT_KEY_ORDER if key_order else 0, _ptr(kf), _ptr(kb), _ptr(ro_f), _ptr(col_f),
            _ptr(ro_b), _ptr(col_b), _ptr(deg), _ptr(norm), _ptr(nc_f), _ptr(nc_b),
            _ptr(old_row_offsets[0]) if old_row_offsets else None, _ptr(old_row_offsets[1]) if old_row_offsets else None)
    with torch.cuda.device(device):
        if queue is not None and queue.defer:
            # this step's emission stays pending (queue); the one found there rides in this step's merge launch
            carry, mine = queue.pending, _C.StoreEmission()
            if carry is not None and carry[1] != device:
                queue.flush()
                carry = None
            _C.check(_C.lib.stg_edgeset_step_deferred_device(*args, ctypes.byref(carry[0]) if carry is not None else None,
                                                             ctypes.byref(mine), _ptr(status), _stream_ptr(device)))
            queue.pending = (mine, device, (kf, kb, ro_f, ro_b, col_f, col_b, norm, nc_f, nc_b))
        else:
            if queue is not None:
                queue.flush()
            _C.check(_C.lib.stg_edgeset_step_device(*args, _ptr(status), _stream_ptr(device)))
    new = EdgeSet(N, kf, kb, status)
    fwd = StoreCSR(new, False, ro_f, col_f, None, deg, key_order)
    bwd = StoreCSR(new, True, ro_b, col_b, None, None, key_order)
    fwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_f)}
    bwd.__dict__["_edge_cache"] = {"norm": (norm, norm._version, nc_b)}
    return new, fwd, bwd, norm


def edgeset_check(es: EdgeSet) -> None:
    """Raise if the update that produced ``es`` violated the stream contract (synchronises)."""
    if es.status is None:
        return
    code = int(es.status.item())
    if code:
        what = "; ".join(t for b, t in _STATUS_TEXT.items() if code & b)
        raise ValueError(f"invalid edge update stream: {what} (libstgraph_hip status {code})")


class StoreCSR(DeviceCSR):
    """CSR emitted from an :class:`EdgeSet`.  ``row_offset`` and ``column_indices`` are emitted eagerly (two
    launches); everything else on first access: ``degrees`` (row lengths), ``node_ids`` (a sort of the degrees --
    only a processing order, ~10 launches a training step on a snapshot never needs), and the edge labels ``eids``
    (0-based, what the launch wrappers index edge tensors with) / ``eids1`` (the reference's 1-based array), which
    the un-weighted GCN kernels never read and which cost a search per edge for the reverse CSR."""

    def __init__(self, es: EdgeSet, reverse: bool, row_offset, column_indices, node_ids=None, degrees=None,
                 key_order: bool = False):
        self.row_offset, self.column_indices, self._node_ids, self._degrees = row_offset, column_indices, node_ids, degrees
        self.degree_sorted = node_ids is not None
        self._es, self._reverse, self._labels, self._key_order = es, bool(reverse), None, bool(key_order)

    @property
    def degrees(self) -> torch.Tensor:
        if self._degrees is None:
            self._degrees = (self.row_offset[1:] - self.row_offset[:-1]).contiguous()
        return self._degrees

    def _emit_labels(self):
        if self._labels is None:
            E = self._es.num_edges
            i32 = dict(dtype=torch.int32, device=self.row_offset.device)
            e1, e0 = torch.empty(E, **i32), torch.empty(E, **i32)
            _emit(self._es, self._reverse, self.row_offset, None, e1, e0, None, None, self._key_order)
            self._labels = (e1, e0)
        return self._labels

    @property
    def keys(self) -> torch.Tensor:
        """The packed 64-bit keys in row order (``row << 32 | column``) -- what the reference's GPMA kernels
        take as ``column_indices`` (tpl_fa_gpma.jinja:5,30-37).  Key-order layout only: there the emitted
        CSR *is* the key array, so this is the store's own tensor, not a copy."""
        if not self._key_order:
            raise ValueError("packed keys follow key order; this CSR was emitted back to front (PCSR layout)")
        return self._es.keys_bwd if self._reverse else self._es.keys_fwd

    @property
    def eids(self) -> torch.Tensor:
        return self._emit_labels()[1]

    @property
    def eids1(self) -> torch.Tensor:
        return self._emit_labels()[0]


EMIT_REVERSE, EMIT_KEY_ORDER = 1, 2          # include/stgraph_hip.h STG_EMIT_*


def _emit(es: EdgeSet, reverse: bool, row_offset, col, eids1, eids0, node_ids, degrees, key_order: bool = False) -> None:
    device, N, E = es.device, es.num_nodes, es.num_edges
    flags = (EMIT_REVERSE if reverse else 0) | (EMIT_KEY_ORDER if key_order else 0)
    args = [_ptr(es.keys_fwd), _ptr(es.keys_bwd), E, N, flags, _ptr(row_offset), _ptr(col),
            _ptr(eids1), _ptr(eids0), _ptr(node_ids), _ptr(degrees)]
    if device.type == "cuda":
        ws_bytes = int(_C.lib.stg_edgeset_emit_csr_workspace_bytes(N))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _C.check(_C.lib.stg_edgeset_emit_csr_device(*args, _ptr(ws), ws_bytes, _stream_ptr(device)))
    else:
        _C.check(_C.lib.stg_edgeset_emit_csr_host(*args))


def edgeset_emit_csr(es: EdgeSet, reverse: bool, key_order: bool = False) -> StoreCSR:
    """The CSR the reference's ``build_csr`` (``reverse=False``, rows = dst) / ``build_reverse_csr``
    (rows = src) emits for this edge set (pcsr.cu:781-879), as a :class:`StoreCSR`.  ``key_order=True``:
    the GPMA view instead (rows and columns ascending, gpma.cu:1121-1188) -- see include/stgraph_hip.h."""
    device, N, E = es.device, es.num_nodes, es.num_edges
    i32 = dict(dtype=torch.int32, device=device)
    ro, col = torch.empty(N + 1, **i32), torch.empty(E, **i32)
    _emit(es, reverse, ro, col, None, None, None, None, key_order)
    return StoreCSR(es, reverse, ro, col, None, None, key_order)


def rows_by_node_ids(graph_type: str) -> bool:
    """Graph types whose kernels visit rows through ``node_ids`` (tpl_fa_csr.jinja / tpl_fa_pcsr.jinja /
    tpl_fa_gpma.jinja, code_gen.py:96-107); the '*_unsorted' types walk rows in id order."""
    return graph_type in ("csr", "pcsr", "gpma")


# ------------------------------------------------------------------------------- GCN
_EDGE_CACHE = True
_LONG_ROWS = True


def set_long_row_path(enabled: bool) -> None:
    """On (default): long rows are summed by wave-per-row, LDS-staged workgroups when a row is narrower than a
    wave (see stg_gcn_agg_edge in include/stgraph_hip.h).  Off: every row goes through the row-group mapping."""
    global _LONG_ROWS
    _LONG_ROWS = bool(enabled)


def set_edge_cache(enabled: bool) -> None:
    """Per-CSR cache of the per-edge scalars in CSR order (norm[col[e]], w[eid[e]]); on by default.
    Off: every launch gathers them itself (stg_gcn_agg), as the reference's kernels do."""
    global _EDGE_CACHE
    _EDGE_CACHE = bool(enabled)


def _edge_gathered(csr: "DeviceCSR", kind: str, table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """table[idx] in CSR order, cached on the CSR object.  The cache entry keeps a reference to the
    source tensor and its version counter, so an in-place update or a new tensor invalidates it and
    a freed tensor's address can never be mistaken for a live one."""
    cache = csr.__dict__.setdefault("_edge_cache", {})
    hit = cache.get(kind)
    if hit is not None and hit[0] is table and hit[1] == table._version:
        return hit[2]
    dst = torch.empty(csr.num_edges, dtype=torch.float32, device=table.device)
    _C.check(_C.lib.stg_edge_gather_f32(_ptr(dst), _ptr(table), _ptr(idx), csr.num_edges,
                                        _stream_ptr(table.device)))
    cache[kind] = (table, table._version, dst)
    return dst


ACT_NONE, ACT_RELU = 0, 1                     # include/stgraph_hip.h STG_ACT_*


def layer_epilogue_usable() -> bool:
    """The bias/activation epilogue rides on the pre-gathered-scalar kernel and writes every column."""
    return _EDGE_CACHE and not reference_compat()


HUB_THRESHOLD = 1024       # rows above it are "hubs" at F >= WIDE_ROW_MIN_F (stg_gcn_agg_edge2's hub plan)
WIDE_ROW_MIN_F = 128       # the narrowest row the kernels map to a whole wave (narrower rows have their own long-row path)


def _hub_plan(csr: "DeviceCSR"):
    """(rows with >= 8192, 2048 .. 8191, HUB_THRESHOLD + 1 .. 2047 edges) of ``csr``, counted once per CSR object (one host
    sync; (0, 0, 0) while a stream capture is running and the count is not known yet: the main launch then takes every row)."""
    plan = csr.__dict__.get("_hub_plan")
    if plan is None:
        if torch.cuda.is_current_stream_capturing():
            return (0, 0, 0)
        deg = csr.row_offset[1:] - csr.row_offset[:-1]
        counts = torch.stack([(deg >= 8192).sum(), ((deg >= 2048) & (deg < 8192)).sum(),
                              ((deg > HUB_THRESHOLD) & (deg < 2048)).sum()]).tolist()
        plan = csr.__dict__["_hub_plan"] = tuple(int(c) for c in counts)
    return plan


def gcn_agg(x: torch.Tensor, norm_row: torch.Tensor, norm_col: torch.Tensor, csr: DeviceCSR,
            ew: torch.Tensor | None = None, use_node_ids: bool = False,
            f_active: int | None = None, bias: torch.Tensor | None = None, act: int = ACT_NONE) -> torch.Tensor:
    """out[r,:] = norm_row[r] * sum_e (norm_col[c] * x[c,:]) * ew[eid]   (stg_gcn_agg);
    with ``bias`` / ``act``: ``act(out + bias)`` in the same launch (stg_gcn_layer_fwd)."""
    x = _f32(x, "x")
    dev = x.device
    N = csr.num_nodes
    if x.dim() < 1 or x.shape[0] != N:
        raise ValueError(f"x has {x.shape[0] if x.dim() else 0} rows, graph has {N} nodes")
    F = int(x[0].numel()) if N > 0 else int(np.prod(x.shape[1:]))
    if F <= 0:
        raise ValueError("empty feature dimension")
    norm_row, norm_col = _f32(norm_row, "norm_row", dev), _f32(norm_col, "norm_col", dev)
    if norm_row.numel() != N or norm_col.numel() != N:
        raise ValueError("norm tensors must hold one value per node")
    if csr.row_offset.device != dev:
        raise RuntimeError(f"graph arrays are on {csr.row_offset.device}, features on {dev}")
    if ew is not None:
        ew = _f32(ew, "edge_weight", dev)
        if ew.numel() < csr.num_edges:
            raise ValueError(f"edge_weight has {ew.numel()} entries, graph has {csr.num_edges} edges")
    fa = F if f_active is None else int(f_active)
    epilogue = bias is not None or act != ACT_NONE
    if epilogue:
        if not _EDGE_CACHE or fa != F:
            raise ValueError("the bias/activation epilogue needs the edge cache and every column active")
        if bias is not None:
            bias = _f32(bias, "bias", dev)
            if bias.numel() != F:
                raise ValueError(f"bias has {bias.numel()} entries, rows have {F}")
    out = (torch.empty_like(x) if fa == F else torch.zeros_like(x))
    nid = _ptr(csr.node_ids_if_ready if use_node_ids else None)     # an order hint: skipped if not sorted yet
    with torch.cuda.device(dev):
        if _EDGE_CACHE:
            nc_e = _edge_gathered(csr, "norm", norm_col, csr.column_indices)
            ew_e = None if ew is None else _edge_gathered(csr, "ew", ew, csr.eids)
        with _Timed("gcn_agg", gcn_agg_algorithmic_bytes(N, csr.num_edges, fa, ew is not None), csr.num_edges * fa):
            if _EDGE_CACHE:
                rbd = csr.node_ids_if_ready if (_LONG_ROWS and csr.degree_sorted) else None
                plan = (0, 0, 0)
                if rbd is not None and fa >= WIDE_ROW_MIN_F:
                    # rows of a wave and wider: the hubs' workgroups are launched from a per-graph count of the long rows
                    # (one sync per CSR object, never inside a capture); without it, or without hubs, one launch for all
                    plan = _hub_plan(csr)
                _C.check(_C.lib.stg_gcn_agg_edge2(
                    _ptr(x), _ptr(norm_row), _ptr(nc_e), _ptr(ew_e), _ptr(bias) if epilogue else None, int(act) if epilogue else ACT_NONE,
                    _ptr(out), _ptr(csr.row_offset), _ptr(csr.column_indices), nid, _ptr(rbd), N, csr.num_edges, F, fa,
                    HUB_THRESHOLD, plan[0], plan[1], plan[2], _stream_ptr(dev)))
            else:
                _C.check(_C.lib.stg_gcn_agg(
                    _ptr(x), _ptr(norm_row), _ptr(norm_col), _ptr(ew), _ptr(out),
                    _ptr(csr.row_offset), _ptr(csr.column_indices), _ptr(csr.eids), nid, N, F, fa,
                    _stream_ptr(dev)))
    return out


def bias_act_fwd_(y: torch.Tensor, bias: torch.Tensor | None, act: int) -> torch.Tensor:
    """``y = act(y + bias)`` in place, one pass (stg_bias_act_fwd)."""
    if y.dtype != torch.float32 or not y.is_cuda or not y.is_contiguous() or y.dim() != 2:
        raise ValueError("y must be a contiguous 2-D fp32 device tensor")
    dev = y.device
    N, F = y.shape
    b = None if bias is None else _f32(bias, "bias", dev)
    if b is not None and b.numel() != F:
        raise ValueError("bias must have one entry per column")
    with torch.cuda.device(dev), _Timed("bias_act_fwd", 8 * N * F, N * F):
        _C.check(_C.lib.stg_bias_act_fwd(_ptr(y), _ptr(b), int(act), N, F, _stream_ptr(dev)))
    return y


def bias_act_bwd(g: torch.Tensor, out: torch.Tensor | None, want_colsum: bool = True):
    """Backward of ``act(y + bias)`` in one pass (stg_bias_act_bwd): returns ``(g_act, colsum)`` with
    ``g_act = g * (out > 0)`` when ``out`` (the ReLU output) is given, else ``g`` itself, and ``colsum`` =
    the bias gradient ``g_act.sum(0)`` (None unless asked for)."""
    g = _f32(g, "g")
    dev = g.device
    N = int(g.shape[0])
    F = int(g[0].numel()) if N > 0 else int(np.prod(g.shape[1:]))
    if out is not None:
        out = _f32(out, "out", dev)
        if out.shape != g.shape:
            raise ValueError("out and g must have the same shape")
    g_act = torch.empty_like(g) if out is not None else None
    colsum = torch.empty(F, dtype=torch.float32, device=dev) if want_colsum else None
    ws_bytes = int(_C.lib.stg_bias_act_bwd_workspace_bytes(N, F)) if want_colsum else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    nbytes = 4 * N * F * (1 + 2 * (out is not None))
    with torch.cuda.device(dev), _Timed("bias_act_bwd", nbytes, N * F):
        _C.check(_C.lib.stg_bias_act_bwd(_ptr(g), _ptr(out), _ptr(g_act), _ptr(colsum), N, F, _ptr(ws), ws_bytes,
                                         _stream_ptr(dev)))
    return (g_act if out is not None else g), colsum


def agg_transform_supported(fin: int, fout: int, big_lds: bool = False) -> bool:
    """``big_lds``: also shapes whose weight needs more than 64 KB of LDS (one workgroup per CU: only for measurements)."""
    return fin % 4 == 0 and fin >= 16 and fout % 32 == 0 and 4 * (64 * (fin + 1) + fin * fout) <= (150 if big_lds else 64) * 1024


def gcn_agg_transform(x: torch.Tensor, W: torch.Tensor, norm_row: torch.Tensor, norm_col: torch.Tensor,
                      csr: DeviceCSR, ew: torch.Tensor | None = None, use_node_ids: bool = False,
                      want_p: bool = True):
    """(out, P) with P = A_hat x (as gcn_agg) and out = P @ W, in one launch (stg_gcn_agg_transform)."""
    x = _f32(x, "x")
    dev = x.device
    W = _f32(W, "W", dev)
    N = csr.num_nodes
    if x.dim() != 2 or x.shape[0] != N or W.dim() != 2 or W.shape[0] != x.shape[1]:
        raise ValueError(f"gcn_agg_transform: x {tuple(x.shape)} / W {tuple(W.shape)} do not match the graph ({N} nodes)")
    fin, fout = int(W.shape[0]), int(W.shape[1])
    if not agg_transform_supported(fin, fout, big_lds=True):
        raise ValueError(f"gcn_agg_transform does not cover {fin} -> {fout}")
    norm_row, norm_col = _f32(norm_row, "norm_row", dev), _f32(norm_col, "norm_col", dev)
    if norm_row.numel() != N or norm_col.numel() != N or csr.row_offset.device != dev:
        raise ValueError("norm tensors must hold one value per node and live with the graph")
    if ew is not None:
        ew = _f32(ew, "edge_weight", dev)
        if ew.numel() < csr.num_edges:
            raise ValueError("edge_weight shorter than the edge list")
    out = torch.empty(N, fout, dtype=torch.float32, device=dev)
    P = torch.empty(N, fin, dtype=torch.float32, device=dev) if want_p else None
    E = csr.num_edges
    with torch.cuda.device(dev):
        nc_e = _edge_gathered(csr, "norm", norm_col, csr.column_indices)
        ew_e = None if ew is None else _edge_gathered(csr, "ew", ew, csr.eids)
        nbytes = gcn_agg_algorithmic_bytes(N, E, fin, ew is not None) + 4 * N * fout + 4 * fin * fout
        with _Timed("gcn_agg_transform", nbytes, E * fin):
            _C.check(_C.lib.stg_gcn_agg_transform(
                _ptr(x), _ptr(norm_row), _ptr(nc_e), _ptr(ew_e), _ptr(W), _ptr(out), _ptr(P),
                _ptr(csr.row_offset), _ptr(csr.column_indices), _ptr(csr.node_ids_if_ready if use_node_ids else None),
                N, fin, fout, _stream_ptr(dev)))
    return out, P


# ------------------------------------------------------------------------------- GAT
_GAT_ONES = True


def set_gat_ones_shortcut(on: bool) -> None:
    """False: the layers' GAT units always materialise and read the per-edge scores A (tests)."""
    global _GAT_ONES
    _GAT_ONES = bool(on)


def gat_fwd(el: torch.Tensor, er: torch.Tensor, feat: torch.Tensor, csr: DeviceCSR,
            slope: float, use_node_ids: bool = False, ones_shortcut: bool = False):
    """Forward units K0 + K1.  Returns (out[N,H,D], A[E,H,1], S[N,H,1]).

    ``ones_shortcut`` (what the layers pass): a device flag says whether every score is finite (stg_gat_score_flag); if
    so A is the constant 1.0f (``emb - max([emb])`` is +0: SURVEY.md D2) and is neither written by K0 nor read by K1 /
    K2 -- same bits.  The returned ``A`` then carries the flag (``A._stg_ones``) for ``gat_bwd`` and its CONTENT IS ONLY
    VALID WHEN THE FLAG IS SET (some non-finite score); callers that read A themselves leave the shortcut off."""
    feat = _f32(feat, "feat_src")
    dev = feat.device
    if feat.dim() != 3:
        raise ValueError("feat_src must be [N, H, D]")
    N, H, D = feat.shape
    el, er = _f32(el, "el", dev), _f32(er, "er", dev)
    if N != csr.num_nodes or el.numel() != N * H or er.numel() != N * H:
        raise ValueError("el/er must be [N, H, 1] and match the graph")
    if csr.row_offset.device != dev:
        raise RuntimeError(f"graph arrays are on {csr.row_offset.device}, features on {dev}")
    E = csr.num_edges
    h_act, hd_act = active_columns(H), active_columns(H * D)
    full = (h_act == H and hd_act == H * D)
    alloc = torch.empty if full else torch.zeros
    A = alloc((E, H, 1), dtype=torch.float32, device=dev)
    S = alloc((N, H, 1), dtype=torch.float32, device=dev)
    out = alloc((N, H, D), dtype=torch.float32, device=dev)
    nid = _ptr(csr.node_ids_if_ready if use_node_ids else None)
    ab = gat_algorithmic_bytes(N, E, H, D)
    flag = None
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        if ones_shortcut and _GAT_ONES and full and not reference_compat():
            flag = torch.empty(1, dtype=torch.int32, device=dev)
            _C.check(_C.lib.stg_gat_score_flag(_ptr(el), _ptr(er), N * H, _ptr(flag), st))
        with _Timed("gat_k0", ab["gat_k0"], E * H):
            _C.check(_C.lib.stg_gat_fwd_k0(_ptr(el), _ptr(er), _ptr(A), _ptr(S), _ptr(csr.row_offset),
                                           _ptr(csr.column_indices), _ptr(csr.eids), nid, N, H, h_act,
                                           float(slope), _ptr(flag), st))
        with _Timed("gat_k1", ab["gat_k1"], E * H * D):
            _C.check(_C.lib.stg_gat_fwd_k1(_ptr(A), _ptr(S), _ptr(feat), _ptr(out), _ptr(csr.row_offset),
                                           _ptr(csr.column_indices), _ptr(csr.eids), nid, N, H, D, hd_act,
                                           _ptr(flag), st))
    if flag is not None:
        A._stg_ones = flag
    return out, A, S


_GAT_UNIFORM = True


def set_gat_uniform_form(on: bool) -> None:
    """True (default): a fused GATConv whose input is narrower than its H*D output runs K1 in the uniform-attention form
    (:func:`gat_fwd_uniform`); False: K1 at full width always (tests compare the two)."""
    global _GAT_UNIFORM
    _GAT_UNIFORM = bool(on)


def gat_uniform_usable(x: torch.Tensor, H: int, D: int) -> bool:
    fin = int(x.shape[1])
    return (_GAT_UNIFORM and _GAT_ONES and not reference_compat() and fin < H * D and gat_fc_supported(fin, H, D)
            and active_columns(H) == H and active_columns(H * D) == H * D and x.data_ptr() % 16 == 0)


def gat_fwd_uniform(x: torch.Tensor, W: torch.Tensor, el: torch.Tensor, er: torch.Tensor, feat: torch.Tensor,
                    csr: DeviceCSR, slope: float, use_node_ids: bool = False, elu: bool = False, feat_unwritten: bool = False):
    """K0 + K1 of a layer whose ``feat = x @ W.T`` ([N, fin] -> [N, H, D], fin < H*D), in the uniform-attention form
    (stgraph_hip.h, ABI 23).  ``emb - max([emb])`` is +0 for every finite score (reference gat_conv.py:50, SURVEY.md
    D2), every A is then 1.0f and K1 is the in-neighbour mean of ``feat`` -- linear in x: the gather runs over x at
    width fin (stg_gat_fwd_k1_uniform, K1's own loop) and the product with W follows (stg_gat_fc_out, which also writes
    ``elu(out)`` when asked).  A device flag decides: with any non-finite score the narrow pass returns at once and the
    full-width K1 (stg_gat_fwd_k1_scored) overwrites the result -- the emitted unit's bits.  Returns (out, act, A, S);
    ``act`` is None unless ``elu``.  A's content is only valid when the flag is set (see :func:`gat_fwd`)."""
    x = _f32(x, "x")
    dev = x.device
    N, fin = x.shape
    feat = _f32(feat, "feat_src", dev)
    _, H, D = feat.shape
    W = _f32(W, "fc.weight", dev)
    el, er = _f32(el, "el", dev), _f32(er, "er", dev)
    if N != csr.num_nodes or feat.shape[0] != N or tuple(W.shape) != (H * D, fin) or el.numel() != N * H or er.numel() != N * H:
        raise ValueError("gat_fwd_uniform: x [N, fin], W [H*D, fin], feat [N, H, D], el / er [N, H, 1] must match the graph")
    if csr.row_offset.device != dev:
        raise RuntimeError(f"graph arrays are on {csr.row_offset.device}, features on {dev}")
    E = csr.num_edges
    A = torch.empty((E, H, 1), dtype=torch.float32, device=dev)
    S = torch.empty((N, H, 1), dtype=torch.float32, device=dev)
    xm = torch.empty((N, fin), dtype=torch.float32, device=dev)
    out = torch.empty((N, H, D), dtype=torch.float32, device=dev)
    act = torch.empty((N, H, D), dtype=torch.float32, device=dev) if elu else None
    nid = _ptr(csr.node_ids_if_ready if use_node_ids else None)
    ab = gat_algorithmic_bytes(N, E, H, D)
    idx = 4 * (N + 1) + 4 * E
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        flag = torch.empty(1, dtype=torch.int32, device=dev)
        _C.check(_C.lib.stg_gat_score_flag(_ptr(el), _ptr(er), N * H, _ptr(flag), st))
        if feat_unwritten:                              # gat_fc_fwd(store_feat=False): the general units below gather feat
            _C.check(_C.lib.stg_gat_fc_feat_if(_ptr(x), _ptr(W), _ptr(feat), N, fin, H, D, _ptr(flag), st))
        with _Timed("gat_k0", ab["gat_k0"], E * H):
            _C.check(_C.lib.stg_gat_fwd_k0(_ptr(el), _ptr(er), _ptr(A), _ptr(S), _ptr(csr.row_offset),
                                           _ptr(csr.column_indices), _ptr(csr.eids), nid, N, H, H, float(slope), _ptr(flag), st))
        with _Timed("gat_k1_uniform", 4 * E * fin + 8 * N * fin + 4 * N + idx, E * fin):
            _C.check(_C.lib.stg_gat_fwd_k1_uniform(_ptr(S), H, _ptr(x), _ptr(xm), _ptr(csr.row_offset),
                                                   _ptr(csr.column_indices), nid, N, fin, _ptr(flag), st))
        with _Timed("gat_fc_out", 4 * N * (fin + (2 if elu else 1) * H * D) + 4 * H * D * fin, 2 * N * fin * H * D):
            _C.check(_C.lib.stg_gat_fc_out(_ptr(xm), _ptr(W), _ptr(out), _ptr(act), N, fin, H, D, st))
        with _Timed("gat_k1_scored", 4, 0):
            _C.check(_C.lib.stg_gat_fwd_k1_scored(_ptr(A), _ptr(S), _ptr(feat), _ptr(out), _ptr(act), _ptr(csr.row_offset),
                                                  _ptr(csr.column_indices), _ptr(csr.eids), nid, N, H, D, _ptr(flag), st))
    A._stg_ones = flag
    A._stg_xm = xm                                     # the in-neighbour mean of x (valid unless *flag): gat_bwd_uniform's g^T xm
    return out, act, A, S


_GAT_FACTORED = True


def set_gat_factored_backward(enabled: bool) -> None:
    """True (default): K2 runs in its factored form (stg_gat_bwd_factored: the target-only term is
    hoisted into a per-vertex pre-pass, halving the gather).  False: the literal per-lane form of the
    reference's emitted kernel (stg_gat_bwd).  Reference-compat mode (D1) always uses the literal form."""
    global _GAT_FACTORED
    _GAT_FACTORED = bool(enabled)


_GAT_REGROUPED_ER = True


def set_gat_regrouped_er(on: bool) -> None:
    """False: the factored backward writes the per-edge terms T and grad_er is summed from them by stg_gat_bwd_er (as
    the emitted units do); True (default): grad_er from the per-vertex pass, slope * (g . out - P * S), T never formed."""
    global _GAT_REGROUPED_ER
    _GAT_REGROUPED_ER = bool(on)


def gat_bwd(A, S, out, g, el, er, feat, fwd: DeviceCSR, bwd: DeviceCSR, slope: float,
            use_node_ids: bool = False, elu: bool = False):
    """Backward unit K2 (+ the dst-major grad_er pass).  Returns (grad_feat, grad_el, grad_er).

    ``elu``: ``g`` is the gradient of ``elu(out)`` (a layer whose activation was fused, :func:`gat_fwd_uniform`): the
    factored form turns it into the gradient of ``out`` inside its per-vertex pass (stg_gat_bwd_factored_elu); the
    literal form gets it from torch's elu_backward first."""
    feat = _f32(feat, "feat_src")
    dev = feat.device
    N, H, D = feat.shape
    A, S, out, g, el, er = (_f32(t, n, dev) for t, n in
                            ((A, "A"), (S, "S"), (out, "out"), (g, "grad_out"), (el, "el"), (er, "er")))
    if g.shape != feat.shape or out.shape != feat.shape:
        raise ValueError("grad_out / out must be [N, H, D]")
    E = bwd.num_edges
    if A.numel() != E * H or S.numel() != N * H or fwd.num_edges != E or bwd.num_nodes != N:
        raise ValueError("A/S do not match the graph")
    h_act, hd_act = active_columns(H), active_columns(H * D)
    full = (h_act == H and hd_act == H * D)
    alloc = torch.empty if full else torch.zeros
    grad_feat = alloc((N, H, D), dtype=torch.float32, device=dev)
    grad_el = alloc((N, H, 1), dtype=torch.float32, device=dev)
    grad_er = alloc((N, H, 1), dtype=torch.float32, device=dev)
    regrouped = full and _GAT_FACTORED and _GAT_REGROUPED_ER
    T = None if regrouped else alloc((E, H), dtype=torch.float32, device=dev)
    flag = getattr(A, "_stg_ones", None)               # set by gat_fwd(ones_shortcut=True): A is 1.0f unless *flag
    ab = gat_algorithmic_bytes(N, E, H, D)
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        with _Timed("gat_bwd", ab["gat_bwd"], E * H * D):
            if full and _GAT_FACTORED:
                P = torch.empty((N, 2 * H), dtype=torch.float32, device=dev)    # scratch: P and 1 / S (or S)
                tail = (_ptr(P), _ptr(bwd.row_offset), _ptr(bwd.column_indices), _ptr(bwd.eids),
                        _ptr(bwd.node_ids_if_ready if use_node_ids else None), N, H, D, float(slope),
                        _ptr(grad_er if regrouped else None), _ptr(flag), st)
                if elu:
                    g_pre = torch.empty_like(g)
                    _C.check(_C.lib.stg_gat_bwd_factored_elu(_ptr(A), _ptr(S), _ptr(out), _ptr(g), _ptr(g_pre), _ptr(feat),
                                                             _ptr(grad_feat), _ptr(grad_el), _ptr(T), *tail))
                else:
                    _C.check(_C.lib.stg_gat_bwd_factored(_ptr(A), _ptr(S), _ptr(out), _ptr(g), _ptr(feat), _ptr(grad_feat),
                                                         _ptr(grad_el), _ptr(T), *tail))
                if regrouped:       # grad_er came out of the per-vertex pass (sum of T over in-edges, regrouped): no T, no
                    return grad_feat, grad_el, grad_er                            # dst-major pass over it
            else:
                if elu:
                    g = torch.ops.aten.elu_backward(g, 1.0, 1.0, 1.0, False, out)
                _C.check(_C.lib.stg_gat_bwd(
                    _ptr(A), _ptr(S), _ptr(out), _ptr(g), _ptr(el), _ptr(er), _ptr(feat),
                    _ptr(grad_feat), _ptr(grad_el), _ptr(T), _ptr(bwd.row_offset), _ptr(bwd.column_indices),
                    _ptr(bwd.eids), _ptr(bwd.node_ids_if_ready if use_node_ids else None), N, H, D, hd_act,
                    float(slope), _ptr(flag), st))
        # heads the backward unit touched: those with at least one active feature column
        h_touched = H if full else min(H, (hd_act + D - 1) // D)
        with _Timed("gat_bwd_er", ab["gat_bwd_er"], E * H):
            _C.check(_C.lib.stg_gat_bwd_er(_ptr(T), _ptr(grad_er), _ptr(fwd.row_offset), _ptr(fwd.eids),
                                           _ptr(fwd.node_ids_if_ready if use_node_ids else None), N, H, h_touched, st))
    return grad_feat, grad_el, grad_er


_GAT_UNIFORM_BWD = os.environ.get("STGRAPH_AMD_GAT_UNIFORM_BWD", "1") != "0"


def set_gat_uniform_backward(on: bool) -> None:
    """False: the fused GATConv's backward unit gathers rows of width H * D per edge (stg_gat_bwd_factored) also when the forward
    ran in the uniform-attention form."""
    global _GAT_UNIFORM_BWD
    _GAT_UNIFORM_BWD = bool(on)


def gat_bwd_uniform_shape(x: torch.Tensor, H: int, D: int) -> bool:
    """Switches and shapes of :func:`gat_bwd_uniform` (what the forward can know)."""
    return (_GAT_UNIFORM_BWD and _GAT_FACTORED and _GAT_REGROUPED_ER and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and x.data_ptr() % 16 == 0 and active_columns(H) == H and active_columns(H * D) == H * D
            and bool(_C.lib.stg_gat_bwd_uniform_supported(int(H), int(D), int(x.shape[1]))))


def gat_bwd_uniform_usable(A: torch.Tensor, x: torch.Tensor, H: int, D: int) -> bool:
    """``A`` from :func:`gat_fwd_uniform` (it carries the flag and the mean of x), shapes of stg_gat_bwd_uniform_supported."""
    return (gat_bwd_uniform_shape(x, H, D) and getattr(A, "_stg_ones", None) is not None
            and getattr(A, "_stg_xm", None) is not None)


def gat_bwd_uniform(A, S, out, g, x, W, feat, fwd: DeviceCSR, bwd: DeviceCSR, slope: float, use_node_ids: bool = False,
                    elu: bool = False):
    """The backward unit K2 of a layer that ran :func:`gat_fwd_uniform`, without a gather of width H * D per edge
    (include/stgraph_hip.h, stg_gat_bwd_uniform_edges).  Returns ``(gxa, grad_el, grad_er, gq, grad_feat, flag, xm)``:
    ``gxa`` [N, fin] = grad_feat @ W when every score is finite (0 otherwise: add grad_feat @ W with
    :func:`gat_bwd_uniform_gx_fallback`), ``gq`` the gradient of the pre-activation ``out`` (the weight gradient is ``gq^T xm``
    -- or ``grad_feat^T x`` when the flag is set: :func:`gemm_tn_gated`), ``grad_feat`` [N, H, D] only written in that case."""
    feat = _f32(feat, "feat_src")
    dev = feat.device
    N, H, D = feat.shape
    fin = int(x.shape[1])
    A, S, out, g, x, W = (_f32(t, n, dev) for t, n in ((A, "A"), (S, "S"), (out, "out"), (g, "grad_out"), (x, "x"), (W, "fc.weight")))
    E = bwd.num_edges
    if (g.shape != feat.shape or out.shape != feat.shape or A.numel() != E * H or S.numel() != N * H or fwd.num_edges != E
            or bwd.num_nodes != N or tuple(W.shape) != (H * D, fin) or x.shape[0] != N):
        raise ValueError("gat_bwd_uniform: operands do not match the graph / the layer")
    flag, xm = A._stg_ones, A._stg_xm
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    pack, grad_er, grad_el = new(N, 16), new(N, H, 1), new(N, H, 1)
    g_pre = torch.empty_like(g) if elu else None
    T, gsW, gxa, grad_feat = new(E, H), new(N, fin), new(N, fin), new(N, H, D)
    idx = 4 * (N + 1) + 8 * E
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        gq = g_pre if elu else g
        aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (out, g, g_pre, W))
        if _GAT_PREPASS_HEADS and N >= ROWGEMM16_MIN_ROWS and aligned and _C.lib.stg_gat_bwd_prepass_heads_supported(N, H, D, fin):
            # the per-vertex pass and gW [H, N, fin] (gW[h, v] = W_h^T g_pre[v, h, :]) from ONE read of g and out (csrc/gat_heads_x3.hip)
            gW = new(H, N, fin)
            with _Timed("gat_bwd_prepass_heads", 4 * N * H * D * (3 if elu else 2) + 4 * N * 16 + 8 * N * H + 4 * N * H * fin + 4 * H * D * fin,
                        2 * N * H * D * fin):
                _C.check(_C.lib.stg_gat_bwd_prepass_heads(_ptr(S), _ptr(out), _ptr(g), _ptr(g_pre), _ptr(pack), _ptr(grad_er), _ptr(W),
                                                          _ptr(gW), N, H, D, fin, float(slope), st))
        else:
            with _Timed("gat_bwd_prepass", 4 * N * H * D * (3 if elu else 2) + 4 * N * 16 + 8 * N * H, 4 * N * H * D):
                _C.check(_C.lib.stg_gat_bwd_prepass(_ptr(S), _ptr(out), _ptr(g), _ptr(g_pre), _ptr(pack), N, H, D, float(slope),
                                                    _ptr(grad_er), st))
            with _Timed("gat_bwd_gw", 4 * N * H * (D + fin) + 4 * H * D * fin, 2 * N * H * D * fin):
                # [H, N, fin]: gW[h, v] = W_h^T g[v, h, :]
                if N >= ROWGEMM16_MIN_ROWS and _C.lib.stg_rowgemm_heads_supported(N, D, fin, H) and gq.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0:
                    gW = new(H, N, fin)
                    _C.check(_C.lib.stg_rowgemm_heads_f32(_ptr(gq), _ptr(W), _ptr(gW), N, D, fin, H, st))
                else:
                    gW = torch.bmm(gq.view(N, H, D).transpose(0, 1), W.view(H, D, fin))
        moved = (4 * N * H * fin + 4 * E * fin + 4 * E * H + 4 * N * fin + 4 * N * 16 + idx            # targets: gW, x[u], T, gsW, pack
                 + 4 * E * H + 4 * E * fin + 4 * N * (fin + H) + idx)                                   # sources: T, gsW[v], grad_el, gxa
        with _Timed("gat_bwd_uniform", moved, E * (2 * fin + H)):      # units: executed gather = x[u] (fin) + gsW[v] (fin) + T[e] (H)
            _C.check(_C.lib.stg_gat_bwd_uniform_edges(
                _ptr(A), _ptr(pack), _ptr(gq), _ptr(feat), _ptr(x), _ptr(gW), _ptr(T), _ptr(gsW), _ptr(grad_feat), _ptr(grad_el),
                _ptr(gxa), _ptr(fwd.row_offset), _ptr(fwd.column_indices), _ptr(fwd.eids),
                _ptr(fwd.node_ids_if_ready if use_node_ids else None), _ptr(bwd.row_offset), _ptr(bwd.column_indices),
                _ptr(bwd.eids), _ptr(bwd.node_ids_if_ready if use_node_ids else None), N, float(slope), _ptr(flag), st))
    return gxa, grad_el, grad_er, gq, grad_feat, flag, xm


def gat_attn_fold_usable(W: torch.Tensor, H: int, D: int, fin: int) -> bool:
    return (_GAT_ATTN_FOLD and W.is_cuda and W.dtype == torch.float32 and W.is_contiguous() and tuple(W.shape) == (H * D, fin)
            and 4 * (D * fin + 2 * fin + 2 * D) <= 64 * 1024)


def gat_attn_fold(W, G, attn_l, attn_r, H: int, D: int, fin: int, want_aw: bool, gw: torch.Tensor | None):
    """(dattn_l [H, D], dattn_r [H, D], Aw [2H, fin] or None) and ``gw += attn (x) G`` in place -- the small products of the projection
    fold of the GAT layer's backward in one launch (stg_gat_attn_fold)."""
    dev = W.device
    G, al, ar = _f32(G, "G", dev), _f32(attn_l, "attn_l", dev).reshape(H, D), _f32(attn_r, "attn_r", dev).reshape(H, D)
    if tuple(G.shape) != (2 * H, fin) or (gw is not None and (tuple(gw.shape) != (H * D, fin) or not gw.is_contiguous())):
        raise ValueError("gat_attn_fold: G [2H, fin], gw [H D, fin] contiguous")
    dal = torch.empty(H, D, dtype=torch.float32, device=dev)
    dar = torch.empty(H, D, dtype=torch.float32, device=dev)
    Aw = torch.empty(2 * H, fin, dtype=torch.float32, device=dev) if want_aw else None
    with torch.cuda.device(dev), _Timed("gat_attn_fold", 4 * (H * D * fin * (3 if gw is not None else 1) + 4 * H * fin), 6 * H * D * fin):
        _C.check(_C.lib.stg_gat_attn_fold(_ptr(W), _ptr(G), _ptr(al), _ptr(ar), _ptr(dal), _ptr(dar), _ptr(Aw), _ptr(gw), H, D, fin,
                                          _stream_ptr(dev)))
    return dal, dar, Aw


_GAT_ATTN_FOLD = True


def set_gat_attn_fold(on: bool) -> None:
    """False: the projection fold's small products run as torch einsums / elementwise launches (rounds 3-4)."""
    global _GAT_ATTN_FOLD
    _GAT_ATTN_FOLD = bool(on)


_GAT_PREPASS_HEADS = True


def set_gat_prepass_heads(on: bool) -> None:
    """False: the uniform-attention backward runs its per-vertex pass and the per-head products g_pre W_h as separate launches
    (rounds 3-4) instead of stg_gat_bwd_prepass_heads."""
    global _GAT_PREPASS_HEADS
    _GAT_PREPASS_HEADS = bool(on)


def gat_bwd_uniform_gx_fallback(grad_feat: torch.Tensor, W: torch.Tensor, gx: torch.Tensor, flag: torch.Tensor) -> None:
    """``gx += grad_feat @ W`` in place when ``*flag`` is set (a non-finite score); a launch that returns at once otherwise."""
    N = int(gx.shape[0])
    with torch.cuda.device(gx.device):
        _C.check(_C.lib.stg_gat_bwd_uniform_gx_fallback(_ptr(grad_feat), _ptr(W), _ptr(gx), N, _ptr(flag), _stream_ptr(gx.device)))


def gemm_tn_gated(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, gate: torch.Tensor, run_if_zero: bool) -> None:
    """``c = a.T @ b`` only if ``*gate == 0`` (``run_if_zero``) / only if ``*gate != 0``; ``c`` is left alone otherwise
    (stg_gemm_tn_gated_f32).  Two calls on the same word, one of each kind, fill ``c`` whichever way the device decides."""
    a, b = _f32(a, "a"), _f32(b, "b", a.device)
    K, M = a.shape
    N = int(b.shape[1])
    if b.shape[0] != K or tuple(c.shape) != (M, N) or not c.is_contiguous() or c.dtype != torch.float32:
        raise ValueError("gemm_tn_gated: a [K, M], b [K, N], c [M, N] contiguous fp32")
    ws_bytes = int(_C.lib.stg_gemm_tn_workspace_bytes(K, M, N))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device), _Timed("gemm_tn", 4 * K * (M + N) + 4 * M * N, 2 * K * M * N):
        _C.check(_C.lib.stg_gemm_tn_gated_f32(_ptr(a), _ptr(b), _ptr(c), K, M, N, _ptr(ws), ws_bytes, _ptr(gate),
                                              1 if run_if_zero else 2, _stream_ptr(a.device)))


# ------------------------------------------------------- dense neighbour: weight gradient
def gat_proj_supported(H: int, D: int) -> bool:
    return bool(_C.lib.stg_gat_proj_supported(int(H), int(D)))


def gat_fc_supported(fin: int, H: int, D: int) -> bool:
    return bool(_C.lib.stg_gat_fc_supported(int(fin), int(H), int(D)))


def gat_fc_fwd(x: torch.Tensor, W: torch.Tensor, attn_l: torch.Tensor, attn_r: torch.Tensor, H: int, D: int, store_feat: bool = True):
    """(feat [N,H,D], el, er [N,H,1]): ``feat = x @ W.T`` and the attention projections from its accumulators, one
    launch (stg_gat_fc_fwd).  ``store_feat=False``: ``feat`` is allocated but NOT written (the uniform-attention form never reads
    it; :func:`gat_fc_feat_if` fills it when the general units are going to run)."""
    x = _f32(x, "x")
    dev = x.device
    N, fin = x.shape
    W = _f32(W, "fc.weight", dev)
    al, ar = _f32(attn_l, "attn_l", dev), _f32(attn_r, "attn_r", dev)
    if tuple(W.shape) != (H * D, fin) or al.numel() != H * D or ar.numel() != H * D:
        raise ValueError("fc.weight must be [H*D, fin], attn_l / attn_r [H, D]")
    feat = torch.empty(N, H, D, dtype=torch.float32, device=dev)
    el = torch.empty(N, H, 1, dtype=torch.float32, device=dev)
    er = torch.empty(N, H, 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("gat_fc", 4 * N * (fin + (H * D if store_feat else 0) + 2 * H) + 4 * H * D * fin, 2 * N * fin * H * D):
        _C.check(_C.lib.stg_gat_fc_fwd(_ptr(x), _ptr(W), _ptr(al), _ptr(ar), _ptr(feat if store_feat else None), _ptr(el), _ptr(er),
                                       N, fin, H, D, _stream_ptr(dev)))
    return feat, el, er


def gat_fc_feat_if(x: torch.Tensor, W: torch.Tensor, feat: torch.Tensor, flag: torch.Tensor | None) -> None:
    """``feat[...] = x @ W.T`` if ``*flag != 0`` (``flag`` None: always) -- the rows :func:`gat_fc_fwd` left unwritten."""
    N, H, D = feat.shape
    fin = int(x.shape[1])
    dev = feat.device
    with torch.cuda.device(dev):
        if flag is None:
            _C.check(_C.lib.stg_gat_fc_out(_ptr(x), _ptr(W), _ptr(feat), None, N, fin, H, D, _stream_ptr(dev)))
        else:
            _C.check(_C.lib.stg_gat_fc_feat_if(_ptr(x), _ptr(W), _ptr(feat), N, fin, H, D, _ptr(flag), _stream_ptr(dev)))


def gat_proj_fwd(feat: torch.Tensor, attn_l: torch.Tensor, attn_r: torch.Tensor):
    """(el, er) [N,H,1] = sum_d feat[n,h,d] * attn_{l,r}[h,d] in one pass (stg_gat_proj_fwd)."""
    feat = _f32(feat, "feat_src")
    dev = feat.device
    N, H, D = feat.shape
    al, ar = _f32(attn_l, "attn_l", dev), _f32(attn_r, "attn_r", dev)
    if al.numel() != H * D or ar.numel() != H * D:
        raise ValueError("attn_l / attn_r must be [H, D]")
    el = torch.empty(N, H, 1, dtype=torch.float32, device=dev)
    er = torch.empty(N, H, 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("gat_proj_fwd", 4 * N * H * (D + 2), 4 * N * H * D):
        _C.check(_C.lib.stg_gat_proj_fwd(_ptr(feat), _ptr(al), _ptr(ar), _ptr(el), _ptr(er), N, H, D, _stream_ptr(dev)))
    return el, er


def gat_proj_bwd(feat, attn_l, attn_r, d_el, d_er, g: torch.Tensor | None, inplace: bool = False):
    """(dfeat, dattn_l, dattn_r): ``dfeat = g + d_el*attn_l + d_er*attn_r`` (into ``g`` itself with ``inplace``),
    ``dattn_* = sum_n d_e*[n,h] feat[n,h,:]`` (stg_gat_proj_bwd)."""
    feat = _f32(feat, "feat_src")
    dev = feat.device
    N, H, D = feat.shape
    al, ar = _f32(attn_l, "attn_l", dev), _f32(attn_r, "attn_r", dev)
    d_el, d_er = _f32(d_el, "d_el", dev), _f32(d_er, "d_er", dev)
    if d_el.numel() != N * H or d_er.numel() != N * H:
        raise ValueError("d_el / d_er must be [N, H, 1]")
    if g is not None:
        g = _f32(g, "g", dev)
        if g.shape != feat.shape:
            raise ValueError("g must have feat's shape")
    dfeat = g if (inplace and g is not None) else torch.empty_like(feat)
    dal = torch.empty(H, D, dtype=torch.float32, device=dev)
    dar = torch.empty(H, D, dtype=torch.float32, device=dev)
    ws_bytes = int(_C.lib.stg_gat_proj_bwd_workspace_bytes(N, H, D))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _Timed("gat_proj_bwd", 4 * N * H * D * (3 if g is not None else 2), 8 * N * H * D):
        _C.check(_C.lib.stg_gat_proj_bwd(_ptr(feat), _ptr(al), _ptr(ar), _ptr(d_el), _ptr(d_er), _ptr(g), _ptr(dfeat),
                                         _ptr(dal), _ptr(dar), N, H, D, _ptr(ws), ws_bytes, _stream_ptr(dev)))
    return dfeat, dal, dar


def gemm_tn(a: torch.Tensor, b: torch.Tensor, colsum: bool = False):
    """``a.T @ b`` for tall-skinny fp32 operands ``a [K, M]``, ``b [K, N]`` (stg_gemm_tn_f32).
    ``colsum=True`` also returns ``a.sum(0)`` from the same launch (stg_gemm_tn_colsum_f32)."""
    a, b = _f32(a, "a"), _f32(b, "b", a.device)
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError(f"gemm_tn expects [K,M] and [K,N], got {tuple(a.shape)} and {tuple(b.shape)}")
    K, M = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    cs = torch.empty(M, dtype=torch.float32, device=a.device) if colsum else None
    if M == 0 or N == 0:
        return (c, cs) if colsum else c
    ws_bytes = int(_C.lib.stg_gemm_tn_workspace_bytes(K, M, N))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device), _Timed("gemm_tn", 4 * K * (M + N) + 4 * M * N, 2 * K * M * N):
        if colsum:
            _C.check(_C.lib.stg_gemm_tn_colsum_f32(_ptr(a), _ptr(b), _ptr(c), _ptr(cs), K, M, N, _ptr(ws), ws_bytes,
                                                   _stream_ptr(a.device)))
        else:
            _C.check(_C.lib.stg_gemm_tn_f32(_ptr(a), _ptr(b), _ptr(c), K, M, N, _ptr(ws), ws_bytes,
                                            _stream_ptr(a.device)))
    return (c, cs) if colsum else c


def gemm_tn_relu_mask(g: torch.Tensor, out: torch.Tensor, x: torch.Tensor, colsum: bool = True):
    """``(g * (out > 0)).T @ x`` and, with ``colsum``, ``(g * (out > 0)).sum(0)`` -- the weight ([M, N], i.e. torch
    Linear layout [out, in]) and bias gradients of ``relu(x W + b)`` -- in one launch, the masked gradient never
    written (stg_gemm_tn_relu_mask_f32)."""
    g, out, x = _f32(g, "g"), _f32(out, "out", g.device), _f32(x, "x", g.device)
    if g.dim() != 2 or out.shape != g.shape or x.dim() != 2 or x.shape[0] != g.shape[0]:
        raise ValueError("gemm_tn_relu_mask expects g, out [K, M] and x [K, N]")
    K, M = g.shape
    N = x.shape[1]
    c = torch.empty(M, N, dtype=torch.float32, device=g.device)
    cs = torch.empty(M, dtype=torch.float32, device=g.device) if colsum else None
    ws_bytes = int(_C.lib.stg_gemm_tn_workspace_bytes(K, M, N))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device), _Timed("gemm_tn", 4 * K * (2 * M + N) + 4 * M * N, 2 * K * M * N):
        _C.check(_C.lib.stg_gemm_tn_relu_mask_f32(_ptr(g), _ptr(out), _ptr(x), _ptr(c), _ptr(cs), K, M, N, _ptr(ws),
                                                  ws_bytes, _stream_ptr(g.device)))
    return (c, cs) if colsum else c


_WIDE_LINEAR = True
_ROWGEMM_MODE = os.environ.get("STGRAPH_AMD_ROWGEMM", "0")     # "0" | "1" | "auto" (only where measured faster)
_ROWGEMM = _ROWGEMM_MODE != "0"


def set_native_rowgemm(enabled: bool) -> None:
    """True: skinny forward / input-gradient GEMMs run on stg_rowgemm_f32; False (default): torch (rocBLAS).
    Measured on MI355X (tools/microbench_rowgemm.py, round 1): 0.5-1.3x rocBLAS depending on the shape, so
    it is not the default yet; the kernel restages W per 64-row tile and does not overlap staging with MFMA."""
    global _ROWGEMM, _ROWGEMM_MODE
    _ROWGEMM, _ROWGEMM_MODE = bool(enabled), ("1" if enabled else "0")


ROWGEMM16_MIN_ROWS = 65536


def rowgemm16_usable(x: torch.Tensor, K: int, M: int) -> bool:
    """Tall fp32 products with K, M in {64, 128} (the dense layers of the GCN / GAT configs): the 16-row row-piece kernel
    of stg_rowgemm_f32 -- 1.14x (128 x 128) to 1.4x (64 -> 128) hipBLASLt's time at 256 K - 1 M rows."""
    return (_ROWGEMM16 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
            and x.shape[0] >= ROWGEMM16_MIN_ROWS and K in (64, 128) and M in (64, 128) and x.data_ptr() % 16 == 0)


_ROWGEMM16 = os.environ.get("STGRAPH_AMD_ROWGEMM16", "1") != "0"


def rowgemm_act(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, trans_w: bool, act: int = ACT_NONE) -> torch.Tensor:
    """``act(x @ op(w) + bias)`` in one launch (stg_rowgemm_act_f32; shapes: :func:`rowgemm16_usable`)."""
    x = _f32(x, "x")
    dev = x.device
    w = _f32(w, "w", dev)
    N, K = x.shape
    M = int(w.shape[0] if trans_w else w.shape[1])
    if int(w.shape[1] if trans_w else w.shape[0]) != K:
        raise ValueError(f"rowgemm_act: x {tuple(x.shape)} and w {tuple(w.shape)} (trans_w={trans_w}) do not match")
    if bias is not None:
        bias = _f32(bias, "bias", dev)
        if bias.numel() != M:
            raise ValueError("rowgemm_act: bias length != output width")
    y = torch.empty(N, M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("rowgemm", 4 * N * (K + M) + 4 * K * M, 2 * N * K * M):
        _C.check(_C.lib.stg_rowgemm_act_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, K, M, int(bool(trans_w)), int(act),
                                            _stream_ptr(dev)))
    return y


def rowgemm_bits_usable(x: torch.Tensor, K: int, M: int) -> bool:
    """The ReLU sign pattern as bits (stg_rowgemm_act_bits_f32): where the split-form row product serves (:func:`rowgemm16_usable`,
    K, M in {64, 128})."""
    return (_RELU_BITS and rowgemm16_usable(x, K, M) and bool(_C.lib.stg_rowgemm_bits_supported(int(x.shape[0]), int(K), int(M))))


_RELU_BITS = os.environ.get("STGRAPH_AMD_RELU_BITS", "1") != "0"


def set_relu_bits(on: bool) -> None:
    """False: the ReLU backward of the GCN input layer re-reads the layer's output (the form before the bit pattern)."""
    global _RELU_BITS
    _RELU_BITS = bool(on)


def rowgemm_relu_bits(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None):
    """``relu(x @ w + bias)`` and the bit pattern ``[out > 0]`` (opaque int32 words, include/stgraph_hip.h) for
    :func:`rowgemm_masked_t`."""
    x = _f32(x, "x")
    dev = x.device
    w = _f32(w, "w", dev)
    N, K = x.shape
    M = int(w.shape[1])
    if int(w.shape[0]) != K:
        raise ValueError(f"rowgemm_relu_bits: x {tuple(x.shape)} and w {tuple(w.shape)} do not match")
    if bias is not None:
        bias = _f32(bias, "bias", dev)
        if bias.numel() != M:
            raise ValueError("rowgemm_relu_bits: bias length != output width")
    y = torch.empty(N, M, dtype=torch.float32, device=dev)
    bits = torch.empty(int(_C.lib.stg_rowgemm_bits_words(N)), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev), _Timed("rowgemm", 4 * N * (K + M) + 4 * K * M + N * M // 8, 2 * N * K * M):
        _C.check(_C.lib.stg_rowgemm_act_bits_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, K, M, 0, int(ACT_RELU), None, _ptr(bits),
                                                 _stream_ptr(dev)))
    return y, bits


def rowgemm_masked_t(g: torch.Tensor, w: torch.Tensor, bits: torch.Tensor) -> torch.Tensor:
    """``(g @ w.T) * pattern`` with ``w`` [M, K] read in place and ``pattern`` the bits :func:`rowgemm_relu_bits` left for a
    ReLU output of shape [N, M]: the gradient with respect to that ReLU's pre-activation, from the launch that forms ``g w^T``."""
    g = _f32(g, "g")
    dev = g.device
    w = _f32(w, "w", dev)
    N, K = g.shape
    M = int(w.shape[0])
    if int(w.shape[1]) != K:
        raise ValueError(f"rowgemm_masked_t: g {tuple(g.shape)} and w {tuple(w.shape)} do not match")
    if bits.dtype != torch.int32 or bits.device != dev or bits.numel() != int(_C.lib.stg_rowgemm_bits_words(N)):
        raise ValueError("rowgemm_masked_t: bits is not the pattern of an [N, M] output on this device")
    y = torch.empty(N, M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("rowgemm", 4 * N * (K + M) + 4 * K * M + N * M // 8, 2 * N * K * M):
        _C.check(_C.lib.stg_rowgemm_act_bits_f32(_ptr(g), _ptr(w), None, _ptr(y), N, K, M, 1, int(ACT_NONE), _ptr(bits), None,
                                                 _stream_ptr(dev)))
    return y


def rowgemm_usable(x: torch.Tensor, K: int, M: int, trans_w: bool = False) -> bool:
    # "auto": in situ (TGCN, |V| = 50K) the kernel beats rocBLAS's 64x32 macro tile on x @ W with W [K, M] wider
    # than deep (19.9 vs 29.7 us for [50K,64] x [64,128]) and loses on the transposed forward shapes (24.4 vs 17.5 us)
    if _ROWGEMM_MODE == "auto" and (trans_w or M < 2 * K):
        return False
    return (_ROWGEMM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 4096
            and bool(_C.lib.stg_rowgemm_supported(int(K), int(M))))


def rowgemm(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, trans_w: bool = False) -> torch.Tensor:
    """``x @ w`` (``trans_w=False``, w [K,M]) or ``x @ w.T`` (``trans_w=True``, w [M,K]) ``+ bias``."""
    x = _f32(x, "x")
    dev = x.device
    w = _f32(w, "w", dev)
    N, K = x.shape
    M = int(w.shape[0] if trans_w else w.shape[1])
    if int(w.shape[1] if trans_w else w.shape[0]) != K:
        raise ValueError(f"rowgemm: x {tuple(x.shape)} and w {tuple(w.shape)} (trans_w={trans_w}) do not match")
    if bias is not None:
        bias = _f32(bias, "bias", dev)
        if bias.numel() != M:
            raise ValueError("rowgemm: bias length != output width")
    y = torch.empty(N, M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("rowgemm", 4 * N * (K + M) + 4 * K * M, 2 * N * K * M):
        _C.check(_C.lib.stg_rowgemm_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), N, K, M, int(bool(trans_w)),
                                        _stream_ptr(dev)))
    return y


WIDE_SLICE = 128


def wide_linear_usable(x: torch.Tensor, w: torch.Tensor) -> bool:
    """``x @ w.T`` with a WIDE output from a narrow input over many rows (GATConv's ``fc``: [256K, 64] -> 512): rocBLAS
    picks a 64x32 macro tile there (0.6 ms where four 128-column launches of it take 0.25); the row kernel computes it
    in 128-column slices of ``w`` written straight into the output (stg_rowgemm_strided_f32)."""
    M, K = int(w.shape[0]), int(w.shape[1])
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 65536 and M >= 256
            and M % WIDE_SLICE == 0 and K <= 64 and w.is_contiguous()
            and bool(_C.lib.stg_rowgemm_supported(K, WIDE_SLICE)))


def wide_linear_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None) -> torch.Tensor:
    x = _f32(x, "x")
    dev = x.device
    N, K = x.shape
    M = int(w.shape[0])
    y = torch.empty(N, M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _Timed("rowgemm_wide", 4 * N * (K + M) + 4 * K * M, 2 * N * K * M):
        for c in range(0, M, WIDE_SLICE):
            _C.check(_C.lib.stg_rowgemm_strided_f32(
                _ptr(x), ctypes.c_void_p(w.data_ptr() + 4 * c * K), ctypes.c_void_p(b.data_ptr() + 4 * c) if b is not None else None,
                ctypes.c_void_p(y.data_ptr() + 4 * c), N, K, WIDE_SLICE, M, 1, _stream_ptr(dev)))
    return y


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None) -> torch.Tensor:
    """``x @ w.T + b`` (torch Linear layout) on the native kernel when the shape is covered."""
    if rowgemm_usable(x, w.shape[1], w.shape[0], True):
        return rowgemm(x, w, b, trans_w=True)
    if _WIDE_LINEAR and wide_linear_usable(x, w):
        return wide_linear_fwd(x, w, b)
    return torch.addmm(b, x, w.t()) if b is not None else torch.mm(x, w.t())


def matmul(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``x @ w`` (w [K,M]) on the native kernel when the shape is covered."""
    if rowgemm_usable(x, w.shape[0], w.shape[1]):
        return rowgemm(x, w, None, trans_w=False)
    return torch.mm(x, w)


def matmul_t(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``x @ w.T`` (w [M,K])."""
    if rowgemm_usable(x, w.shape[1], w.shape[0], True):
        return rowgemm(x, w, None, trans_w=True)
    return torch.mm(x, w.t())


MAX_GEMM_SEGMENTS = 32


def gemm_tn_multi(As, Bs, colsum: bool = False):
    """``sum_t As[t].T @ Bs[t]`` (all ``[K, M]`` / ``[K, N]`` fp32) in one launch per 32 segments
    (stg_gemm_tn_multi_f32); with ``colsum`` also ``sum_t As[t].sum(0)``."""
    if len(As) != len(Bs) or not As:
        raise ValueError("gemm_tn_multi needs equally long, non-empty operand lists")
    dev = As[0].device
    As = [_f32(a, "a", dev) for a in As]
    Bs = [_f32(b, "b", dev) for b in Bs]
    K, M = As[0].shape
    N = Bs[0].shape[1]
    for a, b in zip(As, Bs):
        if a.shape != (K, M) or b.shape != (K, N):
            raise ValueError("gemm_tn_multi: all segments must share one shape")
    c_tot = cs_tot = None
    for i in range(0, len(As), MAX_GEMM_SEGMENTS):
        a_chunk, b_chunk = As[i:i + MAX_GEMM_SEGMENTS], Bs[i:i + MAX_GEMM_SEGMENTS]
        T = len(a_chunk)
        c = torch.empty(M, N, dtype=torch.float32, device=dev)
        cs = torch.empty(M, dtype=torch.float32, device=dev) if colsum else None
        ws_bytes = int(_C.lib.stg_gemm_tn_multi_workspace_bytes(T, K, M, N))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        pa = (ctypes.c_void_p * T)(*[t.data_ptr() for t in a_chunk])
        pb = (ctypes.c_void_p * T)(*[t.data_ptr() for t in b_chunk])
        with torch.cuda.device(dev), _Timed("gemm_tn_multi", 4 * T * K * (M + N) + 4 * M * N, 2 * T * K * M * N):
            _C.check(_C.lib.stg_gemm_tn_multi_f32(pa, pb, T, _ptr(c), _ptr(cs), K, M, N, _ptr(ws), ws_bytes,
                                                  _stream_ptr(dev)))
        c_tot = c if c_tot is None else c_tot + c
        if colsum:
            cs_tot = cs if cs_tot is None else cs_tot + cs
    return (c_tot, cs_tot) if colsum else c_tot


GEMM_B_NONE, GEMM_B_CLAMP, GEMM_B_RELU = 0, 1, 2          # include/stgraph_hip.h STG_GEMM_B_*


def gemm_tn_form(As, Bs, M: int, N: int, B2s=None, nsplit: int | None = None, b_op: int = GEMM_B_NONE, lo: float = 0.0,
                 hi: float = 0.0, colsum: bool = False):
    """``sum_t A_t.T @ [op(b_t[:, :nsplit]) | b2_t]`` with operands taken in place (stg_gemm_tn_form_f32): every
    ``A_t`` is a 2-D fp32 view of M columns (unit column stride, any row stride), ``b_t`` one of ``nsplit`` columns
    (default N) and ``b2_t`` one of ``N - nsplit``; all segments share strides.  ``op``: GEMM_B_CLAMP / GEMM_B_RELU
    applied to ``b_t`` while loading.  With ``colsum`` also ``sum_t A_t.sum(0)``."""
    if not As or len(As) != len(Bs) or (B2s is not None and len(B2s) != len(As)):
        raise ValueError("gemm_tn_form needs equally long, non-empty operand lists")
    nsplit = N if nsplit is None else int(nsplit)
    dev = As[0].device
    K = int(As[0].shape[0])

    def check(ts, cols, name):
        ld = None
        for t in ts:
            if (t.dtype != torch.float32 or not t.is_cuda or t.device != dev or t.dim() != 2 or t.shape[0] != K
                    or t.shape[1] != cols or (cols > 1 and t.stride(1) != 1)):
                raise ValueError(f"gemm_tn_form: {name} must be [K={K}, {cols}] fp32 views with unit column stride on {dev}")
            ldt = int(t.stride(0)) if K > 1 else max(int(t.stride(0)), cols)
            ld = ldt if ld is None else ld
            if ldt != ld:
                raise ValueError(f"gemm_tn_form: all {name} segments must share one row stride")
        return max(ld, cols)
    lda = check(As, M, "A")
    ldb = check(Bs, nsplit, "B")
    ldb2 = check(B2s, N - nsplit, "B2") if nsplit < N else 0
    c_tot = cs_tot = None
    for i in range(0, len(As), MAX_GEMM_SEGMENTS):
        T = len(As[i:i + MAX_GEMM_SEGMENTS])
        c = torch.empty(M, N, dtype=torch.float32, device=dev)
        cs = torch.empty(M, dtype=torch.float32, device=dev) if colsum else None
        ws_bytes = int(_C.lib.stg_gemm_tn_form_workspace_bytes(T, K, M, N, max(lda, ldb, ldb2)))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        pa = (ctypes.c_void_p * T)(*[t.data_ptr() for t in As[i:i + T]])
        pb = (ctypes.c_void_p * T)(*[t.data_ptr() for t in Bs[i:i + T]])
        pb2 = (ctypes.c_void_p * T)(*[t.data_ptr() for t in B2s[i:i + T]]) if nsplit < N else None
        with torch.cuda.device(dev), _Timed("gemm_tn_multi", 4 * T * K * (M + N) + 4 * M * N, 2 * T * K * M * N):
            _C.check(_C.lib.stg_gemm_tn_form_f32(pa, lda, pb, ldb, nsplit, pb2, ldb2, int(b_op), float(lo), float(hi), T,
                                                 _ptr(c), _ptr(cs), K, M, N, _ptr(ws), ws_bytes, _stream_ptr(dev)))
        c_tot = c if c_tot is None else c_tot + c
        if colsum:
            cs_tot = cs if cs_tot is None else cs_tot + cs
    return (c_tot, cs_tot) if colsum else c_tot


GEMM_REDUCE_JOBS = 8


def gemm_tn_form_batch(calls):
    """Several :func:`gemm_tn_form` contractions whose final reductions run as ONE launch (stg_gemm_tn_form_partial_f32 per
    product, then stg_gemm_tn_reduce_multi_f32): ``calls`` is a list of keyword dicts for ``gemm_tn_form``; returns the list of
    its results, same values bit for bit.  Optional keys ``out`` [M, N] / ``colsum_out`` [M]: contiguous fp32 tensors the results
    are written to (a parameter's ``.grad`` view, say).  Or ``out_blocks_t``: a list of M / rows contiguous [N, rows] tensors -- the
    result's row blocks, each transposed (stacked layers' weight gradients in their parameters' layout) -- with ``colsum_blocks``:
    as many [rows] tensors (iff ``colsum``); the product's entry in the returned list is then None.
    Products with more than 32 segments, or more than 8 of them, go one by one."""
    if len(calls) > GEMM_REDUCE_JOBS or any(len(c["As"]) > MAX_GEMM_SEGMENTS for c in calls) or len(calls) < 2:
        res = []
        for c in calls:
            c = dict(c)
            out, cs_out = c.pop("out", None), c.pop("colsum_out", None)
            blocks, cs_blocks = c.pop("out_blocks_t", None), c.pop("colsum_blocks", None)
            r = gemm_tn_form(**c)
            if blocks is not None:
                rows = int(c["M"]) // len(blocks)
                full, cs = (r if c.get("colsum") else (r, None))
                for b, t in enumerate(blocks):
                    t.copy_(full[b * rows:(b + 1) * rows].t())
                    if cs is not None:
                        cs_blocks[b].copy_(cs[b * rows:(b + 1) * rows])
                res.append(None)
                continue
            if out is not None:                              # (``out`` / ``colsum_out``: results written where the caller wants them)
                out.copy_(r[0] if c.get("colsum") else r)
                r = (out, r[1]) if c.get("colsum") else out
            if cs_out is not None and c.get("colsum"):
                cs_out.copy_(r[1])
                r = (r[0], cs_out)
            res.append(r)
        return res
    dev = calls[0]["As"][0].device
    outs, keep = [], []
    slabs_p, c_p, cs_p = [], [], []
    Ms, Ns, Ss, rows_l = [], [], [], []
    nb_max = _C.GEMM_REDUCE_BLOCKS
    blk_p, csblk_p = [], []
    with torch.cuda.device(dev):
        for c in calls:
            As, Bs, M, N = c["As"], c["Bs"], int(c["M"]), int(c["N"])
            B2s, nsplit = c.get("B2s"), c.get("nsplit")
            nsplit = N if nsplit is None else int(nsplit)
            colsum = bool(c.get("colsum", False))
            K, T = int(As[0].shape[0]), len(As)
            lda = max(int(As[0].stride(0)), M)
            ldb = max(int(Bs[0].stride(0)), nsplit)
            ldb2 = max(int(B2s[0].stride(0)), N - nsplit) if nsplit < N else 0
            for ts, cols, ld in ((As, M, lda), (Bs, nsplit, ldb)) + (((B2s, N - nsplit, ldb2),) if nsplit < N else ()):
                for t in ts:
                    if (t.dtype != torch.float32 or t.device != dev or t.dim() != 2 or t.shape != (K, cols) or
                            (cols > 1 and t.stride(1) != 1) or (K > 1 and max(int(t.stride(0)), cols) != ld)):
                        raise ValueError("gemm_tn_form_batch: operands must be [K, cols] fp32 views sharing one row stride")
            ws_bytes = int(_C.lib.stg_gemm_tn_form_workspace_bytes(T, K, M, N, max(lda, ldb, ldb2)))
            ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
            pa = (ctypes.c_void_p * T)(*[t.data_ptr() for t in As])
            pb = (ctypes.c_void_p * T)(*[t.data_ptr() for t in Bs])
            pb2 = (ctypes.c_void_p * T)(*[t.data_ptr() for t in B2s]) if nsplit < N else None
            S = ctypes.c_int32(0)
            with _Timed("gemm_tn_form", 4 * T * K * (M + N) + 4 * M * N, 2 * T * K * M * N):
                _C.check(_C.lib.stg_gemm_tn_form_partial_f32(pa, lda, pb, ldb, nsplit, pb2, ldb2, int(c.get("b_op", GEMM_B_NONE)),
                                                             float(c.get("lo", 0.0)), float(c.get("hi", 0.0)), T, int(colsum), K, M, N,
                                                             _ptr(ws), ws_bytes, ctypes.byref(S), _stream_ptr(dev)))
            blocks, cs_blocks = c.get("out_blocks_t"), c.get("colsum_blocks")
            if blocks is not None:
                nb = len(blocks)
                rows = M // max(nb, 1)
                if (not 0 < nb <= nb_max or rows * nb != M or (colsum and (cs_blocks is None or len(cs_blocks) != nb)) or
                        any(t.dtype != torch.float32 or t.device != dev or tuple(t.shape) != (N, rows) or not t.is_contiguous() for t in blocks) or
                        (colsum and any(t.dtype != torch.float32 or t.device != dev or tuple(t.shape) != (rows,) or not t.is_contiguous()
                                        for t in cs_blocks))):
                    raise ValueError(f"gemm_tn_form_batch: out_blocks_t must be 1 .. {nb_max} contiguous fp32 [N, M / blocks] tensors (+ colsum_blocks)")
                keep.append(ws)
                outs.append(None)
                slabs_p.append(ws.data_ptr()); c_p.append(None); cs_p.append(None)
                Ms.append(M); Ns.append(N); Ss.append(int(S.value)); rows_l.append(rows)
                blk_p += [t.data_ptr() for t in blocks] + [None] * (nb_max - nb)
                csblk_p += ([t.data_ptr() for t in cs_blocks] if colsum else [None] * nb) + [None] * (nb_max - nb)
                continue
            rows_l.append(0)
            blk_p += [None] * nb_max
            csblk_p += [None] * nb_max
            out, cs = c.get("out"), c.get("colsum_out") if colsum else None
            for t, shape, name in ((out, (M, N), "out"), (cs, (M,), "colsum_out")):
                if t is not None and (t.dtype != torch.float32 or t.device != dev or tuple(t.shape) != shape or not t.is_contiguous()):
                    raise ValueError(f"gemm_tn_form_batch: {name} must be a contiguous fp32 {shape} tensor on {dev}")
            if out is None:
                out = torch.empty(M, N, dtype=torch.float32, device=dev)
            if colsum and cs is None:
                cs = torch.empty(M, dtype=torch.float32, device=dev)
            keep.append(ws)
            outs.append((out, cs) if colsum else out)
            slabs_p.append(ws.data_ptr()); c_p.append(out.data_ptr()); cs_p.append(cs.data_ptr() if colsum else None)
            Ms.append(M); Ns.append(N); Ss.append(int(S.value))
        n = len(calls)
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)  # noqa: E731
        i32s = lambda vals: (ctypes.c_int32 * n)(*vals)  # noqa: E731
        if any(rows_l):
            ptrs = lambda vals: (ctypes.c_void_p * len(vals))(*vals)  # noqa: E731
            _C.check(_C.lib.stg_gemm_tn_reduce_multi_blocks_f32(n, arr(slabs_p), arr(c_p), arr(cs_p), i32s(Ms), i32s(Ns), i32s(Ss),
                                                                i32s(rows_l), ptrs(blk_p), ptrs(csblk_p), _stream_ptr(dev)))
        else:
            _C.check(_C.lib.stg_gemm_tn_reduce_multi_f32(n, arr(slabs_p), arr(c_p), arr(cs_p), i32s(Ms), i32s(Ns), i32s(Ss),
                                                         _stream_ptr(dev)))
    return outs


# ------------------------------------------------------------- one TGCN step per launch (csrc/tgcn_step.hip)
# True (default): the window nodes hand the FORWARD step launch the gate Linears with the conv folded in (tgcn_fold_weights) and it
# runs in its folded form (csrc/tgcn_step_fwd.hip, FOLD): the gate products straight from P on the fp32 matrix instruction, 320 per
# tile instead of 512, no x3 formed, the clamp BOUNDED instead of looked at.  The fold is exact only while no conv output is clamped
# (|.| <= 1e6 always holds on sane data): the launch raises a sticky per-device status word otherwise; the epoch functions of
# stgraph_amd.temporal then rerun the epoch in the reference formulation (_FoldGuard).  Needs STEP_WGRAD_FROM_P (nobody may read x3).
STEP_FOLDED = os.environ.get("STGRAPH_AMD_STEP_FOLDED", "1") != "0"


STEP_WGRAD_ZR_TOGETHER = os.environ.get("STGRAPH_AMD_STEP_WGRAD_ZR_TOGETHER", "1") != "0"


def set_step_wgrad_zr_together(on: bool) -> None:
    """True (default): the window nodes keep the gate gradients as column blocks of one [N, 3C] matrix
    (stg_tgcn_step_bwd_args::ld_d) and contract [d_z | d_r] against [H | P] as ONE operand; False: one contraction per gate."""
    global STEP_WGRAD_ZR_TOGETHER
    STEP_WGRAD_ZR_TOGETHER = bool(on)


def set_step_folded(enabled: bool) -> None:
    global STEP_FOLDED
    STEP_FOLDED = bool(enabled)


# True (default): the window nodes form the gate / conv weight gradients from P^T d_g and H^T d_g (temporal._unfold_gate_grads)
# instead of x3^T d_g and P^T da3: the forward launch then stores no x3 and the backward launch no da3 (38 MB each per snapshot
# at |V| = 50 K), and the gate contractions read [P | H] instead of [x3_g | H].  Exact for an inactive clamp (the fp32 forward
# launch raises the same sticky status word as the folded form when an element of x3 is clamped).
STEP_WGRAD_FROM_P = os.environ.get("STGRAPH_AMD_STEP_WGRAD_FROM_P", "1") != "0"


def set_step_wgrad_from_p(enabled: bool) -> None:
    global STEP_WGRAD_FROM_P
    STEP_WGRAD_FROM_P = bool(enabled)


_FOLD_STATUS = {}


def step_ones_mask(N: int, device) -> torch.Tensor:
    """The clamp mask of an inactive clamp ([N, 12] int32 words 0xffff: the fp32 backward launch's layout) -- what a backward launch
    reads after a folded forward launch that did not form x3.  A fresh tensor per call (one small fill): a window captured into a
    HIP graph must own every buffer its launches read."""
    return torch.full((int(N), 12), 0xffff, dtype=torch.int32, device=torch.device(device))


def step_fold_status_word(device) -> torch.Tensor:
    """The device's sticky int32 word the folded step launches OR a 1 into when the fold was not valid."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _FOLD_STATUS.get(key)
    if t is None:
        t = _FOLD_STATUS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def check_step_fold_status(device=None, clear: bool = True) -> None:
    """Raise if a folded step launch since the last check met a clamped conv output (its results are then wrong).  Synchronises.
    For callers that drive the step launches themselves: the epoch functions of stgraph_amd.temporal do not raise -- they keep a
    snapshot of the training state, read this word once per epoch and rerun the epoch in the reference formulation (_FoldGuard)."""
    for key, t in list(_FOLD_STATUS.items()):
        if device is not None:
            dev = torch.device(device)
            if key != (dev.type, dev.index if dev.index is not None else torch.cuda.current_device()):
                continue
        if int(t.item()) != 0:
            if clear:
                t.zero_()
            raise RuntimeError("a TGCN conv output left [-1e6, 1e6] (reference nn/pytorch/temporal/tgcn.py:23 clamps there): the folded step form "
                               "and the weight gradients formed from P are not valid for this data; results since the last check are "
                               "wrong -- rerun with stgraph_amd.kernels.set_step_folded(False) and set_step_wgrad_from_p(False)")


def tgcn_unfold_gate_grads(Rs, css, Wcs, bcs, Wgs, outs=None):
    """Gate + conv parameter gradients of the three gates from ``R_g = d_g^T [Hx | P]`` [C, C + Fin] and ``cs_g`` (one launch:
    stg_tgcn_unfold_gate_grads; see stgraph_hip.h).  ``outs``: None, or per gate (dWg, dbg, dWc, dbc) tensors to fill (the
    parameters' ``.grad``s).  Returns the list of those four per gate."""
    dev = Rs[0].device
    C, Fin = int(Wgs[0].shape[0]), int(Wcs[0].shape[0])
    srcs = [[_f32(t, "input", dev).contiguous() for t in ts] for ts in (Rs, css, Wcs, bcs, Wgs)]
    for g in range(3):
        if (tuple(srcs[0][g].shape) != (C, C + Fin) or srcs[1][g].numel() != C or tuple(srcs[2][g].shape) != (Fin, C)
                or srcs[3][g].numel() != C or tuple(srcs[4][g].shape) != (C, 2 * C)):
            raise ValueError("tgcn_unfold_gate_grads: R [C, C + Fin], cs [C], Wc [Fin, C], bc [C], Wg [C, 2C] per gate")
    if outs is None:
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        outs = [(new(C, 2 * C), new(C), new(Fin, C), new(C)) for _ in range(3)]
    else:
        for g in range(3):
            for t, shape in zip(outs[g], ((C, 2 * C), (C,), (Fin, C), (C,))):
                if t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or tuple(t.shape) != shape:
                    raise TypeError("tgcn_unfold_gate_grads: outputs must be contiguous fp32 (dWg [C, 2C], dbg [C], dWc [Fin, C], dbc [C])")
    tab = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])  # noqa: E731
    with torch.cuda.device(dev):
        _C.check(_C.lib.stg_tgcn_unfold_gate_grads(*[tab(ts) for ts in srcs], *[tab([outs[g][k] for g in range(3)]) for k in range(4)],
                                                   C, Fin, _stream_ptr(dev)))
    return outs


def tgcn_fold_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, bz, Wr, br, Wh, bh, with_bound: bool = False):
    """``(w_fold [3C, Fin + C], b_fold [3C])`` (+ ``bound`` [2] and ``w_fold_t`` [3 Fin, C], the folded backward launch's operand,
    with ``with_bound``) for the folded forward step launch: row
    ``g C + c`` of w_fold is ``[(Wc_g @ Wg[:, :C].T).T[c] | Wg[c, C:]]`` and ``b_fold[g C + c] = (bc_g @ Wg[:, :C].T + bg)[c]`` -- the
    gate pre-activation ``[P Wc_g + bc_g | H] Wg^T + bg`` as one product of ``[P | H]`` (reference nn/pytorch/temporal/tgcn.py:21-41
    without its clamp); ``bound = (max |Wc|, max |bc|)``.  One launch (stg_tgcn_fold_weights)."""
    Wc = [_f32(t, "conv weight").contiguous() for t in (Wcz, Wcr, Wch)]
    dev = Wc[0].device
    bc = [_f32(t, "conv bias", dev).contiguous() for t in (bcz, bcr, bch)]
    Wg = [_f32(t, "gate weight", dev).contiguous() for t in (Wz, Wr, Wh)]
    bg = [_f32(t, "gate bias", dev).contiguous() for t in (bz, br, bh)]
    Fin, C = (int(v) for v in Wc[0].shape)
    if (any(tuple(t.shape) != (Fin, C) for t in Wc) or any(t.numel() != C for t in bc + bg) or any(tuple(t.shape) != (C, 2 * C) for t in Wg)):
        raise ValueError("tgcn_fold_weights: conv weights [Fin, C], conv / gate biases [C], gate weights [C, 2C]")
    w_fold = torch.empty(3 * C, Fin + C, dtype=torch.float32, device=dev)
    b_fold = torch.empty(3 * C, dtype=torch.float32, device=dev)
    bound = torch.empty(2, dtype=torch.float32, device=dev)
    w_fold_t = torch.empty(3 * Fin, C, dtype=torch.float32, device=dev) if with_bound else None     # the backward launch's operand
    tab = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])  # noqa: E731
    with torch.cuda.device(dev):
        _C.check(_C.lib.stg_tgcn_fold_weights(tab(Wc), tab(bc), tab(Wg), tab(bg), _ptr(w_fold), _ptr(b_fold), _ptr(bound), _ptr(w_fold_t),
                                              C, Fin, _stream_ptr(dev)))
    return (w_fold, b_fold, bound, w_fold_t) if with_bound else (w_fold, b_fold)


def tgcn_step_supported(C: int, Fin: int, Fh: int) -> bool:
    return bool(_C.lib.stg_tgcn_step_supported(int(C), int(Fin), int(Fh)))


def tgcn_step_loss_partials(N: int) -> int:
    return int(_C.lib.stg_tgcn_step_loss_partials(int(N)))


_STEP_INT_FIELDS = ("row_offsets", "column_indices", "node_ids", "link_row_ptr", "link_other", "link_eid")


def _fill_step_args(args, what: str, dev: torch.device, tensors: dict, row_stride: dict | None = None) -> None:
    """Check and store the device pointers of a stg_tgcn_step_*_args block (None -> NULL).  ``row_stride``: names whose tensors are
    [N, cols] column blocks of a wider row-major matrix with that row stride (instead of contiguous)."""
    names = {f[0] for f in args._fields_}
    for name, t in tensors.items():
        if name not in names:
            raise TypeError(f"{what}: unknown argument {name!r}")
        if t is None:
            continue
        if name == "w_image":
            raise TypeError(f"{what}: w_image belonged to the retired bf16-split form (ABI 26)")
        want = torch.int32 if name in _STEP_INT_FIELDS or name in ("clamp_mask", "fold_status") else torch.float32
        if row_stride and name in row_stride:
            if (not torch.is_tensor(t) or t.dtype != want or t.device != dev or t.dim() != 2 or t.stride(1) != 1
                    or (t.shape[0] > 1 and t.stride(0) != row_stride[name]) or t.data_ptr() % 16):
                raise TypeError(f"{what}: {name} must be a [N, cols] {want} column block with row stride {row_stride[name]} on {dev}")
            setattr(args, name, t.data_ptr())
            continue
        if not torch.is_tensor(t) or t.dtype != want or not t.is_cuda or t.device != dev or not t.is_contiguous():
            raise TypeError(f"{what}: {name} must be a contiguous {want} tensor on {dev}")
        setattr(args, name, t.data_ptr())


def tgcn_pack_weights(Wcz, Wcr, Wch, bcz, bcr, bch, Wz, Wr, Wh, W1):
    """``(Wcat, WcatT, b3, WzT, WrT, WhT, W1T)``: the layouts tgcn_step_fwd / _bwd take, from the modules' parameters, in one
    launch (stg_tgcn_pack_weights)."""
    src = [_f32(t, "weight") for t in (Wcz, Wcr, Wch, bcz, bcr, bch, Wz, Wr, Wh, W1)]
    dev = src[0].device
    Fin, C = (int(v) for v in src[0].shape)
    Fh = int(src[9].shape[0])
    if (any(t.device != dev for t in src) or any(tuple(t.shape) != (Fin, C) for t in src[:3]) or any(tuple(t.shape) != (C,) for t in src[3:6])
            or any(tuple(t.shape) != (C, 2 * C) for t in src[6:9]) or tuple(src[9].shape) != (Fh, C)):
        raise ValueError("tgcn_pack_weights: conv weights [Fin, C], conv biases [C], gate weights [C, 2C], head weight [Fh, C]")
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    out = (new(Fin, 3 * C), new(3 * C, Fin), new(3 * C), new(2 * C, C), new(2 * C, C), new(2 * C, C), new(C, Fh))
    with torch.cuda.device(dev):
        _C.check(_C.lib.stg_tgcn_pack_weights(*[_ptr(t) for t in src], *[_ptr(t) for t in out], C, Fin, Fh, _stream_ptr(dev)))
    return out


def tgcn_step_fwd(N: int, C: int, Fin: int, Fh: int, head: int, lo: float, hi: float, device, **tensors) -> None:
    """One TGCN step forward in one launch (stg_tgcn_step_fwd); ``tensors``: the pointer fields of
    stg_tgcn_step_fwd_args by name (include/stgraph_hip.h).  Outputs are written in place."""
    dev = torch.device(device)
    a = _C.TgcnStepFwdArgs()
    if tensors.get("w_fold") is None:
        tensors = dict(tensors, b_fold=None, fold_bound=None)
    if tensors.get("w_fold") is not None and tensors.get("fold_status") is None:
        tensors = dict(tensors, fold_status=step_fold_status_word(dev))
    _fill_step_args(a, "tgcn_step_fwd", dev, tensors)
    a.N, a.C, a.Fin, a.Fh, a.head, a.lo, a.hi = int(N), int(C), int(Fin), int(Fh), int(head), float(lo), float(hi)
    # byte model: gathered input rows + index arrays + what the launch reads and writes per row
    per_row = 4 * (Fin + (3 * C if tensors.get("x3") is not None else 0) + 6 * C + (Fh + 2 if head else 0))
    # flops EXECUTED: the folded form multiplies [P | Hx] (K = Fin + C) per gate, the reference formulation x3 (K = Fin) then [x3 | Hx] (K = 2C)
    gate_macs = 3 * C * (Fin + C) if tensors.get("w_fold") is not None else 3 * C * Fin + 6 * C * C
    with torch.cuda.device(dev), _Timed("tgcn_step_fwd", N * per_row, 2 * N * (gate_macs + (Fh * C if head else 0))):
        _C.check(_C.lib.stg_tgcn_step_fwd(ctypes.byref(a), _stream_ptr(dev)))


def tgcn_step_bwd(N: int, C: int, Fin: int, Fh: int, head: int, lo: float, hi: float, device, link_edges: int = 0,
                  ld_d: int = 0, **tensors) -> None:
    """One TGCN step backward in one launch (stg_tgcn_step_bwd); ``tensors``: the pointer fields of
    stg_tgcn_step_bwd_args by name.  With ``link_row_ptr / link_other / link_eid / link_y / link_logits / link_target`` (and
    ``link_edges`` = the number of label edges) the node side of the link loss's backward runs inside the launch."""
    dev = torch.device(device)
    a = _C.TgcnStepBwdArgs()
    wide = int(ld_d) not in (0, int(C))
    _fill_step_args(a, "tgcn_step_bwd", dev, tensors, {"dzl": int(ld_d), "drl": int(ld_d), "dhl": int(ld_d)} if wide else None)
    a.N, a.C, a.Fin, a.Fh, a.head, a.lo, a.hi = int(N), int(C), int(Fin), int(Fh), int(head), float(lo), float(hi)
    a.link_inv_m = 1.0 / float(link_edges) if link_edges else 0.0
    a.ld_d = int(ld_d)                                 # 3 C: dzl / drl / dhl are column blocks of one [N, 3C] matrix
    per_row = 4 * (6 * C + 3 * C + 3 * C + (3 * C if tensors.get("da3") is not None else 0) + C + Fin + (2 * Fh + 3 if head else 0))
    gate_macs = 3 * C * (Fin + C) if tensors.get("da3") is None else 3 * C * Fin + 6 * C * C       # folded: no da3 (see tgcn_step_fwd)
    with torch.cuda.device(dev), _Timed("tgcn_step_bwd", N * per_row, 2 * N * (gate_macs + (Fh * C if head else 0))):
        _C.check(_C.lib.stg_tgcn_step_bwd(ctypes.byref(a), _stream_ptr(dev)))


def tgcn_window_loss(partials: torch.Tensor, steps: int, N: int, step_loss: torch.Tensor | None = None) -> torch.Tensor:
    """cost [1] = sum over ``steps`` rows of ``partials`` of (row sum) / N (stg_tgcn_window_loss)."""
    partials = _f32(partials, "partials")
    dev = partials.device
    if partials.dim() != 2 or partials.shape[0] < steps:
        raise ValueError("partials must be [steps, tiles]")
    cost = torch.empty(1, dtype=torch.float32, device=dev)
    if step_loss is None:
        step_loss = torch.empty(int(steps), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _C.check(_C.lib.stg_tgcn_window_loss(_ptr(partials), int(steps), int(N), int(partials.stride(0)), _ptr(step_loss),
                                             _ptr(cost), _stream_ptr(dev)))
    return cost


def degree_norm(degrees: torch.Tensor | None = None, row_offsets: torch.Tensor | None = None) -> torch.Tensor:
    """``deg ** -0.5`` with 0 for isolated vertices, [N, 1] (stg_degree_norm_f32): from int32 ``degrees`` or from the
    differences of ``row_offsets``."""
    src = degrees if degrees is not None else row_offsets
    if src is None or src.dtype != torch.int32 or not src.is_cuda or not src.is_contiguous():
        raise TypeError("degree_norm needs a contiguous int32 device tensor")
    N = int(src.shape[0]) - (0 if degrees is not None else 1)
    norm = torch.empty(N, 1, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _C.check(_C.lib.stg_degree_norm_f32(_ptr(degrees), _ptr(None if degrees is not None else row_offsets), _ptr(norm), N,
                                            _stream_ptr(src.device)))
    return norm


def partial_sums_loss(partials: torch.Tensor, steps: int, count: int, inv_n: float,
                      step_loss: torch.Tensor | None = None) -> torch.Tensor:
    """cost [1] = sum over the first ``steps`` rows of ``partials`` of (sum of the row's first ``count`` values) * inv_n."""
    dev = partials.device
    cost = torch.empty(1, dtype=torch.float32, device=dev)
    if step_loss is None:
        step_loss = torch.empty(int(steps), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _C.check(_C.lib.stg_partial_sums_loss(_ptr(partials), int(steps), int(count), int(partials.stride(0)), float(inv_n),
                                              _ptr(step_loss), _ptr(cost), _stream_ptr(dev)))
    return cost


def link_decode_fwd(y: torch.Tensor, edge_index: torch.Tensor, target: torch.Tensor, logits: torch.Tensor,
                    partial: torch.Tensor) -> None:
    """logits[e] = <y[src_e], y[dst_e]> and the per-workgroup sums of the BCE-with-logits terms (stg_link_decode_fwd)."""
    M = int(edge_index.shape[1])
    with torch.cuda.device(y.device):
        _C.check(_C.lib.stg_link_decode_fwd(_ptr(y), _ptr(edge_index), _ptr(target), _ptr(logits), _ptr(partial), M,
                                            int(y.shape[1]), _stream_ptr(y.device)))


LINK_DECODE_JOBS = 32


def link_decode_fwd_window(ys, edge_indices, targets, logits, partials) -> None:
    """:func:`link_decode_fwd` for every snapshot of a window in one launch per 32 snapshots (stg_link_decode_fwd_multi):
    lists of per-snapshot tensors, every ``edge_index`` [2, M] with the same M."""
    n = len(ys)
    M = int(edge_indices[0].shape[1])
    dev = ys[0].device
    for e in edge_indices:
        if e.dtype != torch.int64 or tuple(e.shape) != (2, M) or not e.is_contiguous() or e.device != dev:
            raise ValueError("link_decode_fwd_window: every edge_index must be a contiguous int64 [2, M] tensor on one device")
    with torch.cuda.device(dev):
        for i in range(0, n, LINK_DECODE_JOBS):
            k = min(LINK_DECODE_JOBS, n - i)
            arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts[i:i + k]])  # noqa: E731
            _C.check(_C.lib.stg_link_decode_fwd_multi(k, arr(ys), arr(edge_indices), arr(targets), arr(logits), arr(partials), M,
                                                      int(ys[0].shape[1]), _stream_ptr(dev)))


def link_decode_bwd(g_loss: torch.Tensor, y: torch.Tensor, logits: torch.Tensor, target: torch.Tensor, incidence,
                    dy: torch.Tensor) -> None:
    """dy [N, F]: gradient of the mean BCE loss with respect to y, per node over its incident label edges
    (stg_link_decode_bwd; ``incidence`` from ``link_incidence``)."""
    row_ptr, other, eid = incidence
    with torch.cuda.device(y.device):
        _C.check(_C.lib.stg_link_decode_bwd(_ptr(g_loss), _ptr(y), _ptr(logits), _ptr(target), _ptr(row_ptr), _ptr(other),
                                            _ptr(eid), _ptr(dy), int(y.shape[0]), int(logits.shape[0]), int(y.shape[1]),
                                            _stream_ptr(y.device)))


def tgcn_head_supported(C: int, F: int, O: int) -> bool:
    return bool(_C.lib.stg_tgcn_head_supported(int(C), int(F), int(O)))


def tgcn_head_fwd(h, W1, b1, W2, b2, target, loss_in=None):
    """relu -> Linear -> Linear -> mean squared error of one TGCN step in one launch (stg_tgcn_head_fwd).
    Returns (r, y, y_out [N,1], loss [1]); with ``loss_in`` (a one-element tensor: the loop's running cost) the
    returned loss is ``loss_in + mean(...)`` (stg_tgcn_head_fwd_acc)."""
    N, C = h.shape
    F_ = W1.shape[0]
    dev = h.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    r, y, y_out, loss = new(N, C), new(N, F_), new(N, 1), new(1)
    ws_bytes = int(_C.lib.stg_tgcn_head_workspace_bytes(N))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
    nbytes = 4 * N * (2 * C + F_ + 2)
    with torch.cuda.device(dev), _Timed("tgcn_head_fwd", nbytes, 2 * N * F_ * (C + 1)):
        _C.check(_C.lib.stg_tgcn_head_fwd_acc(_ptr(h), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(target), _ptr(loss_in),
                                              _ptr(r), _ptr(y), _ptr(y_out), _ptr(loss), N, C, F_, _ptr(ws), ws_bytes,
                                              _stream_ptr(dev)))
    return r, y, y_out, loss


def tgcn_head_bwd(g_loss, g_y, g_yout, h, y_out, target, W1, W2):
    """Backward of tgcn_head_fwd up to the weight gradients (stg_tgcn_head_bwd).  Returns (dh, dyt, dyo [N,1])."""
    N, C = h.shape
    F_ = W1.shape[0]
    dev = h.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    dh, dyt, dyo = new(N, C), new(N, F_), new(N, 1)
    nbytes = 4 * N * (2 * C + 2 * F_ + 3)
    with torch.cuda.device(dev), _Timed("tgcn_head_bwd", nbytes, 2 * N * F_ * (C + 1)):
        _C.check(_C.lib.stg_tgcn_head_bwd(_ptr(g_loss), _ptr(g_y), _ptr(g_yout), _ptr(h), _ptr(y_out), _ptr(target),
                                          _ptr(W1), _ptr(W2), _ptr(dh), _ptr(dyt), _ptr(dyo), N, C, F_,
                                          _stream_ptr(dev)))
    return dh, dyt, dyo


_XENT_STATUS = {}


def xent_status(device=None, clear: bool = True) -> int:
    """Read (one device sync) and by default clear the sticky label-status word(s) of ``xent_fwd``: non-zero when
    a label that is neither a class index nor ignore_index (-100) reached the kernel since the last clear (torch
    would have raised a device assert).  ``device`` None: OR over every device used so far."""
    word = 0
    for dev, st in list(_XENT_STATUS.items()):
        if device is None or torch.device(device) == dev:
            word |= int(st.item())
            if clear:
                st.zero_()
    return word


def check_xent_status(device=None) -> None:
    """Raise if ``xent_status`` is set (call it wherever a sync is affordable, e.g. once per epoch)."""
    if xent_status(device):
        raise _C.StgError(_C.STG_ERR_INVALID_ARGUMENT, "cross_entropy: a label outside [0, K) other than ignore_index = -100 was seen; "
                          "its row was left out of the loss and received no gradient")


def xent_fwd(logits: torch.Tensor, labels: torch.Tensor, rows: int | None = None):
    """Mean softmax cross-entropy over the first ``rows`` rows (default: all) (stg_xent_fwd).  Returns (loss [1],
    lse [rows], n_counted [1] float: the rows that took part -- label -100 = ignore_index does not, as in torch,
    status [1] int32: non-zero if another out-of-range label was seen in this or an earlier call on the device -- a
    sticky word that is not read here: ``check_xent_status`` pays the sync when the caller wants it)."""
    n, K = logits.shape
    n = n if rows is None else int(rows)
    dev = logits.device
    if not labels.is_cuda or labels.device != dev:
        raise ValueError(f"xent_fwd: labels on {labels.device}, logits on {dev}")
    lse = torch.empty(n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    n_counted = torch.empty(1, dtype=torch.float32, device=dev)
    status = _XENT_STATUS.get(dev)
    if status is None:                          # one sticky word per device, zeroed once (the kernel only ORs into it)
        status = _XENT_STATUS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = int(_C.lib.stg_xent_workspace_bytes(n, K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _Timed("xent_fwd", 4 * n * (K + 3), 4 * n * K):
        _C.check(_C.lib.stg_xent_fwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(loss), _ptr(n_counted), _ptr(status),
                                     n, K, _ptr(ws), ws_bytes, _stream_ptr(dev)))
    return loss, lse, n_counted, status


_XENT_ONE_PASS = os.environ.get("STGRAPH_AMD_XENT_ONE_PASS", "1") != "0"


def set_xent_one_pass(on: bool) -> None:
    """False: the cross-entropy's gradient comes from its own pass over the logits in the backward (the form before
    stg_xent_fwd_grad)."""
    global _XENT_ONE_PASS
    _XENT_ONE_PASS = bool(on)


def xent_fwd_grad_usable(logits: torch.Tensor) -> bool:
    n_total, K = logits.shape
    return (_XENT_ONE_PASS and logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and n_total > 0
            and logits.data_ptr() % 16 == 0 and int(_C.lib.stg_xent_fwd_grad_workspace_bytes(int(n_total), int(K))) > 0)


def xent_fwd_grad(logits: torch.Tensor, labels: torch.Tensor, rows: int | None = None):
    """:func:`xent_fwd` and, from the same pass, the gradient of the whole logits matrix for an upstream gradient of 1 with its
    column sums (stg_xent_fwd_grad).  Returns (loss, lse, n_counted, status, dlogits, colsum); :func:`xent_scale_grad` applies
    the upstream gradient."""
    n_total, K = logits.shape
    n = n_total if rows is None else int(rows)
    dev = logits.device
    if not labels.is_cuda or labels.device != dev:
        raise ValueError(f"xent_fwd_grad: labels on {labels.device}, logits on {dev}")
    lse = torch.empty(n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    n_counted = torch.empty(1, dtype=torch.float32, device=dev)
    status = _XENT_STATUS.get(dev)
    if status is None:
        status = _XENT_STATUS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    d = torch.empty_like(logits)
    cs = torch.empty(K, dtype=torch.float32, device=dev)
    ws_bytes = int(_C.lib.stg_xent_fwd_grad_workspace_bytes(n_total, K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _Timed("xent_fwd_grad", 4 * n * (2 * K + 3) + 8 * n + 4 * (n_total - n) * K, 4 * n * K):
        _C.check(_C.lib.stg_xent_fwd_grad(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(loss), _ptr(n_counted), _ptr(status), _ptr(d),
                                          _ptr(cs), n, n_total, K, _ptr(ws), ws_bytes, _stream_ptr(dev)))
    return loss, lse, n_counted, status, d, cs


_MM_BWD_SMALL = True


def set_mm_bwd_small(on: bool) -> None:
    """False: the backward of a small dense layer runs as two library GEMMs instead of stg_mm_bwd_small."""
    global _MM_BWD_SMALL
    _MM_BWD_SMALL = bool(on)


def mm_bwd_small_usable(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> bool:
    """``y = x @ w`` with x [N, K], w [K, M] small enough for one workgroup (Cora's second layer: 2708 x 16 -> 7)."""
    return (_MM_BWD_SMALL and g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and w.dtype == torch.float32
            and g.dim() == 2 and x.dim() == 2 and w.dim() == 2 and x.is_contiguous() and w.is_contiguous()
            and x.shape[0] == g.shape[0] and tuple(w.shape) == (x.shape[1], g.shape[1])
            and bool(_C.lib.stg_mm_bwd_small_supported(int(x.shape[0]), int(x.shape[1]), int(g.shape[1]))))


def mm_bwd_small(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor, relu_input: bool = False):
    """(g @ w.T, x.T @ g) in ONE launch (stg_mm_bwd_small); shapes of :func:`mm_bwd_small_usable`.  ``relu_input``: x is the output
    of a ReLU layer -- returns ((g @ w.T) * (x > 0), x.T @ g, column sums of the first): the gradient of that layer's
    pre-activation and its bias gradient from the same launch."""
    g = _f32(g, "g")
    N, K = x.shape
    M = int(g.shape[1])
    gx = torch.empty(N, K, dtype=torch.float32, device=g.device)
    gw = torch.empty(K, M, dtype=torch.float32, device=g.device)
    cs = torch.empty(K, dtype=torch.float32, device=g.device) if relu_input else None
    with torch.cuda.device(g.device), _Timed("mm_bwd_small", 4 * N * (2 * K + M) + 8 * K * M, 4 * N * K * M):
        _C.check(_C.lib.stg_mm_bwd_small(_ptr(g), _ptr(x), _ptr(w), _ptr(gx), _ptr(gw), _ptr(cs), N, K, M, _stream_ptr(g.device)))
    return (gx, gw, cs) if relu_input else (gx, gw)


def gemm_tn_small_usable(a: torch.Tensor, b: torch.Tensor) -> bool:
    """``a.T @ b`` for a [N, Ka] of any width and b [N, Mb <= 16] on a small graph (N <= 65536): stg_gemm_tn_small_f32."""
    return (_MM_BWD_SMALL and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
            and a.shape[0] == b.shape[0] and a.is_contiguous()
            and bool(_C.lib.stg_gemm_tn_small_supported(int(a.shape[0]), int(a.shape[1]), int(b.shape[1]))))


def gemm_tn_small(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    b = _f32(b, "b")
    N, Ka = a.shape
    Mb = int(b.shape[1])
    c = torch.empty(Ka, Mb, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), _Timed("gemm_tn_small", 4 * N * (Ka + Mb) + 4 * Ka * Mb, 2 * N * Ka * Mb):
        _C.check(_C.lib.stg_gemm_tn_small_f32(_ptr(a), _ptr(b), _ptr(c), N, Ka, Mb, _stream_ptr(a.device)))
    return c


def xent_small_usable(logits: torch.Tensor) -> bool:
    """A logits matrix one workgroup walks in a few passes (Cora's 2708 x 7): loss and gradient as ONE launch each way."""
    return (_XENT_SMALL and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.is_contiguous()
            and bool(_C.lib.stg_xent_small_supported(int(logits.shape[0]), int(logits.shape[1]))))


_XENT_SMALL = True


def set_xent_small(on: bool) -> None:
    """False: small logits matrices take the general launches too (tests; the A / B of tools/diag/cora_kernels.py)."""
    global _XENT_SMALL
    _XENT_SMALL = bool(on)


def xent_small_fwd(logits: torch.Tensor, labels: torch.Tensor, rows: int | None = None):
    """:func:`xent_fwd` for a small matrix (stg_xent_small_fwd: one launch, no workspace).  Returns (loss, lse, n_counted, status)."""
    n_total, K = logits.shape
    n = n_total if rows is None else int(rows)
    dev = logits.device
    if not labels.is_cuda or labels.device != dev:
        raise ValueError(f"xent_small_fwd: labels on {labels.device}, logits on {dev}")
    lse = torch.empty(n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    n_counted = torch.empty(1, dtype=torch.float32, device=dev)
    status = _XENT_STATUS.get(dev)
    if status is None:
        status = _XENT_STATUS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev), _Timed("xent_small_fwd", 4 * n * (K + 1) + 8 * n, 4 * n * K):
        _C.check(_C.lib.stg_xent_small_fwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(loss), _ptr(n_counted), _ptr(status), n, K,
                                           _stream_ptr(dev)))
    return loss, lse, n_counted, status


def xent_small_bwd(g_loss: torch.Tensor, logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, n_counted: torch.Tensor):
    """(gradient of the whole logits matrix, its column sums) from one launch (stg_xent_small_bwd); shapes of :func:`xent_small_usable`."""
    n_total, K = logits.shape
    n = int(lse.shape[0])
    d = torch.empty_like(logits)
    cs = torch.empty(K, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device), _Timed("xent_small_bwd", 4 * n * K + 4 * n_total * K + 12 * n, 4 * n * K):
        _C.check(_C.lib.stg_xent_small_bwd(_ptr(g_loss), _ptr(logits), _ptr(labels), _ptr(lse), _ptr(n_counted), _ptr(d), _ptr(cs),
                                           n, n_total, K, _stream_ptr(logits.device)))
    return d, cs


def xent_scale_grad(d: torch.Tensor, colsum: torch.Tensor | None, g_loss: torch.Tensor) -> None:
    """``d *= g_loss`` (and ``colsum``) in place; a launch that returns at once when ``g_loss`` is exactly 1."""
    n_total, K = d.shape
    with torch.cuda.device(d.device):
        _C.check(_C.lib.stg_xent_scale_grad(_ptr(d), _ptr(colsum), _ptr(g_loss), n_total, K, _stream_ptr(d.device)))


def xent_bwd(g_loss: torch.Tensor, logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor,
             n_counted: torch.Tensor, want_colsum: bool = False):
    """Gradient of xent_fwd for the WHOLE logits matrix: rows beyond ``lse.shape[0]`` (not part of the loss) and rows
    the forward did not count are zero.  ``want_colsum``: returns ``(gradient, gradient.sum(0))`` from the same launch
    (stg_xent_bwd_colsum) -- the column sums are None for a shape that entry point does not cover."""
    n_total, K = logits.shape
    n = int(lse.shape[0])
    d = torch.empty_like(logits)
    if want_colsum:
        ws_bytes = int(_C.lib.stg_xent_bwd_colsum_workspace_bytes(n_total, K))
        if ws_bytes == 0 or logits.data_ptr() % 16 or d.data_ptr() % 16:
            return xent_bwd(g_loss, logits, labels, lse, n_counted), None
        cs = torch.empty(K, dtype=torch.float32, device=logits.device)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=logits.device)
        with torch.cuda.device(logits.device), _Timed("xent_bwd", 4 * n * (2 * K + 3) + 4 * (n_total - n) * K, 4 * n * K):
            _C.check(_C.lib.stg_xent_bwd_colsum(_ptr(g_loss), _ptr(logits), _ptr(labels), _ptr(lse), _ptr(n_counted), _ptr(d),
                                                _ptr(cs), n, n_total, K, _ptr(ws), ws_bytes, _stream_ptr(logits.device)))
        return d, cs
    with torch.cuda.device(logits.device), _Timed("xent_bwd", 4 * n * (2 * K + 3) + 4 * (n_total - n) * K, 4 * n * K):
        _C.check(_C.lib.stg_xent_bwd(_ptr(g_loss), _ptr(logits), _ptr(labels), _ptr(lse), _ptr(n_counted), _ptr(d), n,
                                     n_total, K, _stream_ptr(logits.device)))
    return d


def link_head_supported(C: int, F: int) -> bool:
    return bool(_C.lib.stg_link_head_supported(int(C), int(F)))


def link_incidence(edge_index: torch.Tensor, N: int):
    """Node-sorted incidence list of ``M`` label edges (``edge_index`` int64 [2, M]): ``row_ptr [N+1]``, and per
    entry the other endpoint and the edge id (int32), entries of a node in ascending (role, edge id) order --
    what stg_link_head_bwd sums over.  Built with a stable sort, once per index tensor."""
    M = int(edge_index.shape[1])
    nodes = torch.cat([edge_index[0], edge_index[1]])
    order = torch.sort(nodes, stable=True).indices
    other = torch.cat([edge_index[1], edge_index[0]])[order].to(torch.int32).contiguous()
    eid = (order % M).to(torch.int32).contiguous()
    counts = torch.bincount(nodes, minlength=N)
    row_ptr = torch.zeros(N + 1, dtype=torch.int32, device=edge_index.device)
    row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return row_ptr, other, eid


def link_head_fwd(h, W1, b1, edge_index, target, loss_in=None):
    """relu -> Linear -> dot-product decode -> BCE-with-logits mean, three launches (stg_link_head_fwd).
    Returns (r, y, logits [M], loss [1])."""
    N, C = h.shape
    F_ = W1.shape[0]
    M = int(edge_index.shape[1])
    dev = h.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    r, y, logits, loss = new(N, C), new(N, F_), new(M), new(1)
    ws_bytes = int(_C.lib.stg_link_head_workspace_bytes(M))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _Timed("link_head_fwd", 4 * N * (2 * C + F_) + 4 * M * (2 * F_ + 6), 2 * N * F_ * C):
        _C.check(_C.lib.stg_link_head_fwd(_ptr(h), _ptr(W1), _ptr(b1), _ptr(edge_index[0]), _ptr(edge_index[1]),
                                          _ptr(target), _ptr(loss_in), _ptr(r), _ptr(y), _ptr(logits), _ptr(loss), N, M, C,
                                          F_, _ptr(ws), ws_bytes, _stream_ptr(dev)))
    return r, y, logits, loss


def link_head_bwd(g_loss, g_y, h, y, logits, target, incidence, W1):
    """Backward of link_head_fwd up to the weight gradients (stg_link_head_bwd): returns (dh, dyt)."""
    N, C = h.shape
    F_ = W1.shape[0]
    M = int(logits.shape[0])
    dev = h.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    dy, dh, dyt = new(N, F_), new(N, C), new(N, F_)
    row_ptr, other, eid = incidence
    with torch.cuda.device(dev), _Timed("link_head_bwd", 4 * N * (2 * C + 3 * F_) + 4 * M * (2 * F_ + 6), 2 * N * F_ * C):
        _C.check(_C.lib.stg_link_head_bwd(_ptr(g_loss), _ptr(g_y), _ptr(h), _ptr(y), _ptr(logits), _ptr(target),
                                          _ptr(row_ptr), _ptr(other), _ptr(eid), _ptr(W1), _ptr(dy), _ptr(dh), _ptr(dyt),
                                          N, M, C, F_, _stream_ptr(dev)))
    return dh, dyt


def tgcn_cell_fused_supported(C: int) -> bool:
    return bool(_C.lib.stg_tgcn_cell_fused_supported(int(C)))


def tgcn_cell_fused_fwd(a3, b3, H, Wz, bz, Wr, br, Wh, bh, lo: float, hi: float):
    """The forward row-local chain of one TGCN step in one launch (stg_tgcn_cell_fused_fwd).
    Returns (Hn, (CZ, CR, CH, Z, R, Ht))."""
    N, C = H.shape
    dev = H.device
    ins = (a3, b3, H, Wz, bz, Wr, br, Wh, bh)
    for t in ins:
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.device != dev:
            raise RuntimeError("tgcn_cell_fused_fwd: operands must be contiguous fp32 tensors on one HIP device")
    if a3.shape != (N, 3 * C) or b3.numel() != 3 * C or any(w.shape != (C, 2 * C) for w in (Wz, Wr, Wh)) or \
            any(b.numel() != C for b in (bz, br, bh)):
        raise ValueError("tgcn_cell_fused_fwd: operand shapes do not match hidden width C")
    new = lambda w: torch.empty(N, w, dtype=torch.float32, device=dev)  # noqa: E731
    CZ, CR, CH, Z, R, Ht, Hn = new(2 * C), new(2 * C), new(2 * C), new(C), new(C), new(C), new(C)
    outs = (CZ, CR, CH, Z, R, Ht, Hn)
    with torch.cuda.device(dev), _Timed("tgcn_cell_fused_fwd", 4 * N * C * 14, 12 * N * C * C):
        _C.check(_C.lib.stg_tgcn_cell_fused_fwd(*[_ptr(t) for t in ins], *[_ptr(t) for t in outs], N, C,
                                                float(lo), float(hi), _stream_ptr(dev)))
    return Hn, (CZ, CR, CH, Z, R, Ht)


def tgcn_cell_fused_bwd_dx_supported(C: int, Fin: int) -> bool:
    return bool(_C.lib.stg_tgcn_cell_fused_bwd_dx_supported(int(C), int(Fin)))


def tgcn_cell_fused_bwd(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, lo: float, hi: float, Wcat=None):
    """The backward row-local chain of one TGCN step in one launch (stg_tgcn_cell_fused_bwd).
    Returns (da3, dH, dzl, drl, dhl); with ``Wcat`` [Fin, 3C] also ``dx = da3 @ Wcat.T`` as a sixth element
    (stg_tgcn_cell_fused_bwd_dx)."""
    N, C = H.shape
    dev = H.device
    ins = (dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh)
    for t in ins:
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.device != dev:
            raise RuntimeError("tgcn_cell_fused_bwd: operands must be contiguous fp32 tensors on one HIP device")
    if a3.shape != (N, 3 * C) or b3.numel() != 3 * C or any(w.shape != (C, 2 * C) for w in (Wz, Wr, Wh)) or \
            any(t.shape != (N, C) for t in (dHn, Z, Ht, R)):
        raise ValueError("tgcn_cell_fused_bwd: operand shapes do not match hidden width C")
    new = lambda w: torch.empty(N, w, dtype=torch.float32, device=dev)  # noqa: E731
    dhl, dzl, drl, da3, dH = new(C), new(C), new(C), new(3 * C), new(C)
    if Wcat is not None:
        Fin = int(Wcat.shape[0])
        if Wcat.shape != (Fin, 3 * C) or not Wcat.is_contiguous() or Wcat.dtype != torch.float32 or Wcat.device != dev:
            raise ValueError("tgcn_cell_fused_bwd: Wcat must be a contiguous fp32 [Fin, 3C] tensor on the same device")
        dx = new(Fin)
        with torch.cuda.device(dev), _Timed("tgcn_cell_fused_bwd", 4 * N * (C * 15 + Fin), 12 * N * C * C + 6 * N * C * Fin):
            _C.check(_C.lib.stg_tgcn_cell_fused_bwd_dx(*[_ptr(t) for t in ins], _ptr(Wcat), _ptr(dhl), _ptr(dzl), _ptr(drl),
                                                       _ptr(da3), _ptr(dH), _ptr(dx), N, C, Fin, float(lo), float(hi),
                                                       _stream_ptr(dev)))
        return da3, dH, dzl, drl, dhl, dx
    with torch.cuda.device(dev), _Timed("tgcn_cell_fused_bwd", 4 * N * C * 15, 12 * N * C * C):
        _C.check(_C.lib.stg_tgcn_cell_fused_bwd(*[_ptr(t) for t in ins], _ptr(dhl), _ptr(dzl), _ptr(drl), _ptr(da3), _ptr(dH),
                                                N, C, float(lo), float(hi), _stream_ptr(dev)))
    return da3, dH, dzl, drl, dhl


def tgcn_cell_call(name: str, tensors, N: int, C: int, *scalars) -> None:
    """Launch one fused TGCN row-local stage (stg_tgcn_cell_<name>); tensors are validated here."""
    dev = tensors[0].device
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"tgcn_cell_{name}: operands must be contiguous fp32 tensors on one HIP device")
    fn = getattr(_C.lib, "stg_tgcn_cell_" + name)
    with torch.cuda.device(dev):
        _C.check(fn(*[_ptr(t) for t in tensors], N, C, *[float(x) for x in scalars], _stream_ptr(dev)))


# ------------------------------------------------------------------------------------------------------- every switch, in one place
# (name, module, attribute, default, environment variable or None, what it selects).  Setters: ``set_<name>`` in the named module
# unless noted; results never depend on a switch beyond fp32 rounding (reference_compat excepted: it reproduces reference defect D1).
_KNOBS = (
    ("reference_compat", "stgraph_amd.kernels", "_REF_COMPAT", False, None, "reproduce reference defect D1 (columns >= the power-of-two width stay 0 for F < 64)"),
    ("direct_build", "stgraph_amd.kernels", "_DIRECT_BUILD", True, None, "counting-sort CSR build for graphs of <= 2M edges (else the radix-sort build)"),
    ("fused_rebuild", "stgraph_amd.kernels", "FUSED_REBUILD", True, None, "re-builds of a validated edge list as one fused launch sequence"),
    ("long_row_path", "stgraph_amd.kernels", "_LONG_ROWS", True, None, "hub rows of the aggregation on their own workgroups"),
    ("edge_cache", "stgraph_amd.kernels", "_EDGE_CACHE", True, None, "per-edge pre-gathered norm / weight scalars kept on the CSR object"),
    ("gat_ones_shortcut", "stgraph_amd.kernels", "_GAT_ONES", True, None, "GAT K0 without the A write when every score is finite (device flag)"),
    ("gat_uniform_form", "stgraph_amd.kernels", "_GAT_UNIFORM", True, None, "GAT K1 at the input width when every A == 1 (SURVEY.md D2)"),
    ("gat_uniform_backward", "stgraph_amd.kernels", "_GAT_UNIFORM_BWD", True, "STGRAPH_AMD_GAT_UNIFORM_BWD", "GAT K2 at the input width, likewise"),
    ("gat_factored_backward", "stgraph_amd.kernels", "_GAT_FACTORED", True, None, "GAT K2 with one E x H x D gather instead of the emitted unit's two"),
    ("gat_regrouped_er", "stgraph_amd.kernels", "_GAT_REGROUPED_ER", True, None, "grad_er summed per target without atomics"),
    ("gat_attn_fold", "stgraph_amd.kernels", "_GAT_ATTN_FOLD", True, None, "GAT projection fold: its small products (attention gradients, A_w, the weight-gradient correction) in one launch"),
    ("gat_prepass_heads", "stgraph_amd.kernels", "_GAT_PREPASS_HEADS", True, None, "uniform GAT backward: per-vertex pass and the per-head products g W_h in one pass over g and out"),
    ("native_rowgemm", "stgraph_amd.kernels", "_ROWGEMM", False, "STGRAPH_AMD_ROWGEMM", "round-3 fp32 row-product kernel for every tall product (off: only the 16-row form below)"),
    ("rowgemm16", "stgraph_amd.kernels", "_ROWGEMM16", True, "STGRAPH_AMD_ROWGEMM16", "tall row products at K, M in {64, 128} on the native kernels (bf16 split from 64 K rows)"),
    ("relu_bits", "stgraph_amd.kernels", "_RELU_BITS", True, "STGRAPH_AMD_RELU_BITS", "ReLU sign pattern as bits; the layer above masks its input gradient in the launch that forms it"),
    ("step_folded", "stgraph_amd.kernels", "STEP_FOLDED", True, "STGRAPH_AMD_STEP_FOLDED", "TGCN step launches with the conv folded into the gate Linears"),
    ("step_wgrad_from_p", "stgraph_amd.kernels", "STEP_WGRAD_FROM_P", True, "STGRAPH_AMD_STEP_WGRAD_FROM_P", "TGCN weight gradients from P (no x3 / da3 stored)"),
    ("step_wgrad_zr_together", "stgraph_amd.kernels", "STEP_WGRAD_ZR_TOGETHER", True, "STGRAPH_AMD_STEP_WGRAD_ZR_TOGETHER", "[d_z | d_r] contracted as one operand"),
    ("xent_one_pass", "stgraph_amd.kernels", "_XENT_ONE_PASS", True, "STGRAPH_AMD_XENT_ONE_PASS", "cross-entropy loss and its gradient in one pass"),
    ("mm_bwd_small", "stgraph_amd.kernels", "_MM_BWD_SMALL", True, None, "backward of a small dense layer (N <= 65536, K, M <= 16) as one launch instead of two library GEMMs"),
    ("xent_small", "stgraph_amd.kernels", "_XENT_SMALL", True, None, "cross-entropy of a small logits matrix (one workgroup holds it in registers; K <= 64) as one launch each way"),
    ("native_weight_grad", "stgraph_amd.nn.functional", "_NATIVE_WGRAD", True, None, "tall-skinny weight gradients on the split-K kernels instead of rocBLAS"),
    ("deferred_weight_grads", "stgraph_amd.nn.functional", "_DEFER", True, None, "weight gradients of a window contracted once, at the end of the backward pass"),
    ("input_layer_reorder", "stgraph_amd.nn.functional", "_INPUT_LAYER", True, None, "a GCNConv whose input needs no gradient aggregates first"),
    ("gat_fc", "stgraph_amd.nn.functional", "_GAT_FC", True, None, "GATConv projection + el / er in one launch"),
    ("gat_proj_fold", "stgraph_amd.nn.functional", "_GAT_PROJ_FOLD", True, None, "GAT projection backward folded into the input / weight gradients"),
    ("fused_head", "stgraph_amd.temporal", "_FUSED_HEAD", True, None, "relu + both Linears + loss of the temporal harness model in one launch"),
    ("fused_window", "stgraph_amd.temporal", "_FUSED_WINDOW", True, None, "one autograd node per BPTT window over the step launches"),
    ("fused_forward", "stgraph_amd.nn.pytorch.temporal.cell", "_FUSED_FWD", True, None, "TGCN cell row-local chain in one launch (per-step API path)"),
    ("fused_backward", "stgraph_amd.nn.pytorch.temporal.cell", "_FUSED_BWD", True, None, "its backward, likewise"),
    ("fused_dx", "stgraph_amd.nn.pytorch.temporal.cell", "_FUSED_DX", True, None, "da3 Wcat^T inside that backward launch"),
    ("pcsr_fused_step", "stgraph_amd.graph.dynamic.pcsr.pcsr", "FUSED_STEP", True, None, "dynamic edge store: merge + CSR emission as one device step (module attribute, no setter)"),
    ("force_generated", "stgraph_amd.compiler.dispatch", "_FORCE_GENERATED", False, None, "always run the generated (hiprtc) kernel instead of a hand-written unit"),
)
ENVIRONMENT = ("STGRAPH_AMD_LIB",) + tuple(k[4] for k in _KNOBS if k[4])


def knobs() -> dict:
    """Every switch of the package and the native library: name -> {value, default, env, module, what} (bench.py prints it into
    bench_detail.json; ``native`` = stg_set_tuning keys, 0 = auto)."""
    import importlib
    out = {}
    for name, module, attr, default, env, what in _KNOBS:
        try:
            value = getattr(importlib.import_module(module), attr)
        except Exception as exc:                                     # noqa: BLE001  (a listing must not fail the caller)
            value = f"unavailable: {type(exc).__name__}"
        out[name] = {"value": value, "default": default, "env": env, "module": module, "what": what}
    out["native"] = _C.tuning_values()
    out["library"] = {"path": _C.LIB_PATH, "env": "STGRAPH_AMD_LIB", "abi": _C.ABI_VERSION}
    return out


def non_default_knobs() -> dict:
    """The switches that differ from their defaults (what a bench line should say about its configuration)."""
    k = knobs()
    out = {n: v["value"] for n, v in k.items() if isinstance(v, dict) and "default" in v and v["value"] != v["default"]}
    out.update({"native." + n: v for n, v in k["native"].items() if v})
    return out
