"""``CSR`` -- counterpart of the reference's pybind class (graph/static/csr.cu:35-201).

Same constructor, same read/write attributes (``row_offset_ptr``,
``column_indices_ptr``, ``eids_ptr``, ``node_ids_ptr``, ``out_degrees``,
``in_degrees``, ``weighted_out_degrees``), ``__copy__``/``__deepcopy__`` that share
the device arrays, and ``get_array(ptr, size)``.  The arrays are built by
``stg_csr_ctor_host`` and uploaded with torch (the reference does host build +
cudaMemcpy as well); whole graphs should use ``kernels.build_graph_csr`` instead,
which builds both directions on the GPU.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch

from ... import kernels

# address -> live tensor, so get_array(ptr, size) can read device arrays (weak: nothing is pinned)
_LIVE: "weakref.WeakValueDictionary[int, torch.Tensor]" = weakref.WeakValueDictionary()


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class CSR:
    def __init__(self, edge_list, edge_weight, num_nodes: int, is_edge_reverse: bool = False,
                 device: torch.device | str | None = None):
        arr = np.asarray(edge_list, dtype=np.int64).reshape(-1, 3)
        if arr.size and (arr.min() < np.iinfo(np.int32).min or arr.max() > np.iinfo(np.int32).max):
            raise ValueError("edge list values do not fit int32")
        h = kernels.csr_ctor_host(arr[:, 0], arr[:, 1], arr[:, 2], edge_weight, num_nodes, is_edge_reverse)
        dev = torch.device(device) if device is not None else default_device()
        up = lambda k: torch.from_numpy(h[k]).to(dev)  # noqa: E731
        self._csr = kernels.DeviceCSR(up("row_offset"), up("column_indices"), up("eids"), up("node_ids"),
                                      degree_sorted=True)      # csr.cu:142-154
        self.out_degrees = h["out_degrees"].tolist()
        self.in_degrees = h["in_degrees"].tolist()
        self.weighted_out_degrees = h["weighted_out_degrees"].tolist()
        self._publish()

    def _publish(self) -> None:
        c = self._csr
        self.row_offset_ptr = c.row_offset_ptr
        self.column_indices_ptr = c.column_indices_ptr
        self.eids_ptr = c.eids_ptr
        self.node_ids_ptr = c.node_ids_ptr
        for t in (c.row_offset, c.column_indices, c.eids, c.node_ids):
            _LIVE[t.data_ptr()] = t

    @property
    def device_csr(self) -> kernels.DeviceCSR:
        return self._csr

    def __copy__(self):                      # csr.cu:193-199: copies share the device arrays
        new = object.__new__(CSR)
        new.__dict__.update(self.__dict__)
        return new

    def __deepcopy__(self, memo):
        return self.__copy__()


def get_array(ptr: int, size: int) -> list:
    """Read ``size`` int32 values starting at address ``ptr`` (csr.cu:172-179)."""
    for base, t in list(_LIVE.items()):
        nbytes = t.numel() * 4
        if base <= ptr < base + max(nbytes, 1):
            off = (ptr - base) // 4
            return t.reshape(-1)[off:off + size].cpu().tolist()
    return list((ctypes.c_int32 * size).from_address(ptr))
