"""``DynamicGraph`` -- timestamp protocol of the reference's dynamic graphs
(graph/dynamic/dynamic_graph.py:16-188).

``get_graph(t)`` moves the forward view to snapshot t (monotonically, as the
training loop advances); the first ``get_backward_graph(t)`` after a forward
phase switches to backprop state and walks the view BACKWARDS, which is how the
executor's timestamp stack replays BPTT (compiler/executor.py:385-387).
Node data is stored per timestamp (dynamic_graph.py:138-153).
"""
from __future__ import annotations

import time
from abc import abstractmethod

from ..stgraph_base import STGraphBase
from ..static.static_graph import edge_arrays


class DynamicGraph(STGraphBase):
    def __init__(self, edge_list, max_num_nodes: int) -> None:
        super().__init__()
        self.max_num_nodes = int(max_num_nodes)
        self._num_timestamps = len(edge_list)
        self._edge_list_ref = edge_list
        self._graph_updates = None          # built on first access (only GPMA/PCSR-style stores need it)
        self._distinct_edges = {}           # t -> len(set(edges_t)), filled by subclasses

        self._is_backprop_state = False
        self.current_timestamp = 0

        self.get_fwd_graph_time = 0
        self.get_bwd_graph_time = 0
        self.move_to_gpu_time = 0

    # -- per-timestamp add/delete lists, sorted by (dst, src) (dynamic_graph.py:56-79) --------------
    @property
    def graph_updates(self) -> dict:
        if self._graph_updates is None:
            sets = []
            for t in range(self._num_timestamps):
                s, d = edge_arrays(self._edge_list_ref[t])
                sets.append(set(zip(map(int, s), map(int, d))))
            key = lambda x: (x[1], x[0])  # noqa: E731
            upd = {"0": {"add": sorted(sets[0], key=key) if sets else [], "delete": []}}
            for t in range(1, len(sets)):
                upd[str(t)] = {"add": sorted(sets[t] - sets[t - 1], key=key),
                               "delete": sorted(sets[t - 1] - sets[t], key=key)}
            self._graph_updates = upd
        return self._graph_updates

    @property
    def graph_attr(self) -> dict:
        return {str(t): (self.max_num_nodes, self._num_edges_at(t)) for t in range(self._num_timestamps)}

    def reset_graph(self) -> None:
        self._get_cached_graph("base")
        self.current_timestamp = 0
        self._is_backprop_state = False
        self._on_timestamp_change()
        self.get_fwd_graph_time = 0
        self.get_bwd_graph_time = 0
        self.move_to_gpu_time = 0

    def get_graph(self, timestamp: int) -> None:
        t0 = time.time()
        self._is_backprop_state = False
        if timestamp < self.current_timestamp:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_forward()")
        if self._get_cached_graph(timestamp - 1):
            self.current_timestamp = timestamp - 1
        if self.current_timestamp == timestamp:
            self._on_timestamp_change()          # re-publish forward pointers after a backward phase
        while self.current_timestamp < timestamp:
            self._update_graph_forward()
            self.current_timestamp += 1
        self.get_fwd_graph_time += time.time() - t0

    def get_backward_graph(self, timestamp: int) -> None:
        t0 = time.time()
        if not self._is_backprop_state:
            self._cache_graph()
            self._is_backprop_state = True
            self._init_reverse_graph()
        if timestamp > self.current_timestamp:
            raise RuntimeError("⏰ Invalid timestamp during STGraphBase.update_graph_backward()")
        while self.current_timestamp > timestamp:
            self._update_graph_backward()
            self.current_timestamp -= 1
        self.get_bwd_graph_time += time.time() - t0

    def get_num_nodes(self) -> int:
        return self.max_num_nodes

    def get_num_edges(self) -> int:
        return self._num_edges_at(self.current_timestamp)

    def get_ndata(self, field: str):
        return self._ndata.get(str(self.current_timestamp), {}).get(field, None)

    def set_ndata(self, field: str, val) -> None:
        self._ndata.setdefault(str(self.current_timestamp), {})[field] = val

    def _on_timestamp_change(self) -> None:
        pass

    @abstractmethod
    def _num_edges_at(self, timestamp: int) -> int: ...

    @abstractmethod
    def in_degrees(self): ...

    @abstractmethod
    def out_degrees(self): ...

    @abstractmethod
    def _cache_graph(self) -> None: ...

    @abstractmethod
    def _get_cached_graph(self, timestamp) -> bool: ...

    @abstractmethod
    def _update_graph_forward(self) -> None: ...

    @abstractmethod
    def _init_reverse_graph(self) -> None: ...

    @abstractmethod
    def _update_graph_backward(self) -> None: ...
