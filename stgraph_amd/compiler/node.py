"""Symbolic neighbourhood handed to a vertex function (reference compiler/node.py:7-26).

``v.innbs`` is a list holding ONE symbolic in-neighbour, ``v.inedges`` ONE symbolic
in-edge with ``.src`` / ``.dst``; only the in-direction is populated with features
(reference compiler/stgraph.py:99-113).  ``outnbs`` / ``outedges`` exist for API
parity and stay empty of features, as in the reference.
"""
from __future__ import annotations


class NbNode:
    def __init__(self, center, direction: str):
        self._central_node = center
        self._direction = direction


class NbEdge:
    def __init__(self, center, direction: str, nbnode):
        self._direction = direction
        if direction == "in":
            self.src, self.dst = nbnode, center
        else:
            self.src, self.dst = center, nbnode


class CentralNode:
    def __init__(self):
        self.innbs = [NbNode(self, "in")]
        self.outnbs = [NbNode(self, "out")]
        self.inedges = [NbEdge(self, "in", nb) for nb in self.innbs]
        self.outedges = [NbEdge(self, "out", nb) for nb in self.outnbs]
