"""PCSR dynamic-graph store (reference: stgraph/graph/dynamic/pcsr/)."""
