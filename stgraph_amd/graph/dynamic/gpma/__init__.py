"""GPMA dynamic-graph store (reference: stgraph/graph/dynamic/gpma/)."""
