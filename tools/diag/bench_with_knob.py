"""bench.py under one stg_set_tuning knob: python tools/diag/bench_with_knob.py <key> <value> [bench.py args...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

torch.cuda.init()
import bench
from stgraph_amd import _C

key, value = sys.argv[1], int(sys.argv[2])
_C.set_tuning(key, value)
sys.argv = ["bench.py"] + sys.argv[3:]
bench.main()
