"""Constants shared with the reference (stgraph/utils/constants.py:6-17)."""
from enum import Enum


class SizeConstants(Enum):
    NODE_NORM_SIZE = 2
