"""reference compiler/backend/pytorch/torch_kernel_wrapper.py:4-6"""
import torch

from ..kernel_wrapper import KernelWrapper


class KernelWrapperTorch(KernelWrapper, torch.autograd.Function):
    pass
