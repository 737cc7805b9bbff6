"""Argument checks of the entry points added with ABI 25: every bad call comes back as an error code with a message (RuntimeError
through _C.check), none launches."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _raises(fn, *args):
    from stgraph_amd import _C
    with pytest.raises(RuntimeError):
        _C.check(fn(*args))


def test_bad_arguments_are_refused(cuda):
    from stgraph_amd import _C
    lib = _C.lib
    x = torch.zeros(64, 64, device=cuda)
    p = x.data_ptr()
    flag = torch.zeros(1, dtype=torch.int32, device=cuda)
    # the uniform-attention backward unit: one head shape only
    assert lib.stg_gat_bwd_uniform_supported(8, 64, 64) and not lib.stg_gat_bwd_uniform_supported(4, 64, 64)
    assert not lib.stg_gat_bwd_uniform_supported(8, 64, 32) and not lib.stg_gat_bwd_uniform_supported(8, 32, 64)
    _raises(lib.stg_gat_bwd_prepass, p, p, p, None, p, 8, 4, 64, 0.2, None, None)                   # H = 4
    _raises(lib.stg_gat_bwd_prepass, None, p, p, None, p, 8, 8, 64, 0.2, None, None)                # NULL S
    _raises(lib.stg_gat_bwd_uniform_edges, *([p] * 10 + [None] + [p] * 8), 8, 0.2, flag.data_ptr(), None)   # NULL gxa
    _raises(lib.stg_gat_bwd_uniform_edges, *([p] * 19), 8, 0.2, None, None)                          # NULL flag
    _raises(lib.stg_gat_bwd_uniform_gx_fallback, p, p, p, 8, None, None)
    _raises(lib.stg_gat_fc_feat_if, p, p, p, 64, 64, 8, 64, None, None)
    # gated contraction: a gate and a polarity
    ws = torch.empty(int(lib.stg_gemm_tn_workspace_bytes(64, 64, 64)) + 16, dtype=torch.uint8, device=cuda)
    _raises(lib.stg_gemm_tn_gated_f32, p, p, p, 64, 64, 64, ws.data_ptr(), ws.numel(), None, 1, None)
    _raises(lib.stg_gemm_tn_gated_f32, p, p, p, 64, 64, 64, ws.data_ptr(), ws.numel(), flag.data_ptr(), 0, None)
    _raises(lib.stg_gemm_tn_gated_f32, p, p, p, 64, 64, 64, ws.data_ptr(), ws.numel(), flag.data_ptr(), 3, None)
    # per-head row products
    assert lib.stg_rowgemm_heads_supported(1000, 64, 64, 8) and not lib.stg_rowgemm_heads_supported(1000, 48, 64, 8)
    assert not lib.stg_rowgemm_heads_supported(1000, 64, 64, 0) and not lib.stg_rowgemm_heads_supported(1 << 26, 128, 64, 8)
    _raises(lib.stg_rowgemm_heads_f32, p, p, p, 64, 48, 64, 1, None)
    _raises(lib.stg_rowgemm_heads_f32, None, p, p, 64, 64, 64, 1, None)
    _raises(lib.stg_rowgemm_heads_f32, p + 4, p, p, 16, 64, 64, 1, None)                             # alignment
    # the ReLU bit pattern
    assert lib.stg_rowgemm_bits_words(0) == 0 and lib.stg_rowgemm_bits_words(33) == 256
    bits = torch.zeros(256, dtype=torch.int32, device=cuda)
    _raises(lib.stg_rowgemm_act_bits_f32, p, p, None, p, 64, 64, 64, 0, 1, bits.data_ptr(), bits.data_ptr(), None)   # both
    _raises(lib.stg_rowgemm_act_bits_f32, p, p, None, p, 64, 64, 64, 0, 0, None, bits.data_ptr(), None)              # bits_out without ReLU
    _raises(lib.stg_rowgemm_act_bits_f32, p, p, None, p, 64, 64, 64, 0, 0, bits.data_ptr(), None, None)              # bits_in without trans_w
    _raises(lib.stg_rowgemm_act_bits_f32, p, p, None, p, 64, 96, 64, 0, 1, None, bits.data_ptr(), None)              # K = 96
    # one-pass cross-entropy
    assert lib.stg_xent_fwd_grad_workspace_bytes(100, 7) == 0 and lib.stg_xent_fwd_grad_workspace_bytes(100, 8) > 0
    _raises(lib.stg_xent_fwd_grad, p, p, p, p, p, p, p, p, 10, 5, 8, p, 1 << 20, None)              # n_total < n
    _raises(lib.stg_xent_fwd_grad, p, p, p, p, p, p, p, p, 10, 10, 7, p, 1 << 20, None)             # K % 4
    _raises(lib.stg_xent_scale_grad, None, None, p, 10, 8, None)


def test_step_backward_refuses_a_wrong_gate_gradient_stride(cuda):
    from stgraph_amd import _C, kernels
    a = _C.TgcnStepBwdArgs()
    a.N, a.C, a.Fin, a.Fh, a.head, a.lo, a.hi = 16, 64, 32, 32, 1, -1e6, 1e6
    a.ld_d = 128                                            # neither C nor 3 C
    with pytest.raises(RuntimeError, match="ld_d"):
        _C.check(_C.lib.stg_tgcn_step_bwd(ctypes.byref(a), None))
