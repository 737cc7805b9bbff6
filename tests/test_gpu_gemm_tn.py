"""stg_gemm_tn_f32 (split-K fp32 MFMA weight gradient) against torch, incl. ragged shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,M,N", [(1, 1, 1), (7, 3, 5), (64, 32, 32), (1000, 33, 65), (4096, 64, 128),
                                   (50_000, 64, 128), (50_000, 32, 192), (50_000, 1, 32), (50_001, 128, 64),
                                   (200_000, 128, 128), (30_000, 200, 300), (2708, 1433, 16)])
def test_matches_torch(cuda, K, M, N):
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(K + M + N)
    a = torch.randn(K, M, device=cuda, generator=gen)
    b = torch.randn(K, N, device=cuda, generator=gen)
    got = kernels.gemm_tn(a, b)
    want = (a.double().t() @ b.double())
    err = (got.double() - want).abs().max().item()
    scale = (a.double().abs().t() @ b.double().abs()).max().item()
    assert err <= 2e-6 * scale + 1e-6, (err, scale)            # fp32 fma-chain accuracy (guide: ~1e-7 * sum|ab|)
    assert torch.equal(got, kernels.gemm_tn(a, b))              # deterministic: fixed slice order, no atomics
    got2, cs = kernels.gemm_tn(a, b, colsum=True)              # weight + bias gradient in one launch
    assert torch.equal(got2, got)
    csd = a.double().sum(0)
    assert (cs.double() - csd).abs().max().item() <= 2e-6 * a.double().abs().sum(0).max().item() + 1e-6


def test_asymmetric_integer_data_exact(cuda):
    """Catches any row/col or lane-map mix-up: exact small-integer data, asymmetric operands."""
    from stgraph_amd import kernels
    K, M, N = 4100, 70, 150
    a = (torch.arange(K * M, device=cuda) % 7 - 3).float().view(K, M)
    b = (torch.arange(K * N, device=cuda) % 5 - 2).float().view(K, N)
    assert torch.equal(kernels.gemm_tn(a, b), (a.double().t() @ b.double()).float())


def test_autograd_wrappers_match_torch(cuda):
    """mm / linear with the native weight gradient against an fp64 reference (the stock fp32 path is
    held to the same bar, so the comparison is about accuracy, not about agreeing with rocBLAS)."""
    from stgraph_amd.nn import functional as SF
    torch.manual_seed(0)
    x = torch.randn(20_000, 48, device=cuda, requires_grad=True)
    w = torch.randn(48, 96, device=cuda, requires_grad=True)
    lw = torch.randn(24, 96, device=cuda, requires_grad=True)
    lb = torch.randn(24, device=cuda, requires_grad=True)
    R = torch.randn(20_000, 24, device=cuda)
    leaves = (x, w, lw, lb)

    def run(native, dtype):
        SF.set_native_weight_grad(native)
        xs = [t.detach().to(dtype).requires_grad_(True) for t in leaves]
        (SF.linear(torch.relu(SF.mm(xs[0], xs[1])), xs[2], xs[3]) * R.to(dtype)).sum().backward()
        return [t.grad.double() for t in xs]
    try:
        ref = run(False, torch.float64)
        for native in (True, False):
            got = run(native, torch.float32)
            for g, r in zip(got, ref):
                assert (g - r).abs().max() <= 2e-5 * r.abs().max() + 1e-5, native
    finally:
        SF.set_native_weight_grad(True)


def test_slices_stay_inside_32bit_addressing(cuda):
    """A launch whose plan would give one wave K/4 rows of an 8192-float-wide operand (>= 2^31 bytes per slice:
    buffer descriptor and scalar row offsets are 32-bit) is cut into more slices instead (integer data: exact)."""
    from stgraph_amd import kernels
    K, M, N, T = 262_144 + 48, 32, 8192, 8              # 64 tiles x 8 segments = 512 blocks: one slice per segment
    a = (torch.arange(K * M, device=cuda) % 5 - 2).float().view(K, M)
    b = (torch.arange(K * N, device=cuda) % 3 - 1).float().view(K, N)
    want = torch.zeros(M, N, dtype=torch.float64, device=cuda)
    for r in range(0, K, 65_536):                        # fp64 in row chunks (b.double() at once would be 17 GB)
        want += a[r:r + 65_536].double().t() @ b[r:r + 65_536].double()
    got, cs = kernels.gemm_tn_multi([a] * T, [b] * T, colsum=True)
    assert torch.equal(got.double(), want * T)
    assert torch.equal(cs.double(), a.double().sum(0) * T)
    del a, b
    torch.cuda.empty_cache()


@pytest.mark.parametrize("K,T", [(50_000, 3), (4099, 2), (64, 1)])
def test_operand_forms_strides_split_and_transforms(cuda, K, T):
    """stg_gemm_tn_form_f32: A and B as column windows of wider matrices, B split over two matrices, clamp / relu applied
    to the first part while loading -- the forms the one-launch TGCN step's weight gradients use."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(K)
    C = 64
    x3 = [torch.randn(K, 3 * C, device=cuda, generator=gen) for _ in range(T)]
    H = [torch.randn(K, C, device=cuda, generator=gen) for _ in range(T)]
    d = [torch.randn(K, C, device=cuda, generator=gen) for _ in range(T)]
    da3 = [torch.randn(K, 3 * C, device=cuda, generator=gen) for _ in range(T)]
    lo, hi = -0.5, 0.8

    def ref(As, Bs):
        return sum(a.double().t() @ b.double() for a, b in zip(As, Bs))

    def close(got, want):
        assert (got.double() - want).abs().max() <= 3e-6 * want.abs().max() * max(1.0, K ** 0.5 / 50) + 1e-6
    for gate in range(3):                        # dW_gate = sum_t d_t^T [clamp(x3_t[:, gate]) | H_t]
        Bs = [x[:, gate * C:(gate + 1) * C] for x in x3]
        got, cs = kernels.gemm_tn_form(d, Bs, C, 2 * C, B2s=H, nsplit=C, b_op=kernels.GEMM_B_CLAMP, lo=lo, hi=hi, colsum=True)
        close(got, ref(d, [torch.cat([b.clamp(lo, hi), h], 1) for b, h in zip(Bs, H)]))
        close(cs, sum(a.double().sum(0) for a in d))
    # relu on a plain operand, A a column window of a wider matrix (lda > M), N = 32
    As = [x[:, 32:64] for x in da3]
    got = kernels.gemm_tn_form(As, H, 32, C, b_op=kernels.GEMM_B_RELU)
    close(got, ref(As, [h.relu() for h in H]))
    # no transform, single operand, M = 192, N = 32 (the conv weight gradient da3^T P) and M = 1 (dyo^T y)
    P = [h[:, :32].contiguous() for h in H]
    close(kernels.gemm_tn_form(da3, P, 3 * C, 32), ref(da3, P))
    one = [x[:, :1].contiguous() for x in d]
    close(kernels.gemm_tn_form(one, P, 1, 32), ref(one, P))
    assert torch.equal(kernels.gemm_tn_form(da3, P, 3 * C, 32), kernels.gemm_tn_multi(da3, P))      # same kernel, same order


@pytest.mark.parametrize("K,M,N", [(100_003, 128, 128), (50_000, 64, 32), (4097, 16, 7), (20_000, 96, 200)])
def test_relu_mask_on_load(cuda, K, M, N):
    """stg_gemm_tn_relu_mask_f32: (g * [out > 0])^T x and its column sums without materialising the masked gradient."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(K + M)
    g = torch.randn(K, M, device=cuda, generator=gen)
    out = torch.relu(torch.randn(K, M, device=cuda, generator=gen))
    x = torch.randn(K, N, device=cuda, generator=gen)
    c, cs = kernels.gemm_tn_relu_mask(g, out, x, colsum=True)
    gm = (g * (out > 0)).double()
    want, want_cs = gm.t() @ x.double(), gm.sum(0)
    scale = float(want.abs().max()) + 1
    torch.testing.assert_close(c.double(), want, rtol=1e-5, atol=1e-6 * scale)
    torch.testing.assert_close(cs.double(), want_cs, rtol=1e-5, atol=1e-6 * (float(want_cs.abs().max()) + 1))
    # identical to the two-launch form on the same kernel: the masked form runs on the dword-per-lane kernel, so the plain form is
    # pinned to it here (its 16-byte-per-lane kernel splits K differently and agrees to fp32 rounding, as checked above)
    from stgraph_amd import _C
    _C.set_tuning("gemm_wide", 1)
    _C.set_tuning("gemm_x3", 1)                       # (nor the bf16-split form, which large plain products take by default)
    try:
        two, two_cs = kernels.gemm_tn((g * (out > 0)), x, colsum=True)
    finally:
        _C.set_tuning("gemm_wide", 0)
        _C.set_tuning("gemm_x3", 0)
    assert torch.equal(c, two) and torch.equal(cs, two_cs)


def test_batched_reduction_equals_the_one_product_form(cuda):
    """kernels.gemm_tn_form_batch (stg_gemm_tn_form_partial_f32 per product + ONE stg_gemm_tn_reduce_multi_f32): a BPTT window's
    weight-gradient contractions -- different widths, operand transforms, with and without column sums -- bit for bit what
    gemm_tn_form returns for each alone."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(7)
    K, C, T = 20_001, 64, 3
    r = lambda *s: torch.randn(*s, device=cuda, generator=gen)  # noqa: E731
    d = [r(K, C) for _ in range(T)]
    x3 = [r(K, 3 * C) for _ in range(T)]
    H = [r(K, C) for _ in range(T)]
    da3 = [r(K, 3 * C) for _ in range(T)]
    P = [r(K, 32) for _ in range(T)]
    one = [r(K) for _ in range(T)]
    calls = [dict(As=d, Bs=[x[:, C:2 * C] for x in x3], M=C, N=2 * C, B2s=H, nsplit=C, b_op=kernels.GEMM_B_CLAMP, lo=-0.5, hi=0.7,
                  colsum=True),
             dict(As=da3, Bs=P, M=3 * C, N=32, colsum=True),
             dict(As=[p[:, :32].contiguous() for p in H], Bs=d, M=32, N=C, b_op=kernels.GEMM_B_RELU),
             dict(As=[o.view(K, 1) for o in one], Bs=P, M=1, N=32, colsum=True)]
    got = kernels.gemm_tn_form_batch(calls)
    for c, g in zip(calls, got):
        want = kernels.gemm_tn_form(**c)
        if c.get("colsum"):
            assert torch.equal(g[0], want[0]) and torch.equal(g[1], want[1])
        else:
            assert torch.equal(g, want)


@pytest.mark.parametrize("colsum", [True, False])
def test_batched_reduction_into_transposed_row_blocks(cuda, colsum):
    """``out_blocks_t`` / ``colsum_blocks`` of kernels.gemm_tn_form_batch (stg_gemm_tn_reduce_multi_blocks_f32): the stacked product
    of three layers' weight gradients leaves the reduction as three transposed [N, C] blocks and three bias slices -- the same
    floats as slicing and transposing the plain result; also through the one-by-one path (a single product)."""
    from stgraph_amd import kernels
    gen = torch.Generator(device=cuda).manual_seed(11)
    K, C, T, Fin = 20_001, 64, 3, 32
    r = lambda *s: torch.randn(*s, device=cuda, generator=gen)  # noqa: E731
    da3, P, d = [r(K, 3 * C) for _ in range(T)], [r(K, Fin) for _ in range(T)], [r(K, C) for _ in range(T)]
    want = kernels.gemm_tn_form(As=da3, Bs=P, M=3 * C, N=Fin, colsum=True)
    other = dict(As=d, Bs=P, M=C, N=Fin, colsum=True)
    for blocks_n in (3, 1, 4):
        rows = 3 * C // blocks_n
        for alone in (False, True):
            blocks = [torch.full((Fin, rows), float("nan"), device=cuda) for _ in range(blocks_n)]
            cs = [torch.full((rows,), float("nan"), device=cuda) for _ in range(blocks_n)] if colsum else None
            call = dict(As=da3, Bs=P, M=3 * C, N=Fin, colsum=colsum, out_blocks_t=blocks, colsum_blocks=cs)
            res = kernels.gemm_tn_form_batch([call] if alone else [other, call])
            assert res[-1] is None
            if not alone:
                w2 = kernels.gemm_tn_form(**other)
                assert torch.equal(res[0][0], w2[0]) and torch.equal(res[0][1], w2[1])
            for b in range(blocks_n):
                assert torch.equal(blocks[b], want[0][b * rows:(b + 1) * rows].t()), (blocks_n, alone, b)
                if colsum:
                    assert torch.equal(cs[b], want[1][b * rows:(b + 1) * rows]), (blocks_n, alone, b)
    with pytest.raises(ValueError):
        kernels.gemm_tn_form_batch([other, dict(As=da3, Bs=P, M=3 * C, N=Fin, colsum=False,
                                                out_blocks_t=[torch.empty(Fin, 3 * C // 5, device=cuda)] * 5)])


@pytest.mark.parametrize("M,N,nsplit,lda,b_op", [(128, 128, 128, 128, 0), (128, 96, 64, 192, 0), (64, 96, 64, 192, 0), (32, 64, 64, 32, 2),
                                                 (64, 128, 128, 64, 1), (128, 64, 64, 128, 0), (32, 96, 64, 32, 0)])
@pytest.mark.parametrize("K,T", [(70_001, 1), (9_000, 9), (50_000, 25)])
def test_split_form_against_fp64_and_the_fp32_form(cuda, M, N, nsplit, lda, b_op, K, T):
    """gemm_tn_x3.hip (knob "gemm_x3" 2): C = sum_t A_t^T [op(B_t) | B2_t] with every product a 3-term bf16 split -- against fp64 inside the
    fp32 form's own bound (both within 1 ulp of sum |a| |b|; the split form may not be more than twice the fp32 form's error + 0.25),
    column sums exact to fp32 summation, operands as column windows of wider matrices (lda > M), ragged K (the last slice's tail
    reads as zeros), clamp / ReLU applied to the first matrix while loading."""
    from stgraph_amd import _C, kernels
    gen = torch.Generator(device=cuda).manual_seed(M + N + K)
    wide = [torch.randn(K, lda, device=cuda, generator=gen) for _ in range(T)]
    As = [w[:, :M] for w in wide]
    Bs = [torch.randn(K, nsplit, device=cuda, generator=gen) for _ in range(T)]
    B2s = [torch.randn(K, N - nsplit, device=cuda, generator=gen) for _ in range(T)] if nsplit < N else None
    lo, hi = -0.5, 0.7
    op = {0: lambda t: t, 1: lambda t: t.clamp(lo, hi), 2: torch.relu}[b_op]
    Bfull = [torch.cat([op(b), b2], 1) if B2s else op(b) for b, b2 in zip(Bs, B2s or Bs)]
    want = sum(a.double().t() @ b.double() for a, b in zip(As, Bfull))
    scale = sum(a.double().abs().t() @ b.double().abs() for a, b in zip(As, Bfull)) * 2.0 ** -24
    want_cs = sum(a.double().sum(0) for a in As)
    err = {}
    for name, knob in (("f32", 1), ("x3", 2)):
        _C.set_tuning("gemm_x3", knob)
        try:
            c, cs = kernels.gemm_tn_form(As, Bs, M, N, B2s=B2s, nsplit=nsplit, b_op=b_op, lo=lo, hi=hi, colsum=True)
            c2 = kernels.gemm_tn_form(As, Bs, M, N, B2s=B2s, nsplit=nsplit, b_op=b_op, lo=lo, hi=hi)
        finally:
            _C.set_tuning("gemm_x3", 0)
        assert torch.equal(c, c2)                                      # with and without the column sums: the same products
        err[name] = float(((c.double() - want).abs() / scale).max())
        assert float((cs.double() - want_cs).abs().max()) <= 1e-5 * float(want_cs.abs().max()) + 1e-3
    assert err["f32"] <= 1.0 and err["x3"] <= 1.0, err
    assert err["x3"] <= 2 * err["f32"] + 0.25, err


def test_split_form_is_exact_on_small_integers(cuda):
    """Integers up to 255 are one bf16 term each: every product and every fp32 partial sum is exact in both forms."""
    from stgraph_amd import _C, kernels
    gen = torch.Generator(device=cuda).manual_seed(3)
    A = torch.randint(-200, 200, (80_000, 128), device=cuda, generator=gen).float()
    B = torch.randint(-3, 4, (80_000, 128), device=cuda, generator=gen).float()
    _C.set_tuning("gemm_x3", 2)
    try:
        c = kernels.gemm_tn(A, B)
    finally:
        _C.set_tuning("gemm_x3", 0)
    assert torch.equal(c.double(), A.double().t() @ B.double())
