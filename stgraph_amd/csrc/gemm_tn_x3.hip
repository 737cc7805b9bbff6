// C[M,N] = sum_t A_t[K,M]^T [op(B_t) | B2_t][K,N] for tall-skinny fp32 operands -- the weight-gradient contractions of gemm_tn.hip --
// with every product taken as a 3-term bf16 split on v_mfma_f32_32x32x16_bf16 (bf16_split.hpp: x = h + m + l exactly to 2^-25 |x|,
// x y taken as hh + hm + mh + hl + lh + mm in fp32; what is dropped is below 2^-23 |x y|).
//
// Why: the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 64 flop / cycle / SIMD) bounds gemm_tn_wide: 1M x 128 x 128 takes 250 us
// against 160 us for reading the two operands once.  Six bf16 instructions of 32 cycles cover K = 16 of a 32 x 32 tile where the
// fp32 form needs eight of 64: 2.7 x less matrix time, and the splits (5.5 vector instructions per value) issue beside them.
//
// Layout: no LDS transpose.  The instruction wants, per lane (i = lane & 31, kgrp = lane >> 5), EIGHT CONSECUTIVE k of column i for
// both operands (A^T as its A, B as its B).  Lane (c, kgrp) loads rows k + 8 kgrp + t, t = 0 .. 7, 16 (8, 4) bytes each at columns
// W c .. W c + W - 1: eight loads that are 512 (256, 128) contiguous bytes per row over the 32 lanes of a group, and whose component r
// over t IS the fragment of "tile r" = columns {W c + r}: the M (and N) dimension is visited in a permuted order that costs
// nothing (gemm_tn_wide's trick).  Tile (ra, rb) accumulates C[WA i + ra][WB j + rb]; over rb a lane's values are W contiguous
// floats of an output row again.
//
// One workgroup = 4 waves = 4 adjacent K sub-slices of ONE slab (the whole M x N output: operands are read from HBM once), added
// in wave order through LDS; slabs are summed by gemm_tn's reduction launches.  One wave per SIMD (up to 256 accumulator + ~200
// operand registers), loads of the next 16 rows in flight under the products of the current ones.
#include "bf16_split.hpp"
#include "gemm_tn.hpp"

namespace stg {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int W>
__device__ __forceinline__ void x3_load(float (&dst)[W], __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    if constexpr (W == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        const unsigned u0 = v[0], u1 = v[1], u2 = v[2], u3 = v[3];       // (element-wise through scalars: see gemm_tn.hip load_vec)
        dst[0] = __uint_as_float(u0), dst[1] = __uint_as_float(u1), dst[2] = __uint_as_float(u2), dst[3] = __uint_as_float(u3);
    } else if constexpr (W == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        const unsigned u0 = v[0], u1 = v[1];
        dst[0] = __uint_as_float(u0), dst[1] = __uint_as_float(u1);
    } else if constexpr (W == 1) {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    }
}

// the fragment (eight k of one column) of each term from eight loaded rows, component r
template <int W>
__device__ __forceinline__ Frag3 frag_col(const float (&v)[8][W], int r)
{
    return frag_of(make_float4(v[0][r], v[1][r], v[2][r], v[3][r]), make_float4(v[4][r], v[5][r], v[6][r], v[7][r]));
}

__device__ __forceinline__ void mfma6_32(f32x16 &acc, const Frag3 &a, const Frag3 &b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[2], acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[2], b.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[1], b.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.t[0], b.t[0], acc, 0, 0, 0);
}

constexpr int kX3Step = 16;                 // rows per step (K of the instruction)

// WA: floats per lane of A (M = 32 WA); WB0 / WB1: of B's first / second matrix (N = 32 (WB0 + WB1); WB1 = 0: one matrix)
template <int WA, int WB0, int WB1, bool CS, int D = (WA * (WB0 + WB1) >= 16 ? 2 : (WA * (WB0 + WB1) >= 12 ? 3 : 4))>
__global__ __launch_bounds__(kBlock, 1) void gemm_tn_x3_kernel(const GemmSegs segs, const GemmForm form, float *__restrict__ slab,
                                                               int64_t K, int64_t kslice_wave, int s_per_seg, int n_groups)
{
    if (gemm_gated_off(form.gate, form.gate_when)) return;
    constexpr int TB = WB0 + WB1, M = 32 * WA, N = 32 * TB;     // N: the columns THIS workgroup covers (n_groups of them side by side)
    extern __shared__ float lds[];                       // one wave's accumulators: WA x TB x 16 x 64 floats (+ WA x 64)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, kgrp = lane >> 5;
    // n_groups == 2 (128 x 128: 16 tiles would need all 256 accumulator registers -- it spilled and ran at 261 us where its loads
    // alone take 185): two workgroups share a K slice, each taking 64 of B's columns.  Workgroups go to the 8 XCDs in turn, so the
    // two of a slice are made xcd, xcd + 8 of a run of 16: same XCD, resident together -- A's second read is an L2 hit.
    int s, nj = 0;                                       // slab = segment * s_per_seg + slice
    if (n_groups == 2) {
        const int b = blockIdx.x, run = b >> 4, within = b & 15, total = (int)gridDim.x >> 1, full = total & ~7;
        if (run * 8 < full) {
            s = run * 8 + (within & 7);
            nj = within >> 3;
        } else {
            const int r = b - 2 * full;
            s = full + (r >> 1);
            nj = r & 1;
        }
    } else {
        s = blockIdx.x;
    }
    const int seg = s / s_per_seg, sl = s - seg * s_per_seg;
    const float *__restrict__ A = segs.a[seg];
    const float *__restrict__ B = segs.b[seg] + nj * N;
    const float *__restrict__ B2 = segs.b2[seg];
    const int lda = form.lda, ldb = form.ldb, ldb2 = form.ldb2;
    const int Nt = N * n_groups;                         // row length of the slab
    const int64_t k0 = ((int64_t)sl * kWavesPerBlock + wave) * kslice_wave;     // wave-uniform, inside the segment
    const int64_t k1 = min(K, k0 + kslice_wave);
    const int64_t rows = k1 > k0 ? k1 - k0 : 0;
    // rows >= k1 read as zero through the descriptors' range check (gemm_tn.hip)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + k0 * lda), 0,
                                                       rows > 0 ? (int)(((rows - 1) * lda + M) * (int64_t)sizeof(float)) : 0, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B + k0 * ldb), 0,
                                                       rows > 0 ? (int)(((rows - 1) * ldb + 32 * WB0) * (int64_t)sizeof(float)) : 0, 0x00020000);
    const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(WB1 ? B2 + k0 * ldb2 : B), 0,
                                                        rows > 0 && WB1 ? (int)(((rows - 1) * ldb2 + 32 * WB1) * (int64_t)sizeof(float)) : 0,
                                                        0x00020000);
    const int voA = (8 * kgrp * lda + WA * c) * (int)sizeof(float);
    const int voB = (8 * kgrp * ldb + WB0 * c) * (int)sizeof(float);
    const int voB2 = (8 * kgrp * ldb2 + WB1 * c) * (int)sizeof(float);
    const int b_op = form.b_op;
    const float blo = form.lo, bhi = form.hi;

    f32x16 acc[WA][TB];
#pragma unroll
    for (int i = 0; i < WA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs[WA];
#pragma unroll
    for (int i = 0; i < WA; ++i) cs[i] = 0.f;

    struct Set {
        float a[8][WA], b[8][WB0], b2[8][WB1 ? WB1 : 1];
    };
    auto load_set = [&](Set &v, int r0) {                                       // r0: first row of the step inside the slice (uniform)
#if defined(STG_X3G_ABLATE) && STG_X3G_ABLATE == 2        // diagnosis build: the splits and products without the loads
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int r = 0; r < WA; ++r) v.a[t][r] = __int_as_float(r0 + t + r + lane);
#pragma unroll
            for (int r = 0; r < WB0; ++r) v.b[t][r] = __int_as_float(r0 - t + r + lane);
            if constexpr (WB1 > 0) {
#pragma unroll
                for (int r = 0; r < WB1; ++r) v.b2[t][r] = __int_as_float(r0 + 3 * t + r + lane);
            }
        }
        return;
#endif
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            x3_load<WA>(v.a[t], rsA, voA, (r0 + t) * lda * (int)sizeof(float));
            x3_load<WB0>(v.b[t], rsB, voB, (r0 + t) * ldb * (int)sizeof(float));
            if constexpr (WB1 > 0) x3_load<WB1>(v.b2[t], rsB2, voB2, (r0 + t) * ldb2 * (int)sizeof(float));
        }
    };
    auto consume = [&](Set &v) {
#if defined(STG_X3G_ABLATE) && STG_X3G_ABLATE == 1        // diagnosis build: the loads without the splits and products
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int r = 0; r < WA; ++r) keep += v.a[t][r];
#pragma unroll
            for (int r = 0; r < WB0; ++r) keep += v.b[t][r];
            if constexpr (WB1 > 0) {
#pragma unroll
                for (int r = 0; r < WB1; ++r) keep += v.b2[t][r];
            }
        }
        acc[0][0][0] += keep;
        return;
#endif
        if (b_op != STG_GEMM_B_NONE) {                                          // wave-uniform: transform of the first matrix' values
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < WB0; ++r) {
                    const float x = v.b[t][r];
                    v.b[t][r] = b_op == STG_GEMM_B_RELU ? (x < 0.f ? 0.f : x) : __builtin_amdgcn_fmed3f(x, blo, bhi);
                }
        }
        Frag3 fa[WA];
#pragma unroll
        for (int i = 0; i < WA; ++i) {
            fa[i] = frag_col<WA>(v.a, i);
            if constexpr (CS) {
#pragma unroll
                for (int t = 0; t < 8; ++t) cs[i] += v.a[t][i];
            }
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const Frag3 fb = j < WB0 ? frag_col<WB0>(v.b, j < WB0 ? j : 0) : frag_col<(WB1 ? WB1 : 1)>(v.b2, j >= WB0 ? j - WB0 : 0);
#pragma unroll
            for (int i = 0; i < WA; ++i) mfma6_32(acc[i][j], fa[i], fb);
        }
        // the splits BETWEEN the matrix instructions, not in front of them: a 32-cycle instruction hides ~5 vector instructions
        // issued behind it (MI355X_MICROARCH.md, cycle constants); left alone the scheduler splits a whole operand set first
#pragma unroll
        for (int g = 0; g < WA * TB * 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
    };

    // D operand sets in flight (D - 1 steps ahead of the products): one wave per SIMD keeps the memory system busy only through the
    // bytes it has outstanding -- at one step ahead (14 KB per wave, 14 MB over the chip) the launch ran at 4.5 TB/s
    const int nsteps = (int)((rows + kX3Step - 1) / kX3Step);
    Set sets[D];
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nsteps) load_set(sets[d], d * kX3Step);
    int i = 0;
    for (; i + 2 * D - 1 <= nsteps; i += D) {               // steady state: straight-line
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_set(sets[(d + D - 1) % D], (i + d + D - 1) * kX3Step);
            consume(sets[d]);
        }
    }
    for (; i < nsteps; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i + d < nsteps) {
                if (i + d + D - 1 < nsteps) load_set(sets[(d + D - 1) % D], (i + d + D - 1) * kX3Step);
                consume(sets[d]);
            }
        }
    }

    // the block's 4 K sub-slices, added in wave order through one wave-sized LDS buffer (as in gemm_tn.hip)
    constexpr int kAccFloats = WA * TB * 16 * kWave;
    for (int w = 1; w < kWavesPerBlock; ++w) {
        if (wave == w) {
            float *dst = lds + lane;
#pragma unroll
            for (int i = 0; i < WA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((i * TB + j) * 16 + r) * kWave] = acc[i][j][r];
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < WA; ++i) lds[kAccFloats + i * kWave + lane] = cs[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float *src = lds + lane;
#pragma unroll
            for (int i = 0; i < WA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * TB + j) * 16 + r) * kWave];
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < WA; ++i) cs[i] += lds[kAccFloats + i * kWave + lane];
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        // acc[ra][rb][r] = C[WA i + ra][chunk + WB j + rb], i = (r & 3) + 8 (r >> 2) + 4 kgrp (the 32 x 32 instruction's C map), j = c
        float *out = slab + (int64_t)s * ((int64_t)M * Nt + (CS ? M : 0)) + nj * N;
#pragma unroll
        for (int ra = 0; ra < WA; ++ra) {
            if constexpr (CS) {
                const float tot = cs[ra] + __shfl_xor(cs[ra], 32, kWave);       // the two row groups of column WA c + ra
                if (kgrp == 0 && nj == 0) out[(int64_t)M * Nt + WA * c + ra] = tot;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = WA * ((r & 3) + 8 * (r >> 2) + 4 * kgrp) + ra;
                float *o = out + (int64_t)m * Nt;
                if constexpr (WB0 == 4) *reinterpret_cast<float4 *>(o + 4 * c) = make_float4(acc[ra][0][r], acc[ra][1][r], acc[ra][2][r], acc[ra][3][r]);
                else if constexpr (WB0 == 2) *reinterpret_cast<float2 *>(o + 2 * c) = make_float2(acc[ra][0][r], acc[ra][1][r]);
                else o[c] = acc[ra][0][r];
                if constexpr (WB1 == 2) *reinterpret_cast<float2 *>(o + 32 * WB0 + 2 * c) = make_float2(acc[ra][WB0][r], acc[ra][WB0 + 1][r]);
                else if constexpr (WB1 == 1) o[32 * WB0 + c] = acc[ra][WB0][r];
            }
        }
    }
}

struct X3Plan {
    int S;                      // slices per segment
    int64_t kslice_wave;
};

X3Plan x3_plan(int64_t K, int T)
{
    // one workgroup per CU (one wave per SIMD): T * S as close to 256 as the segment count allows, never below 64 rows per wave
    int64_t S = std::max<int64_t>(1, 256 / T);
    S = std::max<int64_t>(1, std::min<int64_t>(S, K / (64 * kWavesPerBlock)));
    int64_t per_wave = (K + S * kWavesPerBlock - 1) / (S * kWavesPerBlock);
    per_wave = (per_wave + kX3Step - 1) / kX3Step * kX3Step;
    X3Plan p;
    p.kslice_wave = per_wave;
    p.S = (int)((K + per_wave * kWavesPerBlock - 1) / (per_wave * kWavesPerBlock));
    return p;
}

template <int WA, int WB0, int WB1>
int x3_launch_shape(const GemmSegs &segs, const GemmForm &form, float *slab, int64_t K, int T, bool colsum, const X3Plan &p,
                    hipStream_t stream, int n_groups = 1)
{
    constexpr int TB = WB0 + WB1;
    const size_t lds = ((size_t)WA * TB * 16 + WA) * kWave * sizeof(float);
    const unsigned blocks = (unsigned)(T * p.S * n_groups);
    auto go = [&](auto kern) {
        static PerDeviceOnce once;
        bool *raised = once.slot();
        if (lds > 64 * 1024 && !*raised) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail((int)e, "gemm_tn_x3: %s", hipGetErrorString(e));
            *raised = true;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), lds, stream, segs, form, slab, K, p.kslice_wave, p.S, n_groups);
        return check_launch("gemm_tn_x3");
    };
    return colsum ? go(gemm_tn_x3_kernel<WA, WB0, WB1, true>) : go(gemm_tn_x3_kernel<WA, WB0, WB1, false>);
}

// (WA, WB0, WB1) of a shape, or WA = 0
struct X3Shape {
    int wa, wb0, wb1, n_groups;
};
X3Shape x3_shape(int M, int N, int nsplit)
{
    X3Shape s{0, 0, 0, 1};
    if (M != 32 && M != 64 && M != 128) return s;
    // (the 128 x 128 spill is an ACCUMULATOR tile: 16 tiles are all 256 AccVGPRs and the allocator parks a few other values there.
    //  Loading A's and B's next rows at different times, and scheduling barriers around every B tile's products, left the
    //  allocation unchanged: 132-152 bytes of scratch either way)
    // (128 x 128 as two XCD-paired workgroups of 8 tiles per K slice -- n_groups = 2, no spills -- measured SLOWER than one
    //  workgroup of 16 tiles with its 130-byte spill: 279 against 232-261 us; A's second read through L2 costs more than the spill)
    if (nsplit == N && (N == 64 || N == 128)) s = {M / 32, N / 32, 0, 1};
    else if (nsplit == 64 && N == 96) s = {M / 32, 2, 1, 1};
    return s;
}

}  // namespace

bool gemm_tn_x3_covers(int M, int N, const GemmForm &form, int64_t K, int T)
{
    if (form.a_mask || x3_shape(M, N, form.nsplit).wa == 0) return false;
    if (form.lda % 4 || form.ldb % 4 || (form.nsplit < N && form.ldb2 % 4)) return false;
    const X3Plan p = x3_plan(K, T);
    // a wave addresses its K slice through 32-bit descriptors and scalar row offsets
    const int64_t max_ld = std::max<int64_t>(form.lda, std::max(form.ldb, form.ldb2));
    return p.kslice_wave * max_ld * 4 < (int64_t)INT32_MAX;
}

int gemm_tn_x3_slabs(int M, int N, int64_t K, int T)
{
    (void)M, (void)N;
    return T * x3_plan(K, T).S;
}

int gemm_tn_x3_launch(const GemmSegs &segs, const GemmForm &form, float *slab, int64_t K, int M, int N, int T, bool colsum, int *slabs,
                      hipStream_t stream)
{
    const X3Shape sh = x3_shape(M, N, form.nsplit);
    const X3Plan p = x3_plan(K, T);
    *slabs = T * p.S;
#define STG_X3(WA_, WB0_, WB1_) \
    if (sh.wa == WA_ && sh.wb0 == WB0_ && sh.wb1 == WB1_) return x3_launch_shape<WA_, WB0_, WB1_>(segs, form, slab, K, T, colsum, p, stream, sh.n_groups)
    STG_X3(4, 4, 0);
    STG_X3(4, 2, 0);
    STG_X3(4, 2, 1);
    STG_X3(2, 4, 0);
    STG_X3(2, 2, 0);
    STG_X3(2, 2, 1);
    STG_X3(1, 4, 0);
    STG_X3(1, 2, 0);
    STG_X3(1, 2, 1);
#undef STG_X3
    return fail(STG_ERR_UNSUPPORTED, "gemm_tn_x3: no instantiation for M=%d N=%d nsplit=%d", M, N, form.nsplit);
}

}  // namespace stg
