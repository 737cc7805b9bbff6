// Y[N,M] = act(X[N,K] * op(W) + bias) for many rows and K, M in {64, 128} on the bf16 MATRIX cores of gfx950, every product
// as a 3-term bf16 split with fp32 accumulation (the arithmetic of bf16_split.hpp: x = h + m + l exactly, x w taken as
// h h + h m + m h + h l + l h + m m; what is dropped is below 2^-23 |x w|, one fp32 rounding).
//
// Why: v_mfma_f32_16x16x4_f32 is 157 TFLOP/s and runs on the vector lanes (profiles/r04_coexec_f32mfma.jsonl): cfg2's three
// [1M, 128] x [128, 128] products per step are 209 us of it each, and the fp32 row-piece kernel (rowgemm.hip) takes 270.  Six
// v_mfma_f32_16x16x32_bf16 (16 cycles, K = 32, a pipe of their own) replace eight fp32 ones (32 cycles, K = 4): 78 us of
// matrix time, so the product is bound by its 1 GB of HBM traffic instead.  Here -- unlike the TGCN step, where the split of
// the activations is repeated by every wave that needs them -- a row's K values are split ONCE and
// used against all M output columns: 5.5 vector instructions per value next to 6 M / 16 matrix instructions per 8 values.
//
// Layout: weights are the A operand (row = output column inside a 16-column tile), rows of X the B operand (column = row
// n16 of a 16-row tile), so lane (n16 = lane & 15, kq = lane >> 4) receives output columns 16 ct + 4 kq .. + 3 of ITS row:
// 16-byte stores (the row-piece scheme of tgcn_step.hpp).  The lane's eight k values of K-block b are the CONTIGUOUS
// columns 32 b + 8 kq .. + 7 of its row (two adjacent 16-byte loads; the four kq groups cover 128 contiguous bytes; measured:
// the sector-aligned alternative -- xcol of bf16_split.hpp, 64 contiguous bytes of a row per instruction -- is 15 % SLOWER).  The
// weights are split once per workgroup into an LDS image [ct][b][term][lane] x 16 bytes (96 KB at 128 x 128) and read back
// as three ds_read_b128 per (ct, b); a wave works on TWO 16-row tiles at a time so that each weight fragment read feeds
// twelve matrix instructions (LDS: 62 of 128 bytes per clock).  One workgroup of 8 waves per CU, tile pairs dealt wave-major,
// the rows two K-blocks ahead are in flight under the current block's products.
#include <algorithm>

#include "bf16_split.hpp"

namespace stg {
namespace {

constexpr int kX3Waves = 8;

__device__ __forceinline__ void mfma6x4(f32x4 &a0, f32x4 &a1, f32x4 &b0, f32x4 &b1, const Frag3 &wa, const Frag3 &wb, const Frag3 &x0,
                                        const Frag3 &x1)
{
#define STG_X3_ROUND(TW, TX)                                                              \
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.t[TW], x0.t[TX], a0, 0, 0, 0);        \
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.t[TW], x1.t[TX], a1, 0, 0, 0);        \
    b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb.t[TW], x0.t[TX], b0, 0, 0, 0);        \
    b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb.t[TW], x1.t[TX], b1, 0, 0, 0);
    STG_X3_ROUND(0, 2)                                  // small terms first
    STG_X3_ROUND(2, 0)
    STG_X3_ROUND(1, 1)
    STG_X3_ROUND(0, 1)
    STG_X3_ROUND(1, 0)
    STG_X3_ROUND(0, 0)
#undef STG_X3_ROUND
}

// BITS (whole-line form only): the sign pattern of a ReLU layer's output as one bit per element, in the lanes' own arrangement
// of the output -- element (row, col) is bit 8 (col >> 5) + 4 ((row >> 3) & 1) + (col & 3) of word
// 64 (row >> 4) + (row & 7) + 8 ((col >> 4) & 1) + 16 ((col >> 2) & 3): a lane's 32 outputs of a 16-row tile are ONE word, written
// and read back with no exchange between lanes.  1: leave [y > 0] in `bits` (with RELU); 2: multiply the product by the pattern in
// `bits` before it is stored -- the ReLU backward of the layer below, dX = (g W^T) * [out > 0], in the launch that forms g W^T.
template <int K, int M, bool TRANS_W, bool RELU, bool LINES, int BITS = 0>
__global__ __launch_bounds__(kX3Waves * kWave, 1) void rowgemm_x3_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                                        const float *__restrict__ bias, float *__restrict__ Y,
                                                                        int64_t N, int num_pairs, int ldy, uint32_t *__restrict__ bits,
                                                                        int ldx)
{
    static_assert(BITS == 0 || LINES, "the bit pattern follows the whole-line arrangement");
    constexpr int KB = K / 32, CT = M / 16, AHEAD = 2;
    static_assert(KB >= AHEAD && KB % AHEAD == 0, "the prefetch ring is two K-blocks deep");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *img = lds;                                                   // [ct][b][t][lane] x 16 bytes
    float *bs = reinterpret_cast<float *>(lds + CT * KB * kXTerms * kFragBytes);     // [M] (zeros without a bias)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int total = gridDim.x * kX3Waves;
    int pair = wave * (int)gridDim.x + (int)blockIdx.x;

    // rows of tile `t` (0 / 1) of a pair; lanes past the last row mirror row N - 1 (they rewrite its values), and a wave
    // without a next pair re-reads the last one: no load or store of the loop sits behind a branch, so the waits the
    // compiler places count instructions instead of draining the queue
#ifndef STG_X3_ABLATE
#define STG_X3_ABLATE 0                                               // diagnosis builds only (tools/diag/build_x3_ablate.sh): 1 no stores, 2 loads from
#endif                                                                // one cached tile, 4 no matrix instructions
    auto row_of = [&](int p, int t) { return std::min<int64_t>((int64_t)p * 32 + 16 * t + n16, N - 1); };
    auto row_ld = [&](int p, int t) { return (STG_X3_ABLATE & 2) ? (int64_t)(16 * t + n16) : row_of(p, t); };
    // X through a buffer descriptor (the launcher checks N K < 2^30): a plain global load is free to sink to its first use
    // -- the compiler put every one of them right in front of the split that consumes it, a full memory round trip per
    // K-block -- while the buffer-load intrinsic keeps its place between the scheduling fences below.
    // (ldx >= K: rows of X inside a wider matrix -- a head's columns of [N, H D])
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (int)(((N - 1) * ldx + K) * (int64_t)sizeof(float)),
                                                       0x00020000);
    auto load16 = [&](int64_t row, int col) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)((row * ldx + col) * (int64_t)sizeof(float)), 0, 0);
        const unsigned u0 = v[0], u1 = v[1], u2 = v[2], u3 = v[3];     // (element by element: see load_vec in gemm_tn.hip)
        return make_float4(__uint_as_float(u0), __uint_as_float(u1), __uint_as_float(u2), __uint_as_float(u3));
    };
    // A 128-byte line of a row holds the 32 columns of one K-block.  LINES: one instruction fetches it WHOLE -- lanes
    // (r, kq) and (r + 8, kq) take the two 64-byte halves of row r, a second instruction likewise rows 8 .. 15, and the lanes
    // trade the pieces that belong to their partner's row (lane ^ 8: one DPP row rotation) -- instead of every lane taking
    // two pieces of ITS row, which asks for each line twice, half a line (or four 16-byte slivers) at a time.
    const int r8 = n16 & 7, upper = n16 >> 3;
    auto trade = [&](const float4 &own_if_lower, const float4 &own_if_upper, float4 &lo, float4 &hi) {
        // lower lane (r): lo = its first load, hi = the upper lane's first load; upper lane (r + 8): lo = its second, hi = the lower's second
        auto one = [&](float a1, float a2, float &l, float &h) {
            const int i1 = __float_as_int(a1), i2 = __float_as_int(a2);
            l = __int_as_float(__builtin_amdgcn_update_dpp(i1, i2, 0xE4, 0xF, 0xC, false));              // lanes 8-15: a2
            const int t = __builtin_amdgcn_update_dpp(i1, i1, 0x128, 0xF, 0x3, false);                  // lanes 0-7: partner's a1
            h = __int_as_float(__builtin_amdgcn_update_dpp(t, i2, 0x128, 0xF, 0xC, false));             // lanes 8-15: partner's a2
        };
        one(own_if_lower.x, own_if_upper.x, lo.x, hi.x);
        one(own_if_lower.y, own_if_upper.y, lo.y, hi.y);
        one(own_if_lower.z, own_if_upper.z, lo.z, hi.z);
        one(own_if_lower.w, own_if_upper.w, lo.w, hi.w);
    };
    float4 ring[AHEAD][2][2];                                         // [slot][tile][load]
    auto load_block = [&](int slot, int p, int b) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (LINES) {
                const int64_t base = (int64_t)p * 32 + 16 * t;
                const int64_t ra = (STG_X3_ABLATE & 2) ? r8 : std::min<int64_t>(base + r8, N - 1);
                const int64_t rb = (STG_X3_ABLATE & 2) ? 8 + r8 : std::min<int64_t>(base + 8 + r8, N - 1);
                ring[slot][t][0] = load16(ra, 32 * b + 16 * upper + 4 * kq);
                ring[slot][t][1] = load16(rb, 32 * b + 16 * (1 - upper) + 4 * kq);
            } else {
                ring[slot][t][0] = load16(row_ld(p, t), 32 * b + 8 * kq);
                ring[slot][t][1] = load16(row_ld(p, t), 32 * b + 8 * kq + 4);
            }
        }
    };
    auto frag_block = [&](int slot, int t) {
        if constexpr (LINES) {
            float4 lo, hi;
            trade(ring[slot][t][0], ring[slot][t][1], lo, hi);
            return frag_of(lo, hi);
        } else {
            return frag_of(ring[slot][t][0], ring[slot][t][1]);
        }
    };
    if (pair < num_pairs) {
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) load_block(s, pair, s);
    }

    // the weight image: A[row = output column m][k] = op(W)[k][m], k of (b, kq, i) = 32 b + 8 kq + i
    static_assert((CT * KB) % kX3Waves == 0, "fragments are dealt evenly over the waves");
#pragma unroll
    for (int fi = 0; fi < CT * KB / kX3Waves; ++fi) {                  // (unrolled: one round trip for the lot, not one per fragment)
        const int f = fi * kX3Waves + wave;
        // k of (b, kq, i): LINES xcol(b, kq, i) = 32 b + 16 (i >> 2) + 4 kq + (i & 3), else 32 b + 8 kq + i
        const int ct = f / KB, b = f - ct * KB, m = 16 * ct + n16, k0 = 32 * b + (LINES ? 4 : 8) * kq, kh = LINES ? 16 : 4;
        float4 lo, hi;
        if constexpr (TRANS_W) {                                       // W [M][K]
            lo = *reinterpret_cast<const float4 *>(W + (int64_t)m * K + k0);
            hi = *reinterpret_cast<const float4 *>(W + (int64_t)m * K + k0 + kh);
        } else {                                                       // W [K][M]
            const float *w = W + (int64_t)k0 * M + m;
            lo = make_float4(w[0], w[M], w[2 * M], w[3 * M]);
            hi = make_float4(w[kh * M], w[(kh + 1) * M], w[(kh + 2) * M], w[(kh + 3) * M]);
        }
        const Frag3 fr = frag_of(lo, hi);
#pragma unroll
        for (int t = 0; t < kXTerms; ++t)
            *reinterpret_cast<uint4 *>(img + ((size_t)(f * kXTerms + t) * kWave + lane) * 16) = __builtin_bit_cast(uint4, fr.t[t]);
    }
    for (int i = threadIdx.x; i < M; i += kX3Waves * kWave) bs[i] = bias ? bias[i] : 0.f;
    __syncthreads();

    const auto rsBits = __builtin_amdgcn_make_buffer_rsrc(BITS ? bits : reinterpret_cast<uint32_t *>(Y), 0,
                                                          BITS ? (int)((int64_t)num_pairs * 2 * kWave * sizeof(uint32_t)) : 0, 0x00020000);
    for (; pair < num_pairs; pair += total) {
        const int next = std::min(pair + total, num_pairs - 1);
        unsigned pattern[2] = {0u, 0u};
        if constexpr (BITS == 2) {                                     // (ahead of the ring's loads: long since here at the stores)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                pattern[t] = __builtin_amdgcn_raw_buffer_load_b32(rsBits, ((pair * 2 + t) * kWave + lane) * (int)sizeof(uint32_t), 0, 0);
        }
        f32x4 acc[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 b4 = to_x4(*reinterpret_cast<const float4 *>(bs + 16 * ct + 4 * kq));
            acc[0][ct] = b4;
            acc[1][ct] = b4;
        }
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const int slot = b % AHEAD;
            const Frag3 x0 = frag_block(slot, 0), x1 = frag_block(slot, 1);
            // refill the slot: this pair's block b + AHEAD, or the next pair's block b + AHEAD - KB
            if (b + AHEAD < KB) load_block(slot, pair, b + AHEAD);
            else load_block(slot, next, b + AHEAD - KB);
            // two column tiles at a time: four accumulators in rotation (an instruction's accumulator was written four
            // instructions earlier), the next two tiles' weight fragments on their way from LDS meanwhile.  (Left to itself
            // the scheduler hoists every fragment of the block above the first product: 96 registers, ~390 spilled.)
            Frag3 wa = wfrag_load(img, (0 * KB + b) * kXTerms, lane), wb = wfrag_load(img, (1 * KB + b) * kXTerms, lane);
#pragma unroll
            for (int ct = 0; ct < CT; ct += 2) {
                Frag3 na = wa, nb = wb;
                if (ct + 2 < CT) {
                    na = wfrag_load(img, ((ct + 2) * KB + b) * kXTerms, lane);
                    nb = wfrag_load(img, ((ct + 3) * KB + b) * kXTerms, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STG_X3_ABLATE & 4) {
                    acc[0][ct][0] += __builtin_bit_cast(float4, x0.t[0]).x + __builtin_bit_cast(float4, wa.t[0]).x;
                    acc[1][ct][0] += __builtin_bit_cast(float4, x1.t[2]).x + __builtin_bit_cast(float4, wb.t[2]).x;
                } else {
                    mfma6x4(acc[0][ct], acc[1][ct], acc[0][ct + 1], acc[1][ct + 1], wa, wb, x0, x1);
                }
                __builtin_amdgcn_sched_barrier(0);
                wa = na, wb = nb;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            auto act = [&](const f32x4 &a) {
                float4 o = to_f4(a);
                if constexpr (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                return o;
            };
            if constexpr (LINES) {
                // whole lines again: columns 32 c .. 32 c + 31 of a row are this lane's tiles 2 c and 2 c + 1; the first store
                // writes rows 0 .. 7 (the upper lane stores its partner's tile 2 c + 1), the second rows 8 .. 15
                const int64_t base = (int64_t)pair * 32 + 16 * t;
                float *d1 = Y + std::min<int64_t>(base + r8, N - 1) * ldy + 16 * upper + 4 * kq;
                float *d2 = Y + std::min<int64_t>(base + 8 + r8, N - 1) * ldy + 16 * upper + 4 * kq;
#pragma unroll
                for (int c = 0; c < CT / 2; ++c) {
                    const float4 a = act(acc[t][2 * c]), bq = act(acc[t][2 * c + 1]);
                    float4 s1, s2;
                    auto one = [&](float av, float bv, float &o1, float &o2) {
                        const int ia = __float_as_int(av), ib = __float_as_int(bv);
                        o1 = __int_as_float(__builtin_amdgcn_update_dpp(ia, ib, 0x128, 0xF, 0xC, false));   // upper: partner's 2 c + 1
                        o2 = __int_as_float(__builtin_amdgcn_update_dpp(ib, ia, 0x128, 0xF, 0x3, false));   // lower: partner's 2 c
                    };
                    one(a.x, bq.x, s1.x, s2.x);
                    one(a.y, bq.y, s1.y, s2.y);
                    one(a.z, bq.z, s1.z, s2.z);
                    one(a.w, bq.w, s1.w, s2.w);
                    if constexpr (BITS == 1) {
                        pattern[t] |= ((s1.x > 0.f ? 1u : 0u) | (s1.y > 0.f ? 2u : 0u) | (s1.z > 0.f ? 4u : 0u) | (s1.w > 0.f ? 8u : 0u) |
                                       (s2.x > 0.f ? 16u : 0u) | (s2.y > 0.f ? 32u : 0u) | (s2.z > 0.f ? 64u : 0u) | (s2.w > 0.f ? 128u : 0u))
                                      << (8 * c);
                    } else if constexpr (BITS == 2) {
                        const unsigned pb = pattern[t] >> (8 * c);
                        s1 = make_float4(pb & 1u ? s1.x : 0.f, pb & 2u ? s1.y : 0.f, pb & 4u ? s1.z : 0.f, pb & 8u ? s1.w : 0.f);
                        s2 = make_float4(pb & 16u ? s2.x : 0.f, pb & 32u ? s2.y : 0.f, pb & 64u ? s2.z : 0.f, pb & 128u ? s2.w : 0.f);
                    }
                    if ((STG_X3_ABLATE & 1) && s1.x != 1.2345e33f) continue;
                    *reinterpret_cast<float4 *>(d1 + 32 * c) = s1;
                    *reinterpret_cast<float4 *>(d2 + 32 * c) = s2;
                }
                if constexpr (BITS == 1) bits[((int64_t)pair * 2 + t) * kWave + lane] = pattern[t];
            } else {
                float *dst = Y + row_of(pair, t) * ldy + 4 * kq;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float4 o = act(acc[t][ct]);
                    if ((STG_X3_ABLATE & 1) && o.x != 1.2345e33f) continue;
                    *reinterpret_cast<float4 *>(dst + 16 * ct) = o;
                }
            }
        }
    }
}

template <int K, int M, bool TW, bool RELU, bool LINES, int BITS = 0>
int rowgemm_x3_launch3(const float *X, const float *W, const float *bias, float *Y, int64_t N, hipStream_t st, int ldy, uint32_t *bits = nullptr,
                       int ldx = K)
{
    constexpr size_t lds = (size_t)(M / 16) * (K / 32) * kXTerms * kFragBytes + sizeof(float) * M;
    const int64_t pairs = (N + 31) / 32;
    if (N * ldx >= (int64_t)1 << 30) return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: the split form addresses X with 32 bits (N ldx < 2^30)");
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (lds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_x3_kernel<K, M, TW, RELU, LINES, BITS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "stg_rowgemm_f32: %s", hipGetErrorString(e));
        *raised = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const unsigned blocks = (unsigned)std::min<int64_t>((pairs + kX3Waves - 1) / kX3Waves, cus);
    hipLaunchKernelGGL((rowgemm_x3_kernel<K, M, TW, RELU, LINES, BITS>), dim3(blocks), dim3(kX3Waves * kWave), lds, st, X, W, bias, Y, N,
                       (int)pairs, ldy, bits, ldx);
    return check_launch("stg_rowgemm_f32");
}

template <int K, int M, bool TW, bool RELU>
int rowgemm_x3_launch2(const float *X, const float *W, const float *bias, float *Y, int64_t N, hipStream_t st, int ldy)
{
    // tuning "rowgemm_x3" = 3 (diagnostic): every lane loads and stores pieces of its own row (half lines per instruction)
    return tuning().rowgemm_x3 == 3 ? rowgemm_x3_launch3<K, M, TW, RELU, false>(X, W, bias, Y, N, st, ldy)
                                    : rowgemm_x3_launch3<K, M, TW, RELU, true>(X, W, bias, Y, N, st, ldy);
}

}  // namespace

// (declared in stg_common.hpp; rowgemm.hip dispatches here)
int rowgemm_x3_bits_launch(int K, int M, const float *X, const float *W, const float *bias, float *Y, int64_t N, const uint32_t *bits_in,
                           uint32_t *bits_out, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the two uses: the ReLU layer forward (W [K][M], pattern out) and the input gradient of the layer above it (W [M][K] read in
    // place, pattern in)
#define STG_X3(K_, M_)                                                                                                          \
    if (K == K_ && M == M_)                                                                                                     \
        return bits_out ? rowgemm_x3_launch3<K_, M_, false, true, true, 1>(X, W, bias, Y, N, st, M, bits_out)                    \
                        : rowgemm_x3_launch3<K_, M_, true, false, true, 2>(X, W, bias, Y, N, st, M, const_cast<uint32_t *>(bits_in));
    STG_X3(128, 128)
    STG_X3(64, 128)
    STG_X3(128, 64)
    STG_X3(64, 64)
#undef STG_X3
    return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_act_bits_f32: the split form covers K, M in {64, 128} (got %d, %d)", K, M);
}

// Y_h [N, M] = X[:, h K : (h + 1) K] W_h for the `heads` column blocks of X [N, heads K], W [heads][K][M], Y [heads][N][M]: one
// launch per block (the weight image of one is the workgroup's LDS)
int rowgemm_x3_heads_launch(int K, int M, const float *X, const float *W, float *Y, int64_t N, int heads, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int h = 0; h < heads; ++h) {
        const float *x = X + (int64_t)h * K, *w = W + (int64_t)h * K * M;
        float *y = Y + (int64_t)h * N * M;
        int rc;
        if (K == 64 && M == 64) rc = rowgemm_x3_launch3<64, 64, false, false, true>(x, w, nullptr, y, N, st, M, nullptr, heads * K);
        else if (K == 64 && M == 128) rc = rowgemm_x3_launch3<64, 128, false, false, true>(x, w, nullptr, y, N, st, M, nullptr, heads * K);
        else if (K == 128 && M == 64) rc = rowgemm_x3_launch3<128, 64, false, false, true>(x, w, nullptr, y, N, st, M, nullptr, heads * K);
        else if (K == 128 && M == 128) rc = rowgemm_x3_launch3<128, 128, false, false, true>(x, w, nullptr, y, N, st, M, nullptr, heads * K);
        else return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_heads_f32: K, M in {64, 128} (got %d, %d)", K, M);
        if (rc != 0) return rc;
    }
    return 0;
}

int rowgemm_x3_launch(int K, int M, const float *X, const float *W, const float *bias, float *Y, int64_t N, bool tw, bool relu,
                      void *stream, int ldy)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
#define STG_X3(K_, M_)                                                                                                          \
    if (K == K_ && M == M_) {                                                                                                   \
        if (tw) return relu ? rowgemm_x3_launch2<K_, M_, true, true>(X, W, bias, Y, N, st, ldy)                                 \
                            : rowgemm_x3_launch2<K_, M_, true, false>(X, W, bias, Y, N, st, ldy);                               \
        return relu ? rowgemm_x3_launch2<K_, M_, false, true>(X, W, bias, Y, N, st, ldy)                                        \
                    : rowgemm_x3_launch2<K_, M_, false, false>(X, W, bias, Y, N, st, ldy);                                      \
    }
    STG_X3(128, 128)
    STG_X3(64, 128)
    STG_X3(128, 64)
    STG_X3(64, 64)
#undef STG_X3
    return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: the split form covers K, M in {64, 128} (got %d, %d)", K, M);
}

}  // namespace stg
