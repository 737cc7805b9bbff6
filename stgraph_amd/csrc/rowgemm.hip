// Y[N,M] = X[N,K] * op(W) + bias   with N = number of vertices (large) and K, M = feature widths
// (<= 192): the forward / input-gradient GEMMs of the dense layers next to the Seastar kernels
// (TGCN gate Linears: K = 128 -> M = 64 and back).  rocBLAS/hipBLASLt choose 64x32- or 128x224-wide
// macro tiles for these skinny shapes and land at 2-4x the memory-bound time (measured in situ: 29.7 us
// for [50K,128] x [128,64]^T, whose 38 MB of traffic take 7 us; profiles/r01).
#include <algorithm>

#include "tgcn_step.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---- the same product in the step kernels' row-piece layout (tgcn_step.hpp) for many rows x (K, M <= 128, multiples of 16) ----
// cfg2's three dense products per step are [1M, 128] x [128, 128]: 32.8 GFLOP = 209 us of fp32 MFMA, 1 GB = 170 us of HBM;
// hipBLASLt's 128x128x16 macro tile takes 360 us and the 32x32x2 kernel below 443.  Here a wave owns 16 rows: lane (n16, kq)
// holds the row pieces X[row n16][16 j + 4 kq ..] (K / 16 float4s, loaded one tile ahead), the weight is the A operand of
// v_mfma_f32_16x16x4_f32 read from LDS in its [out][in] layout (one ds_read_b128 per four k steps, rows padded by 8 floats:
// conflict-free), so a lane's four accumulator values are four consecutive columns of its own row: 16-byte stores, no
// transposes.  The output columns are taken in two halves (16 accumulator registers live instead of 32).  One workgroup of
// 12 waves per CU, tiles dealt wave-major.
template <int K, int M, bool TRANS_W, bool RELU>
__global__ __launch_bounds__(12 * kWave) void rowgemm16_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                              const float *__restrict__ bias, float *__restrict__ Y, int64_t N,
                                                              int num_tiles, int ldy)
{
    constexpr int WAVES = 12, NT = WAVES * kWave, PK = K / 16, PM = M / 16, LD = K + 8;
    constexpr int HALF = PM >= 2 && PM % 2 == 0 ? PM / 2 : PM, NH = PM / HALF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds;                                   // [M][LD]: W as [out][in]
    float *bs = Ws + M * LD;                           // [M] (zeros without a bias)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int total = gridDim.x * WAVES;
    int tile = wave * (int)gridDim.x + (int)blockIdx.x;

    auto row_off = [&](int t) {                        // lanes past the last row mirror row N - 1 (they rewrite its values)
        const int64_t r = std::min<int64_t>((int64_t)t * 16 + n16, N - 1);
        return r;
    };
    float4 xn[PK];
    auto load_x = [&](int t) {
        const float *src = X + row_off(t) * K + 4 * kq;
#pragma unroll
        for (int j = 0; j < PK; ++j) xn[j] = *reinterpret_cast<const float4 *>(src + 16 * j);
    };
    if (tile < num_tiles) load_x(tile);

    if constexpr (TRANS_W) {                           // W is [M][K] already: rows copied with the padded stride
        for (int i = threadIdx.x; i < M * K / 4; i += NT) {
            const int m = (i * 4) / K, k = (i * 4) - m * K;
            *reinterpret_cast<float4 *>(Ws + m * LD + k) = *reinterpret_cast<const float4 *>(W + (int64_t)i * 4);
        }
    } else {                                           // W is [K][M]: transposed on the way (coalesced reads, strided LDS writes, once)
        for (int i = threadIdx.x; i < K * M / 4; i += NT) {
            const int k = (i * 4) / M, m = (i * 4) - k * M;
            const float4 v = *reinterpret_cast<const float4 *>(W + (int64_t)i * 4);
            Ws[(m + 0) * LD + k] = v.x;
            Ws[(m + 1) * LD + k] = v.y;
            Ws[(m + 2) * LD + k] = v.z;
            Ws[(m + 3) * LD + k] = v.w;
        }
    }
    for (int i = threadIdx.x; i < M; i += NT) bs[i] = bias ? bias[i] : 0.f;
    __syncthreads();

    const float *const w_l = Ws + pinned((unsigned)(n16 * LD + 4 * kq)), *const b_l = bs + pinned((unsigned)(4 * kq));
    float4 wq[HALF];
    load_w<HALF>(wq, w_l, LD, 0);
    for (; tile < num_tiles; tile += total) {
        float4 x[PK];
#pragma unroll
        for (int j = 0; j < PK; ++j) x[j] = xn[j];
        const int64_t r = row_off(tile);
        if (tile + total < num_tiles) load_x(tile + total);          // the next tile's rows: in flight under this tile's products
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 acc[HALF];
#pragma unroll
            for (int ct = 0; ct < HALF; ++ct) acc[ct] = to_x4(*reinterpret_cast<const float4 *>(b_l + 16 * (h * HALF + ct)));
            float4 w0[HALF];
#pragma unroll
            for (int ct = 0; ct < HALF; ++ct) w0[ct] = wq[ct];
            gemm_chain<HALF, PK>(acc, w_l + h * HALF * 16 * LD, LD, [&](int j) { return x[j]; }, w0,
                                 [&]() { load_w<HALF>(wq, w_l + ((h + 1) % NH) * HALF * 16 * LD, LD, 0); });
            float *dst = Y + r * ldy + 16 * h * HALF + 4 * kq;
#pragma unroll
            for (int ct = 0; ct < HALF; ++ct) {
                float4 o = to_f4(acc[ct]);
                if constexpr (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                *reinterpret_cast<float4 *>(dst + 16 * ct) = o;
            }
        }
    }
}

template <int K, int M, bool TW, bool RELU>
int rowgemm16_launch2(const float *X, const float *W, const float *bias, float *Y, int64_t N, hipStream_t st, int ldy)
{
    constexpr size_t lds = sizeof(float) * ((size_t)M * (K + 8) + M);
    const int64_t tiles = (N + 15) / 16;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: too many rows");
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (lds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm16_kernel<K, M, TW, RELU>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "stg_rowgemm_f32: %s", hipGetErrorString(e));
        *raised = true;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + 11) / 12, 256);
    hipLaunchKernelGGL((rowgemm16_kernel<K, M, TW, RELU>), dim3(blocks), dim3(12 * kWave), lds, st, X, W, bias, Y, N, (int)tiles, ldy);
    return check_launch("stg_rowgemm_f32");
}

template <int K, int M>
int rowgemm16_launch(const float *X, const float *W, const float *bias, float *Y, int64_t N, bool trans_w, bool relu, hipStream_t st, int ldy)
{
    if (trans_w) return relu ? rowgemm16_launch2<K, M, true, true>(X, W, bias, Y, N, st, ldy) : rowgemm16_launch2<K, M, true, false>(X, W, bias, Y, N, st, ldy);
    return relu ? rowgemm16_launch2<K, M, false, true>(X, W, bias, Y, N, st, ldy) : rowgemm16_launch2<K, M, false, false>(X, W, bias, Y, N, st, ldy);
}

// the 3-term bf16 split on the matrix cores (rowgemm_x3.hip) pays once the launch is long enough to be matrix-bound
inline bool rowgemm_x3_wanted(int64_t N, int K)
{
    return N * K < ((int64_t)1 << 30) && (tuning().rowgemm_x3 >= 2 || (tuning().rowgemm_x3 == 0 && N >= 65536));
}

// shapes the row-piece kernel is instantiated for (the dense layers of the GCN / GAT configs)
inline bool rowgemm16_shape(int K, int M) { return (K == 128 && M == 128) || (K == 64 && M == 128) || (K == 128 && M == 64) || (K == 64 && M == 64); }

// One WAVE owns a 32-row tile of X and ALL M output columns (M / 32 accumulators of v_mfma_f32_32x32x2_f32):
//   * its A operands come straight from global memory into registers in MFMA layout -- lane (row, kh) loads the
//     float4s X[row][8 j + 4 kh .. + 3]; the k index is permuted inside each block of 8 (lane half kh supplies
//     k = 8 j + 4 kh + i at step (j, i)), which a sum over k does not care about as long as B follows;
//   * W is staged ONCE per workgroup into LDS as Ws[k][M + 1] (transposing on the way for the torch Linear layout)
//     and read as B operands Ws[8 j + 4 kh + i][32 ct + lane % 32] (conflict-free: odd row stride);
//   * no workgroup barrier after the staging: waves walk tiles independently, so while one wave waits for its X
//     rows the other two or three on the SIMD keep the matrix pipe busy.  The X loads of the first tile are issued
//     before the staging, those of the next tile right after the last MFMA of the current one.
// fp32 in, fp32 accumulate, the k-chain of each output in a fixed order (bias is the initial accumulator).
template <int KBMAX, int MT, bool TRANS_W>
__global__ __launch_bounds__(kBlock) void rowgemm_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ Y,
                                                         int64_t N, int K, int M, int num_tiles, int ldy)
{
    extern __shared__ float Ws[];                      // [KB * 8][M + 1]
    const int ldw = M + 1;
    const int KB = (K + 7) / 8;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int total = gridDim.x * kWavesPerBlock;
    int tile = blockIdx.x * kWavesPerBlock + wave;

    float4 xa[KBMAX];
    auto load_x = [&](int t) {
        const int64_t row = (int64_t)t * 32 + l31;
        const float *src = X + row * K + 4 * kh;
#pragma unroll
        for (int j = 0; j < KBMAX; ++j) {
            xa[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < KB && row < N && 8 * j + 4 * kh < K) xa[j] = *reinterpret_cast<const float4 *>(src + 8 * j);
        }
    };
    if (tile < num_tiles) load_x(tile);

    {   // stage W (zero rows K .. 8 KB): batches of 8 independent 16-B loads per thread before the first LDS write
        const int total4 = K * M / 4;
        for (int base = 0; base < total4; base += 8 * kBlock) {
            float4 w4[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i4 = base + s * kBlock + threadIdx.x;
                w4[s] = i4 < total4 ? *reinterpret_cast<const float4 *>(W + (int64_t)i4 * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i = (base + s * kBlock + threadIdx.x) * 4;
                if (i < K * M) {
                    const float v[4] = {w4[s].x, w4[s].y, w4[s].z, w4[s].w};
                    if constexpr (TRANS_W) {           // W is [M][K]: element i = (m, k), 4 consecutive k
                        const int m = i / K, k = i - m * K;
#pragma unroll
                        for (int q = 0; q < 4; ++q) Ws[(k + q) * ldw + m] = v[q];
                    } else {                           // W is [K][M]: element i = (k, m), 4 consecutive m
                        const int k = i / M, m = i - k * M;
#pragma unroll
                        for (int q = 0; q < 4; ++q) Ws[k * ldw + m + q] = v[q];
                    }
                }
            }
        }
        for (int i = K * ldw + threadIdx.x; i < KB * 8 * ldw; i += kBlock) Ws[i] = 0.f;
    }
    __syncthreads();

    for (; tile < num_tiles; tile += total) {
        f32x16 acc[MT];
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
            const float b = bias ? bias[ct * 32 + l31] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = b;
        }
        const float *pb = Ws + (4 * kh) * ldw + l31;
#pragma unroll
        for (int j = 0; j < KBMAX; ++j) {
            if (j < KB) {
                const float av[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int ct = 0; ct < MT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], pb[(8 * j + i) * ldw + ct * 32], acc[ct], 0, 0, 0);
                }
            }
        }
        const int64_t row_base = (int64_t)tile * 32;
        const int next = tile + total;
        if (next < num_tiles) load_x(next);            // in flight during the stores and the other waves' MFMAs
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t r = row_base + (i & 3) + 8 * (i >> 2) + 4 * kh;
                if (r < N) Y[r * ldy + ct * 32 + l31] = acc[ct][i];
            }
        }
    }
}

struct RowGemmShape {
    int kbmax, mt;
    size_t lds;
};

inline bool rowgemm_shape(int32_t K, int32_t M, RowGemmShape &s)
{
    if (K <= 0 || M <= 0 || K % 4 != 0 || M % 32 != 0) return false;
    const int kb = (K + 7) / 8;
    s.kbmax = kb <= 4 ? 4 : kb <= 8 ? 8 : kb <= 16 ? 16 : kb <= 24 ? 24 : 0;
    s.mt = M / 32;
    if (!s.kbmax || (s.mt != 1 && s.mt != 2 && s.mt != 3 && s.mt != 4 && s.mt != 6)) return false;
    if (4 * s.kbmax + 16 * s.mt > 176) return false;                 // registers: A tile + accumulators
    s.lds = sizeof(float) * (size_t)kb * 8 * (size_t)(M + 1);
    return s.lds <= 150 * 1024;
}

template <int KBMAX, int MT>
int rowgemm_launch(const float *X, const float *W, const float *bias, float *Y, int64_t N, int K, int M, bool trans_w,
                   size_t lds, hipStream_t st, int ldy)
{
    const int64_t tiles = (N + 31) / 32;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: too many rows");
    // dynamic LDS above 64 KiB has to be enabled per kernel (host-side attribute, no sync)
    static PerDeviceOnce once[2];
    bool *raised = once[trans_w ? 1 : 0].slot();
    if (lds > 64 * 1024 && !*raised) {
        const hipError_t e = trans_w
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<KBMAX, MT, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<KBMAX, MT, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail((int)e, "stg_rowgemm_f32: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / (lds + 1024)));
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + kWavesPerBlock - 1) / kWavesPerBlock, 256 * per_cu);
    if (trans_w)
        hipLaunchKernelGGL((rowgemm_kernel<KBMAX, MT, true>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M,
                           (int)tiles, ldy);
    else
        hipLaunchKernelGGL((rowgemm_kernel<KBMAX, MT, false>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M,
                           (int)tiles, ldy);
    return check_launch("stg_rowgemm_f32");
}

template <int KBMAX>
int rowgemm_mt(int mt, const float *X, const float *W, const float *bias, float *Y, int64_t N, int K, int M, bool tw,
               size_t lds, hipStream_t st, int ldy)
{
    switch (mt) {
        case 1: return rowgemm_launch<KBMAX, 1>(X, W, bias, Y, N, K, M, tw, lds, st, ldy);
        case 2: return rowgemm_launch<KBMAX, 2>(X, W, bias, Y, N, K, M, tw, lds, st, ldy);
        case 3: return rowgemm_launch<KBMAX, 3>(X, W, bias, Y, N, K, M, tw, lds, st, ldy);
        case 4: return rowgemm_launch<KBMAX, 4>(X, W, bias, Y, N, K, M, tw, lds, st, ldy);
        default:
            if constexpr (4 * KBMAX + 16 * 6 <= 176) return rowgemm_launch<KBMAX, 6>(X, W, bias, Y, N, K, M, tw, lds, st, ldy);
            else return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: shape not covered");
    }
}

}  // namespace stg

extern "C" int stg_rowgemm_supported(int32_t K, int32_t M)
{
    stg::RowGemmShape s;
    return stg::rowgemm_shape(K, M, s) ? 1 : 0;
}

extern "C" int stg_rowgemm_strided_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N,
                                       int32_t K, int32_t M, int32_t ldy, int trans_w, void *stream);

extern "C" int stg_rowgemm_act_supported(int32_t K, int32_t M) { return stg::rowgemm16_shape(K, M) ? 1 : 0; }

extern "C" int stg_rowgemm_act_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K, int32_t M,
                                   int trans_w, int act, void *stream)
{
    using namespace stg;
    if (act != STG_ACT_NONE && act != STG_ACT_RELU) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_f32: unknown activation %d", act);
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_f32: negative N");
    if (!rowgemm16_shape(K, M)) return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_act_f32: K, M must be 64 or 128 (got %d, %d)", K, M);
    if (N == 0) return 0;
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_f32: X, W and Y must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tw = trans_w != 0, relu = act == STG_ACT_RELU;
    if (rowgemm_x3_wanted(N, K)) return rowgemm_x3_launch(K, M, X, W, bias, Y, N, tw, relu, stream, M);
    if (K == 128 && M == 128) return rowgemm16_launch<128, 128>(X, W, bias, Y, N, tw, relu, st, M);
    if (K == 64 && M == 128) return rowgemm16_launch<64, 128>(X, W, bias, Y, N, tw, relu, st, M);
    if (K == 128 && M == 64) return rowgemm16_launch<128, 64>(X, W, bias, Y, N, tw, relu, st, M);
    return rowgemm16_launch<64, 64>(X, W, bias, Y, N, tw, relu, st, M);
}

// ---- per-head products of a [N, heads K] matrix (rowgemm_x3.hip) ----------------------------------------------------------------
extern "C" int stg_rowgemm_heads_supported(int64_t N, int32_t K, int32_t M, int32_t heads)
{
    using namespace stg;
    const int mode = tuning().rowgemm_x3;
    return rowgemm16_shape(K, M) && heads > 0 && N > 0 && N * heads * K < ((int64_t)1 << 30) && mode != 1 && mode != 3 ? 1 : 0;
}

extern "C" int stg_rowgemm_heads_f32(const float *X, const float *W, float *Y, int64_t N, int32_t K, int32_t M, int32_t heads,
                                     void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_heads_f32: negative N");
    if (N == 0) return 0;
    if (!stg_rowgemm_heads_supported(N, K, M, heads))
        return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_heads_f32: K, M in {64, 128}, N heads K < 2^30 (got N=%lld K=%d M=%d heads=%d)",
                    (long long)N, K, M, heads);
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_heads_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_heads_f32: X, W and Y must be 16-byte aligned");
    return rowgemm_x3_heads_launch(K, M, X, W, Y, N, heads, stream);
}

// ---- the ReLU sign pattern as bits (rowgemm_x3.hip) --------------------------------------------------------------------------
extern "C" size_t stg_rowgemm_bits_words(int64_t N) { return N > 0 ? (size_t)((N + 31) / 32) * 128 : 0; }

extern "C" int stg_rowgemm_bits_supported(int64_t N, int32_t K, int32_t M)
{
    using namespace stg;
    const int mode = tuning().rowgemm_x3;
    return rowgemm16_shape(K, M) && N > 0 && N * K < ((int64_t)1 << 30) && N * M < ((int64_t)1 << 30) && mode != 1 && mode != 3 ? 1 : 0;
}

extern "C" int stg_rowgemm_act_bits_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K, int32_t M,
                                        int trans_w, int act, const uint32_t *bits_in, uint32_t *bits_out, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: negative N");
    if (N == 0) return 0;
    if (!stg_rowgemm_bits_supported(N, K, M))
        return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_act_bits_f32: K, M must be 64 or 128 and N K, N M < 2^30 (got N=%lld K=%d M=%d), "
                    "knob rowgemm_x3 neither 1 nor 3", (long long)N, K, M);
    if ((bits_in != nullptr) == (bits_out != nullptr))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: exactly one of bits_in / bits_out");
    if (bits_out && !(trans_w == 0 && act == STG_ACT_RELU))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: bits_out goes with trans_w = 0 and act = STG_ACT_RELU");
    if (bits_in && !(trans_w != 0 && act == STG_ACT_NONE))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: bits_in goes with trans_w = 1 and act = STG_ACT_NONE");
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_act_bits_f32: X, W and Y must be 16-byte aligned");
    return rowgemm_x3_bits_launch(K, M, X, W, bias, Y, N, bits_in, bits_out, stream);
}

extern "C" int stg_rowgemm_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K,
                               int32_t M, int trans_w, void *stream)
{
    return stg_rowgemm_strided_f32(X, W, bias, Y, N, K, M, M, trans_w, stream);
}

extern "C" int stg_rowgemm_strided_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N,
                                       int32_t K, int32_t M, int32_t ldy, int trans_w, void *stream)
{
    using namespace stg;
    if (ldy < M) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_strided_f32: ldy=%d < M=%d", ldy, M);
    if (N < 0 || K <= 0 || M <= 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: bad shape N=%lld K=%d M=%d", (long long)N, K, M);
    RowGemmShape s;
    if (!rowgemm_shape(K, M, s))
        return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: needs K %% 4 == 0, K <= 192, M %% 32 == 0, M / 32 in {1,2,3,4,6} and "
                    "K / 2 + M / 2 <= 176 registers (got K=%d M=%d)", K, M);
    if (N == 0) return 0;
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: X and W must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rowgemm16_shape(K, M) && N >= 4096 && ldy % 4 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0 && tuning().rowgemm16 != 1) {
        const bool tw = trans_w != 0;
        if (rowgemm_x3_wanted(N, K)) return rowgemm_x3_launch(K, M, X, W, bias, Y, N, tw, false, stream, ldy);
        if (K == 128 && M == 128) return rowgemm16_launch<128, 128>(X, W, bias, Y, N, tw, false, st, ldy);
        if (K == 64 && M == 128) return rowgemm16_launch<64, 128>(X, W, bias, Y, N, tw, false, st, ldy);
        if (K == 128 && M == 64) return rowgemm16_launch<128, 64>(X, W, bias, Y, N, tw, false, st, ldy);
        return rowgemm16_launch<64, 64>(X, W, bias, Y, N, tw, false, st, ldy);
    }
    switch (s.kbmax) {
        case 4: return rowgemm_mt<4>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        case 8: return rowgemm_mt<8>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        case 16: return rowgemm_mt<16>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        default: return rowgemm_mt<24>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
    }
}
