// Y[N,M] = X[N,K] * op(W) + bias   with N = number of vertices (large) and K, M = feature widths
// (<= 192): the forward / input-gradient GEMMs of the dense layers next to the Seastar kernels
// (TGCN gate Linears: K = 128 -> M = 64 and back).  rocBLAS/hipBLASLt choose 64x32- or 128x224-wide
// macro tiles for these skinny shapes and land at 2-4x the memory-bound time (measured in situ: 29.7 us
// for [50K,128] x [128,64]^T, whose 38 MB of traffic take 7 us; profiles/r01).
#include <algorithm>

#include "stg_common.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

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
    switch (s.kbmax) {
        case 4: return rowgemm_mt<4>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        case 8: return rowgemm_mt<8>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        case 16: return rowgemm_mt<16>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
        default: return rowgemm_mt<24>(s.mt, X, W, bias, Y, N, K, M, trans_w != 0, s.lds, st, ldy);
    }
}
