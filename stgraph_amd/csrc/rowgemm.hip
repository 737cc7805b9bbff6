// Y[N,M] = X[N,K] * op(W) + bias   with N = number of vertices (large) and K, M = feature widths
// (<= a few hundred): the forward / input-gradient GEMMs of the dense layers next to the Seastar
// kernels (TGCN gate Linears: K = 128 -> M = 64 and back).  rocBLAS/hipBLASLt choose 64x32- or
// 128x224-wide macro tiles for these skinny shapes and land at 2-4x the memory-bound time (measured:
// 29.7 us for [50K,64] x [64,128], whose 38 MB of traffic take 7 us; profiles/r01).
//
// Workgroup = 256 threads = one 64-row tile of X.  X tile and W are staged in LDS (X rows padded by
// one float so the MFMA A-fragment column reads are conflict-free; W stored [K][M] so B-fragment reads
// are one bank per lane), then v_mfma_f32_32x32x2_f32 over 32x32 output tiles dealt round-robin to
// the 4 waves; each lane's 16 results are stored as 128-B row segments.  fp32 in / fp32 accumulate,
// k-ordered fma chain (same accuracy class as an fp32 BLAS).
#include "stg_common.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kRgRows = 64;

template <bool TRANS_W>
__global__ __launch_bounds__(kBlock) void rowgemm_kernel(const float *__restrict__ X,
                                                         const float *__restrict__ W,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ Y, int64_t N, int K, int M)
{
    extern __shared__ float lds[];
    const int ldx = K + 1;
    const int ldw = M + 1;                    // odd row stride: the transposing stage below is conflict-free
    float *Xs = lds;                          // [kRgRows][K + 1]
    float *Ws = lds + kRgRows * ldx;          // [K][M + 1]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int64_t row_base = (int64_t)blockIdx.x * kRgRows;

    // stage W as Ws[k][m] (no integer division, coalesced global reads, conflict-free LDS writes)
    if constexpr (TRANS_W) {                  // W is [M][K]: a wave streams one W row along k
        for (int m = wave; m < M; m += kWavesPerBlock)
            for (int k = lane; k < K; k += kWave) Ws[k * ldw + m] = W[m * K + k];
    } else {                                  // W is [K][M]: a wave streams one W row along m
        for (int k = wave; k < K; k += kWavesPerBlock)
            for (int m = lane; m < M; m += kWave) Ws[k * ldw + m] = W[k * M + m];
    }
    // stage the X tile: each wave streams whole rows with 16-B loads; rows beyond N are zero
    for (int r = wave; r < kRgRows; r += kWavesPerBlock) {
        const bool ok = row_base + r < N;
        const float *src = X + (row_base + r) * K;
        for (int c = lane * 4; c < K; c += kWave * 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4 *>(src + c);
            float *d = Xs + r * ldx + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();

    const int kh = lane >> 5, l31 = lane & 31;
    const int col_tiles = M / 32;
    const int tiles = (kRgRows / 32) * col_tiles;
    for (int t = wave; t < tiles; t += kWavesPerBlock) {
        const int rt = t / col_tiles, ct = t - rt * col_tiles;
        f32x16 acc;
        const float b = bias ? bias[ct * 32 + l31] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = b;          // bias as the initial accumulator (column = lane)
        const float *pa = Xs + (rt * 32 + l31) * ldx + kh;
        const float *pb = Ws + kh * ldw + ct * 32 + l31;
#pragma unroll 8
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k * ldw], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row_base + rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (r < N) Y[r * M + ct * 32 + l31] = acc[i];
        }
    }
}

}  // namespace stg

extern "C" int stg_rowgemm_supported(int32_t K, int32_t M)
{
    if (K <= 0 || M <= 0 || K % 4 != 0 || M % 32 != 0) return 0;
    const size_t lds = sizeof(float) * ((size_t)stg::kRgRows * (K + 1) + (size_t)K * (M + 1));
    return lds <= 96 * 1024 ? 1 : 0;
}

extern "C" int stg_rowgemm_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K,
                               int32_t M, int trans_w, void *stream)
{
    using namespace stg;
    if (N < 0 || K <= 0 || M <= 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: bad shape N=%lld K=%d M=%d", (long long)N, K, M);
    if (!stg_rowgemm_supported(K, M))
        return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: needs K %% 4 == 0, M %% 32 == 0 and "
                    "4 (64 (K+1) + K M) <= 96 KiB (got K=%d M=%d)", K, M);
    if (N == 0) return 0;
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: X and W must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)kRgRows * (K + 1) + (size_t)K * (M + 1));
    // dynamic LDS above 64 KiB has to be enabled per kernel (idempotent, host-side, no sync)
    static bool raised[2] = {false, false};
    if (lds > 64 * 1024 && !raised[trans_w ? 1 : 0]) {
        const hipError_t e = trans_w
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return fail((int)e, "stg_rowgemm_f32: %s", hipGetErrorString(e));
        raised[trans_w ? 1 : 0] = true;
    }
    const unsigned blocks = (unsigned)((N + kRgRows - 1) / kRgRows);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (trans_w)
        hipLaunchKernelGGL((rowgemm_kernel<true>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M);
    else
        hipLaunchKernelGGL((rowgemm_kernel<false>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M);
    return check_launch("stg_rowgemm_f32");
}
