// Y[N,M] = X[N,K] * op(W) + bias   with N = number of vertices (large) and K, M = feature widths
// (<= 256): the forward / input-gradient GEMMs of the dense layers next to the Seastar kernels
// (TGCN gate Linears: K = 128 -> M = 64 and back).  rocBLAS/hipBLASLt choose 64x32- or 128x224-wide
// macro tiles for these skinny shapes and land at 2-4x the memory-bound time (measured in situ: 29.7 us
// for [50K,64] x [64,128], whose 38 MB of traffic take 7 us; profiles/r01).
//
// Persistent workgroups (256 threads, up to 3 per CU): W is staged into LDS ONCE per workgroup, then
// the workgroup walks 64-row tiles of X: the global loads of tile i+1 are issued (into registers)
// before the MFMA loop of tile i and written to LDS after it (issue-early / write-late), so the matrix
// pipe and the memory pipe overlap inside a workgroup and across the co-resident ones.  X rows are padded by one
// float (conflict-free MFMA A-fragment column reads), W is stored [K][M+1] (conflict-free transposing
// stage and B-fragment reads).  v_mfma_f32_32x32x2_f32 over 32x32 output tiles dealt round-robin to the
// 4 waves, results stored as 128-B row segments.  fp32 in / fp32 accumulate, k-ordered fma chain.
#include "stg_common.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kRgRows = 64;
constexpr int kRgMaxK = 256;
constexpr int kRgRegs = kRgRows * (kRgMaxK / 4) / kBlock;       // float4 registers per thread for one X tile

template <bool TRANS_W>
__global__ __launch_bounds__(kBlock) void rowgemm_kernel(const float *__restrict__ X,
                                                         const float *__restrict__ W,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ Y, int64_t N, int K, int M,
                                                         int num_tiles)
{
    extern __shared__ float lds[];
    const int ldx = K + 1;
    const int ldw = M + 1;
    float *Ws = lds;                                   // [K][M + 1]
    float *Xs = lds + K * ldw;                         // one X tile [kRgRows][K + 1]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int q = K / 4;                               // float4 per row
    const int per_tile = kRgRows * q;                  // float4 per tile

    // stage W as Ws[k][m]: batches of 8 independent 16-B loads per thread are issued before the first
    // LDS write (a one-load-one-store loop serialises on global latency: ~1 us per iteration)
    {
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
                        for (int j = 0; j < 4; ++j) Ws[(k + j) * ldw + m] = v[j];
                    } else {                           // W is [K][M]: element i = (k, m), 4 consecutive m
                        const int k = i / M, m = i - k * M;
#pragma unroll
                        for (int j = 0; j < 4; ++j) Ws[k * ldw + m + j] = v[j];
                    }
                }
            }
        }
    }

    // per-thread slots of an X tile (tile independent: computed once, no division in the loop)
    float4 regs[kRgRegs];
    int goff[kRgRegs], loff[kRgRegs], rrow[kRgRegs];
#pragma unroll
    for (int s = 0; s < kRgRegs; ++s) {
        const int i = threadIdx.x + s * kBlock;
        const int r = i / q, c = (i - r * q) * 4;
        const bool ok = i < per_tile;
        rrow[s] = ok ? r : (1 << 30);                  // a row index that is never < N - row_base
        goff[s] = r * K + c;
        loff[s] = r * ldx + c;
    }
    auto load_tile = [&](int tile) {                   // global -> registers (rows beyond N are zero)
        const int64_t row_base = (int64_t)tile * kRgRows;
        const float *src = X + row_base * K;
        const int64_t rows_left = N - row_base;
#pragma unroll
        for (int s = 0; s < kRgRegs; ++s) {
            regs[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rrow[s] < rows_left) regs[s] = *reinterpret_cast<const float4 *>(src + goff[s]);
        }
    };
    auto store_tile = [&]() {                          // registers -> LDS
#pragma unroll
        for (int s = 0; s < kRgRegs; ++s) {
            if (rrow[s] < kRgRows) {
                float *d = Xs + loff[s];
                d[0] = regs[s].x; d[1] = regs[s].y; d[2] = regs[s].z; d[3] = regs[s].w;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < num_tiles) {
        load_tile(tile);
        store_tile();
    }
    __syncthreads();

    const int kh = lane >> 5, l31 = lane & 31;
    const int col_tiles = M / 32;
    const int tiles = (kRgRows / 32) * col_tiles;
    for (; tile < num_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < num_tiles) load_tile(next);                         // in flight during the MFMA loop below
        const int64_t row_base = (int64_t)tile * kRgRows;
        for (int t = wave; t < tiles; t += kWavesPerBlock) {
            const int rt = t / col_tiles, ct = t - rt * col_tiles;
            f32x16 acc;
            const float b = bias ? bias[ct * 32 + l31] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = b;                   // bias as the initial accumulator
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
        __syncthreads();                                               // every wave is done reading Xs
        if (next < num_tiles) store_tile();
        __syncthreads();
    }
}

}  // namespace stg

extern "C" int stg_rowgemm_supported(int32_t K, int32_t M)
{
    if (K <= 0 || M <= 0 || K % 4 != 0 || K > stg::kRgMaxK || M % 32 != 0) return 0;
    const size_t lds = sizeof(float) * ((size_t)stg::kRgRows * (K + 1) + (size_t)K * (M + 1));
    return lds <= 150 * 1024 ? 1 : 0;
}

extern "C" int stg_rowgemm_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K,
                               int32_t M, int trans_w, void *stream)
{
    using namespace stg;
    if (N < 0 || K <= 0 || M <= 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: bad shape N=%lld K=%d M=%d", (long long)N, K, M);
    if (!stg_rowgemm_supported(K, M))
        return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: needs K %% 4 == 0, K <= %d, M %% 32 == 0 and "
                    "4 (128 (K+1) + K (M+1)) <= 150 KiB (got K=%d M=%d)", kRgMaxK, K, M);
    if (N == 0) return 0;
    if (!X || !W || !Y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_rowgemm_f32: X and W must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)kRgRows * (K + 1) + (size_t)K * (M + 1));
    // dynamic LDS above 64 KiB has to be enabled per kernel (host-side attribute, no sync)
    static bool raised[2] = {false, false};
    if (lds > 64 * 1024 && !raised[trans_w ? 1 : 0]) {
        const hipError_t e = trans_w
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(rowgemm_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return fail((int)e, "stg_rowgemm_f32: %s", hipGetErrorString(e));
        raised[trans_w ? 1 : 0] = true;
    }
    const int64_t num_tiles = (N + kRgRows - 1) / kRgRows;
    if (num_tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_rowgemm_f32: too many rows");
    const int per_cu = lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1);   // 160 KiB of LDS per CU
    const unsigned blocks = (unsigned)std::min<int64_t>(num_tiles, 256 * per_cu);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (trans_w)
        hipLaunchKernelGGL((rowgemm_kernel<true>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M,
                           (int)num_tiles);
    else
        hipLaunchKernelGGL((rowgemm_kernel<false>), dim3(blocks), dim3(kBlock), lds, st, X, W, bias, Y, N, K, M,
                           (int)num_tiles);
    return check_launch("stg_rowgemm_f32");
}
