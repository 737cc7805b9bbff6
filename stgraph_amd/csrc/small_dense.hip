// The backward of a SMALL dense layer y = x W (x [N, K], W [K, M]; Cora's second GCNConv: 2708 x 16 -> 7) as ONE one-workgroup launch:
// gx = g W^T and gw = x^T g.  Inside a replayed HIP graph a launch costs ~ 4.5 us whatever it does and the library GEMM runs each
// of the two on one or two workgroups (K = 2708 reduction: 10.8 us; 4.9 us -- tools/diag/cora_kernels.py); here every load of a thread
// run on the fp32 matrix instruction of one 16-wave workgroup, the waves' partial sums of gw added in LDS in a fixed order (run-to-run
// identical).
// Reference: nn/pytorch/static/gcn_conv.py:158 `torch.mm(h, self.weight)` (its autograd backward: two mm).
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

constexpr int kSdThreads = 1024, kSdWaves = kSdThreads / kWave;

typedef float sd_f32x4 __attribute__((ext_vector_type(4)));

// Both products on v_mfma_f32_16x16x4_f32 (A[i][k]: lane (i = lane % 16, k = lane / 16); B[k][j]: lane (j, k); D[4 (lane / 16) + r][lane % 16]):
//   gw [K, M] = x^T g: A = x^T (16 columns of x by 4 rows: 64 consecutive floats of x, one per lane), B = g (4 rows by 16 columns), one
//     instruction per 4 rows, the waves take 4-row groups in turn and add their 16 x 16 sums in LDS in a fixed order;
//   gx [N, K] = g W^T: A = W (16 rows of W by 4 of its columns), B = g^T (4 columns of g by 16 rows), ceil(M / 4) instructions per
//     16 rows; a lane ends with four consecutive columns of its row.
// (A first version kept rows in lanes and summed the K x M outer products with wave butterflies: 73 us.)
// colsum != NULL (x is the OUTPUT of a ReLU layer, gcn_conv.py:185-188): gx is written as gx * [x > 0] -- the gradient of that layer's
// pre-activation -- and colsum [K] = its column sums, that layer's bias gradient, as sum_m W[k, m] S[k, m] with S = [x > 0]^T g formed
// beside gw (one more matrix instruction per 4 rows).
__global__ __launch_bounds__(kSdThreads) void mm_bwd_small_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                                  const float *__restrict__ W, float *__restrict__ gx, float *__restrict__ gw,
                                                                  int N, int K, int M, float *__restrict__ colsum)
{
    __shared__ float red[kSdWaves][256];
    __shared__ float redm[kSdWaves][256];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    // Every load of a chunk is issued before the first product (a loop of load -> product per 4-row group is one memory round trip
    // per group: measured 33 us for the two products of Cora's layer).  Workgroup 0 forms gw (a chunk = 44 groups per wave: 2816
    // rows), workgroups 1 .. form gx (4 tiles per wave: 1024 rows each) -- together in one workgroup the two sets of loads do not
    // fit the 128 registers of a 16-wave workgroup (302 spilled).
    // Loads through buffer descriptors: ONE offset register per lane, the chunk's stride in a scalar, rows past the end (and lanes
    // past the last column: offset 2^31) read as 0 by the range check -- 64-bit addresses per load were 300 spilled registers.
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((int64_t)N * K * 4), 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, (int)((int64_t)N * M * 4), 0x00020000);
    auto ld = [](const __amdgpu_buffer_rsrc_t &rs, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    };
    constexpr unsigned kNone = 0x80000000u;
    if (blockIdx.x != 0) {
        // ---- gx
        constexpr int UT = 4;
        const int tile0 = ((int)blockIdx.x - 1) * kSdWaves * UT;
        float wa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wa[s] = (n16 < K && 4 * s + kq < M) ? W[n16 * M + 4 * s + kq] : 0.f;
        float gb[UT][4];
        unsigned vg[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) vg[s] = 4 * s + kq < M ? (unsigned)(((tile0 + wave) * 16 + n16) * M + 4 * s + kq) * 4u : kNone;
        const unsigned tstride = (unsigned)(kSdWaves * 16 * M) * 4u;
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) gb[u][s] = ld(rsG, vg[s], u * tstride);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int row = (tile0 + u * kSdWaves + wave) * 16 + n16;
            sd_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (4 * s < M) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], gb[u][s], o, 0, 0, 0);     // (kernel-uniform)
            if (row < N) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kq + r < K) {
                        float v = o[r];
                        if (colsum) v = x[(int64_t)row * K + 4 * kq + r] > 0.f ? v : 0.f;                  // (kernel-uniform branch)
                        gx[(int64_t)row * K + 4 * kq + r] = v;
                    }
            }
        }
        return;
    }
    // ---- gw
    sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accm = {0.f, 0.f, 0.f, 0.f};
    const int groups = (N + 3) / 4;
    constexpr int UG = 44;
    const unsigned vx = n16 < K ? (unsigned)((wave * 4 + kq) * K + n16) * 4u : kNone;
    const unsigned vgw = n16 < M ? (unsigned)((wave * 4 + kq) * M + n16) * 4u : kNone;
    const unsigned sx = (unsigned)(kSdWaves * 4 * K) * 4u, sg = (unsigned)(kSdWaves * 4 * M) * 4u;        // one round of the waves
    for (int base = 0; base < groups; base += kSdWaves * UG) {
        float a[UG], b[UG];
        const unsigned bx = (unsigned)base * 4u * (unsigned)K * 4u, bg = (unsigned)base * 4u * (unsigned)M * 4u;
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            a[u] = ld(rsX, vx, bx + u * sx);
            b[u] = ld(rsG, vgw, bg + u * sg);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            if (colsum) accm = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u] > 0.f ? 1.f : 0.f, b[u], accm, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][(4 * kq + r) * 16 + n16] = acc[r];
        redm[wave][(4 * kq + r) * 16 + n16] = accm[r];
    }
    __syncthreads();
    float sm = 0.f;
    if (tid < 256) {
        const int k = tid >> 4, m = tid & 15;
        if (k < K && m < M) {
            float s = 0.f;
            for (int w = 0; w < kSdWaves; ++w) s = s + red[w][tid];
            gw[k * M + m] = s;
            if (colsum) {
                for (int w = 0; w < kSdWaves; ++w) sm = sm + redm[w][tid];
                sm = sm * W[k * M + m];
            }
        }
    }
    if (colsum) {                                                            // colsum[k] = sum_m W[k, m] S[k, m]: the 16 lanes m of row k, in order
        __syncthreads();
        if (tid < 256) red[0][tid] = sm;
        __syncthreads();
        if (tid < K) {
            float s = 0.f;
            for (int m = 0; m < M; ++m) s = s + red[0][tid * 16 + m];
            colsum[tid] = s;
        }
    }
}

// c [Ka, Mb] = a^T b for a [N, Ka] of any width and b [N, Mb <= 16] (Cora's first layer: x^T g, 1433 x 2708 by 2708 x 16 -- 14.7 us on
// the library GEMM): one 16-wave workgroup per 16 columns of a, the rows' 4-row groups dealt to its waves exactly as in the gw part
// above (every load of a chunk before the first product), the waves' 16 x 16 sums added in LDS in a fixed order.
__global__ __launch_bounds__(kSdThreads) void gemm_tn_small_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                   float *__restrict__ c, int N, int Ka, int Mb)
{
    __shared__ float red[kSdWaves][256];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int m0 = (int)blockIdx.x * 16;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a), 0, (int)((int64_t)N * Ka * 4), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, (int)((int64_t)N * Mb * 4), 0x00020000);
    auto ld = [](const __amdgpu_buffer_rsrc_t &rs, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
    };
    constexpr unsigned kNone = 0x80000000u;
    sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int groups = (N + 3) / 4;
    constexpr int UG = 44;
    const unsigned va = m0 + n16 < Ka ? (unsigned)((wave * 4 + kq) * Ka + m0 + n16) * 4u : kNone;
    const unsigned vb = n16 < Mb ? (unsigned)((wave * 4 + kq) * Mb + n16) * 4u : kNone;
    const unsigned sa = (unsigned)(kSdWaves * 4 * Ka) * 4u, sb = (unsigned)(kSdWaves * 4 * Mb) * 4u;
    for (int base = 0; base < groups; base += kSdWaves * UG) {
        float av[UG], bv[UG];
        const unsigned ba = (unsigned)base * 4u * (unsigned)Ka * 4u, bb = (unsigned)base * 4u * (unsigned)Mb * 4u;
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            av[u] = ld(rsA, va, ba + u * sa);
            bv[u] = ld(rsB, vb, bb + u * sb);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * 16 + n16] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int k = tid >> 4, m = tid & 15;
        if (m0 + k < Ka && m < Mb) {
            float s = 0.f;
            for (int w = 0; w < kSdWaves; ++w) s = s + red[w][tid];
            c[(int64_t)(m0 + k) * Mb + m] = s;
        }
    }
}

}  // namespace
}  // namespace stg

extern "C" int stg_gemm_tn_small_supported(int64_t N, int32_t Ka, int32_t Mb)
{
    // (32-bit byte offsets into a and b)
    return N > 0 && N <= 65536 && Ka > 0 && Mb > 0 && Mb <= 16 && N * (int64_t)Ka < ((int64_t)1 << 29);
}

extern "C" int stg_gemm_tn_small_f32(const float *a, const float *b, float *c, int64_t N, int32_t Ka, int32_t Mb, void *stream_)
{
    using namespace stg;
    if (!stg_gemm_tn_small_supported(N, Ka, Mb))
        return fail(STG_ERR_UNSUPPORTED, "stg_gemm_tn_small_f32: N=%lld Ka=%d Mb=%d (N <= 65536, Mb <= 16, N Ka < 2^29)", (long long)N, Ka, Mb);
    if (!a || !b || !c) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_small_f32: NULL pointer argument");
    hipLaunchKernelGGL(gemm_tn_small_kernel, dim3((unsigned)((Ka + 15) / 16)), dim3(kSdThreads), 0, static_cast<hipStream_t>(stream_), a, b, c,
                       (int)N, Ka, Mb);
    return check_launch("stg_gemm_tn_small_f32");
}

extern "C" int stg_mm_bwd_small_supported(int64_t N, int32_t K, int32_t M)
{
    return N > 0 && N <= 65536 && K > 0 && K <= 16 && M > 0 && M <= 16;
}

extern "C" int stg_mm_bwd_small(const float *g, const float *x, const float *W, float *gx, float *gw, float *relu_colsum, int64_t N,
                                int32_t K, int32_t M, void *stream_)
{
    using namespace stg;
    if (!stg_mm_bwd_small_supported(N, K, M))
        return fail(STG_ERR_UNSUPPORTED, "stg_mm_bwd_small: N=%lld K=%d M=%d (N <= 65536, K, M <= 16)", (long long)N, K, M);
    if (!g || !x || !W || !gx || !gw) return fail(STG_ERR_INVALID_ARGUMENT, "stg_mm_bwd_small: NULL pointer argument");
    const unsigned gx_groups = (unsigned)((N + 1023) / 1024);                // workgroup 0: gw; 1 ..: gx, 1024 rows each
    hipLaunchKernelGGL(mm_bwd_small_kernel, dim3(1 + gx_groups), dim3(kSdThreads), 0, static_cast<hipStream_t>(stream_), g, x, W, gx, gw, (int)N, K, M, relu_colsum);
    return check_launch("stg_mm_bwd_small");
}
