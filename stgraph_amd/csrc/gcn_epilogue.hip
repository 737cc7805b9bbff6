// Backward of GCNConv's tail (bias + ReLU, nn/pytorch/static/gcn_conv.py:185-188) in one pass:
// ReLU mask and bias gradient (column sums) together.  HBM bound: reads g (+ out), writes g_act.
//
// Mapping: L = min(ceil(F / VEC), 256) lanes cover a row (VEC = 4 when F % 4 == 0: 16-B accesses,
// a row is one coalesced request), R = 256 / L rows per pass, a workgroup strides over the rows; each
// thread keeps up to KMAX column slots of running sums; the R row-lanes of a column meet in LDS; one
// partial row per workgroup goes to the workspace and a second tiny kernel adds the partials in a fixed
// order (no atomics: run-to-run identical).
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

constexpr int kMaxSlots = 4;
constexpr int kMaxGrid = 2048;     // partial rows; 8 workgroups per CU with kRowUnroll rows in flight each
constexpr int kRowUnroll = 8;
constexpr int kFinCols = 8, kFinGroups = kBlock / kFinCols;

template <int VEC, bool MASK>
__global__ __launch_bounds__(kBlock) void bias_act_bwd_kernel(const float *__restrict__ g, const float *__restrict__ out,
                                                              float *__restrict__ g_act, float *__restrict__ partial,
                                                              int N, int F, int L, int slots)
{
    __shared__ float red[kBlock * VEC];
    const int R = kBlock / L;
    const int cl = threadIdx.x % L, rl = threadIdx.x / L;
    float acc[kMaxSlots][VEC];
#pragma unroll
    for (int k = 0; k < kMaxSlots; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;

    if (rl < R) {
        // kRowUnroll rows in flight per thread (loads first, sums after, in row order)
        const int64_t step = (int64_t)gridDim.x * R;
        for (int64_t r0 = (int64_t)blockIdx.x * R + rl; r0 < N; r0 += step * kRowUnroll) {
#pragma unroll
            for (int k = 0; k < kMaxSlots; ++k) {
                const int col = (cl + k * L) * VEC;
                if (k < slots && col < F) {
                    float gv[kRowUnroll][VEC], ov[kRowUnroll][VEC];
#pragma unroll
                    for (int u = 0; u < kRowUnroll; ++u) {
                        const int64_t r = r0 + u * step;
                        if (r < N) {
                            vec_load<VEC>(gv[u], g + r * F + col);
                            if constexpr (MASK) vec_load<VEC>(ov[u], out + r * F + col);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kRowUnroll; ++u) {
                        const int64_t r = r0 + u * step;
                        if (r < N) {
                            if constexpr (MASK) {
#pragma unroll
                                for (int i = 0; i < VEC; ++i) gv[u][i] = ov[u][i] > 0.f ? gv[u][i] : 0.f;   // threshold_backward
                                vec_store<VEC>(g_act + r * F + col, gv[u]);
                            }
#pragma unroll
                            for (int i = 0; i < VEC; ++i) acc[k][i] += gv[u][i];
                        }
                    }
                }
            }
        }
    }
    if (!partial) return;
    for (int k = 0; k < slots; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] = acc[k][i];
        __syncthreads();
        const int col = (cl + k * L) * VEC;
        if (rl == 0 && col < F) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float s = 0.f;
                for (int q = 0; q < R; ++q) s += red[(q * L + cl) * VEC + i];
                partial[(int64_t)blockIdx.x * F + col + i] = s;
            }
        }
    }
}

// colsum[f] = sum_b partial[b][f]: kFinCols columns per workgroup, the partial rows dealt to kFinGroups
// thread groups (independent loads), then one fixed-order pass over the groups in LDS.
__global__ __launch_bounds__(kBlock) void colsum_finish_kernel(const float *__restrict__ partial, float *__restrict__ colsum,
                                                              int blocks, int F)
{
    __shared__ float red[kFinGroups][kFinCols];
    const int c = threadIdx.x % kFinCols, grp = threadIdx.x / kFinCols;
    const int f = blockIdx.x * kFinCols + c;
    float s = 0.f;
    if (f < F) {
#pragma unroll 4
        for (int b = grp; b < blocks; b += kFinGroups) s += partial[(int64_t)b * F + f];
    }
    red[grp][c] = s;
    __syncthreads();
    if (grp == 0 && f < F) {
        float t = 0.f;
        for (int q = 0; q < kFinGroups; ++q) t += red[q][c];
        colsum[f] = t;
    }
}

// A matrix ONE workgroup can hold in registers (Cora's 2708 x 16): every load of a thread issued before the first value is used
// (what a small launch costs beyond its ~ 4.5 us is its number of dependent memory round trips), column sums by a tree over the
// row lanes in LDS, written by this launch -- no partials, no finish launch.  L lanes of VEC floats per row, R = 1024 / L rows per
// pass, P passes.
constexpr int kSmallBlock = 1024;

template <int VEC, int P, bool MASK>
__global__ __launch_bounds__(kSmallBlock) void bias_act_bwd_small_kernel(const float *__restrict__ g, const float *__restrict__ out,
                                                                          float *__restrict__ g_act, float *__restrict__ colsum,
                                                                          int N, int F, int L)
{
    __shared__ float red[kSmallBlock * VEC];
    const int R = kSmallBlock / L;
    const int cl = (int)threadIdx.x % L, rl = (int)threadIdx.x / L;
    const int col = cl * VEC;
    const bool live = rl < R && col < F;
    float gv[P][VEC], ov[P][VEC];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = rl + p * R;
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv[p][i] = 0.f, ov[p][i] = 1.f;
        if (live && r < N) {
            vec_load<VEC>(gv[p], g + (int64_t)r * F + col);
            if constexpr (MASK) vec_load<VEC>(ov[p], out + (int64_t)r * F + col);
        }
    }
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = rl + p * R;
        if (live && r < N) {
            if constexpr (MASK) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) gv[p][i] = ov[p][i] > 0.f ? gv[p][i] : 0.f;      // threshold_backward
                vec_store<VEC>(g_act + (int64_t)r * F + col, gv[p]);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += gv[p][i];
        }
    }
    if (!colsum) return;                                         // kernel-uniform
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] = live ? acc[i] : 0.f;
    __syncthreads();
    int top = 1;
    while (top < R) top <<= 1;
    for (int off = top / 2; off > 0; off >>= 1) {                // over the row lanes, fixed order (R need not be a power of two)
        if (rl < off && rl + off < R) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] += red[(threadIdx.x + off * L) * VEC + i];
        }
        __syncthreads();
    }
    if (rl == 0 && col < F) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) colsum[col + i] = red[threadIdx.x * VEC + i];
    }
}

struct Shape {
    int vec, L, slots, grid;
};

bool shape_for(int32_t N, int32_t F, Shape &s)
{
    s.vec = (F % 4 == 0) ? 4 : 1;
    const int lanes = (F + s.vec - 1) / s.vec;
    s.L = std::min(lanes, kBlock);
    s.slots = (lanes + s.L - 1) / s.L;
    if (s.slots > kMaxSlots) return false;
    const int R = kBlock / s.L;
    // enough workgroups to fill 256 CUs several times over, few enough that the partials stay small
    s.grid = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)N + R - 1) / R / kRowUnroll + 1, kMaxGrid));
    return true;
}

}  // namespace
}  // namespace stg

// y = act(y + bias[f]) in place, one pass: the tail of a GCN layer whose aggregation ran BEFORE the weight product
// (nn/functional._InputLayer).  16 B per thread when F % 4 == 0 and y is 16-byte aligned.
namespace stg {
template <int VEC>
__global__ __launch_bounds__(kBlock) void bias_act_fwd_kernel(float *__restrict__ y, const float *__restrict__ bias,
                                                              int64_t total, int F, int act)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock * VEC;
    for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC; i < total; i += stride) {
        float v[VEC];
        vec_load<VEC>(v, y + i);
        const int f = (int)(i % F);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (bias) v[k] = v[k] + bias[f + k];
            if (act == STG_ACT_RELU) v[k] = v[k] < 0.f ? 0.f : v[k];
        }
        vec_store<VEC>(y + i, v);
    }
}
}  // namespace stg

extern "C" int stg_bias_act_fwd(float *y, const float *bias, int32_t act, int32_t N, int32_t F, void *stream_)
{
    using namespace stg;
    if (N < 0 || F <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_fwd: bad shape N=%d F=%d", N, F);
    if (act != STG_ACT_NONE && act != STG_ACT_RELU)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_fwd: unknown activation %d", act);
    if (N == 0 || (!bias && act == STG_ACT_NONE)) return 0;
    if (!y) return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_fwd: NULL pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t total = (int64_t)N * F;
    const bool v4 = F % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
    const int64_t per = (int64_t)kBlock * (v4 ? 4 : 1);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((total + per - 1) / per, 256 * 32));
    if (v4) hipLaunchKernelGGL(bias_act_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, stream, y, bias, total, F, act);
    else hipLaunchKernelGGL(bias_act_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, stream, y, bias, total, F, act);
    return check_launch("stg_bias_act_fwd");
}

extern "C" size_t stg_bias_act_bwd_workspace_bytes(int32_t N, int32_t F)
{
    stg::Shape s;
    if (N <= 0 || F <= 0 || !stg::shape_for(N, F, s)) return 0;
    return (size_t)s.grid * (size_t)F * sizeof(float);
}

extern "C" int stg_bias_act_bwd(const float *g, const float *out, float *g_act, float *colsum, int32_t N, int32_t F,
                                void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0 || F <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_bwd: bad shape N=%d F=%d", N, F);
    if (!out != !g_act) return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_bwd: out and g_act go together");
    if (!out && !colsum) return 0;
    Shape s;
    if (!shape_for(std::max(N, 1), F, s)) return fail(STG_ERR_UNSUPPORTED, "stg_bias_act_bwd: F=%d too wide", F);
    if (N == 0) {
        if (colsum) return zero_async(colsum, sizeof(float) * (size_t)F, stream);
        return 0;
    }
    if (!g) return fail(STG_ERR_INVALID_ARGUMENT, "stg_bias_act_bwd: NULL pointer argument");
    const uintptr_t align = reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(g_act);
    if (s.vec == 4 && align % 16 != 0) {          // unaligned views: scalar lanes
        s.vec = 1;
        s.L = std::min(F, kBlock);
        s.slots = (F + s.L - 1) / s.L;
        if (s.slots > kMaxSlots) return fail(STG_ERR_UNSUPPORTED, "stg_bias_act_bwd: F=%d too wide for unaligned rows", F);
    }
    float *partial = nullptr;
    // a matrix one workgroup holds in registers (Cora's 2708 x 16): one launch, no partials (bias_act_bwd_small_kernel)
    {
        const int vec = (F % 4 == 0 && align % 16 == 0) ? 4 : 1;
        const int L = (F + vec - 1) / vec;
        if (colsum && L <= kSmallBlock) {
            const int R = kSmallBlock / L, passes = (N + R - 1) / R;
            const int pmax = vec == 4 ? 12 : 48;
            if (passes <= pmax) {
#define STG_BAS(V_, P_)                                                                                                            \
    do {                                                                                                                           \
        if (out) hipLaunchKernelGGL((bias_act_bwd_small_kernel<V_, P_, true>), dim3(1), dim3(kSmallBlock), 0, stream, g, out, g_act, colsum, N, F, L); \
        else hipLaunchKernelGGL((bias_act_bwd_small_kernel<V_, P_, false>), dim3(1), dim3(kSmallBlock), 0, stream, g, out, g_act, colsum, N, F, L); \
    } while (0)
                if (vec == 4) {
                    if (passes <= 4) STG_BAS(4, 4);
                    else STG_BAS(4, 12);
                } else {
                    STG_BAS(1, 48);                              // (a 16-pass instance of the scalar form spills 800 registers: compiler)
                }
#undef STG_BAS
                return check_launch("stg_bias_act_bwd");
            }
        }
    }
    if (colsum) {
        const size_t need = (size_t)s.grid * (size_t)F * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(STG_ERR_WORKSPACE, "stg_bias_act_bwd: workspace %zu < required %zu", workspace_bytes, need);
        partial = static_cast<float *>(workspace);
    }
    const dim3 grid(s.grid), block(kBlock);
    if (s.vec == 4) {
        if (out) hipLaunchKernelGGL((bias_act_bwd_kernel<4, true>), grid, block, 0, stream, g, out, g_act, partial, N, F, s.L, s.slots);
        else hipLaunchKernelGGL((bias_act_bwd_kernel<4, false>), grid, block, 0, stream, g, out, g_act, partial, N, F, s.L, s.slots);
    } else {
        if (out) hipLaunchKernelGGL((bias_act_bwd_kernel<1, true>), grid, block, 0, stream, g, out, g_act, partial, N, F, s.L, s.slots);
        else hipLaunchKernelGGL((bias_act_bwd_kernel<1, false>), grid, block, 0, stream, g, out, g_act, partial, N, F, s.L, s.slots);
    }
    if (colsum)
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((F + kFinCols - 1) / kFinCols), block, 0, stream, partial, colsum, s.grid, F);
    return check_launch("stg_bias_act_bwd");
}

// norm = in_deg^-0.5 with inf -> 0: the scripts' `torch.pow(deg, -0.5); norm[isinf(norm)] = 0`
// (benchmarking/gcn/seastar/train.py:53-57; dynamic loop: once per snapshot) as one launch instead of five.
// 1 / sqrt(d), both correctly rounded (torch's pow(x, -0.5) is an rsqrt approximation on the device and libm's pow on
// the host: the three agree to 1 ulp).
namespace stg {
namespace {
__global__ __launch_bounds__(kBlock) void degree_norm_kernel(const int *__restrict__ degrees, const int *__restrict__ row_offsets,
                                                             float *__restrict__ norm, int64_t N)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        const int d = degrees ? degrees[i] : row_offsets[i + 1] - row_offsets[i];
        norm[i] = d > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)d)) : 0.f;
    }
}
}  // namespace
}  // namespace stg

extern "C" int stg_degree_norm_f32(const int32_t *degrees, const int32_t *row_offsets, float *norm, int64_t N, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_degree_norm_f32: negative size");
    if (N == 0) return 0;
    if ((!degrees && !row_offsets) || !norm) return fail(STG_ERR_INVALID_ARGUMENT, "stg_degree_norm_f32: NULL pointer argument");
    const int blocks = (int)std::min<int64_t>((N + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL(degree_norm_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), degrees, row_offsets,
                       norm, N);
    return check_launch("stg_degree_norm_f32");
}
