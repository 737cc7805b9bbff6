// The model head of the static-temporal TGCN training step as two launches (forward, backward).
//
// Reference: benchmarking/static-temporal-tgcn/seastar/model.py:6-18 (relu -> Linear(hidden, feat) ->
// Linear(feat, 1)) and the training loop's per-timestep `torch.mean((y_out - y[t]) ** 2)` (train.py).  In torch that
// is ~20 launches per step forward + backward (relu, two skinny GEMMs with bias, sub, pow, mean, add, and their
// backward counterparts plus the gradient accumulations of the two tensors that fan out), each a few microseconds
// on |V| = 50K -- together as long as the fused TGCN cell itself.
//
//   forward : r = relu(h) [N,C]; y = r W1^T + b1 [N,32]; y_out = y W2^T + b2 [N]; partial[tile] = sum (y_out - t)^2
//             (a one-workgroup finish kernel adds the partials in a fixed order: loss = sum / N)
//   backward: dyo = 2 (y_out - t) / N * gl (+ g_yout); dyt = g_y + dyo W2 [N,32]; dh = (h > 0) (dyt W1) [N,C]
//             dyt and dyo leave the kernel too: the weight gradients dW1 = dyt^T r, db1 = colsum(dyt), dW2 = dyo^T y,
//             db2 = sum(dyo) are tall-skinny contractions done once per backward pass (stg_gemm_tn_multi_f32).
//
// One wave per 32-row tile, both GEMMs on v_mfma_f32_32x32x2_f32.  As in tgcn_cell_fused.hip the A operands are
// read from HBM directly in MFMA layout (lane = (row, k-half), 16-byte pieces, k permuted inside blocks of 8 --
// the B operand is read with the same permutation, so the products pair up correctly); the y_out row sums run on
// the DPP network.  The GEMM summation order differs from rocBLAS': results agree to fp32 rounding (tests: 1e-5).
#include "stg_common.hpp"

namespace stg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHeadF = 32;            // width of y: one 32-column MFMA block
constexpr int64_t kHead16MinRows = 4096;   // N = 10 K .. 400 K: never slower, backward up to -24 % (N = 50 K: 21.4 -> 16.3 us)

// row of accumulator element i for lane half kh (v_mfma_f32_32x32x2_f32 C/D layout)
__device__ __forceinline__ int acc_row(int i, int kh) { return (i & 3) + 8 * (i >> 2) + 4 * kh; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    // lanes without a source (or outside ROW_MASK) add 0
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// sum over each half (32 lanes) of the wave; valid in lanes 31 and 63
__device__ __forceinline__ float half_sum(float v)
{
    v = dpp_add<0x111, 0xf>(v);       // row_shr:1
    v = dpp_add<0x112, 0xf>(v);       // row_shr:2
    v = dpp_add<0x114, 0xf>(v);       // row_shr:4
    v = dpp_add<0x118, 0xf>(v);       // row_shr:8   -> lane 15 of every row of 16 holds the row's sum
    v = dpp_add<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    return v;
}

// OUT = false: only r = relu(h) and y = r W1^T + b1 (the dynamic-temporal model's head, whose loss works on edges).
template <int C, bool OUT = true>
__global__ __launch_bounds__(kBlock) void head_fwd_kernel(
    const float *__restrict__ h, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ target,
    float *__restrict__ r_out, float *__restrict__ y, float *__restrict__ y_out, float *__restrict__ partial,
    int64_t N, int num_tiles)
{
    constexpr int KB = C / 8;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (tile >= num_tiles) return;                                   // whole wave
    const int l31 = lane & 31, kh = lane >> 5;
    const int64_t row = (int64_t)tile * 32 + l31;
    const bool rok = row < N;

    f32x16 acc;
    {
        const float bias = b1[l31];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias;
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int k0 = kb * 8 + kh * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok) {
            a = *reinterpret_cast<const float4 *>(h + row * C + k0);
            a.x = a.x < 0.f ? 0.f : a.x;
            a.y = a.y < 0.f ? 0.f : a.y;
            a.z = a.z < 0.f ? 0.f : a.z;
            a.w = a.w < 0.f ? 0.f : a.w;
            *reinterpret_cast<float4 *>(r_out + row * C + k0) = a;
        }
        const float4 b = *reinterpret_cast<const float4 *>(W1 + l31 * C + k0);      // B[k][n] = W1[n][k]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }

    if constexpr (!OUT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t orow = (int64_t)tile * 32 + acc_row(i, kh);
            if (orow < N) y[orow * kHeadF + l31] = acc[i];
        }
        return;
    }
    const float w2 = W2[l31], bias2 = b2[0];
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t orow = (int64_t)tile * 32 + acc_row(i, kh);
        const bool ok = orow < N;
        if (ok) y[orow * kHeadF + l31] = acc[i];
        const float s = half_sum(acc[i] * w2);                       // every lane takes part
        if (l31 == 31 && ok) {
            const float yo = s + bias2;
            y_out[orow] = yo;
            const float d = yo - target[orow];
            lsum = lsum + d * d;
        }
    }
    const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lsum), 31)) +
                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lsum), 63));
    if (lane == 0) partial[tile] = tot;
}

// ---- 16-row tiles (v_mfma_f32_16x16x4_f32): twice the waves at half the work each -- these kernels are one tile per
// wave and latency bound, so more waves per CU hide more of it.  lane = (n16 = lane & 15, kq = lane >> 4); A piece at
// column 16 j + 4 kq; accumulator element i of 16-column block ct: row 4 kq + i, column 16 ct + n16.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sum over each DPP row of 16 lanes; valid in lanes 15, 31, 47, 63
__device__ __forceinline__ float row16_sum(float v)
{
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}

template <int C, bool OUT = true>
__global__ __launch_bounds__(kBlock) void head_fwd16_kernel(
    const float *__restrict__ h, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ target,
    float *__restrict__ r_out, float *__restrict__ y, float *__restrict__ y_out, float *__restrict__ partial,
    int64_t N, int num_tiles)
{
    constexpr int KQ = C / 16, CT = kHeadF / 16;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (tile >= num_tiles) return;                                   // whole wave
    const int n16 = lane & 15, kq = lane >> 4;
    const int64_t row = (int64_t)tile * 16 + n16;
    const bool rok = row < N;

    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float bias = b1[ct * 16 + n16];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ct][i] = bias;
    }
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int k0 = 16 * j + 4 * kq;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok) {
            a = *reinterpret_cast<const float4 *>(h + row * C + k0);
            a.x = a.x < 0.f ? 0.f : a.x;
            a.y = a.y < 0.f ? 0.f : a.y;
            a.z = a.z < 0.f ? 0.f : a.z;
            a.w = a.w < 0.f ? 0.f : a.w;
            *reinterpret_cast<float4 *>(r_out + row * C + k0) = a;
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float4 b = *reinterpret_cast<const float4 *>(W1 + (ct * 16 + n16) * C + k0);   // B[k][n] = W1[n][k]
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc[ct], 0, 0, 0);
        }
    }

    float w2[CT], bias2 = 0.f, lsum = 0.f;
    if constexpr (OUT) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w2[ct] = W2[ct * 16 + n16];
        bias2 = b2[0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t orow = (int64_t)tile * 16 + 4 * kq + i;
        const bool ok = orow < N;
        if (ok) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) y[orow * kHeadF + ct * 16 + n16] = acc[ct][i];
        }
        if constexpr (OUT) {
            float p = 0.f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) p = p + acc[ct][i] * w2[ct];
            const float s = row16_sum(p);                            // every lane takes part
            if (n16 == 15 && ok) {
                const float yo = s + bias2;
                y_out[orow] = yo;
                const float d = yo - target[orow];
                lsum = lsum + d * d;
            }
        }
    }
    if constexpr (OUT) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) tot = tot + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lsum), 16 * q + 15));
        if (lane == 0) partial[tile] = tot;
    }
}

template <int C>
__global__ __launch_bounds__(kBlock) void head_bwd16_kernel(
    const float *__restrict__ g_loss, const float *__restrict__ g_y, const float *__restrict__ g_yout,
    const float *__restrict__ h, const float *__restrict__ y_out, const float *__restrict__ target,
    const float *__restrict__ W1, const float *__restrict__ W2, float *__restrict__ dh, float *__restrict__ dyt,
    float *__restrict__ dyo, int64_t N, float two_over_n, int num_tiles)
{
    constexpr int KQ = kHeadF / 16, CT = C / 16;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (tile >= num_tiles) return;                                   // whole wave
    const int n16 = lane & 15, kq = lane >> 4;
    const int64_t row = (int64_t)tile * 16 + n16;
    const bool rok = row < N;

    float d = 0.f;
    if (rok) {
        if (g_loss) d = two_over_n * g_loss[0] * (y_out[row] - target[row]);
        if (g_yout) d = d + g_yout[row];
        if (kq == 0 && dyo) dyo[row] = d;
    }

    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ct][i] = 0.f;

#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int k0 = 16 * j + 4 * kq;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_loss || g_yout) {                                     // (wave-uniform) the y_out branch of the head
            const float4 w = *reinterpret_cast<const float4 *>(W2 + k0);
            a = make_float4(d * w.x, d * w.y, d * w.z, d * w.w);
        }
        if (rok) {
            if (g_y) {
                const float4 g = *reinterpret_cast<const float4 *>(g_y + row * kHeadF + k0);
                a = make_float4(g.x + a.x, g.y + a.y, g.z + a.z, g.w + a.w);
            }
            *reinterpret_cast<float4 *>(dyt + row * kHeadF + k0) = a;
        } else {
            a = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float b = W1[(k0 + i) * C + ct * 16 + n16];   // B[k][n] = W1[k][n]
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b, acc[ct], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t orow = (int64_t)tile * 16 + 4 * kq + i;
            if (orow < N) {
                const int64_t at = orow * C + ct * 16 + n16;
                dh[at] = h[at] <= 0.f ? 0.f : acc[ct][i];           // threshold_backward
            }
        }
    }
}

// loss = (sum of the tile partials, in a fixed order) / N
__global__ __launch_bounds__(kBlock) void head_loss_kernel(const float *__restrict__ partial, int num_tiles,
                                                           float inv_n, float *__restrict__ loss,
                                                           const float *__restrict__ loss_in = nullptr)
{
    __shared__ float s[kBlock];
    float v = 0.f;
    for (int t = threadIdx.x; t < num_tiles; t += kBlock) v = v + partial[t];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + off];
        __syncthreads();
    }
    // loss_in: the running cost of the training loop (`cost = cost + mean(...)`): added here instead of by a launch
    if (threadIdx.x == 0) loss[0] = loss_in ? loss_in[0] + s[0] * inv_n : s[0] * inv_n;
}

template <int C>
__global__ __launch_bounds__(kBlock) void head_bwd_kernel(
    const float *__restrict__ g_loss, const float *__restrict__ g_y, const float *__restrict__ g_yout,
    const float *__restrict__ h, const float *__restrict__ y_out, const float *__restrict__ target,
    const float *__restrict__ W1, const float *__restrict__ W2, float *__restrict__ dh, float *__restrict__ dyt,
    float *__restrict__ dyo, int64_t N, float two_over_n, int num_tiles)
{
    constexpr int KB = kHeadF / 8, CT = C / 32;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (tile >= num_tiles) return;                                   // whole wave
    const int l31 = lane & 31, kh = lane >> 5;
    const int64_t row = (int64_t)tile * 32 + l31;
    const bool rok = row < N;

    float d = 0.f;
    if (rok) {
        if (g_loss) d = two_over_n * g_loss[0] * (y_out[row] - target[row]);
        if (g_yout) d = d + g_yout[row];
        if (kh == 0 && dyo) dyo[row] = d;
    }

    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;

#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int k0 = kb * 8 + kh * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_loss || g_yout) {                                     // (wave-uniform) the y_out branch of the head
            const float4 w = *reinterpret_cast<const float4 *>(W2 + k0);
            a = make_float4(d * w.x, d * w.y, d * w.z, d * w.w);
        }
        if (rok) {
            if (g_y) {
                const float4 g = *reinterpret_cast<const float4 *>(g_y + row * kHeadF + k0);
                a = make_float4(g.x + a.x, g.y + a.y, g.z + a.z, g.w + a.w);
            }
            *reinterpret_cast<float4 *>(dyt + row * kHeadF + k0) = a;
        } else {
            a = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float b = W1[(k0 + i) * C + ct * 32 + l31];   // B[k][n] = W1[k][n]
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b, acc[ct], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t orow = (int64_t)tile * 32 + acc_row(i, kh);
            if (orow < N) {
                const int64_t at = orow * C + ct * 32 + l31;
                dh[at] = h[at] <= 0.f ? 0.f : acc[ct][i];           // threshold_backward
            }
        }
    }
}

// ---- link-prediction head of the dynamic-temporal harness (dynamic-temporal-tgcn/seastar/model.py:5-21, train.py:
// decode = (z[src] * z[dst]).sum(-1), BCEWithLogitsLoss) -----------------------------------------------------------
constexpr int kLinkLanes = 8;                 // lanes per edge / node: 8 x 16 bytes = one row of y

__device__ __forceinline__ float group8_sum(float v)
{
    v += __shfl_xor(v, 1, kLinkLanes);
    v += __shfl_xor(v, 2, kLinkLanes);
    v += __shfl_xor(v, 4, kLinkLanes);
    return v;
}

// logits[e] = <y[src[e]], y[dst[e]]>; partial[block] = sum of the stable BCE-with-logits terms of the block's edges
__global__ __launch_bounds__(kBlock) void link_decode_bce_kernel(const float *__restrict__ y, const int64_t *__restrict__ src,
                                                                 const int64_t *__restrict__ dst,
                                                                 const float *__restrict__ target,
                                                                 float *__restrict__ logits, float *__restrict__ partial,
                                                                 int64_t M)
{
    __shared__ float s[kBlock / kLinkLanes];
    const int g = threadIdx.x / kLinkLanes, j = threadIdx.x % kLinkLanes;
    const int64_t e = (int64_t)blockIdx.x * (kBlock / kLinkLanes) + g;
    float term = 0.f;
    if (e < M) {
        const float4 a = *reinterpret_cast<const float4 *>(y + src[e] * kHeadF + j * 4);
        const float4 b = *reinterpret_cast<const float4 *>(y + dst[e] * kHeadF + j * 4);
        float d = a.x * b.x;
        d = d + a.y * b.y;
        d = d + a.z * b.z;
        d = d + a.w * b.w;
        const float x = group8_sum(d);
        if (j == 0) {
            logits[e] = x;
            term = fmaxf(x, 0.f) - x * target[e] + log1pf(__expf(-fabsf(x)));
        }
    } else {
        (void)group8_sum(0.f);
    }
    if (j == 0) s[g] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < kBlock / kLinkLanes; ++i) t = t + s[i];
        partial[blockIdx.x] = t;
    }
}

// The same decode for every snapshot of a BPTT window in ONE launch (blockIdx.y = snapshot): the loss of a snapshot feeds nothing in
// the next one, so the window's decodes can all run behind its last forward step (19 launches of ~5 us less per 20-snapshot window).
constexpr int kDecodeJobs = 32;
struct DecodeJobs {
    const float *y[kDecodeJobs], *target[kDecodeJobs];
    const int64_t *edges[kDecodeJobs];
    float *logits[kDecodeJobs], *partial[kDecodeJobs];
};
__global__ __launch_bounds__(kBlock) void link_decode_bce_multi_kernel(const DecodeJobs jobs, int64_t M)
{
    __shared__ float s[kBlock / kLinkLanes];
    const int t = blockIdx.y;
    const float *__restrict__ y = jobs.y[t];
    const int64_t *__restrict__ src = jobs.edges[t], *__restrict__ dst = jobs.edges[t] + M;
    const int g = threadIdx.x / kLinkLanes, j = threadIdx.x % kLinkLanes;
    const int64_t e = (int64_t)blockIdx.x * (kBlock / kLinkLanes) + g;
    float term = 0.f;
    if (e < M) {                                                   // (arithmetic and order of link_decode_bce_kernel)
        const float4 a = *reinterpret_cast<const float4 *>(y + src[e] * kHeadF + j * 4);
        const float4 b = *reinterpret_cast<const float4 *>(y + dst[e] * kHeadF + j * 4);
        float d = a.x * b.x;
        d = d + a.y * b.y;
        d = d + a.z * b.z;
        d = d + a.w * b.w;
        const float x = group8_sum(d);
        if (j == 0) {
            jobs.logits[t][e] = x;
            term = fmaxf(x, 0.f) - x * jobs.target[t][e] + log1pf(__expf(-fabsf(x)));
        }
    } else {
        (void)group8_sum(0.f);
    }
    if (j == 0) s[g] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tt = 0.f;
        for (int i = 0; i < kBlock / kLinkLanes; ++i) tt = tt + s[i];
        jobs.partial[t][blockIdx.x] = tt;
    }
}

// dy[v] = g_y[v] + sum over the label edges incident to v, in the order of the node-sorted incidence list, of
// (sigmoid(logit) - target) * scale * y[other endpoint]: no atomics, so the gradient is reproducible
__global__ __launch_bounds__(kBlock) void link_bwd_nodes_kernel(const float *__restrict__ g_loss, const float *__restrict__ g_y,
                                                                const float *__restrict__ y, const float *__restrict__ logits,
                                                                const float *__restrict__ target,
                                                                const int *__restrict__ row_ptr, const int *__restrict__ other,
                                                                const int *__restrict__ eid, float *__restrict__ dy, int64_t N,
                                                                float inv_m)
{
    const int g = threadIdx.x / kLinkLanes, j = threadIdx.x % kLinkLanes;
    const int64_t v = (int64_t)blockIdx.x * (kBlock / kLinkLanes) + g;
    if (v >= N) return;
    float4 acc = g_y ? *reinterpret_cast<const float4 *>(g_y + v * kHeadF + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_loss) {
        const float scale = g_loss[0] * inv_m;
        for (int k = row_ptr[v]; k < row_ptr[v + 1]; ++k) {
            const int e = eid[k];
            const float x = logits[e];
            const float sig = 1.0f / (1.0f + __expf(-x));
            const float coef = (sig - target[e]) * scale;
            const float4 o = *reinterpret_cast<const float4 *>(y + (int64_t)other[k] * kHeadF + j * 4);
            acc.x = acc.x + coef * o.x;
            acc.y = acc.y + coef * o.y;
            acc.z = acc.z + coef * o.z;
            acc.w = acc.w + coef * o.w;
        }
    }
    *reinterpret_cast<float4 *>(dy + v * kHeadF + j * 4) = acc;
}

inline int head_tiles(int64_t N) { return (int)((N + 31) / 32); }
inline int head_tiles16(int64_t N) { return (int)((N + 15) / 16); }
// 16-row tiles from this many rows on ("cell_rows" forces either): measured in tools/microbench_head.py
inline bool head_rows16(int64_t N) { return tuning().cell_rows == 16 || (tuning().cell_rows == 0 && N >= kHead16MinRows); }


}  // namespace
}  // namespace stg

extern "C" int stg_tgcn_head_supported(int32_t C, int32_t F, int32_t O)
{
    return (C == 32 || C == 64 || C == 128) && F == stg::kHeadF && O == 1;
}

extern "C" size_t stg_tgcn_head_workspace_bytes(int64_t N)
{
    return N <= 0 ? 0 : sizeof(float) * (size_t)stg::head_tiles16(N);
}

extern "C" int stg_tgcn_head_fwd_acc(const float *h, const float *W1, const float *b1, const float *W2, const float *b2,
                                     const float *target, const float *loss_in, float *r, float *y, float *y_out,
                                     float *loss, int64_t N, int32_t C, int32_t F, void *workspace,
                                     size_t workspace_bytes, void *stream_);

extern "C" int stg_tgcn_head_fwd(const float *h, const float *W1, const float *b1, const float *W2, const float *b2,
                                 const float *target, float *r, float *y, float *y_out, float *loss, int64_t N,
                                 int32_t C, int32_t F, void *workspace, size_t workspace_bytes, void *stream_)
{
    return stg_tgcn_head_fwd_acc(h, W1, b1, W2, b2, target, nullptr, r, y, y_out, loss, N, C, F, workspace, workspace_bytes,
                                 stream_);
}

extern "C" int stg_tgcn_head_fwd_acc(const float *h, const float *W1, const float *b1, const float *W2, const float *b2,
                                     const float *target, const float *loss_in, float *r, float *y, float *y_out,
                                     float *loss, int64_t N, int32_t C, int32_t F, void *workspace,
                                     size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    if (!stg_tgcn_head_supported(C, F, 1))
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_head_fwd: C=%d F=%d not supported", C, F);
    if (N < 0 || N > (int64_t)32 * 0x3fffffff) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_fwd: bad N");
    if (!loss) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_fwd: NULL pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N == 0) {
        hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(kBlock), 0, stream, nullptr, 0, __builtin_nanf(""), loss);   // mean of nothing
        return check_launch("stg_tgcn_head_fwd");
    }
    if (!h || !W1 || !b1 || !W2 || !b2 || !target || !r || !y || !y_out || !workspace)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_fwd: NULL pointer argument");
    if (workspace_bytes < stg_tgcn_head_workspace_bytes(N))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_fwd: workspace too small");
    const bool r16 = head_rows16(N);
    const int tiles = r16 ? head_tiles16(N) : head_tiles(N);
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    float *partial = static_cast<float *>(workspace);
#define STG_HEAD_FWD(CC)                                                                                               \
    if (r16)                                                                                                           \
        hipLaunchKernelGGL((head_fwd16_kernel<CC, true>), dim3(blocks), dim3(kBlock), 0, stream, h, W1, b1, W2, b2,    \
                           target, r, y, y_out, partial, N, tiles);                                                    \
    else                                                                                                               \
        hipLaunchKernelGGL((head_fwd_kernel<CC, true>), dim3(blocks), dim3(kBlock), 0, stream, h, W1, b1, W2, b2,      \
                           target, r, y, y_out, partial, N, tiles)
    switch (C) {
        case 32: STG_HEAD_FWD(32); break;
        case 64: STG_HEAD_FWD(64); break;
        default: STG_HEAD_FWD(128); break;
    }
#undef STG_HEAD_FWD
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(kBlock), 0, stream, partial, tiles, 1.0f / (float)N, loss, loss_in);
    return check_launch("stg_tgcn_head_fwd");
}

extern "C" int stg_tgcn_head_bwd(const float *g_loss, const float *g_y, const float *g_yout, const float *h,
                                 const float *y_out, const float *target, const float *W1, const float *W2, float *dh,
                                 float *dyt, float *dyo, int64_t N, int32_t C, int32_t F, void *stream_)
{
    using namespace stg;
    if (!stg_tgcn_head_supported(C, F, 1))
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_head_bwd: C=%d F=%d not supported", C, F);
    if (N < 0 || N > (int64_t)32 * 0x3fffffff) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_bwd: bad N");
    if (N == 0) return 0;
    if (!h || !y_out || !target || !W1 || !W2 || !dh || !dyt || !dyo)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_head_bwd: NULL pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool r16 = head_rows16(N);
    const int tiles = r16 ? head_tiles16(N) : head_tiles(N);
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const float two_over_n = 2.0f / (float)N;
#define STG_HEAD_BWD(CC)                                                                                               \
    if (r16)                                                                                                           \
        hipLaunchKernelGGL(head_bwd16_kernel<CC>, dim3(blocks), dim3(kBlock), 0, stream, g_loss, g_y, g_yout, h, y_out, \
                           target, W1, W2, dh, dyt, dyo, N, two_over_n, tiles);                                        \
    else                                                                                                               \
        hipLaunchKernelGGL(head_bwd_kernel<CC>, dim3(blocks), dim3(kBlock), 0, stream, g_loss, g_y, g_yout, h, y_out,  \
                           target, W1, W2, dh, dyt, dyo, N, two_over_n, tiles)
    switch (C) {
        case 32: STG_HEAD_BWD(32); break;
        case 64: STG_HEAD_BWD(64); break;
        default: STG_HEAD_BWD(128); break;
    }
#undef STG_HEAD_BWD
    return check_launch("stg_tgcn_head_bwd");
}

extern "C" int stg_link_head_supported(int32_t C, int32_t F) { return (C == 32 || C == 64 || C == 128) && F == stg::kHeadF; }

extern "C" size_t stg_link_head_workspace_bytes(int64_t M)
{
    return M <= 0 ? sizeof(float) : sizeof(float) * (size_t)((M + 31) / 32);
}

extern "C" int stg_link_head_fwd(const float *h, const float *W1, const float *b1, const int64_t *src, const int64_t *dst,
                                 const float *target, const float *loss_in, float *r, float *y, float *logits,
                                 float *loss, int64_t N, int64_t M, int32_t C, int32_t F, void *workspace,
                                 size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    if (!stg_link_head_supported(C, F)) return fail(STG_ERR_UNSUPPORTED, "stg_link_head_fwd: C=%d F=%d not supported", C, F);
    if (N <= 0 || M <= 0 || N > (int64_t)32 * 0x3fffffff || M > (int64_t)32 * 0x3fffffff)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_head_fwd: bad N / M");
    if (!h || !W1 || !b1 || !src || !dst || !target || !r || !y || !logits || !loss || !workspace)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_head_fwd: NULL pointer argument");
    if (workspace_bytes < stg_link_head_workspace_bytes(M))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_head_fwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool r16 = head_rows16(N);
    const int tiles = r16 ? head_tiles16(N) : head_tiles(N);
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    float *partial = static_cast<float *>(workspace);
#define STG_LINK_FWD(CC)                                                                                               \
    if (r16)                                                                                                           \
        hipLaunchKernelGGL((head_fwd16_kernel<CC, false>), dim3(blocks), dim3(kBlock), 0, stream, h, W1, b1, nullptr,  \
                           nullptr, nullptr, r, y, nullptr, nullptr, N, tiles);                                        \
    else                                                                                                               \
        hipLaunchKernelGGL((head_fwd_kernel<CC, false>), dim3(blocks), dim3(kBlock), 0, stream, h, W1, b1, nullptr,    \
                           nullptr, nullptr, r, y, nullptr, nullptr, N, tiles)
    switch (C) {
        case 32: STG_LINK_FWD(32); break;
        case 64: STG_LINK_FWD(64); break;
        default: STG_LINK_FWD(128); break;
    }
#undef STG_LINK_FWD
    const int eblocks = (int)((M + 31) / 32);
    hipLaunchKernelGGL(link_decode_bce_kernel, dim3(eblocks), dim3(kBlock), 0, stream, y, src, dst, target, logits, partial, M);
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(kBlock), 0, stream, partial, eblocks, 1.0f / (float)M, loss, loss_in);
    return check_launch("stg_link_head_fwd");
}

extern "C" int stg_link_head_bwd(const float *g_loss, const float *g_y, const float *h, const float *y,
                                 const float *logits, const float *target, const int32_t *row_ptr, const int32_t *other,
                                 const int32_t *eid, const float *W1, float *dy, float *dh, float *dyt, int64_t N,
                                 int64_t M, int32_t C, int32_t F, void *stream_)
{
    using namespace stg;
    if (!stg_link_head_supported(C, F)) return fail(STG_ERR_UNSUPPORTED, "stg_link_head_bwd: C=%d F=%d not supported", C, F);
    if (N <= 0 || M <= 0 || N > (int64_t)32 * 0x3fffffff) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_head_bwd: bad N / M");
    if (!h || !y || !logits || !target || !row_ptr || !other || !eid || !W1 || !dy || !dh || !dyt)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_head_bwd: NULL pointer argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(link_bwd_nodes_kernel, dim3((unsigned)((N + 31) / 32)), dim3(kBlock), 0, stream, g_loss, g_y, y, logits,
                       target, row_ptr, other, eid, dy, N, 1.0f / (float)M);
    // dh = (h > 0) (dy W1), dyt = dy: the relu -> Linear backward is head_bwd_kernel without its y_out branch (W2 is
    // not read then)
    const bool r16 = head_rows16(N);
    const int tiles = r16 ? head_tiles16(N) : head_tiles(N);
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
#define STG_LINK_BWD(CC)                                                                                               \
    if (r16)                                                                                                           \
        hipLaunchKernelGGL(head_bwd16_kernel<CC>, dim3(blocks), dim3(kBlock), 0, stream, nullptr, dy, nullptr, h,      \
                           nullptr, nullptr, W1, nullptr, dh, dyt, nullptr, N, 0.f, tiles);                            \
    else                                                                                                               \
        hipLaunchKernelGGL(head_bwd_kernel<CC>, dim3(blocks), dim3(kBlock), 0, stream, nullptr, dy, nullptr, h, nullptr, \
                           nullptr, W1, nullptr, dh, dyt, nullptr, N, 0.f, tiles)
    switch (C) {
        case 32: STG_LINK_BWD(32); break;
        case 64: STG_LINK_BWD(64); break;
        default: STG_LINK_BWD(128); break;
    }
#undef STG_LINK_BWD
    return check_launch("stg_link_head_bwd");
}

// The decoder + loss and its node-side backward on their own (the relu -> Linear half of the link head runs inside the
// one-launch TGCN step, csrc/tgcn_step.hpp, when a whole window is one autograd node).
extern "C" int stg_link_decode_fwd(const float *y, const int64_t *edge_index, const float *target, float *logits,
                                   float *partial, int64_t M, int32_t F, void *stream_)
{
    using namespace stg;
    if (F != kHeadF) return fail(STG_ERR_UNSUPPORTED, "stg_link_decode_fwd: F=%d not supported (32)", F);
    if (M <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_fwd: bad M");
    if (!y || !edge_index || !target || !logits || !partial) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_fwd: NULL pointer argument");
    const int eblocks = (int)((M + 31) / 32);
    hipLaunchKernelGGL(link_decode_bce_kernel, dim3(eblocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream_), y, edge_index,
                       edge_index + M, target, logits, partial, M);
    return check_launch("stg_link_decode_fwd");
}

extern "C" int stg_link_decode_fwd_multi(int32_t count, const float *const *y, const int64_t *const *edge_index,
                                         const float *const *target, float *const *logits, float *const *partial, int64_t M,
                                         int32_t F, void *stream_)
{
    using namespace stg;
    if (F != kHeadF) return fail(STG_ERR_UNSUPPORTED, "stg_link_decode_fwd_multi: F=%d not supported (32)", F);
    if (M <= 0 || count <= 0 || count > kDecodeJobs) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_fwd_multi: bad M / count (1 .. %d)", kDecodeJobs);
    if (!y || !edge_index || !target || !logits || !partial) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_fwd_multi: NULL pointer argument");
    DecodeJobs jobs{};
    for (int t = 0; t < count; ++t) {
        if (!y[t] || !edge_index[t] || !target[t] || !logits[t] || !partial[t])
            return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_fwd_multi: NULL pointer in snapshot %d", t);
        jobs.y[t] = y[t]; jobs.edges[t] = edge_index[t]; jobs.target[t] = target[t]; jobs.logits[t] = logits[t]; jobs.partial[t] = partial[t];
    }
    const int eblocks = (int)((M + 31) / 32);
    hipLaunchKernelGGL(link_decode_bce_multi_kernel, dim3((unsigned)eblocks, (unsigned)count), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream_), jobs, M);
    return check_launch("stg_link_decode_fwd_multi");
}

extern "C" int stg_link_decode_bwd(const float *g_loss, const float *y, const float *logits, const float *target,
                                   const int32_t *row_ptr, const int32_t *other, const int32_t *eid, float *dy, int64_t N,
                                   int64_t M, int32_t F, void *stream_)
{
    using namespace stg;
    if (F != kHeadF) return fail(STG_ERR_UNSUPPORTED, "stg_link_decode_bwd: F=%d not supported (32)", F);
    if (N <= 0 || M <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_bwd: bad N / M");
    if (!g_loss || !y || !logits || !target || !row_ptr || !other || !eid || !dy)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_link_decode_bwd: NULL pointer argument");
    hipLaunchKernelGGL(link_bwd_nodes_kernel, dim3((unsigned)((N + 31) / 32)), dim3(kBlock), 0, static_cast<hipStream_t>(stream_),
                       g_loss, nullptr, y, logits, target, row_ptr, other, eid, dy, N, 1.0f / (float)M);
    return check_launch("stg_link_decode_bwd");
}
