// Softmax cross-entropy (mean over rows) of the GCN training scripts as one launch each way.
//
// Reference: benchmarking/gcn/seastar/train.py:63-101 -- `nn.CrossEntropyLoss()(logits[train_mask], labels[train_mask])`.
// In torch that is log_softmax + nll_loss forward and their two backward kernels plus fills; on ROCm the fused 'mean'
// reduction of nll_loss runs in ONE workgroup (10 us on 1624 rows of Cora, 1.3 ms on 600 K rows).  Here:
//   forward : G lanes per row (8 up to 32 classes, 32 up to 256, else 64; 16-byte loads when K % 4 == 0): row max,
//             sum of exp, lse = max + log(sum); loss_row = lse - logits[label]; rows dealt grid-stride to at most
//             2048 workgroups, whose partial sums and counts of counted rows a finish kernel adds in a fixed order;
//             loss = sum / count; lse [n] and the count kept for the backward
//   backward: dlogits[i, c] = (exp(logits[i, c] - lse[i]) - [c == label[i]]) * g / count, zero for a row not counted
//             (by rows like the forward when K % 4 == 0: 16-byte accesses, label and lse once per row)
// Rows whose label is nn.CrossEntropyLoss's default ignore_index (-100) are not counted (no loss term, no gradient,
// not in the mean's denominator), as in torch.  Any other label outside [0, K) -- torch raises a device assert --
// is treated the same way and reported through bit 0 of the status word.
// expf / logf are the accurate library versions: results agree with torch to fp32 rounding (tests: 1e-6 relative).
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

constexpr int64_t kIgnoreIndex = -100;                           // nn.CrossEntropyLoss() default

template <int G>
__device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, G));
    return v;
}

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = v + __shfl_xor(v, off, G);
    return v;
}

// G lanes per row, rows dealt to the lane groups grid-stride (at most kXentGrid workgroups: the partial sums a
// single workgroup adds up afterwards stay few); VEC4: K % 4 == 0, 16-byte loads.
template <int G, bool VEC4>
__global__ __launch_bounds__(kBlock) void xent_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                          float *__restrict__ lse, float *__restrict__ partial,
                                                          int *__restrict__ partial_cnt, int64_t n, int K,
                                                          int *__restrict__ status)
{
    constexpr int ROWS = kBlock / G;
    __shared__ float s[ROWS];
    __shared__ int sc[ROWS];
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    float term = 0.f;                                            // this lane group's rows, in row order
    int counted = 0;
    for (int64_t row = (int64_t)blockIdx.x * ROWS + g; row < n; row += (int64_t)gridDim.x * ROWS) {
        const float *x = logits + row * K;
        float m = -INFINITY, sum = 0.f;
        if constexpr (VEC4) {
            for (int c = j * 4; c < K; c += G * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
            m = group_max<G>(m);
            for (int c = j * 4; c < K; c += G * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                sum = sum + expf(v.x - m);
                sum = sum + expf(v.y - m);
                sum = sum + expf(v.z - m);
                sum = sum + expf(v.w - m);
            }
        } else {
            for (int c = j; c < K; c += G) m = fmaxf(m, x[c]);
            m = group_max<G>(m);
            for (int c = j; c < K; c += G) sum = sum + expf(x[c] - m);
        }
        sum = group_sum<G>(sum);
        if (j == 0) {
            const float l = m + logf(sum);
            lse[row] = l;
            const int64_t t = labels[row];
            if (t >= 0 && t < K) term = term + (l - x[t]), ++counted;
            else if (t != kIgnoreIndex) atomicOr(status, 1);      // neither a class nor ignore_index: reported, not counted
        }
    }
    if (j == 0) s[g] = term, sc[g] = counted;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int c = 0;
        for (int i = 0; i < ROWS; ++i) t = t + s[i], c += sc[i];
        partial[blockIdx.x] = t;
        partial_cnt[blockIdx.x] = c;
    }
}

// The same forward for K <= 4 G CH (K % 4 == 0): a lane keeps its CH float4s of the row in registers between the max pass and the
// exp pass (the plain form reads the row twice), and a lane group has TWO rows in flight -- with one row per trip the launch is a
// chain of load -> reduce -> load (108 us for 600 K x 128, 2.8 TB/s).  Same arithmetic, same order, same results.
template <int G, int CH>
__global__ __launch_bounds__(kBlock) void xent_fwd_reg_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                              float *__restrict__ lse, float *__restrict__ partial,
                                                              int *__restrict__ partial_cnt, int64_t n, int K,
                                                              int *__restrict__ status)
{
    constexpr int ROWS = kBlock / G, R = 2;                      // (four rows in flight: 100 us against 81)
    __shared__ float s[ROWS];
    __shared__ int sc[ROWS];
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    float term = 0.f;                                            // this lane group's rows, in row order
    int counted = 0;
    const int64_t stride = (int64_t)gridDim.x * ROWS;
    for (int64_t row0 = (int64_t)blockIdx.x * ROWS + g; row0 < n; row0 += R * stride) {
        float4 v[R][CH];
        int64_t lab[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * stride;
            lab[r] = (j == 0 && row < n) ? labels[row] : 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int col = (j + c * G) * 4;
                v[r][c] = (row < n && col < K) ? *reinterpret_cast<const float4 *>(logits + row * K + col)
                                               : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * stride;
            if (row >= n) break;                                 // uniform across the lane group
            float m = -INFINITY, sum = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) m = fmaxf(fmaxf(m, fmaxf(v[r][c].x, v[r][c].y)), fmaxf(v[r][c].z, v[r][c].w));
            m = group_max<G>(m);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((j + c * G) * 4 < K) {
                    sum = sum + expf(v[r][c].x - m);
                    sum = sum + expf(v[r][c].y - m);
                    sum = sum + expf(v[r][c].z - m);
                    sum = sum + expf(v[r][c].w - m);
                }
            }
            sum = group_sum<G>(sum);
            if (j == 0) {
                const float l = m + logf(sum);
                lse[row] = l;
                const int64_t t = lab[r];
                if (t >= 0 && t < K) term = term + (l - logits[row * K + t]), ++counted;
                else if (t != kIgnoreIndex) atomicOr(status, 1);  // neither a class nor ignore_index: reported, not counted
            }
        }
    }
    if (j == 0) s[g] = term, sc[g] = counted;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int c = 0;
        for (int i = 0; i < ROWS; ++i) t = t + s[i], c += sc[i];
        partial[blockIdx.x] = t;
        partial_cnt[blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(kBlock) void xent_finish_kernel(const float *__restrict__ partial,
                                                             const int *__restrict__ partial_cnt, int count,
                                                             float *__restrict__ loss, float *__restrict__ n_counted)
{
    __shared__ float s[kBlock];
    __shared__ long long c[kBlock];
    float v = 0.f;
    long long k = 0;
    for (int t = threadIdx.x; t < count; t += kBlock) v = v + partial[t], k += partial_cnt[t];
    s[threadIdx.x] = v;
    c[threadIdx.x] = k;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + off];
            c[threadIdx.x] += c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = (float)c[0];
        n_counted[0] = cnt;
        loss[0] = s[0] / cnt;                                    // no counted row: 0 / 0 = NaN, as torch
    }
}

// rows [n, n_total) of dlogits are zeroed: the loss was taken on a prefix of the logits matrix (the train mask of the
// GCN scripts), and the gradient of the whole matrix comes out of this one launch instead of a fill + a strided copy
__global__ __launch_bounds__(kBlock) void xent_bwd_kernel(const float *__restrict__ g_loss, const float *__restrict__ logits,
                                                          const int64_t *__restrict__ labels, const float *__restrict__ lse,
                                                          float *__restrict__ dlogits, int64_t n, int64_t n_total, int K,
                                                          const float *__restrict__ n_counted)
{
    const int64_t total = n_total * K, live = n * K;
    const float scale = g_loss[0] / n_counted[0];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        float v = 0.f;
        if (i < live) {
            const int64_t row = i / K;
            const int c = (int)(i - row * K);
            const int64_t t = labels[row];
            if (t >= 0 && t < K) {                               // rows not counted by the forward get no gradient
                const float p = expf(logits[i] - lse[row]);
                v = (p - (t == c ? 1.f : 0.f)) * scale;
            }
        }
        dlogits[i] = v;
    }
}

// The same gradient by rows (K % 4 == 0): G lanes per row, 16-byte loads and stores, the row's label and lse read once per row,
// two rows in flight.  The element form above spends a 64-bit division and two scalar gathers per ELEMENT: 270 us for
// [1 M, 128] (600 K live rows), whose 0.82 GB take ~140 us.
// COLSUM: also partial[block][K] = the column sums of the rows this workgroup wrote (K <= 8 G), added up by
// xent_colsum_finish_kernel: the gradient's column sums are the bias gradient of the layer that produced the logits, which
// otherwise re-reads the whole matrix for them (120-150 us at [1 M, 128]).
template <int G, bool COLSUM>
__global__ __launch_bounds__(kBlock) void xent_bwd_rows_kernel(const float *__restrict__ g_loss, const float *__restrict__ logits,
                                                               const int64_t *__restrict__ labels, const float *__restrict__ lse,
                                                               float *__restrict__ dlogits, int64_t n, int64_t n_total, int K,
                                                               const float *__restrict__ n_counted, float *__restrict__ partial)
{
    constexpr int ROWS = kBlock / G, R = 2;
    __shared__ float4 red[COLSUM ? kBlock * 2 : 1];
    float4 cs[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    const float scale = g_loss[0] / n_counted[0];
    const int64_t stride = (int64_t)gridDim.x * ROWS;
    for (int64_t row0 = (int64_t)blockIdx.x * ROWS + g; row0 < n_total; row0 += R * stride) {
        int64_t lab[R];
        float l[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * stride;
            const bool live = row < n;
            lab[r] = live ? labels[row] : -1;
            l[r] = live ? lse[row] : 0.f;
        }
        for (int col = j * 4; col < K; col += G * 4) {
            float4 x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r * stride;
                x[r] = (row < n && lab[r] >= 0 && lab[r] < K) ? *reinterpret_cast<const float4 *>(logits + row * K + col)
                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r * stride;
                if (row >= n_total) break;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                const int64_t t = lab[r];
                if (row < n && t >= 0 && t < K) {                   // rows not counted by the forward get no gradient
                    const int tc = (int)t - col;
                    o.x = (expf(x[r].x - l[r]) - (tc == 0 ? 1.f : 0.f)) * scale;
                    o.y = (expf(x[r].y - l[r]) - (tc == 1 ? 1.f : 0.f)) * scale;
                    o.z = (expf(x[r].z - l[r]) - (tc == 2 ? 1.f : 0.f)) * scale;
                    o.w = (expf(x[r].w - l[r]) - (tc == 3 ? 1.f : 0.f)) * scale;
                }
                *reinterpret_cast<float4 *>(dlogits + row * K + col) = o;
                if constexpr (COLSUM) {
                    float4 &c = cs[col >= G * 4 ? 1 : 0];
                    c = make_float4(c.x + o.x, c.y + o.y, c.z + o.z, c.w + o.w);
                }
            }
        }
    }
    if constexpr (COLSUM) {
        red[threadIdx.x] = cs[0];
        red[kBlock + threadIdx.x] = cs[1];
        __syncthreads();
        // thread (chunk, j) of the first 2 G: the ROWS lane groups' sums of column piece 4 j + 4 G chunk, in group order
        if ((int)threadIdx.x < 2 * G) {
            const int chunk = threadIdx.x / G, jj = threadIdx.x % G, col = (jj + chunk * G) * 4;
            if (col < K) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int q = 0; q < ROWS; ++q) {
                    const float4 v = red[chunk * kBlock + q * G + jj];
                    t = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
                }
                *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * K + col) = t;
            }
        }
    }
}

// colsum[f] = sum_b partial[b][f] in a fixed order: 8 columns per workgroup, the partial rows dealt to 32 thread groups
__global__ __launch_bounds__(kBlock) void xent_colsum_finish_kernel(const float *__restrict__ partial, float *__restrict__ colsum,
                                                                   int blocks, int K)
{
    constexpr int kCols = 8, kGroups = kBlock / kCols;
    __shared__ float red[kGroups][kCols];
    const int c = threadIdx.x % kCols, grp = threadIdx.x / kCols;
    const int f = blockIdx.x * kCols + c;
    float s = 0.f;
    if (f < K) {
#pragma unroll 4
        for (int b = grp; b < blocks; b += kGroups) s += partial[(int64_t)b * K + f];
    }
    red[grp][c] = s;
    __syncthreads();
    if (grp == 0 && f < K) {
        float t = 0.f;
        for (int q = 0; q < kGroups; ++q) t += red[q][c];
        colsum[f] = t;
    }
}

// ---- forward and gradient in ONE pass over the logits ---------------------------------------------------------------------------
// The training step reads the logits twice (lse in the forward, softmax - onehot in the backward: 81 + 140 us at [1 M, 128]
// with 600 K live rows).  When the loss is going to be differentiated, the row is in registers once: lse, the loss term, and
// the gradient for an upstream gradient of 1 -- what `loss.backward()` passes -- written at once, column sums included.  Its
// scale 1 / (rows counted) must be known before the first row: a count launch over the labels alone (8 bytes per row) leaves
// partial counts that every workgroup adds up in the same order.  The backward then only multiplies by g when g != 1
// (xent_scale_grad_kernel: every workgroup reads g and leaves).  The launch is bound by its vector instructions (the accurate
// expf is ~20 of them): the softmax is exp(x - m) / sum from the exponentials of the sum, not a second expf per element.
constexpr int kXentCountParts = 512;

__global__ __launch_bounds__(kBlock) void xent_count_kernel(const int64_t *__restrict__ labels, int64_t n, int K, int *__restrict__ parts)
{
    __shared__ int s[kBlock];
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t t = labels[i];
        c += (t >= 0 && t < K) ? 1 : 0;
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = s[0];
}

template <int G, int CH>
__global__ __launch_bounds__(kBlock) void xent_fwd_grad_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                               float *__restrict__ lse, float *__restrict__ dlogits,
                                                               float *__restrict__ partial, int *__restrict__ partial_cnt,
                                                               float *__restrict__ cs_partial, const int *__restrict__ cnt_parts,
                                                               int n_parts, int64_t n, int64_t n_total, int K, int *__restrict__ status)
{
    constexpr int ROWS = kBlock / G, R = 2;                      // (four rows per trip: 159 us against 148 at [1 M, 128])
    __shared__ float s[ROWS];
    __shared__ int sc[ROWS];
    __shared__ int total_s;
    __shared__ float4 red[kBlock * 2];
    if (threadIdx.x < kWave) {
        int c = 0;
        for (int i = threadIdx.x; i < n_parts; i += kWave) c += cnt_parts[i];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
        if (threadIdx.x == 0) total_s = c;
    }
    __syncthreads();
    const float scale = 1.0f / (float)total_s;                   // = g / n_counted of the backward launch at g = 1
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    float term = 0.f;
    int counted = 0;
    float4 cs[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) cs[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t stride = (int64_t)gridDim.x * ROWS;
    // The NEXT trip's rows are asked for before this trip's are worked on: the chain load -> max -> exp -> sum -> log -> exp -> store
    // of a trip is long, and two rows per lane group in flight do not cover it.  For the loads to stay in flight across the trip
    // they sit behind no branch (a row past the live ones re-reads row n - 1, a column past K its last piece; the values are
    // replaced afterwards) and the row's target logit comes out of the registers (a group sum with one non-zero term) instead of
    // a load of its own, which drains the queue: 170 us at [1 M, 128] with the loads behind their bounds checks.
    auto load_rows = [&](int64_t row0, float4 (&v)[R][CH], int64_t (&lab)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = std::min<int64_t>(row0 + r * stride, n - 1);
            lab[r] = labels[row];
#pragma unroll
            for (int c = 0; c < CH; ++c) v[r][c] = *reinterpret_cast<const float4 *>(logits + row * K + std::min((j + c * G) * 4, K - 4));
        }
    };
    float4 vn[R][CH];
    int64_t labn[R];
    load_rows((int64_t)blockIdx.x * ROWS + g, vn, labn);
    for (int64_t row0 = (int64_t)blockIdx.x * ROWS + g; row0 < n; row0 += R * stride) {
        float4 v[R][CH];
        int64_t lab[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            lab[r] = labn[r];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                v[r][c] = (j + c * G) * 4 < K ? vn[r][c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
        load_rows(row0 + R * stride, vn, labn);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r * stride;
            const int64_t t = lab[r];
            const bool live = row < n, valid = live && t >= 0 && t < K;
            float m = -INFINITY, sum = 0.f, xt = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) m = fmaxf(fmaxf(m, fmaxf(v[r][c].x, v[r][c].y)), fmaxf(v[r][c].z, v[r][c].w));
            m = group_max<G>(m);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((j + c * G) * 4 < K) {
                    const int tc = (int)t - (j + c * G) * 4;      // the lane that holds column t contributes it, the others 0
                    xt += tc == 0 ? v[r][c].x : tc == 1 ? v[r][c].y : tc == 2 ? v[r][c].z : tc == 3 ? v[r][c].w : 0.f;
                    v[r][c] = make_float4(expf(v[r][c].x - m), expf(v[r][c].y - m), expf(v[r][c].z - m), expf(v[r][c].w - m));
                    sum = sum + v[r][c].x;                         // (the order of xent_fwd_reg_kernel)
                    sum = sum + v[r][c].y;
                    sum = sum + v[r][c].z;
                    sum = sum + v[r][c].w;
                }
            }
            sum = group_sum<G>(sum);
            xt = group_sum<G>(xt);
            const float l = m + logf(sum), inv = 1.0f / sum;
            if (j == 0 && live) {
                lse[row] = l;
                if (valid) term = term + (l - xt), ++counted;
                else if (t != kIgnoreIndex) atomicOr(status, 1);
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int col = (j + c * G) * 4;
                    if (col >= K) continue;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid) {                                 // rows not counted get no gradient
                        // softmax = exp(x - m) / sum from the exponentials the sum was made of (the backward launch, which has
                        // only lse, takes exp(x - lse): the same value up to the rounding of lse, ~1e-7 relative either way)
                        const int tc = (int)t - col;
                        o.x = (v[r][c].x * inv - (tc == 0 ? 1.f : 0.f)) * scale;
                        o.y = (v[r][c].y * inv - (tc == 1 ? 1.f : 0.f)) * scale;
                        o.z = (v[r][c].z * inv - (tc == 2 ? 1.f : 0.f)) * scale;
                        o.w = (v[r][c].w * inv - (tc == 3 ? 1.f : 0.f)) * scale;
                    }
                    *reinterpret_cast<float4 *>(dlogits + row * K + col) = o;
                    cs[c] = make_float4(cs[c].x + o.x, cs[c].y + o.y, cs[c].z + o.z, cs[c].w + o.w);
                }
            }
        }
    }
    // rows [n, n_total) took no part in the loss: zeros (the flat range dealt to all threads, 16 bytes each)
    {
        float4 *z = reinterpret_cast<float4 *>(dlogits + n * K);
        const int64_t count = (n_total - n) * (int64_t)(K / 4);
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock)
            z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (j == 0) s[g] = term, sc[g] = counted;
    red[threadIdx.x] = cs[0];
    red[kBlock + threadIdx.x] = CH > 1 ? cs[CH - 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int c = 0;
        for (int i = 0; i < ROWS; ++i) t = t + s[i], c += sc[i];
        partial[blockIdx.x] = t;
        partial_cnt[blockIdx.x] = c;
    }
    if ((int)threadIdx.x < 2 * G) {                              // as xent_bwd_rows_kernel
        const int chunk = threadIdx.x / G, jj = threadIdx.x % G, col = (jj + chunk * G) * 4;
        if (col < K) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < ROWS; ++q) {
                const float4 v = red[chunk * kBlock + q * G + jj];
                t = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
            }
            *reinterpret_cast<float4 *>(cs_partial + (int64_t)blockIdx.x * K + col) = t;
        }
    }
}

// d *= g, colsum *= g -- unless g is exactly 1 (loss.backward()), when every workgroup reads g and leaves
__global__ __launch_bounds__(kBlock) void xent_scale_grad_kernel(float *__restrict__ d, float *__restrict__ colsum,
                                                                 const float *__restrict__ g_loss, int64_t total4, int64_t tail0,
                                                                 int64_t total, int K)
{
    const float g = g_loss[0];
    if (g == 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total4; i += (int64_t)gridDim.x * kBlock) {
        float4 v = reinterpret_cast<float4 *>(d)[i];
        reinterpret_cast<float4 *>(d)[i] = make_float4(v.x * g, v.y * g, v.z * g, v.w * g);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = tail0 + threadIdx.x; i < total; i += kBlock) d[i] = d[i] * g;
        if (colsum)
            for (int k = threadIdx.x; k < K; k += kBlock) colsum[k] = colsum[k] * g;
    }
}

constexpr int kXentGrid = 2048;

// ---- small matrices (the 2708 x 7 logits of Cora): ONE workgroup each way ---------------------------------------------------------
// A launch inside a replayed HIP graph costs ~ 4.5 us whatever it does, and the epoch of a 2708-vertex graph is 18 of them
// (tools/diag/cora_kernels.py).  The general path above is forward + finish + backward + column sums + their finish = 5; for a
// matrix one workgroup can hold in registers the same results take 2.  What such a launch costs beyond its 4.5 us is the number
// of DEPENDENT memory round trips (~ 1.5 us each): every load of a thread is issued before the first value is used (a first
// version that walked the rows in a loop took 7 + 10 us).
constexpr int kSmallBlock = 1024;
constexpr int kSmallPerThread = 64;            // elements a thread may hold (forward: rows x padded columns)
constexpr int kSmallBwdPasses = 32;            // backward: rows per (row lane, column) thread -- 3 registers each of 128

// thread = row (rows t, t + 1024, ..: PASSES of them), the row's KP >= K values in registers
template <int KP, int PASSES>
__global__ __launch_bounds__(kSmallBlock) void xent_small_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                                      float *__restrict__ lse, float *__restrict__ loss,
                                                                      float *__restrict__ n_counted, int *__restrict__ status, int n, int K)
{
    __shared__ float s[kSmallBlock];
    __shared__ int c[kSmallBlock];
    float x[PASSES][KP];
    int64_t lab[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = (int)threadIdx.x + p * kSmallBlock;
        lab[p] = kIgnoreIndex;
        if (row < n) lab[p] = labels[row];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            x[p][k] = -INFINITY;
            if (row < n && k < K) x[p][k] = logits[(int64_t)row * K + k];
        }
    }
    float term = 0.f;
    int counted = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = (int)threadIdx.x + p * kSmallBlock;
        if (row < n) {
            float m = -INFINITY, sum = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) m = fmaxf(m, x[p][k]);
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (k < K) sum = sum + expf(x[p][k] - m);
            const float l = m + logf(sum);
            lse[row] = l;
            const int64_t t = lab[p];
            if (t >= 0 && t < K) {
                float xt = 0.f;
#pragma unroll
                for (int k = 0; k < KP; ++k) xt = k == (int)t ? x[p][k] : xt;
                term = term + (l - xt), ++counted;
            } else if (t != kIgnoreIndex) {
                atomicOr(status, 1);
            }
        }
    }
    s[threadIdx.x] = term;
    c[threadIdx.x] = counted;
    __syncthreads();
    for (int off = kSmallBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + off];
            c[threadIdx.x] += c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = (float)c[0];
        n_counted[0] = cnt;
        loss[0] = s[0] / cnt;                                    // no counted row: 0 / 0 = NaN, as torch
    }
}

// KP = columns padded to a power of two: thread = (row lane r, column c), kSmallBlock / KP rows per pass, PASSES passes
template <int KP, int PASSES>
__global__ __launch_bounds__(kSmallBlock) void xent_small_bwd_kernel(const float *__restrict__ g_loss, const float *__restrict__ logits,
                                                                      const int64_t *__restrict__ labels, const float *__restrict__ lse,
                                                                      const float *__restrict__ n_counted, float *__restrict__ dlogits,
                                                                      float *__restrict__ colsum, int n, int n_total, int K)
{
    constexpr int R = kSmallBlock / KP;
    __shared__ float s[kSmallBlock];
    const int c = (int)threadIdx.x % KP, r = (int)threadIdx.x / KP;
    const float gl = g_loss[0], cnt = n_counted[0];
    float x[PASSES], l[PASSES];
    int lab[PASSES];                                             // -1: not a counted row
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r + p * R;
        x[p] = l[p] = 0.f;
        lab[p] = -1;
        if (row < n && c < K) {
            x[p] = logits[(int64_t)row * K + c], l[p] = lse[row];
            const int64_t t = labels[row];
            lab[p] = t >= 0 && t < K ? (int)t : -1;
        }
    }
    const float scale = gl / cnt;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r + p * R;
        if (row < n_total && c < K) {
            float v = 0.f;
            if (lab[p] >= 0) v = (expf(x[p] - l[p]) - (lab[p] == c ? 1.f : 0.f)) * scale;
            dlogits[(int64_t)row * K + c] = v;
            acc = acc + v;
        }
    }
    if (!colsum) return;                                         // kernel-uniform
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = R / 2; off > 0; off >>= 1) {                  // over the row lanes of a column, fixed order
        if (r < off) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + off * KP];
        __syncthreads();
    }
    if (r == 0 && c < K) colsum[c] = s[c];
}

inline int small_kp(int K) { return K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 64; }

constexpr int kSmallMaxK = 64;

inline int xent_lanes(int K) { return K <= 32 ? 8 : K <= 256 ? 32 : 64; }
inline int xent_blocks(int64_t n, int K)
{
    const int rows = kBlock / xent_lanes(K);
    return (int)std::min<int64_t>((n + rows - 1) / rows, kXentGrid);
}

}  // namespace
}  // namespace stg

extern "C" size_t stg_xent_workspace_bytes(int64_t n, int32_t K)
{
    if (n <= 0 || K <= 0) return 2 * sizeof(float);
    return 2 * sizeof(float) * (size_t)stg::xent_blocks(n, K);
}

extern "C" int stg_xent_fwd(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted,
                            int32_t *status, int64_t n, int32_t K, void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    if (n <= 0 || K <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_fwd: bad shape n=%lld K=%d", (long long)n, K);
    if (!logits || !labels || !lse || !loss || !n_counted || !status || !workspace)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_fwd: NULL pointer argument");
    if (workspace_bytes < stg_xent_workspace_bytes(n, K)) return fail(STG_ERR_WORKSPACE, "stg_xent_fwd: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = xent_blocks(n, K);
    float *partial = static_cast<float *>(workspace);
    int *partial_cnt = reinterpret_cast<int *>(partial + blocks);
    const bool v4 = K % 4 == 0 && reinterpret_cast<uintptr_t>(logits) % 16 == 0;
#define STG_XENT(G)                                                                                                    \
    if (v4)                                                                                                            \
        hipLaunchKernelGGL((xent_fwd_kernel<G, true>), dim3(blocks), dim3(kBlock), 0, stream, logits, labels, lse,     \
                           partial, partial_cnt, n, K, status);                                                        \
    else                                                                                                               \
        hipLaunchKernelGGL((xent_fwd_kernel<G, false>), dim3(blocks), dim3(kBlock), 0, stream, logits, labels, lse,    \
                           partial, partial_cnt, n, K, status)
#define STG_XENT_REG(G, CH)                                                                                            \
    hipLaunchKernelGGL((xent_fwd_reg_kernel<G, CH>), dim3(blocks), dim3(kBlock), 0, stream, logits, labels, lse, partial,  \
                       partial_cnt, n, K, status)
    const int G = xent_lanes(K);
    if (v4 && K <= 8 * G) {                                      // the row fits two float4s per lane: registers, two rows in flight
        const bool one = K <= 4 * G;
        switch (G) {
            case 8: if (one) STG_XENT_REG(8, 1); else STG_XENT_REG(8, 2); break;
            case 32: if (one) STG_XENT_REG(32, 1); else STG_XENT_REG(32, 2); break;
            default: if (one) STG_XENT_REG(64, 1); else STG_XENT_REG(64, 2); break;
        }
    } else {
        switch (G) {
            case 8: STG_XENT(8); break;
            case 32: STG_XENT(32); break;
            default: STG_XENT(64); break;
        }
    }
#undef STG_XENT_REG
#undef STG_XENT
    hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(kBlock), 0, stream, partial, partial_cnt, blocks, loss, n_counted);
    return check_launch("stg_xent_fwd");
}

namespace stg {
namespace {
inline int xent_bwd_row_blocks(int64_t n_total, int K)
{
    const int rows = kBlock / xent_lanes(K);
    return (int)std::min<int64_t>((n_total + rows - 1) / rows, 256 * 16);
}

int xent_bwd_launch(const float *g_loss, const float *logits, const int64_t *labels, const float *lse, const float *n_counted,
                    float *dlogits, float *colsum, float *partial, int64_t n, int64_t n_total, int32_t K, hipStream_t st)
{
    const bool rows_ok = K % 4 == 0 && reinterpret_cast<uintptr_t>(logits) % 16 == 0 && reinterpret_cast<uintptr_t>(dlogits) % 16 == 0;
    if (rows_ok) {
        const int G = xent_lanes(K), rblocks = xent_bwd_row_blocks(n_total, K);
#define STG_XBWD(G_)                                                                                                          \
    if (colsum)                                                                                                               \
        hipLaunchKernelGGL((xent_bwd_rows_kernel<G_, true>), dim3(rblocks), dim3(kBlock), 0, st, g_loss, logits, labels, lse,  \
                           dlogits, n, n_total, K, n_counted, partial);                                                        \
    else                                                                                                                      \
        hipLaunchKernelGGL((xent_bwd_rows_kernel<G_, false>), dim3(rblocks), dim3(kBlock), 0, st, g_loss, logits, labels, lse, \
                           dlogits, n, n_total, K, n_counted, nullptr)
        switch (G) {
            case 8: STG_XBWD(8); break;
            case 32: STG_XBWD(32); break;
            default: STG_XBWD(64); break;
        }
#undef STG_XBWD
        if (colsum)
            hipLaunchKernelGGL(xent_colsum_finish_kernel, dim3((K + 7) / 8), dim3(kBlock), 0, st, partial, colsum, rblocks, K);
        return 0;
    }
    const int64_t total = n_total * (int64_t)K;
    const int blocks = (int)std::min<int64_t>((total + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(blocks), dim3(kBlock), 0, st, g_loss, logits, labels, lse, dlogits, n, n_total, K,
                       n_counted);
    return 0;
}
}  // namespace
}  // namespace stg

extern "C" int stg_xent_bwd(const float *g_loss, const float *logits, const int64_t *labels, const float *lse,
                            const float *n_counted, float *dlogits, int64_t n, int64_t n_total, int32_t K, void *stream_)
{
    using namespace stg;
    if (n <= 0 || K <= 0 || n_total < n) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_bwd: bad shape");
    if (!g_loss || !logits || !labels || !lse || !n_counted || !dlogits) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_bwd: NULL pointer argument");
    xent_bwd_launch(g_loss, logits, labels, lse, n_counted, dlogits, nullptr, nullptr, n, n_total, K, static_cast<hipStream_t>(stream_));
    return check_launch("stg_xent_bwd");
}

extern "C" size_t stg_xent_bwd_colsum_workspace_bytes(int64_t n_total, int32_t K)
{
    if (n_total <= 0 || K <= 0 || K % 4 != 0 || K > 8 * stg::xent_lanes(K)) return 0;          // 0: shape not covered
    return sizeof(float) * (size_t)stg::xent_bwd_row_blocks(n_total, K) * (size_t)K;
}

extern "C" int stg_xent_bwd_colsum(const float *g_loss, const float *logits, const int64_t *labels, const float *lse,
                                   const float *n_counted, float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K,
                                   void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    if (n <= 0 || K <= 0 || n_total < n) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_bwd_colsum: bad shape");
    if (!g_loss || !logits || !labels || !lse || !n_counted || !dlogits || !colsum || !workspace)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_bwd_colsum: NULL pointer argument");
    const size_t need = stg_xent_bwd_colsum_workspace_bytes(n_total, K);
    if (need == 0 || reinterpret_cast<uintptr_t>(logits) % 16 != 0 || reinterpret_cast<uintptr_t>(dlogits) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(workspace) % 16 != 0)
        return fail(STG_ERR_UNSUPPORTED, "stg_xent_bwd_colsum: needs K %% 4 == 0, K <= 8 lanes-per-row and 16-byte aligned matrices (K=%d)", K);
    if (workspace_bytes < need) return fail(STG_ERR_WORKSPACE, "stg_xent_bwd_colsum: workspace %zu < required %zu", workspace_bytes, need);
    xent_bwd_launch(g_loss, logits, labels, lse, n_counted, dlogits, colsum, static_cast<float *>(workspace), n, n_total, K,
                    static_cast<hipStream_t>(stream_));
    return check_launch("stg_xent_bwd_colsum");
}

// ---- one pass: loss, lse and the gradient at g = 1 (see xent_fwd_grad_kernel) -----------------------------------------------------
extern "C" size_t stg_xent_fwd_grad_workspace_bytes(int64_t n_total, int32_t K)
{
    if (n_total <= 0 || K <= 0 || K % 4 != 0 || K > 8 * stg::xent_lanes(K)) return 0;          // 0: shape not covered
    const size_t blocks = (size_t)stg::xent_bwd_row_blocks(n_total, K);
    return sizeof(float) * blocks * (size_t)K + 2 * sizeof(float) * blocks + sizeof(int) * stg::kXentCountParts;
}

extern "C" int stg_xent_fwd_grad(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted,
                                 int32_t *status, float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K,
                                 void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    if (n <= 0 || K <= 0 || n_total < n) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_fwd_grad: bad shape n=%lld n_total=%lld K=%d",
                                                    (long long)n, (long long)n_total, K);
    if (!logits || !labels || !lse || !loss || !n_counted || !status || !dlogits || !colsum || !workspace)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_fwd_grad: NULL pointer argument");
    const size_t need = stg_xent_fwd_grad_workspace_bytes(n_total, K);
    if (need == 0 || reinterpret_cast<uintptr_t>(logits) % 16 != 0 || reinterpret_cast<uintptr_t>(dlogits) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(workspace) % 16 != 0)
        return fail(STG_ERR_UNSUPPORTED, "stg_xent_fwd_grad: needs K %% 4 == 0, K <= 8 lanes-per-row and 16-byte aligned matrices (K=%d)", K);
    if (workspace_bytes < need) return fail(STG_ERR_WORKSPACE, "stg_xent_fwd_grad: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    // one resident round of workgroups (8 per CU): the partial sums the finish launches add up stay few
    const int blocks = std::min(xent_bwd_row_blocks(n_total, K), 2048);
    float *cs_partial = static_cast<float *>(workspace);
    float *partial = cs_partial + (size_t)blocks * K;
    int *partial_cnt = reinterpret_cast<int *>(partial + blocks);
    int *cnt_parts = partial_cnt + blocks;
    const int parts = (int)std::min<int64_t>((n + kBlock * 4 - 1) / (kBlock * 4), kXentCountParts);
    hipLaunchKernelGGL(xent_count_kernel, dim3(parts), dim3(kBlock), 0, st, labels, n, K, cnt_parts);
    const int G = xent_lanes(K);
    const bool one = K <= 4 * G;
#define STG_XFG(G_, CH_)                                                                                                        \
    hipLaunchKernelGGL((xent_fwd_grad_kernel<G_, CH_>), dim3(blocks), dim3(kBlock), 0, st, logits, labels, lse, dlogits, partial,  \
                       partial_cnt, cs_partial, cnt_parts, parts, n, n_total, K, status)
    switch (G) {
        case 8: if (one) STG_XFG(8, 1); else STG_XFG(8, 2); break;
        case 32: if (one) STG_XFG(32, 1); else STG_XFG(32, 2); break;
        default: if (one) STG_XFG(64, 1); else STG_XFG(64, 2); break;
    }
#undef STG_XFG
    hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(kBlock), 0, st, partial, partial_cnt, blocks, loss, n_counted);
    hipLaunchKernelGGL(xent_colsum_finish_kernel, dim3((K + 7) / 8), dim3(kBlock), 0, st, cs_partial, colsum, blocks, K);
    return check_launch("stg_xent_fwd_grad");
}

extern "C" int stg_xent_scale_grad(float *dlogits, float *colsum, const float *g_loss, int64_t n_total, int32_t K, void *stream_)
{
    using namespace stg;
    if (n_total <= 0 || K <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_scale_grad: bad shape");
    if (!dlogits || !g_loss) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_scale_grad: NULL pointer argument");
    const int64_t total = n_total * (int64_t)K;
    const bool v4 = reinterpret_cast<uintptr_t>(dlogits) % 16 == 0;
    const int64_t total4 = v4 ? total / 4 : 0;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>((total4 + kBlock - 1) / kBlock, 1), 256 * 8);
    hipLaunchKernelGGL(xent_scale_grad_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream_), dlogits, colsum, g_loss,
                       total4, total4 * 4, total, K);
    return check_launch("stg_xent_scale_grad");
}

extern "C" int stg_xent_small_supported(int64_t n_total, int32_t K)
{
    // both kernels hold a thread's share of the matrix in registers: (rows per thread) x (padded columns) <= kSmallPerThread
    if (n_total <= 0 || K <= 0 || K > stg::kSmallMaxK) return 0;
    const int kp = stg::small_kp(K);
    const int64_t fwd_passes = (n_total + stg::kSmallBlock - 1) / stg::kSmallBlock;
    const int64_t bwd_passes = (n_total + stg::kSmallBlock / kp - 1) / (stg::kSmallBlock / kp);
    return fwd_passes * kp <= stg::kSmallPerThread && bwd_passes <= stg::kSmallBwdPasses;
}

extern "C" int stg_xent_small_fwd(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted, int32_t *status,
                                  int64_t n, int32_t K, void *stream_)
{
    using namespace stg;
    if (n <= 0 || !stg_xent_small_supported(n, K)) return fail(STG_ERR_UNSUPPORTED, "stg_xent_small_fwd: n=%lld K=%d is not a small matrix", (long long)n, K);
    if (!logits || !labels || !lse || !loss || !n_counted || !status) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_small_fwd: NULL pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int kp = small_kp(K), passes = (int)((n + kSmallBlock - 1) / kSmallBlock);
#define STG_XSF(KP_, P_)                                                                                                          \
    hipLaunchKernelGGL((xent_small_fwd_kernel<KP_, P_>), dim3(1), dim3(kSmallBlock), 0, st, logits, labels, lse, loss, n_counted, status, (int)n, K)
#define STG_XSF_P(KP_)                                                                                                            \
    do {                                                                                                                          \
        if (passes <= 1) STG_XSF(KP_, 1);                                                                                         \
        else if (passes <= 2 && 2 * KP_ <= kSmallPerThread) STG_XSF(KP_, (2 * KP_ <= kSmallPerThread ? 2 : 1));                   \
        else if (passes <= 4 && 4 * KP_ <= kSmallPerThread) STG_XSF(KP_, (4 * KP_ <= kSmallPerThread ? 4 : 1));                   \
        else if (passes <= 8 && 8 * KP_ <= kSmallPerThread) STG_XSF(KP_, (8 * KP_ <= kSmallPerThread ? 8 : 1));                   \
        else STG_XSF(KP_, (16 * KP_ <= kSmallPerThread ? 16 : 1));                                                                \
    } while (0)
    switch (kp) {
        case 4: STG_XSF_P(4); break;
        case 8: STG_XSF_P(8); break;
        case 16: STG_XSF_P(16); break;
        case 32: STG_XSF_P(32); break;
        default: STG_XSF_P(64); break;
    }
#undef STG_XSF_P
#undef STG_XSF
    return check_launch("stg_xent_small_fwd");
}

extern "C" int stg_xent_small_bwd(const float *g_loss, const float *logits, const int64_t *labels, const float *lse, const float *n_counted,
                                  float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K, void *stream_)
{
    using namespace stg;
    if (n <= 0 || n_total < n || !stg_xent_small_supported(n_total, K))
        return fail(STG_ERR_UNSUPPORTED, "stg_xent_small_bwd: n=%lld n_total=%lld K=%d is not a small matrix", (long long)n, (long long)n_total, K);
    if (!g_loss || !logits || !labels || !lse || !n_counted || !dlogits) return fail(STG_ERR_INVALID_ARGUMENT, "stg_xent_small_bwd: NULL pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int kp = small_kp(K), passes = (int)((n_total + kSmallBlock / kp - 1) / (kSmallBlock / kp));
#define STG_XSB(KP_, P_)                                                                                                          \
    hipLaunchKernelGGL((xent_small_bwd_kernel<KP_, P_>), dim3(1), dim3(kSmallBlock), 0, st, g_loss, logits, labels, lse, n_counted,   \
                       dlogits, colsum, (int)n, (int)n_total, K)
#define STG_XSB_P(KP_)                                                                                                            \
    do {                                                                                                                          \
        if (passes <= 8) STG_XSB(KP_, 8);                                                                                         \
        else if (passes <= 24) STG_XSB(KP_, 24);                                                                                  \
        else STG_XSB(KP_, kSmallBwdPasses);                                                                                       \
    } while (0)
    switch (kp) {
        case 4: STG_XSB_P(4); break;
        case 8: STG_XSB_P(8); break;
        case 16: STG_XSB_P(16); break;
        case 32: STG_XSB_P(32); break;
        default: STG_XSB_P(64); break;
    }
#undef STG_XSB_P
#undef STG_XSB
    return check_launch("stg_xent_small_bwd");
}
