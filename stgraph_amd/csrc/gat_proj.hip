// GATConv's attention projections (reference nn/pytorch/static/gat_conv.py:43-45) and their backward:
//   el[n,h] = sum_d feat[n,h,d] * attn_l[h,d]      er[n,h] = sum_d feat[n,h,d] * attn_r[h,d]
//   dfeat[n,h,d] = g[n,h,d] + del[n,h] * attn_l[h,d] + der[n,h] * attn_r[h,d]        (g = gradient from the GAT units)
//   dattn_l[h,d] = sum_n del[n,h] * feat[n,h,d]     dattn_r[h,d] = sum_n der[n,h] * feat[n,h,d]
// In torch these are two broadcast multiplies + two reductions forward and four multiplies, two adds and two
// |V|-long reductions backward, each streaming the [N,H,D] tensor (512 MB at cfg3) through HBM: 3.7 of the
// layer's 10.9 ms.  Here: one pass forward (reads feat once, writes [N,H] twice), one pass backward (reads feat
// and g, writes dfeat; per-workgroup partial column sums, fixed-order finish -- no atomics).  HBM bound.
//
// Mapping: one thread per float4 of a row (HD / 4 per row, consecutive lanes), the D / 4 lanes of a head reduce
// with xor shuffles (D / 4 a power of two <= 64).  Backward: the grid stride is a multiple of HD / 4, so a thread
// keeps its (h, d) position and accumulates its partials in registers.
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

constexpr int kProjMaxGrid = 2048;

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

__global__ __launch_bounds__(kBlock) void gat_proj_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ attn_l,
                                                             const float *__restrict__ attn_r, float *__restrict__ el,
                                                             float *__restrict__ er, int64_t N, int H, int D)
{
    const int q4 = H * D / 4, d4 = D / 4;
    const int64_t total = N * q4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i - (threadIdx.x & (d4 - 1)) < total; i += stride) {
        // all d4 lanes of a head take the same trip count (total is a multiple of d4): shuffles stay converged
        const bool ok = i < total;
        const int64_t row = ok ? i / q4 : 0;
        const int q = ok ? (int)(i - row * q4) : 0;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) f = ld4g(feat + i * 4);
        float sl = dot4(f, ld4g(attn_l + q * 4)), sr = dot4(f, ld4g(attn_r + q * 4));
        for (int off = d4 >> 1; off > 0; off >>= 1) {
            sl += __shfl_xor(sl, off, kWave);
            sr += __shfl_xor(sr, off, kWave);
        }
        if (ok && (q & (d4 - 1)) == 0) {
            const int64_t o = row * H + q / d4;
            el[o] = sl;
            er[o] = sr;
        }
    }
}

// partial_l / partial_r: [gridDim.x][HD]
__global__ __launch_bounds__(kBlock) void gat_proj_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ attn_l,
                                                             const float *__restrict__ attn_r, const float *__restrict__ del,
                                                             const float *__restrict__ der, const float *g /* may alias dfeat */,
                                                             float *dfeat, float *__restrict__ partial_l,
                                                             float *__restrict__ partial_r, int64_t N, int H, int D)
{
    __shared__ float4 red[2][kBlock];
    const int q4 = H * D / 4, d4 = D / 4;
    const int64_t total = N * q4, stride = (int64_t)gridDim.x * blockDim.x;      // stride % q4 == 0 (host)
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(first % q4);                                              // fixed for this thread
    const float4 al = ld4g(attn_l + q * 4), ar = ld4g(attn_r + q * 4);
    const int h = q / d4;
    float4 pl = make_float4(0.f, 0.f, 0.f, 0.f), pr = pl;
    for (int64_t i = first; i < total; i += stride) {
        const int64_t row = i / q4;
        const float a = del[row * H + h], b = der[row * H + h];
        const float4 f = ld4g(feat + i * 4);
        float4 o = g ? ld4g(g + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        o.x += a * al.x + b * ar.x;
        o.y += a * al.y + b * ar.y;
        o.z += a * al.z + b * ar.z;
        o.w += a * al.w + b * ar.w;
        *reinterpret_cast<float4 *>(dfeat + i * 4) = o;
        pl.x += a * f.x; pl.y += a * f.y; pl.z += a * f.z; pl.w += a * f.w;
        pr.x += b * f.x; pr.y += b * f.y; pr.z += b * f.z; pr.w += b * f.w;
    }
    // threads of this workgroup that share a column position: tid, tid + q4, ... (q4 <= 256 divides 256 or vice versa)
    red[0][threadIdx.x] = pl;
    red[1][threadIdx.x] = pr;
    __syncthreads();
    const int per = q4 < kBlock ? q4 : kBlock;                   // distinct column positions in this workgroup
    if ((int)threadIdx.x < per) {
        float4 sl = make_float4(0.f, 0.f, 0.f, 0.f), sr = sl;
        for (int t = threadIdx.x; t < kBlock; t += per) {
            const float4 x = red[0][t], y = red[1][t];
            sl.x += x.x; sl.y += x.y; sl.z += x.z; sl.w += x.w;
            sr.x += y.x; sr.y += y.y; sr.z += y.z; sr.w += y.w;
        }
        const int qq = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % q4);
        float *dl = partial_l + (int64_t)blockIdx.x * H * D + qq * 4, *dr = partial_r + (int64_t)blockIdx.x * H * D + qq * 4;
        *reinterpret_cast<float4 *>(dl) = sl;
        *reinterpret_cast<float4 *>(dr) = sr;
    }
}

// out[f] = sum over the workgroups that hold column f of partial[b][f]; a workgroup covers `per` float4 columns
// starting at (b * 256) % q4, so with q4 > 256 only every (q4 / 256)-th workgroup holds a given column
__global__ __launch_bounds__(kBlock) void gat_proj_finish_kernel(const float *__restrict__ partial_l, const float *__restrict__ partial_r,
                                                                float *__restrict__ dattn_l, float *__restrict__ dattn_r,
                                                                int blocks, int HD)
{
    constexpr int kCols = 8, kGroups = kBlock / kCols;
    __shared__ float red[2][kGroups][kCols];
    const int c = threadIdx.x % kCols, grp = threadIdx.x / kCols;
    const int f = blockIdx.x * kCols + c;
    const int q4 = HD / 4;
    float sl = 0.f, sr = 0.f;
    if (f < HD) {
        const int fq = f / 4;
        // eight partial rows in flight per thread (loads unconditional -- a row that does not hold this column reads a valid word and
        // adds +0 -- then the adds in row order): one row per round trip took 27 us for the 2 x 16 columns of cfg3's second layer
        constexpr int U = 8;
        for (int b0 = grp; b0 < blocks; b0 += kGroups * U) {
            float vl[U], vr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * kGroups;
                const int bb = b < blocks ? b : blocks - 1;
                const int start = (int)(((int64_t)bb * kBlock) % q4);        // first column position of workgroup b
                const int rel = (fq - start + q4) % q4;
                const bool ok = b < blocks && rel < (q4 < kBlock ? q4 : kBlock);
                const float l = partial_l[(int64_t)bb * HD + f], r = partial_r[(int64_t)bb * HD + f];
                vl[u] = ok ? l : 0.f;
                vr[u] = ok ? r : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) sl += vl[u], sr += vr[u];
        }
    }
    red[0][grp][c] = sl;
    red[1][grp][c] = sr;
    __syncthreads();
    if (grp == 0 && f < HD) {
        float tl = 0.f, tr = 0.f;
        for (int k = 0; k < kGroups; ++k) tl += red[0][k][c], tr += red[1][k][c];
        dattn_l[f] = tl;
        dattn_r[f] = tr;
    }
}

bool proj_shape_ok(int32_t H, int32_t D)
{
    if (H <= 0 || D <= 0 || D % 4 != 0) return false;
    const int d4 = D / 4, q4 = H * D / 4;
    if (d4 > kWave || (d4 & (d4 - 1))) return false;                   // a head's lanes reduce inside one wave
    return (q4 <= kBlock && kBlock % q4 == 0) || (q4 > kBlock && q4 % kBlock == 0 && q4 <= 16 * kBlock);
}

int proj_grid(int64_t N, int32_t H, int32_t D)
{
    const int q4 = H * D / 4;
    const int64_t total = N * q4;
    int64_t blocks = std::min<int64_t>((total + kBlock - 1) / kBlock, kProjMaxGrid);
    const int mult = q4 > kBlock ? q4 / kBlock : 1;                     // grid stride must be a multiple of q4
    blocks = std::max<int64_t>(mult, blocks / mult * mult);
    return (int)blocks;
}

}  // namespace
}  // namespace stg

extern "C" int stg_gat_proj_supported(int32_t H, int32_t D) { return stg::proj_shape_ok(H, D) ? 1 : 0; }

extern "C" int stg_gat_proj_fwd(const float *feat, const float *attn_l, const float *attn_r, float *el, float *er,
                                int64_t N, int32_t H, int32_t D, void *stream)
{
    using namespace stg;
    if (N < 0 || !proj_shape_ok(H, D)) return fail(STG_ERR_UNSUPPORTED, "stg_gat_proj_fwd: unsupported shape N=%lld H=%d D=%d", (long long)N, H, D);
    if (N == 0) return 0;
    if (!feat || !attn_l || !attn_r || !el || !er) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_proj_fwd: NULL pointer argument");
    hipLaunchKernelGGL(gat_proj_fwd_kernel, dim3(proj_grid(N, H, D)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), feat, attn_l,
                       attn_r, el, er, N, H, D);
    return check_launch("stg_gat_proj_fwd");
}

extern "C" size_t stg_gat_proj_bwd_workspace_bytes(int64_t N, int32_t H, int32_t D)
{
    if (N <= 0 || !stg::proj_shape_ok(H, D)) return 0;
    return 2 * sizeof(float) * (size_t)stg::proj_grid(N, H, D) * (size_t)H * (size_t)D;
}

extern "C" int stg_gat_proj_bwd(const float *feat, const float *attn_l, const float *attn_r, const float *del, const float *der,
                                const float *g, float *dfeat, float *dattn_l, float *dattn_r, int64_t N, int32_t H, int32_t D,
                                void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0 || !proj_shape_ok(H, D)) return fail(STG_ERR_UNSUPPORTED, "stg_gat_proj_bwd: unsupported shape N=%lld H=%d D=%d", (long long)N, H, D);
    if (!dattn_l || !dattn_r) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_proj_bwd: NULL pointer argument");
    if (N == 0) {
        if (const int rc = zero_async(dattn_l, sizeof(float) * (size_t)H * D, stream)) return rc;
        return zero_async(dattn_r, sizeof(float) * (size_t)H * D, stream);
    }
    if (!feat || !attn_l || !attn_r || !del || !der || !dfeat) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_proj_bwd: NULL pointer argument");
    const int blocks = proj_grid(N, H, D);
    const size_t need = 2 * sizeof(float) * (size_t)blocks * (size_t)H * (size_t)D;
    if (!workspace || workspace_bytes < need) return fail(STG_ERR_WORKSPACE, "stg_gat_proj_bwd: workspace %zu < required %zu", workspace_bytes, need);
    float *pl = static_cast<float *>(workspace), *pr = pl + (size_t)blocks * H * D;
    // partial rows are only written at the column positions a workgroup owns; the finish kernel reads exactly those
    hipLaunchKernelGGL(gat_proj_bwd_kernel, dim3(blocks), dim3(kBlock), 0, stream, feat, attn_l, attn_r, del, der, g, dfeat, pl, pr, N, H, D);
    hipLaunchKernelGGL(gat_proj_finish_kernel, dim3((H * D + 7) / 8), dim3(kBlock), 0, stream, pl, pr, dattn_l, dattn_r, blocks, H * D);
    return check_launch("stg_gat_proj_bwd");
}


// The small products of the projection fold (nn/functional._gat_backward_uniform; gat_conv.py:43-48 differentiated at width H instead of
// H * D), ONE launch, workgroup h = head h with W_h [D][fin] in LDS:
//   dattn_l[h, d] = sum_f W_h[d, f] G[h, f],  dattn_r[h, d] = sum_f W_h[d, f] G[H + h, f]            (G [2H][fin] = [grad_el | grad_er]^T x)
//   Aw[h, f] = sum_d W_h[d, f] attn_l[h, d],  Aw[H + h, f] = sum_d W_h[d, f] attn_r[h, d]            (nullable: gx += [grad_el | grad_er] Aw)
//   gw[h D + d, f] += attn_l[h, d] G[h, f] + attn_r[h, d] G[H + h, f]                                (nullable)
// -- four einsums (two batched GEMMs each with their copies), a cat and five elementwise launches before: ~ 13 launches of ~ 5 us.
namespace stg {
namespace {
__global__ __launch_bounds__(kBlock) void gat_attn_fold_kernel(const float *__restrict__ W, const float *__restrict__ G,
                                                              const float *__restrict__ attn_l, const float *__restrict__ attn_r,
                                                              float *__restrict__ dattn_l, float *__restrict__ dattn_r,
                                                              float *__restrict__ Aw, float *__restrict__ gw, int H, int D, int fin)
{
    extern __shared__ float lds[];
    float *Ws = lds, *gl = Ws + D * fin, *gr = gl + fin, *al = gr + fin, *ar = al + D;
    const int h = (int)blockIdx.x, tid = (int)threadIdx.x;
    const float *Wh = W + (int64_t)h * D * fin;
    for (int i = tid; i < D * fin; i += kBlock) Ws[i] = Wh[i];
    for (int i = tid; i < fin; i += kBlock) gl[i] = G[(int64_t)h * fin + i], gr[i] = G[(int64_t)(H + h) * fin + i];
    for (int i = tid; i < D; i += kBlock) al[i] = attn_l[h * D + i], ar[i] = attn_r[h * D + i];
    __syncthreads();
    for (int d = tid; d < D; d += kBlock) {
        float sl = 0.f, sr = 0.f;
        for (int f = 0; f < fin; ++f) {
            const float w = Ws[d * fin + (f + d) % fin];                     // (rotated start: lanes d, d + 1 on different banks)
            sl = sl + w * gl[(f + d) % fin];
            sr = sr + w * gr[(f + d) % fin];
        }
        dattn_l[h * D + d] = sl;
        dattn_r[h * D + d] = sr;
    }
    if (Aw) {
        for (int f = tid; f < fin; f += kBlock) {
            float sl = 0.f, sr = 0.f;
            for (int d = 0; d < D; ++d) {
                const float w = Ws[d * fin + f];
                sl = sl + w * al[d];
                sr = sr + w * ar[d];
            }
            Aw[(int64_t)h * fin + f] = sl;
            Aw[(int64_t)(H + h) * fin + f] = sr;
        }
    }
    if (gw) {
        float *gh = gw + (int64_t)h * D * fin;
        for (int i = tid; i < D * fin; i += kBlock) {
            const int d = i / fin, f = i - d * fin;
            gh[i] = gh[i] + (al[d] * gl[f] + ar[d] * gr[f]);
        }
    }
}
}  // namespace
}  // namespace stg

extern "C" int stg_gat_attn_fold(const float *W, const float *G, const float *attn_l, const float *attn_r, float *dattn_l, float *dattn_r,
                                 float *Aw, float *gw, int32_t H, int32_t D, int32_t fin, void *stream)
{
    using namespace stg;
    const size_t lds = ((size_t)D * fin + 2 * (size_t)fin + 2 * (size_t)D) * sizeof(float);
    if (H <= 0 || D <= 0 || fin <= 0 || lds > 64 * 1024)
        return fail(STG_ERR_UNSUPPORTED, "stg_gat_attn_fold: H=%d D=%d fin=%d (one head's weights must fit 64 KB of LDS)", H, D, fin);
    if (!W || !G || !attn_l || !attn_r || !dattn_l || !dattn_r) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_attn_fold: NULL pointer argument");
    hipLaunchKernelGGL(gat_attn_fold_kernel, dim3((unsigned)H), dim3(kBlock), lds, static_cast<hipStream_t>(stream), W, G, attn_l, attn_r,
                       dattn_l, dattn_r, Aw, gw, H, D, fin);
    return check_launch("stg_gat_attn_fold");
}
