// Row-local glue of one TGCN step (reference nn/pytorch/temporal/tgcn.py:21-55), fused.
//
// After the neighbour aggregation everything a TGCN step does is local to a vertex row: bias +
// clamp of the three gate pre-activations, the two concatenations [h | H], sigmoid/tanh, the GRU
// blend.  In torch that is ~57 elementwise/cat/fill launches per snapshot (forward + backward), each
// streaming [|V|, 64] tensors through HBM once -- 45 % of the cfg4 step once the aggregation and the
// weight gradients were fixed (profiles/r01_tgcn_cfg4_kernel_stats.csv).  The six kernels below do
// the same arithmetic with each intermediate read/written once; the three gate GEMMs in between stay
// on rocBLAS and the concatenated operands are written IN PLACE into their GEMM input buffers
// ([hz|H], [hr|H], [hh|H*R]) so no cat kernel exists.
//
// All tensors fp32 row-major; C = hidden width (multiple of 4); one thread handles 4 consecutive
// columns of one row (16-B accesses), grid-stride over N*C/4.
#include "stg_common.hpp"

namespace stg {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct F4 {
    float v[4];
};
__device__ __forceinline__ F4 ld4(const float *p)
{
    const float4 t = *reinterpret_cast<const float4 *>(p);
    return {{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void st4(float *p, const F4 &a)
{
    *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

// a3 [N,3C] (aggregated X [Wz|Wr|Wh]), b3 [3C], H [N,C]
//   h = clamp(a3 + b3, lo, hi);  CZ = [hz | H], CR = [hr | H], CH[:, :C] = hh      (each [N,2C])
__global__ void cell_prep_fwd_kernel(const float *__restrict__ a3, const float *__restrict__ b3,
                                     const float *__restrict__ H, float *__restrict__ CZ,
                                     float *__restrict__ CR, float *__restrict__ CH, int64_t N, int C,
                                     float lo, float hi)
{
    const int q = C / 4;
    const int64_t total = N * q, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        const F4 h = ld4(H + r * C + c);
        float *dst[3] = {CZ, CR, CH};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const F4 a = ld4(a3 + r * 3 * C + g * C + c);
            const F4 b = ld4(b3 + g * C + c);
            F4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.v[k] = fminf(fmaxf(a.v[k] + b.v[k], lo), hi);
            st4(dst[g] + r * 2 * C + c, o);
        }
        st4(CZ + r * 2 * C + C + c, h);
        st4(CR + r * 2 * C + C + c, h);
    }
}

// zl, rl [N,C] (gate pre-activations) -> Z = sigmoid(zl), R = sigmoid(rl), CH[:, C:] = H * R
__global__ void cell_gates_fwd_kernel(const float *__restrict__ zl, const float *__restrict__ rl,
                                      const float *__restrict__ H, float *__restrict__ Z,
                                      float *__restrict__ R, float *__restrict__ CH, int64_t N, int C)
{
    const int q = C / 4;
    const int64_t total = N * q, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        const F4 a = ld4(zl + r * C + c), b = ld4(rl + r * C + c), h = ld4(H + r * C + c);
        F4 z, rr, hr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z.v[k] = sigmoidf_(a.v[k]);
            rr.v[k] = sigmoidf_(b.v[k]);
            hr.v[k] = h.v[k] * rr.v[k];
        }
        st4(Z + r * C + c, z);
        st4(R + r * C + c, rr);
        st4(CH + r * 2 * C + C + c, hr);
    }
}

// hl [N,C] -> Ht = tanh(hl); Hn = Z*H + (1 - Z)*Ht
__global__ void cell_update_fwd_kernel(const float *__restrict__ hl, const float *__restrict__ Z,
                                       const float *__restrict__ H, float *__restrict__ Ht,
                                       float *__restrict__ Hn, int64_t N, int C)
{
    const int64_t total = N * C / 4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const F4 a = ld4(hl + i * 4), z = ld4(Z + i * 4), h = ld4(H + i * 4);
        F4 t, n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t.v[k] = tanhf(a.v[k]);
            n.v[k] = z.v[k] * h.v[k] + (1.0f - z.v[k]) * t.v[k];
        }
        st4(Ht + i * 4, t);
        st4(Hn + i * 4, n);
    }
}

// dHn -> dhl = dHn*(1-Z)*(1-Ht^2);  dzl = dHn*(H-Ht)*Z*(1-Z);  dH = dHn*Z
__global__ void cell_update_bwd_kernel(const float *__restrict__ dHn, const float *__restrict__ Z,
                                       const float *__restrict__ H, const float *__restrict__ Ht,
                                       float *__restrict__ dhl, float *__restrict__ dzl,
                                       float *__restrict__ dH, int64_t N, int C)
{
    const int64_t total = N * C / 4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const F4 g = ld4(dHn + i * 4), z = ld4(Z + i * 4), h = ld4(H + i * 4), t = ld4(Ht + i * 4);
        F4 a, b, c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.v[k] = (g.v[k] * (1.0f - z.v[k])) * (1.0f - t.v[k] * t.v[k]);
            b.v[k] = (g.v[k] * (h.v[k] - t.v[k])) * (z.v[k] * (1.0f - z.v[k]));
            c.v[k] = g.v[k] * z.v[k];
        }
        st4(dhl + i * 4, a);
        st4(dzl + i * 4, b);
        st4(dH + i * 4, c);
    }
}

// dCH [N,2C] (grad of [hh | H*R]): drl = dHR*H*R*(1-R);  dH += dHR*R
__global__ void cell_gates_bwd_kernel(const float *__restrict__ dCH, const float *__restrict__ R,
                                      const float *__restrict__ H, float *__restrict__ drl,
                                      float *__restrict__ dH, int64_t N, int C)
{
    const int q = C / 4;
    const int64_t total = N * q, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        const F4 g = ld4(dCH + r * 2 * C + C + c), rr = ld4(R + r * C + c), h = ld4(H + r * C + c);
        F4 d = ld4(dH + r * C + c), o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o.v[k] = (g.v[k] * h.v[k]) * (rr.v[k] * (1.0f - rr.v[k]));
            d.v[k] = d.v[k] + g.v[k] * rr.v[k];
        }
        st4(drl + r * C + c, o);
        st4(dH + r * C + c, d);
    }
}

// da3 = [dCZ[:, :C] | dCR[:, :C] | dCH[:, :C]] masked by lo <= a3 + b3 <= hi (clamp backward);
// dH += dCZ[:, C:] + dCR[:, C:]
__global__ void cell_prep_bwd_kernel(const float *__restrict__ dCZ, const float *__restrict__ dCR,
                                     const float *__restrict__ dCH, const float *__restrict__ a3,
                                     const float *__restrict__ b3, float *__restrict__ da3,
                                     float *__restrict__ dH, int64_t N, int C, float lo, float hi)
{
    const int q = C / 4;
    const int64_t total = N * q, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        const float *src[3] = {dCZ, dCR, dCH};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const F4 d = ld4(src[g] + r * 2 * C + c);
            const F4 a = ld4(a3 + r * 3 * C + g * C + c);
            const F4 b = ld4(b3 + g * C + c);
            F4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = a.v[k] + b.v[k];
                o.v[k] = (v >= lo && v <= hi) ? d.v[k] : 0.f;
            }
            st4(da3 + r * 3 * C + g * C + c, o);
        }
        const F4 x = ld4(dCZ + r * 2 * C + C + c), y = ld4(dCR + r * 2 * C + C + c);
        F4 d = ld4(dH + r * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) d.v[k] = d.v[k] + x.v[k] + y.v[k];
        st4(dH + r * C + c, d);
    }
}

namespace {
inline unsigned cell_grid(int64_t work)
{
    const int64_t b = (work + kBlock - 1) / kBlock;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 256 * 16));
}
inline int cell_check(const char *what, int64_t N, int C, std::initializer_list<const void *> ptrs)
{
    if (N < 0 || C <= 0 || C % 4 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad shape N=%lld C=%d (C must be a multiple of 4)", what,
                    (long long)N, C);
    for (const void *p : ptrs)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15))
            return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL or not 16-byte aligned pointer", what);
    return 0;
}
}  // namespace
}  // namespace stg

#define STG_CELL_LAUNCH(kernel, what, N, C, ...)                                                        \
    if ((N) == 0) return 0;                                                                             \
    hipLaunchKernelGGL(stg::kernel, dim3(stg::cell_grid((int64_t)(N) * (C) / 4)), dim3(stg::kBlock), 0, \
                       static_cast<hipStream_t>(stream), __VA_ARGS__);                                  \
    return stg::check_launch(what)

extern "C" int stg_tgcn_cell_prep_fwd(const float *a3, const float *b3, const float *H, float *CZ, float *CR,
                                      float *CH, int64_t N, int32_t C, float lo, float hi, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_prep_fwd", N, C, {a3, b3, H, CZ, CR, CH})) return rc;
    STG_CELL_LAUNCH(cell_prep_fwd_kernel, "stg_tgcn_cell_prep_fwd", N, C, a3, b3, H, CZ, CR, CH, N, C, lo, hi);
}

extern "C" int stg_tgcn_cell_gates_fwd(const float *zl, const float *rl, const float *H, float *Z, float *R,
                                       float *CH, int64_t N, int32_t C, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_gates_fwd", N, C, {zl, rl, H, Z, R, CH})) return rc;
    STG_CELL_LAUNCH(cell_gates_fwd_kernel, "stg_tgcn_cell_gates_fwd", N, C, zl, rl, H, Z, R, CH, N, C);
}

extern "C" int stg_tgcn_cell_update_fwd(const float *hl, const float *Z, const float *H, float *Ht, float *Hn,
                                        int64_t N, int32_t C, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_update_fwd", N, C, {hl, Z, H, Ht, Hn})) return rc;
    STG_CELL_LAUNCH(cell_update_fwd_kernel, "stg_tgcn_cell_update_fwd", N, C, hl, Z, H, Ht, Hn, N, C);
}

extern "C" int stg_tgcn_cell_update_bwd(const float *dHn, const float *Z, const float *H, const float *Ht,
                                        float *dhl, float *dzl, float *dH, int64_t N, int32_t C, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_update_bwd", N, C, {dHn, Z, H, Ht, dhl, dzl, dH})) return rc;
    STG_CELL_LAUNCH(cell_update_bwd_kernel, "stg_tgcn_cell_update_bwd", N, C, dHn, Z, H, Ht, dhl, dzl, dH, N, C);
}

extern "C" int stg_tgcn_cell_gates_bwd(const float *dCH, const float *R, const float *H, float *drl, float *dH,
                                       int64_t N, int32_t C, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_gates_bwd", N, C, {dCH, R, H, drl, dH})) return rc;
    STG_CELL_LAUNCH(cell_gates_bwd_kernel, "stg_tgcn_cell_gates_bwd", N, C, dCH, R, H, drl, dH, N, C);
}

extern "C" int stg_tgcn_cell_prep_bwd(const float *dCZ, const float *dCR, const float *dCH, const float *a3,
                                      const float *b3, float *da3, float *dH, int64_t N, int32_t C, float lo,
                                      float hi, void *stream)
{
    if (int rc = stg::cell_check("stg_tgcn_cell_prep_bwd", N, C, {dCZ, dCR, dCH, a3, b3, da3, dH})) return rc;
    STG_CELL_LAUNCH(cell_prep_bwd_kernel, "stg_tgcn_cell_prep_bwd", N, C, dCZ, dCR, dCH, a3, b3, da3, dH, N, C, lo,
                    hi);
}
