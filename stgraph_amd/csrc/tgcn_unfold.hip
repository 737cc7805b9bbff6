// The gate and conv parameter gradients of a TGCN window from the three contractions that need neither x3 nor da3
// (stgraph_amd/temporal.py::_unfold_gate_grads, one launch for the three gates instead of ~25 small torch launches per window).
//
// Per gate g (reference nn/pytorch/temporal/tgcn.py:21-41; x3_g = P Wc_g + bc_g unclamped, pre_g = [x3_g | Hx] Wg^T + bg,
// d_g = d cost / d pre_g), given R_g = d_g^T [Hx | P]  [C, C + Fin]  and  cs_g = column sums of d_g  [C]:
//   dWg[:, :C]  = d_g^T x3_g       = MgT Wc_g + cs_g bc_g^T          (MgT = R_g[:, C:],  [C, Fin])
//   dWg[:, C:]  = d_g^T Hx         = R_g[:, :C]
//   dbg         = cs_g
//   dWc_g       = P^T (d_g Wg[:, :C]) = MgT^T Wg[:, :C]              [Fin, C]
//   dbc_g       = cs_g Wg[:, :C]                                     [C]
// One thread per output, sums of at most C terms in index order: deterministic.
#include "stg_common.hpp"

#include "../../include/stgraph_hip.h"

namespace stg {
namespace {

struct UnfoldArgs {
    const float *R[3], *cs[3], *Wc[3], *bc[3], *Wg[3];
    float *dWg[3], *dbg[3], *dWc[3], *dbc[3];
    int C, Fin;
};

__global__ __launch_bounds__(kBlock) void tgcn_unfold_kernel(const UnfoldArgs a)
{
    const int C = a.C, Fin = a.Fin, ldr = C + Fin;
    const int per_gate = C * 2 * C + C + Fin * C + C;
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= 3 * per_gate) return;
    const int g = gid / per_gate;
    int i = gid - g * per_gate;
    const float *__restrict__ R = a.R[g], *__restrict__ cs = a.cs[g], *__restrict__ Wc = a.Wc[g], *__restrict__ bc = a.bc[g],
                *__restrict__ Wg = a.Wg[g];
    // (the sums keep their index order; the loads of eight terms are issued together -- one dependent load + add per term made this
    // 25 us for a few hundred KB)
    auto dot = [&](float v, const float *__restrict__ p, int sp, const float *__restrict__ q, int sq, int n) {
        int k = 0;
        for (; k + 8 <= n; k += 8) {
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = p[(k + u) * sp], y[u] = q[(k + u) * sq];
#pragma unroll
            for (int u = 0; u < 8; ++u) v = v + x[u] * y[u];
        }
        for (; k < n; ++k) v = v + p[k * sp] * q[k * sq];
        return v;
    };
    if (i < C * 2 * C) {
        const int o = i / (2 * C), col = i - o * 2 * C;
        // torch.addmm(outer(cs, bc), MgT, Wc): the outer product first
        a.dWg[g][i] = col < C ? dot(cs[o] * bc[col], R + o * ldr + C, 1, Wc + col, C, Fin) : R[o * ldr + (col - C)];
    } else if ((i -= C * 2 * C) < C) {
        a.dbg[g][i] = cs[i];
    } else if ((i -= C) < Fin * C) {
        const int f = i / C, col = i - f * C;
        a.dWc[g][i] = dot(0.f, R + C + f, ldr, Wg + col, 2 * C, C);
    } else {
        i -= Fin * C;
        a.dbc[g][i] = dot(0.f, cs, 1, Wg + i, 2 * C, C);
    }
}

// ---- the other direction, once per window: the gate Linears with the conv folded in (stg_tgcn_fold_weights) ----------------------
//   w_fold[g C + c][f]       = sum_k Wg[c][k] Wc_g[f][k]   (f < Fin: the P part of [P | Hx] Ag^T)
//   w_fold[g C + c][Fin + j] = Wg[c][C + j]
//   b_fold[g C + c]          = sum_k Wg[c][k] bc_g[k] + bg[c]
//   bound = {max |Wc_g|, max |bc_g|} over the three gates (what a step launch that does not form x3 bounds it with)
struct FoldArgs {
    const float *Wc[3], *bc[3], *Wg[3], *bg[3];
    float *w_fold, *b_fold, *bound, *w_fold_t;
    int C, Fin;
};

__global__ __launch_bounds__(kBlock) void tgcn_fold_kernel(const FoldArgs a)
{
    const int C = a.C, Fin = a.Fin, K = Fin + C;
    const int total = 3 * C * K + 3 * C;
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid < 3 * C * K) {
        const int row = gid / K, col = gid - row * K, g = row / C, c = row - g * C;
        const float *__restrict__ Wg = a.Wg[g] + c * 2 * C;
        float v;
        if (col < Fin) {
            const float *__restrict__ Wc = a.Wc[g] + col * C;
            v = 0.f;
            for (int k = 0; k < C; k += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(Wg + k), y = *reinterpret_cast<const float4 *>(Wc + k);
                v = v + x.x * y.x;
                v = v + x.y * y.y;
                v = v + x.z * y.z;
                v = v + x.w * y.w;
            }
            if (a.w_fold_t) a.w_fold_t[(g * Fin + col) * C + c] = v;
        } else {
            v = Wg[C + (col - Fin)];
        }
        a.w_fold[gid] = v;
    } else if (gid < total) {
        const int row = gid - 3 * C * K, g = row / C, c = row - g * C;
        const float *__restrict__ Wg = a.Wg[g] + c * 2 * C, *__restrict__ bc = a.bc[g];
        float v = a.bg[g][c];
        for (int k = 0; k < C; ++k) v = v + Wg[k] * bc[k];
        a.b_fold[row] = v;
    }
    // the two maxima: every thread brings one element (|.| >= 0: the float's bits order like unsigned integers), a wave its maximum
    // with one atomic (bound was zeroed by the launcher)
    {
        float mw = 0.f, mb = 0.f;
        if (gid < 3 * Fin * C) mw = fabsf(a.Wc[gid / (Fin * C)][gid % (Fin * C)]);
        if (gid < 3 * C) mb = fabsf(a.bc[gid / C][gid % C]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mw = fmaxf(mw, __shfl_xor(mw, o, kWave));
            mb = fmaxf(mb, __shfl_xor(mb, o, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            if (mw > 0.f) atomicMax(reinterpret_cast<unsigned *>(a.bound), __float_as_uint(mw));
            if (mb > 0.f) atomicMax(reinterpret_cast<unsigned *>(a.bound) + 1, __float_as_uint(mb));
        }
    }
}

}  // namespace
}  // namespace stg

extern "C" int stg_tgcn_fold_weights(const float *const *Wc, const float *const *bc, const float *const *Wg, const float *const *bg,
                                     float *w_fold, float *b_fold, float *bound, float *w_fold_t, int32_t C, int32_t Fin, void *stream)
{
    using namespace stg;
    if (C <= 0 || Fin <= 0 || C % 4 != 0 || C > 1024 || Fin > 1024)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_fold_weights: bad shape C=%d Fin=%d (C %% 4 == 0)", C, Fin);
    if (!Wc || !bc || !Wg || !bg || !w_fold || !b_fold || !bound)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_fold_weights: NULL pointer argument");
    FoldArgs a{};
    for (int g = 0; g < 3; ++g) {
        if (!Wc[g] || !bc[g] || !Wg[g] || !bg[g]) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_fold_weights: NULL pointer for gate %d", g);
        if ((reinterpret_cast<uintptr_t>(Wc[g]) | reinterpret_cast<uintptr_t>(Wg[g])) & 15)
            return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_fold_weights: weights must be 16-byte aligned");
        a.Wc[g] = Wc[g]; a.bc[g] = bc[g]; a.Wg[g] = Wg[g]; a.bg[g] = bg[g];
    }
    a.w_fold = w_fold; a.b_fold = b_fold; a.bound = bound; a.w_fold_t = w_fold_t; a.C = C; a.Fin = Fin;
    const int total = 3 * C * (Fin + C) + 3 * C;
    if (const int rc = zero_async(bound, 2 * sizeof(float), static_cast<hipStream_t>(stream))) return rc;
    hipLaunchKernelGGL(tgcn_fold_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("stg_tgcn_fold_weights");
}

extern "C" int stg_tgcn_unfold_gate_grads(const float *const *R, const float *const *cs, const float *const *Wc,
                                          const float *const *bc, const float *const *Wg, float *const *dWg, float *const *dbg,
                                          float *const *dWc, float *const *dbc, int32_t C, int32_t Fin, void *stream)
{
    using namespace stg;
    if (C <= 0 || Fin <= 0 || C > 1024 || Fin > 1024)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_unfold_gate_grads: bad shape C=%d Fin=%d", C, Fin);
    if (!R || !cs || !Wc || !bc || !Wg || !dWg || !dbg || !dWc || !dbc)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_unfold_gate_grads: NULL pointer table");
    UnfoldArgs a{};
    for (int g = 0; g < 3; ++g) {
        if (!R[g] || !cs[g] || !Wc[g] || !bc[g] || !Wg[g] || !dWg[g] || !dbg[g] || !dWc[g] || !dbc[g])
            return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_unfold_gate_grads: NULL pointer for gate %d", g);
        a.R[g] = R[g]; a.cs[g] = cs[g]; a.Wc[g] = Wc[g]; a.bc[g] = bc[g]; a.Wg[g] = Wg[g];
        a.dWg[g] = dWg[g]; a.dbg[g] = dbg[g]; a.dWc[g] = dWc[g]; a.dbc[g] = dbc[g];
    }
    a.C = C; a.Fin = Fin;
    const int total = 3 * (C * 2 * C + C + Fin * C + C);
    hipLaunchKernelGGL(tgcn_unfold_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("stg_tgcn_unfold_gate_grads");
}
