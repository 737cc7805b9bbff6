// Fused aggregate-then-transform:  out = (A_hat X) W   in ONE kernel  (SURVEY.md 8(f) rank 1).
//
//   P[r,:]  = norm_row[r] * sum_{e in row r} (nc[e] * x[col[e],:]) * w[e]        (as gcn_agg_kernel)
//   out[r,:] = P[r,:] W                                                          (fp32 MFMA)
//
// A GCN layer computes A_hat (X W); aggregation is linear, so (A_hat X) W is the same map and when
// F_in < F_out the gather runs at the NARROW width: TGCN's fused gates gather 32 floats per edge
// instead of 192 (6x less gather traffic), and the [N, F_out] intermediate X W is never written or
// re-read.  Rounding differs from the reference's order (A_hat applied after W) at the 1e-7 level;
// parity is 1e-4 (tests/test_gpu_agg_transform.py).
//
// Workgroup = 256 threads = 64 CSR rows.
//   phase 1  each wave aggregates its rows exactly like gcn_agg_kernel (G lanes per row, 16-B row
//            gathers, UNROLL in flight, CSR order, one accumulator per feature) and drops the
//            scaled row into an LDS tile Ps[64][F_in + 1] (the +1 pad makes the column reads of
//            phase 2 conflict-free); W [F_in, F_out] is staged into LDS once per workgroup.
//   phase 2  v_mfma_f32_32x32x2_f32 over the tile: A fragment = Ps[row = l&31][k + (l>>5)] (LDS),
//            B fragment = Ws[k + (l>>5)][n0 + (l&31)] (LDS, bank = column), 32x32 output tiles
//            dealt round-robin to the 4 waves; results stored as 128-B row segments.
// LDS = 4 (64 (F_in+1) + F_in F_out) bytes: 33 KB at 32 -> 192, so 4 workgroups (16 waves) per CU.
// Optionally P itself is written out ([N, F_in]): the backward pass needs it for dW = P^T dOut.
#include "stg_common.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kXwRowsMax = 64;
constexpr bool kXwWide = false;      // measured (TGCN cfg4, Fin = 32): 8-wave workgroups 39 us vs 35 us with 4

// WAVES per workgroup: 4, or 8 when a row takes 8+ lanes (then 4 waves would need several passes over the 64-row
// tile, one after the other; 8 waves halve that chain and put twice the gathers in flight per tile).
// kXwRows rows per workgroup: 64, or 32 ("xw_rows") -- twice the workgroups at half the tile each.
template <int LOG2G, bool HAS_EW, int WAVES, int kXwRows = 64>
__global__ __launch_bounds__(WAVES * kWave) void gcn_agg_xw_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, const float *__restrict__ W, float *__restrict__ out,
    float *__restrict__ P_out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ node_ids, int N, int Fin, int Fout)
{
    constexpr int G = 1 << LOG2G;
    constexpr int VEC = 4;
    constexpr int ROWS_PER_WAVE = kWave / G;
    constexpr int ROWS_PER_PASS = ROWS_PER_WAVE * WAVES;
    constexpr int PASSES = kXwRows / ROWS_PER_PASS > 0 ? kXwRows / ROWS_PER_PASS : 1;
    constexpr int NT = WAVES * kWave;
    constexpr int R = G < 16 ? 16 : G, I = R / G, U = 8;
    extern __shared__ float lds[];
    const int ldp = Fin + 1;
    float *Ps = lds;                                  // [kXwRows][Fin + 1]
    float *Ws = lds + kXwRows * ldp;                  // [Fin][Fout]

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int j = lane & (G - 1);
    const int row_base = blockIdx.x * kXwRows;

    // stage W (read once per workgroup, L2 resident)
    // (8 loads in flight per thread before the first LDS write: one round trip per 8 NT float4s instead of one each)
    for (int base = threadIdx.x * 4; base < Fin * Fout; base += 8 * NT * 4) {
        float4 w4[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = base + s * NT * 4;
            w4[s] = i < Fin * Fout ? *reinterpret_cast<const float4 *>(W + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = base + s * NT * 4;
            if (i < Fin * Fout) *reinterpret_cast<float4 *>(Ws + i) = w4[s];
        }
    }

    // ---- phase 1: aggregate into the LDS tile
    const int foff = j * VEC;
    const bool fok = foff < Fin;
    for (int pass = 0; pass < PASSES; ++pass) {
        const int lr = pass * ROWS_PER_PASS + wave * ROWS_PER_WAVE + (lane >> LOG2G);   // row inside the tile
        const int idx = row_base + lr;
        const bool row_valid = idx < N && lr < kXwRows;
        int r = 0, beg = 0, deg = 0;
        float nr = 0.f;
        if (row_valid) {
            r = node_ids ? node_ids[idx] : idx;
            beg = row_offsets[r];
            deg = row_offsets[r + 1] - beg;
            nr = norm_row[r];
        }
        const int max_deg = wave_max_nonneg(deg);
        float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
        // One round = R edges per row (lane j holds edges j, j + G, ...): at least 16, so that rows of typical
        // degree take ONE index round trip; the gathers go U at a time, later batches under a wave-uniform guard.
        for (int base = 0; base < max_deg; base += R) {
            const int cnt = deg - base;
            const int cnt_max = min(R, max_deg - base);
            int c[I];
            float nc[I], w[I];
#pragma unroll
            for (int i = 0; i < I; ++i) {
                c[i] = 0;
                nc[i] = 0.f;
                w[i] = 1.f;
                if (i * G + j < cnt) {
                    const int e = beg + base + i * G + j;
                    c[i] = column_indices[e];
                    nc[i] = nc_edge[e];
                    if constexpr (HAS_EW) w[i] = ew_edge[e];
                }
            }
#pragma unroll
            for (int k = 0; k < R; k += U) {
                if (k < cnt_max) {
                    float v[U][VEC];
                    float ncs[U], ws[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int kk = k + u, el = kk >> LOG2G, src = kk & (G - 1);
                        const int ck = __shfl(c[el], src, G);
                        ncs[u] = __shfl(nc[el], src, G);
                        if constexpr (HAS_EW) ws[u] = __shfl(w[el], src, G);
                        if (kk < cnt && fok) {
                            vec_load<VEC>(v[u], x + (int64_t)ck * Fin + foff);
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (k + u < cnt) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                float t = ncs[u] * v[u][i];
                                if constexpr (HAS_EW) t = t * ws[u];
                                acc[i] = acc[i] + t;
                            }
                        }
                    }
                }
            }
        }
        if (fok && lr < kXwRows) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) Ps[lr * ldp + foff + i] = acc[i] * nr;       // rows >= N hold zeros
            if (P_out && row_valid) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = acc[i] * nr;
                vec_store<VEC>(P_out + (int64_t)r * Fin + foff, o);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: out tile = Ps (64 x Fin) * Ws (Fin x Fout) on the fp32 matrix cores
    const int kh = lane >> 5, l31 = lane & 31;
    const int col_tiles = Fout / 32;
    const int tiles = (kXwRows / 32) * col_tiles;
    for (int t = wave; t < tiles; t += WAVES) {
        const int rt = t / col_tiles, ct = t - rt * col_tiles;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float *pa = Ps + (rt * 32 + l31) * ldp + kh;
        const float *pb = Ws + kh * Fout + ct * 32 + l31;
#pragma unroll 4
        for (int k = 0; k < Fin; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k * Fout], acc, 0, 0, 0);
        // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            const int idx = row_base + lr;
            if (idx < N) {
                const int r = node_ids ? node_ids[idx] : idx;
                out[(int64_t)r * Fout + ct * 32 + l31] = acc[i];
            }
        }
    }
}

}  // namespace stg

extern "C" int stg_gcn_agg_transform(const float *x, const float *norm_row, const float *norm_col_edge,
                                     const float *ew_edge, const float *W, float *out, float *P_out,
                                     const int32_t *row_offsets, const int32_t *column_indices,
                                     const int32_t *node_ids, int32_t N, int32_t Fin, int32_t Fout, void *stream)
{
    using namespace stg;
    if (N < 0 || Fin <= 0 || Fout <= 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gcn_agg_transform: bad shape N=%d Fin=%d Fout=%d", N, Fin, Fout);
    if (Fin % 4 != 0 || Fin < 16 || Fin > 256 || Fout % 32 != 0)
        return fail(STG_ERR_UNSUPPORTED,
                    "stg_gcn_agg_transform: needs Fin %% 4 == 0, 16 <= Fin <= 256, Fout %% 32 == 0 (got %d -> %d)", Fin, Fout);
    // measured (tools/microbench_xw.py): N = 1M / E = 16M: 456 -> 423 us with 32-row workgroups, N = 50 K: equal
    const int rows = tuning().xw_rows == 32 || (tuning().xw_rows == 0 && N >= 400000) ? 32 : 64;
    const size_t lds = sizeof(float) * ((size_t)rows * (Fin + 1) + (size_t)Fin * Fout);
    // up to 64 KB without asking; larger weights (one workgroup per CU then: measured slower than GEMM + aggregation at
    // 128 -> 128, profiles/r02_agg_transform_128.jsonl, so the layer never picks it) need the limit raised per kernel
    if (lds > 150 * 1024)
        return fail(STG_ERR_UNSUPPORTED, "stg_gcn_agg_transform: W (%d x %d) does not fit the LDS", Fin, Fout);
    if (N == 0) return 0;
    if (!x || !norm_row || !W || !out || !row_offsets)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gcn_agg_transform: NULL pointer argument");
    const uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) |
                            reinterpret_cast<uintptr_t>(P_out);
    if (align % 16 != 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gcn_agg_transform: operands must be 16-byte aligned");
    const int lanes = Fin / 4;
    const int log2g = ilog2_ceil(lanes);
    const unsigned blocks = (unsigned)(((int64_t)N + rows - 1) / rows);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define STG_XW_RAISE(KERN)                                                                                     \
    if (lds > 64 * 1024) {                                                                                     \
        const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(KERN),                       \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        if (e_ != hipSuccess) return fail((int)e_, "stg_gcn_agg_transform: %s", hipGetErrorString(e_));        \
    }
#define STG_XW_(LG, WVS)                                                                                           \
    {                                                                                                          \
        constexpr int WV = WVS;                                                                                \
        STG_XW_RAISE((gcn_agg_xw_kernel<LG, true, WV, 32>))                                                    \
        STG_XW_RAISE((gcn_agg_xw_kernel<LG, false, WV, 32>))                                                   \
        STG_XW_RAISE((gcn_agg_xw_kernel<LG, true, WV>))                                                        \
        STG_XW_RAISE((gcn_agg_xw_kernel<LG, false, WV>))                                                       \
        if (rows == 32) {                                                                                      \
            if (ew_edge)                                                                                       \
                hipLaunchKernelGGL((gcn_agg_xw_kernel<LG, true, WV, 32>), dim3(blocks), dim3(WV * kWave), lds, st, x, \
                                   norm_row, norm_col_edge, ew_edge, W, out, P_out, row_offsets, column_indices, \
                                   node_ids, N, Fin, Fout);                                                    \
            else                                                                                               \
                hipLaunchKernelGGL((gcn_agg_xw_kernel<LG, false, WV, 32>), dim3(blocks), dim3(WV * kWave), lds, st, x, \
                                   norm_row, norm_col_edge, ew_edge, W, out, P_out, row_offsets, column_indices, \
                                   node_ids, N, Fin, Fout);                                                    \
        } else if (ew_edge)                                                                                    \
            hipLaunchKernelGGL((gcn_agg_xw_kernel<LG, true, WV>), dim3(blocks), dim3(WV * kWave), lds, st, x,  \
                               norm_row, norm_col_edge, ew_edge, W, out, P_out, row_offsets, column_indices,   \
                               node_ids, N, Fin, Fout);                                                        \
        else                                                                                                   \
            hipLaunchKernelGGL((gcn_agg_xw_kernel<LG, false, WV>), dim3(blocks), dim3(WV * kWave), lds, st, x, \
                               norm_row, norm_col_edge, ew_edge, W, out, P_out, row_offsets, column_indices,   \
                               node_ids, N, Fin, Fout);                                                        \
    }
    // 8 waves per workgroup where 4 would need several passes over the tile (rows of 8+ lanes); "xw_waves" forces
#define STG_XW(LG)                                                   \
    if (tuning().xw_waves == 8 || (tuning().xw_waves == 0 && kXwWide && LG >= 3)) STG_XW_(LG, 8) else STG_XW_(LG, 4)
    switch (log2g) {
        case 0: STG_XW(0); break;
        case 1: STG_XW(1); break;
        case 2: STG_XW(2); break;
        case 3: STG_XW(3); break;
        case 4: STG_XW(4); break;
        case 5: STG_XW(5); break;
        default: STG_XW(6); break;
    }
#undef STG_XW
#undef STG_XW_
#undef STG_XW_RAISE
    return check_launch("stg_gcn_agg_transform");
}
