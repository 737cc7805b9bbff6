// Fused GCN aggregation for gfx950 (MI355X): CSR-driven neighbour gather +
// (edge-)weighted sum + row scaling in one pass.  One kernel serves the forward
// unit (dst-major CSR) and the backward unit (src-major CSR) that the reference
// emits per GCNConv (stgraph_hip.h: stg_gcn_agg).
//
// Mapping (wave64):
//   * G = 2^LOG2G consecutive lanes own one CSR row; a wave carries 64/G rows.
//     Each lane owns VEC contiguous features per chunk (VEC*4 B per load: 16 B
//     when F % 4 == 0), so one row gather is a single fully coalesced
//     G*VEC*4-byte request per chunk.
//   * The row's column indices / per-edge coefficients are fetched G at a time,
//     one per lane (coalesced), then broadcast lane -> row-group with v_readlane
//     (G = 64, the index lands in an SGPR and the row base address becomes
//     scalar) or ds_bpermute (G < 64).
//   * UNROLL row gathers are issued back to back before the first is consumed,
//     so each wave keeps UNROLL * 64/G rows in flight; the accumulation itself
//     stays strictly in CSR order with ONE fp32 accumulator per (row, feature)
//     and no FMA contraction (-ffp-contract=off), i.e. bit-identical to the
//     reference's sequential loop.
//   * All loop bounds are wave-uniform (SGPR); per-row raggedness is handled by
//     predication, so no lane ever leaves a shuffle early.
//
// PRE (stg_gcn_agg_edge): norm[col[e]] and w[eid[e]] arrive already gathered into
// CSR order (stg_edge_gather_f32, done once per graph by the caller).  The two
// scattered 4-byte gathers per edge -- each of which costs a whole 64-B sector of
// fabric traffic and a dependent round trip -- become coalesced streams; the
// values, and therefore the results, are identical.
#include "stg_common.hpp"

namespace stg {

template <int G>
__device__ __forceinline__ int bcast_i(int v, int src)
{
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, src);
    else if constexpr (G == 1) return v;
    else return __shfl(v, src, G);
}

template <int G>
__device__ __forceinline__ float bcast_f(float v, int src)
{
    return __int_as_float(bcast_i<G>(__float_as_int(v), src));
}

template <int VEC, int LOG2G, int CHUNKS, bool HAS_EW, bool PRE, int UNROLL, bool EPI = false>
__global__ __launch_bounds__(kBlock) void gcn_agg_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row,
    const float *__restrict__ norm_col, const float *__restrict__ ew, float *__restrict__ out,
    const int *__restrict__ row_offsets, const int *__restrict__ column_indices,
    const int *__restrict__ eids, const int *__restrict__ node_ids, int N, int F, int F_active,
    const float *__restrict__ bias, int act)
{
    constexpr int G = 1 << LOG2G;
    constexpr int ROWS_PER_WAVE = kWave / G;
    constexpr int U = UNROLL < G ? UNROLL : G;

    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & (G - 1);
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int idx = wave_global * ROWS_PER_WAVE + (lane >> LOG2G);
    const bool row_valid = idx < N;

    int r = 0, beg = 0, deg = 0;
    float nr = 0.f;
    if (row_valid) {
        r = node_ids ? node_ids[idx] : idx;
        beg = row_offsets[r];
        deg = row_offsets[r + 1] - beg;
        nr = norm_row[r];
    }
    const int max_deg = __builtin_amdgcn_readfirstlane(wave_max(deg));

    for (int fbase = 0; fbase < F_active; fbase += G * VEC * CHUNKS) {
        float acc[CHUNKS][VEC];
        int foff[CHUNKS];
        bool fok[CHUNKS];
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            foff[ch] = fbase + (ch * G + j) * VEC;
            fok[ch] = foff[ch] < F_active;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[ch][i] = 0.f;
        }

        for (int base = 0; base < max_deg; base += G) {
            const int cnt = deg - base;                       // edges left in MY row (may be <= 0)
            const int cnt_max = min(G, max_deg - base);       // wave-uniform
            int c = 0;
            float nc = 0.f, w = 1.f;
            if (j < cnt) {
                const int e = beg + base + j;
                c = column_indices[e];
                if constexpr (PRE) {
                    nc = norm_col[e];                         // = norm[col[e]], gathered once per graph
                    if constexpr (HAS_EW) w = ew[e];          // = w[eid[e]]
                } else {
                    nc = norm_col[c];
                    if constexpr (HAS_EW) w = ew[eids[e]];
                }
            }
            for (int k = 0; k < cnt_max; k += U) {
                float v[U][CHUNKS][VEC];
                float ncs[U], ws[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ck = bcast_i<G>(c, kk & (G - 1));
                    ncs[u] = bcast_f<G>(nc, kk & (G - 1));
                    if constexpr (HAS_EW) ws[u] = bcast_f<G>(w, kk & (G - 1));
                    const float *row = x + (int64_t)ck * F;
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if (kk < cnt && fok[ch]) {
                            vec_load<VEC>(v[u][ch], row + foff[ch]);
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[u][ch][i] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < cnt) {
#pragma unroll
                        for (int ch = 0; ch < CHUNKS; ++ch) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                float t = ncs[u] * v[u][ch][i];          // Mul(norm_inb, h_inb)
                                if constexpr (HAS_EW) t = t * ws[u];     // Mul(., edge_weight)
                                acc[ch][i] = acc[ch][i] + t;             // AggSum
                            }
                        }
                    }
                }
            }
        }

        if (row_valid) {
            float *orow = out + (int64_t)r * F;
#pragma unroll
            for (int ch = 0; ch < CHUNKS; ++ch) {
                if (fok[ch]) {
                    float o[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] = acc[ch][i] * nr;   // Mul(., norm_cen)
                    if constexpr (EPI) {   // layer epilogue (gcn_conv.py:185-188), wave-uniform switches: + bias, ReLU
                        if (bias) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) o[i] = o[i] + bias[foff[ch] + i];
                        }
                        if (act == STG_ACT_RELU) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) o[i] = o[i] < 0.f ? 0.f : o[i];
                        }
                    }
                    vec_store<VEC>(orow + foff[ch], o);
                }
            }
        }
    }
}

// dst[i] = table[idx[i]]
__global__ void edge_gather_kernel(float *__restrict__ dst, const float *__restrict__ table,
                                   const int *__restrict__ idx, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = table[idx[i]];
}

namespace {

struct GcnArgs {
    const float *x, *norm_row, *norm_col, *ew;
    float *out;
    const int *row_offsets, *column_indices, *eids, *node_ids;
    int N, F, F_active;
    bool pre;
    hipStream_t stream;
    int64_t E = 0;      // number of edges if the caller knows it (mapping heuristic only), else 0
    const float *bias = nullptr;   // layer epilogue: out = act(out + bias)
    int act = STG_ACT_NONE;
};

template <int VEC, int LOG2G, int CHUNKS, bool HAS_EW, bool PRE, int UNROLL, bool EPI = false>
void launch(const GcnArgs &a)
{
    constexpr int rows_per_block = (kWave >> LOG2G) * kWavesPerBlock;
    const int64_t blocks = ((int64_t)a.N + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((gcn_agg_kernel<VEC, LOG2G, CHUNKS, HAS_EW, PRE, UNROLL, EPI>), dim3((unsigned)blocks),
                       dim3(kBlock), 0, a.stream, a.x, a.norm_row, a.norm_col, a.ew, a.out,
                       a.row_offsets, a.column_indices, a.eids, a.node_ids, a.N, a.F, a.F_active, a.bias, a.act);
}

template <int VEC, int LOG2G, int CHUNKS, int UNROLL>
void launch_ew(const GcnArgs &a)
{
    if (a.pre && (a.bias || a.act != STG_ACT_NONE)) {
        if (a.ew) launch<VEC, LOG2G, CHUNKS, true, true, UNROLL, true>(a);
        else launch<VEC, LOG2G, CHUNKS, false, true, UNROLL, true>(a);
    } else if (a.pre) {
        if (a.ew) launch<VEC, LOG2G, CHUNKS, true, true, UNROLL>(a);
        else launch<VEC, LOG2G, CHUNKS, false, true, UNROLL>(a);
    } else {
        if (a.ew) launch<VEC, LOG2G, CHUNKS, true, false, UNROLL>(a);
        else launch<VEC, LOG2G, CHUNKS, false, false, UNROLL>(a);
    }
}

template <int VEC, int LOG2G, int CHUNKS>
void launch_unroll(const GcnArgs &a, int unroll)
{
    switch (unroll) {
        case 2: launch_ew<VEC, LOG2G, CHUNKS, 2>(a); break;
        case 4: launch_ew<VEC, LOG2G, CHUNKS, 4>(a); break;
        default: launch_ew<VEC, LOG2G, CHUNKS, 8>(a); break;
    }
}

template <int VEC>
void launch_vec(const GcnArgs &a, int log2g, int chunks, int unroll)
{
    if (chunks == 4) { launch_unroll<VEC, 6, 4>(a, unroll > 4 ? 4 : unroll); return; }
    if (chunks == 2) { launch_unroll<VEC, 6, 2>(a, unroll); return; }
    switch (log2g) {
        case 0: launch_ew<VEC, 0, 1, 1>(a); break;
        case 1: launch_ew<VEC, 1, 1, 2>(a); break;
        case 2: launch_ew<VEC, 2, 1, 4>(a); break;
        case 3: launch_unroll<VEC, 3, 1>(a, unroll); break;
        case 4: launch_unroll<VEC, 4, 1>(a, unroll); break;
        case 5: launch_unroll<VEC, 5, 1>(a, unroll); break;
        default: launch_unroll<VEC, 6, 1>(a, unroll); break;
    }
}

int gcn_agg_dispatch(GcnArgs a, const char *what)
{
    if (a.N < 0 || a.F <= 0 || a.F_active < 0 || a.F_active > a.F)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad shape N=%d F=%d F_active=%d", what, a.N, a.F, a.F_active);
    if (a.N == 0 || a.F_active == 0) return 0;
    // column_indices / eids / per-edge arrays may be NULL for a graph without edges
    if (!a.x || !a.norm_row || !a.out || !a.row_offsets || (!a.pre && !a.norm_col))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", what);
    if (!a.pre && a.ew && !a.eids)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: edge weights given without eids", what);
    if (a.act != STG_ACT_NONE && a.act != STG_ACT_RELU)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: unknown activation %d", what, a.act);
    if ((a.bias || a.act != STG_ACT_NONE) && (a.F_active != a.F || !a.pre))
        return fail(STG_ERR_UNSUPPORTED, "%s: the layer epilogue needs every column active and pre-gathered scalars", what);

    const uintptr_t align = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.out);
    int vec = 1;
    if (a.F % 4 == 0 && a.F_active % 4 == 0 && align % 16 == 0) vec = 4;
    else if (a.F % 2 == 0 && a.F_active % 2 == 0 && align % 8 == 0) vec = 2;

    // performance knob: force the number of lanes per row (shrinks VEC if the row is too narrow)
    const int forced = tuning().gcn_lanes_per_row;
    if (forced > 0) {
        while (vec > 1 && a.F_active / vec < forced) vec /= 2;
    } else if (vec == 4 && a.F_active == 128 && a.E >= 8 * (int64_t)a.N) {
        vec = 2;      // measured (profiles/r01): at average degree >= 8 one 512-B row per wave (G = 64,
                      // 8 B/lane, scalar row base via v_readlane) beats two rows per wave (G = 32,
                      // 16 B/lane) by ~3 %; on low-degree graphs (Cora-shaped, degree ~4) it loses 25 %
    }
    const int lanes = (a.F_active + vec - 1) / vec;
    int log2g = ilog2_ceil(lanes);
    int chunks = 1;
    if (log2g > 6) {
        log2g = 6;
        const int need = (lanes + kWave - 1) / kWave;
        chunks = need >= 3 ? 4 : 2;           // wider rows loop over super-chunks inside the kernel
    }
    const int unroll = tuning().gcn_unroll > 0 ? tuning().gcn_unroll : (log2g == 6 && chunks == 1 ? 4 : 8);
    switch (vec) {
        case 4: launch_vec<4>(a, log2g, chunks, unroll); break;
        case 2: launch_vec<2>(a, log2g, chunks, unroll); break;
        default: launch_vec<1>(a, log2g, chunks, unroll); break;
    }
    return check_launch(what);
}

}  // namespace
}  // namespace stg

extern "C" int stg_gcn_agg(const float *x, const float *norm_row, const float *norm_col,
                           const float *ew, float *out, const int32_t *row_offsets,
                           const int32_t *column_indices, const int32_t *eids,
                           const int32_t *node_ids, int32_t N, int32_t F, int32_t F_active,
                           void *stream)
{
    return stg::gcn_agg_dispatch({x, norm_row, norm_col, ew, out, row_offsets, column_indices, eids, node_ids,
                                  N, F, F_active, false, static_cast<hipStream_t>(stream)}, "stg_gcn_agg");
}

extern "C" int stg_gcn_agg_edge(const float *x, const float *norm_row, const float *norm_col_edge,
                                const float *ew_edge, float *out, const int32_t *row_offsets,
                                const int32_t *column_indices, const int32_t *node_ids, int32_t N,
                                int64_t E, int32_t F, int32_t F_active, void *stream)
{
    return stg::gcn_agg_dispatch({x, norm_row, norm_col_edge, ew_edge, out, row_offsets, column_indices, nullptr,
                                  node_ids, N, F, F_active, true, static_cast<hipStream_t>(stream), E},
                                 "stg_gcn_agg_edge");
}

extern "C" int stg_gcn_layer_fwd(const float *x, const float *norm_row, const float *norm_col_edge,
                                 const float *ew_edge, const float *bias, int32_t act, float *out,
                                 const int32_t *row_offsets, const int32_t *column_indices,
                                 const int32_t *node_ids, int32_t N, int64_t E, int32_t F, void *stream)
{
    stg::GcnArgs a{x, norm_row, norm_col_edge, ew_edge, out, row_offsets, column_indices, nullptr,
                   node_ids, N, F, F, true, static_cast<hipStream_t>(stream), E};
    a.bias = bias;
    a.act = act;
    return stg::gcn_agg_dispatch(a, "stg_gcn_layer_fwd");
}

extern "C" int stg_edge_gather_f32(float *dst, const float *table, const int32_t *idx, int64_t n,
                                   void *stream)
{
    using namespace stg;
    if (n < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_edge_gather_f32: negative size");
    if (n == 0) return 0;
    if (!dst || !table || !idx) return fail(STG_ERR_INVALID_ARGUMENT, "stg_edge_gather_f32: NULL pointer argument");
    const int blocks = (int)std::min<int64_t>((n + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(edge_gather_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), dst,
                       table, idx, n);
    return check_launch("stg_edge_gather_f32");
}
