/*
 * stg_oracle.c -- CPU restatement of STGraph's Seastar hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under stgraph_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU
 * baseline, never as the product.
 *
 * Every function restates one piece of the reference (paths relative to
 * /root/reference/stgraph) loop-for-loop, with the reference's summation
 * order: ONE fp32 accumulator per (vertex, feature), edges visited in CSR row
 * order.  Build with -ffp-contract=off so a*b+c is never fused: the golden
 * vectors were produced by the reference's own emitted kernels compiled for
 * x86-64 without FMA (tests/golden/make_golden.py).
 *
 * Pinning: see oracle/README.md.  The emitted-kernel restatements (gcn_agg,
 * gat_*) are checked against golden vectors generated from the reference's
 * Python compiler stack; csr.cu itself is NOT buildable in this image (needs
 * cuda_runtime.h, thrust, cub), so orc_csr_ctor is pinned by the reference's
 * Python callers (static_graph.py / naive_graph.py run unmodified on top of
 * it in make_golden.py) plus the CSR example recorded in SURVEY.md 8(c).
 *
 * OpenMP (optional, -fopenmp) parallelises over rows only; every (row,
 * feature) sum stays sequential, so results do not depend on thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * CSR::CSR  -- graph/static/csr.cu:68-157
 *
 * Input: E triples (a[i], b[i], eid[i]) in the order the Python caller
 * prepared them (static_graph.py:65-78), edge weights indexed BY EID
 * (csr.cu:126), num_nodes, is_edge_reverse.
 * Output: row_offset[N+1], column_indices[E], eids[E], node_ids[N],
 * in_degrees[N], out_degrees[N], weighted_out_degrees[N].
 * ---------------------------------------------------------------------- */
typedef struct { int deg; int id; } orc_deg_id;

static int orc_cmp_deg_desc(const void *pa, const void *pb)
{
    const orc_deg_id *a = (const orc_deg_id *)pa, *b = (const orc_deg_id *)pb;
    /* csr.cu:148-150 compares degree only (std::sort, tie order unspecified);
       ties are broken by ascending id here so the oracle is deterministic. */
    if (a->deg != b->deg) return (a->deg > b->deg) ? -1 : 1;
    return (a->id < b->id) ? -1 : (a->id > b->id);
}

int orc_csr_ctor(const int *a, const int *b, const int *eid_in, const float *edge_weight,
                 int64_t E, int N, int is_edge_reverse,
                 int *row_offset, int *column_indices, int *eids, int *node_ids,
                 int *in_degrees, int *out_degrees, float *weighted_out_degrees)
{
    /* csr.cu:83-90 */
    for (int i = 0; i < N; ++i) { in_degrees[i] = 0; out_degrees[i] = 0; weighted_out_degrees[i] = 0.f; }
    for (int i = 0; i <= N; ++i) row_offset[i] = -1;
    row_offset[0] = 0;

    int current_src = 0; /* csr.cu:91 leaves this uninitialised for E==0 (UB) */
    int beg = 0, end = 0;

    /* csr.cu:96-127 */
    for (int64_t i = 0; i < E; ++i) {
        int src = is_edge_reverse ? b[i] : a[i];
        int dst = is_edge_reverse ? a[i] : b[i];
        int eid = eid_in[i];
        if (src < 0 || src >= N || dst < 0 || dst >= N) return -1;
        if (beg == 0 && end == 0) current_src = src;
        if (current_src != src) {
            row_offset[current_src] = beg;
            row_offset[current_src + 1] = end;
            current_src = src;
            beg = end;
        }
        column_indices[i] = dst;
        eids[i] = eid;
        end += 1;
        out_degrees[src] += 1;
        in_degrees[dst] += 1;
        weighted_out_degrees[src] += edge_weight ? edge_weight[eid] : 1.0f;
    }
    /* csr.cu:129 */
    if (E > 0) row_offset[current_src + 1] = end;

    /* csr.cu:131-140: replace the -1 gaps */
    int curr_val = row_offset[0];
    for (int i = 1; i <= N; ++i) {
        if (row_offset[i] != curr_val && row_offset[i] != -1) curr_val = row_offset[i];
        if (row_offset[i] == -1) row_offset[i] = curr_val;
    }

    /* csr.cu:142-154: node ids by descending (out) degree */
    orc_deg_id *p = (orc_deg_id *)malloc(sizeof(orc_deg_id) * (size_t)(N > 0 ? N : 1));
    if (!p) return -2;
    for (int i = 0; i < N; ++i) { p[i].deg = out_degrees[i]; p[i].id = i; }
    qsort(p, (size_t)N, sizeof(orc_deg_id), orc_cmp_deg_desc);
    for (int i = 0; i < N; ++i) node_ids[i] = p[i].id;
    free(p);
    return 0;
}

/* ------------------------------------------------------------------------
 * Launch geometry -- compiler/execution_unit.py:92-106
 * Number of feature columns the reference's FA kernel actually computes
 * (SURVEY Appendix A, D1): for feat_size < 64 only the largest power of two
 * <= feat_size, because thrs_per_group lanes stride by blockDim.x = 64.
 * ---------------------------------------------------------------------- */
int orc_ref_active_columns(int feat_size)
{
    if (feat_size >= 64) return feat_size;
    int ub = 64;
    while (ub > feat_size) ub /= 2;     /* first_pow2_less_than_n, execution_unit.py:113-116 */
    return ub < 1 ? 1 : ub;
}

/* ------------------------------------------------------------------------
 * Emitted GCN units K0/K1 (with and without edge weight)
 *   tracer        nn/pytorch/static/gcn_conv.py:162-182
 *   template      compiler/code_gen/templates/fa/tpl_fa_csr{,_unsorted}.jinja
 *   statements    compiler/registry.py:255-293 (Mul, AggSum)
 *   listing       SURVEY.md Appendix B.1 / B.2
 *
 *   out[r,f] = norm_row[r] * sum_{e in row r} ((norm_col[c] * x[c,f]) * w[eid[e]])
 *
 * forward : CSR = dst-major, norm_row = norm_col = norm          (K0)
 * backward: CSR = src-major, x = grad_out                         (K1)
 * node_ids != NULL reproduces the 'csr' template (row = node_ids[i]).
 * Columns >= F_active are left untouched (caller zero-fills, as
 * executor.py:293-307 does).
 * ---------------------------------------------------------------------- */
void orc_gcn_agg(const float *x, const float *norm_row, const float *norm_col, const float *ew,
                 float *out,
                 const int *row_offsets, const int *column_indices, const int *eids,
                 const int *node_ids, int N, int F, int F_active)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < N; ++idx) {
        const int r = node_ids ? node_ids[idx] : idx;
        const int beg = row_offsets[r], end = row_offsets[r + 1];
        for (int tx = 0; tx < F_active; ++tx) {
            float acc = 0.f;
            for (int e = beg; e < end; ++e) {
                const int c = column_indices[e];
                float t = norm_col[c] * x[(int64_t)c * F + tx];
                if (ew) t = t * ew[eids[e]];
                acc += t;
            }
            out[(int64_t)r * F + tx] = acc * norm_row[r];
        }
    }
}

/* ------------------------------------------------------------------------
 * Emitted GAT forward unit K0 -- nn/pytorch/static/gat_conv.py:48-53,
 * registry.py:195-252, SURVEY.md Appendix B.3.  max_dims = [H,1].
 *   s = el[u,h] + er[v,h];  z = s - s  (python max() over a 1-element list,
 *   SURVEY D2);  a = exp(z > 0 ? z : slope*z);  A[eid,h] = a;  S[v,h] = sum a
 * ---------------------------------------------------------------------- */
void orc_gat_k0(const float *el, const float *er, float *A, float *S,
                const int *row_offsets, const int *column_indices, const int *eids,
                const int *node_ids, int N, int H, int H_active, float slope)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < N; ++idx) {
        const int v = node_ids ? node_ids[idx] : idx;
        const int beg = row_offsets[v], end = row_offsets[v + 1];
        for (int tx = 0; tx < H_active; ++tx) {
            float acc = 0.f;
            for (int e = beg; e < end; ++e) {
                const int u = column_indices[e];
                const int eid = eids[e];
                const float s = el[(int64_t)u * H + tx] + er[(int64_t)v * H + tx];
                const volatile float sv = s;          /* keep s - s literal (NaN/inf propagate) */
                const float z = sv - sv;
                const float l = z > 0 ? z : slope * z;
                const float a = expf(l);
                A[(int64_t)eid * H + tx] = a;
                acc += a;
            }
            S[(int64_t)v * H + tx] = acc;
        }
    }
}

/* K1: out[v,h,d] = sum_e (A[eid,h] / S[v,h]) * feat[u,h,d]   (Appendix B.3) */
void orc_gat_k1(const float *A, const float *S, const float *feat, float *out,
                const int *row_offsets, const int *column_indices, const int *eids,
                const int *node_ids, int N, int H, int D, int HD_active)
{
    const int HD = H * D;
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < N; ++idx) {
        const int v = node_ids ? node_ids[idx] : idx;
        const int beg = row_offsets[v], end = row_offsets[v + 1];
        for (int tx = 0; tx < HD_active; ++tx) {
            const int h = tx / D;
            float acc = 0.f;
            for (int e = beg; e < end; ++e) {
                const int u = column_indices[e];
                const float alpha = A[(int64_t)eids[e] * H + h] / S[(int64_t)v * H + h];
                acc += alpha * feat[(int64_t)u * HD + tx];
            }
            out[(int64_t)v * HD + tx] = acc;
        }
    }
}

/* K2 (SrcParallel on the backward CSR) -- SURVEY.md Appendix B.3.
 * grad_el / grad_er must be zero on entry (executor.py:309-322); the
 * reference accumulates them with atomicAdd, here sequentially in
 * (row, tx, edge) order.  Not parallelised: the grad_er scatter races. */
void orc_gat_bwd(const float *A, const float *S, const float *out, const float *g,
                 const float *el, const float *er, const float *feat,
                 float *grad_feat, float *grad_el, float *grad_er,
                 const int *row_offsets, const int *column_indices, const int *eids,
                 const int *node_ids, int N, int H, int D, int HD_active, float slope)
{
    const int HD = H * D;
    for (int idx = 0; idx < N; ++idx) {
        const int u = node_ids ? node_ids[idx] : idx;
        const int beg = row_offsets[u], end = row_offsets[u + 1];
        for (int tx = 0; tx < HD_active; ++tx) {
            const int h = tx / D;
            float a29 = 0.f, a13 = 0.f;
            for (int e = beg; e < end; ++e) {
                const int v = column_indices[e];
                const int eid = eids[e];
                const float V3 = A[(int64_t)eid * H + h];
                const float V4 = S[(int64_t)v * H + h];
                const float V8 = g[(int64_t)v * HD + tx];
                const float V5 = V3 / V4;
                const float V12 = V8 * V5;
                a13 += V12;
                const volatile float V0 = el[(int64_t)u * H + h] + er[(int64_t)v * H + h];
                const float V1 = V0 - V0;
                const float V10 = V8 * feat[(int64_t)u * HD + tx];
                const float V14 = 1.0f / V4;
                const float V15 = V10 * V14;
                const float V16 = V8 / V4;
                const float V17 = V16 * out[(int64_t)v * HD + tx];
                const float V18 = -1.0f * V17;
                const float V22 = V15 + V18;
                const float V23 = V22 * V3;
                const float V24 = V1 > 0 ? 1.0f : slope;
                const float V25 = V23 * V24;
                a29 += V25;
                grad_er[(int64_t)v * H + h] += V25;
            }
            grad_el[(int64_t)u * H + h] += a29;
            grad_feat[(int64_t)u * HD + tx] = a13;
        }
    }
}

/* ------------------------------------------------------------------------
 * Edge-list preparation -- graph/static/static_graph.py:65-78 (and
 * naive_graph.py:76-94): forward order = stable sort by (dst, src) with
 * eid = rank; backward order = lexicographic sort of (src, dst, eid).
 * perm_fwd[j] = index in the caller's list of the edge that gets eid j.
 * ---------------------------------------------------------------------- */
typedef struct { int k0, k1, k2; int64_t pos; } orc_key3;

static int orc_cmp_key3(const void *pa, const void *pb)
{
    const orc_key3 *a = (const orc_key3 *)pa, *b = (const orc_key3 *)pb;
    if (a->k0 != b->k0) return a->k0 < b->k0 ? -1 : 1;
    if (a->k1 != b->k1) return a->k1 < b->k1 ? -1 : 1;
    if (a->k2 != b->k2) return a->k2 < b->k2 ? -1 : 1;
    return a->pos < b->pos ? -1 : (a->pos > b->pos);   /* stability (python list.sort is stable) */
}

int orc_prepare_edge_lists(const int *src, const int *dst, int64_t E,
                           int64_t *perm_fwd,
                           int *fwd_src, int *fwd_dst, int *fwd_eid,
                           int *bwd_src, int *bwd_dst, int *bwd_eid)
{
    orc_key3 *k = (orc_key3 *)malloc(sizeof(orc_key3) * (size_t)(E > 0 ? E : 1));
    if (!k) return -2;
    for (int64_t i = 0; i < E; ++i) { k[i].k0 = dst[i]; k[i].k1 = src[i]; k[i].k2 = 0; k[i].pos = i; }
    qsort(k, (size_t)E, sizeof(orc_key3), orc_cmp_key3);
    for (int64_t j = 0; j < E; ++j) {
        perm_fwd[j] = k[j].pos;
        fwd_src[j] = k[j].k1; fwd_dst[j] = k[j].k0; fwd_eid[j] = (int)j;
    }
    for (int64_t j = 0; j < E; ++j) { k[j].k0 = fwd_src[j]; k[j].k1 = fwd_dst[j]; k[j].k2 = (int)j; k[j].pos = j; }
    qsort(k, (size_t)E, sizeof(orc_key3), orc_cmp_key3);
    for (int64_t j = 0; j < E; ++j) { bwd_src[j] = k[j].k0; bwd_dst[j] = k[j].k1; bwd_eid[j] = k[j].k2; }
    free(k);
    return 0;
}
