/*
 * stg_gpma_oracle.c -- CPU restatement of what the reference's GPMA graph hands to its kernels.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as stg_oracle.c: tests/, smoke() and bench.py's cpu_baseline
 * leg may use it as the checker; nothing under stgraph_amd/ may).
 *
 * PARITY UNPINNED.  The reference ships no build of this component (graph/dynamic/gpma/gpma.so is
 * listed in .MISSING_LARGE_BLOBS) and gpma.cu needs nvcc, thrust, cub and device-side launches
 * (-rdc=true), so it is unbuildable in this image; the reference has no test or fixture for it either.
 * What follows restates, loop for loop, the functions that define the array contract between the GPMA
 * store and the emitted kernels (paths relative to /root/reference/stgraph):
 *
 *   orc_gpma_label_edges         graph/dynamic/gpma/gpma.cu:1121-1163  (label_edges_kernel + scan)
 *   orc_gpma_degrees             gpma.cu:1034-1062                      (update_node_degrees)
 *   orc_gpma_build_backward_csr  gpma.cu:1165-1231                      (count_sort_kernel, run serially)
 *   orc_gpma_node_ids            gpma.cu:1239-1270                      (get_csr_ptrs' degree sort)
 *   orc_gpma_gcn_agg             compiler/code_gen/templates/fa/tpl_fa_gpma.jinja:1-68 with the GCN
 *                                statements of SURVEY.md Appendix B.1/B.2
 *
 * and one generator that is NOT a restatement: orc_gpma_image lays a given live edge set out as a
 * gapped array obeying the structural rules of the store (row walls (r<<32)|0xFFFFFFFF with value 1,
 * gpma.cu:936-978; row_offset[r+1] = slot of wall r, :385-390; two KEY_MAX guards at the end, :925-926;
 * empty slots KEY_NONE, lazily deleted entries keep their key with value 0, :197-212), with holes and
 * tombstones placed by a seeded LCG instead of the PMA's density-driven rebalancing.  The position of
 * the gaps is the one thing the kernels' results do not depend on.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KEY_NONE 0xFFFFFFFFFFFFFFFFull
#define ORC_KEY_MAX 0xFFFFFFFFFFFFFFFEull
#define ORC_COL_IDX_NONE 0xFFFFFFFFu
#define ORC_VALUE_NONE 0u

static uint32_t orc_lcg(uint64_t *s)
{
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(*s >> 33);
}

/* Gapped image of the sorted live keys `live[E]` ((row << 32) | col, strictly ascending, row < N).
 * `dead[D]`: keys (ascending, disjoint from live) that stay in the array as lazily deleted entries.
 * hole_pct: chance (0..100) of an empty slot before each entry.  Returns the number of slots written
 * (<= capacity) or -1 if capacity is too small.  Values of live entries are 1 (what an insert writes,
 * gpma.cu:1087) -- label_edges assigns the real labels. */
int64_t orc_gpma_image(const uint64_t *live, int64_t E, const uint64_t *dead, int64_t D, int N, int hole_pct,
                       uint64_t seed, uint64_t *keys, uint32_t *values, uint32_t *row_offset, int64_t capacity)
{
    int64_t p = 0, il = 0, id = 0;
    uint64_t s = seed;
    row_offset[0] = 0;
    for (int r = 0; r <= N; ++r) {
        const uint64_t wall = r < N ? (((uint64_t)r << 32) | ORC_COL_IDX_NONE) : ORC_KEY_MAX;
        for (;;) {
            const int hl = il < E && live[il] < wall, hd = id < D && dead[id] < wall;
            if (!hl && !hd) break;
            const int take_live = hl && (!hd || live[il] < dead[id]);
            while ((int)(orc_lcg(&s) % 100) < hole_pct) {
                if (p >= capacity) return -1;
                keys[p] = ORC_KEY_NONE, values[p++] = 0;
            }
            if (p >= capacity) return -1;
            keys[p] = take_live ? live[il++] : dead[id++];
            values[p++] = take_live ? 1u : ORC_VALUE_NONE;
        }
        if (p + 2 > capacity) return -1;
        if (r < N) {
            keys[p] = wall, values[p] = 1;
            row_offset[r + 1] = (uint32_t)p++;
        } else {
            keys[p] = ORC_KEY_MAX, values[p++] = 1;
            keys[p] = ORC_KEY_MAX, values[p++] = 1;
        }
    }
    return p;
}

/* update_node_degrees (gpma.cu:1034-1062) applied to the whole live set: in_degree[(uint32)key]++,
 * out_degree[key >> 32]++. */
void orc_gpma_degrees(const uint64_t *live, int64_t E, int N, uint32_t *in_degree, uint32_t *out_degree)
{
    memset(in_degree, 0, sizeof(uint32_t) * (size_t)N);
    memset(out_degree, 0, sizeof(uint32_t) * (size_t)N);
    for (int64_t i = 0; i < E; ++i) {
        in_degree[(uint32_t)live[i]]++;
        out_degree[(uint32_t)(live[i] >> 32)]++;
    }
}

/* label_edges (gpma.cu:1148-1163): cum_out_degree = inclusive scan of out_degree; label_edges_kernel
 * (:1121-1146): row `index` numbers its live entries from cum_out_degree[index-1] + 1. */
void orc_gpma_label_edges(const uint32_t *row_offset, const uint64_t *keys, uint32_t *values,
                          const uint32_t *out_degree, int N)
{
    uint32_t *cum = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N > 0 ? N : 1));
    uint32_t run = 0;
    for (int i = 0; i < N; ++i) cum[i] = (run += out_degree[i]);
    for (int index = 0; index < N; ++index) {
        int edge_count = 1;
        const int beg = (int)row_offset[index], end = (int)row_offset[index + 1];
        if (index > 0) edge_count = (int)cum[index - 1] + 1;
        for (int i = beg; i < end; ++i) {
            const uint64_t key = keys[i];
            const uint32_t value = values[i];
            if (key != ORC_KEY_MAX && (key & 0xffffffffull) != ORC_COL_IDX_NONE && value != ORC_VALUE_NONE) {
                values[i] = (uint32_t)edge_count;
                ++edge_count;
            }
        }
    }
    free(cum);
}

/* build_backward_csr (gpma.cu:1190-1231): bwd_row_offset = inclusive scan of in_degree, [N] = edge_count;
 * count_sort_kernel (:1165-1188) with its threads run one after another in index order, so a row is
 * filled from its END in forward-array order.  (On the GPU the in-row order is whatever atomicSub hands
 * out -- "THIS IS NO LONGER A STABLE SORT", :1168 -- so only the row CONTENT is defined.) */
void orc_gpma_build_backward_csr(const uint32_t *row_offset, const uint64_t *keys, const uint32_t *values,
                                 const uint32_t *in_degree, int N, uint32_t edge_count,
                                 uint32_t *bwd_row_offset, uint64_t *bwd_keys, uint32_t *bwd_values)
{
    uint32_t run = 0;
    for (int i = 0; i < N; ++i) bwd_row_offset[i] = (run += in_degree[i]);
    bwd_row_offset[N] = edge_count;
    for (int index = 0; index < N; ++index) {                      /* row_offset_size - 1 == N */
        const int beg = (int)row_offset[index], end = (int)row_offset[index + 1];
        for (int i = beg; i < end; ++i) {
            const uint64_t key = keys[i];
            const uint32_t value = values[i];
            const uint32_t src = (uint32_t)key;
            if (key != ORC_KEY_MAX && src != ORC_COL_IDX_NONE && value != ORC_VALUE_NONE) {
                const uint32_t pos = (bwd_row_offset[src]--) - 1;  /* atomicSub returns the old value */
                bwd_keys[pos] = ((uint64_t)src << 32) + (key >> 32);
                bwd_values[pos] = value;
            }
        }
    }
}

/* get_csr_ptrs (gpma.cu:1239-1270): node_ids = sequence sorted by degree, greater<int>.  thrust leaves
 * the order of equal degrees open; ascending id here. */
typedef struct { uint32_t deg; uint32_t id; } orc_gpma_di;
static int orc_gpma_cmp(const void *a, const void *b)
{
    const orc_gpma_di *x = (const orc_gpma_di *)a, *y = (const orc_gpma_di *)b;
    if (x->deg != y->deg) return x->deg > y->deg ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}
void orc_gpma_node_ids(const uint32_t *degree, int N, uint32_t *node_ids)
{
    orc_gpma_di *t = (orc_gpma_di *)malloc(sizeof(orc_gpma_di) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; ++i) t[i].deg = degree[i], t[i].id = (uint32_t)i;
    qsort(t, (size_t)N, sizeof(orc_gpma_di), orc_gpma_cmp);
    for (int i = 0; i < N; ++i) node_ids[i] = t[i].id;
    free(t);
}

/* The emitted GCN unit on a 'gpma' graph: tpl_fa_gpma.jinja:13-68 around the statements of Appendix
 * B.1 (ew == NULL) / B.2.  `literal` != 0 keeps the template's predicate exactly as written --
 * `eid = label - 1` FIRST (:34), then `... && eid != 0` (:43) -- which skips the edge labelled 1 and
 * admits tombstones (label 0 -> eid 0xFFFFFFFF, an out-of-range edge-tensor index: they are only
 * admitted here when ew == NULL).  `literal` == 0 applies what gpma.cu itself tests everywhere
 * (label != VALUE_NONE, e.g. :1138, :1181): the live edges. */
void orc_gpma_gcn_agg(const float *x, const float *norm_row, const float *norm_col, const float *ew, float *out,
                      const uint32_t *row_offsets, const uint32_t *eids, const uint64_t *column_indices,
                      const uint32_t *node_ids, int N, int F, int F_active, int literal)
{
    for (int idx = 0; idx < N; ++idx) {
        const uint32_t r = node_ids ? node_ids[idx] : (uint32_t)idx;
        const uint32_t beg = row_offsets[r], end = row_offsets[r + 1];
        for (int tx = 0; tx < F_active; ++tx) {
            float acc = 0.f;
            for (uint32_t e = beg; e < end; ++e) {
                const uint64_t key = column_indices[e];
                const uint32_t eid = eids[e] - 1u;
                const uint32_t c = (uint32_t)(key & 0xffffffffull);
                const int pass = literal ? (eid != 0u) : (eids[e] != ORC_VALUE_NONE);
                if (key != ORC_KEY_MAX && c != ORC_COL_IDX_NONE && pass) {
                    float t = norm_col[c] * x[(int64_t)c * F + tx];
                    if (ew) t = t * ew[eid];
                    acc += t;
                }
            }
            out[(int64_t)r * F + tx] = acc * norm_row[r];
        }
    }
}
