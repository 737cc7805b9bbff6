/*
 * stgraph_hip.h -- C ABI of libstgraph_hip.so, the MI355X (gfx950) replacement
 * for the native side of STGraph's Seastar hot path.
 *
 * Every entry point is extern "C", takes plain pointers and sizes, returns
 * 0 on success or a non-zero code (a hipError_t value, or STG_ERR_* below), and
 * leaves a human readable message in stg_last_error_string() (thread local).
 * Device pointers are marked [dev], host pointers [host].  `stream` is a
 * hipStream_t passed as void* (NULL = the null stream, which is what the
 * reference launches on: compiler/execution_unit.py:363-370).
 * No entry point allocates or synchronises; scratch memory is passed in.
 *
 * "Replaces" cites the reference interface (paths relative to
 * /root/reference/stgraph) that a maintainer would rebind to this symbol; the
 * binding stubs are shown in INTEGRATION.md.
 */
#ifndef STGRAPH_HIP_H
#define STGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature or contract changes (stgraph_amd/_C.py checks it at load):
 *   3: stg_xent_fwd / stg_xent_bwd count the rows (ignore_index = -100, n_counted); round-1 changes to
 *      stg_link_head_fwd (loss_in), stg_tgcn_head_fwd_acc and the xent status contract folded in.
 *   4: stg_tgcn_step_fwd / _bwd, stg_tgcn_window_loss, stg_gemm_tn_form_f32 added.
 *  22: stg_tgcn_step_*_args gain `w_image` (last field); stg_tgcn_pack_weights_x3, stg_tgcn_step_image_bytes; knob "step_impl".
 *  23: stg_gat_fwd_k1_uniform, stg_gat_fc_out, stg_gat_fwd_k1_scored, stg_gat_bwd_factored_elu; knob "rowgemm_x3".
 *  24: stg_tgcn_step_fwd_args gains w_fold, b_fold, fold_status (last fields): the folded form of the forward step launch; x3 / da3 of
 *      the step launches optional; stg_tgcn_unfold_gate_grads.
 *  25: stg_rowgemm_act_bits_f32, stg_rowgemm_bits_words, stg_rowgemm_bits_supported: a ReLU layer's sign pattern as bits;
 *      stg_xent_fwd_grad, stg_xent_scale_grad: cross-entropy forward and gradient in one pass; stg_gat_bwd_uniform_*,
 *      stg_gat_bwd_prepass, stg_gemm_tn_gated_f32, stg_rowgemm_heads_f32: the GAT backward unit in the uniform-attention form;
 *      stg_gat_fc_fwd accepts feat == NULL, stg_gat_fc_feat_if; stg_tgcn_step_bwd_args gains ld_d (last field).
 *  26: the bf16-split forms of the step launches are RETIRED (measured slower than the fp32 / folded fp32 forms: profiles/r04_stepx_*,
 *      r04_stepf_*; sources in git history before round 5): stg_tgcn_pack_weights_x3, stg_tgcn_step_image_bytes and the knobs
 *      "step_impl" / "step_fold" are gone, `w_image` of both argument blocks is reserved and must be NULL, w_fold now REQUIRES
 *      fold_bound (the folded launch runs on the fp32 matrix instruction only).
 *  27: stg_xent_small_supported / _fwd / _bwd: softmax cross-entropy of a SMALL logits matrix as one launch each way (the graph of
 *      a captured Cora epoch is launch-count bound); stg_bias_act_bwd finishes the column sums in its own launch when one
 *      workgroup covers the matrix (no signature change); stg_gat_bwd_prepass_heads(_supported); stg_gat_fc_fwd / stg_gat_fc_out take the
 *      3-term bf16 split form at H % 4 == 0 heads of 64 over 64 inputs (results equal to fp32 rounding); stg_build_job gains `id`
 *      (last field); stg_mm_bwd_small(_supported), stg_gemm_tn_small_f32(_supported), stg_gat_attn_fold. */
#define STG_ABI_VERSION 27

#define STG_ERR_INVALID_ARGUMENT 10001   /* NULL pointer, negative size, bad shape  */
#define STG_ERR_UNSUPPORTED      10002   /* shape outside what the kernels cover     */
#define STG_ERR_VERTEX_RANGE     10003   /* an edge endpoint is outside [0, N)       */
#define STG_ERR_WORKSPACE        10004   /* workspace too small                       */
#define STG_ERR_JIT              10005   /* run-time compilation of a generated kernel failed */

int         stg_abi_version(void);
const char *stg_last_error_string(void);

/* Launch-time knobs (performance only, never results).  Unknown keys return
 * STG_ERR_INVALID_ARGUMENT.  Keys: "gcn_lanes_per_row" (0 = auto, else a power
 * of two <= 64), "gcn_unroll" (0 = auto, 2/4/8), "gcn_long_threshold" (0 = auto; rows with more
 * edges take the wave-per-row path of stg_gcn_agg_edge), "gcn_xcd_tile" (0 = auto; T >= 1: each XCD takes runs
 * of T consecutive workgroups' rows, 1 = plain round robin), "gcn_addr32" (0 = auto, 1 = never use 32-bit gather
 * offsets), "gcn_block" (0 = auto; 64 / 128 / 256 threads per workgroup of the plain stg_gcn_agg* launch),
 * "gcn_tile" (edge-dealt kernel for rows of <= 32 floats: 0 = auto, 1 = never, 2 = whenever legal),
 * "gcn_tile_rows" (its rows per workgroup: 0 = one per lane group, else 8 .. 256), "gcn_tile_pipe" (its
 * persistent, software-pipelined form: 0 / 1 = off, 2 = whenever the tile kernel runs; measured equal),
 * "xw_waves" (0 = auto; 4 / 8 waves per workgroup of stg_gcn_agg_transform), "cell_rows" (0 = auto; 16 / 32 rows
 * per tile of stg_tgcn_cell_fused_fwd), "step_waves" (stg_tgcn_step_*: 0 = auto, 12 / 16 waves per workgroup),
 * "rowgemm_x3" (stg_rowgemm_f32 / _strided_f32 / _act_f32 at K, M in {64, 128} and N K < 2^30: 0 = from 64 K rows
 * every product as a 3-term bf16 split on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, 1 = always v_mfma_f32_16x16x4_f32, 2 = the
 * split form at every N, 3 = its lane-owns-row-pieces load / store variant (diagnosis); the one knob that selects between two
 * ARITHMETICS: both forms inside the fp32 kernel's error bound against fp64, integer data exact in both), "step_spread"
 * (0 = one workgroup per CU when there are fewer tiles than wave slots, 1 = packed grid), "step_coop" (stg_tgcn_step_*, tiles of the
 * last partial round: 0 = each shared by four waves of its workgroup -- bit-identical results --, 1 = one wave each), "gemm_wide" (tall-skinny weight
 * gradients: 0 = the 16-byte-per-lane form where the widths allow, 1 = never), "gemm_x3" (the same contractions at M in {32, 64, 128},
 * N in {64, 96, 128}: 0 = every product as a 3-term bf16 split on v_mfma_f32_32x32x16_bf16 from 64 K rows in all, 1 = never, 2 = always;
 * like "rowgemm_x3" a choice between two arithmetics inside the fp32 form's error bound against fp64), "gemm_cyclic" (its row-group hand-out: 0 .. 2), "gemm_xcd_pair" (its
 * workgroup order when M x N takes several workgroups per K slice: 0 = those of a slice on one XCD, 1 = dealt in turn),
 * "gcn_wide_long" (rows of >= 1024 edges at F >= 128: 0 = feature-sliced workgroups beside the main launch, 1 = never, 2 = behind
 * it on the same stream), "build_lds_count" (stg_graph_build_direct2_device: 0 = histograms in LDS when |V| <= 40 K and the graph
 * is dense enough, 1 = whenever |V| fits, 2 = never), "store_rows" (stg_edgeset_step_device given the old set's row offsets: 0 = the
 * new row offsets are derived from them inside the merge launch -- two launches per step --, 1 = searched in the merged keys by a
 * launch of their own, as without the offsets). */
int stg_set_tuning(const char *key, int value);

/* ---------------------------------------------------------------- CSR, host
 * Replaces the pybind class `CSR(edge_list, edge_weight, num_nodes,
 * is_edge_reverse)` -- graph/static/csr.cu:68-157, bound at csr.cu:181-201.
 * (a[i], b[i], eid[i]) are the triples in the order the Python caller prepared
 * them; edge_weight is indexed by eid and may be NULL (all ones).  All arrays
 * [host]; outputs sized N+1, E, E, N, N, N, N.
 */
int stg_csr_ctor_host(const int32_t *a, const int32_t *b, const int32_t *eid,
                      const float *edge_weight, int64_t E, int32_t N, int is_edge_reverse,
                      int32_t *row_offset, int32_t *column_indices, int32_t *eids,
                      int32_t *node_ids, int32_t *in_degrees, int32_t *out_degrees,
                      float *weighted_out_degrees);

/* Replaces StaticGraph._prepare_edge_lst_fwd/_bwd + the two CSR constructions
 * of StaticGraph.__init__ -- graph/static/static_graph.py:40-78 (and, per
 * snapshot, NaiveGraph.__init__ -- graph/dynamic/naive/naive_graph.py:45-94).
 * Input: E (src,dst) pairs in caller order [host].
 * Output [host]: perm_fwd[E] (caller position of the edge that received eid j,
 * i.e. the (dst,src)-stable-sorted order the reference sorts the caller's list
 * into), forward CSR (rows = dst, cols = src, eids = 0..E-1), backward CSR
 * (rows = src, cols = dst, eids = forward positions), node_ids of both
 * (rows by non-increasing degree, ties by ascending id), and the graph's
 * in/out degrees.
 */
int stg_graph_build_host(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                         int64_t *perm_fwd,
                         int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                         int32_t *fwd_eids, int32_t *fwd_node_ids,
                         int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                         int32_t *bwd_eids, int32_t *bwd_node_ids,
                         int32_t *in_degrees, int32_t *out_degrees);

/* -------------------------------------------------------------- CSR, device
 * Same contract as stg_graph_build_host with every array [dev]; runs entirely
 * on `stream` (radix sort + binary-search row offsets), no host round trip.
 * This is the per-snapshot rebuild of the dynamic-temporal configuration.
 * Additionally validates endpoints: *status [dev, int32] is set to 0, or to
 * STG_ERR_VERTEX_RANGE if any endpoint is outside [0, N).
 * workspace [dev] must hold stg_graph_build_device_workspace_bytes(E, N) bytes.
 */
size_t stg_graph_build_device_workspace_bytes(int64_t E, int32_t N);
int stg_graph_build_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                           int64_t *perm_fwd,
                           int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                           int32_t *fwd_eids, int32_t *fwd_node_ids,
                           int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                           int32_t *bwd_eids, int32_t *bwd_node_ids,
                           int32_t *in_degrees, int32_t *out_degrees,
                           int32_t *status, void *workspace, size_t workspace_bytes,
                           void *stream);

/* The same build without a global sort, for graphs whose rows are short (the per-snapshot rebuild of dynamic graphs:
 * BASELINE configs[4] runs it once per timestep): degrees by atomic histogram, row offsets by a one-workgroup scan,
 * entries scattered into their rows and ranked inside the row by counting smaller keys (ties broken by the caller
 * position, i.e. the stable (dst, src) order of static_graph.py:65-72) -- 6 launches instead of the 59 three rocPRIM
 * sorts take at |E| = 250K.  Outputs are bit-identical to stg_graph_build_device.  If a row is longer than 2048
 * entries nothing but the degrees and row offsets is produced and *status gets STG_BUILD_NEEDS_SORT: call
 * stg_graph_build_device instead.  fwd_node_ids / bwd_node_ids may both be NULL (they only fix a processing
 * order, and sorting |V| degrees costs more launches than everything else here). */
/* node_ids of a CSR after the fact: rows by non-increasing degree, ties by ascending id (csr.cu:142-154 leaves
 * tie order open) -- for builds that skipped them. */
size_t stg_rows_by_degree_workspace_bytes(int32_t N);
int stg_rows_by_degree_device(const int32_t *degrees, int32_t N, int32_t *node_ids, void *workspace,
                              size_t workspace_bytes, void *stream);
#define STG_BUILD_NEEDS_SORT 32
size_t stg_graph_build_direct_workspace_bytes(int64_t E, int32_t N);
int stg_graph_build_direct_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                  int64_t *perm_fwd,
                                  int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                                  int32_t *fwd_eids, int32_t *fwd_node_ids,
                                  int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                                  int32_t *bwd_eids, int32_t *bwd_node_ids,
                                  int32_t *in_degrees, int32_t *out_degrees,
                                  int32_t *status, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* The same build (same arrays, same bits) for a snapshot that is REBUILT -- NaiveGraph(resident=False), reference
 * graph/dynamic/naive/naive_graph.py:45-74 built once per snapshot; here once per snapshot and epoch -- in five launches
 * with one atomic pass, together with what the training loop derives from every new CSR
 * (dynamic-temporal-tgcn/seastar/train.py:213-218): norm [N] = in_deg^-1/2 (nullable) and norm gathered through the
 * forward / backward columns [E] (nullable; need norm).  zero_counters: 2 * ((N + 3) & ~3) ints owned by the caller, all
 * zero on entry and all zero again on exit; the per-vertex arrays 16-byte aligned.  sticky_status is OR-ed into, never cleared (codes as above): for edge lists an
 * earlier stg_graph_build_direct_device call has validated. */
int stg_graph_build_direct2_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N, int64_t *perm_fwd,
                                   int32_t *fwd_row_offset, int32_t *fwd_column_indices, int32_t *fwd_eids,
                                   int32_t *fwd_node_ids, int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                                   int32_t *bwd_eids, int32_t *bwd_node_ids, int32_t *in_degrees, int32_t *out_degrees,
                                   float *norm, float *norm_col_fwd, float *norm_col_bwd, int32_t *zero_counters,
                                   int32_t *sticky_status, void *workspace, size_t workspace_bytes, void *stream);

/* The snapshots of one BPTT window (dynamic-temporal-tgcn/seastar/train.py:205-232 walks them one get_graph() at a time) are
 * independent builds over the same |V|: up to STG_BUILD_BATCH_MAX of them in the SAME six launches (blockIdx.z = snapshot),
 * every output bit-identical to n_jobs calls of stg_graph_build_direct2_device without node_ids.  Each job brings its own
 * zero_counters and its own workspace (>= stg_graph_build_direct_workspace_bytes(E, N)); N > 0 is common to the batch. */
#define STG_BUILD_BATCH_MAX 16
typedef struct stg_build_job {
    const int32_t *src, *dst;
    int64_t E;
    int64_t *perm_fwd;
    int32_t *fwd_row_offset, *fwd_column_indices, *fwd_eids;
    int32_t *bwd_row_offset, *bwd_column_indices, *bwd_eids;
    int32_t *in_degrees, *out_degrees;
    float *norm, *norm_col_fwd, *norm_col_bwd;       /* nullable, as above */
    int32_t *zero_counters;
    void *workspace;
    size_t workspace_bytes;
    /* The caller's name for this build, 1 .. 2^23 - 1 (0: none; ABI 27).  The sticky word is shared by every build of a device:
     * the FIRST build that fails leaves (id << 8) beside its code bits (bits 0 .. 7: 1 = endpoint out of range,
     * STG_BUILD_NEEDS_SORT), later failures only OR their code bits in -- whoever reads the word can name the edge list. */
    int32_t id;
} stg_build_job;
int stg_graph_build_direct2_batch_device(const stg_build_job *jobs, int32_t n_jobs, int32_t N, int32_t *sticky_status,
                                         void *stream);

/* ------------------------------------------------------- dynamic edge store (PCSR, GPMA)
 * Replaces the reference's PCSR class: graph/dynamic/pcsr/pcsr.cu:273-939
 * (PCSR::edge_update_list :759-779, label_edges :745-757, build_csr :829-879,
 * build_reverse_csr :781-827, move_pinned_to_gpu :881-886), as driven by
 * graph/dynamic/pcsr/pcsr_graph.py:46-166 -- and the reference's GPMA functions:
 * graph/dynamic/gpma/gpma.cu (update_gpma :838-911 as called by edge_update_t
 * :1064-1119, label_edges :1121-1163, count_sort_kernel + build_backward_csr
 * :1165-1231, get_csr_ptrs :1239-1270), as driven by gpma_graph.py:58-152.
 *
 * State = the current edge SET as two dense, ascending arrays of packed keys
 *   keys_fwd[i] = (uint64)dst << 32 | src        keys_bwd[i] = (uint64)src << 32 | dst
 * (the PMA's "source" is the graph's dst: every call site passes is_reverse_edge=True).
 *
 * update: keys_out = (keys_in \ del) U add for both orientations, out of place
 *   (E_out = E + n_add - n_del entries each).  Contract, as for the reference's
 *   own update streams (dynamic_graph.py:56-79): added edges are absent, deleted
 *   edges are present, ids < N, no duplicates inside a batch.  *status [dev for
 *   _device / host for _host] = 0, or a bit set: 1 vertex id out of range, 2 added
 *   edge already present, 4 deleted edge absent, 8 edge both added and deleted;
 *   the outputs are unspecified then.  Batches may arrive in any order.
 * emit: flags = STG_EMIT_REVERSE? | STG_EMIT_KEY_ORDER?.  Without KEY_ORDER, the
 *   CSR the reference's build_csr (rows = dst) or build_reverse_csr (REVERSE:
 *   rows = src) would have produced: columns of a
 *   row in DESCENDING order (the PMA row is emitted back to front), eids1 = the
 *   1-based labels of label_edges (1 + rank in (dst, src) order; what
 *   tpl_fa_pcsr.jinja:32-34 subtracts 1 from), eids0 = eids1 - 1 (what the
 *   stg_gcn_agg / stg_gat_* entry points take).  node_ids = rows by non-increasing
 *   length (ties: ascending id); degrees = row lengths.  row_offset is always
 *   written; column_indices, eids1, eids0 may each be NULL and node_ids + degrees
 *   may both be NULL -- those parts are skipped (the un-weighted GCN kernels never
 *   read eids, so a training step emits structure only and labels on demand).
 *   With STG_EMIT_KEY_ORDER the slot of an edge is its position in the key array:
 *   columns of a row ASCENDING.  That is the GPMA view: the forward array the
 *   reference's kernels walk (tpl_fa_gpma.jinja:28-43) with its holes squeezed out,
 *   labels = label_edges' running count in key order (gpma.cu:1121-1146); REVERSE =
 *   build_backward_csr's rows (gpma.cu:1165-1188), whose in-row order the reference
 *   leaves to atomicSub ("no longer a stable sort") and this build fixes ascending.
 * All arrays [dev] for *_device, [host] for *_host; no call synchronises.
 */
size_t stg_edgeset_update_workspace_bytes(int64_t n_add, int64_t n_del);
int stg_edgeset_update_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                              const int32_t *add_src, const int32_t *add_dst, int64_t n_add,
                              const int32_t *del_src, const int32_t *del_dst, int64_t n_del, int32_t N,
                              uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *status,
                              void *workspace, size_t workspace_bytes, void *stream);
int stg_edgeset_update_host(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                            const int32_t *add_src, const int32_t *add_dst, int64_t n_add,
                            const int32_t *del_src, const int32_t *del_dst, int64_t n_del, int32_t N,
                            uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *status);
/* One orientation, batches already packed and strictly ascending (a PCSRGraph packs and sorts its
 * per-timestamp add/delete lists once, at construction): keys_out = (keys_in \ del) U add in one
 * scatter pass, no workspace.  status as above, plus 16 = a batch was not ascending.
 * The caller zeroes *status (it accumulates over the calls of one update). */
int stg_edgeset_merge_device(const uint64_t *keys_in, int64_t E, const uint64_t *add_sorted, int64_t n_add,
                             const uint64_t *del_sorted, int64_t n_del, uint64_t *keys_out,
                             int32_t *status, void *stream);

/* One timestamp of a delta store (PCSRGraph / GPMAGraph._update_graph_forward / _backward: reference
 * graph/dynamic/pcsr/pcsr_graph.py:121-166 = edge_update_list x 2 + label_edges + build_csr / build_reverse_csr, and the
 * norm the training loop derives from the new in-degrees, dynamic-temporal-tgcn/seastar/train.py:213-218) as THREE
 * launches: (old \ del) U add in both orientations from batches packed + sorted up front, both CSRs (row offsets +
 * columns: rows back to front, or ascending with STG_EMIT_KEY_ORDER), in_degrees [N] (nullable), norm [N] = in_deg^-1/2
 * (0 for isolated rows; nullable) and norm gathered through either CSR's columns [E_out] (nullable; need norm).
 * fwd_row_offset_in / bwd_row_offset_in (both or neither; nullable): EXACTLY the row offsets [N + 1] of keys_fwd_in / keys_bwd_in, as
 * a previous step emitted them.  With them a batch key is placed inside its own row of the old set, and the new row offsets, degrees
 * and norm are derived from them and the batches (old offset + additions below the row - deletions below it; exact for every step that
 * leaves status clean) by blocks of the merge launch: TWO launches per step.  Offsets that are not those of the input set give
 * undefined outputs.
 * status is OR-ed into, never cleared (codes as stg_edgeset_update_device): one word may serve a whole store. */
int stg_edgeset_step_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E, const uint64_t *add_fwd,
                            const uint64_t *add_bwd, int64_t n_add, const uint64_t *del_fwd, const uint64_t *del_bwd,
                            int64_t n_del, int32_t N, int flags, uint64_t *keys_fwd_out, uint64_t *keys_bwd_out,
                            int32_t *fwd_row_offset, int32_t *fwd_column_indices, int32_t *bwd_row_offset,
                            int32_t *bwd_column_indices, int32_t *in_degrees, float *norm, float *norm_col_fwd,
                            float *norm_col_bwd, const int32_t *fwd_row_offset_in, const int32_t *bwd_row_offset_in,
                            int32_t *status, void *stream);

/* Several steps issued back to back -- the graph updates of a BPTT window, which the training loop performs before the window's
 * first model step -- need not pay a launch for each emission: a step's emission (columns + per-edge norm of both CSRs) depends
 * only on ITS merge launch, and the next step's merge does not depend on it.  stg_edgeset_step_deferred_device is
 * stg_edgeset_step_device except that (i) it describes its own emission in *pending_out instead of launching it and (ii) it
 * carries `carry` (nullable: an earlier step's pending emission) as extra blocks of its merge launch: ONE launch per step.  The
 * last pending emission is issued by stg_edgeset_emit_pending_device.  Outputs bit-identical to the undeferred calls; the
 * column / norm_col arrays of a step hold nothing until its emission has run. */
typedef struct stg_store_emission {
    const uint64_t *keys_fwd, *keys_bwd;
    int64_t E;
    const int32_t *fwd_row_offset, *bwd_row_offset;
    int32_t *fwd_column_indices, *bwd_column_indices;
    const float *norm;
    float *norm_col_fwd, *norm_col_bwd;
    int flags;
} stg_store_emission;
int stg_edgeset_step_deferred_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E, const uint64_t *add_fwd,
                                     const uint64_t *add_bwd, int64_t n_add, const uint64_t *del_fwd, const uint64_t *del_bwd,
                                     int64_t n_del, int32_t N, int flags, uint64_t *keys_fwd_out, uint64_t *keys_bwd_out,
                                     int32_t *fwd_row_offset, int32_t *fwd_column_indices, int32_t *bwd_row_offset,
                                     int32_t *bwd_column_indices, int32_t *in_degrees, float *norm, float *norm_col_fwd,
                                     float *norm_col_bwd, const int32_t *fwd_row_offset_in, const int32_t *bwd_row_offset_in,
                                     const stg_store_emission *carry, stg_store_emission *pending_out, int32_t *status, void *stream);
int stg_edgeset_emit_pending_device(const stg_store_emission *pending, void *stream);
#define STG_EMIT_REVERSE   1   /* rows = src (build_reverse_csr / build_backward_csr) */
#define STG_EMIT_KEY_ORDER 2   /* GPMA view: rows and columns in key order; default = PCSR's back-to-front rows */
size_t stg_edgeset_emit_csr_workspace_bytes(int32_t N);
int stg_edgeset_emit_csr_device(const uint64_t *keys_fwd, const uint64_t *keys_bwd, int64_t E, int32_t N,
                                int flags, int32_t *row_offset, int32_t *column_indices,
                                int32_t *eids1, int32_t *eids0, int32_t *node_ids, int32_t *degrees,
                                void *workspace, size_t workspace_bytes, void *stream);
int stg_edgeset_emit_csr_host(const uint64_t *keys_fwd, const uint64_t *keys_bwd, int64_t E, int32_t N,
                              int flags, int32_t *row_offset, int32_t *column_indices,
                              int32_t *eids1, int32_t *eids0, int32_t *node_ids, int32_t *degrees);

/* ------------------------------------------------------- JIT for generated kernels
 * Replaces the reference's run-time kernel build and launch for vertex functions
 * that are not one of the hand-written units: nvcc -> PTX (compiler/code_gen/
 * compiler.py:14-44), cuModuleLoadData / cuModuleGetFunction (compiler/
 * execution_unit.py:241-269) and cuLaunchKernel (execution_unit.py:359-372).
 * The HIP source is generated by stgraph_amd/compiler/codegen.py from the traced GIR.
 *
 * stg_jit_compile: hiprtc, --offload-arch=gfx950 -O3 -ffp-contract=off; needs no GPU.
 *   *code_out (malloc'ed code object, release with stg_jit_free), optional *log_out.
 * stg_jit_load / stg_jit_get_function / stg_jit_unload: hipModule API on the
 *   current device.  stg_jit_launch: 1-D launch; the generated kernels take
 *   n_ptr pointer arguments followed by n_int int32 arguments.
 */
int  stg_jit_compile(const char *source, const char *name, char **code_out, size_t *code_size_out, char **log_out);
void stg_jit_free(void *p);
int  stg_jit_load(const void *code, void **module_out);
int  stg_jit_get_function(void *module, const char *name, void **function_out);
int  stg_jit_unload(void *module);
int  stg_jit_launch(void *function, uint32_t grid, uint32_t block, const void *const *ptr_args, int32_t n_ptr,
                    const int32_t *int_args, int32_t n_int, void *stream);

/* ------------------------------------------------------- fused GCN aggregation
 * Replaces the compiler-emitted FA kernels K0/K1 of GCNConv (tracer
 * nn/pytorch/static/gcn_conv.py:162-182; template
 * compiler/code_gen/templates/fa/tpl_fa_csr{,_unsorted}.jinja; launch
 * compiler/execution_unit.py:359-372,407-415).  All pointers [dev], fp32 / int32.
 *
 *   out[r,f] = norm_row[r] * sum_{e in row r, c = column_indices[e]}
 *                              ((norm_col[c] * x[c,f]) * (ew ? ew[eids[e]] : 1))
 *   for f in [0, F_active); columns [F_active, F) of `out` are not written.
 *
 * One fp32 accumulator per (r,f), edges in CSR order, no FMA contraction: the
 * result is bit-identical to the reference's sequential loop.
 * forward : dst-major CSR, x = h,        norm_row = norm_col = norm
 * backward: src-major CSR, x = grad_out, norm_row = norm_col = norm
 * ew may be NULL; node_ids may be NULL ('csr_unsorted' graphs) or the row
 * processing order of 'csr' graphs (tpl_fa_csr.jinja:13-18).
 * x and out rows have stride F floats.  Edge arrays (column_indices, eids, ew) may be
 * NULL only for a graph without edges.
 */
int stg_gcn_agg(const float *x, const float *norm_row, const float *norm_col, const float *ew,
                float *out,
                const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                const int32_t *node_ids, int32_t N, int32_t F, int32_t F_active, void *stream);

/* Same computation with the two per-edge scalars already gathered into CSR order:
 *   norm_col_edge[e] = norm_col[column_indices[e]],  ew_edge[e] = ew[eids[e]] (or NULL)
 * (build them once per graph with stg_edge_gather_f32).  The scattered 4-byte gathers of
 * stg_gcn_agg -- a 64-B sector and a dependent round trip each -- become coalesced streams;
 * values and results are bit-identical.  This is what the Python executor launches.
 * rows_by_degree (nullable): the rows in non-increasing degree order -- the `node_ids` array every CSR
 * builder emits (csr.cu:142-154).  When given and a row is narrower than a wave (F_active / vector width
 * < 64 lanes), long rows are taken out of the row-group mapping and summed by the launch's FIRST workgroups,
 * one wave per row: the wave gathers 16-128 of the row's edges at a time into an LDS tile [edge][feature]
 * and adds them in CSR order, one feature per lane (same additions, same bits).  "Long" = more than 16
 * edges while the whole grid is resident at once (the launch then lasts as long as its longest row:
 * 19 -> 6 us on a Cora-shaped graph with a degree-168 vertex), more than 1024 edges on larger graphs.
 */
int stg_gcn_agg_edge(const float *x, const float *norm_row, const float *norm_col_edge,
                     const float *ew_edge, float *out,
                     const int32_t *row_offsets, const int32_t *column_indices,
                     const int32_t *node_ids, const int32_t *rows_by_degree,
                     int32_t N, int64_t E, int32_t F, int32_t F_active,
                     void *stream);      /* E = number of edges (lane-mapping heuristic only; 0 = unknown) */

/* GCNConv's tail fused into the aggregation's store (nn/pytorch/static/gcn_conv.py:185-188:
 * `h = h + self.bias`, `h = self.activation(h)`):
 *   out[r,f] = act( stg_gcn_agg_edge(...)[r,f] + bias[f] )      bias [F] or NULL, act = STG_ACT_*
 * The sum is formed exactly as above and the two extra roundings happen in the order torch applies
 * them, so `out` is bit-identical to aggregation -> torch add -> torch relu; two elementwise passes
 * over [N,F] (4 x 4NF bytes of traffic) disappear.  Every column is active (no F_active).
 */
#define STG_ACT_NONE 0
#define STG_ACT_RELU 1
int stg_gcn_layer_fwd(const float *x, const float *norm_row, const float *norm_col_edge,
                      const float *ew_edge, const float *bias, int32_t act, float *out,
                      const int32_t *row_offsets, const int32_t *column_indices,
                      const int32_t *node_ids, const int32_t *rows_by_degree,
                      int32_t N, int64_t E, int32_t F, void *stream);

/* stg_gcn_agg_edge and stg_gcn_layer_fwd in one signature (bias NULL + STG_ACT_NONE = the bare aggregation; with an epilogue
 * F_active must equal F), plus the caller's HUB PLAN for rows of a whole wave and wider (F >= 128, F % 4 == 0, F <= 256): of
 * the first rows of rows_by_degree, hub_rows_16 have >= 8192 edges, the next hub_rows_4 have 2048 .. 8191 and the next
 * hub_rows_1 have hub_threshold + 1 .. 2047 (counted once per graph on the host).  Those rows are skipped by the main launch
 * (every row above hub_threshold is) and summed by feature-sliced workgroups in CSR order -- bit-identical to the main path --
 * in a second launch of exactly 16 hub_rows_16 + 4 hub_rows_4 + hub_rows_1 workgroups.  All three 0: one launch, no skip.
 * The reference treats every row alike (compiler/execution_unit.py:92-100: one block per vertex). */
int stg_gcn_agg_edge2(const float *x, const float *norm_row, const float *norm_col_edge, const float *ew_edge,
                      const float *bias, int32_t act, float *out, const int32_t *row_offsets,
                      const int32_t *column_indices, const int32_t *node_ids, const int32_t *rows_by_degree, int32_t N,
                      int64_t E, int32_t F, int32_t F_active, int32_t hub_threshold, int32_t hub_rows_16, int32_t hub_rows_4,
                      int32_t hub_rows_1, void *stream);

/* y[n,f] = act(y[n,f] + bias[f]) in place (bias may be NULL; act = STG_ACT_NONE / STG_ACT_RELU): the `h + self.bias`,
 * `self.activation(h)` tail of GCNConv (reference nn/pytorch/static/gcn_conv.py:184-188) as one pass, for a layer
 * whose aggregation ran before its weight product. */
int stg_bias_act_fwd(float *y, const float *bias, int32_t act, int32_t N, int32_t F, void *stream);

/* Backward of that tail in one pass over [N,F] (torch: threshold_backward + sum(0), two passes + a
 * single-block-per-column reduction):
 *   g_act[r,f] = out ? (out[r,f] > 0 ? g[r,f] : 0) : (not written)      -- ReLU mask, `out` = the layer output
 *   colsum[f]  = sum_r (out ? g_act[r,f] : g[r,f])                       -- the bias gradient
 * out == NULL: no activation (g_act must be NULL too: g itself is what flows on).  colsum may be NULL.
 * Deterministic: fixed grid, per-workgroup partial sums in `workspace`, one final reduction per column.
 */
size_t stg_bias_act_bwd_workspace_bytes(int32_t N, int32_t F);
int stg_bias_act_bwd(const float *g, const float *out, float *g_act, float *colsum,
                     int32_t N, int32_t F, void *workspace, size_t workspace_bytes, void *stream);

/* Fused aggregate-then-transform:  out[N,Fout] = (A_hat x) W  in one kernel:
 * the rows of A_hat x (aggregated exactly as stg_gcn_agg_edge does, width Fin) are staged in an LDS
 * tile and multiplied by W [Fin,Fout] on the fp32 matrix cores.  Equals the layer's A_hat (x W) up to
 * fp32 rounding (aggregation is linear); pays when Fin < Fout (TGCN gates: gather 32 instead of 192
 * floats per edge).  P_out (nullable) [N,Fin] receives A_hat x for the backward pass.
 * Needs Fin % 4 == 0, 16 <= Fin, Fout % 32 == 0, 4 (64 (Fin+1) + Fin Fout) <= 64 KiB, else
 * STG_ERR_UNSUPPORTED (callers then run the two-kernel form). */
int stg_gcn_agg_transform(const float *x, const float *norm_row, const float *norm_col_edge,
                          const float *ew_edge, const float *W, float *out, float *P_out,
                          const int32_t *row_offsets, const int32_t *column_indices,
                          const int32_t *node_ids, int32_t N, int32_t Fin, int32_t Fout, void *stream);

/* dst[i] = table[idx[i]], i < n.  All [dev]. */
int stg_edge_gather_f32(float *dst, const float *table, const int32_t *idx, int64_t n, void *stream);

/* ones_flag (optional device word, NULL = not known) of the GAT entry points below: when *ones_flag == 0 every
 * attention score el[u] + er[v] is finite, so the vertex function's `emb - max([emb])` (reference
 * nn/pytorch/static/gat_conv.py:50) is +0, A[e,h] = exp(leaky(0)) = 1.0f exactly and S[v,h] = min(in-degree, 2^24);
 * stg_gat_fwd_k0 then writes S without visiting an edge (A is left untouched) and the other units take A as 1.0f
 * instead of loading it -- bit-identical to the general path.  stg_gat_score_flag sets it: 1 iff some |el| or |er| is
 * not below 1e38 (inf and NaN included). */
int stg_gat_score_flag(const float *el, const float *er, int64_t n, int32_t *flag, void *stream);

/* ----------------------------------------------------------------- fused GAT
 * Replace the emitted units K0, K1 (forward) and K2 (backward) of GATConv
 * (tracer nn/pytorch/static/gat_conv.py:48-56; listing SURVEY.md Appendix B.3).
 * Layouts: el, er, S, grad_el, grad_er [N,H]; A [E,H]; feat, out, g, grad_feat
 * [N,H,D]; all [dev] fp32.
 *
 * k0 : s = el[u,h] + er[v,h]; z = s - s; a = exp(z > 0 ? z : slope*z);
 *      A[eid,h] = a; S[v,h] = sum_e a            (dst-major CSR, h < H_active)
 * k1 : out[v,h,d] = sum_e (A[eid,h] / S[v,h]) * feat[u,h,d]      (tx < HD_active)
 * bwd: src-major CSR.  grad_feat[u,h,d] = sum_e g[v,h,d] * (A/S);
 *      t = ((g*feat[u])*(1/S[v]) + (-1*((g/S[v])*out[v]))) * A * (z>0 ? 1 : slope)
 *      grad_el[u,h] = sum_e sum_d t   (in-wave reduction, no atomics)
 *      T[eid,h]     = sum_d t         (edge scratch, [E,H])
 * bwd_er: grad_er[v,h] = sum_{e in in(v)} T[eid,h]   (dst-major CSR; replaces the
 *      reference's atomicAdd inside the edge loop, deterministic)
 */
int stg_gat_fwd_k0(const float *el, const float *er, float *A, float *S,
                   const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                   const int32_t *node_ids, int32_t N, int32_t H, int32_t H_active, float slope,
                   const int32_t *ones_flag, void *stream);
int stg_gat_fwd_k1(const float *A, const float *S, const float *feat, float *out,
                   const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                   const int32_t *node_ids, int32_t N, int32_t H, int32_t D, int32_t HD_active,
                   const int32_t *ones_flag, void *stream);
int stg_gat_bwd(const float *A, const float *S, const float *out, const float *g,
                const float *el, const float *er, const float *feat,
                float *grad_feat, float *grad_el, float *T,
                const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                const int32_t *node_ids, int32_t N, int32_t H, int32_t D, int32_t HD_active,
                float slope, const int32_t *ones_flag, void *stream);
/* K2 with the target-only part hoisted: P[v,h] = sum_d (g/S)*out (one pass over the vertices,
 * P is a scratch array of 2 N H floats: P and 1/S, or, for H = 8, D = 64, S | P in 64 bytes per vertex) and per edge T = ((sum_d g*feat[u]) / S - P) * A * slope, so the
 * per-edge out[v] row gather (half of K2's traffic) disappears.  Same outputs as stg_gat_bwd
 * (grad_feat bit-identical; grad_el / T regrouped sums -- the reference forms them with atomicAdd
 * in an undefined order).  Computes all H*D columns (no *_active argument).  grad_er (optional, [N,H]): the sum of
 * T over every vertex's in-edges, regrouped into the per-vertex pass as slope * (g . out - P * S) -- what stg_gat_bwd_er
 * computes from T (rounding noise around 0 in either form); with it T may be NULL and is then neither written nor
 * needed. */
int stg_gat_bwd_factored(const float *A, const float *S, const float *out, const float *g,
                         const float *feat, float *grad_feat, float *grad_el, float *T, float *P,
                         const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                         const int32_t *node_ids, int32_t N, int32_t H, int32_t D, float slope, float *grad_er,
                         const int32_t *ones_flag, void *stream);
/* K2 of the GATConv whose attention is uniform (every A = 1.0f: ones_flag clear) and whose features are feat = x fc.weight^T, at
 * H = 8, D = 64, fin = 64 (stg_gat_bwd_uniform_supported) -- BASELINE configs[2].  No row of width H D is gathered per edge:
 * the dot product of T is  g[v,h,:] . feat[u,h,:] = gW[v,h,:] . x[u,:]  with gW [H][N][fin] = g[:,h,:] W_h (the caller's batched
 * product), and grad_feat is only used through grad_feat W = sum over out-edges of gsW[v] (gsW = sum_h gW[v,h,:] / S[v,h]) and
 * grad_feat^T x = g^T xm (the forward's mean of x).  Call order:
 *   stg_gat_bwd_prepass(S, out, g, g_pre, pack, ...)      pack [N][16] = S | P; g_pre (nullable) = gradient through the fused elu;
 *                                                         grad_er (nullable) as stg_gat_bwd_factored
 *   gW = batched g_pre[:, h, :] @ W_h                     (library GEMM, H batches)
 *   stg_gat_bwd_uniform_edges(...)                        forward CSR: T [E][H], gsW [N][fin]; backward CSR: grad_el [N][H],
 *                                                         gxa [N][fin] = grad_feat W
 * When ones_flag is set (a non-finite score) the general unit runs inside the same call instead: it fills grad_feat [N][H D]
 * and T, gsW / gxa come out 0, and the caller adds grad_feat W (stg_gat_bwd_uniform_gx_fallback, a no-op otherwise) and takes
 * the weight gradient from grad_feat^T x instead of g^T xm -- both decided on the device (stg_gemm_tn_gated_f32). */
int stg_gat_bwd_uniform_supported(int32_t H, int32_t D, int32_t fin);
int stg_gat_bwd_prepass(const float *S, const float *out, const float *g, float *g_pre, float *pack, int32_t N, int32_t H,
                        int32_t D, float slope, float *grad_er, void *stream);
/* The first two steps above as ONE pass over g and out (H % 4 == 0 heads of D = 64 over fin = 64, N H D < 2^29; knob "rowgemm_x3" not
 * 1 or 3: stg_gat_bwd_prepass_heads_supported): g_pre, pack and grad_er as stg_gat_bwd_prepass (equal to fp32 rounding: the
 * per-head dot products add the same terms in another order), and gW [H][N][fin] = g_pre[:, h, :] W_h in the 3-term bf16 split
 * form of stg_rowgemm_heads_f32, four heads per launch.  out, g, g_pre, W, gW 16-byte aligned; g_pre == NULL: g is already the
 * gradient of `out`. */
/* The small products of that fold in ONE launch (W [H D][fin] = fc.weight, G [2H][fin] = [grad_el | grad_er]^T h, attn_l / attn_r [H][D]):
 * dattn_l[h, d] = sum_f W[h D + d, f] G[h, f] (dattn_r with G[H + h]); Aw [2H][fin] (nullable) with Aw[h, f] = sum_d W[h D + d, f]
 * attn_l[h, d] (rows H .. with attn_r); gw [H D][fin] (nullable) += attn_l[h, d] G[h, f] + attn_r[h, d] G[H + h, f].  D fin <= ~16 K. */
int stg_gat_attn_fold(const float *W, const float *G, const float *attn_l, const float *attn_r, float *dattn_l, float *dattn_r,
                      float *Aw, float *gw, int32_t H, int32_t D, int32_t fin, void *stream);
int stg_gat_bwd_prepass_heads_supported(int64_t N, int32_t H, int32_t D, int32_t fin);
int stg_gat_bwd_prepass_heads(const float *S, const float *out, const float *g, float *g_pre, float *pack, float *grad_er,
                              const float *W, float *gW, int64_t N, int32_t H, int32_t D, int32_t fin, float slope, void *stream);
int stg_gat_bwd_uniform_edges(const float *A, const float *pack, const float *gq, const float *feat, const float *x,
                              const float *gW, float *T, float *gsW, float *grad_feat, float *grad_el, float *gxa,
                              const int32_t *fwd_row_offsets, const int32_t *fwd_column_indices, const int32_t *fwd_eids,
                              const int32_t *fwd_node_ids, const int32_t *bwd_row_offsets, const int32_t *bwd_column_indices,
                              const int32_t *bwd_eids, const int32_t *bwd_node_ids, int32_t N, float slope,
                              const int32_t *ones_flag, void *stream);
int stg_gat_bwd_uniform_gx_fallback(const float *grad_feat, const float *W, float *gx, int32_t N, const int32_t *ones_flag,
                                    void *stream);
int stg_gat_bwd_er(const float *T, float *grad_er,
                   const int32_t *row_offsets, const int32_t *eids, const int32_t *node_ids,
                   int32_t N, int32_t H, int32_t H_active, void *stream);

/* The uniform-attention form of the layer (ABI 23).  The vertex function's `emb - max([emb])` is emb - emb (reference
 * nn/pytorch/static/gat_conv.py:50): with every score finite (*ones_flag == 0, stg_gat_score_flag) each
 * A is 1.0f and K1 is out[v] = sum_u (1.0f / S[v]) * feat[u] with feat = x W^T -- linear in x, so the gather can run
 * at the INPUT width:
 *   stg_gat_fwd_k1_uniform : xm[v, 0..F) = sum_{u in in(v)} (1.0f / S[v*H]) * x[u, 0..F)  (K1's own loop, one "head" of F
 *                            columns; S [N,H] from K0, all H equal).  Runs only if *ones_flag == 0, else returns at once.
 *   stg_gat_fc_out         : out = xm W^T (the kernel of stg_gat_fc_fwd without the projections; same shape support),
 *                            and, when act_out != NULL, act_out = elu(out) beside it (torch's elu, alpha 1: x <= 0 ?
 *                            exp(x) - 1 : x) -- the layer's `activation`.
 *   stg_gat_fwd_k1_scored  : the full-width K1 (stg_gat_fwd_k1 over all H*D columns) that OVERWRITES out (and act_out,
 *                            nullable) when *ones_flag != 0 -- some score is inf / NaN -- and returns at once otherwise.
 * Launched in this order the three leave exactly K1's result for non-finite scores and sum-then-product instead of
 * product-then-sum (fp32 rounding, ~1e-7 relative) for finite ones, at 1/8 of the gather bytes for 64 -> 8 x 64.
 * stg_gat_bwd_factored_elu: stg_gat_bwd_factored for a layer whose output went through that elu: g_act is the gradient
 * of elu(out); the per-vertex pass forms g_pre = g_act * (out <= 0 ? exp(out) : 1) (torch's elu_backward), stores it in
 * g_pre [N, H*D] and everything downstream uses it. */
int stg_gat_fwd_k1_uniform(const float *S, int32_t H, const float *x, float *xm, const int32_t *row_offsets,
                           const int32_t *column_indices, const int32_t *node_ids, int32_t N, int32_t F,
                           const int32_t *ones_flag, void *stream);
int stg_gat_fc_out(const float *xm, const float *W, float *out, float *act_out, int32_t N, int32_t fin, int32_t H,
                   int32_t D, void *stream);
/* The input side without its feat: stg_gat_fc_fwd accepts feat == NULL (el / er only, from the same accumulators) -- the
 * uniform-attention form never reads feat; stg_gat_fc_feat_if writes feat = x W^T afterwards only if *only_if != 0 (the flag of
 * stg_gat_score_flag: some score is not finite and the general units, which gather feat, are going to run). */
int stg_gat_fc_feat_if(const float *x, const float *W, float *feat, int32_t N, int32_t fin, int32_t H, int32_t D,
                       const int32_t *only_if, void *stream);
int stg_gat_fwd_k1_scored(const float *A, const float *S, const float *feat, float *out, float *act_out,
                          const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                          const int32_t *node_ids, int32_t N, int32_t H, int32_t D, const int32_t *ones_flag,
                          void *stream);
int stg_gat_bwd_factored_elu(const float *A, const float *S, const float *out, const float *g_act, float *g_pre,
                             const float *feat, float *grad_feat, float *grad_el, float *T, float *P,
                             const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                             const int32_t *node_ids, int32_t N, int32_t H, int32_t D, float slope, float *grad_er,
                             const int32_t *ones_flag, void *stream);

/* ------------------------------------------------- dense neighbour: weight gradient
 * C[M,N] = A[K,M]^T * B[K,N], all fp32 row-major [dev]; K = number of vertices (large), M, N =
 * feature widths (small).  This is dW = X^T dY of `torch.mm(h, self.weight)`
 * (nn/pytorch/static/gcn_conv.py:158) and of TGCN's gate Linears (nn/pytorch/temporal/tgcn.py:21-47),
 * which the reference leaves to cuBLAS.  Split-K over the whole chip on fp32 matrix cores
 * (exact fp32 fma chain per slice), slices added in a fixed order: deterministic, no atomics.
 * workspace [dev] must hold stg_gemm_tn_workspace_bytes(K, M, N) bytes.
 */
size_t stg_gemm_tn_workspace_bytes(int64_t K, int32_t M, int32_t N);
int stg_gemm_tn_f32(const float *A, const float *B, float *C, int64_t K, int32_t M, int32_t N,
                    void *workspace, size_t workspace_bytes, void *stream);

/* stg_gemm_tn_f32 decided on the device: the launch (slabs and reduction) runs only if *gate == 0 (gate_when = 1) or only if
 * *gate != 0 (gate_when = 2) and leaves C alone otherwise.  Two calls on the same word, one of each kind, with the same C: the
 * product that applies is chosen without a host read (inside a HIP graph too).  Workspace: stg_gemm_tn_workspace_bytes. */
int stg_gemm_tn_gated_f32(const float *A, const float *B, float *C, int64_t K, int32_t M, int32_t N, void *workspace,
                          size_t workspace_bytes, const int32_t *gate, int gate_when, void *stream);
/* Same, additionally colsum_A[m] = sum_k A[k][m] (the bias gradient that goes with the weight
 * gradient) from one extra MFMA per k-pair; colsum_A [dev, M floats]. */
int stg_gemm_tn_colsum_f32(const float *A, const float *B, float *C, float *colsum_A, int64_t K, int32_t M,
                           int32_t N, void *workspace, size_t workspace_bytes, void *stream);

/* GATConv's input side as one launch: feat [N, H*D] = x [N, fin] W^T (W [H*D, fin], the layer's bias-free fc:
 * reference nn/pytorch/static/gat_conv.py:43-48) and, from the accumulators, el / er [N, H] = sum_d feat[n,h,d] *
 * attn_{l,r}[h,d] (what stg_gat_proj_fwd computes from feat).  MFMA fp32; results agree with x @ W^T followed by
 * stg_gat_proj_fwd to fp32 rounding.  Supported: D = 64, even H, fin 32 or 64, H*64*(fin+8)*4 + 512 H <= 160 KB;
 * else STG_ERR_UNSUPPORTED (callers run the GEMM and stg_gat_proj_fwd). */
int stg_gat_fc_supported(int32_t fin, int32_t H, int32_t D);
int stg_gat_fc_fwd(const float *x, const float *W, const float *attn_l, const float *attn_r, float *feat,
                   float *el, float *er, int32_t N, int32_t fin, int32_t H, int32_t D, void *stream);

/* GATConv's attention projections and their backward (nn/pytorch/static/gat_conv.py:43-45; torch ops in the
 * reference), one streaming pass each over feat [N,H,D]:
 *   fwd: el[n,h] = sum_d feat[n,h,d] attn_l[h,d],  er likewise with attn_r                  (el, er: [N,H])
 *   bwd: dfeat = g + del (x) attn_l + der (x) attn_r   (g = gradient w.r.t. feat from stg_gat_bwd*, nullable = 0,
 *        may alias dfeat);  dattn_l[h,d] = sum_n del[n,h] feat[n,h,d],  dattn_r likewise    (fixed-order partial sums)
 * stg_gat_proj_supported(H, D) != 0 iff D % 4 == 0, D / 4 a power of two <= 64 and H D / 4 divides or is a
 * multiple (<= 16x) of 256. */
int stg_gat_proj_supported(int32_t H, int32_t D);
int stg_gat_proj_fwd(const float *feat, const float *attn_l, const float *attn_r, float *el, float *er,
                     int64_t N, int32_t H, int32_t D, void *stream);
size_t stg_gat_proj_bwd_workspace_bytes(int64_t N, int32_t H, int32_t D);
int stg_gat_proj_bwd(const float *feat, const float *attn_l, const float *attn_r, const float *del, const float *der,
                     const float *g, float *dfeat, float *dattn_l, float *dattn_r, int64_t N, int32_t H, int32_t D,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Y[N,M] = X[N,K] * op(W) + bias (bias nullable): forward / input-gradient GEMM of the dense layers
 * for N = number of vertices and small K, M.  op(W) = W [K,M] (trans_w = 0) or W^T with W [M,K]
 * (trans_w = 1, i.e. torch's Linear weight layout).  64-row X tile + W in LDS, fp32 matrix cores.
 * stg_rowgemm_supported(K, M) != 0 iff K % 4 == 0, K <= 192, M / 32 in {1,2,3,4,6} and the A tile plus the
 * accumulators fit the register budget (K / 2 + M / 2 <= 176). */
int stg_rowgemm_supported(int32_t K, int32_t M);
int stg_rowgemm_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K,
                    int32_t M, int trans_w, void *stream);
/* Same with an output row stride ldy >= M (floats): writes the column block Y[:, 0:M] of a wider row-major matrix,
 * so an output wider than one launch covers (M > 192) is produced in column slices of W / bias. */
int stg_rowgemm_strided_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K,
                            int32_t M, int32_t ldy, int trans_w, void *stream);

/* Y = act(X op(W) + bias) for K, M in {64, 128} (stg_rowgemm_act_supported) and many rows: the dense layer of the GCN / GAT
 * configs (gcn_conv.py:158-188: `torch.mm(h, self.weight)`, `+ self.bias`, activation) as ONE launch in the 16-row row-piece
 * layout of the step kernels; act = STG_ACT_NONE / STG_ACT_RELU.  X, W, Y 16-byte aligned, Y contiguous [N, M]. */
int stg_rowgemm_act_supported(int32_t K, int32_t M);
int stg_rowgemm_act_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K, int32_t M,
                        int trans_w, int act, void *stream);

/* The same product with the ReLU's sign pattern as ONE BIT per output element, so that the backward pass never re-reads the layer's
 * output for it (gcn_conv.py:186-188 `h = self.activation(h)`; autograd's threshold_backward reads [N, M] floats, 0.5 GB at cfg2):
 *   forward  (trans_w = 0, act = STG_ACT_RELU, bits_out given): Y = relu(X W + bias) and bits_out = [Y > 0];
 *   backward (trans_w = 1, act = STG_ACT_NONE, bits_in given): Y = (X W^T) * pattern -- the gradient with respect to the ReLU
 *   layer's pre-activation, formed in the launch of the layer ABOVE that computes its input gradient g W^T (M = the ReLU
 *   layer's width in both calls, same N).
 * bits: stg_rowgemm_bits_words(N) uint32 words (16 bytes per row, rounded up to 32 rows).  Element (row, col) is bit
 * 8 (col >> 5) + 4 ((row >> 3) & 1) + (col & 3) of word 64 (row >> 4) + (row & 7) + 8 ((col >> 4) & 1) + 16 ((col >> 2) & 3) -- the
 * arrangement in which a lane of the kernel holds its 32 outputs of a 16-row tile.  Always the 3-term bf16 split on the matrix
 * cores; stg_rowgemm_bits_supported: K, M in {64, 128}, N K and N M < 2^30, knob "rowgemm_x3" neither 1 nor 3. */
/* Y [heads][N][M]: Y_h = X[:, h K : (h + 1) K] W_h for the column blocks of X [N, heads K] and W [heads][K][M] -- the per-head
 * products of a multi-head layer's gradient with its fc weight (GATConv: g[:, h, :] W_h, stg_gat_bwd_uniform_edges' gW) as
 * `heads` launches of the split-form row product; K, M in {64, 128}, N heads K < 2^30 (stg_rowgemm_heads_supported). */
int stg_rowgemm_heads_supported(int64_t N, int32_t K, int32_t M, int32_t heads);
int stg_rowgemm_heads_f32(const float *X, const float *W, float *Y, int64_t N, int32_t K, int32_t M, int32_t heads, void *stream);
size_t stg_rowgemm_bits_words(int64_t N);
int stg_rowgemm_bits_supported(int64_t N, int32_t K, int32_t M);
int stg_rowgemm_act_bits_f32(const float *X, const float *W, const float *bias, float *Y, int64_t N, int32_t K, int32_t M,
                             int trans_w, int act, const uint32_t *bits_in, uint32_t *bits_out, void *stream);

/* C = sum_{t < T} A_t^T B_t (and colsum_A = sum_t colsum(A_t), nullable) in ONE launch: A, B are HOST
 * arrays of T <= 32 device pointers, every A_t [K,M], B_t [K,N].  The pointers travel by value in
 * the kernel arguments, so the call is HIP-graph capturable.  Used to turn a BPTT window's weight
 * gradient (one contribution per timestep) into a single split-K launch. */
size_t stg_gemm_tn_multi_workspace_bytes(int32_t T, int64_t K, int32_t M, int32_t N);
int stg_gemm_tn_multi_f32(const float *const *A, const float *const *B, int32_t T, float *C, float *colsum_A,
                          int64_t K, int32_t M, int32_t N, void *workspace, size_t workspace_bytes,
                          void *stream);

/* C [M,N] = (A * [mask > 0])^T B and, if colsum_A, its column sums sum_k (A * [mask > 0])[k][m]: the weight and bias
 * gradients of `act(X W + b)` with a ReLU (A = upstream gradient [K,M], mask = the layer's output [K,M], B = the layer's
 * input [K,N]) in one launch -- the masked gradient is formed in the operand registers and never written
 * (replaces stg_bias_act_bwd + stg_gemm_tn_colsum_f32 where nothing else needs the masked gradient).  All contiguous;
 * workspace as stg_gemm_tn_workspace_bytes(K, M, N). */
int stg_gemm_tn_relu_mask_f32(const float *A, const float *mask, const float *B, float *C, float *colsum_A,
                              int64_t K, int32_t M, int32_t N, void *workspace, size_t workspace_bytes, void *stream);

/* The same contraction with operands taken where the one-launch TGCN step leaves them (csrc/tgcn_step.hpp):
 * A_t [K, M] with row stride lda;  B_t [K, N] = [ op(b_t[:, 0:nsplit]) | b2_t[:, 0:N-nsplit] ] with row strides ldb /
 * ldb2 (nsplit a multiple of 32, or N: then B2 / ldb2 are unused), op = b_op applied to the values of b_t as they
 * are loaded: STG_GEMM_B_CLAMP: min(max(v, lo), hi), STG_GEMM_B_RELU: max(v, 0).  E.g. dWz = sum_t dzl_t^T
 * [clamp(x3_t[:, 0:C]) | H_t] (reference: the autograd of nn.Linear inside nn/pytorch/temporal/tgcn.py:24-25).
 * workspace: stg_gemm_tn_form_workspace_bytes(T, K, M, N, max(lda, ldb, ldb2)). */
#define STG_GEMM_B_NONE  0
#define STG_GEMM_B_CLAMP 1
#define STG_GEMM_B_RELU  2
size_t stg_gemm_tn_form_workspace_bytes(int32_t T, int64_t K, int32_t M, int32_t N, int32_t max_ld);
int stg_gemm_tn_form_f32(const float *const *A, int32_t lda, const float *const *B, int32_t ldb, int32_t nsplit,
                         const float *const *B2, int32_t ldb2, int32_t b_op, float lo, float hi, int32_t T, float *C,
                         float *colsum_A, int64_t K, int32_t M, int32_t N, void *workspace, size_t workspace_bytes,
                         void *stream);
/* The same contraction WITHOUT its final reduction: the split-K slabs stay in `workspace` (size as stg_gemm_tn_form_workspace_bytes)
 * and their count is written to *slabs; stg_gemm_tn_reduce_multi_f32 then sums the slabs of up to 8 products in ONE launch into
 * C[i] [M[i], N[i]] and colsum[i] [M[i]] (NULL where the product was run with want_colsum = 0) -- same arithmetic and order as the
 * one-product form.  A BPTT window's six weight gradients take one reduction launch instead of six. */
int stg_gemm_tn_form_partial_f32(const float *const *A, int32_t lda, const float *const *B, int32_t ldb, int32_t nsplit,
                                 const float *const *B2, int32_t ldb2, int32_t b_op, float lo, float hi, int32_t T,
                                 int32_t want_colsum, int64_t K, int32_t M, int32_t N, void *workspace, size_t workspace_bytes,
                                 int32_t *slabs, void *stream);
int stg_gemm_tn_reduce_multi_f32(int32_t count, const float *const *slabs, float *const *C, float *const *colsum,
                                 const int32_t *M, const int32_t *N, const int32_t *S, void *stream);
/* The same reduction with a product's C [M, N] leaving as M / block_rows[i] row blocks, each TRANSPOSED into its own contiguous
 * [N, block_rows[i]] array C_blocks[i * STG_GEMM_REDUCE_BLOCKS + b] (and its slice of the column sums into colsum_blocks[...]; all NULL
 * = no column sums; the product must then have been formed without them): three GCNConv layers' weight gradients computed as ONE
 * stacked product (dynamic-temporal TGCN: conv_z / conv_r / conv_h share their input) land in the layout of their parameters'
 * .grad without a transposing copy each.  block_rows[i] = 0 (or block_rows NULL): C[i] / colsum[i] as above.  Same sums, bit for bit. */
#define STG_GEMM_REDUCE_BLOCKS 4
int stg_gemm_tn_reduce_multi_blocks_f32(int32_t count, const float *const *slabs, float *const *C, float *const *colsum,
                                        const int32_t *M, const int32_t *N, const int32_t *S, const int32_t *block_rows,
                                        float *const *C_blocks, float *const *colsum_blocks, void *stream);

/* The whole forward chain of the six stages below in ONE launch for C = 32 or 64 (hidden width): bias + clamp,
 * the three gate GEMMs on the fp32 matrix cores with the gate weights (torch Linear layout [C][2C]) resident in
 * LDS, sigmoid / tanh and the GRU blend -- one wave per 32-row tile, see csrc/tgcn_cell_fused.hip.  Outputs are
 * exactly what prep_fwd + gates_fwd + update_fwd and the three Linears produce: CZ = [hz|H], CR = [hr|H],
 * CH = [hh|H*R] ([N,2C] each, read again by the weight gradients), Z, R, Ht, Hn ([N,C]).  Elementwise formulas
 * are identical; the GEMM k-order differs from rocBLAS', so results agree to fp32 rounding (tested at 1e-5). */
int stg_tgcn_cell_fused_supported(int32_t C);
/* ... and the backward chain likewise (update_bwd + gates_bwd + prep_bwd and the three input-gradient GEMMs
 * dhl Wh, dzl Wz, drl Wr, whose [N,2C] results never leave the chip): outputs dhl, dzl, drl [N,C] (the
 * pre-activation gradients the weight gradients contract with CH, CZ, CR), da3 [N,3C] (clamp mask applied) and
 * dH [N,C]. */
int stg_tgcn_cell_fused_bwd(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R,
                            const float *a3, const float *b3, const float *Wz, const float *Wr, const float *Wh,
                            float *dhl, float *dzl, float *drl, float *da3, float *dH, int64_t N, int32_t C,
                            float lo, float hi, void *stream);
int stg_tgcn_cell_fused_fwd(const float *a3, const float *b3, const float *H, const float *Wz, const float *bz,
                            const float *Wr, const float *br, const float *Wh, const float *bh, float *CZ,
                            float *CR, float *CH, float *Z, float *R, float *Ht, float *Hn, int64_t N, int32_t C,
                            float lo, float hi, void *stream);
/* stg_tgcn_cell_fused_bwd that also returns dx = da3 Wcat^T [N,Fin] -- the gradient reaching the aggregated input of
 * the fused gate aggregation (Wcat [Fin][3C] = the three GCNConv weights side by side): da3's row pieces are MFMA
 * operands at the moment they are stored, so the product rides along instead of a GEMM launch that re-reads da3.
 * Supported: C in {32, 64}, Fin = 32 (stg_tgcn_cell_fused_bwd_dx_supported). */
int stg_tgcn_cell_fused_bwd_dx_supported(int32_t C, int32_t Fin);
int stg_tgcn_cell_fused_bwd_dx(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R,
                               const float *a3, const float *b3, const float *Wz, const float *Wr, const float *Wh,
                               const float *Wcat, float *dhl, float *dzl, float *drl, float *da3, float *dH, float *dx,
                               int64_t N, int32_t C, int32_t Fin, float lo, float hi, void *stream);


/* ----------------------------------------------- one TGCN training step as one launch each way
 * The per-snapshot body of the temporal harnesses: nn/pytorch/temporal/tgcn.py:21-55 (three GCNConv gates over the same
 * graph and input, clamp to [lo, hi], gate Linears, GRU blend) under benchmarking/static-temporal-tgcn/seastar/model.py:6-18
 * (relu -> Linear(C, Fh) -> Linear(Fh, 1)) and its loop's `cost += mean((y_out - y[t]) ** 2)`; the dynamic-temporal model
 * (dynamic-temporal-tgcn/seastar/model.py:5-21) is the same step with head = 1.  Replaces, per snapshot, the three emitted
 * GCN units forward + backward, ~100 torch launches and their autograd nodes.  See csrc/tgcn_step.hip.
 * All pointers [dev], fp32 / int32, row-major contiguous; C = 64, Fin = Fh = 32 (stg_tgcn_step_supported).
 *
 * forward:  x [N,Fin] (NULL: a3 [N,3C] = A_hat (x Wcat) is given instead and no graph is read); forward CSR
 *   (row_offsets, column_indices, norm_col_edge = norm[col[e]] and ew_edge = w[eid[e]] in CSR order as for
 *   stg_gcn_agg_edge, ew_edge NULL = unweighted; node_ids NULL = rows in vertex order); norm [N]; H [N,C] (NULL = zeros);
 *   WcatT [3C,Fin] = [Wz_conv | Wr_conv | Wh_conv]^T, b3 [3C]; Wz/Wr/Wh [C,2C] and bz/br/bh [C] (torch Linear layout);
 *   head >= 1: W1 [Fh,C], b1 [Fh] -> y [N,Fh];  head == 2: W2 [Fh], b2 [1], target [N] -> y_out [N] and
 *   loss_partial [stg_tgcn_step_loss_partials(N)] (per-tile sums of (y_out - target)^2: stg_tgcn_window_loss adds them).
 *   outputs kept for the backward pass and the weight gradients: P [N,Fin] = A_hat x, x3 [N,3C] = P Wcat + b3 (before
 *   the clamp), Z, R, Ht, Hn, HR = H*R [N,C]; clamp_mask [N, 12] uint32 (NULL: not wanted): one bit per column of x3,
 *   set where lo <= x3 <= hi, in the kernels' own piece order (word 4 g + q: 16 bits for gate g, lane q of the row's
 *   four lanes) -- the backward launch then reads 48 bytes per row instead of x3's 12C.
 * backward: backward CSR (rows = sources) with its per-edge arrays; zn [N,Fin] = the NEXT step's z (NULL: none);
 *   g_y [N,Fh] = a gradient reaching y directly (NULL: none); dHn [N,C] = gradient reaching Hn from the next step (NULL: 0);
 *   g_cost [1]; the forward's Z, R, Ht, H, Hn, x3 (or clamp_mask, which is preferred when both are given), y_out, target; WzT/WrT/WhT [2C,C] (transposed gate weights),
 *   Wcat [Fin,3C], W1T [C,Fh], W2 [Fh].  head == 2: dyo = 2 (y_out - target) / N * g_cost; dyt = A_hat^T zn + g_y + dyo W2;
 *   dHn += (Hn > 0) (dyt W1).  outputs: dzl, drl, dhl [N,C] (pre-activation gradients of the gate Linears), da3 [N,3C]
 *   (gradient of x3 after the clamp mask), dH [N,C] (to the previous step's Hn), z [N,Fin] = da3 Wcat^T (NULL: not
 *   wanted) -- the gradient of this step's INPUT is A_hat^T z, taken by the previous step's backward launch (or by
 *   stg_gcn_agg_edge) --, dyt [N,Fh], dyo [N]. */
typedef struct stg_tgcn_step_fwd_args {
    const int32_t *row_offsets, *column_indices, *node_ids;
    const float *norm_col_edge, *ew_edge, *norm;
    const float *x, *a3, *H, *target;
    const float *WcatT, *b3, *Wz, *bz, *Wr, *br, *Wh, *bh, *W1, *b1, *W2, *b2;
    float *P, *x3, *Z, *R, *Ht, *Hn, *HR, *y, *y_out, *loss_partial;
    uint32_t *clamp_mask;
    int64_t N;
    int32_t C, Fin, Fh, head;
    float lo, hi;
    /* Reserved: must be NULL (ABI 22-25: the weight image of the retired bf16-split form). */
    const void *w_image;
    /* Optional (NULL: not used; ABI 24).  The FOLDED form of the forward launch: w_fold [3C][Fin + C], rows
     * g C + c = [ (Wc_g Wg[:, :C]^T)^T | Wg[:, C:] ] and b_fold [3C] = bc_g Wg[:, :C]^T + bg -- the conv output folded into the gate
     * Linears, formed once per window by the caller (stg_tgcn_fold_weights).  Needs x != NULL, head >= 1, x3 == NULL and all four of
     * w_fold, b_fold, fold_bound, fold_status.  The gate products run straight from P on the folded weights -- 320 fp32 matrix
     * instructions per tile instead of 512, no x3 formed.  The fold is only valid while no element of x3 is clamped, and x3 cannot be
     * looked at, so the launch bounds it: |P[r, :]|_1 fold_bound[0] + fold_bound[1] ({max |Wcat|, max |b3|}: stg_tgcn_fold_weights
     * writes them) outside [lo, hi] sets *fold_status |= 1 (sticky, never cleared by the library) and THAT launch's Z / R / Ht / Hn /
     * HR / y are wrong -- the caller must check it and redo the work without w_fold.  clamp_mask is not written (an inactive clamp's
     * mask is all ones: 0xffff in every word).  fold_status alone may be given to the un-folded form, which then ORs a 1 into it when an
     * element of x3 is clamped (callers that form the weight gradients from P^T d_g instead of x3 and da3 -- both then optional: x3
     * may be NULL when clamp_mask is given, stg_tgcn_step_bwd_args::da3 may be NULL when z is wanted -- need to know). */
    const float *w_fold, *b_fold;
    int32_t *fold_status;
    const float *fold_bound;
} stg_tgcn_step_fwd_args;
typedef struct stg_tgcn_step_bwd_args {
    const int32_t *row_offsets, *column_indices, *node_ids;
    const float *norm_col_edge, *ew_edge, *norm;
    const float *zn, *g_y, *dHn, *g_cost;
    const float *Z, *R, *Ht, *H, *Hn, *x3, *y_out, *target;
    const float *WzT, *WrT, *WhT, *Wcat, *W1T, *W2;
    float *dzl, *drl, *dhl, *da3, *dH, *z, *dyt, *dyo;
    const uint32_t *clamp_mask;
    int64_t N;
    int32_t C, Fin, Fh, head;
    float lo, hi;
    /* head == 1, optional (link_row_ptr NULL: not used): the node side of the link-prediction loss's backward taken in this
     * launch instead of stg_link_decode_bwd before it -- g_y[v] += sum over the label edges incident to v, in the order of
     * the node-sorted incidence list (link_row_ptr [N + 1], link_other, link_eid: stg_link_head_bwd's), of
     * (sigmoid(link_logits[e]) - link_target[e]) * g_cost[0] * link_inv_m * link_y[other end] (link_y = this step's y [N,Fh]). */
    const int32_t *link_row_ptr, *link_other, *link_eid;
    const float *link_y, *link_logits, *link_target;
    float link_inv_m;
    /* Reserved: must be NULL (ABI 22-25: the weight image of the retired bf16-split form). */
    const void *w_image;
    /* Optional (NULL: not used; ABI 24): [3 Fin][C], rows g Fin + f = the folded gate weights' P part transposed (stg_tgcn_fold_weights'
     * w_fold_t).  With it, da3 == NULL and head >= 1 the launch takes its FOLDED form: z = sum_g d_g w_fold_t_g^T instead of da3 Wcat^T
     * with da3_g = d_g Wg[:, :C] (320 matrix instructions per tile instead of 512), da3 is not formed and neither x3 nor clamp_mask
     * read -- exact for an inactive clamp (stg_tgcn_step_fwd_args::fold_status tells).  Wcat may then be NULL. */
    const float *w_fold_t;
    /* Row stride of dzl / drl / dhl in floats (ABI 25): 0 or C = three [N, C] matrices; 3 C = the column blocks of ONE [N, 3C] matrix
     * (dzl = D, drl = D + C, dhl = D + 2 C, say), so that a window's weight gradients contract [d_z | d_r] against [H | P] as one
     * operand -- the shared operand is read once (stg_gemm_tn_form_f32 with lda = 3 C). */
    int32_t ld_d;
} stg_tgcn_step_bwd_args;
int    stg_tgcn_step_supported(int32_t C, int32_t Fin, int32_t Fh);
size_t stg_tgcn_step_loss_partials(int64_t N);
int    stg_tgcn_step_fwd(const stg_tgcn_step_fwd_args *args, void *stream);
/* The weight layouts the two step launches take, from the modules' parameters, in one launch per window: Wcat [Fin,3C] =
 * [Wcz | Wcr | Wch] (GCNConv weights [Fin,C]), WcatT [3C,Fin], b3 [3C] = [bcz | bcr | bch], WzT / WrT / WhT [2C,C] (transposed gate
 * Linear weights [C,2C]), W1T [C,Fh] (transposed head weight [Fh,C]). */
int    stg_tgcn_pack_weights(const float *Wcz, const float *Wcr, const float *Wch, const float *bcz, const float *bcr,
                             const float *bch, const float *Wz, const float *Wr, const float *Wh, const float *W1, float *Wcat,
                             float *WcatT, float *b3, float *WzT, float *WrT, float *WhT, float *W1T, int32_t C, int32_t Fin,
                             int32_t Fh, void *stream);
int    stg_tgcn_step_bwd(const stg_tgcn_step_bwd_args *args, void *stream);
/* The gate and conv parameter gradients of a window from contractions that need neither x3 nor da3 (ABI 24).  Tables of three
 * device pointers, one per gate (z, r, h).  Inputs: R_g [C, C + Fin] = d_g^T [Hx | P] and cs_g [C] = column sums of d_g (d_g: gradient
 * of the gate's pre-activation over the window's rows -- stg_gemm_tn_form_f32 with the column sums), the conv weight Wc_g [Fin, C]
 * and bias bc_g [C], the gate Linear's weight Wg [C, 2C].  Outputs (fully overwritten): dWg [C, 2C] = [R_g[:, C:] Wc_g + cs_g bc_g^T
 * | R_g[:, :C]], dbg [C] = cs_g, dWc_g [Fin, C] = R_g[:, C:]^T Wg[:, :C], dbc_g [C] = cs_g Wg[:, :C] -- exact when no element of x3 was
 * clamped (stg_tgcn_step_fwd_args::fold_status tells).  One launch, sums in index order. */
/* ... and the gate Linears with the conv folded in, once per window, for the folded forms of the forward step launch
 * (stg_tgcn_step_fwd_args::w_fold): w_fold [3C][Fin + C], rows g C + c = [ Wg[c, :C] . Wc_g[f, :] (f < Fin) | Wg[c, C:] ];
 * b_fold [3C] = Wg[:, :C] bc_g + bg;  bound [2] = {max |Wc_g|, max |bc_g|} over the gates (stg_tgcn_step_fwd_args::fold_bound);
 * w_fold_t [3 Fin][C] (nullable), rows g Fin + f = w_fold[g C + :, f] (stg_tgcn_step_bwd_args::w_fold_t).
 * Tables of three device pointers (gates z, r, h): Wc_g [Fin, C], bc_g [C], Wg [C, 2C], bg [C]. */
int    stg_tgcn_fold_weights(const float *const *Wc, const float *const *bc, const float *const *Wg, const float *const *bg,
                             float *w_fold, float *b_fold, float *bound, float *w_fold_t, int32_t C, int32_t Fin, void *stream);
int    stg_tgcn_unfold_gate_grads(const float *const *R, const float *const *cs, const float *const *Wc, const float *const *bc,
                                  const float *const *Wg, float *const *dWg, float *const *dbg, float *const *dWc,
                                  float *const *dbc, int32_t C, int32_t Fin, void *stream);
/* cost[0] = sum over the window's `steps` steps, in order, of (sum of that step's partials) / N; step_loss [steps]
 * (required: the terms, and the scratch of the final sum).  partials: `steps` rows of step_stride floats. */
int    stg_tgcn_window_loss(const float *partials, int32_t steps, int64_t N, int64_t step_stride, float *step_loss,
                            float *cost, void *stream);

/* norm[i] = d_i^-0.5 (0 where d_i = 0), d_i = degrees[i] or row_offsets[i + 1] - row_offsets[i] (degrees NULL): the
 * scripts' `norm = torch.pow(in_degrees, -0.5); norm[isinf(norm)] = 0` (benchmarking/gcn/seastar/train.py:53-57) in one
 * launch; 1 / sqrt(d) correctly rounded. */
int    stg_degree_norm_f32(const int32_t *degrees, const int32_t *row_offsets, float *norm, int64_t N, void *stream);
/* cost[0] = sum over `steps` rows of `partials` (count values each, rows step_stride floats apart), in order, of
 * (row sum, fixed order) * inv_n; step_loss [steps] (required) gets the terms.  The general form of
 * stg_tgcn_window_loss: the dynamic-temporal loop's `cost += BCEWithLogitsLoss()(...)` over a window. */
int    stg_partial_sums_loss(const float *partials, int32_t steps, int32_t count, int64_t step_stride, float inv_n,
                             float *step_loss, float *cost, void *stream);
/* The link-prediction decoder and loss of benchmarking/dynamic-temporal-tgcn/seastar/model.py:19-21 + train loop on their
 * own (stg_link_head_fwd / _bwd also run the relu -> Linear half, which the one-launch step does itself):
 *   fwd: logits[e] = <y[src_e], y[dst_e]>, edge_index int64 [2, M] (sources then destinations);
 *        partial[(M + 31) / 32] = per-workgroup sums of the stable BCE-with-logits terms;
 *   bwd: dy [N, F] = sum over the label edges incident to each node, in the order of the incidence list
 *        (row_ptr [N + 1], other, eid: see stg_link_head_bwd), of (sigmoid(logit) - target) * g_loss / M * y[other].
 * F = 32. */
int    stg_link_decode_fwd(const float *y, const int64_t *edge_index, const float *target, float *logits, float *partial,
                           int64_t M, int32_t F, void *stream);
/* The same for up to 32 snapshots of a BPTT window in one launch (host arrays of per-snapshot device pointers, passed by value:
 * capturable): every snapshot has M label edges; partial[t] has (M + 31) / 32 entries. */
int    stg_link_decode_fwd_multi(int32_t count, const float *const *y, const int64_t *const *edge_index, const float *const *target,
                                 float *const *logits, float *const *partial, int64_t M, int32_t F, void *stream);
int    stg_link_decode_bwd(const float *g_loss, const float *y, const float *logits, const float *target,
                           const int32_t *row_ptr, const int32_t *other, const int32_t *eid, float *dy, int64_t N, int64_t M,
                           int32_t F, void *stream);

/* ----------------------------------------------- dense neighbour: softmax cross-entropy
 * `nn.CrossEntropyLoss()(logits, labels)` of the GCN training scripts (benchmarking/gcn/seastar/train.py:63-101),
 * mean over the n rows, one launch each way (+ a one-workgroup finish): see csrc/xent.hip.
 *   fwd: lse[i] = logsumexp(logits[i, :]) [n] (kept for the backward); a row is COUNTED when 0 <= labels[i] < K;
 *        loss[0] = sum over counted rows of (lse[i] - logits[i, labels[i]]) / n_counted[0], n_counted[0] = their number
 *        (as a float; kept for the backward).  labels[i] == -100 (nn.CrossEntropyLoss's default ignore_index) is not
 *        counted, as in torch; any other label outside [0, K) (torch: device assert) is not counted either and sets
 *        status[0] |= 1: OR-ed into, never cleared here -- the caller zeroes the word once and may keep it across
 *        calls (a sticky flag, no launch per call).  labels int64 [n].
 *   bwd: dlogits[i, c] = (exp(logits[i, c] - lse[i]) - [c == labels[i]]) * g_loss[0] / n_counted[0] for counted rows
 *        i < n, 0 for rows not counted and for the rows n <= i < n_total (the loss of the scripts is taken on the
 *        train-mask prefix of an [n_total, K] matrix).
 * All [dev]; workspace: stg_xent_workspace_bytes(n, K). */
size_t stg_xent_workspace_bytes(int64_t n, int32_t K);
int stg_xent_fwd(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted, int32_t *status,
                 int64_t n, int32_t K, void *workspace, size_t workspace_bytes, void *stream);
int stg_xent_bwd(const float *g_loss, const float *logits, const int64_t *labels, const float *lse,
                 const float *n_counted, float *dlogits, int64_t n, int64_t n_total, int32_t K, void *stream);
/* The same gradient together with its column sums colsum [K] = dlogits.sum(0) -- the bias gradient of the layer that produced the
 * logits (gcn_conv.py:186 `h + self.bias`), which otherwise re-reads the whole gradient for them.  K % 4 == 0 and K <= 8 x the
 * kernel's lanes per row (stg_xent_bwd_colsum_workspace_bytes returns 0 for a shape that is not covered); deterministic. */
size_t stg_xent_bwd_colsum_workspace_bytes(int64_t n_total, int32_t K);
int stg_xent_bwd_colsum(const float *g_loss, const float *logits, const int64_t *labels, const float *lse,
                        const float *n_counted, float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K,
                        void *workspace, size_t workspace_bytes, void *stream);
/* Forward and gradient in ONE pass over the logits, for a loss that is going to be differentiated (train.py:63-101:
 * `loss = loss_fcn(...)`, `loss.backward()`): loss, lse and n_counted as stg_xent_fwd, and dlogits [n_total, K] / colsum [K] as
 * stg_xent_bwd_colsum would return them for g_loss = 1 -- a count launch over the labels first (the gradient's 1 / n_counted must
 * be known before the first row), then each row in registers once.  stg_xent_scale_grad multiplies dlogits (and colsum, nullable)
 * by g_loss [dev scalar] in place in the backward pass; when g_loss is exactly 1 -- what `loss.backward()` passes -- every
 * workgroup reads it and leaves.  Shapes as stg_xent_bwd_colsum (workspace bytes 0 = not covered). */
size_t stg_xent_fwd_grad_workspace_bytes(int64_t n_total, int32_t K);
int stg_xent_fwd_grad(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted, int32_t *status,
                      float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K, void *workspace,
                      size_t workspace_bytes, void *stream);
int stg_xent_scale_grad(float *dlogits, float *colsum, const float *g_loss, int64_t n_total, int32_t K, void *stream);
/* The same loss and gradient for a SMALL matrix (stg_xent_small_supported: K <= 64 and about n_total * K' <= 32768, K' = K rounded up
 * to a power of two -- one workgroup holds it in registers; the 2708 x 7 logits of
 * Cora, benchmarking/gcn/seastar/train.py:63-101) as ONE one-workgroup launch each way: fwd = stg_xent_fwd (no workspace, no finish
 * launch), bwd = stg_xent_bwd_colsum for any K (colsum nullable).  Inside a replayed HIP graph a launch costs ~ 4.5 us whatever
 * it does; the general entry points need five for what these two do.  Sums in a fixed order: run-to-run identical; against the
 * general path equal to fp32 rounding (another order of the same additions). */
/* The backward of a SMALL dense layer y = x W (nn/pytorch/static/gcn_conv.py:158 `torch.mm(h, self.weight)`; x [N, K], W [K, M], g = dy
 * [N, M]) as ONE one-workgroup launch: gx [N, K] = g W^T and gw [K, M] = x^T g (N <= 65536, K, M <= 16: Cora's second
 * layer, 2708 x 16 -> 7).  Fixed summation order (run-to-run identical); against the library GEMMs equal to fp32 rounding.
 * relu_colsum [K] (nullable): x is the OUTPUT of a ReLU layer (gcn_conv.py:185-188 below this one): gx comes out as
 * (g W^T) * [x > 0], the gradient of that layer's pre-activation, and relu_colsum = its column sums, that layer's bias gradient. */
/* c [Ka, Mb] = a^T b for a [N, Ka] of any width, b [N, Mb <= 16], N <= 65536 (the weight gradient x^T g of a layer on a small graph:
 * Cora's first, 1433 x 2708 by 2708 x 16) -- one workgroup per 16 columns of a; fixed summation order. */
int stg_gemm_tn_small_supported(int64_t N, int32_t Ka, int32_t Mb);
int stg_gemm_tn_small_f32(const float *a, const float *b, float *c, int64_t N, int32_t Ka, int32_t Mb, void *stream);
int stg_mm_bwd_small_supported(int64_t N, int32_t K, int32_t M);
int stg_mm_bwd_small(const float *g, const float *x, const float *W, float *gx, float *gw, float *relu_colsum, int64_t N, int32_t K,
                     int32_t M, void *stream);
int stg_xent_small_supported(int64_t n_total, int32_t K);
int stg_xent_small_fwd(const float *logits, const int64_t *labels, float *lse, float *loss, float *n_counted, int32_t *status,
                       int64_t n, int32_t K, void *stream);
int stg_xent_small_bwd(const float *g_loss, const float *logits, const int64_t *labels, const float *lse, const float *n_counted,
                       float *dlogits, float *colsum, int64_t n, int64_t n_total, int32_t K, void *stream);

/* ----------------------------------------------- dense neighbour: the TGCN harness head
 * The model head and loss of the static-temporal TGCN training step
 * (benchmarking/static-temporal-tgcn/seastar/model.py:6-18: relu -> Linear(C, F) -> Linear(F, 1); train.py:
 * `cost = cost + torch.mean((y_out - y[t]) ** 2)`) as one launch forward and one backward, see csrc/tgcn_head.hip.
 * h [N,C]; W1 [F,C], b1 [F] (torch.nn.Linear layout); W2 [1,F], b2 [1]; target [N]; all [dev] fp32, 16-byte aligned.
 *   fwd: r = relu(h) [N,C], y = r W1^T + b1 [N,F], y_out = y W2^T + b2 [N], loss[0] = mean((y_out - target)^2);
 *        workspace: stg_tgcn_head_workspace_bytes(N) (per-tile partial sums, added in a fixed order).
 *   bwd: g_loss [1] = d cost / d loss, g_y [N,F] = gradient reaching y from its other consumer (the next step's
 *        input), g_yout [N] likewise for y_out; each may be NULL (= zero).  dyo = 2 (y_out - target) / N * g_loss
 *        + g_yout [N]; dyt = g_y + dyo W2 [N,F]; dh = (h > 0) (dyt W1) [N,C].  The weight gradients are
 *        dW1 = dyt^T r, db1 = colsum(dyt), dW2 = dyo^T y, db2 = sum(dyo) (stg_gemm_tn_*).
 * Supported: C in {32, 64, 128}, F = 32, one output column (stg_tgcn_head_supported); anything else is the
 * caller's torch composition. */
int stg_tgcn_head_supported(int32_t C, int32_t F, int32_t O);
size_t stg_tgcn_head_workspace_bytes(int64_t N);
int stg_tgcn_head_fwd(const float *h, const float *W1, const float *b1, const float *W2, const float *b2,
                      const float *target, float *r, float *y, float *y_out, float *loss, int64_t N, int32_t C,
                      int32_t F, void *workspace, size_t workspace_bytes, void *stream);
/* ... with the training loop's `cost = cost + loss` folded in: loss[0] = loss_in[0] + mean(...) (loss_in may be NULL). */
int stg_tgcn_head_fwd_acc(const float *h, const float *W1, const float *b1, const float *W2, const float *b2,
                          const float *target, const float *loss_in, float *r, float *y, float *y_out, float *loss,
                          int64_t N, int32_t C, int32_t F, void *workspace, size_t workspace_bytes, void *stream);
int stg_tgcn_head_bwd(const float *g_loss, const float *g_y, const float *g_yout, const float *h, const float *y_out,
                      const float *target, const float *W1, const float *W2, float *dh, float *dyt, float *dyo,
                      int64_t N, int32_t C, int32_t F, void *stream);

/* The link-prediction head of the dynamic-temporal harness (benchmarking/dynamic-temporal-tgcn/seastar/model.py:5-21:
 * relu -> Linear(C, F); decode = (z[src] * z[dst]).sum(-1); train.py: BCEWithLogitsLoss, mean over the M label edges).
 *   fwd: r = relu(h) [N,C], y = r W1^T + b1 [N,F], logits[e] = <y[src[e]], y[dst[e]]> [M],
 *        loss[0] = (loss_in ? loss_in[0] : 0) + mean(max(x,0) - x t + log1p(exp(-|x|))) -- loss_in: the loop's running
 *        cost; src / dst int64 [M] (a [2,M] index tensor's rows).
 *   bwd: dy[v] = g_y[v] + sum over the label edges incident to v of (sigmoid(logit) - t) / M * g_loss * y[other end],
 *        taken in the order of a node-sorted incidence list (row_ptr [N+1], other [2M], eid [2M], int32, built once
 *        per index tensor by the caller) -- no atomics; dyt = dy; dh = (h > 0) (dy W1).  g_y / g_loss may be NULL.
 * Weight gradients as for stg_tgcn_head_*: dW1 = dyt^T r, db1 = colsum(dyt).  C in {32, 64, 128}, F = 32. */
int stg_link_head_supported(int32_t C, int32_t F);
size_t stg_link_head_workspace_bytes(int64_t M);
int stg_link_head_fwd(const float *h, const float *W1, const float *b1, const int64_t *src, const int64_t *dst,
                      const float *target, const float *loss_in, float *r, float *y, float *logits, float *loss,
                      int64_t N, int64_t M, int32_t C, int32_t F, void *workspace, size_t workspace_bytes, void *stream);
int stg_link_head_bwd(const float *g_loss, const float *g_y, const float *h, const float *y, const float *logits,
                      const float *target, const int32_t *row_ptr, const int32_t *other, const int32_t *eid,
                      const float *W1, float *dy, float *dh, float *dyt, int64_t N, int64_t M, int32_t C, int32_t F,
                      void *stream);

/* ----------------------------------------------- dense neighbour: TGCN row-local glue
 * Fused elementwise stages of one TGCN step (nn/pytorch/temporal/tgcn.py:21-55); the three gate
 * GEMMs between them stay on rocBLAS.  C = hidden width (multiple of 4), all [dev] fp32 row-major,
 * 16-byte aligned.  a3 [N,3C] = aggregated X [Wz|Wr|Wh]; b3 [3C]; H, Z, R, Ht, Hn, zl, rl, hl [N,C];
 * CZ, CR, CH [N,2C] are the GEMM operands [hz|H], [hr|H], [hh|H*R] written in place (no cat).
 *   prep_fwd  : h = clamp(a3 + b3, lo, hi); CZ = [hz|H]; CR = [hr|H]; CH[:, :C] = hh
 *   gates_fwd : Z = sigmoid(zl); R = sigmoid(rl); CH[:, C:] = H * R
 *   update_fwd: Ht = tanh(hl); Hn = Z*H + (1-Z)*Ht
 *   update_bwd: dhl = dHn (1-Z)(1-Ht^2); dzl = dHn (H-Ht) Z (1-Z); dH = dHn Z
 *   gates_bwd : drl = dCH[:, C:] H R (1-R); dH += dCH[:, C:] R
 *   prep_bwd  : da3 = [dCZ[:, :C]|dCR[:, :C]|dCH[:, :C]] where lo <= a3+b3 <= hi else 0;
 *               dH += dCZ[:, C:] + dCR[:, C:]
 */
int stg_tgcn_cell_prep_fwd(const float *a3, const float *b3, const float *H, float *CZ, float *CR, float *CH,
                           int64_t N, int32_t C, float lo, float hi, void *stream);
int stg_tgcn_cell_gates_fwd(const float *zl, const float *rl, const float *H, float *Z, float *R, float *CH,
                            int64_t N, int32_t C, void *stream);
int stg_tgcn_cell_update_fwd(const float *hl, const float *Z, const float *H, float *Ht, float *Hn, int64_t N,
                             int32_t C, void *stream);
int stg_tgcn_cell_update_bwd(const float *dHn, const float *Z, const float *H, const float *Ht, float *dhl,
                             float *dzl, float *dH, int64_t N, int32_t C, void *stream);
int stg_tgcn_cell_gates_bwd(const float *dCH, const float *R, const float *H, float *drl, float *dH, int64_t N,
                            int32_t C, void *stream);
int stg_tgcn_cell_prep_bwd(const float *dCZ, const float *dCR, const float *dCH, const float *a3, const float *b3,
                           float *da3, float *dH, int64_t N, int32_t C, float lo, float hi, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STGRAPH_HIP_H */
