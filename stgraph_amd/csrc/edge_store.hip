// Dynamic edge store -- the MI355X counterpart of the reference's PCSR class
// (graph/dynamic/pcsr/pcsr.cu:273-939; driven by graph/dynamic/pcsr/pcsr_graph.py:46-166).
//
// What the reference does per timestamp: insert / delete the update lists one edge at a time into a
// host packed-memory array (O(log^2) amortised each, pointer chasing), relabel every slot
// (label_edges: O(capacity)), walk the whole array again to emit a CSR into pinned memory
// (build_csr / build_reverse_csr: O(capacity)), then four cudaMemcpy H2D.  Since the CSR is re-emitted
// from scratch at every step anyway, the gapped array buys nothing here; what the kernels need is the
// emitted CSR, which for a valid update stream is a pure function of the current edge SET
// (tests/test_oracle_pcsr.py::test_pcsr_csr_is_the_static_csr_with_reversed_rows_and_one_based_eids):
//
//   forward  CSR: rows = dst, columns = src DESCENDING (the PMA row is emitted back to front,
//                 pcsr.cu:842-853), eids = 1 + rank of (dst, src) in ascending order (label_edges)
//   backward CSR: rows = src, columns = dst DESCENDING, eids = the same labels       (pcsr.cu:791-804)
//   node_ids    : rows by non-increasing length (ties unspecified in the reference; ascending id here)
//
// State here: the edge set as TWO dense sorted arrays of packed keys, resident in HBM:
//   keys_fwd = (dst << 32 | src) ascending,   keys_bwd = (src << 32 | dst) ascending.
// update : new = (old \ del) U add by ONE scatter pass per orientation -- every surviving old key and
//          every added key computes its output slot from binary searches over the (small, sorted,
//          L2-resident) batches; no atomics, no temporaries of size E, deterministic.  16 B of HBM
//          traffic per stored edge per orientation.
// emit   : row offsets by binary search over the keys, then one pass that writes column / label into
//          the row-reversed slot.  Labels of the backward CSR are found by searching keys_fwd.
// Everything is stream ordered and capturable; violations of the stream contract (adding a present
// edge, deleting an absent one, a vertex id out of range) are reported through a device status word.
#include "stg_common.hpp"
#include "csr_kernels.hpp"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <numeric>
#include <vector>

namespace stg {
namespace {

constexpr int kStoreBits = 32;

__device__ __forceinline__ int64_t lower_bound_dev(const uint64_t *__restrict__ a, int64_t n, uint64_t k)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Two independent lower bounds in ONE loop: a binary search is a chain of dependent loads (18 of them over 250 K keys, each an
// L2 round trip), and two searches written one after the other are a chain of both lengths added.
__device__ __forceinline__ void lower_bound2_dev(const uint64_t *__restrict__ a, int64_t na, uint64_t ka, const uint64_t *__restrict__ b,
                                                 int64_t nb, uint64_t kb, int64_t &ra, int64_t &rb)
{
    int64_t alo = 0, ahi = na, blo = 0, bhi = nb;
    while (alo < ahi || blo < bhi) {
        const int64_t am = (alo + ahi) >> 1, bm = (blo + bhi) >> 1;
        const bool ago = alo < ahi, bgo = blo < bhi;
        const uint64_t av = ago ? a[am] : 0, bv = bgo ? b[bm] : 0;
        if (ago) {
            if (av < ka) alo = am + 1;
            else ahi = am;
        }
        if (bgo) {
            if (bv < kb) blo = bm + 1;
            else bhi = bm;
        }
    }
    ra = alo;
    rb = blo;
}

// Pack an update batch in both orientations.
__global__ void pack_batch(const int *__restrict__ src, const int *__restrict__ dst, int64_t n, int N,
                           uint64_t *__restrict__ kf, uint64_t *__restrict__ kb, int *__restrict__ status)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned s = (unsigned)src[i], d = (unsigned)dst[i];
        if (s >= (unsigned)N || d >= (unsigned)N) atomicOr(status, 1);
        kf[i] = ((uint64_t)d << kStoreBits) | s;
        kb[i] = ((uint64_t)s << kStoreBits) | d;
    }
}

// Surviving old keys: slot = i - #(deleted keys below) + #(added keys below).  A workgroup owns a
// contiguous tile of old keys; four lanes bracket the tile inside the two batches first, so the per-key
// searches run over a handful of batch entries (tile x churn) that stay in L1 instead of over the
// whole batch.
constexpr int kMergeItems = 8;
constexpr int kMergeSlice = 1024;        // batch entries per tile kept in LDS (2 x 8 KiB)

// lower (UPPER = false) / upper bound of k in a[0 .. n) by one WAVE: every round the 64 lanes probe the last entries of 64 equal
// chunks and a ballot keeps one chunk -- three dependent loads for the 6 K entries of a batch where a lane's binary search has 13.
template <bool UPPER>
__device__ __forceinline__ int64_t wave_bound_dev(const uint64_t *__restrict__ a, int64_t n, uint64_t k)
{
    const int lane = threadIdx.x & (kWave - 1);
    int64_t lo = 0, hi = n;                                // the answer is in [lo, hi]
    while (hi - lo > kWave) {
        const int64_t chunk = (hi - lo + kWave - 1) / kWave;
        const int64_t last = std::min<int64_t>(hi, lo + (lane + 1) * chunk) - 1;   // last entry of this lane's chunk
        const bool have = lo + lane * chunk < hi;
        const uint64_t v = have ? a[last] : ~0ull;
        const bool below = have && (UPPER ? v <= k : v < k);      // the whole chunk is below the bound
        const int c = __popcll(__ballot(below));                  // chunks entirely below: they form a prefix (a is sorted)
        lo = std::min<int64_t>(hi, lo + c * chunk);
        hi = std::min<int64_t>(hi, lo + chunk);
    }
    const bool have = lo + lane < hi;
    const uint64_t v = have ? a[lo + lane] : ~0ull;
    return lo + __popcll(__ballot(have && (UPPER ? v <= k : v < k)));
}

__device__ __forceinline__ void scatter_old_tile(const uint64_t *__restrict__ old, int64_t E,
                                                 const uint64_t *__restrict__ add, int64_t na,
                                                 const uint64_t *__restrict__ del, int64_t nd,
                                                 uint64_t *__restrict__ out, int64_t E_out,
                                                 int *__restrict__ status, int64_t tile)
{
    __shared__ int64_t bounds[4];
    const int64_t base = tile * (kBlock * kMergeItems);
    const int64_t last = min(base + (int64_t)kBlock * kMergeItems, E) - 1;
    {                                                      // wave w of the four: bracket w, searched by the whole wave
        const int w = threadIdx.x / kWave;
        const bool hi = w & 1, use_del = w & 2;
        const uint64_t *arr = use_del ? del : add;
        const int64_t n = use_del ? nd : na;
        const int64_t r = hi ? wave_bound_dev<true>(arr, n, old[last]) : wave_bound_dev<false>(arr, n, old[base]);
        if ((threadIdx.x & (kWave - 1)) == 0) bounds[w] = r;
    }
    __syncthreads();
    const int64_t a0 = bounds[0], a1 = bounds[1], d0 = bounds[2], d1 = bounds[3];
    // the slices of the two batches that can touch this tile, staged in LDS when they fit (the usual case:
    // tile x churn entries); the per-key searches then never leave the CU
    __shared__ uint64_t slice[2][kMergeSlice];
    const bool staged = (a1 - a0) <= kMergeSlice && (d1 - d0) <= kMergeSlice;
    if (staged) {
        for (int64_t t = threadIdx.x; t < a1 - a0; t += kBlock) slice[0][t] = add[a0 + t];
        for (int64_t t = threadIdx.x; t < d1 - d0; t += kBlock) slice[1][t] = del[d0 + t];
        __syncthreads();
    }
    uint64_t key[kMergeItems];
#pragma unroll
    for (int j = 0; j < kMergeItems; ++j) {                            // all loads of the tile in flight first
        const int64_t i = base + (int64_t)j * kBlock + threadIdx.x;
        key[j] = i < E ? old[i] : 0;
    }
    auto place = [&](const uint64_t *addp, const uint64_t *delp) {
        const int64_t an = a1 - a0, dn = d1 - d0;
#pragma unroll
        for (int j = 0; j < kMergeItems; ++j) {
            const int64_t i = base + (int64_t)j * kBlock + threadIdx.x;
            if (i >= E) break;
            const uint64_t k = key[j];
            const int64_t d = lower_bound_dev(delp, dn, k);
            if (d < dn && delp[d] == k) continue;                          // deleted
            const int64_t a = lower_bound_dev(addp, an, k);
            if (a < an && addp[a] == k) { atomicOr(status, 2); continue; } // adding an edge that is present
            const int64_t o = i - (d0 + d) + (a0 + a);
            if (o < E_out) out[o] = k;
        }
    };
    if (staged) place(slice[0], slice[1]);                                 // ds_read searches
    else place(add + a0, del + d0);
}

__global__ __launch_bounds__(kBlock) void scatter_old(const uint64_t *__restrict__ old, int64_t E,
                                                      const uint64_t *__restrict__ add, int64_t na,
                                                      const uint64_t *__restrict__ del, int64_t nd,
                                                      uint64_t *__restrict__ out, int64_t E_out,
                                                      int *__restrict__ status)
{
    scatter_old_tile(old, E, add, na, del, nd, out, E_out, status, (int64_t)blockIdx.x);
}

// Added keys: slot = j + #(old keys below) - #(deleted keys below); deleted keys: must exist.
__device__ __forceinline__ void scatter_add_range(const uint64_t *__restrict__ old, int64_t E,
                                                  const uint64_t *__restrict__ add, int64_t na,
                                                  const uint64_t *__restrict__ del, int64_t nd,
                                                  uint64_t *__restrict__ out, int64_t E_out, int *__restrict__ status,
                                                  int64_t block, int64_t nblocks, const int *__restrict__ ro_old = nullptr, int n_rows = 0)
{
    // ro_old (nullable): the row offsets of `old` -- a key's place among the old keys is then searched inside its own row
    // (two loads + ~4 steps instead of 18 dependent steps over the whole array)
    const int64_t stride = nblocks * blockDim.x;
    for (int64_t j = block * blockDim.x + threadIdx.x; j < na + nd; j += stride) {
        const uint64_t kk = j < na ? add[j] : del[j - na];
        int64_t rlo = 0, rn = E;
        if (ro_old) {
            const int64_t row = (int64_t)(kk >> kStoreBits);
            if (row < n_rows) {                                // (a key past the last row: searched over the whole array, reported as before)
                rlo = ro_old[row];
                rn = (int64_t)ro_old[row + 1] - rlo;
            }
        }
        if (j < na) {
            const uint64_t k = kk;
            if (j > 0 && add[j - 1] >= k) { atomicOr(status, add[j - 1] == k ? 2 : 16); continue; }   // duplicate / unsorted
            int64_t o_lt, d_lt;
            lower_bound2_dev(old + rlo, rn, k, del, nd, k, o_lt, d_lt);
            o_lt += rlo;
            if (d_lt < nd && del[d_lt] == k) atomicOr(status, 8);                  // added and deleted at once
            const int64_t o = j + o_lt - d_lt;
            if (o >= 0 && o < E_out) out[o] = k;
        } else {
            const int64_t q = j - na;
            const uint64_t k = kk;
            const int64_t o = rlo + lower_bound_dev(old + rlo, rn, k);
            if (q > 0 && del[q - 1] > k) atomicOr(status, 16);                              // batch not sorted
            if (o >= E || old[o] != k || (q > 0 && del[q - 1] == k)) atomicOr(status, 4);   // deleting an absent edge
        }
    }
}

__global__ void scatter_add_check_del(const uint64_t *__restrict__ old, int64_t E,
                                      const uint64_t *__restrict__ add, int64_t na,
                                      const uint64_t *__restrict__ del, int64_t nd,
                                      uint64_t *__restrict__ out, int64_t E_out, int *__restrict__ status)
{
    scatter_add_range(old, E, add, na, del, nd, out, E_out, status, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// ---- one timestamp of a delta store as THREE launches (round 3) ---------------------------------------------------------
// What a training step of the dynamic loop needs from the store after an update: the new key arrays, both CSRs (row
// offsets + columns; labels stay lazy), the in-degrees, norm = in_deg^-1/2 and norm gathered per edge of either CSR.
// Issued piecewise that is two scatter launches per orientation, two row-offset searches, two emissions, a degree
// difference, the norm and two gathers -- ~14 launches of 4-20 us at |E| = 250 K, 100 us in all.  Here: (1) both
// orientations' merges in one grid, (2) both row-offset arrays + degrees + norm, (3) both emissions + per-edge norm.
struct StepArgs {
    const uint64_t *old[2], *add[2], *del[2];
    uint64_t *out[2];
    int *ro[2], *col[2];
    const int *ro_old[2];                                  // row offsets of old[side] (nullable): search hints
    float *nc[2];
    int *in_deg;
    float *norm;
    int *status;
    int64_t E, na, nd, E_out;
    int N, nb_old, nb_add;
    int nb_rows;                                           // > 0: the row offsets are derived from ro_old by that many extra blocks per side of the merge launch
};

// The emission of a step (columns + per-edge norm of both CSRs from its merged keys, row offsets and norm) as its own argument
// block: it depends on that step's merge launch only, and the NEXT step's merge does not depend on it -- so a caller that issues
// several steps back to back (a BPTT window's graph updates) may leave it pending and hand it to the next step, whose merge launch
// then carries it as extra blocks (nb > 0): ONE launch per step.
struct EmitArgs {
    const uint64_t *keys[2];
    const int *ro[2];
    int *col[2];
    float *nc[2];
    const float *norm;
    int64_t E;
    int key_order, nb;
};

template <bool KEY_ORDER>
__device__ __forceinline__ void step_emit_blocks(const EmitArgs &a, int64_t block, int64_t nblocks)
{
    const int64_t stride = nblocks * kBlock;
    for (int64_t t = block * kBlock + threadIdx.x; t < 2 * a.E; t += stride) {
        const int side = t >= a.E ? 1 : 0;
        const int64_t i = t - (side ? a.E : 0);
        const uint64_t k = a.keys[side][i];
        const unsigned row = (unsigned)(k >> kStoreBits), c = (unsigned)k;
        const int *ro = a.ro[side];
        const int64_t o = KEY_ORDER ? i : (int64_t)ro[row] + ((int64_t)ro[row + 1] - 1 - i);
        a.col[side][o] = (int)c;
        if (a.nc[side]) a.nc[side][o] = a.norm[c];
    }
}

template <bool KEY_ORDER>
__global__ __launch_bounds__(kBlock) void step_emit_kernel(const EmitArgs a)
{
    step_emit_blocks<KEY_ORDER>(a, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// Row offsets, in-degrees and norm of the NEW set without reading it: ro_new[v] = ro_old[v] + #(added keys below row v) - #(deleted
// keys below row v) -- exact whenever the step is valid (every deletion present, no addition present: anything else raises a status
// bit in the merge blocks).  Two 13-step searches over the L2-resident batches instead of an 18-step one over the merged keys, and
// -- the point -- no dependence on the merge: these blocks ride in ITS launch and the step is two launches, not three.
// Block b of a side takes rows [255 b, 255 b + 255]; the last thread's row is only the next row of its neighbour's degree.
__device__ __forceinline__ void step_rows_from_old(const StepArgs &a, int side, int b)
{
    __shared__ int first[kBlock];
    const int t = (int)threadIdx.x;
    const int64_t v = (int64_t)b * (kBlock - 1) + t;
    int r = 0;
    if (v <= a.N) {
        const uint64_t kv = (uint64_t)(unsigned)v << kStoreBits;
        int64_t al, dl;
        lower_bound2_dev(a.add[side], a.na, kv, a.del[side], a.nd, kv, al, dl);
        r = a.ro_old[side][v] + (int)al - (int)dl;
    }
    first[t] = r;
    __syncthreads();
    if (t == kBlock - 1 || v > a.N) return;
    a.ro[side][v] = r;
    if (side == 0 && v < a.N && (a.in_deg || a.norm)) {
        const int d = first[t + 1] - r;
        if (a.in_deg) a.in_deg[v] = d;
        if (a.norm) a.norm[v] = d > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)d)) : 0.f;     // = degree_norm_kernel
    }
}

__global__ __launch_bounds__(kBlock) void step_merge_kernel(const StepArgs a, const EmitArgs prev)
{
    const int per = a.nb_old + a.nb_add;
    if ((int)blockIdx.x >= 2 * (per + a.nb_rows)) {                                 // (block-uniform, as everything below)
        const int64_t eb = (int64_t)blockIdx.x - 2 * (per + a.nb_rows);            // an earlier step's pending emission
        if (prev.key_order) step_emit_blocks<true>(prev, eb, prev.nb);
        else step_emit_blocks<false>(prev, eb, prev.nb);
        return;
    }
    if ((int)blockIdx.x >= 2 * per) {
        const int rb = (int)blockIdx.x - 2 * per;
        step_rows_from_old(a, rb / a.nb_rows, rb % a.nb_rows);
        return;
    }
    const int side = (int)blockIdx.x / per, r = (int)blockIdx.x - side * per;
    if (r < a.nb_old)
        scatter_old_tile(a.old[side], a.E, a.add[side], a.na, a.del[side], a.nd, a.out[side], a.E_out, a.status, r);
    else
        scatter_add_range(a.old[side], a.E, a.add[side], a.na, a.del[side], a.nd, a.out[side], a.E_out, a.status,
                          r - a.nb_old, a.nb_add, a.ro_old[side], a.N);
}

__global__ __launch_bounds__(kBlock) void step_rows_kernel(const StepArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int side = idx > a.N ? 1 : 0;
    const int v = (int)(idx - (side ? a.N + 1 : 0));
    if (v > a.N) return;
    // first key of row v and (forward side: the degree) of row v + 1, searched together (lower_bound2_dev)
    const uint64_t kv = (uint64_t)(unsigned)v << kStoreBits, kn = (uint64_t)((unsigned)v + 1u) << kStoreBits;
    const bool want_deg = side == 0 && v < a.N && (a.in_deg || a.norm);
    int64_t lo64, nx64;
    lower_bound2_dev(a.out[side], a.E_out, kv, a.out[side], want_deg ? a.E_out : 0, kn, lo64, nx64);
    const int lo = (int)lo64;
    a.ro[side][v] = lo;
    if (want_deg) {
        const int d = (int)nx64 - lo;
        if (a.in_deg) a.in_deg[v] = d;
        if (a.norm) a.norm[v] = d > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)d)) : 0.f;     // = degree_norm_kernel
    }
}

// One pass over the sorted keys of one orientation: column + label into the row-reversed slot.
// Reverse CSR: the label of (src -> dst) is its rank in keys_fwd; row_offset_fwd[dst] (L2 resident) narrows the
// search to that destination's row, i.e. to one or two cache lines of keys_fwd.
template <bool REVERSE, bool KEY_ORDER>
__global__ void emit_rows(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ keys_fwd, int64_t E,
                          const int *__restrict__ row_offset, const int *__restrict__ row_offset_fwd,
                          int *__restrict__ col, int *__restrict__ eids1, int *__restrict__ eids0)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const uint64_t k = keys[i];
        const unsigned row = (unsigned)(k >> kStoreBits), c = (unsigned)k;
        // PCSR layout: rows back to front (the PMA walk, pcsr.cu:784-876); key-order layout: slot = position
        const int64_t o = KEY_ORDER ? i : (int64_t)row_offset[row] + ((int64_t)row_offset[row + 1] - 1 - i);
        int64_t rank = i;                                                    // forward: the label is the position
        if (REVERSE && (eids1 || eids0)) {
            const int64_t lo = row_offset_fwd[c];
            rank = lo + lower_bound_dev(keys_fwd + lo, (int64_t)row_offset_fwd[c + 1] - lo, ((uint64_t)c << kStoreBits) | row);
        }
        if (col) col[o] = (int)c;
        if (eids1) eids1[o] = (int)(rank + 1);
        if (eids0) eids0[o] = (int)rank;
    }
}

struct StoreLayout {
    size_t add_f, add_b, del_f, del_b, sorted, sort_tmp, total, sort_tmp_bytes;
};

StoreLayout update_layout(int64_t n_add, int64_t n_del)
{
    StoreLayout L{};
    const size_t na = (size_t)std::max<int64_t>(n_add, 1), nd = (size_t)std::max<int64_t>(n_del, 1);
    size_t t = 0;
    (void)rocprim::radix_sort_keys(nullptr, t, (uint64_t *)nullptr, (uint64_t *)nullptr, std::max(na, nd), 0, 64);
    L.sort_tmp_bytes = t;
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
    L.add_f = take(na * 8);
    L.add_b = take(na * 8);
    L.del_f = take(nd * 8);
    L.del_b = take(nd * 8);
    L.sorted = take(std::max(na, nd) * 8);         // unsorted staging for one batch at a time
    L.sort_tmp = take(t);
    L.total = off;
    return L;
}

struct EmitLayout {
    size_t key_a, key_b, iota, ro_fwd, sort_tmp, total, sort_tmp_bytes;
};

EmitLayout emit_layout(int32_t N)
{
    EmitLayout L{};
    const size_t n = (size_t)std::max<int32_t>(N, 1);
    size_t t = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, t, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                                         (int *)nullptr, n, 0, 32);
    L.sort_tmp_bytes = t;
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
    L.key_a = take(n * 4);
    L.key_b = take(n * 4);
    L.iota = take(n * 4);
    L.ro_fwd = take((n + 1) * 4);
    L.sort_tmp = take(t);
    L.total = off;
    return L;
}

void launch_merge(const uint64_t *old, int64_t E, const uint64_t *add, int64_t na, const uint64_t *del, int64_t nd,
                  uint64_t *out, int64_t E_out, int32_t *status, hipStream_t stream);

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, 256 * 16)); }

void launch_merge(const uint64_t *old, int64_t E, const uint64_t *add, int64_t na, const uint64_t *del, int64_t nd,
                  uint64_t *out, int64_t E_out, int32_t *status, hipStream_t stream)
{
    if (E > 0) {
        const int64_t tile = (int64_t)kBlock * kMergeItems;
        hipLaunchKernelGGL(scatter_old, dim3((unsigned)((E + tile - 1) / tile)), dim3(kBlock), 0, stream, old, E, add,
                           na, del, nd, out, E_out, status);
    }
    if (na + nd > 0)
        hipLaunchKernelGGL(scatter_add_check_del, dim3(grid_for(na + nd)), dim3(kBlock), 0, stream, old, E, add, na,
                           del, nd, out, E_out, status);
}

}  // namespace
}  // namespace stg

// ------------------------------------------------------------------------------------- host
// Same contract on host arrays (used when the graph lives on the CPU: tests without a GPU).
extern "C" int stg_edgeset_update_host(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                                       const int32_t *add_src, const int32_t *add_dst, int64_t n_add,
                                       const int32_t *del_src, const int32_t *del_dst, int64_t n_del, int32_t N,
                                       uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *status)
{
    using namespace stg;
    if (E < 0 || n_add < 0 || n_del < 0 || N < 0 || !status)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_host: bad argument");
    if (E + n_add - n_del < 0 || E + n_add >= (int64_t(1) << 31))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_host: edge count out of range");
    int st = 0;
    for (int side = 0; side < 2; ++side) {
        std::vector<uint64_t> add((size_t)n_add), del((size_t)n_del);
        auto pack = [&](const int32_t *s, const int32_t *d, int64_t n, std::vector<uint64_t> &out) {
            for (int64_t i = 0; i < n; ++i) {
                const unsigned a = (unsigned)s[i], b = (unsigned)d[i];
                if (a >= (unsigned)N || b >= (unsigned)N) st |= 1;
                out[(size_t)i] = side == 0 ? ((uint64_t)b << 32) | a : ((uint64_t)a << 32) | b;
            }
            std::sort(out.begin(), out.end());
        };
        pack(add_src, add_dst, n_add, add);
        pack(del_src, del_dst, n_del, del);
        const uint64_t *old = side == 0 ? keys_fwd_in : keys_bwd_in;
        uint64_t *out = side == 0 ? keys_fwd_out : keys_bwd_out;
        if (std::adjacent_find(add.begin(), add.end()) != add.end()) st |= 2;
        if (std::adjacent_find(del.begin(), del.end()) != del.end()) st |= 4;
        std::vector<uint64_t> kept;
        kept.reserve((size_t)E);
        size_t q = 0, found = 0;
        for (int64_t i = 0; i < E; ++i) {
            while (q < del.size() && del[q] < old[i]) ++q;
            if (q < del.size() && del[q] == old[i]) { ++found; continue; }
            kept.push_back(old[i]);
        }
        if (found != del.size()) st |= 4;
        for (uint64_t k : add) {
            if (std::binary_search(old, old + E, k)) st |= 2;
            if (std::binary_search(del.begin(), del.end(), k)) st |= 8;
        }
        if (st) continue;                                      // contract violated: leave the outputs untouched
        std::merge(kept.begin(), kept.end(), add.begin(), add.end(), out);
    }
    *status = st;
    return 0;
}

extern "C" int stg_edgeset_emit_csr_host(const uint64_t *keys_fwd, const uint64_t *keys_bwd, int64_t E, int32_t N,
                                         int flags, int32_t *row_offset, int32_t *column_indices, int32_t *eids1,
                                         int32_t *eids0, int32_t *node_ids, int32_t *degrees)
{
    using namespace stg;
    if (E < 0 || N < 0 || !row_offset || (E > 0 && (!keys_fwd || !keys_bwd)) || (!node_ids != !degrees) ||
        (flags & ~(STG_EMIT_REVERSE | STG_EMIT_KEY_ORDER)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_emit_csr_host: bad argument");
    const bool reverse = flags & STG_EMIT_REVERSE, key_order = flags & STG_EMIT_KEY_ORDER;
    const uint64_t *keys = reverse ? keys_bwd : keys_fwd;
    int64_t p = 0;
    for (int32_t v = 0; v <= N; ++v) {
        while (p < E && (int64_t)(keys[p] >> 32) < (int64_t)v) ++p;
        row_offset[v] = (int32_t)p;
    }
    for (int64_t i = 0; i < E; ++i) {
        const uint64_t k = keys[i];
        const unsigned row = (unsigned)(k >> 32), c = (unsigned)k;
        if (row >= (unsigned)N) return fail(STG_ERR_VERTEX_RANGE, "stg_edgeset_emit_csr_host: vertex id out of range");
        const int64_t o = key_order ? i : (int64_t)row_offset[row] + ((int64_t)row_offset[row + 1] - 1 - i);
        int64_t rank = i;
        if (reverse && (eids1 || eids0)) rank = std::lower_bound(keys_fwd, keys_fwd + E, ((uint64_t)c << 32) | row) - keys_fwd;
        if (column_indices) column_indices[o] = (int32_t)c;
        if (eids1) eids1[o] = (int32_t)(rank + 1);
        if (eids0) eids0[o] = (int32_t)rank;
    }
    if (!degrees) return 0;
    for (int32_t v = 0; v < N; ++v) degrees[v] = row_offset[v + 1] - row_offset[v];
    if (N > 0) {
        std::iota(node_ids, node_ids + N, 0);
        std::stable_sort(node_ids, node_ids + N, [degrees](int32_t l, int32_t r) { return degrees[l] > degrees[r]; });
    }
    return 0;
}

// ----------------------------------------------------------------------------------- device
extern "C" size_t stg_edgeset_update_workspace_bytes(int64_t n_add, int64_t n_del)
{
    if (n_add < 0 || n_del < 0) return 0;
    return stg::update_layout(n_add, n_del).total;
}

extern "C" int stg_edgeset_update_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                                         const int32_t *add_src, const int32_t *add_dst, int64_t n_add,
                                         const int32_t *del_src, const int32_t *del_dst, int64_t n_del, int32_t N,
                                         uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *status,
                                         void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E < 0 || n_add < 0 || n_del < 0 || N < 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_device: negative size");
    const int64_t E_out = E + n_add - n_del;
    if (E_out < 0 || E + n_add >= (int64_t(1) << 31))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_device: edge count %lld out of range", (long long)E_out);
    if (!status || !workspace || (E > 0 && (!keys_fwd_in || !keys_bwd_in)) ||
        (E_out > 0 && (!keys_fwd_out || !keys_bwd_out)) || (n_add > 0 && (!add_src || !add_dst)) ||
        (n_del > 0 && (!del_src || !del_dst)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_device: NULL pointer argument");
    if (keys_fwd_out == keys_fwd_in || keys_bwd_out == keys_bwd_in)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_update_device: the update is out of place");
    const StoreLayout L = update_layout(n_add, n_del);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_edgeset_update_device: workspace %zu < required %zu", workspace_bytes, L.total);
    char *ws = static_cast<char *>(workspace);
    uint64_t *batch[4] = {reinterpret_cast<uint64_t *>(ws + L.add_f), reinterpret_cast<uint64_t *>(ws + L.add_b),
                          reinterpret_cast<uint64_t *>(ws + L.del_f), reinterpret_cast<uint64_t *>(ws + L.del_b)};
    auto *staging = reinterpret_cast<uint64_t *>(ws + L.sorted);
    void *sort_tmp = ws + L.sort_tmp;
    const unsigned end_bit = (unsigned)(kStoreBits + key_bits_for(N));

    if (const int rc = zero_async(status, sizeof(int32_t), stream)) return rc;
    hipError_t e = hipSuccess;

    // pack + sort the two batches in both orientations (batches are small next to E)
    for (int which = 0; which < 2; ++which) {
        const int64_t n = which == 0 ? n_add : n_del;
        if (n == 0) continue;
        const int32_t *s = which == 0 ? add_src : del_src, *d = which == 0 ? add_dst : del_dst;
        uint64_t *kf = batch[2 * which], *kb = batch[2 * which + 1];
        // pack forward keys into the staging buffer and backward keys into kb; sort staging -> kf; then kb -> staging -> kb
        hipLaunchKernelGGL(pack_batch, dim3(grid_for(n)), dim3(kBlock), 0, stream, s, d, n, N, staging, kb, status);
        size_t tmp = L.sort_tmp_bytes;
        e = rocprim::radix_sort_keys(sort_tmp, tmp, staging, kf, (size_t)n, 0, end_bit, stream);
        if (e != hipSuccess) return fail((int)e, "stg_edgeset_update_device: sort: %s", hipGetErrorString(e));
        tmp = L.sort_tmp_bytes;
        e = rocprim::radix_sort_keys(sort_tmp, tmp, kb, staging, (size_t)n, 0, end_bit, stream);
        if (e != hipSuccess) return fail((int)e, "stg_edgeset_update_device: sort: %s", hipGetErrorString(e));
        if (const int rc = copy_async(kb, staging, sizeof(uint64_t) * (size_t)n, stream)) return rc;
    }
    for (int side = 0; side < 2; ++side) {
        const uint64_t *old = side == 0 ? keys_fwd_in : keys_bwd_in;
        uint64_t *out = side == 0 ? keys_fwd_out : keys_bwd_out;
        const uint64_t *add = batch[side], *del = batch[2 + side];
        launch_merge(old, E, add, n_add, del, n_del, out, E_out, status, stream);
    }
    return check_launch("stg_edgeset_update_device");
}

extern "C" int stg_edgeset_merge_device(const uint64_t *keys_in, int64_t E, const uint64_t *add_sorted, int64_t n_add,
                                        const uint64_t *del_sorted, int64_t n_del, uint64_t *keys_out,
                                        int32_t *status, void *stream_)
{
    using namespace stg;
    if (E < 0 || n_add < 0 || n_del < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_merge_device: negative size");
    const int64_t E_out = E + n_add - n_del;
    if (E_out < 0 || E + n_add >= (int64_t(1) << 31))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_merge_device: edge count %lld out of range", (long long)E_out);
    if (!status || (E > 0 && !keys_in) || (E_out > 0 && !keys_out) || (n_add > 0 && !add_sorted) ||
        (n_del > 0 && !del_sorted))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_merge_device: NULL pointer argument");
    if (keys_out == keys_in) return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_merge_device: the merge is out of place");
    launch_merge(keys_in, E, add_sorted, n_add, del_sorted, n_del, keys_out, E_out, status,
                 static_cast<hipStream_t>(stream_));
    return check_launch("stg_edgeset_merge_device");
}

namespace stg {
namespace {

EmitArgs emit_args(const stg_store_emission &e)
{
    EmitArgs p{};
    p.keys[0] = e.keys_fwd; p.keys[1] = e.keys_bwd; p.ro[0] = e.fwd_row_offset; p.ro[1] = e.bwd_row_offset;
    p.col[0] = e.fwd_column_indices; p.col[1] = e.bwd_column_indices; p.nc[0] = e.norm_col_fwd; p.nc[1] = e.norm_col_bwd;
    p.norm = e.norm; p.E = e.E; p.key_order = (e.flags & STG_EMIT_KEY_ORDER) ? 1 : 0;
    p.nb = e.E > 0 ? grid_for(2 * e.E) : 0;
    return p;
}

int emission_ok(const stg_store_emission &e, const char *who)
{
    if (e.E < 0 || e.E >= (int64_t(1) << 30) || (e.flags & ~STG_EMIT_KEY_ORDER))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad pending emission (E=%lld, flags=%d)", who, (long long)e.E, e.flags);
    if (e.E > 0 && (!e.keys_fwd || !e.keys_bwd || !e.fwd_row_offset || !e.bwd_row_offset || !e.fwd_column_indices ||
                    !e.bwd_column_indices || ((e.norm_col_fwd || e.norm_col_bwd) && !e.norm)))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer in a pending emission", who);
    return 0;
}

void launch_emission(const EmitArgs &p, hipStream_t stream)
{
    if (p.nb == 0) return;
    if (p.key_order) hipLaunchKernelGGL((step_emit_kernel<true>), dim3(p.nb), dim3(kBlock), 0, stream, p);
    else hipLaunchKernelGGL((step_emit_kernel<false>), dim3(p.nb), dim3(kBlock), 0, stream, p);
}

// carry (nullable): an earlier step's emission, issued with this step's merge; pending_out (nullable): NULL = this step's
// emission is launched here, else it is described there and left to the caller.
int edgeset_step_run(const char *who, const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E, const uint64_t *add_fwd,
                     const uint64_t *add_bwd, int64_t n_add, const uint64_t *del_fwd, const uint64_t *del_bwd, int64_t n_del, int32_t N,
                     int flags, uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                     int32_t *bwd_row_offset, int32_t *bwd_column_indices, int32_t *in_degrees, float *norm, float *norm_col_fwd,
                     float *norm_col_bwd, const int32_t *fwd_row_offset_in, const int32_t *bwd_row_offset_in,
                     const stg_store_emission *carry, stg_store_emission *pending_out, int32_t *status, hipStream_t stream)
{
    if (E < 0 || n_add < 0 || n_del < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "%s: negative size", who);
    const int64_t E_out = E + n_add - n_del;
    if (E_out < 0 || E + n_add >= (int64_t(1) << 30))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: edge count %lld out of range", who, (long long)E_out);
    if (flags & ~STG_EMIT_KEY_ORDER) return fail(STG_ERR_INVALID_ARGUMENT, "%s: unknown flag", who);
    if (!status || !fwd_row_offset || !bwd_row_offset || (E > 0 && (!keys_fwd_in || !keys_bwd_in)) ||
        (E_out > 0 && (!keys_fwd_out || !keys_bwd_out || !fwd_column_indices || !bwd_column_indices)) ||
        (n_add > 0 && (!add_fwd || !add_bwd)) || (n_del > 0 && (!del_fwd || !del_bwd)) ||
        ((norm_col_fwd || norm_col_bwd) && !norm))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", who);
    if (keys_fwd_out == keys_fwd_in || keys_bwd_out == keys_bwd_in)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: the update is out of place", who);
    EmitArgs prev{};
    if (carry) {
        if (const int rc = emission_ok(*carry, who)) return rc;
        prev = emit_args(*carry);
    }
    StepArgs a{};
    a.old[0] = keys_fwd_in; a.old[1] = keys_bwd_in; a.add[0] = add_fwd; a.add[1] = add_bwd; a.del[0] = del_fwd; a.del[1] = del_bwd;
    a.out[0] = keys_fwd_out; a.out[1] = keys_bwd_out; a.ro[0] = fwd_row_offset; a.ro[1] = bwd_row_offset;
    a.col[0] = fwd_column_indices; a.col[1] = bwd_column_indices; a.nc[0] = norm_col_fwd; a.nc[1] = norm_col_bwd;
    a.in_deg = in_degrees; a.norm = norm; a.status = status;
    const bool hints = fwd_row_offset_in && bwd_row_offset_in;
    a.ro_old[0] = hints ? fwd_row_offset_in : nullptr; a.ro_old[1] = hints ? bwd_row_offset_in : nullptr;
    a.E = E; a.na = n_add; a.nd = n_del; a.E_out = E_out; a.N = N;
    const int64_t tile = (int64_t)kBlock * kMergeItems;
    a.nb_old = (int)((E + tile - 1) / tile);
    a.nb_add = n_add + n_del > 0 ? grid_for(n_add + n_del) : 0;
    // with the old set's row offsets at hand the new ones do not wait for the merge: their blocks join its launch
    const bool rows_in_merge = hints && a.nb_old + a.nb_add > 0 && tuning().store_rows != 1 && fwd_row_offset_in != fwd_row_offset &&
                               bwd_row_offset_in != bwd_row_offset;
    a.nb_rows = rows_in_merge ? (int)(((int64_t)N + 1 + kBlock - 2) / (kBlock - 1)) : 0;
    if (a.nb_old + a.nb_add > 0)
        hipLaunchKernelGGL(step_merge_kernel, dim3(2u * (unsigned)(a.nb_old + a.nb_add + a.nb_rows) + (unsigned)prev.nb), dim3(kBlock), 0,
                           stream, a, prev);
    else
        launch_emission(prev, stream);
    if (!rows_in_merge)
        hipLaunchKernelGGL(step_rows_kernel, dim3((unsigned)((2 * ((int64_t)N + 1) + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, a);
    stg_store_emission mine{};
    mine.keys_fwd = keys_fwd_out; mine.keys_bwd = keys_bwd_out; mine.E = E_out; mine.fwd_row_offset = fwd_row_offset;
    mine.bwd_row_offset = bwd_row_offset; mine.fwd_column_indices = fwd_column_indices; mine.bwd_column_indices = bwd_column_indices;
    mine.norm = norm; mine.norm_col_fwd = norm_col_fwd; mine.norm_col_bwd = norm_col_bwd; mine.flags = flags;
    if (pending_out) *pending_out = mine;
    else launch_emission(emit_args(mine), stream);
    return check_launch(who);
}

}  // namespace
}  // namespace stg

extern "C" int stg_edgeset_step_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                                       const uint64_t *add_fwd, const uint64_t *add_bwd, int64_t n_add,
                                       const uint64_t *del_fwd, const uint64_t *del_bwd, int64_t n_del, int32_t N, int flags,
                                       uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *fwd_row_offset,
                                       int32_t *fwd_column_indices, int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                                       int32_t *in_degrees, float *norm, float *norm_col_fwd, float *norm_col_bwd,
                                       const int32_t *fwd_row_offset_in, const int32_t *bwd_row_offset_in, int32_t *status,
                                       void *stream_)
{
    return stg::edgeset_step_run("stg_edgeset_step_device", keys_fwd_in, keys_bwd_in, E, add_fwd, add_bwd, n_add, del_fwd, del_bwd, n_del,
                                 N, flags, keys_fwd_out, keys_bwd_out, fwd_row_offset, fwd_column_indices, bwd_row_offset,
                                 bwd_column_indices, in_degrees, norm, norm_col_fwd, norm_col_bwd, fwd_row_offset_in, bwd_row_offset_in,
                                 nullptr, nullptr, status, static_cast<hipStream_t>(stream_));
}

extern "C" int stg_edgeset_step_deferred_device(const uint64_t *keys_fwd_in, const uint64_t *keys_bwd_in, int64_t E,
                                                const uint64_t *add_fwd, const uint64_t *add_bwd, int64_t n_add,
                                                const uint64_t *del_fwd, const uint64_t *del_bwd, int64_t n_del, int32_t N, int flags,
                                                uint64_t *keys_fwd_out, uint64_t *keys_bwd_out, int32_t *fwd_row_offset,
                                                int32_t *fwd_column_indices, int32_t *bwd_row_offset, int32_t *bwd_column_indices,
                                                int32_t *in_degrees, float *norm, float *norm_col_fwd, float *norm_col_bwd,
                                                const int32_t *fwd_row_offset_in, const int32_t *bwd_row_offset_in,
                                                const stg_store_emission *carry, stg_store_emission *pending_out, int32_t *status,
                                                void *stream_)
{
    if (!pending_out) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_step_deferred_device: NULL pending_out");
    return stg::edgeset_step_run("stg_edgeset_step_deferred_device", keys_fwd_in, keys_bwd_in, E, add_fwd, add_bwd, n_add, del_fwd, del_bwd,
                                 n_del, N, flags, keys_fwd_out, keys_bwd_out, fwd_row_offset, fwd_column_indices, bwd_row_offset,
                                 bwd_column_indices, in_degrees, norm, norm_col_fwd, norm_col_bwd, fwd_row_offset_in, bwd_row_offset_in,
                                 carry, pending_out, status, static_cast<hipStream_t>(stream_));
}

extern "C" int stg_edgeset_emit_pending_device(const stg_store_emission *pending, void *stream_)
{
    if (!pending) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_emit_pending_device: NULL argument");
    if (const int rc = stg::emission_ok(*pending, "stg_edgeset_emit_pending_device")) return rc;
    stg::launch_emission(stg::emit_args(*pending), static_cast<hipStream_t>(stream_));
    return stg::check_launch("stg_edgeset_emit_pending_device");
}

extern "C" size_t stg_edgeset_emit_csr_workspace_bytes(int32_t N)
{
    if (N < 0) return 0;
    return stg::emit_layout(N).total;
}

extern "C" int stg_edgeset_emit_csr_device(const uint64_t *keys_fwd, const uint64_t *keys_bwd, int64_t E, int32_t N,
                                           int flags, int32_t *row_offset, int32_t *column_indices, int32_t *eids1,
                                           int32_t *eids0, int32_t *node_ids, int32_t *degrees, void *workspace,
                                           size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_emit_csr_device: negative size");
    if (E >= (int64_t(1) << 31)) return fail(STG_ERR_UNSUPPORTED, "stg_edgeset_emit_csr_device: E does not fit int32");
    if (flags & ~(STG_EMIT_REVERSE | STG_EMIT_KEY_ORDER))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_emit_csr_device: unknown flag");
    const bool reverse = flags & STG_EMIT_REVERSE, key_order = flags & STG_EMIT_KEY_ORDER;
    if (!row_offset || !workspace || (E > 0 && (!keys_fwd || !keys_bwd)) || (!node_ids != !degrees))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_edgeset_emit_csr_device: NULL pointer argument");
    const EmitLayout L = emit_layout(N);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_edgeset_emit_csr_device: workspace %zu < required %zu", workspace_bytes, L.total);
    char *ws = static_cast<char *>(workspace);
    const uint64_t *keys = reverse ? keys_bwd : keys_fwd;
    hipLaunchKernelGGL(row_offsets_by_search, dim3((N + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, keys, E,
                       kStoreBits, N, row_offset);
    if (E > 0 && (column_indices || eids1 || eids0)) {
        const int *ro_fwd = nullptr;
        int *e1 = eids1, *e0 = eids0;
        if (reverse && (eids1 || eids0)) {
            auto *ro = reinterpret_cast<int *>(ws + L.ro_fwd);
            hipLaunchKernelGGL(row_offsets_by_search, dim3((N + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                               keys_fwd, E, kStoreBits, N, ro);
            ro_fwd = ro;
        }
        const dim3 grid(grid_for(E)), block(kBlock);
        if (reverse && key_order)
            hipLaunchKernelGGL((emit_rows<true, true>), grid, block, 0, stream, keys, keys_fwd, E, row_offset, ro_fwd,
                               column_indices, e1, e0);
        else if (reverse)
            hipLaunchKernelGGL((emit_rows<true, false>), grid, block, 0, stream, keys, keys_fwd, E, row_offset, ro_fwd,
                               column_indices, e1, e0);
        else if (key_order)
            hipLaunchKernelGGL((emit_rows<false, true>), grid, block, 0, stream, keys, keys_fwd, E, row_offset, ro_fwd,
                               column_indices, e1, e0);
        else
            hipLaunchKernelGGL((emit_rows<false, false>), grid, block, 0, stream, keys, keys_fwd, E, row_offset, ro_fwd,
                               column_indices, e1, e0);
    }
    if (N > 0 && node_ids) {
        auto *key_a = reinterpret_cast<unsigned *>(ws + L.key_a);
        auto *key_b = reinterpret_cast<unsigned *>(ws + L.key_b);
        auto *iota = reinterpret_cast<int *>(ws + L.iota);
        hipLaunchKernelGGL(degrees_and_iota, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, row_offset, N,
                           degrees, key_a, iota);
        size_t tmp = L.sort_tmp_bytes;
        const hipError_t e = rocprim::radix_sort_pairs_desc(ws + L.sort_tmp, tmp, key_a, key_b, iota, node_ids,
                                                            (size_t)N, 0, 32, stream);
        if (e != hipSuccess) return fail((int)e, "stg_edgeset_emit_csr_device: node_ids sort: %s", hipGetErrorString(e));
    }
    return check_launch("stg_edgeset_emit_csr_device");
}
