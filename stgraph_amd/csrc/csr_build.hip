// CSR construction for the Seastar graph surface (stgraph_hip.h, "CSR" sections).
//
//  * stg_csr_ctor_host      : loop-for-loop counterpart of the reference's pybind
//                             CSR constructor (host arrays in, host arrays out).
//  * stg_graph_build_host   : whole StaticGraph build on the host (sort + both CSRs).
//  * stg_graph_build_device : the MI355X path -- everything on the GPU, stream
//                             ordered: two stable LSD radix sorts (rocPRIM) on
//                             packed (row,col) keys, row offsets by binary search
//                             over the sorted keys (no atomics, deterministic),
//                             degree-sorted node_ids by a third stable sort.
//
// Ordering contract (bit-exact with the reference's Python+C++ pipeline):
//   forward  = stable sort by (dst, src), eid = rank         static_graph.py:65-72
//   backward = sort of (src, dst, eid)                       static_graph.py:75-78
//   node_ids = rows by non-increasing degree (csr.cu:142-154; ties are
//              unspecified there, ascending id here).
#include "stg_common.hpp"
#include "csr_kernels.hpp"

#include <cstring>   // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <numeric>
#include <vector>

namespace stg {
namespace {

void node_ids_by_degree_host(const int32_t *deg, int32_t N, int32_t *node_ids)
{
    std::iota(node_ids, node_ids + N, 0);
    std::stable_sort(node_ids, node_ids + N, [deg](int32_t l, int32_t r) { return deg[l] > deg[r]; });
}

// LSD radix sort of (key, value) pairs on the host, stable, 16 bits per pass.
void radix_sort_pairs_host(std::vector<uint64_t> &keys, std::vector<int64_t> &vals, int key_bits, int begin_bit = 0)
{
    const size_t n = keys.size();
    std::vector<uint64_t> k2(n);
    std::vector<int64_t> v2(n);
    for (int shift = begin_bit; shift < key_bits; shift += 16) {
        std::vector<size_t> hist(65536 + 1, 0);
        for (size_t i = 0; i < n; ++i) ++hist[((keys[i] >> shift) & 0xFFFF) + 1];
        for (size_t b = 0; b < 65536; ++b) hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; ++i) {
            const size_t p = hist[(keys[i] >> shift) & 0xFFFF]++;
            k2[p] = keys[i];
            v2[p] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

// ------------------------------------------------------------------ device kernels
__global__ void make_fwd_keys(const int *__restrict__ src, const int *__restrict__ dst, int64_t E,
                              int N, int bits, uint64_t *__restrict__ keys,
                              int64_t *__restrict__ pos, int *__restrict__ status)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int s = src[i], d = dst[i];
        if ((unsigned)s >= (unsigned)N || (unsigned)d >= (unsigned)N) *status = STG_ERR_VERTEX_RANGE;
        keys[i] = ((uint64_t)(unsigned)d << bits) | (uint64_t)(unsigned)s;
        pos[i] = i;
    }
}

// From the forward-sorted keys: forward columns / identity eids, and the keys of the backward sort.
__global__ void split_fwd_make_bwd(const uint64_t *__restrict__ fkeys, int64_t E, int bits,
                                   int *__restrict__ fwd_col, int *__restrict__ fwd_eid,
                                   uint64_t *__restrict__ bkeys, int *__restrict__ bvals)
{
    const uint64_t mask = (uint64_t(1) << bits) - 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += stride) {
        const uint64_t k = fkeys[j];
        const uint64_t s = k & mask, d = k >> bits;
        fwd_col[j] = (int)s;
        fwd_eid[j] = (int)j;
        bkeys[j] = (s << bits) | d;
        bvals[j] = (int)j;
    }
}

__global__ void split_bwd(const uint64_t *__restrict__ bkeys, int64_t E, int bits,
                          int *__restrict__ bwd_col)
{
    const uint64_t mask = (uint64_t(1) << bits) - 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += stride)
        bwd_col[j] = (int)(bkeys[j] & mask);
}


struct DeviceLayout {
    size_t keys_a, keys_b, pos_b, vals_a, vals_b, deg_key_a, deg_key_b, iota, deg_tmp, sort_tmp, total;
    size_t sort_tmp_bytes;
};

DeviceLayout device_layout(int64_t E, int32_t N)
{
    DeviceLayout L{};
    const size_t e = (size_t)std::max<int64_t>(E, 1), n = (size_t)std::max<int32_t>(N, 1);
    size_t t1 = 0, t2 = 0, t3 = 0;
    const unsigned end_bit = (unsigned)(2 * key_bits_for(N));
    (void)rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (int64_t *)nullptr, (int64_t *)nullptr, e, 0, end_bit);
    (void)rocprim::radix_sort_pairs(nullptr, t2, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (int *)nullptr, (int *)nullptr, e, 0, end_bit);
    (void)rocprim::radix_sort_pairs_desc(nullptr, t3, (unsigned *)nullptr, (unsigned *)nullptr,
                                         (int *)nullptr, (int *)nullptr, n, 0, 32);
    L.sort_tmp_bytes = std::max(t1, std::max(t2, t3));
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
    L.keys_a = take(e * 8);
    L.keys_b = take(e * 8);
    L.pos_b = take(e * 8);       // int64 positions, input side of sort 1 (output goes to perm_fwd)
    L.vals_a = take(e * 4);
    L.vals_b = take(e * 4);
    L.deg_key_a = take(n * 4);
    L.deg_key_b = take(n * 4);
    L.iota = take(n * 4);
    L.deg_tmp = take(n * 4);
    L.sort_tmp = take(L.sort_tmp_bytes);
    L.total = off;
    return L;
}

}  // namespace
}  // namespace stg

// ------------------------------------------------------------------------------ host
extern "C" int stg_csr_ctor_host(const int32_t *a, const int32_t *b, const int32_t *eid,
                                 const float *edge_weight, int64_t E, int32_t N,
                                 int is_edge_reverse, int32_t *row_offset, int32_t *column_indices,
                                 int32_t *eids, int32_t *node_ids, int32_t *in_degrees,
                                 int32_t *out_degrees, float *weighted_out_degrees)
{
    using namespace stg;
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_csr_ctor_host: negative size");
    if ((E > 0 && (!a || !b || !eid || !column_indices || !eids)) || !row_offset ||
        (N > 0 && (!node_ids || !in_degrees || !out_degrees || !weighted_out_degrees)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_csr_ctor_host: NULL pointer argument");

    std::fill(in_degrees, in_degrees + N, 0);
    std::fill(out_degrees, out_degrees + N, 0);
    std::fill(weighted_out_degrees, weighted_out_degrees + N, 0.f);

    // Rows arrive grouped (the caller sorted by row); count then prefix-sum.  For grouped input
    // this yields exactly the array the reference derives by tracking row changes and
    // back-filling the gaps (csr.cu:96-140).
    for (int64_t i = 0; i < E; ++i) {
        const int32_t row = is_edge_reverse ? b[i] : a[i];
        const int32_t col = is_edge_reverse ? a[i] : b[i];
        if ((uint32_t)row >= (uint32_t)N || (uint32_t)col >= (uint32_t)N)
            return fail(STG_ERR_VERTEX_RANGE, "stg_csr_ctor_host: edge %lld = (%d,%d) outside [0,%d)",
                        (long long)i, a[i], b[i], N);
        if (i > 0) {
            const int32_t prev = is_edge_reverse ? b[i - 1] : a[i - 1];
            if (row < prev)
                return fail(STG_ERR_INVALID_ARGUMENT,
                            "stg_csr_ctor_host: edge list is not grouped by ascending row at %lld", (long long)i);
        }
        column_indices[i] = col;
        eids[i] = eid[i];
        out_degrees[row] += 1;
        in_degrees[col] += 1;
        weighted_out_degrees[row] += edge_weight ? edge_weight[eid[i]] : 1.0f;
    }
    row_offset[0] = 0;
    for (int32_t v = 0; v < N; ++v) row_offset[v + 1] = row_offset[v] + out_degrees[v];
    node_ids_by_degree_host(out_degrees, N, node_ids);
    return 0;
}

extern "C" int stg_graph_build_host(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                    int64_t *perm_fwd, int32_t *fwd_row_offset,
                                    int32_t *fwd_column_indices, int32_t *fwd_eids,
                                    int32_t *fwd_node_ids, int32_t *bwd_row_offset,
                                    int32_t *bwd_column_indices, int32_t *bwd_eids,
                                    int32_t *bwd_node_ids, int32_t *in_degrees,
                                    int32_t *out_degrees)
{
    using namespace stg;
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_host: negative size");
    if (E >= (int64_t(1) << 31))
        return fail(STG_ERR_UNSUPPORTED, "stg_graph_build_host: E=%lld does not fit int32 edge ids", (long long)E);
    if ((E > 0 && (!src || !dst || !perm_fwd || !fwd_column_indices || !fwd_eids ||
                   !bwd_column_indices || !bwd_eids)) ||
        !fwd_row_offset || !bwd_row_offset ||
        (N > 0 && (!fwd_node_ids || !bwd_node_ids || !in_degrees || !out_degrees)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_host: NULL pointer argument");

    const int bits = key_bits_for(N);
    std::vector<uint64_t> keys((size_t)E);
    std::vector<int64_t> vals((size_t)E);
    for (int64_t i = 0; i < E; ++i) {
        if ((uint32_t)src[i] >= (uint32_t)N || (uint32_t)dst[i] >= (uint32_t)N)
            return fail(STG_ERR_VERTEX_RANGE, "stg_graph_build_host: edge %lld = (%d,%d) outside [0,%d)",
                        (long long)i, src[i], dst[i], N);
        keys[i] = ((uint64_t)(uint32_t)dst[i] << bits) | (uint32_t)src[i];
        vals[i] = i;
    }
    radix_sort_pairs_host(keys, vals, 2 * bits);
    const uint64_t mask = (uint64_t(1) << bits) - 1;
    std::fill(in_degrees, in_degrees + N, 0);
    std::fill(out_degrees, out_degrees + N, 0);
    std::vector<uint64_t> bkeys((size_t)E);
    std::vector<int64_t> bvals((size_t)E);
    for (int64_t j = 0; j < E; ++j) {
        const uint64_t s = keys[j] & mask, d = keys[j] >> bits;
        perm_fwd[j] = vals[j];
        fwd_column_indices[j] = (int32_t)s;
        fwd_eids[j] = (int32_t)j;
        in_degrees[d] += 1;
        out_degrees[s] += 1;
        bkeys[j] = (s << bits) | d;
        bvals[j] = j;
    }
    // the input is in forward order, i.e. already sorted by (dst, eid): a STABLE sort on the src bits alone
    // yields (src, dst, eid) -- half the passes
    radix_sort_pairs_host(bkeys, bvals, 2 * bits, bits);
    for (int64_t j = 0; j < E; ++j) {
        bwd_column_indices[j] = (int32_t)(bkeys[j] & mask);
        bwd_eids[j] = (int32_t)bvals[j];
    }
    fwd_row_offset[0] = 0;
    bwd_row_offset[0] = 0;
    for (int32_t v = 0; v < N; ++v) {
        fwd_row_offset[v + 1] = fwd_row_offset[v] + in_degrees[v];
        bwd_row_offset[v + 1] = bwd_row_offset[v] + out_degrees[v];
    }
    node_ids_by_degree_host(in_degrees, N, fwd_node_ids);    // forward CSR rows = dst
    node_ids_by_degree_host(out_degrees, N, bwd_node_ids);   // backward CSR rows = src
    return 0;
}

// ---------------------------------------------------------------------------- device
extern "C" size_t stg_graph_build_device_workspace_bytes(int64_t E, int32_t N)
{
    if (E < 0 || N < 0) return 0;
    return stg::device_layout(E, N).total;
}

extern "C" int stg_graph_build_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                      int64_t *perm_fwd, int32_t *fwd_row_offset,
                                      int32_t *fwd_column_indices, int32_t *fwd_eids,
                                      int32_t *fwd_node_ids, int32_t *bwd_row_offset,
                                      int32_t *bwd_column_indices, int32_t *bwd_eids,
                                      int32_t *bwd_node_ids, int32_t *in_degrees,
                                      int32_t *out_degrees, int32_t *status, void *workspace,
                                      size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_device: negative size");
    if (E >= (int64_t(1) << 31))
        return fail(STG_ERR_UNSUPPORTED, "stg_graph_build_device: E=%lld does not fit int32 edge ids", (long long)E);
    if ((E > 0 && (!src || !dst || !perm_fwd || !fwd_column_indices || !fwd_eids ||
                   !bwd_column_indices || !bwd_eids)) ||
        !fwd_row_offset || !bwd_row_offset || !status || !workspace ||
        (N > 0 && (!fwd_node_ids || !bwd_node_ids || !in_degrees || !out_degrees)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_device: NULL pointer argument");
    const DeviceLayout L = device_layout(E, N);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_graph_build_device: workspace %zu < required %zu", workspace_bytes, L.total);

    char *ws = static_cast<char *>(workspace);
    auto *keys_a = reinterpret_cast<uint64_t *>(ws + L.keys_a);
    auto *keys_b = reinterpret_cast<uint64_t *>(ws + L.keys_b);
    auto *pos_b = reinterpret_cast<int64_t *>(ws + L.pos_b);
    auto *vals_a = reinterpret_cast<int *>(ws + L.vals_a);
    auto *deg_key_a = reinterpret_cast<unsigned *>(ws + L.deg_key_a);
    auto *deg_key_b = reinterpret_cast<unsigned *>(ws + L.deg_key_b);
    auto *iota = reinterpret_cast<int *>(ws + L.iota);
    void *sort_tmp = ws + L.sort_tmp;
    size_t sort_tmp_bytes = L.sort_tmp_bytes;

    const int bits = key_bits_for(N);
    if (const int rc = zero_async(status, sizeof(int32_t), stream)) return rc;
    hipError_t e = hipSuccess;

    const int threads = kBlock;
    const int eblocks = (int)std::min<int64_t>((E + threads - 1) / threads, 256 * 16);
    const int nblocks = (N + 1 + threads - 1) / threads;

    if (E > 0) {
        // forward: stable sort by (dst, src); the carried value is the caller position
        hipLaunchKernelGGL(make_fwd_keys, dim3(eblocks), dim3(threads), 0, stream, src, dst, E, N, bits,
                           keys_a, pos_b, status);
        e = rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys_a, keys_b, pos_b, perm_fwd,
                                      (size_t)E, 0, (unsigned)(2 * bits), stream);
        if (e != hipSuccess) return fail((int)e, "stg_graph_build_device: forward sort: %s", hipGetErrorString(e));
        // keys_b = forward-sorted keys.  Emit forward arrays + backward keys (into keys_a).
        hipLaunchKernelGGL(split_fwd_make_bwd, dim3(eblocks), dim3(threads), 0, stream, keys_b, E, bits,
                           fwd_column_indices, fwd_eids, keys_a, vals_a);
    }
    hipLaunchKernelGGL(row_offsets_by_search, dim3(nblocks), dim3(threads), 0, stream, keys_b, E, bits,
                       N, fwd_row_offset);
    if (E > 0) {
        // backward: the keys arrive in forward order, i.e. sorted by (dst, eid), so a STABLE sort on the src
        // bits alone gives (src, dst, eid) lexicographic -- 3 radix passes instead of 5 at |V| = 1M
        sort_tmp_bytes = L.sort_tmp_bytes;
        e = rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys_a, keys_b, vals_a, bwd_eids,
                                      (size_t)E, (unsigned)bits, (unsigned)(2 * bits), stream);
        if (e != hipSuccess) return fail((int)e, "stg_graph_build_device: backward sort: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(split_bwd, dim3(eblocks), dim3(threads), 0, stream, keys_b, E, bits,
                           bwd_column_indices);
    }
    hipLaunchKernelGGL(row_offsets_by_search, dim3(nblocks), dim3(threads), 0, stream, keys_b, E, bits,
                       N, bwd_row_offset);

    if (N > 0) {
        const int vblocks = (N + threads - 1) / threads;
        // graph in-degree = forward row length; out-degree = backward row length
        hipLaunchKernelGGL(degrees_and_iota, dim3(vblocks), dim3(threads), 0, stream, fwd_row_offset, N,
                           in_degrees, deg_key_a, iota);
        sort_tmp_bytes = L.sort_tmp_bytes;
        e = rocprim::radix_sort_pairs_desc(sort_tmp, sort_tmp_bytes, deg_key_a, deg_key_b, iota,
                                           fwd_node_ids, (size_t)N, 0, 32, stream);
        if (e != hipSuccess) return fail((int)e, "stg_graph_build_device: node_ids sort: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(degrees_and_iota, dim3(vblocks), dim3(threads), 0, stream, bwd_row_offset, N,
                           out_degrees, deg_key_a, iota);
        sort_tmp_bytes = L.sort_tmp_bytes;
        e = rocprim::radix_sort_pairs_desc(sort_tmp, sort_tmp_bytes, deg_key_a, deg_key_b, iota,
                                           bwd_node_ids, (size_t)N, 0, 32, stream);
        if (e != hipSuccess) return fail((int)e, "stg_graph_build_device: node_ids sort: %s", hipGetErrorString(e));
    }
    return check_launch("stg_graph_build_device");
}
