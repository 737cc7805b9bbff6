// Kernels shared by the CSR builders (csr_build.hip, edge_store.hip): row offsets by binary search
// over sorted packed (row, col) keys, degrees, small helpers.  Internal linkage: one copy per TU.
#pragma once
#include "stg_common.hpp"

namespace stg {
namespace {

inline int key_bits_for(int32_t N)
{
    int b = 1;
    while (b < 31 && (int64_t(1) << b) < int64_t(N)) ++b;
    return b;
}

// row_offset[v] = first position whose row (key >> bits) is >= v   (v in [0, N])
__global__ void row_offsets_by_search(const uint64_t *__restrict__ keys, int64_t E, int bits, int N,
                                      int *__restrict__ row_offset)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > N) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> bits) < (int64_t)v) lo = mid + 1;
        else hi = mid;
    }
    row_offset[v] = (int)lo;
}

__global__ void degrees_and_iota(const int *__restrict__ row_offset, int N, int *__restrict__ deg,
                                 unsigned *__restrict__ sort_key, int *__restrict__ iota)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int d = row_offset[v + 1] - row_offset[v];
    deg[v] = d;
    sort_key[v] = (unsigned)d;
    iota[v] = v;
}

inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace
}  // namespace stg
