// Direct (counting) build of both CSRs of a graph -- same inputs, outputs and bits as stg_graph_build_device
// (csr_build.hip; reference graph/static/csr.cu:68-157 + graph/static/static_graph.py:40-78), without a global sort.
//
// The sort-based builder runs three rocPRIM radix sorts.  Below ~1M items rocPRIM sorts by block sort + log2(n/1K)
// merge passes, so the per-snapshot rebuild of a small dynamic graph (BASELINE configs[4]: |V| = 25K, |E| = 250K)
// is 59 launches of 5-6 us: 0.35 ms of launch gaps, a third of a training step.  Rows of a graph are short, so here
// the order inside a row is found by COUNTING instead:
//   1. degrees by atomic histogram (+ id validation)                                             1 launch
//   2. row offsets: one workgroup per direction scans its degree array, notes the longest row    1 launch
//   3. forward scatter: edge i -> any free slot of row dst[i]  (key = src << 32 | i)             1 launch
//   4. forward rank: the slot of an entry inside its row = number of smaller keys in the row
//      (keys are unique: ties in src are broken by the caller position i, which is exactly the stable
//      (dst, src) order static_graph.py:65-72 defines); writes column, eid, perm_fwd and scatters the entry
//      into its backward row (key = eid << 32 | dst)                                             1 launch
//   5. backward rank: by eid, which for a fixed src is the (dst, eid) order of static_graph.py:75-78   1 launch
//   6. node_ids: rocPRIM sort of |V| degrees per direction (optional: NULL skips it)
// Ranking costs (row length) reads per entry, so rows longer than kDirectMaxRow are left to the sort-based
// builder: the scan raises status bit STG_BUILD_NEEDS_SORT and the caller falls back.
#include "stg_common.hpp"
#include "csr_kernels.hpp"

#include <cstring>   // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace stg {
namespace {

constexpr int kDirectMaxRow = 2048;
constexpr int kScanThreads = 1024;

__global__ void direct_init(int *__restrict__ status, int *__restrict__ cursors, int *__restrict__ in_deg,
                            int *__restrict__ out_deg, int N)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) *status = 0;
    if (v < N) {
        cursors[v] = 0;
        cursors[N + v] = 0;
        in_deg[v] = 0;
        out_deg[v] = 0;
    }
}

__global__ void direct_histogram(const int *__restrict__ src, const int *__restrict__ dst, int64_t E, int N,
                                 int *__restrict__ in_deg, int *__restrict__ out_deg, int *__restrict__ status)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int s = src[i], d = dst[i];
        if ((unsigned)s >= (unsigned)N || (unsigned)d >= (unsigned)N) {
            atomicOr(status, 1);               // same code as make_fwd_keys: endpoint out of range
            continue;
        }
        atomicAdd(in_deg + d, 1);
        atomicAdd(out_deg + s, 1);
    }
}

// blockIdx.x = 0: forward (rows = dst, lengths = in_deg); 1: backward.  Exclusive scan of N lengths by one workgroup:
// tiles of 4096 lengths, four consecutive ones per thread (coalesced), a shuffle scan inside each wave, the 16 wave
// totals through LDS, a running carry between tiles (|V| = 25 K: 7 tiles, ~ 6 us; the first version gave every thread
// one contiguous chunk -- strided loads and a 10-step Hillis-Steele over 1024 partials with 20 barriers: 31.6 us).
__global__ __launch_bounds__(kScanThreads) void direct_scan(const int *__restrict__ in_deg, const int *__restrict__ out_deg,
                                                           int N, int *__restrict__ fwd_ro, int *__restrict__ bwd_ro,
                                                           int *__restrict__ status)
{
    constexpr int kWavesScan = kScanThreads / 64;
    __shared__ int wsum[kWavesScan];
    const int *deg = blockIdx.x == 0 ? in_deg : out_deg;
    int *ro = blockIdx.x == 0 ? fwd_ro : bwd_ro;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0, longest = 0;
    for (int base = 0; base < N; base += 4 * kScanThreads) {
        const int v0 = base + 4 * tid;
        int d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = v0 + i < N ? deg[v0 + i] : 0;
        longest = max(max(longest, max(d[0], d[1])), max(d[2], d[3]));
        const int tot = d[0] + d[1] + d[2] + d[3];
        int x = tot;                                                  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kWavesScan; ++w) {
            const int t = wsum[w];
            before += w < wave ? t : 0;
            all += t;
        }
        int run = carry + before + (x - tot);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (v0 + i < N) ro[v0 + i] = run;
            run += d[i];
        }
        carry += all;
        __syncthreads();                                              // wsum is rewritten by the next tile
    }
    if (tid == 0) ro[N] = carry;
    if (longest > kDirectMaxRow) atomicOr(status, STG_BUILD_NEEDS_SORT);
}

__global__ void direct_scatter_fwd(const int *__restrict__ src, const int *__restrict__ dst, int64_t E, int N,
                                   const int *__restrict__ fwd_ro, int *__restrict__ cursor,
                                   uint64_t *__restrict__ key, int *__restrict__ row)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int s = src[i], d = dst[i];
        if ((unsigned)s >= (unsigned)N || (unsigned)d >= (unsigned)N) continue;
        const int slot = fwd_ro[d] + atomicAdd(cursor + d, 1);
        key[slot] = ((uint64_t)(unsigned)s << 32) | (uint64_t)(unsigned)i;
        row[slot] = d;
    }
}

// number of keys of [beg, end) smaller than `mine`
__device__ __forceinline__ int rank_in_row(const uint64_t *__restrict__ key, int beg, int end, uint64_t mine)
{
    int r = 0;
    int t = beg;
    for (; t + 8 <= end; t += 8) {                     // eight independent loads in flight: the loop is latency, not bandwidth
        uint64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = key[t + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) r += k[u] < mine ? 1 : 0;
    }
    for (; t < end; ++t) r += key[t] < mine ? 1 : 0;
    return r;
}

// The batched build keeps a placed edge as ONE 16-byte record {key (8 bytes), row, pad}: placing an edge is then one scattered
// store instead of two (an 8-byte key and a 4-byte row each paid a 64-byte sector of their own: direct2_place 71 -> see
// profiles/r05_build_records.json), and the rank passes read key and row of their entry with one coalesced 16-byte load.
struct __attribute__((aligned(16))) EdgeRec {
    uint64_t key;
    int row, aux;              // aux (forward records): the edge's slot in its backward row
};
__device__ __forceinline__ int rank_in_row(const EdgeRec *__restrict__ rec, int beg, int end, uint64_t mine)
{
    int r = 0;
    int t = beg;
    for (; t + 8 <= end; t += 8) {                     // eight independent loads in flight (see above)
        uint64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = rec[t + u].key;
#pragma unroll
        for (int u = 0; u < 8; ++u) r += k[u] < mine ? 1 : 0;
    }
    for (; t < end; ++t) r += rec[t].key < mine ? 1 : 0;
    return r;
}

__global__ void direct_rank_fwd(const uint64_t *__restrict__ key, const int *__restrict__ row, int64_t E,
                                const int *__restrict__ fwd_ro, const int *__restrict__ bwd_ro,
                                int *__restrict__ cursor_b, int *__restrict__ fwd_col, int *__restrict__ fwd_eid,
                                int64_t *__restrict__ perm_fwd, uint64_t *__restrict__ key_b, int *__restrict__ row_b,
                                const int *__restrict__ status)
{
    if (*status) return;                        // a row too long to rank by counting, or an endpoint out of range
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < E; t += stride) {
        const uint64_t mine = key[t];
        const int d = row[t];
        const int beg = fwd_ro[d];
        const int e = beg + rank_in_row(key, beg, fwd_ro[d + 1], mine);   // = eid
        const int s = (int)(mine >> 32);
        fwd_col[e] = s;
        fwd_eid[e] = e;
        perm_fwd[e] = (int64_t)(unsigned)mine;
        const int slot = bwd_ro[s] + atomicAdd(cursor_b + s, 1);
        key_b[slot] = ((uint64_t)(unsigned)e << 32) | (uint64_t)(unsigned)d;
        row_b[slot] = s;
    }
}

__global__ void direct_rank_bwd(const uint64_t *__restrict__ key_b, const int *__restrict__ row_b, int64_t E,
                                const int *__restrict__ bwd_ro, int *__restrict__ bwd_col, int *__restrict__ bwd_eid,
                                const int *__restrict__ status)
{
    if (*status) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < E; t += stride) {
        const uint64_t mine = key_b[t];
        const int s = row_b[t];
        const int beg = bwd_ro[s];
        const int o = beg + rank_in_row(key_b, beg, bwd_ro[s + 1], mine);
        bwd_eid[o] = (int)(mine >> 32);
        bwd_col[o] = (int)(unsigned)mine;
    }
}

__global__ void direct_sort_keys(const int *__restrict__ deg, int N, unsigned *__restrict__ sort_key, int *__restrict__ iota)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    sort_key[v] = (unsigned)deg[v];
    iota[v] = v;
}

// ---- the same build in FIVE launches with ONE atomic pass (round 3) -------------------------------------------------------
// A snapshot rebuilt every epoch (NaiveGraph(resident=False)) pays the six launches above plus what follows every new CSR
// in the training loop: the in-degree norm and its per-edge gathers (3 more launches) -- 88 + 14 us at |E| = 250 K, of
// which the three random-address atomic passes (histogram, two scatter cursors) are 60.  Here the histogram pass RETURNS
// each edge's arrival index inside its two rows, so placing needs no cursor; the scan also writes norm = in_deg^-1/2; the
// two ranking passes also write norm gathered through their columns; the last pass leaves the counters zero again for the
// next build (they live in a caller-owned buffer that is zero between builds: no init launch).
//
// Every kernel of this build takes its operands as jobs.j[blockIdx.z]: ONE build is a batch of one (CAP = 1), and the
// snapshots of a whole BPTT window -- independent builds over the same |V| -- go through the SAME launches as up to
// kBuildBatchMax jobs (stg_graph_build_direct2_batch_device): a 250 K-edge build is six launches that each fill a fraction of
// the chip for 5-10 us, so eight of them side by side cost little more than one.
struct BJob {
    const int *src, *dst;
    int64_t E;
    int64_t *perm;
    int *fwd_ro, *fwd_col, *fwd_eid, *bwd_ro, *bwd_col, *bwd_eid, *in_deg, *out_deg;
    float *norm, *nc_f, *nc_b;
    int *cnt;                               // 2 npad counters, zero between builds
    EdgeRec *rec_f, *rec_b;                  // placed edges, forward / backward rows (16-byte records)
    int *pos_f, *pos_b, *part;
    int chunk_shift, lds;                   // lds != 0: counted by direct3_count (per-chunk histograms in LDS) and combined
    int id;                                 // the caller's name for this build (stg_build_job::id; 0: none)
};

// The sticky word is shared by every build of a device: the FIRST build that fails leaves its id in bits 8 .. 30 beside the code
// bits (0 .. 7), later failures only add their code bits -- the reader can name the build (a bad edge list was
// reported by whichever graph's check ran next).
__device__ __forceinline__ void report_build(int *status, int code, int id)
{
    const int old = atomicCAS(status, 0, code | ((id & 0x7fffff) << 8));
    if (old != 0) atomicOr(status, code);
}
constexpr int kBuildBatchMax = STG_BUILD_BATCH_MAX;
template <int CAP>
struct BuildJobs {
    BJob j[CAP];
};
static_assert(sizeof(BuildJobs<kBuildBatchMax>) <= 3584, "the batch travels as a kernel argument");

template <int CAP>
__global__ __launch_bounds__(kBlock) void direct2_count(const BuildJobs<CAP> jobs, int N, int npad, int *__restrict__ status)
{
    const BJob &J = jobs.j[blockIdx.z];
    if (J.lds) return;
    const int *__restrict__ src = J.src, *__restrict__ dst = J.dst;
    int *__restrict__ cnt = J.cnt, *__restrict__ pos_f = J.pos_f, *__restrict__ pos_b = J.pos_b;
    const int64_t E = J.E;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int s = src[i], d = dst[i];
        if ((unsigned)s >= (unsigned)N || (unsigned)d >= (unsigned)N) {
            report_build(status, 1, J.id);
            pos_f[i] = -1;
            continue;
        }
        pos_f[i] = atomicAdd(cnt + d, 1);
        pos_b[i] = atomicAdd(cnt + npad + s, 1);
    }
}

// The same pass with the histograms in LDS (|V| <= kLdsCountMaxN): the 2 E returning atomics of direct2_count go to random
// addresses of a 200 KB table in L2 and take 23.6 us at |E| = 250 K -- the chip's scattered-atomic rate.  Here workgroup
// (chunk c, side) counts ITS kLdsChunks-th of the edges into a private histogram of |V| ints in LDS (side 0: by destination,
// side 1: by source), leaves every edge's arrival index inside (chunk, row) in pos_f / pos_b and the histogram in
// part[side][c][.]; direct3_combine turns the chunks' counts of a row into exclusive bases (in place) and the row's total
// (cnt, as direct2_count leaves it); placing adds base[chunk of the edge][row] to the arrival index.
constexpr int kLdsChunks = 16;
constexpr int kLdsCountMaxN = 40 * 1024;                        // 160 KB of LDS
template <int CAP>
__global__ __launch_bounds__(kScanThreads) void direct3_count(const BuildJobs<CAP> jobs, int N, int npad, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) int hist[];  // npad = |V| rounded up to 4 ints (also the row stride of part)
    const BJob &J = jobs.j[blockIdx.z];
    if (!J.lds) return;
    const int *__restrict__ src = J.src, *__restrict__ dst = J.dst;
    int *__restrict__ part = J.part, *__restrict__ pos_f = J.pos_f, *__restrict__ pos_b = J.pos_b;
    const int64_t E = J.E, chunk_len = (int64_t)1 << J.chunk_shift;
    const int c = (int)blockIdx.x, side = (int)blockIdx.y;
    for (int v = 4 * threadIdx.x; v < npad; v += 4 * kScanThreads) *reinterpret_cast<int4 *>(hist + v) = make_int4(0, 0, 0, 0);
    __syncthreads();
    const int64_t lo = c * chunk_len, hi = std::min<int64_t>(E, lo + chunk_len);
    int *pos = side ? pos_b : pos_f;
    constexpr int U = 16;                                  // edges per thread in flight (|E| = 250 K: a chunk in ONE round): the loop is a chain of load -> atomic -> store
    for (int64_t base = lo + threadIdx.x; base < hi; base += (int64_t)U * kScanThreads) {
        int s[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * kScanThreads;
            s[u] = i < hi ? src[i] : 0;
            d[u] = i < hi ? dst[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * kScanThreads;
            if (i >= hi) continue;
            if ((unsigned)s[u] >= (unsigned)N || (unsigned)d[u] >= (unsigned)N) {
                if (side == 0) {
                    report_build(status, 1, J.id);
                    pos_f[i] = -1;
                }
                continue;
            }
            pos[i] = atomicAdd(hist + (side ? s[u] : d[u]), 1);
        }
    }
    __syncthreads();
    int *out = part + ((int64_t)side * kLdsChunks + c) * npad;
    for (int v = 4 * threadIdx.x; v < npad; v += 4 * kScanThreads) *reinterpret_cast<int4 *>(out + v) = *reinterpret_cast<const int4 *>(hist + v);
}

template <int CAP>
__global__ __launch_bounds__(kBlock) void direct3_combine(const BuildJobs<CAP> jobs, int npad)
{
    const BJob &J = jobs.j[blockIdx.z];
    if (!J.lds) return;
    int *__restrict__ part = J.part, *__restrict__ cnt = J.cnt;
    const int v = 4 * (blockIdx.x * blockDim.x + threadIdx.x), side = (int)blockIdx.y;      // four rows per thread
    if (v >= npad) return;
    int *p = part + (int64_t)side * kLdsChunks * npad + v;
    int4 x[kLdsChunks];
#pragma unroll
    for (int c = 0; c < kLdsChunks; ++c) x[c] = *reinterpret_cast<const int4 *>(p + (int64_t)c * npad);     // all loads first
    int4 run = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < kLdsChunks; ++c) {
        *reinterpret_cast<int4 *>(p + (int64_t)c * npad) = run;
        run = make_int4(run.x + x[c].x, run.y + x[c].y, run.z + x[c].z, run.w + x[c].w);
    }
    *reinterpret_cast<int4 *>(cnt + side * npad + v) = run;          // (rows past |V| inside the padding: zero, as the scan expects)
}

// blockIdx.x = 0: forward rows (lengths cnt[0 .. N)), 1: backward rows (cnt[npad .. npad + N)), npad = N rounded up to 4.
// Up to 16 tiles of 4096 lengths are held in registers at once (|V| <= 65536: one pass): every tile is loaded with one
// coalesced int4 per thread, the 16 tiles' wave scans run interleaved on the shuffle network, ONE barrier publishes the
// wave totals, and offsets / degrees / norm leave as coalesced 16-byte stores.  (A contiguous chunk of rows per thread --
// strided 4-byte loads and stores -- took 47 us at |V| = 25 K; the seven-tile sequential loop of direct_scan 16.)
constexpr int kScan2Tiles = 16;
template <int CAP>
__global__ __launch_bounds__(kScanThreads) void direct2_scan(const BuildJobs<CAP> jobs, int npad, int N, int *__restrict__ status)
{
    const BJob &J = jobs.j[blockIdx.z];
    const int *__restrict__ cnt = J.cnt;
    int *__restrict__ fwd_ro = J.fwd_ro, *__restrict__ bwd_ro = J.bwd_ro, *__restrict__ in_deg = J.in_deg, *__restrict__ out_deg = J.out_deg;
    float *__restrict__ norm = J.norm;
    constexpr int kWavesScan = kScanThreads / 64;                 // 16
    constexpr int kFlat = kScan2Tiles * kWavesScan;               // 256 (tile, wave) totals = 4 waves of entries
    __shared__ int wsum[kFlat];                                   // [tile * 16 + wave]: totals, then exclusive bases
    __shared__ int wtot[kFlat / 64];
    const int side = blockIdx.x;
    const int *deg = cnt + (side ? npad : 0);
    int *ro = side ? bwd_ro : fwd_ro;
    int *dout = side ? out_deg : in_deg;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0, longest = 0;
    for (int g0 = 0; g0 < N; g0 += kScan2Tiles * 4 * kScanThreads) {
        const int nt = min(kScan2Tiles, (N - g0 + 4 * kScanThreads - 1) / (4 * kScanThreads));     // tiles in use (uniform)
        int4 d[kScan2Tiles];
        int tot[kScan2Tiles], x[kScan2Tiles];
#pragma unroll
        for (int j = 0; j < kScan2Tiles; ++j) {
            d[j] = make_int4(0, 0, 0, 0);
            if (j < nt) {
                const int v0 = g0 + j * 4 * kScanThreads + 4 * tid;
                // counters past N (inside the padding) are zero by contract, so a whole int4 is safe wherever v0 < npad
                if (v0 < npad) d[j] = *reinterpret_cast<const int4 *>(deg + v0);
            }
            tot[j] = d[j].x + d[j].y + d[j].z + d[j].w;
            longest = max(max(longest, max(d[j].x, d[j].y)), max(d[j].z, d[j].w));
            x[j] = tot[j];
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < kScan2Tiles; ++j) {
                if (j < nt) {
                    const int y = __shfl_up(x[j], off, 64);
                    if (lane >= off) x[j] += y;
                }
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < kScan2Tiles; ++j)
                if (j < nt) wsum[j * kWavesScan + wave] = x[j];
        }
        __syncthreads();
        // exclusive scan of the (tile, wave) totals in tile-major order by the first four waves
        int e = 0, incl = 0;
        if (tid < kFlat) {
            e = tid < nt * kWavesScan ? wsum[tid] : 0;
            incl = e;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off, 64);
                if (lane >= off) incl += y;
            }
            if (lane == 63) wtot[wave] = incl;
        }
        __syncthreads();
        const int t0 = wtot[0], t1 = wtot[1], t2 = wtot[2], t3 = wtot[3];
        if (tid < kFlat) wsum[tid] = (wave > 0 ? t0 : 0) + (wave > 1 ? t1 : 0) + (wave > 2 ? t2 : 0) + incl - e;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kScan2Tiles; ++j) {
            if (j >= nt) continue;
            const int v0 = g0 + j * 4 * kScanThreads + 4 * tid;
            const int o0 = carry + wsum[j * kWavesScan + wave] + (x[j] - tot[j]);
            const int o1 = o0 + d[j].x, o2 = o1 + d[j].y, o3 = o2 + d[j].z;
            if (v0 + 3 < N) {
                *reinterpret_cast<int4 *>(ro + v0) = make_int4(o0, o1, o2, o3);
                *reinterpret_cast<int4 *>(dout + v0) = d[j];
                if (side == 0 && norm) {
                    const int dd[4] = {d[j].x, d[j].y, d[j].z, d[j].w};
                    float nn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) nn[i] = dd[i] > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)dd[i])) : 0.f;   // = degree_norm_kernel
                    *reinterpret_cast<float4 *>(norm + v0) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                }
            } else if (v0 < N) {
                const int oo[4] = {o0, o1, o2, o3}, dd[4] = {d[j].x, d[j].y, d[j].z, d[j].w};
                for (int i = 0; i < 4 && v0 + i < N; ++i) {
                    ro[v0 + i] = oo[i];
                    dout[v0 + i] = dd[i];
                    if (side == 0 && norm) norm[v0 + i] = dd[i] > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)dd[i])) : 0.f;
                }
            }
        }
        carry += t0 + t1 + t2 + t3;
        __syncthreads();                                              // wsum / wtot are rewritten by the next group
    }
    if (tid == 0) ro[N] = carry;
    if (longest > kDirectMaxRow) report_build(status, STG_BUILD_NEEDS_SORT, J.id);
}

// (base != NULL: pos_f / pos_b count inside (chunk, row) -- direct3_count -- and base[chunk][row] is added; npad = its row stride)
template <int CAP>
__global__ __launch_bounds__(kBlock) void direct2_place(const BuildJobs<CAP> jobs, int npad, const int *__restrict__ status)
{
    if (*status) return;
    const BJob &J = jobs.j[blockIdx.z];
    const int *__restrict__ src = J.src, *__restrict__ dst = J.dst, *__restrict__ fwd_ro = J.fwd_ro, *__restrict__ pos_f = J.pos_f;
    const int *__restrict__ bwd_ro = J.bwd_ro, *__restrict__ pos_b = J.pos_b;
    EdgeRec *__restrict__ rec = J.rec_f;
    const int *__restrict__ base_f = J.lds ? J.part : nullptr;
    const int *__restrict__ base_b = J.lds ? J.part + (size_t)kLdsChunks * npad : nullptr;
    const int chunk_shift = J.chunk_shift;
    const int64_t E = J.E;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += stride) {
        const int d = dst[i];
        int slot = fwd_ro[d] + pos_f[i];
        if (base_f) slot += base_f[(i >> chunk_shift) * npad + d];
        // the edge's slot in its BACKWARD row is known here too (pos_b[i] is a coalesced read in this pass, a random one in the
        // rank pass): it rides in the record's fourth word
        const int s = src[i];
        int slot_b = bwd_ro[s] + pos_b[i];
        if (base_b) slot_b += base_b[(i >> chunk_shift) * npad + s];
        // one 16-byte store: {key = src << 32 | i, row = d, backward slot}
        *reinterpret_cast<uint4 *>(rec + slot) = make_uint4((unsigned)i, (unsigned)s, (unsigned)d, (unsigned)slot_b);
    }
}

template <int CAP>
__global__ __launch_bounds__(kBlock) void direct2_rank_fwd(const BuildJobs<CAP> jobs, int npad, const int *__restrict__ status)
{
    if (*status) return;
    const BJob &J = jobs.j[blockIdx.z];
    const EdgeRec *__restrict__ rec = J.rec_f;
    const int *__restrict__ fwd_ro = J.fwd_ro;
    int *__restrict__ fwd_col = J.fwd_col, *__restrict__ fwd_eid = J.fwd_eid;
    int64_t *__restrict__ perm_fwd = J.perm;
    EdgeRec *__restrict__ rec_b = J.rec_b;
    const float *__restrict__ norm = J.norm;
    float *__restrict__ nc_fwd = J.nc_f;
    const int64_t E = J.E;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < E; t += stride) {
        const uint4 r4 = *reinterpret_cast<const uint4 *>(rec + t);
        const uint64_t mine = ((uint64_t)r4.y << 32) | r4.x;
        const int d = (int)r4.z;
        const int beg = fwd_ro[d];
        const int e = beg + rank_in_row(rec, beg, fwd_ro[d + 1], mine);   // = eid
        const int s = (int)r4.y;
        const unsigned i = r4.x;                                          // caller position
        fwd_col[e] = s;
        fwd_eid[e] = e;
        perm_fwd[e] = (int64_t)i;
        if (nc_fwd) nc_fwd[e] = norm[s];
        // {key = eid << 32 | dst, row = s} into its backward row (slot found by direct2_place), one store
        *reinterpret_cast<uint4 *>(rec_b + r4.w) = make_uint4((unsigned)d, (unsigned)e, (unsigned)s, 0u);
    }
}

template <int CAP>
__global__ __launch_bounds__(kBlock) void direct2_rank_bwd(const BuildJobs<CAP> jobs, int npad, const int *__restrict__ status)
{
    const BJob &J = jobs.j[blockIdx.z];
    const EdgeRec *__restrict__ rec_b = J.rec_b;
    const int *__restrict__ bwd_ro = J.bwd_ro;
    int *__restrict__ bwd_col = J.bwd_col, *__restrict__ bwd_eid = J.bwd_eid, *__restrict__ cnt = J.cnt;
    const float *__restrict__ norm = J.norm;
    float *__restrict__ nc_bwd = J.nc_b;
    const int64_t E = J.E;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t v = first; v < 2 * (int64_t)npad; v += stride) cnt[v] = 0;  // nobody reads the counters any more
    if (*status) return;
    for (int64_t t = first; t < E; t += stride) {
        const uint4 r4 = *reinterpret_cast<const uint4 *>(rec_b + t);
        const uint64_t mine = ((uint64_t)r4.y << 32) | r4.x;
        const int s = (int)r4.z;
        const int beg = bwd_ro[s];
        const int o = beg + rank_in_row(rec_b, beg, bwd_ro[s + 1], mine);
        const int d = (int)r4.x;
        bwd_eid[o] = (int)r4.y;
        bwd_col[o] = d;
        if (nc_bwd) nc_bwd[o] = norm[d];
    }
}

struct DirectLayout {
    size_t key_f, row_f, key_b, row_b, cursors, pos_f, pos_b, deg_key_a, deg_key_b, iota, sort_tmp, part, total, sort_tmp_bytes;
};

DirectLayout direct_layout(int64_t E, int32_t N)
{
    DirectLayout L{};
    const size_t e = (size_t)std::max<int64_t>(E, 1), n = (size_t)std::max<int32_t>(N, 1);
    size_t t = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, t, (unsigned *)nullptr, (unsigned *)nullptr, (int *)nullptr,
                                         (int *)nullptr, n, 0, 32);
    L.sort_tmp_bytes = t;
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
    L.key_f = take(e * 16);             // the batched build's 16-byte records (the single build uses the first 8 E bytes as keys)
    L.row_f = take(e * 4);
    L.key_b = take(e * 16);
    L.row_b = take(e * 4);
    L.cursors = take(2 * n * 4);
    L.pos_f = take(e * 4);
    L.pos_b = take(e * 4);
    L.deg_key_a = take(n * 4);
    L.deg_key_b = take(n * 4);
    L.iota = take(n * 4);
    L.sort_tmp = take(t);
    L.part = take(N <= kLdsCountMaxN ? (size_t)2 * kLdsChunks * ((n + 3) & ~(size_t)3) * 4 : 0);      // direct3_count's per-chunk histograms
    L.total = off;
    return L;
}


BJob make_job(const DirectLayout &L, char *ws, const int32_t *src, const int32_t *dst, int64_t E, int32_t N, int64_t *perm_fwd,
              int32_t *fwd_ro, int32_t *fwd_col, int32_t *fwd_eid, int32_t *bwd_ro, int32_t *bwd_col, int32_t *bwd_eid,
              int32_t *in_deg, int32_t *out_deg, float *norm, float *nc_f, float *nc_b, int32_t *counters)
{
    BJob j{};
    j.src = src, j.dst = dst, j.E = E, j.perm = perm_fwd;
    j.fwd_ro = fwd_ro, j.fwd_col = fwd_col, j.fwd_eid = fwd_eid, j.bwd_ro = bwd_ro, j.bwd_col = bwd_col, j.bwd_eid = bwd_eid;
    j.in_deg = in_deg, j.out_deg = out_deg, j.norm = norm, j.nc_f = nc_f, j.nc_b = nc_b, j.cnt = counters;
    j.rec_f = reinterpret_cast<EdgeRec *>(ws + L.key_f), j.rec_b = reinterpret_cast<EdgeRec *>(ws + L.key_b);
    j.pos_f = reinterpret_cast<int *>(ws + L.pos_f), j.pos_b = reinterpret_cast<int *>(ws + L.pos_b);
    j.part = reinterpret_cast<int *>(ws + L.part);
    // histograms in LDS when |V| fits it and there are enough edges per vertex to pay for writing and combining 2 x 16 x |V| counts
    const int mode = tuning().build_lds_count;
    j.lds = E > 0 && N <= kLdsCountMaxN && mode != 2 && (mode == 1 || (E >= 2 * (int64_t)N && E >= 32768));
    j.chunk_shift = 0;                                     // chunks of 2^shift edges (the chunk of an edge is a shift in the placing passes)
    while (((int64_t)kLdsChunks << j.chunk_shift) < E) ++j.chunk_shift;
    return j;
}

// The six launches over jobs.j[0 .. n): blockIdx.z = job; a job with E = 0 still gets its row offsets, degrees and norm.
template <int CAP>
int run_direct2(const BuildJobs<CAP> &jobs, int n, int32_t N, int32_t *sticky_status, hipStream_t stream, const char *who)
{
    const int npad = (std::max(N, 1) + 3) & ~3;
    int64_t emax = 0;
    bool any_lds = false, any_flat = false;
    for (int i = 0; i < n; ++i) {
        emax = std::max(emax, jobs.j[i].E);
        if (jobs.j[i].lds) any_lds = true;
        else if (jobs.j[i].E > 0) any_flat = true;
    }
    const int eblocks = (int)std::max<int64_t>(1, std::min<int64_t>((emax + kBlock - 1) / kBlock, 256 * 16));
    if (any_lds) {
        const size_t lds = (size_t)npad * sizeof(int);
        static PerDeviceOnce once;
        bool *raised = once.slot();
        if (lds > 64 * 1024 && !*raised) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(direct3_count<CAP>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsCountMaxN * (int)sizeof(int));
            if (e != hipSuccess) return fail((int)e, "%s: %s", who, hipGetErrorString(e));
            *raised = true;
        }
        hipLaunchKernelGGL(direct3_count<CAP>, dim3(kLdsChunks, 2, n), dim3(kScanThreads), lds, stream, jobs, N, npad, sticky_status);
        hipLaunchKernelGGL(direct3_combine<CAP>, dim3((npad / 4 + kBlock - 1) / kBlock, 2, n), dim3(kBlock), 0, stream, jobs, npad);
    }
    if (any_flat)
        hipLaunchKernelGGL(direct2_count<CAP>, dim3(eblocks, 1, n), dim3(kBlock), 0, stream, jobs, N, npad, sticky_status);
    hipLaunchKernelGGL(direct2_scan<CAP>, dim3(2, 1, n), dim3(kScanThreads), 0, stream, jobs, npad, N, sticky_status);
    if (emax > 0) {
        hipLaunchKernelGGL(direct2_place<CAP>, dim3(eblocks, 1, n), dim3(kBlock), 0, stream, jobs, npad, sticky_status);
        hipLaunchKernelGGL(direct2_rank_fwd<CAP>, dim3(eblocks, 1, n), dim3(kBlock), 0, stream, jobs, npad, sticky_status);
    }
    // (also re-zeroes the counters: launched even for E = 0)
    hipLaunchKernelGGL(direct2_rank_bwd<CAP>, dim3(std::max(eblocks, (2 * npad + kBlock - 1) / kBlock), 1, n), dim3(kBlock), 0, stream,
                       jobs, npad, sticky_status);
    return 0;
}

}  // namespace
}  // namespace stg

extern "C" size_t stg_graph_build_direct_workspace_bytes(int64_t E, int32_t N)
{
    if (E < 0 || N < 0) return 0;
    return stg::direct_layout(E, N).total;
}

extern "C" int stg_graph_build_direct_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                             int64_t *perm_fwd, int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                                             int32_t *fwd_eids, int32_t *fwd_node_ids, int32_t *bwd_row_offset,
                                             int32_t *bwd_column_indices, int32_t *bwd_eids, int32_t *bwd_node_ids,
                                             int32_t *in_degrees, int32_t *out_degrees, int32_t *status, void *workspace,
                                             size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_direct_device: negative size");
    if (E >= (int64_t(1) << 31))
        return fail(STG_ERR_UNSUPPORTED, "stg_graph_build_direct_device: E=%lld does not fit int32 edge ids", (long long)E);
    if ((E > 0 && (!src || !dst || !perm_fwd || !fwd_column_indices || !fwd_eids || !bwd_column_indices || !bwd_eids)) ||
        !fwd_row_offset || !bwd_row_offset || !status || !workspace || (N > 0 && (!in_degrees || !out_degrees)) ||
        (!fwd_node_ids != !bwd_node_ids))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_direct_device: NULL pointer argument");
    const DirectLayout L = direct_layout(E, N);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_graph_build_direct_device: workspace %zu < required %zu", workspace_bytes, L.total);
    char *ws = static_cast<char *>(workspace);
    auto *key_f = reinterpret_cast<uint64_t *>(ws + L.key_f);
    auto *row_f = reinterpret_cast<int *>(ws + L.row_f);
    auto *key_b = reinterpret_cast<uint64_t *>(ws + L.key_b);
    auto *row_b = reinterpret_cast<int *>(ws + L.row_b);
    auto *cursors = reinterpret_cast<int *>(ws + L.cursors);

    hipError_t e = hipSuccess;
    hipLaunchKernelGGL(direct_init, dim3((std::max(N, 1) + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, status, cursors,
                       in_degrees, out_degrees, N);

    const int eblocks = (int)std::max<int64_t>(1, std::min<int64_t>((E + kBlock - 1) / kBlock, 256 * 16));
    if (E > 0)
        hipLaunchKernelGGL(direct_histogram, dim3(eblocks), dim3(kBlock), 0, stream, src, dst, E, N, in_degrees, out_degrees, status);
    hipLaunchKernelGGL(direct_scan, dim3(2), dim3(kScanThreads), 0, stream, in_degrees, out_degrees, N, fwd_row_offset,
                       bwd_row_offset, status);
    if (E > 0) {
        hipLaunchKernelGGL(direct_scatter_fwd, dim3(eblocks), dim3(kBlock), 0, stream, src, dst, E, N, fwd_row_offset, cursors,
                           key_f, row_f);
        hipLaunchKernelGGL(direct_rank_fwd, dim3(eblocks), dim3(kBlock), 0, stream, key_f, row_f, E, fwd_row_offset,
                           bwd_row_offset, cursors + N, fwd_column_indices, fwd_eids, perm_fwd, key_b, row_b, status);
        hipLaunchKernelGGL(direct_rank_bwd, dim3(eblocks), dim3(kBlock), 0, stream, key_b, row_b, E, bwd_row_offset,
                           bwd_column_indices, bwd_eids, status);
    }
    if (N > 0 && fwd_node_ids) {
        auto *ka = reinterpret_cast<unsigned *>(ws + L.deg_key_a);
        auto *kb = reinterpret_cast<unsigned *>(ws + L.deg_key_b);
        auto *iota = reinterpret_cast<int *>(ws + L.iota);
        const int vblocks = (N + kBlock - 1) / kBlock;
        const int *degs[2] = {in_degrees, out_degrees};
        int32_t *outs[2] = {fwd_node_ids, bwd_node_ids};
        for (int side = 0; side < 2; ++side) {
            hipLaunchKernelGGL(direct_sort_keys, dim3(vblocks), dim3(kBlock), 0, stream, degs[side], N, ka, iota);
            size_t tmp = L.sort_tmp_bytes;
            e = rocprim::radix_sort_pairs_desc(ws + L.sort_tmp, tmp, ka, kb, iota, outs[side], (size_t)N, 0, 32, stream);
            if (e != hipSuccess) return fail((int)e, "stg_graph_build_direct_device: node_ids sort: %s", hipGetErrorString(e));
        }
    }
    return check_launch("stg_graph_build_direct_device");
}

extern "C" int stg_graph_build_direct2_device(const int32_t *src, const int32_t *dst, int64_t E, int32_t N,
                                              int64_t *perm_fwd, int32_t *fwd_row_offset, int32_t *fwd_column_indices,
                                              int32_t *fwd_eids, int32_t *fwd_node_ids, int32_t *bwd_row_offset,
                                              int32_t *bwd_column_indices, int32_t *bwd_eids, int32_t *bwd_node_ids,
                                              int32_t *in_degrees, int32_t *out_degrees, float *norm, float *norm_col_fwd,
                                              float *norm_col_bwd, int32_t *zero_counters, int32_t *sticky_status,
                                              void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E < 0 || N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_direct2_device: negative size");
    if (E >= (int64_t(1) << 31))
        return fail(STG_ERR_UNSUPPORTED, "stg_graph_build_direct2_device: E=%lld does not fit int32 edge ids", (long long)E);
    if ((E > 0 && (!src || !dst || !perm_fwd || !fwd_column_indices || !fwd_eids || !bwd_column_indices || !bwd_eids)) ||
        !fwd_row_offset || !bwd_row_offset || !sticky_status || !workspace || (N > 0 && (!in_degrees || !out_degrees || !zero_counters)) ||
        (!fwd_node_ids != !bwd_node_ids) || ((norm_col_fwd || norm_col_bwd) && !norm))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_direct2_device: NULL pointer argument");
    if (((uintptr_t)zero_counters | (uintptr_t)fwd_row_offset | (uintptr_t)bwd_row_offset | (uintptr_t)in_degrees |
         (uintptr_t)out_degrees | (uintptr_t)norm) & 15)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_graph_build_direct2_device: the per-vertex arrays must be 16-byte aligned");
    const DirectLayout L = direct_layout(E, N);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_graph_build_direct2_device: workspace %zu < required %zu", workspace_bytes, L.total);
    char *ws = static_cast<char *>(workspace);
    BuildJobs<1> jobs;
    jobs.j[0] = make_job(L, ws, src, dst, E, N, perm_fwd, fwd_row_offset, fwd_column_indices, fwd_eids, bwd_row_offset,
                         bwd_column_indices, bwd_eids, in_degrees, out_degrees, norm, norm_col_fwd, norm_col_bwd, zero_counters);
    if (const int rc = run_direct2(jobs, 1, N, sticky_status, stream, "stg_graph_build_direct2_device")) return rc;
    if (N > 0 && fwd_node_ids) {
        auto *ka = reinterpret_cast<unsigned *>(ws + L.deg_key_a);
        auto *kb = reinterpret_cast<unsigned *>(ws + L.deg_key_b);
        auto *iota = reinterpret_cast<int *>(ws + L.iota);
        const int vblocks = (N + kBlock - 1) / kBlock;
        const int *degs[2] = {in_degrees, out_degrees};
        int32_t *outs[2] = {fwd_node_ids, bwd_node_ids};
        for (int side = 0; side < 2; ++side) {
            hipLaunchKernelGGL(direct_sort_keys, dim3(vblocks), dim3(kBlock), 0, stream, degs[side], N, ka, iota);
            size_t tmp = L.sort_tmp_bytes;
            const hipError_t e = rocprim::radix_sort_pairs_desc(ws + L.sort_tmp, tmp, ka, kb, iota, outs[side], (size_t)N, 0, 32, stream);
            if (e != hipSuccess) return fail((int)e, "stg_graph_build_direct2_device: node_ids sort: %s", hipGetErrorString(e));
        }
    }
    return check_launch("stg_graph_build_direct2_device");
}

extern "C" int stg_graph_build_direct2_batch_device(const stg_build_job *jobs_in, int32_t n_jobs, int32_t N, int32_t *sticky_status,
                                                    void *stream_)
{
    using namespace stg;
    const char *who = "stg_graph_build_direct2_batch_device";
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_jobs < 0 || N <= 0 || (n_jobs > 0 && !jobs_in) || !sticky_status)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad argument (n_jobs=%d, N=%d)", who, n_jobs, N);
    if (n_jobs > kBuildBatchMax) return fail(STG_ERR_UNSUPPORTED, "%s: %d jobs > STG_BUILD_BATCH_MAX = %d", who, n_jobs, kBuildBatchMax);
    if (n_jobs == 0) return 0;
    BuildJobs<kBuildBatchMax> jobs{};
    for (int i = 0; i < n_jobs; ++i) {
        const stg_build_job &q = jobs_in[i];
        if (q.E < 0) return fail(STG_ERR_INVALID_ARGUMENT, "%s: job %d: negative size", who, i);
        if (q.E >= (int64_t(1) << 31)) return fail(STG_ERR_UNSUPPORTED, "%s: job %d: E=%lld does not fit int32 edge ids", who, i, (long long)q.E);
        if ((q.E > 0 && (!q.src || !q.dst || !q.perm_fwd || !q.fwd_column_indices || !q.fwd_eids || !q.bwd_column_indices || !q.bwd_eids)) ||
            !q.fwd_row_offset || !q.bwd_row_offset || !q.workspace || !q.in_degrees || !q.out_degrees || !q.zero_counters ||
            ((q.norm_col_fwd || q.norm_col_bwd) && !q.norm))
            return fail(STG_ERR_INVALID_ARGUMENT, "%s: job %d: NULL pointer argument", who, i);
        if (((uintptr_t)q.zero_counters | (uintptr_t)q.fwd_row_offset | (uintptr_t)q.bwd_row_offset | (uintptr_t)q.in_degrees |
             (uintptr_t)q.out_degrees | (uintptr_t)q.norm) & 15)
            return fail(STG_ERR_INVALID_ARGUMENT, "%s: job %d: the per-vertex arrays must be 16-byte aligned", who, i);
        for (int k = 0; k < i; ++k)
            if (jobs_in[k].zero_counters == q.zero_counters || jobs_in[k].workspace == q.workspace)
                return fail(STG_ERR_INVALID_ARGUMENT, "%s: jobs %d and %d share counters or workspace", who, k, i);
        const DirectLayout L = direct_layout(q.E, N);
        if (q.workspace_bytes < L.total)
            return fail(STG_ERR_WORKSPACE, "%s: job %d: workspace %zu < required %zu", who, i, q.workspace_bytes, L.total);
        jobs.j[i] = make_job(L, static_cast<char *>(q.workspace), q.src, q.dst, q.E, N, q.perm_fwd, q.fwd_row_offset, q.fwd_column_indices,
                             q.fwd_eids, q.bwd_row_offset, q.bwd_column_indices, q.bwd_eids, q.in_degrees, q.out_degrees, q.norm,
                             q.norm_col_fwd, q.norm_col_bwd, q.zero_counters);
        jobs.j[i].id = q.id;
    }
    if (const int rc = run_direct2(jobs, n_jobs, N, sticky_status, stream, who)) return rc;
    return check_launch(who);
}

extern "C" size_t stg_rows_by_degree_workspace_bytes(int32_t N)
{
    if (N < 0) return 0;
    const stg::DirectLayout L = stg::direct_layout(1, N);
    return L.total;
}

extern "C" int stg_rows_by_degree_device(const int32_t *degrees, int32_t N, int32_t *node_ids, void *workspace,
                                         size_t workspace_bytes, void *stream_)
{
    using namespace stg;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rows_by_degree_device: negative N");
    if (N == 0) return 0;
    if (!degrees || !node_ids || !workspace) return fail(STG_ERR_INVALID_ARGUMENT, "stg_rows_by_degree_device: NULL pointer argument");
    const DirectLayout L = direct_layout(1, N);
    if (workspace_bytes < L.total)
        return fail(STG_ERR_WORKSPACE, "stg_rows_by_degree_device: workspace %zu < required %zu", workspace_bytes, L.total);
    char *ws = static_cast<char *>(workspace);
    auto *ka = reinterpret_cast<unsigned *>(ws + L.deg_key_a);
    auto *kb = reinterpret_cast<unsigned *>(ws + L.deg_key_b);
    auto *iota = reinterpret_cast<int *>(ws + L.iota);
    hipLaunchKernelGGL(direct_sort_keys, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, degrees, N, ka, iota);
    size_t tmp = L.sort_tmp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs_desc(ws + L.sort_tmp, tmp, ka, kb, iota, node_ids, (size_t)N, 0, 32, stream);
    if (e != hipSuccess) return fail((int)e, "stg_rows_by_degree_device: %s", hipGetErrorString(e));
    return check_launch("stg_rows_by_degree_device");
}
