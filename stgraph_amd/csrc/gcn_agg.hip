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
#include <algorithm>
#include <type_traits>

#include "stg_common.hpp"
#include <cstdlib>

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

// Broadcast inside a G-lane row group from a source lane that is a compile-time constant after unrolling: quad
// permutes (DPP, no LDS crossbar trip) for G = 2 and 4.
template <int CTRL>
__device__ __forceinline__ int dpp_quad(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);       // every lane of a quad has a source
}

template <int G>
__device__ __forceinline__ int bcast_const_i(int v, int src)
{
    if constexpr (G == 1) {
        return v;
    } else if constexpr (G == 2) {
        return src ? dpp_quad<0xF5>(v) : dpp_quad<0xA0>(v);          // lanes {0,1} <- src, {2,3} <- 2 + src
    } else if constexpr (G == 4) {
        switch (src) {
            case 0: return dpp_quad<0x00>(v);
            case 1: return dpp_quad<0x55>(v);
            case 2: return dpp_quad<0xAA>(v);
            default: return dpp_quad<0xFF>(v);
        }
    } else {
        return bcast_i<G>(v, src);
    }
}

template <int G>
__device__ __forceinline__ float bcast_const_f(float v, int src)
{
    return __int_as_float(bcast_const_i<G>(__float_as_int(v), src));
}

// ---- long rows ------------------------------------------------------------------------------------------
// Rows with more than `threshold` edges when a row is narrower than a wave (L = 2^LOG2L < 64 lanes per row).
// In the main path below a row's L lanes walk its edges UNROLL at a time, so a hub of degree d costs
// d / UNROLL dependent round trips (a degree-168 row at F = 16: 21 of the launch's 26 us on a Cora-shaped
// graph, whose other rows need 5).  Here ONE WAVE owns one long row: its S = 64 / L lane groups fetch
// B = S * U different edges' rows at once (the coefficient products nc * x (* w) are formed by the fetching
// lane exactly as in the main path), stage the B products in an LDS tile [edge][feature], and lane group 0
// adds them to its single accumulator in CSR order -- the same sequence of fp32 additions as the
// reference's loop, with B rows in flight instead of UNROLL.  Two-deep pipeline: the index loads of batch
// b + 2 and the row gathers of batch b + 1 are in flight while batch b is summed.
// Long rows are found through `rows_by_degree` (non-increasing degree: every CSR builder emits it as
// node_ids); the FIRST `long_blocks` workgroups of the launch stride over that list and stop at the first
// row that is not long, so hubs start first and overlap the short rows instead of trailing them.
#ifndef STG_GCN_ROUND
#define STG_GCN_ROUND 8
#endif
#ifndef STG_TILE_U
#define STG_TILE_U 4
#endif
#ifndef STG_GCN_ROUND4
#define STG_GCN_ROUND4 16
#endif

template <int VEC, int LOG2L>
struct LongTile {
    static constexpr int L = 1 << LOG2L, S = kWave / L, W = L * VEC;     // W floats per staged row
    static constexpr int kFloats = 1024;                                  // 4 KB per wave, 16 KB per workgroup
    static constexpr int B0 = kFloats / W < 128 ? kFloats / W : 128;
    static constexpr int B = B0 < S ? S : B0;                             // edges per batch, >= one per lane group
    static constexpr int U = B / S;
    static_assert(U >= 1 && U * S == B && (B & (B - 1)) == 0, "batch must be a power of two multiple of S");
};

template <int VEC, int LOG2L, bool HAS_EW, bool EPI>
__device__ __forceinline__ void gcn_agg_long_rows(
    float *__restrict__ tile, int first_wave, int total_waves,
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ rows_by_degree, int N, int F, int F_active,
    const float *__restrict__ bias, int act, int threshold)
{
    using T = LongTile<VEC, LOG2L>;
    constexpr int L = T::L, S = T::S, W = T::W, B = T::B, U = T::U;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane >> LOG2L, j = lane & (L - 1);
    const int foff = j * VEC;                                        // position in the staged row
    const bool fok = foff < F_active;
    const int goff = VEC > 1 ? min(foff, F_active - VEC) : foff;     // ragged width: overlapping last window
    // feature f sits at tile position f, except in the last (shifted) window
    const int last = ((F_active + VEC - 1) / VEC - 1) * VEC;
    const int shift = last + VEC - F_active;

    for (int i = first_wave; i < N; i += total_waves) {
        const int r = rows_by_degree[i];
        const int beg = row_offsets[r];
        const int deg = row_offsets[r + 1] - beg;                    // wave-uniform
        if (deg <= threshold) break;

        // summation mapping: lane f owns feature f (+ 64 m): one LDS dword and one add per edge and lane, so
        // the order-preserving chain costs ~one dependent v_add per edge however wide the row is
        constexpr int M = (W + kWave - 1) / kWave;
        float acc[M];
        int pos[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc[m] = 0.f;
            const int f = lane + m * kWave;
            pos[m] = (f >= last ? f + shift : f) & (W - 1);
        }
        int ci[U];
        float nc[U], w[U], v[U][VEC];
        auto load_idx = [&](int base) {                              // coalesced: column, norm[col], w[eid]
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = base + u * S + sub;
                ci[u] = -1;
                nc[u] = 0.f;
                w[u] = 1.f;
                if (k < deg) {
                    const int e = beg + k;
                    ci[u] = column_indices[e];
                    nc[u] = nc_edge[e];
                    if constexpr (HAS_EW) w[u] = ew_edge[e];
                }
            }
        };
        auto gather = [&](float (&dst)[U][VEC]) {                    // the rows the current ci[] name
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) dst[u][q] = 0.f;
                if (ci[u] >= 0 && fok) vec_load_g<VEC>(dst[u], x + (int64_t)ci[u] * F + goff);
            }
        };
        load_idx(0);
        gather(v);
        float ncb[U], wb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ncb[u] = nc[u], wb[u] = w[u];
        if (B < deg) load_idx(B);
        for (int base = 0; base < deg; base += B) {
            const int cnt = min(B, deg - base);
#pragma unroll
            for (int u = 0; u < U; ++u) {                            // products -> LDS tile [edge][feature]
                float t[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    t[q] = ncb[u] * v[u][q];                         // Mul(norm_inb, h_inb)
                    if constexpr (HAS_EW) t[q] = t[q] * wb[u];       // Mul(., edge_weight)
                }
                if (fok) vec_store<VEC>(tile + (u * S + sub) * W + foff, t);
            }
            wave_lds_fence();
            if (base + B < deg) {                                    // next batch's gathers fly during the sums
                gather(v);
#pragma unroll
                for (int u = 0; u < U; ++u) ncb[u] = nc[u], wb[u] = w[u];
                if (base + 2 * B < deg) load_idx(base + 2 * B);
            }
            {
                constexpr int kRead = B < 16 ? B : 16;               // LDS reads in flight ahead of the add chain
                for (int k = 0; k < cnt; k += kRead) {
                    float t[kRead][M];
#pragma unroll
                    for (int u = 0; u < kRead; ++u)                  // (slots past cnt hold stale data, never added)
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            t[u][m] = tile[((k + u) & (B - 1)) * W + pos[m]];
#pragma unroll
                    for (int u = 0; u < kRead; ++u) {
                        if (k + u < cnt) {
#pragma unroll
                            for (int m = 0; m < M; ++m) acc[m] = acc[m] + t[u][m];     // AggSum, CSR order
                        }
                    }
                }
            }
            wave_lds_fence();
        }
        const float nr = norm_row[r];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int f = lane + m * kWave;
            if (f < W && f < F_active) {
                float o = acc[m] * nr;                               // Mul(., norm_cen)
                if constexpr (EPI) {
                    if (bias) o = o + bias[f];
                    if (act == STG_ACT_RELU) o = o < 0.f ? 0.f : o;
                }
                out[(int64_t)r * F + f] = o;
            }
        }
    }
}

// The long-row workgroups as a launch of their own: used behind the main kernel on graphs too large to be
// resident at once, where only giant hubs are taken out (there the main kernel must keep its registers and LDS
// to itself: inlining this path costs it 2x at F = 7, |V| = 2.8M).
template <int VEC, int LOG2L, bool HAS_EW, bool EPI>
__global__ __launch_bounds__(kBlock) void gcn_agg_long_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ rows_by_degree, int N, int F, int F_active,
    const float *__restrict__ bias, int act, int threshold)
{
    __shared__ __attribute__((aligned(16))) float tiles[kWavesPerBlock][LongTile<VEC, LOG2L>::kFloats];
    gcn_agg_long_rows<VEC, LOG2L, HAS_EW, EPI>(
        tiles[threadIdx.x >> 6], blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), gridDim.x * kWavesPerBlock,
        x, norm_row, nc_edge, ew_edge, out, row_offsets, column_indices, rows_by_degree, N, F, F_active, bias, act,
        threshold);
}

// ---- long rows WIDER than half a wave (F >= 128; round 3) ------------------------------------------------------------------
// gcn_agg_long_rows gives a long row ONE wave (rows up to 32 lanes wide).  At F = 128 the main kernel walks a row 4-8
// edges per index round trip with one wave, so a hub of 10^5 in-edges is ~2 x 10^4 dependent round trips: tens of ms
// behind a 1.1 ms launch.  Here a hub gets WORKGROUPS, and more than one: the order-preserving part of the work is one
// dependent fp32 add per edge and FEATURE, so the features of a long row are cut into `fs` slices (1, 4 or 16 by row
// length) and each slice is summed by its own workgroup:
//   gather  all 256 threads fetch 16-byte pieces of the slice for a batch of up to 1024 edges (one 32-byte sector per
//           edge at 8 floats per slice; up to 8 pieces in flight per thread) and write the PRODUCTS nc * x (* w), formed
//           exactly as in the main kernel, into an LDS tile [edge][slice width];
//   sum     thread f of the slice adds its column of the tile in CSR order: one accumulator per (row, feature), the same
//           sequence of fp32 additions as every other path -- bit-identical output; the next batch's gathers are in flight
//           meanwhile.
// The host knows how many rows fall in each class (counted once per graph: the caller's plan), so the grid is exactly the
// work items, longest rows first.  (Launching 1024 x 16 workgroups that find their work -- or none -- on the device cost the
// hubs' launch 0.6 ms of dispatch and a uniform graph 19 us per aggregation.)
constexpr int kWideTileFloats = 8192;           // products per batch (32 KB) ...
constexpr int kWideTileAlloc = kWideTileFloats + 4 * 256;   // ... in rows of B + 4 floats (one per feature of the slice)
constexpr int kWideMaxRounds = 8;

template <bool HAS_EW, bool EPI>
__global__ __launch_bounds__(kBlock) void gcn_agg_wide_long_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ rows_by_degree, int N, int F, int F_active,
    const float *__restrict__ bias, int act, int n16, int n4, int n1)
{
    __shared__ __attribute__((aligned(16))) float tile[kWideTileAlloc];
    const int t = (int)threadIdx.x;
    // Columns in super-blocks of at most 256 floats (any F_active >= 4): nsb of them, equal widths that are multiples of 4 except the
    // last.  A super-block of Ws floats is PT = ceil(Ws / 4) 16-byte pieces; when Ws % 4 != 0 the LAST piece is the window that
    // starts at Ws - 4 (gfx950 loads 16 bytes at any dword address), so up to three features are formed twice -- the same sums,
    // the same stores.
    const int nsb = (F_active + 255) / 256;
    const int sbw = 4 * ((F_active + 4 * nsb - 1) / (4 * nsb));
    // work items = (row, slice, super-block), longest rows first: rows_by_degree[0 .. n16) in 16 slices, the next n4 in 4, the next n1 whole
    for (int item0 = (int)blockIdx.x; item0 < (16 * n16 + 4 * n4 + n1) * nsb; item0 += (int)gridDim.x) {
        const int sb = item0 % nsb, item = item0 / nsb;
        const int c0 = sb * sbw, Ws = min(sbw, F_active - c0);
        const int PT = (Ws + 3) >> 2;
        int i, slice, fs;
        if (item < 16 * n16) {
            i = item >> 4, slice = item & 15, fs = 16;
        } else if (item < 16 * n16 + 4 * n4) {
            const int j = item - 16 * n16;
            i = n16 + (j >> 2), slice = j & 3, fs = 4;
        } else {
            i = n16 + n4 + (item - 16 * n16 - 4 * n4), slice = 0, fs = 1;
        }
        const int r = rows_by_degree[i];
        const int beg = row_offsets[r];
        const int deg = row_offsets[r + 1] - beg;                   // block-uniform
        if (deg <= 0 || Ws < 4) continue;
        while (fs > 1 && PT / fs < 2) fs >>= 1;                     // a slice is at least one 32-byte sector wide
        if (slice >= fs) continue;
        const int pp = (PT + fs - 1) / fs;
        const int p0 = slice * pp, P = min(pp, PT - p0);            // this slice: pieces [p0, p0 + P), P <= 64
        if (P <= 0) continue;
        const int Wc = 4 * P;
        // first float (inside the super-block) of piece q of this slice: the super-block's last piece may be the overlapping window
        auto piece_f = [&](int q) { return (p0 + q == PT - 1 && (Ws & 3)) ? Ws - 4 : 4 * (p0 + q); };
        const int ER = kBlock / P;                                  // edges per gather round
        int rounds = min(kWideMaxRounds, min(1024, kWideTileFloats / Wc) / ER);
        if (rounds < 1) rounds = 1;
        const int B = rounds * ER;                                  // edges per batch (B * Wc <= kWideTileFloats)
        const int LDB = B + 4;                                      // tile row (one feature): + 4 floats against bank conflicts
        const int piece = t % P, e_in = t / P;
        const bool gthread = t < ER * P;

        // (Two batches of gathers in flight and a hand-pipelined add loop were tried: 5.3 ms against 3.7 ms for this simple
        // form on tools/bench_powerlaw.py's graph -- the extra registers and branches cost more than the latency they hide.)
        float4 v[kWideMaxRounds];
        float nc[kWideMaxRounds], w[kWideMaxRounds];
        // Index and row-piece loads are separate steps, no per-lane branches (slots past the row's end are clamped to its last
        // edge and never staged): the indices of batch k + 2 and the row pieces of batch k + 1 are issued while batch k is summed,
        // so neither of the two dependent round trips sits on the critical path.  (As one guarded block per round the compiler
        // could not move a load across the guards: 2 x rounds DEPENDENT round trips per batch, 34 us per 64-edge batch at F = 128.)
        int cn[kWideMaxRounds];
        float ncn[kWideMaxRounds], wn[kWideMaxRounds];
        auto load_idx = [&](int base) {
#pragma unroll
            for (int rr = 0; rr < kWideMaxRounds; ++rr) {
                cn[rr] = 0;
                ncn[rr] = 0.f;
                wn[rr] = 1.f;
                if (rr < rounds) {                                   // block-uniform
                    const int e = beg + min(base + rr * ER + e_in, deg - 1);
                    cn[rr] = column_indices[e];
                    ncn[rr] = nc_edge[e];
                    if constexpr (HAS_EW) wn[rr] = ew_edge[e];
                }
            }
        };
        auto load_x = [&]() {                                        // rows named by cn[]; takes over ncn / wn as the batch's
#pragma unroll
            for (int rr = 0; rr < kWideMaxRounds; ++rr) {
                v[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
                nc[rr] = ncn[rr];
                w[rr] = wn[rr];
                if (rr < rounds) {
                    const stg_f4u u4 = *reinterpret_cast<const stg_f4u *>(x + (int64_t)cn[rr] * F + c0 + piece_f(min(piece, P - 1)));
                    v[rr] = make_float4(u4.x, u4.y, u4.z, u4.w);
                }
            }
        };
        float acc = 0.f;
        load_idx(0);
        load_x();
        if (B < deg) load_idx(B);
        for (int base = 0; base < deg; base += B) {
            const int cnt = min(B, deg - base);
#pragma unroll
            for (int rr = 0; rr < kWideMaxRounds; ++rr) {
                const int el = rr * ER + e_in;
                if (rr < rounds && gthread && el < cnt) {
                    float4 p;
                    p.x = nc[rr] * v[rr].x;                          // Mul(norm_inb, h_inb)
                    p.y = nc[rr] * v[rr].y;
                    p.z = nc[rr] * v[rr].z;
                    p.w = nc[rr] * v[rr].w;
                    if constexpr (HAS_EW) {                          // Mul(., edge_weight)
                        p.x = p.x * w[rr];
                        p.y = p.y * w[rr];
                        p.z = p.z * w[rr];
                        p.w = p.w * w[rr];
                    }
                    // the tile is FEATURE-major, [slice feature][edge] with rows of B + 4 floats: the summing thread of a
                    // feature then reads four consecutive edges per ds_read_b128 and spends ~1.25 instructions per edge (edge-major,
                    // one ds_read_b32 + index arithmetic + a guard per edge, was ~10 instructions = 50 cycles per edge for the one
                    // wave that sums: 2.2 ms for a row of 9.1e4 edges)
                    float *dstp = tile + (4 * piece) * LDB + el;
                    dstp[0] = p.x;
                    dstp[LDB] = p.y;
                    dstp[2 * LDB] = p.z;
                    dstp[3 * LDB] = p.w;
                }
            }
            __syncthreads();
            if (base + B < deg) {                                    // in flight during the sums
                load_x();
                if (base + 2 * B < deg) load_idx(base + 2 * B);
            }
            if (t < Wc) {
                const float *colp = tile + t * LDB;
                int k = 0;
                if (cnt >= 16) {
                    // the next 16 edges' LDS reads are in flight while the current 16 are added (one dependent v_add per edge)
                    float4 a0 = *reinterpret_cast<const float4 *>(colp), a1 = *reinterpret_cast<const float4 *>(colp + 4);
                    float4 a2 = *reinterpret_cast<const float4 *>(colp + 8), a3 = *reinterpret_cast<const float4 *>(colp + 12);
                    for (; k + 16 <= cnt; k += 16) {
                        const int kn = min(k + 16, B - 12);          // (past the last full group: a valid address, values unused)
                        const float4 b0 = *reinterpret_cast<const float4 *>(colp + kn), b1 = *reinterpret_cast<const float4 *>(colp + kn + 4);
                        const float4 b2 = *reinterpret_cast<const float4 *>(colp + kn + 8), b3 = *reinterpret_cast<const float4 *>(colp + kn + 12);
                        acc = acc + a0.x; acc = acc + a0.y; acc = acc + a0.z; acc = acc + a0.w;      // AggSum, CSR order
                        acc = acc + a1.x; acc = acc + a1.y; acc = acc + a1.z; acc = acc + a1.w;
                        acc = acc + a2.x; acc = acc + a2.y; acc = acc + a2.z; acc = acc + a2.w;
                        acc = acc + a3.x; acc = acc + a3.y; acc = acc + a3.z; acc = acc + a3.w;
                        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
                    }
                }
                for (; k < cnt; ++k) acc = acc + colp[k];
            }
            __syncthreads();
        }
        if (t < Wc) {
            const int f = c0 + piece_f(t >> 2) + (t & 3);            // the feature this tile row holds
            float o = acc * norm_row[r];                             // Mul(., norm_cen)
            if constexpr (EPI) {
                if (bias) o = o + bias[f];
                if (act == STG_ACT_RELU) o = o < 0.f ? 0.f : o;
            }
            out[(int64_t)r * F + f] = o;
        }
    }
}

template <int VEC, int LOG2G, int CHUNKS, bool HAS_EW, bool PRE, int UNROLL, bool EPI = false, bool LONG = false,
          bool A32 = false>
__global__ __launch_bounds__(kBlock) void gcn_agg_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row,
    const float *__restrict__ norm_col, const float *__restrict__ ew, float *__restrict__ out,
    const int *__restrict__ row_offsets, const int *__restrict__ column_indices,
    const int *__restrict__ eids, const int *__restrict__ node_ids, int N, int F, int F_active,
    const float *__restrict__ bias, int act, const int *__restrict__ rows_by_degree, int long_blocks,
    int long_threshold, int xcd_tile)
{
    constexpr int G = 1 << LOG2G;
    constexpr int ROWS_PER_WAVE = kWave / G;
    // One round = the R edges whose indices a row's G lanes hold at once, i.e. the gathers one index round trip
    // buys.  Narrow rows (G < 8 lanes) hold several edges per lane so that a round is still 8 edges: with R = G a
    // 2-lane row (F = 7) would need deg / 2 dependent index -> gather round trips.
    // Rows of four lanes hold 16 edges per round (two batches of 8 gathers, the second under a wave-uniform guard):
    // with 16 rows per wave some row usually has more than 8 edges, and a second ROUND is two dependent round trips.
    constexpr int R = (G == 4 || G == 8) ? STG_GCN_ROUND4 : (G < STG_GCN_ROUND ? STG_GCN_ROUND : G);
    constexpr int I = R / G;
    constexpr int U = I > 1 ? STG_GCN_ROUND : (UNROLL < G ? UNROLL : G);

    if constexpr (LONG) {
        // the first `long_blocks` workgroups take the long rows (see gcn_agg_long_rows)
        __shared__ __attribute__((aligned(16))) float tiles[kWavesPerBlock][LongTile<VEC, LOG2G>::kFloats];
        if ((int)blockIdx.x < long_blocks) {
            gcn_agg_long_rows<VEC, LOG2G, HAS_EW, EPI>(
                tiles[threadIdx.x >> 6], blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), long_blocks * kWavesPerBlock,
                x, norm_row, norm_col, ew, out, row_offsets, column_indices, rows_by_degree, N, F, F_active, bias, act,
                long_threshold);
            return;
        }
    }

    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & (G - 1);
    // Workgroups go round-robin to the 8 XCDs, each with an L2 of its own.  Deal the rows to the XCDs in runs of
    // `xcd_tile` consecutive workgroups (grid = a multiple of 8 * xcd_tile) instead of one workgroup at a time: on
    // graphs with locality (neighbours near the row in vertex order) a source row is then fetched into one L2
    // instead of eight, while runs stay short enough that a degree trend along the vertex order still spreads.
    int vb = (int)blockIdx.x - long_blocks;
    if (xcd_tile > 1) {
        const int s = vb >> 3;
        vb = ((s / xcd_tile) * 8 + (vb & 7)) * xcd_tile + s % xcd_tile;
    }
    const int wave_global = vb * ((int)blockDim.x >> 6) + (threadIdx.x >> 6);     // LONG: always kBlock threads
    const int idx = wave_global * ROWS_PER_WAVE + (lane >> LOG2G);

    int r = 0, beg = 0, deg = 0;
    float nr = 0.f;
    bool row_valid = idx < N;
    if (row_valid) {
        r = node_ids ? node_ids[idx] : idx;
        beg = row_offsets[r];
        deg = row_offsets[r + 1] - beg;
        nr = norm_row[r];
        if (deg > long_threshold) {          // taken by the long-row workgroups of this launch
            deg = 0;
            row_valid = false;
        }
    }
    const int max_deg = wave_max_nonneg(deg);

    for (int fbase = 0; fbase < F_active; fbase += G * VEC * CHUNKS) {
        float acc[CHUNKS][VEC];
        int foff[CHUNKS];
        bool fok[CHUNKS];
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            foff[ch] = fbase + (ch * G + j) * VEC;
            fok[ch] = foff[ch] < F_active;
            if constexpr (VEC > 1) foff[ch] = min(foff[ch], F_active - VEC);   // ragged width: overlapping last window
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[ch][i] = 0.f;
        }

        for (int base = 0; base < max_deg; base += R) {
            const int cnt = deg - base;                       // edges left in MY row (may be <= 0)
            const int cnt_max = min(R, max_deg - base);       // wave-uniform
            int c[I];
            float nc[I], w[I];
#pragma unroll
            for (int i = 0; i < I; ++i) {                     // lane j holds edges j, j + G, ... of the round
                c[i] = 0;
                nc[i] = 0.f;
                w[i] = 1.f;
                if (i * G + j < cnt) {
                    const int e = beg + base + i * G + j;
                    c[i] = column_indices[e];
                    if constexpr (PRE) {
                        nc[i] = norm_col[e];                  // = norm[col[e]], gathered once per graph
                        if constexpr (HAS_EW) w[i] = ew[e];   // = w[eid[e]]
                    } else {
                        nc[i] = norm_col[c[i]];
                        if constexpr (HAS_EW) w[i] = ew[eids[e]];
                    }
                }
            }
            // one batch = U gathers in flight, then their sums.  I > 1: k is the compile-time constant KC (so that the
            // element of c[] and the source lane are constants); I == 1: KC = 0 and k is the loop variable.
            auto batch = [&](auto kc, int k) {
                constexpr int KC = decltype(kc)::value;
                float v[U][CHUNKS][VEC];
                float ncs[U], ws[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int el = I == 1 ? 0 : (KC + u) >> LOG2G;
                    int ck;
                    if constexpr (I > 1) {                    // source lane u & (G - 1) is a constant here
                        ck = bcast_const_i<G>(c[el], u & (G - 1));
                        ncs[u] = bcast_const_f<G>(nc[el], u & (G - 1));
                        if constexpr (HAS_EW) ws[u] = bcast_const_f<G>(w[el], u & (G - 1));
                    } else {
                        ck = bcast_i<G>(c[el], kk & (G - 1));
                        ncs[u] = bcast_f<G>(nc[el], kk & (G - 1));
                        if constexpr (HAS_EW) ws[u] = bcast_f<G>(w[el], kk & (G - 1));
                    }
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if (kk < cnt && fok[ch]) {
                            if constexpr (A32) {
                                // whole matrix < 4 GB and < 2^24 rows (host-checked): one full-rate 24-bit multiply-add
                                // yields the byte offset and the load takes the scalar base, instead of a 64-bit
                                // multiply-add plus a 64-bit shift-add (each several issue cycles) per gather
                                const uint32_t off = __umul24((uint32_t)ck, (uint32_t)F * 4u) + (uint32_t)foff[ch] * 4u;
                                vec_load_g<VEC>(v[u][ch], reinterpret_cast<const float *>(
                                                              reinterpret_cast<const char *>(x) + off));
                            } else {
                                vec_load_g<VEC>(v[u][ch], x + (int64_t)ck * F + foff[ch]);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[u][ch][i] = 0.f;
                        }
                    }
                }
                // A slot past the end of a row holds nc = +0 and v = +0 (w = 1): its term is +0, and adding +0 to an
                // accumulator that started at +0 changes nothing (such a sum is never -0), so the sums need no guard.
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if constexpr (VEC == 1) {
                            float t = ncs[u] * v[u][ch][0];          // Mul(norm_inb, h_inb)
                            if constexpr (HAS_EW) t = t * ws[u];     // Mul(., edge_weight)
                            acc[ch][0] = acc[ch][0] + t;             // AggSum
                        } else {
                            // the same three operations on float pairs (packed fp32 instructions)
#pragma unroll
                            for (int i = 0; i < VEC; i += 2) {
                                gvec_t<2> t = {v[u][ch][i], v[u][ch][i + 1]};
                                gvec_t<2> a = {acc[ch][i], acc[ch][i + 1]};
                                t = t * ncs[u];
                                if constexpr (HAS_EW) t = t * ws[u];
                                a = a + t;
                                acc[ch][i] = a.x;
                                acc[ch][i + 1] = a.y;
                            }
                        }
                    }
                }
            };
            if constexpr (I > 1) {
                batch(std::integral_constant<int, 0>{}, 0);
                if constexpr (R > U) {
                    if (U < cnt_max) batch(std::integral_constant<int, U>{}, U);
                }
            } else {
                for (int k = 0; k < cnt_max; k += U) batch(std::integral_constant<int, 0>{}, k);
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
                    vec_store_g<VEC>(orow + foff[ch], o);
                }
            }
        }
    }
}

// ---- narrow rows, edge-dealt: a workgroup's rows through an LDS tile ------------------------------------------
// The row-group kernel above runs a wave to the LONGEST of its 8-64 rows, slot by slot: on a low-degree graph about
// half of the 8 slots of a round are empty and one longer row sends the wave into a second (index, gather) round
// trip (Cora-shaped, 16 rows per wave: 69 % of the waves; measured there: 38 % VALU utilisation, 65 % of a wave's
// life spent waiting).  Here a workgroup owns 256 / G consecutive rows = ONE CONTIGUOUS EDGE RANGE, and deals the
// EDGES, not the rows, to its 256 / G lane groups:
//   A  row offsets of the workgroup's rows -> LDS
//   B  lane group g takes edges g, g + groups, ... of a chunk of CAP = 4 x groups edges (a 16 KB tile): index + coefficients (coalesced, every
//      slot used), the 16-byte row pieces, and writes the PRODUCT nc * x (* w) -- formed exactly as above -- into an
//      LDS tile [edge][feature]
//   C  the lane group of a row adds its row's products from the tile in CSR order (one accumulator per feature, the
//      same sequence of fp32 additions); chunks follow each other in edge order
//   D  scale by the row norm (+ bias, ReLU) and store
// Hubs need no special path: their edges are dealt like any others and summed at one LDS read per edge.
// Measured on Cora x 1024 against the row-group kernel (same process): F = 4: 0.37 -> 0.58 of the roofline, F = 7:
// 0.48 -> 0.53; F = 12 / 16 / 32 (4 and 8 lanes per row, fewer rows per wave, so less to gain): 0.58 -> 0.50,
// 0.72 -> 0.63, 0.92 -> 0.72 -- used for rows of one or two lanes only.  A one-wave-per-workgroup variant (no
// barriers, 4 KB tile) was slower at every width (F = 4: 0.43, F = 7: 0.44).
// VPL = 16-byte pieces per lane.  Only 1 is instantiated: with 2 (rows of 9-16 floats as two lanes, a 32 KB tile)
// F = 16 drops to 0.54 and F = 12 to 0.45 -- the time follows the bytes through LDS, not the lane count.
template <int LOG2G, int VPL, bool HAS_EW, bool EPI, bool A32>
__global__ __launch_bounds__(kBlock) void gcn_agg_tile_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, int N, int F, int F_active, const float *__restrict__ bias, int act,
    int xcd_tile, int rows)
{
    constexpr int G = 1 << LOG2G, VEC = 4, W = G * VEC * VPL;
    constexpr int RB = kBlock / G;                  // lane groups per workgroup; it owns `rows` <= RB rows (see launch)
    constexpr int U = STG_TILE_U;                   // 4: measured best (8: F = 7 0.53 -> 0.46)                           // edges per lane group and chunk
    constexpr int CAP = U * RB;                     // edges per chunk: a 16 KB (VPL = 1) / 32 KB tile
    __shared__ int offs[RB + 1];
    __shared__ __attribute__((aligned(16))) float tile[CAP * W];

    int vb = (int)blockIdx.x;
    if (xcd_tile > 1) {
        const int s = vb >> 3;
        vb = ((s / xcd_tile) * 8 + (vb & 7)) * xcd_tile + s % xcd_tile;
    }
    const int r0 = vb * rows;
    if (r0 >= N) return;                            // whole workgroup (grid padded to a multiple of 8 runs)
    const int group = threadIdx.x >> LOG2G, j = threadIdx.x & (G - 1);
    int foff[VPL], goff[VPL];
    bool fok[VPL];
#pragma unroll
    for (int p = 0; p < VPL; ++p) {
        foff[p] = (j * VPL + p) * VEC;
        fok[p] = foff[p] < F_active;
        goff[p] = min(foff[p], F_active - VEC);     // ragged width: overlapping last window
    }

    for (int i = threadIdx.x; i <= rows; i += kBlock) offs[i] = row_offsets[min(r0 + i, N)];
    __syncthreads();
    const int e0 = offs[0], e1 = offs[rows];
    const bool has_row = group < rows;
    const int row = has_row ? r0 + group : N;
    const int rb = has_row ? offs[group] : 0, re = has_row ? offs[group + 1] : 0;   // rows >= N: empty (both = row_offsets[N])

    float acc[VPL][VEC];
#pragma unroll
    for (int p = 0; p < VPL; ++p)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[p][q] = 0.f;
    for (int cb = e0; cb < e1; cb += CAP) {
        const int cnt = min(CAP, e1 - cb);
        // ---- B
        {
            int c[U];
            float nc[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int el = group + u * RB;
                c[u] = 0;
                nc[u] = 0.f;
                w[u] = 1.f;
                if (el < cnt) {
                    c[u] = column_indices[cb + el];
                    nc[u] = nc_edge[cb + el];
                    if constexpr (HAS_EW) w[u] = ew_edge[cb + el];
                }
            }
            float v[U][VPL][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int p = 0; p < VPL; ++p) {
                    if (group + u * RB < cnt && fok[p]) {
                        if constexpr (A32) {
                            const uint32_t off = __umul24((uint32_t)c[u], (uint32_t)F * 4u) + (uint32_t)goff[p] * 4u;
                            vec_load_g<VEC>(v[u][p],
                                            reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + off));
                        } else {
                            vec_load_g<VEC>(v[u][p], x + (int64_t)c[u] * F + goff[p]);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) v[u][p][q] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int el = group + u * RB;
#pragma unroll
                for (int p = 0; p < VPL; ++p) {
                    if (el < cnt && fok[p]) {
                        float t[VEC];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            t[q] = nc[u] * v[u][p][q];                   // Mul(norm_inb, h_inb)
                            if constexpr (HAS_EW) t[q] = t[q] * w[u];    // Mul(., edge_weight)
                        }
                        vec_store<VEC>(tile + el * W + foff[p], t);
                    }
                }
            }
        }
        __syncthreads();
        // ---- C
        {
            const int lo = max(rb, cb) - cb, hi = min(re, cb + cnt) - cb;
            for (int el = lo; el < hi; el += 4) {
                float t[4][VPL][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int p = 0; p < VPL; ++p)
                        vec_load<VEC>(t[u][p], tile + min(el + u, CAP - 1) * W + foff[p]);   // past hi: not added
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (el + u < hi) {
#pragma unroll
                        for (int p = 0; p < VPL; ++p)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) acc[p][q] = acc[p][q] + t[u][p][q];   // AggSum, CSR order
                    }
                }
            }
        }
        if (cb + CAP < e1) __syncthreads();         // the tile is rewritten by the next chunk
    }
    // ---- D
    if (row < N) {
        const float nr = norm_row[row];
#pragma unroll
        for (int p = 0; p < VPL; ++p) {
            if (fok[p]) {
                float o[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) o[q] = acc[p][q] * nr;    // Mul(., norm_cen)
                if constexpr (EPI) {
                    if (bias) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) o[q] = o[q] + bias[goff[p] + q];
                    }
                    if (act == STG_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) o[q] = o[q] < 0.f ? 0.f : o[q];
                    }
                }
                vec_store_g<VEC>(out + (int64_t)row * F + goff[p], o);
            }
        }
    }
}

// ---- the same, software-pipelined over a workgroup's SEQUENCE of row blocks ------------------------------------------
// A workgroup of the kernel above lives through three dependent memory round trips (row offsets -> index and
// coefficients -> gathered rows) and two barriers for ~ 500 edges; on the Cora-shaped roofline graph the L2 request
// rate is 14 % of its peak and the HBM rate 0.36 (r02_pmc_cora_l2.json): the time is those round trips, not bytes.  Here
// the grid is what is resident at once and a workgroup walks row blocks s, s + S, ... of its XCD's list (the same
// XCD-local runs of 64 blocks as above); while the gathers of one chunk of edges are in flight it already holds the
// NEXT chunk's index and coefficients (issued one chunk ahead, of the same block or the first of the next) and the
// row offsets of the block after next (an LDS ring of three), so that one gather round trip per chunk remains
// exposed.  Products, the order of the additions and the epilogue are those of gcn_agg_tile_kernel: bit-identical.
// MEASURED (Cora x 1024, round 2): F = 7 0.540 against 0.536 of the roofline, F = 4 0.562 / 0.563, F = 8 0.569 / 0.578
// -- no gain, so the exposed round trips are not what bounds the narrow rows: kept behind `gcn_tile_pipe` = 2 (tested,
// never taken by default).  What the counters show instead (profiles/r02_pmc_cora_l1.json): every 28-byte row costs
// a 128-byte line through the CU's vector L1 (9.4 M L2 read requests = 1.2 GB per launch for 0.38 GB of rows, L1 hit
// rate 51 %, L2 hit rate 74 %, L2 read latency 376 clk), and the L1's in-order pipe spends 56 % of its cycles stalled
// on lines that are still in flight (TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1).  Dealing each chunk's edges in column
// order (a per-graph plan of sorted columns + tile slots, so that one gather instruction covers neighbouring lines)
// was bit-identical and moved F = 7 from 0.493 to 0.508 and F = 4 from 0.56 to 0.63 -- not worth a second copy of
// the index; dropped.
constexpr int kTileRun = 64;

template <int LOG2G, bool HAS_EW, bool EPI, bool A32>
__global__ __launch_bounds__(kBlock) void gcn_agg_tile_pipe_kernel(
    const float *__restrict__ x, const float *__restrict__ norm_row, const float *__restrict__ nc_edge,
    const float *__restrict__ ew_edge, float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, int N, int F, int F_active, const float *__restrict__ bias, int act,
    int nblocks, int rows, int last_edge)
{
    constexpr int G = 1 << LOG2G, VEC = 4, W = G * VEC;
    constexpr int RB = kBlock / G;                  // lane groups; a block is `rows` <= RB rows
    constexpr int U = STG_TILE_U;
    constexpr int CAP = U * RB;
    __shared__ int offs[3][RB + 1];
    __shared__ __attribute__((aligned(16))) float tile[CAP * W];

    const int xcd = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, S = (int)gridDim.x >> 3;
    auto block_of = [&](int p) { return ((p / kTileRun) * 8 + xcd) * kTileRun + p % kTileRun; };
    const int tid = (int)threadIdx.x;
    const int group = tid >> LOG2G, j = tid & (G - 1);
    const int foff = j * VEC;
    const bool fok = foff < F_active;
    const int goff = min(foff, F_active - VEC);     // ragged width: overlapping last window
    const bool has_row = group < rows;

    int p = s;
    int b = block_of(p);
    if (b >= nblocks) return;
    // Every global load below is UNCONDITIONAL (clamped address, value dropped afterwards): loads under a branch make
    // the compiler wait for all outstanding loads (s_waitcnt vmcnt(0)) at the join, which would serialise exactly the
    // round trips this kernel overlaps.
    // row offsets of a block: entry tid by every thread, entry RB (only when rows can be kBlock) by all, same address
    auto offs_load = [&](int blk, int &o0, int &o1) {
        const int r0 = min(blk, nblocks - 1) * rows;
        o0 = row_offsets[min(r0 + tid, N)];
        o1 = 0;
        if constexpr (RB == kBlock) o1 = row_offsets[min(r0 + RB, N)];
    };
    auto offs_store = [&](int slot, int o0, int o1) {
        if (tid <= RB) offs[slot][tid] = o0;
        if (RB == kBlock && tid == 0) offs[slot][RB] = o1;
    };
    {
        int o0, o1, q0, q1;
        offs_load(b, o0, o1);
        offs_load(block_of(p + S), q0, q1);
        offs_store(0, o0, o1);
        offs_store(1, q0, q1);
    }
    __syncthreads();

    int cN[U];
    float ncN[U], wN[U];
    auto idx_load = [&](int cb, int cnt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int el = group + u * RB;
            const int at = min(cb + min(el, max(cnt - 1, 0)), last_edge);
            const int cc = column_indices[at];
            const float nn = nc_edge[at];
            cN[u] = el < cnt ? cc : 0;
            ncN[u] = el < cnt ? nn : 0.f;
            wN[u] = 1.f;
            if constexpr (HAS_EW) {
                const float ww = ew_edge[at];
                wN[u] = el < cnt ? ww : 1.f;
            }
        }
    };
    idx_load(offs[0][0], min(CAP, offs[0][rows] - offs[0][0]));
    float bv[VEC] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI) {
        if (bias) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) bv[q] = bias[goff + q];
        }
    }

    for (int k = 0;; ++k, p += S) {
        const int cur = k % 3, nxt = (k + 1) % 3, nn = (k + 2) % 3;
        b = block_of(p);
        const bool has1 = block_of(p + S) < nblocks;
        const int b2 = block_of(p + 2 * S);
        const bool has2 = b2 < nblocks;
        int o0, o1;
        offs_load(b2, o0, o1);
        const int e0 = offs[cur][0], e1 = offs[cur][rows];
        const int row = has_row ? b * rows + group : N;
        const int rb = has_row ? offs[cur][group] : 0, re = has_row ? offs[cur][group + 1] : 0;
        const float nr = norm_row[min(row, N - 1)];                 // early: its latency hides behind the chunk

        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        int cb = e0;
        do {
            const int cnt = max(0, min(CAP, e1 - cb));
            int c[U];
            float nc[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = cN[u];
                nc[u] = ncN[u];
                w[u] = wN[u];
            }
            // ---- B: this chunk's gathers, then the next chunk's index and coefficients behind them
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (A32) {
                    const uint32_t off = __umul24((uint32_t)c[u], (uint32_t)F * 4u) + (uint32_t)goff * 4u;
                    vec_load_g<VEC>(v[u], reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + off));
                } else {
                    vec_load_g<VEC>(v[u], x + (int64_t)c[u] * F + goff);
                }
            }
            {
                int ncb = cb + CAP, ncnt = 0;
                if (ncb < e1) ncnt = min(CAP, e1 - ncb);
                else if (has1) {
                    ncb = offs[nxt][0];
                    ncnt = min(CAP, offs[nxt][rows] - ncb);
                }
                idx_load(ncb, ncnt);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int el = group + u * RB;
                float t[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    t[q] = nc[u] * v[u][q];                         // Mul(norm_inb, h_inb)
                    if constexpr (HAS_EW) t[q] = t[q] * w[u];       // Mul(., edge_weight)
                }
                if (el < cnt && fok) vec_store<VEC>(tile + el * W + foff, t);
            }
            __syncthreads();
            // ---- C
            {
                const int lo = max(rb, cb) - cb, hi = min(re, cb + cnt) - cb;
                for (int el = lo; el < hi; el += 4) {
                    float t[4][VEC];
#pragma unroll
                    for (int u = 0; u < 4; ++u) vec_load<VEC>(t[u], tile + min(el + u, CAP - 1) * W + foff);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (el + u < hi) {
#pragma unroll
                            for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + t[u][q];      // AggSum, CSR order
                        }
                    }
                }
            }
            cb += CAP;
            if (cb < e1) __syncthreads();           // the tile is rewritten by the next chunk
        } while (cb < e1);
        // ---- D
        {
            float o[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) o[q] = acc[q] * nr;       // Mul(., norm_cen)
            if constexpr (EPI) {
                if (bias) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o[q] = o[q] + bv[q];
                }
                if (act == STG_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) o[q] = o[q] < 0.f ? 0.f : o[q];
                }
            }
            if (row < N && fok) vec_store_g<VEC>(out + (int64_t)row * F + goff, o);
        }
        if (!has1) break;
        if (has2) offs_store(nn, o0, o1);
        __syncthreads();                            // ring slot nn visible; tile free for the next block
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
    const int *rows_by_degree = nullptr;   // rows by non-increasing degree: enables the long-row launch
    // rows of a wave and wider (F >= 128): how many of the first rows of rows_by_degree have >= 8192 / 2048..8191 /
    // hub_threshold + 1 .. 2047 edges (counted by the caller, once per graph); 0 rows: no hub launch
    int hub_threshold = 0, hub_n16 = 0, hub_n4 = 0, hub_n1 = 0;
};

constexpr int kLongRowThreshold = 16;      // edges; rows above it go to gcn_agg_long_kernel (G < 64 only)
constexpr int kGiantRowThreshold = 1024;
constexpr int kNoLongRows = 0x7fffffff;
constexpr int kXcdTile = 64;
constexpr int kPlainBlock = 64;

inline bool long_rows_enabled(const GcnArgs &a, int log2g) { return a.pre && a.rows_by_degree && log2g < 6; }

// one helper stream + two events per device, created on first use (never destroyed: process lifetime)
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = false, tried = false;
};

inline SideStream *side_stream()
{
    static SideStream per_device[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    SideStream &s = per_device[dev];
    if (!s.tried) {
        s.tried = true;
        s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
    }
    return s.ok ? &s : nullptr;
}

template <int VEC, int LOG2G, int CHUNKS, bool HAS_EW, bool PRE, int UNROLL, bool EPI = false>
void launch(const GcnArgs &a)
{
    constexpr bool kCanLong = PRE && LOG2G < 6 && CHUNKS == 1;
    constexpr bool kCanA32 = PRE && LOG2G <= 3 && CHUNKS == 1;      // narrow rows: instruction-, not bandwidth-bound
    // Workgroup size of the plain (non-merged) launch: a workgroup's wave slots are refilled only when ALL of its
    // waves have left, and the waves of this kernel are short-lived with ragged lifetimes -- one-wave workgroups
    // keep the SIMDs fuller (measured occupancy 5.5 -> see DESIGN.md) than 4-wave ones.
    int threads = tuning().gcn_block > 0 ? tuning().gcn_block : kPlainBlock;
    const int64_t blocks256 = ((int64_t)a.N + (kWave >> LOG2G) * kWavesPerBlock - 1) / ((kWave >> LOG2G) * kWavesPerBlock);
    const bool merged = kCanLong && long_rows_enabled(a, LOG2G) && blocks256 <= 256 * 8;
    if (merged) threads = kBlock;                                   // its LDS tiles are laid out for kBlock threads
    const int rows_per_block = (kWave >> LOG2G) * (threads / kWave);
    int64_t blocks = ((int64_t)a.N + rows_per_block - 1) / rows_per_block;
    // XCD runs (see the kernel): whole eighths of a grid that is resident at once, runs of 64 (x 256 threads)
    // workgroups otherwise
    int xcd_tile = tuning().gcn_xcd_tile;
    if (xcd_tile == 0) xcd_tile = blocks256 <= 256 * 8 ? (int)((blocks + 7) / 8) : kXcdTile * (kBlock / threads);
    if (blocks < 16) xcd_tile = 1;
    if (xcd_tile > 1) blocks = (blocks + 8 * xcd_tile - 1) / (8 * xcd_tile) * (8 * xcd_tile);
    const dim3 block(threads);
    const bool a32 = kCanA32 && tuning().gcn_addr32 != 1 && a.N <= (1 << 24) && a.F < (1 << 22) &&
                     (uint64_t)a.N * (uint64_t)a.F * 4u <= 0xffffffffull;

    auto main_kernel = [&](auto long_tag, int64_t grid, const int *rows_by_degree, int long_blocks, int threshold) {
        constexpr bool L = decltype(long_tag)::value;
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((unsigned)grid), block, 0, a.stream, a.x, a.norm_row, a.norm_col, a.ew, a.out,
                               a.row_offsets, a.column_indices, a.eids, a.node_ids, a.N, a.F, a.F_active, a.bias, a.act,
                               rows_by_degree, long_blocks, threshold, xcd_tile);
        };
        if constexpr (kCanA32) {
            if (a32) {
                go(gcn_agg_kernel<VEC, LOG2G, CHUNKS, HAS_EW, PRE, UNROLL, EPI, L, true>);
                return;
            }
        }
        go(gcn_agg_kernel<VEC, LOG2G, CHUNKS, HAS_EW, PRE, UNROLL, EPI, L, false>);
    };

    if constexpr (kCanA32 && VEC == 4) {
        // rows of one or two lanes in vertex order on a graph larger than one resident grid: deal edges, not rows
        if (!a.node_ids && a.F_active >= 4 &&
            (tuning().gcn_tile == 2 || (tuning().gcn_tile == 0 && LOG2G <= 1 && !merged && blocks256 > 256 * 8))) {
            constexpr int TLG = LOG2G, VPL = 1;
            // Rows per workgroup: as many as its lane groups.  Fewer (`gcn_tile_rows`), so that a mean workgroup's edges
            // fit ONE chunk of the tile (Cora-shaped, 4.9 edges per row with the self loops: 128 rows = 627 edges
            // against 512 per chunk; the lane groups without a row still take their share of the edges), measured
            // the same within noise (F = 7: 0.51-0.54 at 64 .. 128 rows): the second, short chunk is not the cost.
            constexpr int lane_groups = kBlock >> TLG;
            int rb = lane_groups;
            if (tuning().gcn_tile_rows > 0) rb = std::min(tuning().gcn_tile_rows, lane_groups);
            int64_t tb = ((int64_t)a.N + rb - 1) / rb;
            int tt = tuning().gcn_xcd_tile > 0 ? tuning().gcn_xcd_tile : kXcdTile;
            if (tb < 16 * tt) tt = 1;
            if (tt > 1) tb = (tb + 8 * tt - 1) / (8 * tt) * (8 * tt);
            if constexpr (VPL == 1) {
                // more row blocks than are resident at once: the persistent, software-pipelined form
                auto pipe = [&](auto kernel, int *cache) {
                    int dev = 0;
                    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
                    if (cache[dev] == 0) {
                        int per_cu = 0, cus = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess ||
                            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev == 63 ? 0 : dev) !=
                                hipSuccess)
                            per_cu = 4, cus = 256;
                        cache[dev] = std::max(per_cu * cus, 8);
                    }
                    const int64_t nb = ((int64_t)a.N + rb - 1) / rb;
                    // (its unconditional loads need one valid edge: a.E is 0 when unknown or when there is none)
                    if (a.E <= 0 || a.E > 0x7fffffffll || tuning().gcn_tile_pipe != 2) return false;
                    const int64_t grid = (std::min<int64_t>(cache[dev], nb) + 7) / 8 * 8;
                    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kBlock), 0, a.stream, a.x, a.norm_row,
                                       a.norm_col, a.ew, a.out, a.row_offsets, a.column_indices, a.N, a.F,
                                       a.F_active, a.bias, a.act, (int)nb, rb, (int)(a.E - 1));
                    return true;
                };
                static int resident32[64] = {}, resident64[64] = {};

                if (a32 ? pipe(gcn_agg_tile_pipe_kernel<TLG, HAS_EW, EPI, true>, resident32)
                        : pipe(gcn_agg_tile_pipe_kernel<TLG, HAS_EW, EPI, false>, resident64))
                    return;
            }
            auto go = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3((unsigned)tb), dim3(kBlock), 0, a.stream, a.x, a.norm_row, a.norm_col, a.ew,
                                   a.out, a.row_offsets, a.column_indices, a.N, a.F, a.F_active, a.bias, a.act, tt, rb);
            };
            if (a32) go(gcn_agg_tile_kernel<TLG, VPL, HAS_EW, EPI, true>);
            else go(gcn_agg_tile_kernel<TLG, VPL, HAS_EW, EPI, false>);
            return;
        }
    }
    if constexpr (kCanLong) {
        if (long_rows_enabled(a, LOG2G)) {
            // Long-row workgroups: how many rows are long is not known on the host (no sync), so one wave per 16
            // rows, at least 64 and at most 16384 waves; a wave that meets a short row first leaves at once.
            const int long_blocks = (int)std::max<int64_t>(
                std::min<int64_t>({((int64_t)a.N + kWavesPerBlock - 1) / kWavesPerBlock,
                                   std::max<int64_t>((int64_t)a.N / 64, 16), (int64_t)4096}), 1);
            const int forced = tuning().gcn_long_threshold;
            if (merged) {
                // The whole grid is resident at once: the launch lasts as long as its longest row, so every row
                // above 16 edges gets a wave of its own, in the SAME launch (long rows first, overlapping the rest).
                main_kernel(std::true_type{}, blocks + long_blocks, a.rows_by_degree, long_blocks,
                            forced > 0 ? forced : kLongRowThreshold);
                return;
            }
            // Larger graphs: the main path's many waves hide row latency; only giant hubs (> 1024 edges) would
            // still trail the launch.  They get their own launch so the main kernel keeps its registers and LDS.
            const int threshold = forced > 0 ? forced : kGiantRowThreshold;
            main_kernel(std::false_type{}, blocks, a.rows_by_degree, 0, threshold);
            hipLaunchKernelGGL((gcn_agg_long_kernel<VEC, LOG2G, HAS_EW, EPI>), dim3((unsigned)std::min(long_blocks, 512)),
                               dim3(kBlock), 0, a.stream, a.x, a.norm_row, a.norm_col, a.ew, a.out, a.row_offsets,
                               a.column_indices, a.rows_by_degree, a.N, a.F, a.F_active, a.bias, a.act, threshold);
            return;
        }
    }
    if constexpr (PRE && LOG2G == 6) {
        // rows of a whole wave and wider: hubs above 1024 edges go to feature-sliced workgroups (gcn_agg_wide_long_kernel)
        const int64_t items = 16 * (int64_t)a.hub_n16 + 4 * (int64_t)a.hub_n4 + a.hub_n1;
        if (a.rows_by_degree && items > 0 && a.hub_threshold > 0 && tuning().gcn_wide_long != 1 && a.F_active >= 4) {
            // The hubs' workgroups run BESIDE the main kernel when a helper stream is available: forked and joined through two
            // events (a stream capture records them as graph edges).  Only graphs WITH hubs pay the fork.
            SideStream *ss = tuning().gcn_wide_long == 2 ? nullptr : side_stream();
            hipStream_t wide_stream = a.stream;
            if (ss && hipEventRecord(ss->fork, a.stream) == hipSuccess && hipStreamWaitEvent(ss->stream, ss->fork, 0) == hipSuccess)
                wide_stream = ss->stream;
            const int64_t witems = items * ((a.F_active + 255) / 256);            // x column super-blocks of <= 256 floats
            hipLaunchKernelGGL((gcn_agg_wide_long_kernel<HAS_EW, EPI>), dim3((unsigned)std::min<int64_t>(witems, 1 << 20)), dim3(kBlock), 0,
                               wide_stream, a.x, a.norm_row, a.norm_col, a.ew, a.out, a.row_offsets, a.column_indices, a.rows_by_degree,
                               a.N, a.F, a.F_active, a.bias, a.act, a.hub_n16, a.hub_n4, a.hub_n1);
            main_kernel(std::false_type{}, blocks, a.rows_by_degree, 0, a.hub_threshold);
            if (wide_stream != a.stream) {
                (void)hipEventRecord(ss->join, ss->stream);
                (void)hipStreamWaitEvent(a.stream, ss->join, 0);
            }
            return;
        }
    }
    main_kernel(std::false_type{}, blocks, nullptr, 0, kNoLongRows);
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

    // Lanes own VEC consecutive features; global dwordx2 / dwordx4 accesses only need dword alignment on gfx950 and
    // a width that is not a multiple of VEC is covered by an overlapping last window, so any row of >= 4 floats is
    // read 16 B per lane (F = 7: 2 lanes per row, 32 rows per gather instruction instead of 8).
    int vec = a.F_active >= 4 ? 4 : a.F_active >= 2 ? 2 : 1;

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
                                const int32_t *column_indices, const int32_t *node_ids,
                                const int32_t *rows_by_degree, int32_t N, int64_t E, int32_t F, int32_t F_active,
                                void *stream)
{
    stg::GcnArgs a{x, norm_row, norm_col_edge, ew_edge, out, row_offsets, column_indices, nullptr,
                   node_ids, N, F, F_active, true, static_cast<hipStream_t>(stream), E};
    a.rows_by_degree = rows_by_degree;
    return stg::gcn_agg_dispatch(a, "stg_gcn_agg_edge");
}

extern "C" int stg_gcn_layer_fwd(const float *x, const float *norm_row, const float *norm_col_edge,
                                 const float *ew_edge, const float *bias, int32_t act, float *out,
                                 const int32_t *row_offsets, const int32_t *column_indices,
                                 const int32_t *node_ids, const int32_t *rows_by_degree, int32_t N, int64_t E,
                                 int32_t F, void *stream)
{
    stg::GcnArgs a{x, norm_row, norm_col_edge, ew_edge, out, row_offsets, column_indices, nullptr,
                   node_ids, N, F, F, true, static_cast<hipStream_t>(stream), E};
    a.rows_by_degree = rows_by_degree;
    a.bias = bias;
    a.act = act;
    return stg::gcn_agg_dispatch(a, "stg_gcn_layer_fwd");
}

extern "C" int stg_gcn_agg_edge2(const float *x, const float *norm_row, const float *norm_col_edge, const float *ew_edge,
                                 const float *bias, int32_t act, float *out, const int32_t *row_offsets,
                                 const int32_t *column_indices, const int32_t *node_ids, const int32_t *rows_by_degree,
                                 int32_t N, int64_t E, int32_t F, int32_t F_active, int32_t hub_threshold, int32_t hub_rows_16,
                                 int32_t hub_rows_4, int32_t hub_rows_1, void *stream)
{
    if (hub_threshold < 0 || hub_rows_16 < 0 || hub_rows_4 < 0 || hub_rows_1 < 0 ||
        (int64_t)hub_rows_16 + hub_rows_4 + hub_rows_1 > N)
        return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gcn_agg_edge2: bad hub plan");
    stg::GcnArgs a{x, norm_row, norm_col_edge, ew_edge, out, row_offsets, column_indices, nullptr,
                   node_ids, N, F, F_active, true, static_cast<hipStream_t>(stream), E};
    a.rows_by_degree = rows_by_degree;
    a.bias = bias;
    a.act = act;
    a.hub_threshold = hub_threshold; a.hub_n16 = hub_rows_16; a.hub_n4 = hub_rows_4; a.hub_n1 = hub_rows_1;
    return stg::gcn_agg_dispatch(a, "stg_gcn_agg_edge2");
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
