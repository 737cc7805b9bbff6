// Forward launch of the one-launch TGCN step on the matrix cores (3-term bf16 split): design notes in tgcn_stepx.hpp.
// Same inputs, outputs and saved tensors as tgcn_step_fwd.hip (reference: nn/pytorch/temporal/tgcn.py:21-55 under
// benchmarking/static-temporal-tgcn/seastar/model.py:6-18 and dynamic-temporal-tgcn/seastar/model.py:5-21); the products agree
// with the fp32 form to fp32 rounding (tests/test_gpu_tgcn_step.py: 1e-5 against fp64), P bit for bit.
#include "tgcn_stepx.hpp"

namespace stg {
namespace {

struct FwdXArgs {
    const int *row_offsets, *column_indices;
    const float *nc_edge, *ew_edge, *norm;
    const float *x, *H, *target;
    const char *img;                                   // stg_tgcn_pack_weights_x3's forward image
    float *P, *x3, *Z, *R, *Ht, *Hn, *HR, *y, *y_out, *partial;
    unsigned char *mask;                               // [N][3 gates][4 kq][4 ct] one byte each (low nibble: the piece's 4 columns)
    int64_t N;
    float lo, hi;
    int num_tiles;
};

constexpr int kPLd = 36;                                               // Pbuf row stride in floats
constexpr int kActImg = 2 * kXTerms * kFragBytes;                      // a 64-column activation as fragments: 2 K-blocks = 6144 B
constexpr int kTeamBytes = 16 * kPLd * 4 + 3 * kActImg + 3 * kActImg + kGatherTableBytes;  // Pbuf | hz hr hh | H, HR, relu(Hn) | edge records
constexpr int kFwdLdsCat = 0;
constexpr int kFwdLdsHead = kFwdLdsCat + 4 * kFwdCatFrags * kFragBytes;
constexpr int kFwdLdsBias = kFwdLdsHead + 2 * kFwdHeadFrags * kFragBytes;
constexpr int kFwdLdsTeam = kFwdLdsBias + 2048;
constexpr int kFwdLds = kFwdLdsTeam + 2 * kTeamBytes;
static_assert(4 * kFwdBiasFloats <= 2048 && kFwdLds <= 160 * 1024, "LDS budget");

template <bool HAS_EW, int HEAD>
__global__ __launch_bounds__(512) void tgcn_stepx_fwd_kernel(const FwdXArgs a)
{
    constexpr int C = kXC, FIN = kXFin, FH = kXFh;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int team = wave >> 2, ct = wave & 3;
    const int n16 = lane & 15, kq = lane >> 4;
    char *const sCat = lds + kFwdLdsCat, *const sHead = lds + kFwdLdsHead;
    const float *const sBias = reinterpret_cast<const float *>(lds + kFwdLdsBias);
    char *const tm = lds + kFwdLdsTeam + team * kTeamBytes;
    float *const Pbuf = reinterpret_cast<float *>(tm);
    char *const sFhg = tm + 16 * kPLd * 4;                                  // hz | hr | hh, kActImg each
    char *const sFH = sFhg + 3 * kActImg, *const sFHR = sFH + kActImg, *const sFHn = sFHR + kActImg;
    // the 32 edge records of this lane's gather row (row 4 ct + rl of the tile; the 16 lanes of a row share them)
    uint4 *const trow = reinterpret_cast<uint4 *>(sFHn + kActImg) + (4 * ct + (lane >> 4)) * kGatherTableEdges;

    // ---- the small weights and the biases into LDS; this wave's rows of the gate Linears into registers -------------------
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.img + kFwdImgCat);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        constexpr int n16b = (kFwdImgBias - kFwdImgCat) / 16;               // Wcat + W1 sections: contiguous in image and LDS
        static_assert(n16b % 512 == 0, "whole rounds of the workgroup");
        uint4 v[n16b / 512];                                                // every load in flight before the first LDS store
#pragma unroll
        for (int k = 0; k < n16b / 512; ++k) v[k] = src[threadIdx.x + 512 * k];
#pragma unroll
        for (int k = 0; k < n16b / 512; ++k) dst[threadIdx.x + 512 * k] = v[k];
        const float *bsrc = reinterpret_cast<const float *>(a.img + kFwdImgBias);
        float *bdst = reinterpret_cast<float *>(lds + kFwdLdsBias);
        for (int i = threadIdx.x; i < kFwdBiasFloats; i += 512) bdst[i] = bsrc[i];
    }
    Frag3 Wg[3][4];                                                         // [gate][K-block]: 36 fragments, 144 registers
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int b = 0; b < 4; ++b) Wg[g][b] = wfrag_load(a.img + kFwdImgGate, ct * kFwdGateFrags + (g * 4 + b) * kXTerms, lane);

    // tiles of this team: (2 k + team) * grid + block, k = 0, 1, ...
    const int G = (int)gridDim.x, first = team * G + (int)blockIdx.x;
    const int n_mine = first < a.num_tiles ? (a.num_tiles - first + 2 * G - 1) / (2 * G) : 0;
    const int n_other = ((1 - team) * G + (int)blockIdx.x) < a.num_tiles
                            ? (a.num_tiles - ((1 - team) * G + (int)blockIdx.x) + 2 * G - 1) / (2 * G) : 0;
    const int n0 = team == 0 ? n_mine : n_other, n1 = team == 0 ? n_other : n_mine;
    // steps of a team: 0 = first gather, then 4 per tile; team 1 runs two intervals behind team 0
    const int total = max(n0 ? 1 + 4 * n0 : 0, n1 ? 3 + 4 * n1 : 0);
    const float lo = a.lo, hi = a.hi;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 h4 = zero4, z4 = zero4;                                          // own pieces of H and Z, carried across intervals

    // The gather of a tile's P rows (four rows per wave) in its steps: see RowGatherX.  `gather_finish` ends it: P to global
    // memory and to Pbuf, and this lane's piece of the tile's H (split into fragments by the tile's first phase).
    RowGatherX<HAS_EW> rg;
    const int c2 = lane & 15;
    auto grow = [&](int t) { return (int)min((int64_t)t * 16 + 4 * ct + (lane >> 4), a.N - 1); };
    auto gather_finish = [&](int t) -> float4 {
        const int row = grow(t);
        const float2 p = rg.run(trow, a.x, a.column_indices, a.nc_edge, a.ew_edge, c2);
        *reinterpret_cast<float2 *>(reinterpret_cast<char *>(a.P) + ((size_t)(unsigned)row * (FIN * 4u) + 8u * c2)) = p;
        *reinterpret_cast<float2 *>(Pbuf + (4 * ct + (lane >> 4)) * kPLd + 2 * c2) = p;
        const unsigned hrow = (unsigned)min((int64_t)t * 16 + n16, a.N - 1);
        return a.H ? ld_f4(a.H, (hrow * C + 16u * ct + 4u * kq) * 4u, 0) : zero4;
    };

    // The weight fragments are used in every interval: one wait for them HERE (an asm use the compiler must honour), so that no
    // interval starts by waiting for "maybe still pending" loads with vmcnt(0).
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < kXTerms; ++k) asm volatile("" ::"v"(Wg[g][b].t[k]));
    __syncthreads();
    // The intervals of a team in program order -- [first gather] then per tile [x3] [gates z, r] [gate h] [head + next gather] --
    // as STRAIGHT-LINE code per tile (no phase switch inside the loop): the compiler's s_waitcnt insertion counts exactly inside an
    // iteration, while a wait that has to look across a loop back edge becomes vmcnt(0) -- a full drain of every load and store in
    // flight at the head of every phase (measured: 2 - 3 us per interval).  Team 1 runs two intervals behind team 0, so that one
    // team's matrix phase sits beside the other's gather / store phase; every wave of the workgroup passes `total` barriers.
    int itc = 0;                                                                // barriers passed (= trace interval)
    auto interval_end = [&]() {
        STGX_MARK(2 * itc + 1);
        lds_barrier();
        ++itc;
        STGX_MARK(2 * itc);
    };
    STGX_MARK(0);
    for (int k = 0; k < 2 * team; ++k) interval_end();
    if (n_mine > 0) {
        rg.extent(a.row_offsets, a.norm, grow(first));                       // the first tile's gather: its steps back to back
        rg.indices(a.column_indices, a.nc_edge, a.ew_edge, c2);
        rg.stash(trow, c2);
        wave_lds_fence();                                                    // the table rows are this wave's own
        h4 = gather_finish(first);
    }
    interval_end();
    for (int j = 0; j < n_mine; ++j) {
        const int tile = first + j * 2 * G;
        const int64_t idx = (int64_t)tile * 16 + n16;
        const unsigned row = (unsigned)min(idx, a.N - 1);
        const unsigned oC = (row * C + 16u * ct + 4u * kq) * 4u;              // this lane's piece in a row of C floats
        const bool more = j + 1 < n_mine;                                   // uniform per team
        // Order inside an interval: the matrix work, then whatever consumes an earlier interval's loads, the interval's own global
        // stores, and the loads for later intervals last.  sched_barrier pins the sections.
        {
            // ---- x3 = P Wcat + b3 (this wave's 16 columns of each gate), clamp, mask; hg as fragments -------------
            const float *pb = Pbuf + n16 * kPLd + 4 * kq;
            const Frag3 fp = frag_of(*reinterpret_cast<const float4 *>(pb), *reinterpret_cast<const float4 *>(pb + 16));
            float4 x3v[3];
            unsigned mk[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                mfma6(acc, wfrag_load(sCat, (ct * 3 + g) * kXTerms, lane), fp);
                const float4 b = *reinterpret_cast<const float4 *>(sBias + g * C + 16 * ct + 4 * kq);
                const float4 v = make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
                const float4 hg = make_float4(clamp3(v.x, lo, hi), clamp3(v.y, lo, hi), clamp3(v.z, lo, hi), clamp3(v.w, lo, hi));
                mk[g] = (hg.x == v.x ? 1u : 0u) | (hg.y == v.y ? 2u : 0u) | (hg.z == v.z ? 4u : 0u) | (hg.w == v.w ? 8u : 0u);
                x3v[g] = v;
                frag_store_piece(sFhg + g * kActImg, ct, lane, split4(hg));
            }
            __builtin_amdgcn_sched_barrier(0);
            frag_store_piece(sFH, ct, lane, split4(h4));                    // this tile's H piece (loaded an interval ago)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 3; ++g) st_f4(a.x3, (row * (3u * C) + 4u * kq) * 4u, 4 * (g * C + 16 * ct), x3v[g]);
            if (a.mask) {
#pragma unroll
                for (int g = 0; g < 3; ++g) a.mask[(size_t)row * 48u + (4u * g + kq) * 4u + ct] = (unsigned char)mk[g];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) rg.extent(a.row_offsets, a.norm, grow(tile + 2 * G));    // next tile's gather, first round trip: issued LAST
        }
        interval_end();
        {
            // ---- Z = sigmoid([hz | H] Wz^T + bz),  R = sigmoid([hr | H] Wr^T + br) ------------------------------------
            f32x4 az = to_x4(*reinterpret_cast<const float4 *>(sBias + 3 * C + 16 * ct + 4 * kq));
            f32x4 ar = to_x4(*reinterpret_cast<const float4 *>(sBias + 4 * C + 16 * ct + 4 * kq));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                mfma6(az, Wg[0][b], frag_load(sFhg, b, lane));
                mfma6(ar, Wg[1][b], frag_load(sFhg + kActImg, b, lane));
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const Frag3 fh = frag_load(sFH, b, lane);
                mfma6(az, Wg[0][2 + b], fh);
                mfma6(ar, Wg[1][2 + b], fh);
            }
            z4 = make_float4(sigmoid_(az[0]), sigmoid_(az[1]), sigmoid_(az[2]), sigmoid_(az[3]));
            const float4 r = make_float4(sigmoid_(ar[0]), sigmoid_(ar[1]), sigmoid_(ar[2]), sigmoid_(ar[3]));
            const float4 hr = make_float4(h4.x * r.x, h4.y * r.y, h4.z * r.z, h4.w * r.w);
            frag_store_piece(sFHR, ct, lane, split4(hr));
            __builtin_amdgcn_sched_barrier(0);
            st_f4(a.Z, oC, 0, z4);
            st_f4(a.R, oC, 0, r);
            st_f4(a.HR, oC, 0, hr);
            __builtin_amdgcn_sched_barrier(0);
            if (more) rg.indices(a.column_indices, a.nc_edge, a.ew_edge, c2);   // second round trip (needs the extent: an interval old), LAST
        }
        interval_end();
        {
            // ---- Ht = tanh([hh | H*R] Wh^T + bh);  Hn = Z*H + (1 - Z)*Ht -----------------------------------------------
            f32x4 ah = to_x4(*reinterpret_cast<const float4 *>(sBias + 5 * C + 16 * ct + 4 * kq));
#pragma unroll
            for (int b = 0; b < 2; ++b) mfma6(ah, Wg[2][b], frag_load(sFhg + 2 * kActImg, b, lane));
#pragma unroll
            for (int b = 0; b < 2; ++b) mfma6(ah, Wg[2][2 + b], frag_load(sFHR, b, lane));
            const float4 t = make_float4(tanh_(ah[0]), tanh_(ah[1]), tanh_(ah[2]), tanh_(ah[3]));
            const float4 hn = make_float4(z4.x * h4.x + (1.0f - z4.x) * t.x, z4.y * h4.y + (1.0f - z4.y) * t.y,
                                          z4.z * h4.z + (1.0f - z4.z) * t.z, z4.w * h4.w + (1.0f - z4.w) * t.w);
            frag_store_piece(sFHn, ct, lane, split4(make_float4(hn.x < 0.f ? 0.f : hn.x, hn.y < 0.f ? 0.f : hn.y,
                                                               hn.z < 0.f ? 0.f : hn.z, hn.w < 0.f ? 0.f : hn.w)));
            __builtin_amdgcn_sched_barrier(0);
            if (more) rg.stash(trow, c2);                                       // the edge records (an interval old) into the table
            __builtin_amdgcn_sched_barrier(0);
            st_f4(a.Ht, oC, 0, t);
            st_f4(a.Hn, oC, 0, hn);
        }
        interval_end();
        {
            // ---- head of this tile (one wave of the team, in turn), the next tile's gather, then the stores and the H load ----
            const bool head_wave = HEAD != 0 && ct == (j & 3);
            f32x4 ay[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            float yo = 0.f, tg = 0.f;
            if (head_wave) {
                if constexpr (HEAD == 2) tg = ld_f1(a.target, row * 4u);       // consumed after the gather
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) ay[ft] = to_x4(*reinterpret_cast<const float4 *>(sBias + 6 * C + 16 * ft + 4 * kq));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const Frag3 f = frag_load(sFHn, b, lane);
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) mfma6(ay[ft], wfrag_load(sHead, (ft * 2 + b) * kXTerms, lane), f);
                }
                if constexpr (HEAD == 2) {
                    float sdot = 0.f;
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        const float4 w2 = *reinterpret_cast<const float4 *>(sBias + 6 * C + FH + 16 * ft + 4 * kq);
                        sdot = sdot + ay[ft][0] * w2.x;
                        sdot = sdot + ay[ft][1] * w2.y;
                        sdot = sdot + ay[ft][2] * w2.z;
                        sdot = sdot + ay[ft][3] * w2.w;
                    }
                    sdot = sdot + __shfl_xor(sdot, 16, kWave);               // the row's four kq lanes
                    sdot = sdot + __shfl_xor(sdot, 32, kWave);
                    yo = sdot + sBias[6 * C + 2 * FH];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 hnext = zero4;
            if (more) hnext = gather_finish(tile + 2 * G);                   // third round trip: the neighbour rows; P; the H load
            __builtin_amdgcn_sched_barrier(0);
            if (head_wave) {
                const unsigned oF = (row * FH + 4u * kq) * 4u;
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) st_f4(a.y, oF, 64 * ft, to_f4(ay[ft]));
                if constexpr (HEAD == 2) {
                    if (kq == 0) st_f1(a.y_out, row * 4u, yo);
                    const float dlt = yo - tg;
                    float sq = (idx < a.N && kq == 0) ? dlt * dlt : 0.f;
                    sq = row16_sum(sq);                                   // lanes 0..15: the tile's 16 rows, in lane order
                    if (lane == 15) a.partial[tile] = sq;
                }
            }
            if (more) h4 = hnext;
        }
        interval_end();
    }
    while (itc < total) interval_end();
}

template <bool HAS_EW, int HEAD>
int launch_stepx_fwd(const FwdXArgs &a, hipStream_t stream)
{
    auto kern = tgcn_stepx_fwd_kernel<HAS_EW, HEAD>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (!*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_step_fwd (matrix-core form): %s", hipGetErrorString(e));
        *raised = true;
    }
    // one workgroup (two teams) per CU; fewer when there are fewer than two tiles per workgroup
    const unsigned blocks = (unsigned)std::max(1, std::min(256, (a.num_tiles + 1) / 2));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), kFwdLds, stream, a);
    return check_launch("stg_tgcn_step_fwd (matrix-core form)");
}

// ---- the weight images ---------------------------------------------------------------------------------------------------------
struct PackXArgs {
    const float *Wc[3], *bc[3], *Wg[3], *bg[3], *W1, *b1, *W2, *b2;
    char *fwd, *bwd;
};

__device__ __forceinline__ void split1(float v, unsigned short (&t)[kXTerms])
{
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) {
        const unsigned p = pk_bf16(v, 0.f);
        t[k] = (unsigned short)(p & 0xffffu);
        v = v - bf16_lo(p);
    }
}

// one thread per (fragment without its term index, lane): eight source weights -> three 16-byte entries
__global__ __launch_bounds__(kBlock) void tgcn_pack_weights_x3_kernel(const PackXArgs a)
{
    constexpr int C = kXC;
    // fragment groups (each = kXTerms consecutive fragments of one image)
    constexpr int nFg = 4 * 3 * 4, nFc = 4 * 3, nFh = 2 * 2;               // forward: gate, Wcat, head
    constexpr int nBg = 4 * 3 * 2 * 2, nBh = 4, nBc = 2 * 6;              // backward: gate, W1T, Wcat
    constexpr int nGroups = nFg + nFc + nFh + nBg + nBh + nBc;
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int grp = gid >> 6, lane = gid & 63;
    if (grp < nGroups) {
        const int m16 = lane & 15, kq = lane >> 4;
        float v[8];
        char *dst;
        int q = grp;
        if (q < nFg) {                                  // (ct, g, b): W_g[16 ct + m16][xcol(b, kq, i)]
            const int ctv = q / 12, g = (q % 12) / 4, b = q % 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.Wg[g][(16 * ctv + m16) * 2 * C + xcol(b, kq, i)];
            dst = a.fwd + kFwdImgGate + (size_t)(q * kXTerms) * kFragBytes;
        } else if ((q -= nFg) < nFc) {                  // (ct, g): Wc_g[f = xcol(0, kq, i)][16 ct + m16]
            const int ctv = q / 3, g = q % 3;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.Wc[g][xcol(0, kq, i) * C + 16 * ctv + m16];
            dst = a.fwd + kFwdImgCat + (size_t)(q * kXTerms) * kFragBytes;
        } else if ((q -= nFc) < nFh) {                  // (ct', b): W1[16 ct' + m16][xcol(b, kq, i)]
            const int ft = q / 2, b = q % 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.W1[(16 * ft + m16) * C + xcol(b, kq, i)];
            dst = a.fwd + kFwdImgHead + (size_t)(q * kXTerms) * kFragBytes;
        } else if ((q -= nFh) < nBg) {                  // (ct, g, half, b): W_g[c = xcol(b, kq, i)][half C + 16 ct + m16]
            const int ctv = q / 12, g = (q % 12) / 4, half = (q % 4) / 2, b = q % 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.Wg[g][xcol(b, kq, i) * 2 * C + half * C + 16 * ctv + m16];
            dst = a.bwd + kBwdImgGate + (size_t)(q * kXTerms) * kFragBytes;
        } else if ((q -= nBg) < nBh) {                  // (ct): W1[f = xcol(0, kq, i)][16 ct + m16]
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.W1[xcol(0, kq, i) * C + 16 * q + m16];
            dst = a.bwd + kBwdImgHead + (size_t)(q * kXTerms) * kFragBytes;
        } else {                                        // (ct', b): Wcat[f = 16 ct' + m16][c = xcol(b, kq, i)], c = g C + c'
            q -= nBh;
            const int ft = q / 6, b = q % 6;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = xcol(b, kq, i);
                v[i] = a.Wc[c / C][(16 * ft + m16) * C + (c % C)];
            }
            dst = a.bwd + kBwdImgCat + (size_t)(q * kXTerms) * kFragBytes;
        }
        unsigned short t[8][kXTerms];
#pragma unroll
        for (int i = 0; i < 8; ++i) split1(v[i], t[i]);
#pragma unroll
        for (int k = 0; k < kXTerms; ++k) {
            uint4 o;
            o.x = (unsigned)t[0][k] | ((unsigned)t[1][k] << 16);
            o.y = (unsigned)t[2][k] | ((unsigned)t[3][k] << 16);
            o.z = (unsigned)t[4][k] | ((unsigned)t[5][k] << 16);
            o.w = (unsigned)t[6][k] | ((unsigned)t[7][k] << 16);
            *reinterpret_cast<uint4 *>(dst + (size_t)k * kFragBytes + lane * 16) = o;
        }
    }
    // the fp32 tails: forward b3 | bz br bh | b1 | W2 | b2, backward W2
    if (gid < kFwdBiasFloats) {
        float *f = reinterpret_cast<float *>(a.fwd + kFwdImgBias);
        float v = 0.f;
        if (gid < 3 * C) v = a.bc[gid / C][gid % C];
        else if (gid < 6 * C) v = a.bg[(gid - 3 * C) / C][gid % C];
        else if (gid < 6 * C + kXFh) v = a.b1[gid - 6 * C];
        else if (gid < 6 * C + 2 * kXFh) v = a.W2 ? a.W2[gid - 6 * C - kXFh] : 0.f;
        else if (gid == 6 * C + 2 * kXFh) v = a.b2 ? a.b2[0] : 0.f;
        f[gid] = v;
    }
    if (gid < kBwdBiasFloats) reinterpret_cast<float *>(a.bwd + kBwdImgBias)[gid] = a.W2 ? a.W2[gid] : 0.f;
}

}  // namespace
}  // namespace stg

extern "C" size_t stg_tgcn_step_image_bytes(int32_t backward) { return backward ? (size_t)stg::kBwdImgBytes : (size_t)stg::kFwdImgBytes; }

extern "C" int stg_tgcn_pack_weights_x3(const float *Wcz, const float *Wcr, const float *Wch, const float *bcz, const float *bcr,
                                        const float *bch, const float *Wz, const float *bz, const float *Wr, const float *br,
                                        const float *Wh, const float *bh, const float *W1, const float *b1, const float *W2,
                                        const float *b2, void *fwd_image, void *bwd_image, int32_t C, int32_t Fin, int32_t Fh, void *stream)
{
    using namespace stg;
    if (C != kXC || Fin != kXFin || Fh != kXFh)
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_pack_weights_x3: C=%d Fin=%d Fh=%d not supported (64 / 32 / 32)", C, Fin, Fh);
    if (!Wcz || !Wcr || !Wch || !bcz || !bcr || !bch || !Wz || !bz || !Wr || !br || !Wh || !bh || !W1 || !b1 || !fwd_image || !bwd_image)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_pack_weights_x3: NULL pointer argument (only W2 / b2 may be NULL)");
    if ((reinterpret_cast<uintptr_t>(fwd_image) | reinterpret_cast<uintptr_t>(bwd_image)) & 15)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_pack_weights_x3: the images must be 16-byte aligned");
    PackXArgs a{{Wcz, Wcr, Wch}, {bcz, bcr, bch}, {Wz, Wr, Wh}, {bz, br, bh}, W1, b1, W2, b2,
                static_cast<char *>(fwd_image), static_cast<char *>(bwd_image)};
    constexpr int groups = 4 * 3 * 4 + 4 * 3 + 2 * 2 + 4 * 3 * 2 * 2 + 4 + 2 * 6;
    constexpr int threads = groups * 64 > kFwdBiasFloats ? groups * 64 : kFwdBiasFloats;
    hipLaunchKernelGGL(tgcn_pack_weights_x3_kernel, dim3((threads + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("stg_tgcn_pack_weights_x3");
}

#ifdef STG_STEPX_TRACE
extern "C" int stg_debug_set_stepx_trace_fwd(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(stg::g_stepx_trace), &buf, sizeof(buf)); }
#endif

// dispatch target of stg_tgcn_step_fwd (tgcn_step_fwd.hip) when the argument block carries a weight image
int stg_tgcn_stepx_fwd_launch(const stg_tgcn_step_fwd_args *p, void *stream_)
{
    using namespace stg;
    FwdXArgs a{};
    a.row_offsets = p->row_offsets; a.column_indices = p->column_indices;
    a.nc_edge = p->norm_col_edge; a.ew_edge = p->ew_edge; a.norm = p->norm;
    a.x = p->x; a.H = p->H; a.target = p->target;
    a.img = static_cast<const char *>(p->w_image);
    a.P = p->P; a.x3 = p->x3; a.Z = p->Z; a.R = p->R; a.Ht = p->Ht; a.Hn = p->Hn; a.HR = p->HR; a.y = p->y;
    a.y_out = p->y_out; a.partial = p->loss_partial; a.mask = reinterpret_cast<unsigned char *>(p->clamp_mask);
    a.N = p->N; a.lo = p->lo; a.hi = p->hi; a.num_tiles = (int)((p->N + 15) / 16);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (p->ew_edge) return p->head == 1 ? launch_stepx_fwd<true, 1>(a, st) : launch_stepx_fwd<true, 2>(a, st);
    return p->head == 1 ? launch_stepx_fwd<false, 1>(a, st) : launch_stepx_fwd<false, 2>(a, st);
}
