// Backward launch of the one-launch-per-step TGCN kernels: layout and design notes in tgcn_step.hpp.
#include "tgcn_step.hpp"

namespace stg {
namespace {

// ------------------------------------------------------------------------------------------------------ backward
struct BwdArgs {
    const int *row_offsets, *column_indices, *node_ids;        // BACKWARD CSR (rows = sources)
    const float *nc_edge, *ew_edge, *norm;
    const float *zn, *gy, *dHn, *g_cost;
    const float *Z, *R, *Ht, *H, *Hn, *x3, *y_out, *target;
    const float *WzT, *WrT, *WhT, *Wcat, *W1T, *W2;
    const float *At;                                           // FOLD: [3 Fin][C], rows g Fin + f = the folded gate weights' P part, transposed
    float *dzl, *drl, *dhl, *da3, *dH, *z, *dyt, *dyo;
    int d_wide;                    // dzl / drl / dhl are column blocks of ONE [N, 3C] matrix (rows 3C floats apart)
    const unsigned *mask;
    const int *link_row_ptr, *link_other, *link_eid;           // node side of the link loss's backward (head == 1; NULL: not here)
    const float *link_y, *link_logits, *link_target;
    float link_inv_m;
    int64_t N;
    float lo, hi, two_over_n;
    int num_tiles;
    int no_coop;                                               // knob "step_coop" 1
};

// FOLD: the input gradient z = da3 Wcat^T with da3_g = d_g Wg[:, :C] is  sum_g d_g At_g^T  with At_g [Fin][C] = (the folded gate
// weights' P part)^T (stg_tgcn_fold_weights): three products of K = C onto Fin columns (96 matrix instructions per tile) instead of
// three onto C columns + da3 Wcat^T (288); da3 is not formed and the clamp mask not read (an inactive clamp is the fold's premise:
// the forward launch raises the status word otherwise).  LDS holds the gate Linears' H halves only.
template <int C, int FIN, int FH, int WAVES, bool GATHER, int HEAD, bool FOLD = false>
struct BwdShape {
    static constexpr int K2 = 2 * C, LDB = C + 8, LDX = 3 * C + 8, LDT = FH + 8;   // row strides = 8 mod 16 dwords: see tgcn_step.hpp
    static constexpr int kGate = FOLD ? 3 * C * LDB : 3 * K2 * LDB;   // WzT | WrT | WhT, each [2C][LDB] (FOLD: rows C .. 2C - 1 of each)
    static constexpr int kCat = FOLD ? 3 * FIN * LDB : FIN * LDX;     // Wcat [FIN][LDX] (FOLD: At [3 FIN][LDB])
    static constexpr int kHead = HEAD ? C * LDT : 0;          // W1T [C][LDT]
    static constexpr int kBias = FH + 4;                      // W2
    // shared tiles of the partial round (tgcn_step_fwd.hip): per group of four waves three [16][LDXB] exchange buffers (d_h, d_z, d_r)
    static constexpr int GROUPS = WAVES / 8, LDXB = C + 8;
    static constexpr int kCoop = (FOLD && GATHER && HEAD == 2) ? GROUPS * 3 * 16 * LDXB + 8 : 0;
    static constexpr int kFloats = kGate + kCat + kHead + kBias + kCoop;
    static constexpr size_t kLds = sizeof(float) * (size_t)kFloats;
    static_assert(kLds <= 160 * 1024, "the weights must fit one CU's LDS");
};

template <int C, int FIN, int FH, int WAVES, bool GATHER, bool HAS_EW, int HEAD, bool FOLD = false>
__global__ __launch_bounds__(WAVES *kWave) void tgcn_step_bwd_kernel(const BwdArgs a)
{
    using S = BwdShape<C, FIN, FH, WAVES, GATHER, HEAD, FOLD>;
    constexpr int NT = WAVES * kWave, PC = C / 16, PF = FIN / 16, PH = FH / 16;
    constexpr int LDB = S::LDB, LDX = S::LDX, LDT = S::LDT;
    static_assert(FIN == 32 && FH == FIN, "the head's output is the next step's input");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *WgT = lds;                                  // [3][2C][LDB]
    float *Wc = WgT + S::kGate;                        // [FIN][LDX]
    float *W1T = Wc + S::kCat;                         // [C][LDT]
    float *bs = W1T + S::kHead;                        // W2 [FH]
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;

    // tile hand-out and staging: see tgcn_step_fwd.hip
    int *const next_w = reinterpret_cast<int *>(bs + FH);
    STG_TRACE_MARK(0);
    STG_TRACE_MARK(14);
    if (threadIdx.x == 0) *next_w = WAVES;
    // hand-out order, full rounds and the partial round shared four waves to a tile: see tgcn_step_fwd.hip
    const int grid = (int)gridDim.x, blk = (int)blockIdx.x;
    const int full_rounds = a.num_tiles / (WAVES * grid);
    const int rem = a.num_tiles - full_rounds * WAVES * grid;
    const int cnt_b = rem > blk ? (rem - blk - 1) / grid + 1 : 0;
    constexpr bool kCanCoop = FOLD && GATHER && HEAD == 2;
    const bool coop = kCanCoop && full_rounds >= 1 && cnt_b > 0 && 8 * cnt_b <= WAVES && !a.node_ids && !a.no_coop && a.z != nullptr;
    const int seq_end = full_rounds * WAVES + (coop ? 0 : cnt_b);
    int seq = wave;
    int *const coop_cnt = reinterpret_cast<int *>(bs + S::kBias);
    float *const xbuf = bs + S::kBias + 8;
    if constexpr (kCanCoop) {
        if (threadIdx.x < 8) coop_cnt[threadIdx.x] = 0;
    }
    int tile = seq * grid + blk;
    const bool want_z = a.z != nullptr;                // block-uniform
    constexpr int kStage4 = ((FOLD ? 3 * C * C : 3 * 2 * C * C) + FIN * 3 * C + (HEAD ? C * FH : 0)) / 4;
    constexpr int GR = FOLD ? C : 2 * C;                // rows of a gate's block in LDS (FOLD: its H half, rows C .. 2C - 1 of W_g^T)
    const StageSeg segs[5] = {{FOLD ? a.WzT + C * C : a.WzT, WgT, GR, C, LDB}, {FOLD ? a.WrT + C * C : a.WrT, WgT + GR * LDB, GR, C, LDB},
                              {FOLD ? a.WhT + C * C : a.WhT, WgT + 2 * GR * LDB, GR, C, LDB},
                              {FOLD ? a.At : a.Wcat, Wc, want_z ? (FOLD ? 3 * FIN : FIN) : 0, FOLD ? C : 3 * C, FOLD ? LDB : LDX},
                              {a.W1T, W1T, HEAD ? C : 0, FH, LDT}};
    Stager<NT, 5, (kStage4 + NT - 1) / NT> stager;
    const bool do_gather = GATHER && HEAD != 0 && a.zn != nullptr;      // block-uniform
    const int q = lane & 3, grow = lane >> 2;
    // Lanes past the last row MIRROR row N - 1 (gather, loads, arithmetic, stores): they recompute and rewrite that row's values
    // bit for bit, so no load or store of the tile body sits behind a per-lane guard -- a guarded store is a basic block of its
    // own, and the loads of the next phase could not be scheduled above it (round 2's form exposed ~12 round trips per tile).
    auto gather_row = [&](int t) {
        const int64_t gidx = std::min<int64_t>((int64_t)t * 16 + grow, a.N - 1);
        return a.node_ids ? a.node_ids[gidx] : (int)gidx;
    };
    stager.issue(segs);
    float sv = 0.f;                                    // W2: its load in flight with the staging loads (tgcn_step_fwd.hip)
    if constexpr (HEAD == 2) {
        if ((int)threadIdx.x < FH) sv = a.W2[threadIdx.x];
    }
    stager.commit(segs);
    if constexpr (HEAD == 2) {
        if ((int)threadIdx.x < FH) bs[threadIdx.x] = sv;
    }
    __syncthreads();
    STG_TRACE_MARK(1);

    const float lo = a.lo, hi = a.hi;
    // this lane's row / piece inside each LDS matrix (tgcn_step.hpp: pinned)
    const float *const wg_l = WgT + pinned((unsigned)(n16 * LDB + 4 * kq)), *const wcrow = Wc + pinned((unsigned)(n16 * (FOLD ? LDB : LDX) + 4 * kq));
    const float *const w1_l = W1T + pinned((unsigned)(n16 * LDT + 4 * kq)), *const bs_l = bs + pinned((unsigned)(4 * kq));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- the partial round first, four waves per tile, each one 16-column block of every product (tgcn_step_fwd.hip); what a later
    // product needs at full width -- d_h and d_z, then d_r -- crosses through LDS.  Same chains in the same order: bit-identical.
    if constexpr (kCanCoop) {
        if (coop && (wave >> 2) < cnt_b) {
            constexpr int LDXB = S::LDXB;
            const int grp = wave >> 2, cb = wave & 3;
            float *const xb_l = xbuf + grp * 3 * 16 * LDXB + pinned((unsigned)(n16 * LDXB + 4 * kq));
            volatile int *const cnt = coop_cnt + grp;
            int phase = 0;
            auto group_sync = [&]() {
                phase += 4;
                __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes before its counter bump
                if (lane == 0) atomicAdd(const_cast<int *>(cnt), 1);
                while (*cnt < phase) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_s_waitcnt(0xc07f);
            };
            for (int ct = grp; ct < cnt_b; ct += S::GROUPS) {
                const int ctile = (full_rounds * WAVES + ct) * grid + blk;
                STG_TRACE_MARK(10);
                float4 gp[PH];
#pragma unroll
                for (int j = 0; j < PH; ++j) gp[j] = zero4;
                if (do_gather) {
                    RowGather32<HAS_EW> rg;
                    rg.begin(a.row_offsets, a.norm, gather_row(ctile));
                    rg.indices(a.column_indices, a.nc_edge, a.ew_edge, 0, q);
                    float p8[8];
                    rg.run(p8, a.zn, a.column_indices, a.nc_edge, a.ew_edge, q);
                    gather_to_pieces(p8, gp, n16, kq);
                }
                const unsigned row = (unsigned)std::min<int64_t>((int64_t)ctile * 16 + n16, a.N - 1);
                const unsigned oC = (row * C + 4u * kq) * 4u + 64u * cb, o3 = (row * (3u * C) + 4u * kq) * 4u + 64u * cb;
                const unsigned oF = (row * FH + 4u * kq) * 4u, oD = a.d_wide ? o3 : oC;
                // this wave's column block of the row-local operands; the head's input at full width (every wave forms it)
                float4 gy[PH];
#pragma unroll
                for (int j = 0; j < PH; ++j) gy[j] = a.gy ? ld_f4(a.gy, oF, 64 * j) : zero4;
                const float yo = ld_f1(a.y_out, row * 4u), tg = ld_f1(a.target, row * 4u), gc = a.g_cost[0];
                const float4 hn = ld_f4(a.Hn, oC, 0), zz = ld_f4(a.Z, oC, 0), tt = ld_f4(a.Ht, oC, 0), rr = ld_f4(a.R, oC, 0);
                float4 dhn = a.dHn ? ld_f4(a.dHn, oC, 0) : zero4;
                const float4 hh = a.H ? ld_f4(a.H, oC, 0) : zero4;
#pragma unroll
                for (int j = 0; j < PH; ++j) gy[j] = make_float4(gy[j].x + gp[j].x, gy[j].y + gp[j].y, gy[j].z + gp[j].z, gy[j].w + gp[j].w);
                const float dyo = ((yo - tg) * a.two_over_n) * gc;
                if (cb == 0 && kq == 0) st_f1(a.dyo, row * 4u, dyo);
#pragma unroll
                for (int j = 0; j < PH; ++j) {
                    const float4 w2 = *reinterpret_cast<const float4 *>(bs_l + 16 * j);
                    gy[j] = make_float4(gy[j].x + dyo * w2.x, gy[j].y + dyo * w2.y, gy[j].z + dyo * w2.z, gy[j].w + dyo * w2.w);
                }
                if (cb == 0) {
#pragma unroll
                    for (int j = 0; j < PH; ++j) st_f4(a.dyt, oF, 64 * j, gy[j]);
                }
                f32x4 acc[1];
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_pieces<1, PH, false>(acc, w1_l + 16 * cb * LDT, LDT, [&](int j) { return gy[j]; });
                dhn.x = dhn.x + (hn.x > 0.f ? acc[0][0] : 0.f);
                dhn.y = dhn.y + (hn.y > 0.f ? acc[0][1] : 0.f);
                dhn.z = dhn.z + (hn.z > 0.f ? acc[0][2] : 0.f);
                dhn.w = dhn.w + (hn.w > 0.f ? acc[0][3] : 0.f);
                const float4 g = dhn, z = zz, t = tt, h = hh;
                const float4 dhl1 = make_float4((g.x * (1.0f - z.x)) * (1.0f - t.x * t.x), (g.y * (1.0f - z.y)) * (1.0f - t.y * t.y),
                                                (g.z * (1.0f - z.z)) * (1.0f - t.z * t.z), (g.w * (1.0f - z.w)) * (1.0f - t.w * t.w));
                const float4 dz1 = make_float4((g.x * (h.x - t.x)) * (z.x * (1.0f - z.x)), (g.y * (h.y - t.y)) * (z.y * (1.0f - z.y)),
                                               (g.z * (h.z - t.z)) * (z.z * (1.0f - z.z)), (g.w * (h.w - t.w)) * (z.w * (1.0f - z.w)));
                float4 dHa = make_float4(g.x * z.x, g.y * z.y, g.z * z.z, g.w * z.w);
                st_f4(a.dhl, oD, 0, dhl1);
                st_f4(a.dzl, oD, 0, dz1);
                *reinterpret_cast<float4 *>(xb_l + 16 * cb) = dhl1;
                *reinterpret_cast<float4 *>(xb_l + 16 * LDXB + 16 * cb) = dz1;
                group_sync();
                float4 dhl[PC], dzl[PC];
#pragma unroll
                for (int j = 0; j < PC; ++j) dhl[j] = *reinterpret_cast<const float4 *>(xb_l + 16 * j), dzl[j] = *reinterpret_cast<const float4 *>(xb_l + 16 * LDXB + 16 * j);
                // z (FIN = 2 column blocks: waves 0 and 1 of the group), gates in the tile loop's order h, z, r
                f32x4 zacc[1];
                zacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cb < PF) gemm_pieces<1, PC, false>(zacc, wcrow + (2 * FIN + 16 * cb) * LDB, LDB, [&](int j) { return dhl[j]; });
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_pieces<1, PC, false>(acc, wg_l + (2 * C + 16 * cb) * LDB, LDB, [&](int j) { return dhl[j]; });
                const float4 d = to_f4(acc[0]);
                const float4 drl1 = make_float4((d.x * h.x) * (rr.x * (1.0f - rr.x)), (d.y * h.y) * (rr.y * (1.0f - rr.y)),
                                                (d.z * h.z) * (rr.z * (1.0f - rr.z)), (d.w * h.w) * (rr.w * (1.0f - rr.w)));
                dHa = make_float4(dHa.x + d.x * rr.x, dHa.y + d.y * rr.y, dHa.z + d.z * rr.z, dHa.w + d.w * rr.w);
                st_f4(a.drl, oD, 0, drl1);
                *reinterpret_cast<float4 *>(xb_l + 2 * 16 * LDXB + 16 * cb) = drl1;
                if (cb < PF) gemm_pieces<1, PC, false>(zacc, wcrow + (0 * FIN + 16 * cb) * LDB, LDB, [&](int j) { return dzl[j]; });
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_pieces<1, PC, false>(acc, wg_l + (0 * C + 16 * cb) * LDB, LDB, [&](int j) { return dzl[j]; });
                dHa = make_float4(dHa.x + acc[0][0], dHa.y + acc[0][1], dHa.z + acc[0][2], dHa.w + acc[0][3]);
                group_sync();
                float4 drl[PC];
#pragma unroll
                for (int j = 0; j < PC; ++j) drl[j] = *reinterpret_cast<const float4 *>(xb_l + 2 * 16 * LDXB + 16 * j);
                if (cb < PF) gemm_pieces<1, PC, false>(zacc, wcrow + (1 * FIN + 16 * cb) * LDB, LDB, [&](int j) { return drl[j]; });
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_pieces<1, PC, false>(acc, wg_l + (1 * C + 16 * cb) * LDB, LDB, [&](int j) { return drl[j]; });
                st_f4(a.dH, oC, 0, make_float4(dHa.x + acc[0][0], dHa.y + acc[0][1], dHa.z + acc[0][2], dHa.w + acc[0][3]));
                if (cb < PF) st_f4(a.z, oF + 64u * cb, 0, to_f4(zacc[0]));
                STG_TRACE_MARK(11);
            }
        }
    }

    while (seq < seq_end) {
        // A_hat^T zn of the tile's rows (the next step's input gradient, aggregated here) as row pieces.  First: the gather loop
        // is the register-hungry part of the kernel and nothing else is live yet.
        float4 gp[PH];
#pragma unroll
        for (int j = 0; j < PH; ++j) gp[j] = zero4;
        if constexpr (GATHER && HEAD != 0) {
            if (do_gather) {
                RowGather32<HAS_EW> rg;
                rg.begin(a.row_offsets, a.norm, gather_row(tile));
                rg.indices(a.column_indices, a.nc_edge, a.ew_edge, 0, q);
                float p8[8];
                rg.run(p8, a.zn, a.column_indices, a.nc_edge, a.ew_edge, q);
                gather_to_pieces(p8, gp, n16, kq);
            }
        }
        STG_TRACE_MARK(2);
        // Element offsets are 32-bit (N 3C < 2^30, checked on the host): one VGPR per row stride next to scalar base pointers.
        unsigned row = (unsigned)std::min<int64_t>((int64_t)tile * 16 + n16, a.N - 1);
        if (a.node_ids) row = (unsigned)a.node_ids[row];
        // byte offsets of this lane's first piece in a row of C, 3C and FH floats (one VGPR each; columns are immediates)
        const unsigned oC = (row * C + 4u * kq) * 4u, o3 = (row * (3u * C) + 4u * kq) * 4u, oF = (row * FH + 4u * kq) * 4u;
        auto ldC = [&](const float *p, int j) { return ld_f4(p, oC, 64 * j); };          // p must not be NULL
        auto stC = [&](float *p, int j, const float4 &v) { st_f4(p, oC, 64 * j, v); };
        // the gate gradients: [N, C] each, or the column blocks of one [N, 3C] matrix (stg_tgcn_step_bwd_args::ld_d), so that
        // the window's weight gradients read d_z | d_r as ONE operand against [H | P]
        const unsigned oD = a.d_wide ? o3 : oC;
        auto ldD = [&](const float *p, int j) { return ld_f4(p, oD, 64 * j); };
        auto stD = [&](float *p, int j, const float4 &v) { st_f4(p, oD, 64 * j, v); };

        // clamp mask of the 3C columns of x3 as 48 bits (used by three later phases: 2 registers)
        unsigned mlo = 0u, mhi = 0u;
        unsigned m0 = 0u, m1 = 0u, m2 = 0u;
        if (!FOLD && !a.mask) {                                      // wave-uniform; the test-only form (no mask from the forward launch)
            float4 v[3 * PC];
#pragma unroll
            for (int c = 0; c < 3 * PC; ++c) v[c] = ld_f4(a.x3, o3, 64 * c);
#pragma unroll
            for (int c = 0; c < 3 * PC; ++c) {
                const unsigned b = (v[c].x >= lo && v[c].x <= hi ? 1u : 0u) | (v[c].y >= lo && v[c].y <= hi ? 2u : 0u) |
                                   (v[c].z >= lo && v[c].z <= hi ? 4u : 0u) | (v[c].w >= lo && v[c].w <= hi ? 8u : 0u);
                if (4 * c < 32) mlo |= b << ((4 * c) & 31);
                else mhi |= b << ((4 * c) & 31);
            }
        }
        // ---- every operand the tile reads before its first gate product, in flight at once (ONE round trip) -------------
        float4 dhn[PC], gy[PH], hn[PC], hh[PC], zz[PC], tt[PC];
        float yo = 0.f, tg = 0.f, gc = 0.f;
        {
            // word 4 g + kq of the row: 16 bits for gate g; packed below as bit 16 g + 4 blk + i (gates 0, 1 in mlo, 2 in mhi).
            // Loaded without a branch (from Z's rows, discarded, when there is no mask) so that the unpacking stays below the batch.
            if constexpr (!FOLD) {
                const unsigned *mbase = a.mask ? a.mask : reinterpret_cast<const unsigned *>(a.Z);
                const unsigned *mp = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(mbase) + (size_t)((row * 12u + kq) * 4u));
                m0 = mp[0], m1 = mp[4], m2 = mp[8];
            }
        }
#pragma unroll
        for (int j = 0; j < PC; ++j) dhn[j] = hh[j] = hn[j] = zero4;
#pragma unroll
        for (int j = 0; j < PH; ++j) gy[j] = zero4;
        if constexpr (HEAD != 0) {
            if (a.gy) {
#pragma unroll
                for (int j = 0; j < PH; ++j) gy[j] = ld_f4(a.gy, oF, 64 * j);
            }
            if constexpr (HEAD == 2) yo = ld_f1(a.y_out, row * 4u), tg = ld_f1(a.target, row * 4u), gc = a.g_cost[0];
#pragma unroll
            for (int j = 0; j < PC; ++j) hn[j] = ldC(a.Hn, j);
        }
        if (a.dHn) {                                                 // wave-uniform: one branch for the group of loads
#pragma unroll
            for (int j = 0; j < PC; ++j) dhn[j] = ldC(a.dHn, j);
        }
        if (a.H) {
#pragma unroll
            for (int j = 0; j < PC; ++j) hh[j] = ldC(a.H, j);
        }
#pragma unroll
        for (int j = 0; j < PC; ++j) zz[j] = ldC(a.Z, j), tt[j] = ldC(a.Ht, j);
        __builtin_amdgcn_sched_barrier(0);
        if (a.mask) mlo = (m0 & 0xffffu) | (m1 << 16), mhi = m2 & 0xffffu;

        // ---- gradient reaching Hn: from the next step (dHn) and through the head --------------------------------
        if constexpr (HEAD == 1) {
            // The node side of the link-prediction loss (stg_link_decode_bwd's sum, term for term and in its order): the label
            // edges incident to this lane's row, four at a time -- indices first, then logits / targets / the other ends' rows of y.
            if (a.link_row_ptr) {                                    // block-uniform
                const int kb = a.link_row_ptr[row], ke = a.link_row_ptr[row + 1];
                const float scale = a.g_cost[0] * a.link_inv_m;
                const int kmax = wave_max_nonneg(ke - kb);
                for (int k0 = 0; k0 < kmax; k0 += 4) {
                    int e[4], o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = kb + k0 + u < ke;
                        e[u] = ok ? a.link_eid[kb + k0 + u] : 0;
                        o[u] = ok ? a.link_other[kb + k0 + u] : 0;
                    }
                    float x[4], t[4];
                    float4 yo[4][PH];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[u] = a.link_logits[e[u]];
                        t[u] = a.link_target[e[u]];
#pragma unroll
                        for (int j = 0; j < PH; ++j) yo[u][j] = ld_f4(a.link_y, ((unsigned)o[u] * FH + 4u * kq) * 4u, 64 * j);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (kb + k0 + u < ke) {
                            const float sig = 1.0f / (1.0f + __expf(-x[u]));
                            const float coef = (sig - t[u]) * scale;
#pragma unroll
                            for (int j = 0; j < PH; ++j)
                                gy[j] = make_float4(gy[j].x + coef * yo[u][j].x, gy[j].y + coef * yo[u][j].y,
                                                    gy[j].z + coef * yo[u][j].z, gy[j].w + coef * yo[u][j].w);
                        }
                    }
                }
            }
        }
        if constexpr (HEAD != 0) {
#pragma unroll
            for (int j = 0; j < PH; ++j) gy[j] = make_float4(gy[j].x + gp[j].x, gy[j].y + gp[j].y, gy[j].z + gp[j].z, gy[j].w + gp[j].w);
            if constexpr (HEAD == 2) {
                const float dyo = ((yo - tg) * a.two_over_n) * gc;
                if (kq == 0) st_f1(a.dyo, row * 4u, dyo);
#pragma unroll
                for (int j = 0; j < PH; ++j) {
                    const float4 w2 = *reinterpret_cast<const float4 *>(bs_l + 16 * j);
                    gy[j] = make_float4(gy[j].x + dyo * w2.x, gy[j].y + dyo * w2.y, gy[j].z + dyo * w2.z, gy[j].w + dyo * w2.w);
                }
            }
#pragma unroll
            for (int j = 0; j < PH; ++j) st_f4(a.dyt, oF, 64 * j, gy[j]);
            f32x4 acc[PC];
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_pieces<PC, PH, false>(acc, w1_l, LDT, [&](int j) { return gy[j]; });
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const float4 h = hn[j];
                dhn[j].x = dhn[j].x + (h.x > 0.f ? acc[j][0] : 0.f);
                dhn[j].y = dhn[j].y + (h.y > 0.f ? acc[j][1] : 0.f);
                dhn[j].z = dhn[j].z + (h.z > 0.f ? acc[j][2] : 0.f);
                dhn[j].w = dhn[j].w + (h.w > 0.f ? acc[j][3] : 0.f);
            }
        }

        STG_TRACE_MARK(3);
        // ---- GRU update backward ---------------------------------------------------------------------------------
        // (dzl is stored here and read back by the same lane -- an L2 hit, issued a whole product ahead of its use -- instead of
        // living through the first two products, and H likewise: 32 registers)
        float4 dhl[PC], dHa[PC];
#pragma unroll
        for (int j = 0; j < PC; ++j) {
            const float4 g = dhn[j], z = zz[j], t = tt[j], h = hh[j];
            dhl[j] = make_float4((g.x * (1.0f - z.x)) * (1.0f - t.x * t.x), (g.y * (1.0f - z.y)) * (1.0f - t.y * t.y),
                                 (g.z * (1.0f - z.z)) * (1.0f - t.z * t.z), (g.w * (1.0f - z.w)) * (1.0f - t.w * t.w));
            const float4 dz = make_float4((g.x * (h.x - t.x)) * (z.x * (1.0f - z.x)), (g.y * (h.y - t.y)) * (z.y * (1.0f - z.y)),
                                          (g.z * (h.z - t.z)) * (z.z * (1.0f - z.z)), (g.w * (h.w - t.w)) * (z.w * (1.0f - z.w)));
            dHa[j] = make_float4(g.x * z.x, g.y * z.y, g.z * z.z, g.w * z.w);
            stD(a.dhl, j, dhl[j]);
            stD(a.dzl, j, dz);
        }
        // (left free, `g * z` is sunk to the dHR stage, keeping g and z -- 32 registers, 12 of them spilled -- instead of dHa)
#pragma unroll
        for (int j = 0; j < PC; ++j) materialize(dHa[j]);
        f32x4 zacc[PF];
#pragma unroll
        for (int ft = 0; ft < PF; ++ft) zacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        // out[blk] = (in (K = C) x W_g[:, half C + 16 blk ..]) as row pieces; W_g^T is [2C][LDB] in LDS
        auto gemm = [&](const float4 (&in)[PC], int g, int half, f32x4 (&acc)[PC]) {
            const float *w = wg_l + (FOLD ? g * C : g * 2 * C + half * C) * LDB;      // FOLD keeps the H halves only (half == 1)
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_pieces<PC, PC, (WAVES <= 12)>(acc, w, LDB, [&](int j) { return in[j]; });
        };
        // da3[:, g C + 16 blk ..] = clamp mask * piece; z += that piece x Wcat^T
        auto emit_da3 = [&](int g, const f32x4 (&acc)[PC]) {
#pragma unroll
            for (int blk = 0; blk < PC; ++blk) {
                const int bit = 4 * (g * PC + blk);
                const unsigned m = (bit < 32 ? mlo : mhi) >> (bit & 31);
                float4 o;
                o.x = (m & 1u) ? acc[blk][0] : 0.f;
                o.y = (m & 2u) ? acc[blk][1] : 0.f;
                o.z = (m & 4u) ? acc[blk][2] : 0.f;
                o.w = (m & 8u) ? acc[blk][3] : 0.f;
                if (a.da3) st_f4(a.da3, o3, 4 * (g * C + 16 * blk), o);           // kernel-uniform (optional: the conv gradients can come from P^T d_g)
                if (want_z) {
                    mfma_piece<PF>(zacc, wcrow, LDX, g * PC + blk, o);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // FOLD: z += d_g At_g^T (At_g [FIN][C] in LDS at wcrow + g FIN LDB): the gate's whole contribution to the input gradient
        auto emit_z = [&](int g, const float4 (&d)[PC]) {
            if (want_z) gemm_pieces<PF, PC, (WAVES <= 12)>(zacc, wcrow + g * FIN * LDB, LDB, [&](int j) { return d[j]; });
        };
        f32x4 acc[PC];
        // ---- dCH = dhl Wh: d(hh) -> da3[:, 2C..];  dHR -> drl, dH ---------------------------------------------------
        if constexpr (FOLD) {
            emit_z(2, dhl);
        } else {
            gemm(dhl, 2, 0, acc);
            emit_da3(2, acc);
        }
        STG_TRACE_MARK(4);
        float4 rr[PC], hb[PC], dzl[PC];                          // for the dHR stage: in flight under the next product
#pragma unroll
        for (int j = 0; j < PC; ++j) rr[j] = ldC(a.R, j);
#pragma unroll
        for (int j = 0; j < PC; ++j) hb[j] = zero4, dzl[j] = ldD(a.dzl, j);      // this lane's own stores, above
        if (a.H) {
#pragma unroll
            for (int j = 0; j < PC; ++j) hb[j] = ldC(a.H, j);
        }
        gemm(dhl, 2, 1, acc);
        float4 drl[PC];
#pragma unroll
        for (int blk = 0; blk < PC; ++blk) {
            const float4 r = rr[blk], h = hb[blk];
            const float4 d = to_f4(acc[blk]);
            drl[blk] = make_float4((d.x * h.x) * (r.x * (1.0f - r.x)), (d.y * h.y) * (r.y * (1.0f - r.y)),
                                   (d.z * h.z) * (r.z * (1.0f - r.z)), (d.w * h.w) * (r.w * (1.0f - r.w)));
            dHa[blk] = make_float4(dHa[blk].x + d.x * r.x, dHa[blk].y + d.y * r.y, dHa[blk].z + d.z * r.z,
                                   dHa[blk].w + d.w * r.w);
            stD(a.drl, blk, drl[blk]);
        }
        // ---- dCZ = dzl Wz,  dCR = drl Wr: d(hz), d(hr) -> da3;  second halves -> dH (dCZ's first, then dCR's) -------
        STG_TRACE_MARK(5);
        if constexpr (FOLD) {
            emit_z(0, dzl);
        } else {
            gemm(dzl, 0, 0, acc);
            emit_da3(0, acc);
        }
        gemm(dzl, 0, 1, acc);
#pragma unroll
        for (int blk = 0; blk < PC; ++blk)
            dHa[blk] = make_float4(dHa[blk].x + acc[blk][0], dHa[blk].y + acc[blk][1], dHa[blk].z + acc[blk][2],
                                   dHa[blk].w + acc[blk][3]);
        STG_TRACE_MARK(6);
        if constexpr (FOLD) {
            emit_z(1, drl);
        } else {
            gemm(drl, 1, 0, acc);
            emit_da3(1, acc);
        }
        gemm(drl, 1, 1, acc);
#pragma unroll
        for (int blk = 0; blk < PC; ++blk)
            stC(a.dH, blk, make_float4(dHa[blk].x + acc[blk][0], dHa[blk].y + acc[blk][1], dHa[blk].z + acc[blk][2],
                                       dHa[blk].w + acc[blk][3]));
        if (want_z) {
#pragma unroll
            for (int ft = 0; ft < PF; ++ft) st_f4(a.z, oF, 64 * ft, to_f4(zacc[ft]));
        }
        STG_TRACE_MARK(7);
        STG_TRACE_MARK(15);
        int w = 0;
        if (lane == 0) w = atomicAdd(next_w, 1);
        seq = __builtin_amdgcn_readfirstlane(w);
        tile = seq * grid + blk;
    }
}

template <int C, int FIN, int FH, int WAVES, bool GATHER, bool HAS_EW, int HEAD, bool FOLD = false>
int launch_step_bwd(const BwdArgs &a, hipStream_t stream)
{
    using S = BwdShape<C, FIN, FH, WAVES, GATHER, HEAD, FOLD>;
    auto kern = tgcn_step_bwd_kernel<C, FIN, FH, WAVES, GATHER, HAS_EW, HEAD, FOLD>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_step_bwd: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (S::kLds + 512), 32 / WAVES));
    // Tiles are dealt wave-major over the grid (tile = wave * grid + block), so a grid of one workgroup per CU spreads FEWER
    // tiles than wave slots over all CUs (|V| = 25 K: 6 tiles on each of 256 CUs instead of 12 on 131): the launch lasts as long
    // as one SIMD's share of the matrix work.  `step_spread` = 1 restores the packed grid.
    const int64_t packed = ((int64_t)a.num_tiles + WAVES - 1) / WAVES;
    const int64_t spread = tuning().step_spread == 1 ? packed : std::min<int64_t>(a.num_tiles, 256 * per_cu);
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max(packed, spread), 256 * per_cu);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * kWave), S::kLds, stream, a);
    return check_launch("stg_tgcn_step_bwd");
}


}  // namespace
}  // namespace stg

#ifdef STG_STEP_TRACE
extern "C" int stg_debug_set_step_trace_bwd(void *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(stg::g_step_trace), &buf, sizeof(buf));
}
#endif

extern "C" int stg_tgcn_step_bwd(const stg_tgcn_step_bwd_args *p, void *stream_)
{
    using namespace stg;
    if (!p) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: NULL argument block");
    if (!stg_tgcn_step_supported(p->C, p->Fin, p->Fh))
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_bwd: C=%d Fin=%d Fh=%d not supported (64 / 32 / 32)", p->C, p->Fin, p->Fh);
    if (p->N < 0 || p->N > (int64_t)16 * 0x7ffffff0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: bad N");
    if (p->head < 0 || p->head > 2) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: head must be 0, 1 or 2");
    if (p->ld_d != 0 && p->ld_d != p->C && p->ld_d != 3 * p->C)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: ld_d must be 0 / C ([N, C] each) or 3 C (column blocks of one [N, 3C] matrix)");
    if (p->N == 0) return 0;
    const bool gather = p->head != 0 && p->zn != nullptr;
    if (gather && (!p->row_offsets || !p->column_indices || !p->norm_col_edge || !p->norm))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: NULL graph pointer");
    if ((int64_t)p->N * 3 * p->C >= ((int64_t)1 << 30)) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_bwd: too many rows for 32-bit offsets");
    if (p->w_image) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_bwd: w_image (the bf16-split form, ABI 22-25) was retired in ABI 26: pass NULL");
    // folded form: w_fold_t given, no da3 asked for, head >= 1
    const bool fold = p->w_fold_t && !p->da3 && p->head >= 1;
    if (!p->Z || !p->R || !p->Ht || (!p->x3 && !p->clamp_mask && !fold) || !p->WzT || !p->WrT || !p->WhT || !p->dzl || !p->drl || !p->dhl || (!p->da3 && !p->z) || !p->dH)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: NULL cell pointer");
    if (p->z && !p->Wcat && !fold) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: z wanted but Wcat is NULL");
    if (p->head >= 1 && (!p->W1T || !p->Hn || !p->dyt)) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: NULL head pointer");
    if (p->head == 2 && (!p->W2 || !p->y_out || !p->target || !p->g_cost || !p->dyo))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: NULL loss pointer");
    if (p->link_row_ptr && (p->head != 1 || !p->link_other || !p->link_eid || !p->link_y || !p->link_logits || !p->link_target || !p->g_cost ||
                            !(p->link_inv_m > 0.f)))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: the link-loss arguments need head == 1, g_cost and every link_* field");
    const bool d_wide = p->ld_d == 3 * p->C;
    BwdArgs a{};
    a.row_offsets = p->row_offsets; a.column_indices = p->column_indices; a.node_ids = p->node_ids;
    a.nc_edge = p->norm_col_edge; a.ew_edge = p->ew_edge; a.norm = p->norm;
    a.zn = p->zn; a.gy = p->g_y; a.dHn = p->dHn; a.g_cost = p->g_cost;
    a.Z = p->Z; a.R = p->R; a.Ht = p->Ht; a.H = p->H; a.Hn = p->Hn; a.x3 = p->x3; a.y_out = p->y_out; a.target = p->target;
    a.WzT = p->WzT; a.WrT = p->WrT; a.WhT = p->WhT; a.Wcat = p->Wcat; a.W1T = p->W1T; a.W2 = p->W2;
    a.At = p->w_fold_t;
    a.mask = p->clamp_mask;
    if (p->link_row_ptr) {
        if (p->head != 1 || !p->link_other || !p->link_eid || !p->link_y || !p->link_logits || !p->link_target || !p->g_cost ||
            !(p->link_inv_m > 0.f))
            return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_bwd: the link-loss arguments need head == 1, g_cost and every link_* field");
        a.link_row_ptr = p->link_row_ptr; a.link_other = p->link_other; a.link_eid = p->link_eid;
        a.link_y = p->link_y; a.link_logits = p->link_logits; a.link_target = p->link_target; a.link_inv_m = p->link_inv_m;
    }
    a.dzl = p->dzl; a.drl = p->drl; a.dhl = p->dhl; a.da3 = p->da3; a.dH = p->dH; a.z = p->z; a.dyt = p->dyt; a.dyo = p->dyo;
    a.d_wide = d_wide ? 1 : 0;
    a.N = p->N; a.lo = p->lo; a.hi = p->hi; a.two_over_n = 2.0f / (float)p->N; a.num_tiles = (int)((p->N + 15) / 16);
    a.no_coop = tuning().step_coop;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const bool w16 = tuning().step_waves == 16;
#define STG_STEP_BWD(G_, EW_, HD_)                                                         \
    return w16 ? launch_step_bwd<64, 32, 32, 16, G_, EW_, HD_>(a, st) : launch_step_bwd<64, 32, 32, 12, G_, EW_, HD_>(a, st)
    if (fold) {
#define STG_STEP_BWD_F(G_, EW_, HD_)                                                                                 \
    return w16 ? launch_step_bwd<64, 32, 32, 16, G_, EW_, HD_, true>(a, st) : launch_step_bwd<64, 32, 32, 12, G_, EW_, HD_, true>(a, st)
        if (gather) {
            if (p->ew_edge) { if (p->head == 1) STG_STEP_BWD_F(true, true, 1); else STG_STEP_BWD_F(true, true, 2); }
            if (p->head == 1) STG_STEP_BWD_F(true, false, 1); else STG_STEP_BWD_F(true, false, 2);
        }
        if (p->head == 1) STG_STEP_BWD_F(false, false, 1); else STG_STEP_BWD_F(false, false, 2);
#undef STG_STEP_BWD_F
    }
    if (p->head == 0) STG_STEP_BWD(false, false, 0);
    if (gather) {
        if (p->ew_edge) {
            if (p->head == 1) STG_STEP_BWD(true, true, 1);
            STG_STEP_BWD(true, true, 2);
        }
        if (p->head == 1) STG_STEP_BWD(true, false, 1);
        STG_STEP_BWD(true, false, 2);
    }
    if (p->head == 1) STG_STEP_BWD(false, false, 1);
    STG_STEP_BWD(false, false, 2);
#undef STG_STEP_BWD
}

