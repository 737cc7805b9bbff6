// Forward launch of the one-launch-per-step TGCN kernels: layout and design notes in tgcn_step.hpp.
#include "tgcn_step.hpp"

namespace stg {
namespace {

struct FwdArgs {
    const int *row_offsets, *column_indices, *node_ids;
    const float *nc_edge, *ew_edge, *norm;
    const float *x, *a3, *H, *target;
    const float *WcatT, *b3, *Wz, *bz, *Wr, *br, *Wh, *bh, *W1, *b1, *W2, *b2;
    const float *Ag, *bgf, *bound;                     // FOLD: gate Linears with the conv folded in [3C][Fin + C], their biases [3C], {max |Wcat|, max |b3|}
    float *P, *x3, *Z, *R, *Ht, *Hn, *HR, *y, *y_out, *partial;
    unsigned *mask;
    int *status;                                       // optional: |= 1 when an element of x3 is clamped (stg_tgcn_step_fwd_args::fold_status)
    int64_t N;
    float lo, hi;
    int num_tiles;
    int no_coop;                                       // knob "step_coop" 1
};

// FOLD (GATHER only): the conv output enters the gates only through their Linears, so the gate products run on
// Ag = [(Wc_g Wg[:, :C]^T)^T | Wg[:, C:]] (K = Fin + C = 96 instead of 2 C = 128, stg_tgcn_fold_weights) straight from P: no x3 product
// (96 of the tile's 512 matrix instructions), no Wcat in LDS.  Valid while no element of x3 WOULD be clamped; x3 is not formed,
// so the launch checks a bound instead: |x3[r, :]| <= |P[r, :]|_1 max |Wcat| + max |b3| must stay inside [lo, hi], else *status |= 1.
template <int C, int FIN, int FH, int WAVES, bool GATHER, int HEAD, bool FOLD = false>
struct FwdShape {
    static constexpr int K2 = FOLD ? FIN + C : 2 * C, LDW = K2 + 8, LDC = FIN + 8, LD1 = C + 8;   // row strides = 8 mod 16 dwords: see tgcn_step.hpp
    static constexpr int kGate = 3 * C * LDW;
    static constexpr int kCat = (GATHER && !FOLD) ? 3 * C * LDC : 0;
    static constexpr int kHead = HEAD ? FH * LD1 : 0;
    static constexpr int kBias = 6 * C + 2 * FH + 4;          // b3 | bz br bh | b1 | W2 | b2
    // cooperative tiles (FOLD): per group of four waves two [16][LDXB] exchange buffers (H*R, relu(Hn)) + one counter per group
    static constexpr int GROUPS = WAVES / 8, LDXB = C + 8;    // groups that can be busy: a shared tile pays while 8 cnt_b <= WAVES
    static constexpr int kCoop = FOLD ? GROUPS * 2 * 16 * LDXB + 8 : 0;
    static constexpr int kFloats = kGate + kCat + kHead + kBias + kCoop;
    static constexpr size_t kLds = sizeof(float) * (size_t)kFloats;
    static_assert(kLds <= 160 * 1024, "the weights must fit one CU's LDS");
};

template <int C, int FIN, int FH, int WAVES, bool GATHER, bool HAS_EW, int HEAD, bool FOLD = false>
__global__ __launch_bounds__(WAVES *kWave) void tgcn_step_fwd_kernel(const FwdArgs a)
{
    static_assert(!FOLD || GATHER, "the folded form gathers P itself");
    using S = FwdShape<C, FIN, FH, WAVES, GATHER, HEAD, FOLD>;
    constexpr int NT = WAVES * kWave, PC = C / 16, PF = FIN / 16, PH = FH / 16;
    constexpr int LDW = S::LDW, LDC = S::LDC, LD1 = S::LD1;
    static_assert(FIN == 32 && C % 16 == 0 && FH % 16 == 0, "shapes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wg = lds;                                   // Wz | Wr | Wh, each [C][LDW]
    float *WcT = Wg + S::kGate;                        // [3C][LDC]
    float *W1s = WcT + S::kCat;                        // [FH][LD1]
    float *bs = W1s + S::kHead;                        // b3 [3C] | bz br bh [3C] | b1 [FH] | W2 [FH] | b2
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;

    // Tiles of a workgroup: (w * grid + block), w = 0, 1, ...  Wave w starts on tile w; whichever wave finishes FIRST takes the
    // workgroup's next one off a counter in LDS (|V| = 50 K is 12 tiles on every CU and a 13th on 53 of them: taken by the
    // earliest finisher it runs beside the other waves' last phases instead of alone after wave 0's own tile).
    int *const next_w = reinterpret_cast<int *>(bs + (6 * C + 2 * FH + 1));
    STG_TRACE_MARK(0);
    STG_TRACE_MARK(14);
    if (threadIdx.x == 0) *next_w = WAVES;
    // The workgroup's tiles in hand-out order: seq k <-> tile k * grid + block.  Rounds of WAVES tiles; every wave of every workgroup
    // has a tile in a FULL round.  The last, partial round gives this workgroup cnt_b tiles (|V| = 50 K on 256 CUs: 12 full + one on
    // 53 workgroups): run by one wave each they leave ONE SIMD of the CU with a fourth tile while 971 SIMDs are done -- 4 x 7 us of
    // matrix work where every other SIMD has 3 x 7 (profiles/r04_step_folded_trace.json: last SIMD 42 us, median 32.5).  FOLD: a
    // tile of the partial round is instead shared by the FOUR waves of a group (waves 4 g .. 4 g + 3: one per SIMD), each taking one
    // 16-column block of every gate (a quarter of the matrix instructions); H * R and relu(Hn), which the next product needs at
    // full width, cross through LDS.  Every column block is the same chain of products in the same order: results bit-identical.
    const int grid = (int)gridDim.x, blk = (int)blockIdx.x;
    const int full_rounds = a.num_tiles / (WAVES * grid);
    const int rem = a.num_tiles - full_rounds * WAVES * grid;
    const int cnt_b = rem > blk ? (rem - blk - 1) / grid + 1 : 0;
    const bool coop = FOLD && HEAD == 2 && full_rounds >= 1 && cnt_b > 0 && 8 * cnt_b <= WAVES && !a.node_ids && !a.no_coop;
    const int seq_end = full_rounds * WAVES + (coop ? 0 : cnt_b);
    int seq = wave;
    int *const coop_cnt = reinterpret_cast<int *>(bs + S::kBias);
    float *const xbuf = bs + S::kBias + 8;
    if constexpr (FOLD) {
        if (threadIdx.x < 8) coop_cnt[threadIdx.x] = 0;
    }
    int tile = seq * grid + blk;

    // Weights and biases: every global load of the staging in flight at once, then the LDS writes.  Measured and dropped, both
    // SLOWER: gathering the first tile between the two halves (the workgroup barrier then waits for the slowest wave's gather,
    // and waves that start their products at different times overlap better), and -- round 3 -- no barrier at all (a count in LDS
    // each wave bumps after its share, looked at after the wave's first gather, with that gather's extent and index loads
    // issued around the staging loads: the gather ends 1.4 us earlier, the staging 0.5-1.4 us later, the launch +1 us).
    constexpr int kStage4 = (3 * C * S::K2 + ((GATHER && !FOLD) ? 3 * C * FIN : 0) + (HEAD ? FH * C : 0)) / 4;
    const StageSeg segs[5] = {{FOLD ? a.Ag : a.Wz, Wg, C, S::K2, LDW},
                              {FOLD ? a.Ag + C * S::K2 : a.Wr, Wg + C * LDW, C, S::K2, LDW},
                              {FOLD ? a.Ag + 2 * C * S::K2 : a.Wh, Wg + 2 * C * LDW, C, S::K2, LDW},
                              {a.WcatT, WcT, (GATHER && !FOLD) ? 3 * C : 0, FIN, LDC}, {a.W1, W1s, HEAD ? FH : 0, C, LD1}};
    Stager<NT, 5, (kStage4 + NT - 1) / NT> stager;
    const int q = lane & 3, grow = lane >> 2;
    // Lanes past the last row MIRROR row N - 1 (they recompute and rewrite its values bit for bit): no load or store of the tile
    // body sits behind a per-lane guard, so the scheduler sees straight-line code between the products (tgcn_step_bwd.hip).
    auto gather_row = [&](int t) {
        const int64_t gidx = std::min<int64_t>((int64_t)t * 16 + grow, a.N - 1);
        return a.node_ids ? a.node_ids[gidx] : (int)gidx;
    };
    // (measured and dropped, round 5: the first gather's extent / index loads issued around the staging loads -- 36.0-36.7 us
    //  against 36.3; the three waves of a SIMD issuing their first row loads one after the other -- 37.7-38.2: the head of the
    //  launch is bound by the L2 / fabric rate of all gathers together, not by the order of their requests)
    stager.issue(segs);
    // the small vectors (biases, W2): their loads in flight WITH the staging loads, not a round trip of their own after the LDS writes
    static_assert(NT >= 6 * C && NT >= 2 * FH + 1, "one element per thread");
    const int tid = (int)threadIdx.x;
    float sv0 = 0.f, sv1 = 0.f;
    if constexpr (FOLD) {
        if (tid < 3 * C) sv0 = a.bgf[tid];                                   // -> bs[3C + tid]
    } else {
        if (tid < 3 * C) sv0 = a.b3[tid];                                    // -> bs[tid]
        else if (tid < 4 * C) sv0 = a.bz[tid - 3 * C];                       // -> bs[tid]  (bz | br | bh follow b3)
        else if (tid < 5 * C) sv0 = a.br[tid - 4 * C];
        else if (tid < 6 * C) sv0 = a.bh[tid - 5 * C];
    }
    if constexpr (HEAD != 0) {
        if (tid < FH) sv1 = a.b1[tid];                                       // -> bs[6C + tid]
        else if (HEAD == 2 && tid < 2 * FH) sv1 = a.W2[tid - FH];            // -> bs[6C + tid]
        else if (HEAD == 2 && tid == 2 * FH) sv1 = a.b2[0];                  // -> bs[6C + 2 FH]
    }
    stager.commit(segs);
    if constexpr (FOLD) {
        if (tid < 3 * C) bs[3 * C + tid] = sv0;
    } else {
        if (tid < 6 * C) bs[tid] = sv0;
    }
    if constexpr (HEAD != 0) {
        if (tid < (HEAD == 2 ? 2 * FH + 1 : FH)) bs[6 * C + tid] = sv1;
    }
    __syncthreads();
    STG_TRACE_MARK(1);

    const float lo = a.lo, hi = a.hi;
    // this lane's row / piece inside each LDS matrix (tgcn_step.hpp: pinned)
    const float *const wg_l = Wg + pinned((unsigned)(n16 * LDW + 4 * kq)), *const wc_l = WcT + pinned((unsigned)(n16 * LDC + 4 * kq));
    const float *const w1_l = W1s + pinned((unsigned)(n16 * LD1 + 4 * kq)), *const bs_l = bs + pinned((unsigned)(4 * kq));
    // ---- the partial round FIRST, four waves per tile (see above): run after the full rounds it would start when the launch should
    // end (measured: the four waves reach it at 24-34 us and need 7-14 us for it -- a gather of three dependent round trips and two
    // group syncs; profiles/r05_step_coop_trace.jsonl) -----------------------------------------------------------------------------
    if constexpr (FOLD && HEAD == 2) {
        if (coop) {
            constexpr int LDXB = S::LDXB;
            const int grp = wave >> 2, cb = wave & 3;
            float *const xb = xbuf + grp * 2 * 16 * LDXB;
            volatile int *const cnt = coop_cnt + grp;
            float *const xb_l = xb + pinned((unsigned)(n16 * LDXB + 4 * kq));
            int phase = 0;
            auto group_sync = [&]() {
                // this wave's LDS writes are ordered before its counter bump (one in-order LDS pipe per CU); waves of a workgroup are
                // all resident, so the wait always ends
                phase += 4;
                __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
                if (lane == 0) atomicAdd(const_cast<int *>(cnt), 1);
                while (*cnt < phase) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_s_waitcnt(0xc07f);
            };
            for (int ct = grp; ct < cnt_b; ct += S::GROUPS) {
                const int ctile = (full_rounds * WAVES + ct) * grid + blk;
                STG_TRACE_MARK(10);
                float4 p[PF];
                {
                    const int gr = gather_row(ctile);
                    RowGather32<HAS_EW> rg;
                    rg.begin(a.row_offsets, a.norm, gr);
                    rg.indices(a.column_indices, a.nc_edge, a.ew_edge, 0, q);
                    float p8[8];
                    rg.run(p8, a.x, a.column_indices, a.nc_edge, a.ew_edge, q);
                    if (cb == 0) {
                        const unsigned off = (unsigned)gr * (FIN * 4u) + 32u * q;
                        st_f4(a.P, off, 0, make_float4(p8[0], p8[1], p8[2], p8[3]));
                        st_f4(a.P, off, 16, make_float4(p8[4], p8[5], p8[6], p8[7]));
                    }
                    gather_to_pieces(p8, p, n16, kq);
                }
                if (cb == 0) {
                    float s1 = 0.f;
#pragma unroll
                    for (int j = 0; j < PF; ++j) s1 = s1 + ((fabsf(p[j].x) + fabsf(p[j].y)) + (fabsf(p[j].z) + fabsf(p[j].w)));
                    s1 = s1 + __shfl_xor(s1, 16, kWave);
                    s1 = s1 + __shfl_xor(s1, 32, kWave);
                    const float bnd = s1 * a.bound[0] + a.bound[1];
                    if (!(bnd <= a.hi && -bnd >= a.lo)) atomicOr(a.status, 1);
                }
                const int64_t idx = (int64_t)ctile * 16 + n16;
                const bool rok = idx < a.N;
                const unsigned row = (unsigned)std::min<int64_t>(idx, a.N - 1);
                const unsigned oC = (row * C + 4u * kq) * 4u, oF = (row * FH + 4u * kq) * 4u;
                float4 hh[PC];
#pragma unroll
                for (int j = 0; j < PC; ++j) {
                    hh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.H) hh[j] = ld_f4(a.H, oC, 64 * j);
                }
                float4 hmine = hh[0];                                     // this wave's own block of H: piece cb
#pragma unroll
                for (int j = 1; j < PC; ++j) hmine = cb == j ? hh[j] : hmine;
                // one 16-column block of gate g: bg'[16 cb ..] + [P | second] Ag_g[16 cb .., :]^T
                auto gate1 = [&](int g, const float4 (&second)[PC], f32x4 (&acc)[1]) {
                    acc[0] = to_x4(*reinterpret_cast<const float4 *>(bs_l + (3 + g) * C + 16 * cb));
                    gemm_pieces<1, PF + PC>(acc, wg_l + (g * C + 16 * cb) * LDW, LDW, [&](int j) { return j < PF ? p[j % PF] : second[(j - PF) % PC]; });
                };
                f32x4 acc[1];
                gate1(0, hh, acc);
                const float4 zz = make_float4(sigmoid_(acc[0][0]), sigmoid_(acc[0][1]), sigmoid_(acc[0][2]), sigmoid_(acc[0][3]));
                st_f4(a.Z, oC + 64u * cb, 0, zz);
                gate1(1, hh, acc);
                const float4 r = make_float4(sigmoid_(acc[0][0]), sigmoid_(acc[0][1]), sigmoid_(acc[0][2]), sigmoid_(acc[0][3]));
                const float4 hr1 = make_float4(hmine.x * r.x, hmine.y * r.y, hmine.z * r.z, hmine.w * r.w);
                st_f4(a.R, oC + 64u * cb, 0, r);
                st_f4(a.HR, oC + 64u * cb, 0, hr1);
                *reinterpret_cast<float4 *>(xb_l + 16 * cb) = hr1;
                group_sync();
                float4 hr[PC];
#pragma unroll
                for (int j = 0; j < PC; ++j) hr[j] = *reinterpret_cast<const float4 *>(xb_l + 16 * j);
                gate1(2, hr, acc);
                const float4 t = make_float4(tanh_(acc[0][0]), tanh_(acc[0][1]), tanh_(acc[0][2]), tanh_(acc[0][3]));
                const float4 hn1 = make_float4(zz.x * hmine.x + (1.0f - zz.x) * t.x, zz.y * hmine.y + (1.0f - zz.y) * t.y,
                                               zz.z * hmine.z + (1.0f - zz.z) * t.z, zz.w * hmine.w + (1.0f - zz.w) * t.w);
                st_f4(a.Ht, oC + 64u * cb, 0, t);
                st_f4(a.Hn, oC + 64u * cb, 0, hn1);
                *reinterpret_cast<float4 *>(xb_l + 16 * LDXB + 16 * cb) =
                    make_float4(hn1.x < 0.f ? 0.f : hn1.x, hn1.y < 0.f ? 0.f : hn1.y, hn1.z < 0.f ? 0.f : hn1.z, hn1.w < 0.f ? 0.f : hn1.w);
                group_sync();
                if (cb == 0) {                                            // the head (32 matrix instructions) by one wave, as in the tile loop
                    const float tg = ld_f1(a.target, row * 4u);
                    float4 hn[PC];
#pragma unroll
                    for (int j = 0; j < PC; ++j) hn[j] = *reinterpret_cast<const float4 *>(xb_l + 16 * LDXB + 16 * j);
                    f32x4 accy[PH];
#pragma unroll
                    for (int ft = 0; ft < PH; ++ft) accy[ft] = to_x4(*reinterpret_cast<const float4 *>(bs_l + 6 * C + 16 * ft));
                    gemm_pieces<PH, PC>(accy, w1_l, LD1, [&](int j) { return hn[j]; });
#pragma unroll
                    for (int ft = 0; ft < PH; ++ft) st_f4(a.y, oF, 64 * ft, to_f4(accy[ft]));
                    float s = 0.f;
#pragma unroll
                    for (int ft = 0; ft < PH; ++ft) {
                        const float4 w2 = *reinterpret_cast<const float4 *>(bs_l + 6 * C + FH + 16 * ft);
                        s = s + accy[ft][0] * w2.x;
                        s = s + accy[ft][1] * w2.y;
                        s = s + accy[ft][2] * w2.z;
                        s = s + accy[ft][3] * w2.w;
                    }
                    s = s + __shfl_xor(s, 16, kWave);
                    s = s + __shfl_xor(s, 32, kWave);
                    const float yo = s + bs[6 * C + 2 * FH];
                    if (kq == 0) st_f1(a.y_out, row * 4u, yo);
                    const float d = yo - tg;
                    float sq = (rok && kq == 0) ? d * d : 0.f;
                    sq = row16_sum(sq);
                    if (lane == 15) a.partial[ctile] = sq;
                }
                STG_TRACE_MARK(11);
            }
        }
    }

    while (seq < seq_end) {
        // P = A_hat x of the tile, handed from the gather layout to row pieces on the LDS crossbar (no LDS memory)
        float4 p[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) p[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (GATHER) {
            const int gr = gather_row(tile);
            RowGather32<HAS_EW> rg;
            rg.begin(a.row_offsets, a.norm, gr);
            rg.indices(a.column_indices, a.nc_edge, a.ew_edge, 0, q);
            float p8[8];
            rg.run(p8, a.x, a.column_indices, a.nc_edge, a.ew_edge, q);
            const unsigned off = (unsigned)gr * (FIN * 4u) + 32u * q;
            st_f4(a.P, off, 0, make_float4(p8[0], p8[1], p8[2], p8[3]));
            st_f4(a.P, off, 16, make_float4(p8[4], p8[5], p8[6], p8[7]));
            gather_to_pieces(p8, p, n16, kq);
        }
        STG_TRACE_MARK(2);
        if constexpr (FOLD) {
            // the fold assumes an inactive clamp: bound this row's |x3| (pieces of the row sit in its four kq lanes)
            float s1 = 0.f;
#pragma unroll
            for (int j = 0; j < PF; ++j) s1 = s1 + ((fabsf(p[j].x) + fabsf(p[j].y)) + (fabsf(p[j].z) + fabsf(p[j].w)));
            s1 = s1 + __shfl_xor(s1, 16, kWave);
            s1 = s1 + __shfl_xor(s1, 32, kWave);
            const float bnd = s1 * a.bound[0] + a.bound[1];
            if (!(bnd <= a.hi && -bnd >= a.lo)) atomicOr(a.status, 1);          // (also catches NaN)
        }
        const int64_t idx = (int64_t)tile * 16 + n16;
        const bool rok = idx < a.N;                               // only the loss partial looks at it
        // Element offsets are 32-bit (N 3C < 2^30, checked on the host); byte offsets of this lane's first piece in a row of
        // C, 3C and FH floats: one VGPR each next to scalar base pointers, the column is an immediate.
        unsigned row = (unsigned)std::min<int64_t>(idx, a.N - 1);
        if (a.node_ids) row = (unsigned)a.node_ids[row];
        const unsigned oC = (row * C + 4u * kq) * 4u, o3 = (row * (3u * C) + 4u * kq) * 4u, oF = (row * FH + 4u * kq) * 4u;
        float4 hh[PC];
        float tg = 0.f;
#pragma unroll
        for (int j = 0; j < PC; ++j) {
            hh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.H) hh[j] = ld_f4(a.H, oC, 64 * j);
        }
        if constexpr (HEAD == 2) tg = ld_f1(a.target, row * 4u);
        // hg = clamp(x3[:, g C ..]) with x3 = P Wcat + b3 (or a3 + b3), one gate at a time (16 live registers instead
        // of 48); x3 itself (before the clamp) is what the backward pass and the weight gradients read
        // (wq: the step-0 weights of the next product, read from LDS one phase ahead -- tgcn_step.hpp, gemm_chain)
        auto gate_input = [&](int g, float4 (&hg)[PC], float4 (&wq)[PC]) {
            if constexpr (GATHER) {
                f32x4 acc[PC];
#pragma unroll
                for (int ct = 0; ct < PC; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                float4 w0[PC];
#pragma unroll
                for (int ct = 0; ct < PC; ++ct) w0[ct] = wq[ct];
                gemm_chain<PC, PF>(acc, wc_l + g * C * LDC, LDC, [&](int j) { return p[j]; }, w0,
                                   [&]() { load_w<PC>(wq, wg_l + g * C * LDW, LDW, 0); });
#pragma unroll
                for (int ct = 0; ct < PC; ++ct) {
                    const float4 b = *reinterpret_cast<const float4 *>(bs_l + g * C + 16 * ct);
                    hg[ct] = make_float4(acc[ct][0] + b.x, acc[ct][1] + b.y, acc[ct][2] + b.z, acc[ct][3] + b.w);
                }
            } else {
                load_w<PC>(wq, wg_l + g * C * LDW, LDW, 0);
#pragma unroll
                for (int ct = 0; ct < PC; ++ct) {
                    const float4 v = ld_f4(a.a3, o3, 4 * (g * C + 16 * ct));
                    const float4 b = *reinterpret_cast<const float4 *>(bs_l + g * C + 16 * ct);
                    hg[ct] = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
                }
            }
            unsigned gm = 0u;             // clamp mask of this gate's columns: bit 4 ct + i <-> column 16 ct + 4 kq + i
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) {
                if (a.x3) st_f4(a.x3, o3, 4 * (g * C + 16 * ct), hg[ct]);     // kernel-uniform (x3 is optional when the mask is taken)
                const float4 v = hg[ct];
                hg[ct] = make_float4(clamp3(v.x, lo, hi), clamp3(v.y, lo, hi), clamp3(v.z, lo, hi), clamp3(v.w, lo, hi));
                // inside [lo, hi]  <=>  the clamp left the value alone
                gm |= ((hg[ct].x == v.x ? 1u : 0u) | (hg[ct].y == v.y ? 2u : 0u) | (hg[ct].z == v.z ? 4u : 0u) | (hg[ct].w == v.w ? 8u : 0u)) << (4 * ct);
            }
            if (a.mask) a.mask[row * 12u + 4 * g + kq] = gm;
            if (a.status && gm != (PC == 4 ? 0xffffu : (1u << (4 * PC)) - 1u)) atomicOr(a.status, 1);
        };
        // acc = bias + [hg | second] W_g^T   (W_g [C][2C] in LDS, torch Linear layout)
        // on return wq holds the step-0 weights of what follows: the next gate's x3 product, or the head (in wq[0 .. PH))
        auto gate = [&](int g, const float4 (&hg)[PC], const float4 (&second)[PC], f32x4 (&acc)[PC], float4 (&wq)[PC]) {
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) acc[ct] = to_x4(*reinterpret_cast<const float4 *>(bs_l + (3 + g) * C + 16 * ct));
            float4 w0[PC];
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) w0[ct] = wq[ct];
            gemm_chain<PC, 2 * PC>(acc, wg_l + g * C * LDW, LDW, [&](int j) { return j < PC ? hg[j % PC] : second[j % PC]; }, w0, [&]() {
                if (g < 2) {
                    if constexpr (GATHER) load_w<PC>(wq, wc_l + (g + 1) * C * LDC, LDC, 0);
                } else if constexpr (HEAD != 0) {
                    float4 wh[PH];
                    load_w<PH>(wh, w1_l, LD1, 0);
#pragma unroll
                    for (int ft = 0; ft < PH; ++ft) wq[ft] = wh[ft];
                }
            });
        };

        // FOLD: acc = bg'_g + [P | second] Ag_g^T   (Ag_g [C][Fin + C] in LDS); wq as above (the next gate's, or the head's, step 0)
        auto gate_f = [&](int g, const float4 (&second)[PC], f32x4 (&acc)[PC], float4 (&wq)[PC]) {
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) acc[ct] = to_x4(*reinterpret_cast<const float4 *>(bs_l + (3 + g) * C + 16 * ct));
            float4 w0[PC];
#pragma unroll
            for (int ct = 0; ct < PC; ++ct) w0[ct] = wq[ct];
            gemm_chain<PC, PF + PC>(acc, wg_l + g * C * LDW, LDW, [&](int j) { return j < PF ? p[j % PF] : second[(j - PF) % PC]; }, w0, [&]() {
                if (g < 2) {
                    load_w<PC>(wq, wg_l + (g + 1) * C * LDW, LDW, 0);
                } else if constexpr (HEAD != 0) {
                    float4 wh[PH];
                    load_w<PH>(wh, w1_l, LD1, 0);
#pragma unroll
                    for (int ft = 0; ft < PH; ++ft) wq[ft] = wh[ft];
                }
            });
        };

        float4 wq[PC];
        if constexpr (FOLD) load_w<PC>(wq, wg_l, LDW, 0);
        else if constexpr (GATHER) load_w<PC>(wq, wc_l, LDC, 0);
        // ---- Z = sigmoid([hz | H] Wz^T + bz),  R = sigmoid([hr | H] Wr^T + br) -----------------------------------
        float4 zz[PC], hr[PC];
        {
            float4 hg[PC];
            f32x4 acc[PC];
            if constexpr (FOLD) {
                gate_f(0, hh, acc, wq);
            } else {
                gate_input(0, hg, wq);
                STG_TRACE_MARK(8);                          // (trace build only) the x3 product, its stores, clamp and mask
                gate(0, hg, hh, acc, wq);
            }
            STG_TRACE_MARK(9);                              // the gate product
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                zz[j] = make_float4(sigmoid_(acc[j][0]), sigmoid_(acc[j][1]), sigmoid_(acc[j][2]), sigmoid_(acc[j][3]));
                st_f4(a.Z, oC, 64 * j, zz[j]);
            }
            STG_TRACE_MARK(3);
            if constexpr (FOLD) {
                gate_f(1, hh, acc, wq);
            } else {
                gate_input(1, hg, wq);
                gate(1, hg, hh, acc, wq);
            }
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const float4 r = make_float4(sigmoid_(acc[j][0]), sigmoid_(acc[j][1]), sigmoid_(acc[j][2]), sigmoid_(acc[j][3]));
                hr[j] = make_float4(hh[j].x * r.x, hh[j].y * r.y, hh[j].z * r.z, hh[j].w * r.w);
                st_f4(a.R, oC, 64 * j, r);
                st_f4(a.HR, oC, 64 * j, hr[j]);
            }
        }

        STG_TRACE_MARK(4);
        // ---- Ht = tanh([hh | H*R] Wh^T + bh);  Hn = Z*H + (1 - Z)*Ht ----------------------------------------------
        float4 hn[PC];
        {
            float4 hg[PC];
            f32x4 acc[PC];
            if constexpr (FOLD) {
                gate_f(2, hr, acc, wq);
            } else {
                gate_input(2, hg, wq);
                gate(2, hg, hr, acc, wq);
            }
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const float4 t = make_float4(tanh_(acc[j][0]), tanh_(acc[j][1]), tanh_(acc[j][2]), tanh_(acc[j][3]));
                const float4 z = zz[j], h = hh[j];
                hn[j] = make_float4(z.x * h.x + (1.0f - z.x) * t.x, z.y * h.y + (1.0f - z.y) * t.y,
                                    z.z * h.z + (1.0f - z.z) * t.z, z.w * h.w + (1.0f - z.w) * t.w);
                st_f4(a.Ht, oC, 64 * j, t);
                st_f4(a.Hn, oC, 64 * j, hn[j]);
            }
        }

        STG_TRACE_MARK(5);
        // ---- head: y = relu(Hn) W1^T + b1;  y_out = y W2^T + b2;  partial[tile] = sum (y_out - target)^2 ------------
        if constexpr (HEAD != 0) {
            f32x4 accy[PH];
#pragma unroll
            for (int ft = 0; ft < PH; ++ft) accy[ft] = to_x4(*reinterpret_cast<const float4 *>(bs_l + 6 * C + 16 * ft));
            float4 wh[PH];
#pragma unroll
            for (int ft = 0; ft < PH; ++ft) wh[ft] = wq[ft];
            gemm_chain<PH, PC>(accy, w1_l, LD1, [&](int j) {
                return make_float4(hn[j].x < 0.f ? 0.f : hn[j].x, hn[j].y < 0.f ? 0.f : hn[j].y,
                                   hn[j].z < 0.f ? 0.f : hn[j].z, hn[j].w < 0.f ? 0.f : hn[j].w);
            }, wh, []() {});
#pragma unroll
            for (int ft = 0; ft < PH; ++ft) st_f4(a.y, oF, 64 * ft, to_f4(accy[ft]));
            if constexpr (HEAD == 2) {
                float s = 0.f;
#pragma unroll
                for (int ft = 0; ft < PH; ++ft) {
                    const float4 w2 = *reinterpret_cast<const float4 *>(bs_l + 6 * C + FH + 16 * ft);
                    s = s + accy[ft][0] * w2.x;
                    s = s + accy[ft][1] * w2.y;
                    s = s + accy[ft][2] * w2.z;
                    s = s + accy[ft][3] * w2.w;
                }
                s = s + __shfl_xor(s, 16, kWave);                   // the row's four kq lanes
                s = s + __shfl_xor(s, 32, kWave);
                const float yo = s + bs[6 * C + 2 * FH];
                if (kq == 0) st_f1(a.y_out, row * 4u, yo);
                const float d = yo - tg;
                float sq = (rok && kq == 0) ? d * d : 0.f;
                sq = row16_sum(sq);                                 // lanes 0..15 (kq = 0): the tile's 16 rows, in lane order
                if (lane == 15) a.partial[tile] = sq;
            }
        }
        STG_TRACE_MARK(6);
        STG_TRACE_MARK(15);
        int w = 0;
        if (lane == 0) w = atomicAdd(next_w, 1);
        seq = __builtin_amdgcn_readfirstlane(w);
        tile = seq * grid + blk;
    }
}

template <int C, int FIN, int FH, int WAVES, bool GATHER, bool HAS_EW, int HEAD, bool FOLD = false>
int launch_step_fwd(const FwdArgs &a, hipStream_t stream)
{
    using S = FwdShape<C, FIN, FH, WAVES, GATHER, HEAD, FOLD>;
    auto kern = tgcn_step_fwd_kernel<C, FIN, FH, WAVES, GATHER, HAS_EW, HEAD, FOLD>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_step_fwd: %s", hipGetErrorString(e));
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
    return check_launch("stg_tgcn_step_fwd");
}


// cost = sum_t mean_t, mean_t = (sum of step t's tile partials, in a fixed order) / N.  One workgroup per step adds that
// step's partials (the steps in parallel: a single workgroup walking 25 x 3125 partials took 138 us per window), then
// one thread adds the per-step terms in order, as the loop's `cost = cost + loss` does.
__global__ __launch_bounds__(kBlock) void window_loss_kernel(const float *__restrict__ partial, int num_tiles, int64_t stride,
                                                             float inv_n, float *__restrict__ step_loss)
{
    __shared__ float s[kBlock];
    const float *p = partial + (int64_t)blockIdx.x * stride;
    float v = 0.f;
    for (int i = threadIdx.x; i < num_tiles; i += kBlock) v = v + p[i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) step_loss[blockIdx.x] = s[0] * inv_n;
}

__global__ void window_cost_kernel(const float *__restrict__ step_loss, int steps, float *__restrict__ cost)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float total = 0.f;
        for (int t = 0; t < steps; ++t) total = total + step_loss[t];
        cost[0] = total;
    }
}

int window_loss_launch(const float *partials, int steps, int count, int64_t stride, float inv_n, float *step_loss, float *cost,
                       hipStream_t stream, const char *what)
{
    if (!step_loss) return fail(STG_ERR_INVALID_ARGUMENT, "%s: step_loss [steps] is required (it is the scratch of the sum)", what);
    hipLaunchKernelGGL(window_loss_kernel, dim3((unsigned)steps), dim3(kBlock), 0, stream, partials, count, stride, inv_n, step_loss);
    hipLaunchKernelGGL(window_cost_kernel, dim3(1), dim3(kWave), 0, stream, step_loss, steps, cost);
    return check_launch(what);
}

}  // namespace
}  // namespace stg

#ifdef STG_STEP_TRACE
extern "C" int stg_debug_set_step_trace_fwd(void *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(stg::g_step_trace), &buf, sizeof(buf));
}
#endif

extern "C" int stg_tgcn_step_supported(int32_t C, int32_t Fin, int32_t Fh)
{
    return C == 64 && Fin == 32 && Fh == 32;
}

extern "C" size_t stg_tgcn_step_loss_partials(int64_t N) { return N > 0 ? (size_t)((N + 15) / 16) : 0; }

extern "C" int stg_tgcn_step_fwd(const stg_tgcn_step_fwd_args *p, void *stream_)
{
    using namespace stg;
    if (!p) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: NULL argument block");
    if (!stg_tgcn_step_supported(p->C, p->Fin, p->Fh))
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_fwd: C=%d Fin=%d Fh=%d not supported (64 / 32 / 32)", p->C, p->Fin, p->Fh);
    if (p->N < 0 || p->N > (int64_t)16 * 0x7ffffff0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: bad N");
    if (p->head < 0 || p->head > 2) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: head must be 0, 1 or 2");
    if (p->N == 0) return 0;
    const bool gather = p->x != nullptr;
    if (p->w_image) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_fwd: w_image (the bf16-split form, ABI 22-25) was retired in ABI 26: pass NULL");
    // folded gate weights: x3 is then not formed (it must not be asked for) and the launch bounds it instead
    const bool fold32 = p->w_fold != nullptr;
    if (fold32 && (!gather || p->head < 1 || p->x3 || !p->b_fold || !p->fold_bound || !p->fold_status))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: the folded form needs x, head >= 1, x3 == NULL and w_fold, b_fold, fold_bound, fold_status together");
    if (gather ? (!p->row_offsets || !p->column_indices || !p->norm_col_edge || !p->norm || !p->WcatT || !p->P) : !p->a3)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: NULL graph / input pointer");
    if (!p->b3 || !p->Wz || !p->bz || !p->Wr || !p->br || !p->Wh || !p->bh || (!p->x3 && !p->clamp_mask && !fold32) || !p->Z || !p->R || !p->Ht || !p->Hn || !p->HR)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: NULL cell pointer");
    if (p->head >= 1 && (!p->W1 || !p->b1 || !p->y)) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: NULL head pointer");
    if (p->head == 2 && (!p->W2 || !p->b2 || !p->y_out || !p->target || !p->loss_partial))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd: NULL loss pointer");
    if ((int64_t)p->N * 3 * p->C >= ((int64_t)1 << 30)) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_fwd: too many rows for 32-bit offsets");

    FwdArgs a{};
    a.row_offsets = p->row_offsets; a.column_indices = p->column_indices; a.node_ids = p->node_ids;
    a.nc_edge = p->norm_col_edge; a.ew_edge = p->ew_edge; a.norm = p->norm;
    a.x = p->x; a.a3 = p->a3; a.H = p->H; a.target = p->target;
    a.WcatT = p->WcatT; a.b3 = p->b3; a.Wz = p->Wz; a.bz = p->bz; a.Wr = p->Wr; a.br = p->br; a.Wh = p->Wh; a.bh = p->bh;
    a.W1 = p->W1; a.b1 = p->b1; a.W2 = p->W2; a.b2 = p->b2;
    a.Ag = p->w_fold; a.bgf = p->b_fold; a.bound = p->fold_bound;
    a.P = p->P; a.x3 = p->x3; a.Z = p->Z; a.R = p->R; a.Ht = p->Ht; a.Hn = p->Hn; a.HR = p->HR; a.y = p->y;
    a.y_out = p->y_out; a.partial = p->loss_partial; a.mask = p->clamp_mask; a.status = p->fold_status;
    a.N = p->N; a.lo = p->lo; a.hi = p->hi; a.num_tiles = (int)((p->N + 15) / 16);
    a.no_coop = tuning().step_coop;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const bool w16 = tuning().step_waves == 16;
#define STG_STEP_FWD(G_, EW_, HD_)                                                         \
    return w16 ? launch_step_fwd<64, 32, 32, 16, G_, EW_, HD_>(a, st) : launch_step_fwd<64, 32, 32, 12, G_, EW_, HD_>(a, st)
#define STG_STEP_FWD_H(G_, EW_)                                                            \
    switch (p->head) {                                                                     \
        case 0: STG_STEP_FWD(G_, EW_, 0);                                                  \
        case 1: STG_STEP_FWD(G_, EW_, 1);                                                  \
        default: STG_STEP_FWD(G_, EW_, 2);                                                 \
    }
    if (fold32) {
#define STG_STEP_FWD_F(EW_, HD_)                                                                                     \
    return w16 ? launch_step_fwd<64, 32, 32, 16, true, EW_, HD_, true>(a, st) : launch_step_fwd<64, 32, 32, 12, true, EW_, HD_, true>(a, st)
        if (p->ew_edge) { if (p->head == 1) STG_STEP_FWD_F(true, 1); else STG_STEP_FWD_F(true, 2); }
        if (p->head == 1) STG_STEP_FWD_F(false, 1); else STG_STEP_FWD_F(false, 2);
#undef STG_STEP_FWD_F
    }
    if (gather) {
        if (p->ew_edge) { STG_STEP_FWD_H(true, true) } else { STG_STEP_FWD_H(true, false) }
    }
    STG_STEP_FWD_H(false, false)
#undef STG_STEP_FWD_H
#undef STG_STEP_FWD
}

extern "C" int stg_tgcn_window_loss(const float *partials, int32_t steps, int64_t N, int64_t step_stride, float *step_loss,
                                    float *cost, void *stream)
{
    using namespace stg;
    if (steps <= 0 || N <= 0 || step_stride < (int64_t)stg_tgcn_step_loss_partials(N))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_window_loss: bad shape steps=%d N=%lld stride=%lld", steps,
                    (long long)N, (long long)step_stride);
    if (!partials || !cost) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_window_loss: NULL pointer argument");
    return window_loss_launch(partials, steps, (int)stg_tgcn_step_loss_partials(N), step_stride, 1.0f / (float)N, step_loss, cost,
                              static_cast<hipStream_t>(stream), "stg_tgcn_window_loss");
}

extern "C" int stg_partial_sums_loss(const float *partials, int32_t steps, int32_t count, int64_t step_stride, float inv_n,
                                     float *step_loss, float *cost, void *stream)
{
    using namespace stg;
    if (steps <= 0 || count <= 0 || step_stride < count)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_partial_sums_loss: bad shape steps=%d count=%d stride=%lld", steps, count,
                    (long long)step_stride);
    if (!partials || !cost) return fail(STG_ERR_INVALID_ARGUMENT, "stg_partial_sums_loss: NULL pointer argument");
    return window_loss_launch(partials, steps, count, step_stride, inv_n, step_loss, cost, static_cast<hipStream_t>(stream),
                              "stg_partial_sums_loss");
}

// The layouts the two step launches want of a window's weights, made in ONE launch (the window loop made them with two
// torch.cat, one transpose for the forward pass and four for the backward pass: seven launches of ~4.6 us per window):
//   Wcat [Fin, 3C] = [Wcz | Wcr | Wch], WcatT [3C, Fin], b3 [3C] = [bcz | bcr | bch], WzT / WrT / WhT [2C, C], W1T [C, Fh].
namespace stg {
namespace {
struct PackArgs {
    const float *Wc[3], *bc[3], *Wg[3], *W1;
    float *Wcat, *WcatT, *b3, *WgT[3], *W1T;
    int C, Fin, Fh;
};
__global__ __launch_bounds__(kBlock) void tgcn_pack_weights_kernel(const PackArgs a)
{
    const int C = a.C, Fin = a.Fin, Fh = a.Fh;
    const int n_cat = Fin * 3 * C, n_b = 3 * C, n_g = 2 * C * C, n_1 = C * Fh;
    const int total = n_cat + n_b + 3 * n_g + n_1;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        int j = i;
        if (j < n_cat) {                                   // (f, c) of Wcat: gate c / C
            const int f = j / (3 * C), c = j - f * 3 * C, g = c / C;
            const float v = a.Wc[g][f * C + (c - g * C)];
            a.Wcat[j] = v;
            a.WcatT[c * Fin + f] = v;
            continue;
        }
        j -= n_cat;
        if (j < n_b) {
            a.b3[j] = a.bc[j / C][j % C];
            continue;
        }
        j -= n_b;
        if (j < 3 * n_g) {                                 // gate Linear weights [C][2C] -> [2C][C]
            const int g = j / n_g, r = j - g * n_g, o = r / (2 * C), k = r - o * 2 * C;
            a.WgT[g][k * C + o] = a.Wg[g][r];
            continue;
        }
        j -= 3 * n_g;
        const int o = j / C, k = j - o * C;                // W1 [Fh][C] -> [C][Fh]
        a.W1T[k * Fh + o] = a.W1[j];
    }
}
}  // namespace
}  // namespace stg

extern "C" int stg_tgcn_pack_weights(const float *Wcz, const float *Wcr, const float *Wch, const float *bcz, const float *bcr,
                                     const float *bch, const float *Wz, const float *Wr, const float *Wh, const float *W1,
                                     float *Wcat, float *WcatT, float *b3, float *WzT, float *WrT, float *WhT, float *W1T,
                                     int32_t C, int32_t Fin, int32_t Fh, void *stream)
{
    using namespace stg;
    if (C <= 0 || Fin <= 0 || Fh <= 0 || (int64_t)C * (Fin * 3 + 6 * C + Fh + 3) > (1 << 28))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_pack_weights: bad shape C=%d Fin=%d Fh=%d", C, Fin, Fh);
    if (!Wcz || !Wcr || !Wch || !bcz || !bcr || !bch || !Wz || !Wr || !Wh || !W1 || !Wcat || !WcatT || !b3 || !WzT || !WrT || !WhT || !W1T)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_pack_weights: NULL pointer argument");
    PackArgs a{{Wcz, Wcr, Wch}, {bcz, bcr, bch}, {Wz, Wr, Wh}, W1, Wcat, WcatT, b3, {WzT, WrT, WhT}, W1T, C, Fin, Fh};
    const int total = Fin * 3 * C + 3 * C + 3 * 2 * C * C + C * Fh;
    hipLaunchKernelGGL(tgcn_pack_weights_kernel, dim3((unsigned)std::min((total + kBlock - 1) / kBlock, 256)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("stg_tgcn_pack_weights");
}
