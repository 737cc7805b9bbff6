// Forward launch of the one-launch TGCN step, FOLDED form: every product a 3-term bf16 split on v_mfma_f32_16x16x32_bf16 (the
// arithmetic of tgcn_stepx.hpp), every wave owning ALL output columns of its 16 rows, all weights in LDS.
//
// Why another form.  The fp32 form (tgcn_step_fwd.hip) is bound by v_mfma_f32_16x16x4_f32 sharing the vector lanes (DESIGN.md
// section 0.2); the first matrix-core form (tgcn_stepx_fwd.hip) had to cut the output columns over four waves because the gate
// weights as bf16 triples (3 x 128 x 64 x 6 B = 147 KB) + Wcat (36 KB) + W1 (12 KB) exceed a CU's LDS, and the cut multiplied the
// vector work.  What makes everything fit here: the conv output only ever enters the gates through a Linear,
//     [clamp(P Wc_g + bc_g) | H] Wg^T + bg  =  P (Wc_g Wg[:, :C]^T) + H Wg[:, C:]^T + (bc_g Wg[:, :C]^T + bg)      (no clamping),
// so the gate products run on the FOLDED weights Ag = [(Wc_g Wg[:, :C]^T)^T | Wg[:, C:]] of K = Fin + C = 96 instead of 2 C = 128
// (formed once per window on the host side): 3 x 96 x 64 x 6 B = 108 KB, + Wcat 36 KB + W1 12 KB = 156 KB of fragments.  x3 = P Wcat
// + b3 is still computed (it is a saved tensor: the weight gradients and the backward launch read it) and with it the EXACT clamp
// mask (reference nn/pytorch/temporal/tgcn.py:23,31,39: clamp to +-1e6); a clamped element -- it never happens on sane data --
// invalidates the fold and raises *status (sticky; the host side checks it and reruns / refuses).  Products per 16-row tile:
// 72 (x3) + 216 (gates) + 24 (head) matrix instructions of 16 cycles = 5.0 K cycles, against 512 x 32 = 16.4 K in the fp32 form.
//
// Layout: the row-piece scheme with the xcol k-order of tgcn_stepx.hpp -- lane (n16, kq) holds columns 16 ct + 4 kq .. + 3 of row
// n16 for every column tile ct, and the eight k values of K-block b ARE its pieces 2 b and 2 b + 1 -- so the output of one product
// is the operand of the next without leaving the lane's registers: no LDS exchange, no barrier after the weights are staged.
// The gather has the same shape: the four lanes of a row walk its edges together, each taking 2 x 16 bytes of the 128-byte
// neighbour row (P's arithmetic and order are gcn_agg's: bit-identical P).  One workgroup of 16 waves per CU, one tile per wave at
// |V| = 50 K.  Weights are split into the LDS image by the workgroup itself from the fp32 matrices (112 KB out of L2 per CU).
#include "tgcn_stepx.hpp"

#include "../../include/stgraph_hip.h"

namespace stg {
namespace {

struct FwdFArgs {
    const int *row_offsets, *column_indices;
    const float *nc_edge, *ew_edge, *norm;
    const float *x, *H, *target;
    const float *Acat, *Ag, *W1;                       // [3C][Fin] (= WcatT), [3C][Fin + C] folded, [Fh][C]
    const float *b3, *bg, *b1, *W2, *b2;               // [3C], [3C] folded, [Fh], [Fh], [1]
    float *P, *x3, *Z, *R, *Ht, *Hn, *HR, *y, *y_out, *partial;
    unsigned *mask;                                    // the fp32 form's layout: [row][3 gates][4 kq], bit 4 ct + i <-> column 16 ct + 4 kq + i
    int *status;
    int64_t N;
    float lo, hi;
    int num_tiles;
};

#ifndef STG_STEPF_WAVES
#define STG_STEPF_WAVES 12
#endif
constexpr int kFWaves = STG_STEPF_WAVES;
#ifdef STG_STEPF_NOFENCE
#define STG_STEPF_FENCE() ((void)0)
#else
#define STG_STEPF_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int kFKg = (kXFin + kXC) / 32;                                // K-blocks of a folded gate product: 3
constexpr int kFSecCat = 0;                                           // fragment triples: [ct 0..11]
constexpr int kFSecGate = kFSecCat + 12;                              // [ct 0..11][kb 0..2]
constexpr int kFSecHead = kFSecGate + 12 * kFKg;                      // [ct' 0..1][kb 0..1]
constexpr int kFTriples = kFSecHead + 4;                              // 52
constexpr int kFLdsBias = kFTriples * kXTerms * kFragBytes;           // 159 744
constexpr int kFBiasFloats = 6 * kXC + 2 * kXFh + 4;                  // b3 | bg | b1 | W2 | b2
constexpr int kFLds = kFLdsBias + 4 * kFBiasFloats;
static_assert(kFLds <= 160 * 1024, "LDS budget");

__device__ __forceinline__ Frag3 triple(const char *img, int idx, int lane) { return wfrag_load(img, idx * kXTerms, lane); }

// a[0], a[1] += W(ct0), W(ct0 + 1) x X over one K-block: two accumulators in turn
__device__ __forceinline__ void mfma6x2w(f32x4 &a0, f32x4 &a1, const Frag3 &w0, const Frag3 &w1, const Frag3 &x)
{
#define STG_F_ROUND(TW, TX)                                                              \
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.t[TW], x.t[TX], a0, 0, 0, 0);        \
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.t[TW], x.t[TX], a1, 0, 0, 0);
    STG_F_ROUND(0, 2)
    STG_F_ROUND(2, 0)
    STG_F_ROUND(1, 1)
    STG_F_ROUND(0, 1)
    STG_F_ROUND(1, 0)
    STG_F_ROUND(0, 0)
#undef STG_F_ROUND
}

template <bool HAS_EW, int HEAD, bool GATHER>
__global__ __launch_bounds__(kFWaves * 64) void tgcn_stepf_fwd_kernel(const FwdFArgs a)
{
    constexpr int C = kXC, FIN = kXFin, FH = kXFh, KG = FIN + C;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const float *const sB = reinterpret_cast<const float *>(lds + kFLdsBias);       // b3 [0,192) | bg [192,384) | b1 | W2 | b2
    STGX_MARK(0);

    // ---- the weights: fp32 matrices -> fragment image (A operand: row = output column, k = xcol(b, kq, i)) ------------------
    {
        constexpr int kIts = (kFTriples + kFWaves - 1) / kFWaves;
        float4 lo[kIts], hi[kIts];
#pragma unroll
        for (int it = 0; it < kIts; ++it) {                                         // all loads first: one round trip
            const int f = it * kFWaves + wave;
            lo[it] = hi[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < kFTriples) {
                const float *src;
                if (f < kFSecGate) src = a.Acat + (16 * f + n16) * FIN + 4 * kq;
                else if (f < kFSecHead) src = a.Ag + (16 * ((f - kFSecGate) / kFKg) + n16) * KG + 32 * ((f - kFSecGate) % kFKg) + 4 * kq;
                else src = a.W1 + (16 * ((f - kFSecHead) >> 1) + n16) * C + 32 * ((f - kFSecHead) & 1) + 4 * kq;
                lo[it] = *reinterpret_cast<const float4 *>(src);
                hi[it] = *reinterpret_cast<const float4 *>(src + 16);
            }
        }
#pragma unroll
        for (int it = 0; it < kIts; ++it) {
            const int f = it * kFWaves + wave;
            if (f < kFTriples) {
                const Frag3 fr = frag_of(lo[it], hi[it]);
#pragma unroll
                for (int t = 0; t < kXTerms; ++t)
                    *reinterpret_cast<uint4 *>(lds + ((size_t)(f * kXTerms + t) * 64 + lane) * 16) = __builtin_bit_cast(uint4, fr.t[t]);
            }
        }
        float *bd = reinterpret_cast<float *>(lds + kFLdsBias);
        for (int i = threadIdx.x; i < kFBiasFloats; i += kFWaves * 64) {
            float v = 0.f;
            if (i < 3 * C) v = a.b3[i];
            else if (i < 6 * C) v = a.bg[i - 3 * C];
            else if (i < 6 * C + FH) v = a.b1[i - 6 * C];
            else if (i < 6 * C + 2 * FH) v = a.W2 ? a.W2[i - 6 * C - FH] : 0.f;
            else if (i == 6 * C + 2 * FH) v = a.b2 ? a.b2[0] : 0.f;
            bd[i] = v;
        }
    }
    __syncthreads();

    const float lo_c = a.lo, hi_c = a.hi;
    STGX_MARK(1);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = wave * (int)gridDim.x + (int)blockIdx.x; tile < a.num_tiles; tile += (int)gridDim.x * kFWaves) {
        const int64_t idx = (int64_t)tile * 16 + n16;
        const unsigned row = (unsigned)min(idx, a.N - 1);                    // lanes past the last row mirror row N - 1
        const unsigned oC = (row * C + 4u * kq) * 4u;                        // + 64 ct: this lane's piece ct of a row of C floats
        // ---- the gather of P: the four lanes of a row walk its edges together.  Three dependent round trips, then a stream: the
        // row's extent; the records (column, norm, weight) of its first 32 edges, eight per lane of the row, all at once; the
        // neighbour rows eight edges at a time, each edge's record handed to the row's other lanes by ds_bpermute.  (A first version
        // fetched four records, then four rows, per loop iteration: two round trips per four edges, 22 us median and 44 us worst
        // for a tile's gather -- tools/diag/stepf_trace.py.)  Edges beyond 32 of a row take that slow loop.
        float4 plo = zero4, phi = zero4;
        if constexpr (!GATHER) {                                            // P = A_hat x given (stg_gcn_agg_edge at width Fin)
            const unsigned oPi = (row * FIN + 4u * kq) * 4u;
            plo = ld_f4(a.P, oPi, 0);
            phi = ld_f4(a.P, oPi, 64);
        } else {
            const int beg = a.row_offsets[row], end = a.row_offsets[row + 1];
            const float nr = a.norm[row];
            const int deg = end - beg, last = max(end - 1, 0);
            const int max_deg = wave_max_nonneg(deg);
            int rc[8];
            float rn[8], rw[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {                                // past the end of its row a lane re-reads the last edge and drops the term
                const int e = min(beg + 8 * kq + jj, last);
                rc[jj] = a.column_indices[e];
                rn[jj] = a.nc_edge[e];
                rw[jj] = 1.f;
                if constexpr (HAS_EW) rw[jj] = a.ew_edge[e];
            }
            auto add_edge = [&](bool ok, float nc, float w, const float4 &vl, const float4 &vh) {
                auto acc1 = [&](float &acc, float v) {                       // (nc * x) * w, added in edge order: gcn_agg's arithmetic
                    float t = nc * v;
                    if constexpr (HAS_EW) t = t * w;
                    acc = ok ? acc + t : acc;
                };
                acc1(plo.x, vl.x); acc1(plo.y, vl.y); acc1(plo.z, vl.z); acc1(plo.w, vl.w);
                acc1(phi.x, vh.x); acc1(phi.y, vh.y); acc1(phi.z, vh.z); acc1(phi.w, vh.w);
            };
            const int fast = min(max_deg, 32);
            for (int base = 0; base < fast; base += 8) {
                const int from = ((base >> 3) * 16 + n16) * 4;               // the lane of this row that holds edges base .. base + 7
                float nc[8], w[8];
                float4 vl[8], vh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = __builtin_amdgcn_ds_bpermute(from, rc[u]);
                    nc[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rn[u])));
                    w[u] = 1.f;
                    if constexpr (HAS_EW) w[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rw[u])));
                    const unsigned off = (unsigned)c * (FIN * 4u) + 16u * kq;
                    vl[u] = ld_f4(a.x, off, 0);
                    vh[u] = ld_f4(a.x, off, 64);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) add_edge(base + u < deg, nc[u], w[u], vl[u], vh[u]);
            }
            for (int base = 32; base < max_deg; base += 4) {
                int c[4];
                float nc[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = min(beg + base + u, last);
                    c[u] = a.column_indices[e];
                    nc[u] = a.nc_edge[e];
                    w[u] = 1.f;
                    if constexpr (HAS_EW) w[u] = a.ew_edge[e];
                }
                float4 vl[4], vh[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned off = (unsigned)c[u] * (FIN * 4u) + 16u * kq;
                    vl[u] = ld_f4(a.x, off, 0);
                    vh[u] = ld_f4(a.x, off, 64);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) add_edge(base + u < deg, nc[u], w[u], vl[u], vh[u]);
            }
            plo = make_float4(plo.x * nr, plo.y * nr, plo.z * nr, plo.w * nr);
            phi = make_float4(phi.x * nr, phi.y * nr, phi.z * nr, phi.w * nr);
        }
        STGX_MARK(2);
        // H (four pieces): on its way under the x3 product
        float4 h[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) h[ct] = a.H ? ld_f4(a.H, oC, 64 * ct) : zero4;
        if constexpr (GATHER) {
            const unsigned oP = (row * FIN + 4u * kq) * 4u;
            st_f4(a.P, oP, 0, plo);
            st_f4(a.P, oP, 64, phi);
        }
        const Frag3 fp = frag_of(plo, phi);
        STGX_MARK(3);

        // ---- the products as ONE software-pipelined sequence of 26 steps of 12 matrix instructions (two column tiles x six split
        // terms).  Per step, in program order: the NEXT step's two weight triples are requested from LDS, a fence, this step's 12
        // instructions, and then a piece of vector work that does not depend on them -- the epilogue of an EARLIER step (clamp + mask,
        // sigmoid, tanh, the split of a finished piece) -- which sched_group_barrier deals into the matrix instructions' shadow, two
        // vector instructions behind each (what v_mfma_f32_16x16x32_bf16 hides: profiles/r04_coexec_f32mfma.jsonl).  Without this
        // every step began with an exposed LDS round trip and all vector work ran with the matrix pipe idle (tools/diag/stepf_trace.py:
        // 25 us for the products of a launch whose matrix instructions are 6 us).
        Frag3 wa = triple(lds, kFSecCat + 0, lane), wb = triple(lds, kFSecCat + 1, lane), na, nb;
        auto prefetch = [&](int i0, int i1) {
            na = triple(lds, i0, lane);
            nb = triple(lds, i1, lane);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto interleave = [&]() {                                              // 12 x (1 matrix instruction, 2 vector instructions)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        };
        auto step_end = [&]() {
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            wa = na, wb = nb;
        };
        auto bias2 = [&](int off, int ct, f32x4 &a0, f32x4 &a1) {
            a0 = to_x4(*reinterpret_cast<const float4 *>(sB + off + 16 * ct + 4 * kq));
            a1 = to_x4(*reinterpret_cast<const float4 *>(sB + off + 16 * (ct + 1) + 4 * kq));
        };
        auto mul4 = [](const float4 &p, const float4 &q) { return make_float4(p.x * q.x, p.y * q.y, p.z * q.z, p.w * q.w); };
        auto sig4 = [](const f32x4 &v) { return make_float4(sigmoid_(v[0]), sigmoid_(v[1]), sigmoid_(v[2]), sigmoid_(v[3])); };
        const unsigned oX = (row * (3u * C) + 4u * kq) * 4u;
        unsigned bad = 0u, gm[3] = {0u, 0u, 0u};
        // epilogue of an x3 step: two pieces of gate g -> clamp mask bits, the (unclamped) values stored
        auto x3_out = [&](int g, int cp, const f32x4 &a0, const f32x4 &a1) {
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const float4 v = to_f4(sidx ? a1 : a0);
                const float4 hg = make_float4(clamp3(v.x, lo_c, hi_c), clamp3(v.y, lo_c, hi_c), clamp3(v.z, lo_c, hi_c), clamp3(v.w, lo_c, hi_c));
                const unsigned m4 = (hg.x == v.x ? 1u : 0u) | (hg.y == v.y ? 2u : 0u) | (hg.z == v.z ? 4u : 0u) | (hg.w == v.w ? 8u : 0u);
                gm[g] |= m4 << (4 * (cp + sidx));
                if (a.x3) st_f4(a.x3, oX, 4 * (g * C + 16 * (cp + sidx)), v);       // kernel-uniform: x3 is optional
            }
        };
        auto gidx = [](int g, int ct, int kb) { return kFSecGate + (g * 4 + ct) * kFKg + kb; };

        // -- x3: six steps; H's fragments are split in the shadow of the first two
        f32x4 xa[2][2];                                                         // [parity of the step][piece]
        Frag3 fh0, fh1;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int g = st >> 1, cp = 2 * (st & 1);
            if (st < 5) prefetch(kFSecCat + (st + 1) * 2, kFSecCat + (st + 1) * 2 + 1);
            else prefetch(gidx(1, 0, 0), gidx(1, 1, 0));                        // gate r, column tiles 0 and 1, K-block 0
            bias2(g * C, cp, xa[st & 1][0], xa[st & 1][1]);
            mfma6x2w(xa[st & 1][0], xa[st & 1][1], wa, wb, fp);
            if (st > 0) x3_out((st - 1) >> 1, 2 * ((st - 1) & 1), xa[(st - 1) & 1][0], xa[(st - 1) & 1][1]);
            if (st == 0) fh0 = frag_of(h[0], h[1]);
            if (st == 1) fh1 = frag_of(h[2], h[3]);
            step_end();
        }
        // -- gates: acc[ct] = bg_g + Ag_g [P | X]; steps (cp, kb) with cp = 0, 2 and kb = 0 .. 2
        f32x4 ar[4], az[4], ah[4];
        float4 hr[4], z[4], rl[4];
        Frag3 fr0, fr1;
        auto xk = [&](int kb, const Frag3 &x1, const Frag3 &x2) -> const Frag3 & { return kb == 0 ? fp : (kb == 1 ? x1 : x2); };
        // gate r (its first step also finishes x3)
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int cp = 2 * (st / 3), kb = st % 3;
            if (st < 5) prefetch(gidx(1, 2 * ((st + 1) / 3), (st + 1) % 3), gidx(1, 2 * ((st + 1) / 3) + 1, (st + 1) % 3));
            else prefetch(gidx(0, 0, 0), gidx(0, 1, 0));
            if (kb == 0) bias2(3 * C + 1 * C, cp, ar[cp], ar[cp + 1]);
            mfma6x2w(ar[cp], ar[cp + 1], wa, wb, xk(kb, fh0, fh1));
            if (st == 0) {
                x3_out(2, 2, xa[1][0], xa[1][1]);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (a.mask) a.mask[row * 12u + 4u * g + kq] = gm[g];
                    bad |= gm[g] ^ 0xffffu;
                }
            }
            step_end();
        }
        if (bad) atomicOr(a.status, 1);
        // gate z; in its shadow R = sigmoid(.), H * R and the split of H * R, piece by piece
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int cp = 2 * (st / 3), kb = st % 3;
            if (st < 5) prefetch(gidx(0, 2 * ((st + 1) / 3), (st + 1) % 3), gidx(0, 2 * ((st + 1) / 3) + 1, (st + 1) % 3));
            else prefetch(gidx(2, 0, 0), gidx(2, 1, 0));
            if (kb == 0) bias2(3 * C + 0 * C, cp, az[cp], az[cp + 1]);
            mfma6x2w(az[cp], az[cp + 1], wa, wb, xk(kb, fh0, fh1));
            if (st < 4) {
                const float4 r = sig4(ar[st]);
                hr[st] = mul4(h[st], r);
                st_f4(a.R, oC, 64 * st, r);
                st_f4(a.HR, oC, 64 * st, hr[st]);
            }
            if (st == 4) fr0 = frag_of(hr[0], hr[1]);
            if (st == 5) fr1 = frag_of(hr[2], hr[3]);
            step_end();
        }
        STGX_MARK(5);
        // gate h; in its shadow Z = sigmoid(.)
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int cp = 2 * (st / 3), kb = st % 3;
            if (st < 5) prefetch(gidx(2, 2 * ((st + 1) / 3), (st + 1) % 3), gidx(2, 2 * ((st + 1) / 3) + 1, (st + 1) % 3));
            else prefetch(kFSecHead + 0, kFSecHead + 2);
            if (kb == 0) bias2(3 * C + 2 * C, cp, ah[cp], ah[cp + 1]);
            mfma6x2w(ah[cp], ah[cp + 1], wa, wb, xk(kb, fr0, fr1));
            if (st < 4) {
                z[st] = sig4(az[st]);
                st_f4(a.Z, oC, 64 * st, z[st]);
            }
            step_end();
        }
        STGX_MARK(6);
        // ---- Ht = tanh(.);  Hn = Z * H + (1 - Z) * Ht (nothing left to hide it under) -------------------------------------------
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float4 t = make_float4(tanh_(ah[ct][0]), tanh_(ah[ct][1]), tanh_(ah[ct][2]), tanh_(ah[ct][3]));
            const float4 hn = make_float4(z[ct].x * h[ct].x + (1.0f - z[ct].x) * t.x, z[ct].y * h[ct].y + (1.0f - z[ct].y) * t.y,
                                          z[ct].z * h[ct].z + (1.0f - z[ct].z) * t.z, z[ct].w * h[ct].w + (1.0f - z[ct].w) * t.w);
            st_f4(a.Ht, oC, 64 * ct, t);
            st_f4(a.Hn, oC, 64 * ct, hn);
            rl[ct] = make_float4(hn.x < 0.f ? 0.f : hn.x, hn.y < 0.f ? 0.f : hn.y, hn.z < 0.f ? 0.f : hn.z, hn.w < 0.f ? 0.f : hn.w);
        }
        STGX_MARK(7);
        // ---- head: y = relu(Hn) W1^T + b1;  y_out = y W2^T + b2;  partial[tile] = sum (y_out - target)^2 -------------------------
        if constexpr (HEAD != 0) {
            fr0 = frag_of(rl[0], rl[1]);
            f32x4 ay0, ay1;
            bias2(6 * C, 0, ay0, ay1);
            prefetch(kFSecHead + 1, kFSecHead + 3);
            mfma6x2w(ay0, ay1, wa, wb, fr0);
            fr1 = frag_of(rl[2], rl[3]);                                        // the second K-block's split under the first's products
            step_end();
            mfma6x2w(ay0, ay1, wa, wb, fr1);
            const unsigned oF = (row * FH + 4u * kq) * 4u;
            st_f4(a.y, oF, 0, to_f4(ay0));
            st_f4(a.y, oF, 64, to_f4(ay1));
            if constexpr (HEAD == 2) {
                const float4 w20 = *reinterpret_cast<const float4 *>(sB + 6 * C + FH + 4 * kq);
                const float4 w21 = *reinterpret_cast<const float4 *>(sB + 6 * C + FH + 16 + 4 * kq);
                float sdot = 0.f;
                sdot = sdot + ay0[0] * w20.x;
                sdot = sdot + ay0[1] * w20.y;
                sdot = sdot + ay0[2] * w20.z;
                sdot = sdot + ay0[3] * w20.w;
                sdot = sdot + ay1[0] * w21.x;
                sdot = sdot + ay1[1] * w21.y;
                sdot = sdot + ay1[2] * w21.z;
                sdot = sdot + ay1[3] * w21.w;
                sdot = sdot + __shfl_xor(sdot, 16, kWave);                     // the row's four kq lanes
                sdot = sdot + __shfl_xor(sdot, 32, kWave);
                const float yo = sdot + sB[6 * C + 2 * FH];
                const float tg = ld_f1(a.target, row * 4u);
                if (kq == 0) st_f1(a.y_out, row * 4u, yo);
                const float dlt = yo - tg;
                float sq = (idx < a.N && kq == 0) ? dlt * dlt : 0.f;
                sq = row16_sum(sq);                                            // lanes 0..15: the tile's 16 rows, in lane order
                if (lane == 15) a.partial[tile] = sq;
            }
        }
        STGX_MARK(8);
    }
}

template <bool HAS_EW, int HEAD, bool GATHER = true>
int launch_stepf_fwd(const FwdFArgs &a, hipStream_t stream)
{
    auto kern = tgcn_stepf_fwd_kernel<HAS_EW, HEAD, GATHER>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (!*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kFLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_step_fwd (folded form): %s", hipGetErrorString(e));
        *raised = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // one workgroup per CU as soon as there is a tile for each (tiles are dealt wave-major: at |V| = 50 K a CU runs 12-13 of its 16
    // waves; packing 16 tiles per workgroup instead left 60 CUs idle)
    const unsigned blocks = (unsigned)std::max(1, std::min(cus, a.num_tiles));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kFWaves * 64), kFLds, stream, a);
    return check_launch("stg_tgcn_step_fwd (folded form)");
}

}  // namespace
}  // namespace stg

#ifdef STG_STEPX_TRACE
extern "C" int stg_debug_set_stepf_trace_fwd(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(stg::g_stepx_trace), &buf, sizeof(buf)); }
#endif

// dispatch target of stg_tgcn_step_fwd (tgcn_step_fwd.hip) when the argument block carries folded gate weights
int stg_tgcn_stepf_fwd_launch(const stg_tgcn_step_fwd_args *p, void *stream_)
{
    using namespace stg;
    if (!p->w_fold || !p->b_fold || !p->fold_status)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_step_fwd (folded form): w_fold, b_fold and fold_status go together");
    if (p->N * (int64_t)(3 * kXC) >= (int64_t)1 << 30)
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_step_fwd (folded form): N too large for 32-bit byte offsets");
    FwdFArgs a{};
    a.row_offsets = p->row_offsets; a.column_indices = p->column_indices;
    a.nc_edge = p->norm_col_edge; a.ew_edge = p->ew_edge; a.norm = p->norm;
    a.x = p->x; a.H = p->H; a.target = p->target;
    a.Acat = p->WcatT; a.Ag = p->w_fold; a.W1 = p->W1;
    a.b3 = p->b3; a.bg = p->b_fold; a.b1 = p->b1; a.W2 = p->W2; a.b2 = p->b2;
    a.P = p->P; a.x3 = p->x3; a.Z = p->Z; a.R = p->R; a.Ht = p->Ht; a.Hn = p->Hn; a.HR = p->HR; a.y = p->y;
    a.y_out = p->y_out; a.partial = p->loss_partial; a.mask = p->clamp_mask; a.status = p->fold_status;
    a.N = p->N; a.lo = p->lo; a.hi = p->hi; a.num_tiles = (int)((p->N + 15) / 16);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (!p->x) return p->head == 1 ? launch_stepf_fwd<false, 1, false>(a, st) : launch_stepf_fwd<false, 2, false>(a, st);
    if (p->ew_edge) return p->head == 1 ? launch_stepf_fwd<true, 1>(a, st) : launch_stepf_fwd<true, 2>(a, st);
    return p->head == 1 ? launch_stepf_fwd<false, 1>(a, st) : launch_stepf_fwd<false, 2>(a, st);
}
