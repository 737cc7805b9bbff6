// Backward launch of the one-launch TGCN step on the matrix cores (3-term bf16 split): design notes in tgcn_stepx.hpp, the
// arithmetic it restates in tgcn_step.hpp / tgcn_step_bwd.hip (same inputs and outputs; reference: autograd through
// nn/pytorch/temporal/tgcn.py:21-55 and the heads of benchmarking/{static,dynamic}-temporal-tgcn/seastar/model.py).
//
// Per 16-row tile, wave (team, ct) owns columns 16 ct .. + 15 of every [16, C] tensor (and of each gate's third of da3):
//   I1  dyt = g_y + A_hat^T z_next + dyo W2 (+ the link loss's node side), by every wave for its lanes' row (8 values: cheap,
//       and it is the B operand all four need);  dHn += (Hn > 0)(dyt W1)[own columns];  GRU backward -> dhl, dzl (stored), dHa
//   I2  dCH = dhl Wh: first half -> da3[:, 2C + own] (clamp mask), second half = dHR -> drl (stored), dHa += dHR R
//   I3  dCZ = dzl Wz, dCR = drl Wr: first halves -> da3[:, own], da3[:, C + own]; dH = dHa + second halves
//   I4  z = da3 Wcat^T (K = 3C: two waves of the team, one per 16 output columns), the next tile's gather and operand loads
// The weights of the three gate products (as [2C outputs][C] A operands) live in registers, W1^T and Wcat^T fragments in LDS.
#include "tgcn_stepx.hpp"

namespace stg {
namespace {

struct BwdXArgs {
    const int *row_offsets, *column_indices;               // BACKWARD CSR (rows = sources)
    const float *nc_edge, *ew_edge, *norm;
    const float *zn, *gy, *dHn, *g_cost;
    const float *Z, *R, *Ht, *H, *Hn, *y_out, *target;
    const char *img;                                       // stg_tgcn_pack_weights_x3's backward image
    float *dzl, *drl, *dhl, *da3, *dH, *z, *dyt, *dyo;
    const unsigned char *mask;                             // the matrix-core forward launch's layout: tgcn_stepx_fwd.hip
    const int *link_row_ptr, *link_other, *link_eid;
    const float *link_y, *link_logits, *link_target;
    float link_inv_m;
    int64_t N;
    float two_over_n;
    int num_tiles;
};

constexpr int kBPLd = 36;
constexpr int kBActImg = 2 * kXTerms * kFragBytes;                      // a 64-column operand as fragments
constexpr int kBTeamBytes = 16 * kBPLd * 4 + 3 * kBActImg + 3 * kBActImg + kGatherTableBytes;   // Gbuf | dhl dzl drl | da3 (z, r, h thirds) | edge records
constexpr int kBwdLdsHead = 0;                                          // W1^T fragments [ct][t]
constexpr int kBwdLdsCat = kBwdLdsHead + 4 * kBwdHeadFrags * kFragBytes;    // Wcat fragments [ct'][b][t]
constexpr int kBwdLdsBias = kBwdLdsCat + 2 * kBwdCatFrags * kFragBytes;     // W2 [Fh]
constexpr int kBwdLdsTeam = kBwdLdsBias + 256;
constexpr int kBwdLds = kBwdLdsTeam + 2 * kBTeamBytes;
static_assert(4 * kBwdBiasFloats <= 256 && kBwdLds <= 160 * 1024, "LDS budget");

template <bool HAS_EW, int HEAD>
__global__ __launch_bounds__(512) void tgcn_stepx_bwd_kernel(const BwdXArgs a)
{
    constexpr int C = kXC, FIN = kXFin, FHW = kXFh;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int team = wave >> 2, ct = wave & 3;
    const int n16 = lane & 15, kq = lane >> 4;
    char *const sHead = lds + kBwdLdsHead, *const sCat = lds + kBwdLdsCat;
    const float *const sW2 = reinterpret_cast<const float *>(lds + kBwdLdsBias);
    char *const tm = lds + kBwdLdsTeam + team * kBTeamBytes;
    float *const Gbuf = reinterpret_cast<float *>(tm);
    char *const sFdh = tm + 16 * kBPLd * 4, *const sFdz = sFdh + kBActImg, *const sFdr = sFdz + kBActImg;
    char *const sFda = sFdr + kBActImg;                                      // da3: gate g at sFda + g * kBActImg
    uint4 *const trow = reinterpret_cast<uint4 *>(sFda + 3 * kBActImg) + (4 * ct + (lane >> 4)) * kGatherTableEdges;

    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.img + kBwdImgHead);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        constexpr int n16b = (kBwdImgBias - kBwdImgHead) / 16;              // W1^T + Wcat sections: contiguous in image and LDS
        static_assert(n16b % 512 == 0, "whole rounds of the workgroup");
        uint4 v[n16b / 512];
#pragma unroll
        for (int k = 0; k < n16b / 512; ++k) v[k] = src[threadIdx.x + 512 * k];
#pragma unroll
        for (int k = 0; k < n16b / 512; ++k) dst[threadIdx.x + 512 * k] = v[k];
        if (threadIdx.x < kBwdBiasFloats)
            reinterpret_cast<float *>(lds + kBwdLdsBias)[threadIdx.x] = reinterpret_cast<const float *>(a.img + kBwdImgBias)[threadIdx.x];
    }
    Frag3 Wg[3][2][2];                                                      // [gate][half of the 2C outputs][K-block]: 144 registers
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                Wg[g][h][b] = wfrag_load(a.img + kBwdImgGate, ct * kBwdGateFrags + ((g * 2 + h) * 2 + b) * kXTerms, lane);

    const int G = (int)gridDim.x, first = team * G + (int)blockIdx.x;
    const int n_mine = first < a.num_tiles ? (a.num_tiles - first + 2 * G - 1) / (2 * G) : 0;
    const int other = (1 - team) * G + (int)blockIdx.x;
    const int n_other = other < a.num_tiles ? (a.num_tiles - other + 2 * G - 1) / (2 * G) : 0;
    const int n0 = team == 0 ? n_mine : n_other, n1 = team == 0 ? n_other : n_mine;
    const int total = max(n0 ? 1 + 4 * n0 : 0, n1 ? 3 + 4 * n1 : 0);
    const bool want_z = a.z != nullptr, do_gather = a.zn != nullptr;         // block-uniform
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // carried across the intervals of a tile: the own pieces the tile reads, dHa, the clamp-mask nibbles
    float4 p_dhn = zero4, p_hn = zero4, p_z = zero4, p_t = zero4, p_h = zero4, p_r = zero4, dHa = zero4;
    unsigned mz = 0u, mr = 0u, mh = 0u;
    float p_yo = 0.f, p_tg = 0.f;                                            // HEAD == 2: the row's y_out and target

    // The gather of A_hat^T z_next for a tile's rows in its steps (RowGatherX); `gather_finish` ends it (Gbuf) and puts this
    // lane's pieces of the tile's saved tensors in flight.
    RowGatherX<HAS_EW> rg;
    const int c2 = lane & 15;
    auto grow = [&](int t) { return (int)min((int64_t)t * 16 + 4 * ct + (lane >> 4), a.N - 1); };
    auto gather_finish = [&](int t, float4 &o_dhn, float4 &o_hn, float4 &o_z, float4 &o_t, float4 &o_h, unsigned &o_mz, unsigned &o_mr,
                             unsigned &o_mh, float &o_yo, float &o_tg) {
        float2 p = make_float2(0.f, 0.f);
        if (do_gather) p = rg.run(trow, a.zn, a.column_indices, a.nc_edge, a.ew_edge, c2);
        *reinterpret_cast<float2 *>(Gbuf + (4 * ct + (lane >> 4)) * kBPLd + 2 * c2) = p;
        const unsigned row = (unsigned)min((int64_t)t * 16 + n16, a.N - 1);
        const unsigned oC = (row * C + 16u * ct + 4u * kq) * 4u;
        o_dhn = a.dHn ? ld_f4(a.dHn, oC, 0) : zero4;
        o_h = a.H ? ld_f4(a.H, oC, 0) : zero4;
        o_hn = ld_f4(a.Hn, oC, 0);
        o_z = ld_f4(a.Z, oC, 0);
        o_t = ld_f4(a.Ht, oC, 0);
        const unsigned char *mp = a.mask + (size_t)row * 48u + 4u * kq + ct;
        o_mz = mp[0], o_mr = mp[16], o_mh = mp[32];
        if constexpr (HEAD == 2) o_yo = ld_f1(a.y_out, row * 4u), o_tg = ld_f1(a.target, row * 4u);
    };
    auto masked = [](const f32x4 &v, unsigned m) {
        return make_float4((m & 1u) ? v[0] : 0.f, (m & 2u) ? v[1] : 0.f, (m & 4u) ? v[2] : 0.f, (m & 8u) ? v[3] : 0.f);
    };

    // one wait for the weight fragments here, none in the intervals (tgcn_stepx_fwd.hip)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int k = 0; k < kXTerms; ++k) asm volatile("" ::"v"(Wg[g][h][b].t[k]));
    __syncthreads();
    // Straight-line intervals per tile (no phase switch inside the loop): see tgcn_stepx_fwd.hip.
    int itc = 0;
    auto interval_end = [&]() {
        STGX_MARK(2 * itc + 1);
        lds_barrier();
        ++itc;
        STGX_MARK(2 * itc);
    };
    STGX_MARK(0);
    for (int k = 0; k < 2 * team; ++k) interval_end();
    if (n_mine > 0) {
        if (do_gather) {                                                     // the first tile's gather: its steps back to back
            rg.extent(a.row_offsets, a.norm, grow(first));
            rg.indices(a.column_indices, a.nc_edge, a.ew_edge, c2);
            rg.stash(trow, c2);
            wave_lds_fence();
        }
        gather_finish(first, p_dhn, p_hn, p_z, p_t, p_h, mz, mr, mh, p_yo, p_tg);
    }
    interval_end();
    for (int j = 0; j < n_mine; ++j) {
        const int tile = first + j * 2 * G;
        const unsigned row = (unsigned)min((int64_t)tile * 16 + n16, a.N - 1);
        const unsigned oC = (row * C + 16u * ct + 4u * kq) * 4u;
        const unsigned o3 = (row * (3u * C) + 16u * ct + 4u * kq) * 4u;
        const bool more = j + 1 < n_mine, pre = more && do_gather;           // uniform per team
        {
            // ---- dyt (this lane's two row pieces of its row), dHn through the head, GRU backward ------------------------
            const unsigned oF = (row * FHW + 4u * kq) * 4u;
            float4 gy[2] = {zero4, zero4};
            if (a.gy) gy[0] = ld_f4(a.gy, oF, 0), gy[1] = ld_f4(a.gy, oF, 64);
            if constexpr (HEAD == 1) {
                // node side of the link-prediction loss (stg_link_decode_bwd's sum, term for term and in its order)
                if (a.link_row_ptr) {
                    const int kb = a.link_row_ptr[row], ke = a.link_row_ptr[row + 1];
                    const float scale = a.g_cost[0] * a.link_inv_m;
                    const int kmax = wave_max_nonneg(ke - kb);
                    const int klast = max(ke - 1, 0);             // no guarded loads: lanes past their list re-read its last entry
                    for (int k0 = 0; k0 < kmax; k0 += 4) {
                        int e[4], o[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = min(kb + k0 + u, klast);
                            e[u] = a.link_eid[k];
                            o[u] = a.link_other[k];
                        }
                        float x[4], tg[4];
                        float4 yo[4][2];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            x[u] = a.link_logits[e[u]];
                            tg[u] = a.link_target[e[u]];
#pragma unroll
                            for (int q = 0; q < 2; ++q) yo[u][q] = ld_f4(a.link_y, ((unsigned)o[u] * FHW + 4u * kq) * 4u, 64 * q);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool ok = kb + k0 + u < ke;
                            const float sig = 1.0f / (1.0f + __expf(-x[u]));
                            const float coef = (sig - tg[u]) * scale;
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                gy[q] = make_float4(ok ? gy[q].x + coef * yo[u][q].x : gy[q].x, ok ? gy[q].y + coef * yo[u][q].y : gy[q].y,
                                                    ok ? gy[q].z + coef * yo[u][q].z : gy[q].z, ok ? gy[q].w + coef * yo[u][q].w : gy[q].w);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 gp = *reinterpret_cast<const float4 *>(Gbuf + n16 * kBPLd + 16 * q + 4 * kq);
                gy[q] = make_float4(gy[q].x + gp.x, gy[q].y + gp.y, gy[q].z + gp.z, gy[q].w + gp.w);
            }
            float dyo = 0.f;
            if constexpr (HEAD == 2) {
                dyo = ((p_yo - p_tg) * a.two_over_n) * a.g_cost[0];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 w2 = *reinterpret_cast<const float4 *>(sW2 + 16 * q + 4 * kq);
                    gy[q] = make_float4(gy[q].x + dyo * w2.x, gy[q].y + dyo * w2.y, gy[q].z + dyo * w2.z, gy[q].w + dyo * w2.w);
                }
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mfma6(acc, wfrag_load(sHead, ct * kXTerms, lane), frag_of(gy[0], gy[1]));
            const float4 g = make_float4(p_dhn.x + (p_hn.x > 0.f ? acc[0] : 0.f), p_dhn.y + (p_hn.y > 0.f ? acc[1] : 0.f),
                                         p_dhn.z + (p_hn.z > 0.f ? acc[2] : 0.f), p_dhn.w + (p_hn.w > 0.f ? acc[3] : 0.f));
            const float4 z = p_z, t = p_t, h = p_h;
            const float4 dhl = make_float4((g.x * (1.0f - z.x)) * (1.0f - t.x * t.x), (g.y * (1.0f - z.y)) * (1.0f - t.y * t.y),
                                           (g.z * (1.0f - z.z)) * (1.0f - t.z * t.z), (g.w * (1.0f - z.w)) * (1.0f - t.w * t.w));
            const float4 dz = make_float4((g.x * (h.x - t.x)) * (z.x * (1.0f - z.x)), (g.y * (h.y - t.y)) * (z.y * (1.0f - z.y)),
                                          (g.z * (h.z - t.z)) * (z.z * (1.0f - z.z)), (g.w * (h.w - t.w)) * (z.w * (1.0f - z.w)));
            dHa = make_float4(g.x * z.x, g.y * z.y, g.z * z.z, g.w * z.w);
            frag_store_piece(sFdh, ct, lane, split4(dhl));
            frag_store_piece(sFdz, ct, lane, split4(dz));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HEAD == 2) {
                if (ct == 0 && kq == 0) st_f1(a.dyo, row * 4u, dyo);
            }
            if (ct < 2) st_f4(a.dyt, oF, 64 * ct, ct == 0 ? gy[0] : gy[1]);       // waves 0 and 1 store one piece each
            st_f4(a.dhl, oC, 0, dhl);
            st_f4(a.dzl, oC, 0, dz);
            p_r = ld_f4(a.R, oC, 0);                                     // for the next interval: in flight across the barrier
            if (pre) rg.extent(a.row_offsets, a.norm, grow(tile + 2 * G));        // next tile's gather, first round trip
        }
        interval_end();
        {
            // ---- dCH = dhl Wh: d(hh) -> da3[:, 2C + own];  dHR -> drl, dHa ------------------------------------------------
            f32x4 aa = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const Frag3 f = frag_load(sFdh, b, lane);
                mfma6(aa, Wg[2][0][b], f);
                mfma6(ab, Wg[2][1][b], f);
            }
            const float4 d3 = masked(aa, mh);
            frag_store_piece(sFda + 2 * kBActImg, ct, lane, split4(d3));
            const float4 r = p_r, h = p_h;
            const float4 drl = make_float4((ab[0] * h.x) * (r.x * (1.0f - r.x)), (ab[1] * h.y) * (r.y * (1.0f - r.y)),
                                           (ab[2] * h.z) * (r.z * (1.0f - r.z)), (ab[3] * h.w) * (r.w * (1.0f - r.w)));
            dHa = make_float4(dHa.x + ab[0] * r.x, dHa.y + ab[1] * r.y, dHa.z + ab[2] * r.z, dHa.w + ab[3] * r.w);
            frag_store_piece(sFdr, ct, lane, split4(drl));
            __builtin_amdgcn_sched_barrier(0);
            st_f4(a.da3, o3, 4 * (2 * C), d3);
            st_f4(a.drl, oC, 0, drl);
            __builtin_amdgcn_sched_barrier(0);
            if (pre) rg.indices(a.column_indices, a.nc_edge, a.ew_edge, c2);        // second round trip (needs the extent: an interval old)
        }
        interval_end();
        {
            // ---- dCZ = dzl Wz, dCR = drl Wr: first halves -> da3, second halves -> dH (dCZ's first, then dCR's) -----------
            f32x4 za = {0.f, 0.f, 0.f, 0.f}, zb = za, ra = za, rb = za;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const Frag3 fz = frag_load(sFdz, b, lane);
                mfma6(za, Wg[0][0][b], fz);
                mfma6(zb, Wg[0][1][b], fz);
                const Frag3 fr = frag_load(sFdr, b, lane);
                mfma6(ra, Wg[1][0][b], fr);
                mfma6(rb, Wg[1][1][b], fr);
            }
            const float4 dzc = masked(za, mz), drc = masked(ra, mr);
            frag_store_piece(sFda, ct, lane, split4(dzc));
            frag_store_piece(sFda + kBActImg, ct, lane, split4(drc));
            const float4 d1 = make_float4(dHa.x + zb[0], dHa.y + zb[1], dHa.z + zb[2], dHa.w + zb[3]);
            __builtin_amdgcn_sched_barrier(0);
            if (pre) rg.stash(trow, c2);                                           // the edge records (an interval old) into the table
            __builtin_amdgcn_sched_barrier(0);
            st_f4(a.da3, o3, 0, dzc);
            st_f4(a.da3, o3, 4 * C, drc);
            st_f4(a.dH, oC, 0, make_float4(d1.x + rb[0], d1.y + rb[1], d1.z + rb[2], d1.w + rb[3]));
        }
        interval_end();
        {
            // ---- z = da3 Wcat^T (two waves of the team, 16 output columns each, in turn), then the next tile ---------------
            const bool z_wave = want_z && (ct >> 1) == (j & 1);
            const int ft = ct & 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (z_wave) {
#pragma unroll
                for (int b = 0; b < 6; ++b)
                    mfma6(acc, wfrag_load(sCat, (ft * 6 + b) * kXTerms, lane), frag_load(sFda + (b >> 1) * kBActImg, b & 1, lane));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) gather_finish(tile + 2 * G, p_dhn, p_hn, p_z, p_t, p_h, mz, mr, mh, p_yo, p_tg);
            __builtin_amdgcn_sched_barrier(0);
            if (z_wave) st_f4(a.z, (row * FIN + 4u * kq) * 4u, 64 * ft, to_f4(acc));
        }
        interval_end();
    }
    while (itc < total) interval_end();
}

template <bool HAS_EW, int HEAD>
int launch_stepx_bwd(const BwdXArgs &a, hipStream_t stream)
{
    auto kern = tgcn_stepx_bwd_kernel<HAS_EW, HEAD>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (!*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_step_bwd (matrix-core form): %s", hipGetErrorString(e));
        *raised = true;
    }
    const unsigned blocks = (unsigned)std::max(1, std::min(256, (a.num_tiles + 1) / 2));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), kBwdLds, stream, a);
    return check_launch("stg_tgcn_step_bwd (matrix-core form)");
}

}  // namespace
}  // namespace stg

#ifdef STG_STEPX_TRACE
extern "C" int stg_debug_set_stepx_trace_bwd(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(stg::g_stepx_trace), &buf, sizeof(buf)); }
#endif

// dispatch target of stg_tgcn_step_bwd (tgcn_step_bwd.hip) when the argument block carries a weight image
int stg_tgcn_stepx_bwd_launch(const stg_tgcn_step_bwd_args *p, void *stream_)
{
    using namespace stg;
    BwdXArgs a{};
    a.row_offsets = p->row_offsets; a.column_indices = p->column_indices;
    a.nc_edge = p->norm_col_edge; a.ew_edge = p->ew_edge; a.norm = p->norm;
    a.zn = p->zn; a.gy = p->g_y; a.dHn = p->dHn; a.g_cost = p->g_cost;
    a.Z = p->Z; a.R = p->R; a.Ht = p->Ht; a.H = p->H; a.Hn = p->Hn; a.y_out = p->y_out; a.target = p->target;
    a.img = static_cast<const char *>(p->w_image);
    a.dzl = p->dzl; a.drl = p->drl; a.dhl = p->dhl; a.da3 = p->da3; a.dH = p->dH; a.z = p->z; a.dyt = p->dyt; a.dyo = p->dyo;
    a.mask = reinterpret_cast<const unsigned char *>(p->clamp_mask);
    if (p->link_row_ptr) {
        a.link_row_ptr = p->link_row_ptr; a.link_other = p->link_other; a.link_eid = p->link_eid;
        a.link_y = p->link_y; a.link_logits = p->link_logits; a.link_target = p->link_target; a.link_inv_m = p->link_inv_m;
    }
    a.N = p->N; a.two_over_n = 2.0f / (float)p->N; a.num_tiles = (int)((p->N + 15) / 16);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const bool ew = p->zn && p->ew_edge;
    if (ew) return p->head == 1 ? launch_stepx_bwd<true, 1>(a, st) : launch_stepx_bwd<true, 2>(a, st);
    return p->head == 1 ? launch_stepx_bwd<false, 1>(a, st) : launch_stepx_bwd<false, 2>(a, st);
}
