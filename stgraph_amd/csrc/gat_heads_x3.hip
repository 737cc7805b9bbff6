// The per-head products of a GAT layer at H heads of D = 64 over fin = 64 inputs (BASELINE configs[2]: 8 x 64 over 64) on the bf16
// matrix cores, every product as the 3-term bf16 split with fp32 accumulation of rowgemm_x3.hip (same arithmetic, same lane
// layout, same whole-line loads and stores), FOUR heads per launch: their weight images (4 x 24 KB) are the workgroup's LDS, a
// wave takes a pair of 16-row tiles through the four heads before it moves on.
//
//   stg_gat_fc_out / stg_gat_fc_fwd (nn/pytorch/static/gat_conv.py:43-48: `self.fc(h).view(-1, H, D)`, the attention projections;
//   the layer's output side in its uniform-attention form): out[:, h, :] = x W_h^T from ONE read of the pair's rows of x, with
//   elu(out) and / or el, er from the same accumulators.  The fp32-instruction kernel (gat_fc.hip) is bound by its 512
//   v_mfma_f32_16x16x4_f32 per tile: 310 us for the 1.1 GB it moves at |V| = 256 K; here the product is 96 bf16 instructions per
//   head and tile pair and the launch is bound by its stores.
//
//   stg_gat_bwd_prepass_heads (the uniform-attention backward, kernels.gat_bwd_uniform): the per-vertex pass of the factored
//   backward unit (gat.hip, gat_bwd_prepass_h8d64_kernel: g_pre = g * elu'(out), pack = S | sum_d (g_pre / S) out,
//   grad_er = slope (g_pre . out - pack S)) AND gW[h, v, :] = W_h^T g_pre[v, h, :] from one read of g and out -- the two used to be
//   a 1.5 GB pass followed by eight launches that read g_pre again (294 + 246 us).
#include <algorithm>

#include "bf16_split.hpp"

namespace stg {
namespace {

constexpr int kHxWaves = 8, kHxD = 64, kHxKB = 2, kHxCT = 4, kHxHG = 4;
constexpr int kHxHeadFrags = kHxCT * kHxKB;                                        // fragment triples of one head's 64 x 64 weights
constexpr size_t kHxImgBytes = (size_t)kHxHG * kHxHeadFrags * kXTerms * kFragBytes;   // 96 KB

__device__ __forceinline__ void hx_mfma6x4(f32x4 &a0, f32x4 &a1, f32x4 &b0, f32x4 &b1, const Frag3 &wa, const Frag3 &wb, const Frag3 &x0,
                                           const Frag3 &x1)
{
#define STG_HX_ROUND(TW, TX)                                                              \
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.t[TW], x0.t[TX], a0, 0, 0, 0);        \
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.t[TW], x1.t[TX], a1, 0, 0, 0);        \
    b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb.t[TW], x0.t[TX], b0, 0, 0, 0);        \
    b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb.t[TW], x1.t[TX], b1, 0, 0, 0);
    STG_HX_ROUND(0, 2)                                  // small terms first (rowgemm_x3.hip)
    STG_HX_ROUND(2, 0)
    STG_HX_ROUND(1, 1)
    STG_HX_ROUND(0, 1)
    STG_HX_ROUND(1, 0)
    STG_HX_ROUND(0, 0)
#undef STG_HX_ROUND
}

// Whole 128-byte lines (rowgemm_x3.hip, LINES): lanes (r, kq) and (r + 8, kq) load the two 64-byte halves of row r with one
// instruction and of row r + 8 with a second, and trade the pieces that belong to the partner's row (one DPP row rotation).
// lower lane (r): lo = its first load, hi = the upper lane's first load; upper lane (r + 8): lo = its second, hi = the lower's second
__device__ __forceinline__ void hx_trade(const float4 &own_if_lower, const float4 &own_if_upper, float4 &lo, float4 &hi)
{
    auto one = [&](float a1, float a2, float &l, float &h) {
        const int i1 = __float_as_int(a1), i2 = __float_as_int(a2);
        l = __int_as_float(__builtin_amdgcn_update_dpp(i1, i2, 0xE4, 0xF, 0xC, false));              // lanes 8-15: a2
        const int t = __builtin_amdgcn_update_dpp(i1, i1, 0x128, 0xF, 0x3, false);                  // lanes 0-7: partner's a1
        h = __int_as_float(__builtin_amdgcn_update_dpp(t, i2, 0x128, 0xF, 0xC, false));             // lanes 8-15: partner's a2
    };
    one(own_if_lower.x, own_if_upper.x, lo.x, hi.x);
    one(own_if_lower.y, own_if_upper.y, lo.y, hi.y);
    one(own_if_lower.z, own_if_upper.z, lo.z, hi.z);
    one(own_if_lower.w, own_if_upper.w, lo.w, hi.w);
}

// the way back: a lane's column tiles 2 c (a) and 2 c + 1 (bq) of its row -> s1 = the 16 bytes it stores into row r8 (first store),
// s2 = into row r8 + 8 (second store), both at columns 32 c + 16 upper + 4 kq
__device__ __forceinline__ void hx_untrade(const float4 &a, const float4 &bq, float4 &s1, float4 &s2)
{
    auto one = [&](float av, float bv, float &o1, float &o2) {
        const int ia = __float_as_int(av), ib = __float_as_int(bv);
        o1 = __int_as_float(__builtin_amdgcn_update_dpp(ia, ib, 0x128, 0xF, 0xC, false));   // upper: partner's 2 c + 1
        o2 = __int_as_float(__builtin_amdgcn_update_dpp(ib, ia, 0x128, 0xF, 0x3, false));   // lower: partner's 2 c
    };
    one(a.x, bq.x, s1.x, s2.x);
    one(a.y, bq.y, s1.y, s2.y);
    one(a.z, bq.z, s1.z, s2.z);
    one(a.w, bq.w, s1.w, s2.w);
}

__device__ __forceinline__ float4 hx_load16(const __amdgpu_buffer_rsrc_t &rs, int64_t elem)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * (int64_t)sizeof(float)), 0, 0);
    const unsigned u0 = v[0], u1 = v[1], u2 = v[2], u3 = v[3];
    return make_float4(__uint_as_float(u0), __uint_as_float(u1), __uint_as_float(u2), __uint_as_float(u3));
}

__device__ __forceinline__ float hx_elu(float x) { return x <= 0.f ? expf(x) - 1.0f : x; }          // torch's elu (gat.hip: elu1)
__device__ __forceinline__ float hx_elu_bwd(float g, float x) { return x <= 0.f ? g * expf(x) : g; }   // (gat.hip: elu1_bwd)

// The weight image of head group [h0, h0 + 4): fragment triple (hh, ct, b) at ((hh * 8 + ct * 2 + b) * 3 + term) x 1 KB.
// TRANS (fc: W [H * 64][64] in torch Linear layout, output column m = row of W_h): A[m][k] = W[(h 64 + m) 64 + k];
// else (the backward: output column m = input feature f, k = d): A[m][k] = W[(h 64 + k) 64 + m].
template <bool TRANS>
__device__ __forceinline__ void hx_stage_weights(char *img, const float *__restrict__ W, int h0, int lane, int wave)
{
    const int n16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int fi = 0; fi < kHxHG * kHxHeadFrags / kHxWaves; ++fi) {
        const int f = fi * kHxWaves + wave;
        const int hh = f >> 3, ct = (f & 7) >> 1, b = f & 1, m = 16 * ct + n16, k0 = 32 * b + 4 * kq;      // k of (b, kq, i) = xcol(b, kq, i)
        const float *wh = W + (int64_t)(h0 + hh) * kHxD * kHxD;
        float4 lo, hi;
        if constexpr (TRANS) {
            lo = *reinterpret_cast<const float4 *>(wh + m * kHxD + k0);
            hi = *reinterpret_cast<const float4 *>(wh + m * kHxD + k0 + 16);
        } else {
            const float *w = wh + k0 * kHxD + m;
            lo = make_float4(w[0], w[kHxD], w[2 * kHxD], w[3 * kHxD]);
            hi = make_float4(w[16 * kHxD], w[17 * kHxD], w[18 * kHxD], w[19 * kHxD]);
        }
        const Frag3 fr = frag_of(lo, hi);
#pragma unroll
        for (int t = 0; t < kXTerms; ++t)
            *reinterpret_cast<uint4 *>(img + ((size_t)(f * kXTerms + t) * kWave + lane) * 16) = __builtin_bit_cast(uint4, fr.t[t]);
    }
}

// acc[t][ct] += the head's 64 x 64 product for the pair's two tiles
__device__ __forceinline__ void hx_product(f32x4 (&acc)[2][kHxCT], const char *sec, const Frag3 (&xf)[kHxKB][2], int lane)
{
#pragma unroll
    for (int b = 0; b < kHxKB; ++b) {
        Frag3 wa = wfrag_load(sec, (0 * kHxKB + b) * kXTerms, lane), wb = wfrag_load(sec, (1 * kHxKB + b) * kXTerms, lane);
#pragma unroll
        for (int ct = 0; ct < kHxCT; ct += 2) {
            Frag3 na = wa, nb = wb;
            if (ct + 2 < kHxCT) {
                na = wfrag_load(sec, ((ct + 2) * kHxKB + b) * kXTerms, lane);
                nb = wfrag_load(sec, ((ct + 3) * kHxKB + b) * kXTerms, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            hx_mfma6x4(acc[0][ct], acc[1][ct], acc[0][ct + 1], acc[1][ct + 1], wa, wb, xf[b][0], xf[b][1]);
            __builtin_amdgcn_sched_barrier(0);
            wa = na, wb = nb;
        }
    }
}

struct HeadsFcArgs {
    const float *x, *W, *attn_l, *attn_r;
    float *out, *act, *el, *er;
    int64_t N;
    int num_pairs, h0, H;
};

// STORE: out[:, h, :]; ELU: act = elu(out); PROJ: el / er
template <bool STORE, bool ELU, bool PROJ>
__global__ __launch_bounds__(kHxWaves *kWave, 1) void gat_heads_fc_x3_kernel(const HeadsFcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *img = lds;
    float *av = reinterpret_cast<float *>(lds + kHxImgBytes);                    // attn_l [4][64] | attn_r [4][64] of the group
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4, r8 = n16 & 7, upper = n16 >> 3;
    const int total = (int)gridDim.x * kHxWaves;
    int pair = wave * (int)gridDim.x + (int)blockIdx.x;
    const int64_t N = a.N;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)(N * kHxD * (int64_t)sizeof(float)), 0x00020000);
    float4 ring[kHxKB][2][2];                                                    // [K-block][tile][load]
    auto load_pair = [&](int p) {
#pragma unroll
        for (int b = 0; b < kHxKB; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t base = (int64_t)p * 32 + 16 * t;
                const int64_t ra = std::min<int64_t>(base + r8, N - 1), rb = std::min<int64_t>(base + 8 + r8, N - 1);
                ring[b][t][0] = hx_load16(rsX, ra * kHxD + 32 * b + 16 * upper + 4 * kq);
                ring[b][t][1] = hx_load16(rsX, rb * kHxD + 32 * b + 16 * (1 - upper) + 4 * kq);
            }
    };
    if (pair < a.num_pairs) load_pair(pair);
    hx_stage_weights<true>(img, a.W, a.h0, lane, wave);
    if constexpr (PROJ) {
        for (int i = threadIdx.x; i < kHxHG * kHxD; i += kHxWaves * kWave) {
            av[i] = a.attn_l[a.h0 * kHxD + i];
            av[kHxHG * kHxD + i] = a.attn_r[a.h0 * kHxD + i];
        }
    }
    __syncthreads();
    const int ldy = a.H * kHxD;
    for (; pair < a.num_pairs; pair += total) {
        const int next = std::min(pair + total, a.num_pairs - 1);
        Frag3 xf[kHxKB][2];
#pragma unroll
        for (int b = 0; b < kHxKB; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 lo, hi;
                hx_trade(ring[b][t][0], ring[b][t][1], lo, hi);
                xf[b][t] = frag_of(lo, hi);
            }
        load_pair(next);                                                         // under the four heads' products
#pragma unroll 1
        for (int hh = 0; hh < kHxHG; ++hh) {
            f32x4 acc[2][kHxCT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < kHxCT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            hx_product(acc, img + (size_t)hh * kHxHeadFrags * kXTerms * kFragBytes, xf, lane);
            const int col0 = (a.h0 + hh) * kHxD;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (PROJ) {
                    // the sums of gat_fc.hip's epilogue, term for term: 16 multiply-adds per lane, then the row's four kq lanes
                    float l = 0.f, r = 0.f;
#pragma unroll
                    for (int ct = 0; ct < kHxCT; ++ct) {
                        const float4 p = *reinterpret_cast<const float4 *>(av + hh * kHxD + 16 * ct + 4 * kq);
                        const float4 q = *reinterpret_cast<const float4 *>(av + kHxHG * kHxD + hh * kHxD + 16 * ct + 4 * kq);
                        l += acc[t][ct][0] * p.x + acc[t][ct][1] * p.y + acc[t][ct][2] * p.z + acc[t][ct][3] * p.w;
                        r += acc[t][ct][0] * q.x + acc[t][ct][1] * q.y + acc[t][ct][2] * q.z + acc[t][ct][3] * q.w;
                    }
                    l = l + __shfl_xor(l, 16, 64);
                    r = r + __shfl_xor(r, 16, 64);
                    l = l + __shfl_xor(l, 32, 64);
                    r = r + __shfl_xor(r, 32, 64);
                    if (kq == 0) {
                        const int64_t row = std::min<int64_t>((int64_t)pair * 32 + 16 * t + n16, N - 1);
                        a.el[row * a.H + a.h0 + hh] = l;
                        a.er[row * a.H + a.h0 + hh] = r;
                    }
                }
                if constexpr (STORE || ELU) {
                    const int64_t base = (int64_t)pair * 32 + 16 * t;
                    const int64_t o1 = std::min<int64_t>(base + r8, N - 1) * ldy + col0 + 16 * upper + 4 * kq;
                    const int64_t o2 = std::min<int64_t>(base + 8 + r8, N - 1) * ldy + col0 + 16 * upper + 4 * kq;
#pragma unroll
                    for (int c = 0; c < kHxCT / 2; ++c) {
                        float4 s1, s2;
                        hx_untrade(to_f4(acc[t][2 * c]), to_f4(acc[t][2 * c + 1]), s1, s2);
                        if constexpr (STORE) {
                            *reinterpret_cast<float4 *>(a.out + o1 + 32 * c) = s1;
                            *reinterpret_cast<float4 *>(a.out + o2 + 32 * c) = s2;
                        }
                        if constexpr (ELU) {
                            *reinterpret_cast<float4 *>(a.act + o1 + 32 * c) = make_float4(hx_elu(s1.x), hx_elu(s1.y), hx_elu(s1.z), hx_elu(s1.w));
                            *reinterpret_cast<float4 *>(a.act + o2 + 32 * c) = make_float4(hx_elu(s2.x), hx_elu(s2.y), hx_elu(s2.z), hx_elu(s2.w));
                        }
                    }
                }
            }
        }
    }
}

// 1: P = sum_d (g_pre / S) out with a division per element, as gat_bwd_prepass_h8d64_kernel forms it; 0: (g_pre . out) / S -- equal to
// fp32 rounding, 31 divisions per lane and head fewer
#ifndef STG_HX_DIV_PER_ELEMENT
#define STG_HX_DIV_PER_ELEMENT 0
#endif

struct HeadsBwdArgs {
    const float *g, *outp, *S, *W;
    float *g_pre, *pack, *grad_er, *gW;
    int64_t N;
    int num_pairs, h0, H;
    float slope;
};

// ELU: g is the gradient of elu(out) (g_pre = g * elu'(out) is written); else g is the gradient of out (g_pre unused)
template <bool ELU>
__global__ __launch_bounds__(kHxWaves *kWave, 1) void gat_heads_bwd_x3_kernel(const HeadsBwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *img = lds;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4, r8 = n16 & 7, upper = n16 >> 3;
    const int total = (int)gridDim.x * kHxWaves;
    int pair = wave * (int)gridDim.x + (int)blockIdx.x;
    const int64_t N = a.N;
    const int HD = a.H * kHxD;
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.g), 0, (int)(N * HD * (int64_t)sizeof(float)), 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.outp), 0, (int)(N * HD * (int64_t)sizeof(float)), 0x00020000);
    float4 G[kHxKB][2][2], O[kHxKB][2][2];                                       // [K-block][tile][load]: g and out of ONE head
    float sv[2][2];                                                              // S of rows (tile, load) for that head
    auto rows_of = [&](int p, int t, int64_t &ra, int64_t &rb) {
        const int64_t base = (int64_t)p * 32 + 16 * t;
        ra = std::min<int64_t>(base + r8, N - 1), rb = std::min<int64_t>(base + 8 + r8, N - 1);
    };
    auto load_item = [&](int p, int hh) {
        const int col0 = (a.h0 + hh) * kHxD;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int64_t ra, rb;
            rows_of(p, t, ra, rb);
            sv[t][0] = a.S[ra * a.H + a.h0 + hh];
            sv[t][1] = a.S[rb * a.H + a.h0 + hh];
#pragma unroll
            for (int b = 0; b < kHxKB; ++b) {
                const int64_t ea = ra * HD + col0 + 32 * b + 16 * upper + 4 * kq, eb = rb * HD + col0 + 32 * b + 16 * (1 - upper) + 4 * kq;
                G[b][t][0] = hx_load16(rsG, ea);
                G[b][t][1] = hx_load16(rsG, eb);
                O[b][t][0] = hx_load16(rsO, ea);
                O[b][t][1] = hx_load16(rsO, eb);
            }
        }
    };
    if (pair < a.num_pairs) load_item(pair, 0);
    hx_stage_weights<false>(img, a.W, a.h0, lane, wave);
    __syncthreads();
    for (; pair < a.num_pairs; pair += total) {
        const int next = std::min(pair + total, a.num_pairs - 1);
#pragma unroll 1
        for (int hh = 0; hh < kHxHG; ++hh) {
            const int h = a.h0 + hh, col0 = h * kHxD;
            // ---- the per-vertex pass on the pieces as loaded (a piece = four consecutive columns of row ra (load 0) / rb (load 1))
            float p[2][2], q[2][2];
            Frag3 xf[kHxKB][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int64_t ra, rb;
                rows_of(pair, t, ra, rb);
                p[t][0] = p[t][1] = q[t][0] = q[t][1] = 0.f;
#pragma unroll
                for (int b = 0; b < kHxKB; ++b) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float4 gv = G[b][t][j];
                        const float4 ov = O[b][t][j];
                        if constexpr (ELU) {
                            gv = make_float4(hx_elu_bwd(gv.x, ov.x), hx_elu_bwd(gv.y, ov.y), hx_elu_bwd(gv.z, ov.z), hx_elu_bwd(gv.w, ov.w));
                            const int64_t e = (j == 0 ? ra : rb) * HD + col0 + 32 * b + 16 * (j == 0 ? upper : 1 - upper) + 4 * kq;
                            *reinterpret_cast<float4 *>(a.g_pre + e) = gv;
                            G[b][t][j] = gv;
                        }
#if STG_HX_DIV_PER_ELEMENT
                        const float sh = sv[t][j];
                        p[t][j] = p[t][j] + (gv.x / sh) * ov.x;
                        p[t][j] = p[t][j] + (gv.y / sh) * ov.y;
                        p[t][j] = p[t][j] + (gv.z / sh) * ov.z;
                        p[t][j] = p[t][j] + (gv.w / sh) * ov.w;
#endif
                        q[t][j] = q[t][j] + gv.x * ov.x;
                        q[t][j] = q[t][j] + gv.y * ov.y;
                        q[t][j] = q[t][j] + gv.z * ov.z;
                        q[t][j] = q[t][j] + gv.w * ov.w;
                    }
                    float4 lo, hi;
                    hx_trade(G[b][t][0], G[b][t][1], lo, hi);
                    xf[b][t] = frag_of(lo, hi);
                }
                // a row's 64 columns of the head sit in the eight lanes (upper, kq) of its r8: lanes ^ 8, ^ 16, ^ 32
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float pv = p[t][j], qv = q[t][j];
                    pv = pv + __shfl_xor(pv, 8, 64);
                    qv = qv + __shfl_xor(qv, 8, 64);
                    pv = pv + __shfl_xor(pv, 16, 64);
                    qv = qv + __shfl_xor(qv, 16, 64);
                    pv = pv + __shfl_xor(pv, 32, 64);
                    qv = qv + __shfl_xor(qv, 32, 64);
                    if (lane < 8) {                                              // upper = 0, kq = 0: r8 = lane
                        const int64_t row = j == 0 ? ra : rb;
                        const float sh = sv[t][j];
#if !STG_HX_DIV_PER_ELEMENT
                        pv = qv / sh;                                            // sum_d (g / S) out = (g . out) / S: ONE division per (vertex, head)
#endif
                        a.pack[row * 16 + h] = sh;
                        a.pack[row * 16 + 8 + h] = pv;
                        if (a.grad_er) a.grad_er[row * a.H + h] = sh == 0.f ? 0.f : a.slope * (qv - pv * sh);   // S = 0: no in-edge
                    }
                }
            }
            // the next head's (or the next pair's first head's) rows: in flight under this head's products
            if (hh + 1 < kHxHG) load_item(pair, hh + 1);
            else load_item(next, 0);
            f32x4 acc[2][kHxCT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < kHxCT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            hx_product(acc, img + (size_t)hh * kHxHeadFrags * kXTerms * kFragBytes, xf, lane);
            float *y = a.gW + (int64_t)h * N * kHxD;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int64_t ra, rb;
                rows_of(pair, t, ra, rb);
                float *d1 = y + ra * kHxD + 16 * upper + 4 * kq, *d2 = y + rb * kHxD + 16 * upper + 4 * kq;
#pragma unroll
                for (int c = 0; c < kHxCT / 2; ++c) {
                    float4 s1, s2;
                    hx_untrade(to_f4(acc[t][2 * c]), to_f4(acc[t][2 * c + 1]), s1, s2);
                    *reinterpret_cast<float4 *>(d1 + 32 * c) = s1;
                    *reinterpret_cast<float4 *>(d2 + 32 * c) = s2;
                }
            }
        }
    }
}

inline bool hx_shape_ok(int64_t N, int H, int D, int fin)
{
    return D == kHxD && fin == kHxD && H >= kHxHG && H % kHxHG == 0 && N > 0 && N * (int64_t)H * D < ((int64_t)1 << 29);
}

template <typename Kernel>
int hx_raise_lds(Kernel kernel, size_t lds, PerDeviceOnce &once, const char *what)
{
    bool *done = once.slot();
    if (!*done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
        *done = true;
    }
    return 0;
}

inline unsigned hx_grid(int64_t pairs)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (unsigned)std::min<int64_t>((pairs + kHxWaves - 1) / kHxWaves, cus);
}

}  // namespace

// (declared in stg_common.hpp) whether the split form takes this layer: the shape, 16-byte aligned operands, knob "rowgemm_x3"
bool gat_heads_x3_wanted(int64_t N, int H, int D, int fin)
{
    return hx_shape_ok(N, H, D, fin) && tuning().rowgemm_x3 != 1 && tuning().rowgemm_x3 != 3;
}

// out (nullable) [N, H, 64] = x W^T, act (nullable) = elu of it, el / er (both or neither) the attention projections
int gat_heads_fc_x3_launch(const char *what, const float *x, const float *W, const float *attn_l, const float *attn_r, float *out, float *act,
                           float *el, float *er, int64_t N, int H, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = kHxImgBytes + 2 * kHxHG * kHxD * sizeof(float);
    const int64_t pairs = (N + 31) / 32;
    const bool proj = el != nullptr;
    static PerDeviceOnce once[6];
    for (int h0 = 0; h0 < H; h0 += kHxHG) {
        const HeadsFcArgs a{x, W, attn_l, attn_r, out, act, el, er, N, (int)pairs, h0, H};
        int rc = 0;
#define STG_HX_FC(S_, E_, P_, I_)                                                                                               \
    do {                                                                                                                        \
        rc = hx_raise_lds(gat_heads_fc_x3_kernel<S_, E_, P_>, lds, once[I_], what);                                             \
        if (rc == 0) hipLaunchKernelGGL((gat_heads_fc_x3_kernel<S_, E_, P_>), dim3(hx_grid(pairs)), dim3(kHxWaves * kWave), lds, st, a); \
    } while (0)
        if (proj && out) STG_HX_FC(true, false, true, 0);
        else if (proj) STG_HX_FC(false, false, true, 1);
        else if (out && act) STG_HX_FC(true, true, false, 2);
        else if (out) STG_HX_FC(true, false, false, 3);
        else return fail(STG_ERR_INVALID_ARGUMENT, "%s: nothing to compute", what);
#undef STG_HX_FC
        if (rc != 0) return rc;
    }
    return check_launch(what);
}

int gat_heads_bwd_x3_launch(const char *what, const float *S, const float *outp, const float *g, float *g_pre, float *pack, float *grad_er,
                            const float *W, float *gW, int64_t N, int H, float slope, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = kHxImgBytes;
    const int64_t pairs = (N + 31) / 32;
    static PerDeviceOnce once[2];
    for (int h0 = 0; h0 < H; h0 += kHxHG) {
        const HeadsBwdArgs a{g, outp, S, W, g_pre, pack, grad_er, gW, N, (int)pairs, h0, H, slope};
        int rc;
        if (g_pre) {
            rc = hx_raise_lds(gat_heads_bwd_x3_kernel<true>, lds, once[0], what);
            if (rc == 0) hipLaunchKernelGGL(gat_heads_bwd_x3_kernel<true>, dim3(hx_grid(pairs)), dim3(kHxWaves * kWave), lds, st, a);
        } else {
            rc = hx_raise_lds(gat_heads_bwd_x3_kernel<false>, lds, once[1], what);
            if (rc == 0) hipLaunchKernelGGL(gat_heads_bwd_x3_kernel<false>, dim3(hx_grid(pairs)), dim3(kHxWaves * kWave), lds, st, a);
        }
        if (rc != 0) return rc;
    }
    return check_launch(what);
}

}  // namespace stg

extern "C" int stg_gat_bwd_prepass_heads_supported(int64_t N, int32_t H, int32_t D, int32_t fin)
{
    return stg::gat_heads_x3_wanted(N, H, D, fin) ? 1 : 0;
}

extern "C" int stg_gat_bwd_prepass_heads(const float *S, const float *out, const float *g, float *g_pre, float *pack, float *grad_er,
                                         const float *W, float *gW, int64_t N, int32_t H, int32_t D, int32_t fin, float slope, void *stream)
{
    using namespace stg;
    if (!hx_shape_ok(N, H, D, fin))
        return fail(STG_ERR_UNSUPPORTED, "stg_gat_bwd_prepass_heads: H %% 4 == 0 heads of D = 64 over fin = 64, N H D < 2^29 (got N=%lld H=%d D=%d fin=%d)",
                    (long long)N, H, D, fin);
    if (!S || !out || !g || !pack || !W || !gW) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_prepass_heads: NULL pointer argument");
    const uintptr_t align = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g_pre) |
                            reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(gW);
    if (align % 16 != 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_prepass_heads: out, g, g_pre, W, gW must be 16-byte aligned");
    return gat_heads_bwd_x3_launch("stg_gat_bwd_prepass_heads", S, out, g, g_pre, pack, grad_er, W, gW, N, H, slope, stream);
}
