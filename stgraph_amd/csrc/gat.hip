// Fused GAT units for gfx950 (MI355X) -- stgraph_hip.h "fused GAT".
//
// Same wave64 row mapping as gcn_agg.hip: G lanes own one CSR row, a batch of G
// (column, eid) pairs is fetched one-per-lane and broadcast, UNROLL neighbour-row
// gathers are in flight per row before the first is consumed, accumulation is
// sequential in CSR order with one fp32 accumulator per (row, feature).
//
//   k0     : edge score a = exp(leaky_relu(s - s)), per-dst sum S        (lane = head)
//   k1     : out = sum_e (A/S) * feat[u]                                 (lane = VEC features)
//   bwd    : grad_feat, grad_el and the per-edge scalar T = sum_d t      (src-major CSR)
//   bwd_er : grad_er[v] = sum_{e in in(v)} T[e]                          (dst-major CSR)
//
// The reference accumulates grad_el / grad_er with fp32 atomicAdd from every
// feature lane.  Here the sum over the D lanes of a
// head is an in-wave butterfly (ds_swizzle/DPP via __shfl_xor) and the sum over a
// vertex's in-edges is a second, tiny dst-major pass: no atomics, run-to-run
// deterministic.
#include "stg_common.hpp"

namespace stg {

template <int G>
__device__ __forceinline__ int gbcast_i(int v, int src)
{
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, src);
    else if constexpr (G == 1) return v;
    else return __shfl(v, src, G);
}

struct RowInfo {
    int r, beg, deg, max_deg;
    bool valid;
};

// (vblock: the workgroup's position in the grid that has one workgroup per row block -- blockIdx.x unless a launch with FEWER
//  workgroups walks the row blocks with a stride: the fallback launches of the uniform-attention form, whose 64 K workgroups
//  otherwise cost 16 us to dispatch only to return on the device flag)
template <int LOG2G>
__device__ __forceinline__ RowInfo row_prologue(const int *__restrict__ row_offsets,
                                                const int *__restrict__ node_ids, int N, int vblock = -1)
{
    constexpr int G = 1 << LOG2G;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_global = (vblock < 0 ? (int)blockIdx.x : vblock) * kWavesPerBlock + (threadIdx.x >> 6);
    const int idx = wave_global * (kWave / G) + (lane >> LOG2G);
    RowInfo ri{0, 0, 0, 0, idx < N};
    if (ri.valid) {
        ri.r = node_ids ? node_ids[idx] : idx;
        ri.beg = row_offsets[ri.r];
        ri.deg = row_offsets[ri.r + 1] - ri.beg;
    }
    ri.max_deg = __builtin_amdgcn_readfirstlane(wave_max(ri.deg));
    return ri;
}

// ------------------------------------------------------------------------------ "all ones" flag
// The vertex function's `emb - max([emb])` (reference gat_conv.py:50) is s - s with s = el[u] + er[v]:
// +0 for every finite s, hence A[e,h] = exp(leaky(0)) = 1.0f EXACTLY and S[v,h] = the in-degree, unless some score is
// inf / NaN.  *flag (zeroed by the caller) is set when any |el| or |er| is not below 1e38 -- then, and only then, can a
// sum be non-finite.  With the flag clear K0 writes S = min(deg, 2^24) (what the sequential fp32 sum of ones gives)
// without visiting an edge, and K1 / K2 take A as the constant 1.0f instead of loading E*H of them (in K2 through a
// scattered edge id): the same bits as the emitted units in every case, 0.3 ms of the cfg3 layer.
__global__ __launch_bounds__(kBlock) void gat_score_flag_kernel(const float *__restrict__ el, const float *__restrict__ er,
                                                                int64_t n, int *__restrict__ flag)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        bad |= !(fabsf(el[i]) < 1e38f) || !(fabsf(er[i]) < 1e38f);
    if (bad) atomicOr(flag, 1);
}

__device__ __forceinline__ bool all_ones(const int *__restrict__ flag)
{
    return flag != nullptr && __builtin_amdgcn_readfirstlane(*flag) == 0;
}

// ------------------------------------------------------------------------------ K0
template <int LOG2G>
__global__ __launch_bounds__(kBlock) void gat_k0_kernel(
    const float *__restrict__ el, const float *__restrict__ er, float *__restrict__ A,
    float *__restrict__ S, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ eids,
    const int *__restrict__ node_ids, int N, int H, int H_active, float slope, const int *__restrict__ flag)
{
    constexpr int G = 1 << LOG2G;
    constexpr int U = G < 4 ? G : 4;
    const int j = threadIdx.x & (G - 1);
    const RowInfo ri = row_prologue<LOG2G>(row_offsets, node_ids, N);
    if (all_ones(flag)) {                                   // kernel-uniform
        const float s = (float)min(ri.deg, 1 << 24);
        for (int h = j; h < H_active; h += G)
            if (ri.valid) S[(int64_t)ri.r * H + h] = s;
        return;
    }

    for (int hbase = 0; hbase < H_active; hbase += G) {
        const int h = hbase + j;
        const bool hok = h < H_active;
        const float erv = (ri.valid && hok) ? er[(int64_t)ri.r * H + h] : 0.f;
        float acc = 0.f;
        for (int base = 0; base < ri.max_deg; base += G) {
            const int cnt = ri.deg - base;
            const int cnt_max = min(G, ri.max_deg - base);
            int c = 0, ev = 0;
            if (j < cnt) {
                c = column_indices[ri.beg + base + j];
                ev = eids[ri.beg + base + j];
            }
            for (int k = 0; k < cnt_max; k += U) {
                float elv[U];
                int ek[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ck = gbcast_i<G>(c, kk & (G - 1));
                    ek[u] = gbcast_i<G>(ev, kk & (G - 1));
                    elv[u] = (kk < cnt && hok) ? el[(int64_t)ck * H + h] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < cnt && hok) {
                        const float s = elv[u] + erv;       // Add(el_inb, er_cen)
                        const float z = s - s;              // Sub(emb, max([emb])) == emb - emb
                        const float l = z > 0 ? z : slope * z;
                        const float a = expf(l);
                        A[(int64_t)ek[u] * H + h] = a;
                        acc = acc + a;
                    }
                }
            }
        }
        if (ri.valid && hok) S[(int64_t)ri.r * H + h] = acc;
    }
}

// ------------------------------------------------------------------------------ K1
enum { kK1Always = 0, kK1IfUniform = 1, kK1UnlessUniform = 2 };
constexpr unsigned kFallbackGrid = 4096;           // workgroups of a launch that almost always returns on the device flag

// torch's elu (alpha = 1) as its device kernel forms it: x <= 0 ? exp(x) - 1 : x
__device__ __forceinline__ float elu1(float x) { return x <= 0.f ? expf(x) - 1.0f : x; }
// ... and its elu_backward from the pre-activation value: g * (x <= 0 ? exp(x) : 1)
__device__ __forceinline__ float elu1_bwd(float g, float x) { return x <= 0.f ? g * expf(x) : g; }

template <int VEC, int LOG2G, int CHUNKS, int UNROLL>
__global__ __launch_bounds__(kBlock) void gat_k1_kernel(
    const float *__restrict__ A, const float *__restrict__ S, const float *__restrict__ feat,
    float *__restrict__ out, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ eids,
    const int *__restrict__ node_ids, int N, int H, int D, int HD_active, const int *__restrict__ flag,
    int s_stride, int when, float *__restrict__ act_out, int nblocks)
{
    constexpr int G = 1 << LOG2G;
    constexpr int U = UNROLL < G ? UNROLL : G;
    const int j = threadIdx.x & (G - 1);
    const int HD = H * D;
    const bool ones = all_ones(flag);
    // when: kK1Always | kK1IfUniform (the narrow-width pass of the uniform-attention form: only meaningful when every
    // A is 1.0f) | kK1UnlessUniform (the full-width pass that replaces its result otherwise): kernel-uniform exits
    if ((when == kK1IfUniform && !ones) || (when == kK1UnlessUniform && ones)) return;
    // (nblocks row blocks on gridDim.x workgroups: one each, except for the capped grid of the kK1UnlessUniform launch)
    for (int vb = (int)blockIdx.x; vb < nblocks; vb += (int)gridDim.x) {
    const RowInfo ri = row_prologue<LOG2G>(row_offsets, node_ids, N, vb);

    for (int fbase = 0; fbase < HD_active; fbase += G * VEC * CHUNKS) {
        float acc[CHUNKS][VEC];
        float sv[CHUNKS];
        int foff[CHUNKS], hh[CHUNKS];
        bool fok[CHUNKS];
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            foff[ch] = fbase + (ch * G + j) * VEC;
            fok[ch] = foff[ch] < HD_active;
            hh[ch] = fok[ch] ? foff[ch] / D : 0;
            sv[ch] = (ri.valid && fok[ch]) ? S[(int64_t)ri.r * s_stride + hh[ch]] : 1.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[ch][i] = 0.f;
        }
        for (int base = 0; base < ri.max_deg; base += G) {
            const int cnt = ri.deg - base;
            const int cnt_max = min(G, ri.max_deg - base);
            int c = 0, ev = 0;
            if (j < cnt) {
                c = column_indices[ri.beg + base + j];
                if (!ones) ev = eids[ri.beg + base + j];
            }
            for (int k = 0; k < cnt_max; k += U) {
                float v[U][CHUNKS][VEC];
                float a[U][CHUNKS];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ck = gbcast_i<G>(c, kk & (G - 1));
                    const int ek = gbcast_i<G>(ev, kk & (G - 1));
                    const float *row = feat + (int64_t)ck * HD;
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if (kk < cnt && fok[ch]) {
                            a[u][ch] = ones ? 1.0f : A[(int64_t)ek * H + hh[ch]];
                            vec_load<VEC>(v[u][ch], row + foff[ch]);
                        } else {
                            a[u][ch] = 0.f;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[u][ch][i] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k + u < cnt) {
#pragma unroll
                        for (int ch = 0; ch < CHUNKS; ++ch) {
                            if (fok[ch]) {
                                const float alpha = a[u][ch] / sv[ch];          // TrueDiv(c, s)
#pragma unroll
                                for (int i = 0; i < VEC; ++i)
                                    acc[ch][i] = acc[ch][i] + alpha * v[u][ch][i];
                            }
                        }
                    }
                }
            }
        }
        if (ri.valid) {
            float *orow = out + (int64_t)ri.r * HD;
#pragma unroll
            for (int ch = 0; ch < CHUNKS; ++ch)
                if (fok[ch]) vec_store<VEC>(orow + foff[ch], acc[ch]);
            if (act_out) {                                   // kernel-uniform: the layer's ELU beside the pre-activation rows
                float *arow = act_out + (int64_t)ri.r * HD;
#pragma unroll
                for (int ch = 0; ch < CHUNKS; ++ch) {
                    if (!fok[ch]) continue;
                    float y[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) y[i] = elu1(acc[ch][i]);
                    vec_store<VEC>(arow + foff[ch], y);
                }
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------ K2
// Sum of `p` over the LH consecutive lanes that share a head.  LH is a power of two
// (butterfly, every lane gets the total) or arbitrary (LDS staging, ascending-lane
// order; only the head's first lane gets the total).
template <bool POW2>
__device__ __forceinline__ float head_sum(float p, int LH, float *lds_wave, int lane, int j, int G)
{
    if constexpr (POW2) {
        for (int off = LH >> 1; off > 0; off >>= 1) p = p + __shfl_xor(p, off, 64);
        return p;
    } else {
        lds_wave[lane] = p;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        float s = 0.f;
        for (int q = 0; q < LH && j + q < G; ++q) s = s + lds_wave[lane + q];
        __builtin_amdgcn_wave_barrier();
        return s;
    }
}

template <int VEC, int LOG2G, int CHUNKS, int UNROLL, bool POW2>
__global__ __launch_bounds__(kBlock) void gat_bwd_kernel(
    const float *__restrict__ A, const float *__restrict__ S, const float *__restrict__ outp,
    const float *__restrict__ g, const float *__restrict__ el, const float *__restrict__ er,
    const float *__restrict__ feat, float *__restrict__ grad_feat, float *__restrict__ grad_el,
    float *__restrict__ T, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ eids,
    const int *__restrict__ node_ids, int N, int H, int D, int HD_active, float slope, const int *__restrict__ flag)
{
    constexpr int G = 1 << LOG2G;
    constexpr int U = UNROLL < G ? UNROLL : G;
    __shared__ float lds[POW2 ? 1 : kBlock];
    float *lds_wave = lds + (POW2 ? 0 : (threadIdx.x & ~(kWave - 1)));
    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & (G - 1);
    const int HD = H * D;
    const int LH = D / VEC;                       // lanes per head (host guarantees D % VEC == 0)
    const RowInfo ri = row_prologue<LOG2G>(row_offsets, node_ids, N);
    const bool ones = all_ones(flag);

    for (int fbase = 0; fbase < HD_active; fbase += G * VEC * CHUNKS) {
        float a13[CHUNKS][VEC], a29[CHUNKS][VEC], fu[CHUNKS][VEC];
        float elu[CHUNKS];
        int foff[CHUNKS], hh[CHUNKS];
        bool fok[CHUNKS], lead[CHUNKS];
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            foff[ch] = fbase + (ch * G + j) * VEC;
            fok[ch] = foff[ch] < HD_active;
            hh[ch] = foff[ch] < HD ? foff[ch] / D : 0;
            lead[ch] = foff[ch] < HD && (foff[ch] % D) == 0;
            elu[ch] = (ri.valid && fok[ch]) ? el[(int64_t)ri.r * H + hh[ch]] : 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { a13[ch][i] = 0.f; a29[ch][i] = 0.f; fu[ch][i] = 0.f; }
            if (ri.valid && fok[ch]) vec_load<VEC>(fu[ch], feat + (int64_t)ri.r * HD + foff[ch]);
        }
        for (int base = 0; base < ri.max_deg; base += G) {
            const int cnt = ri.deg - base;
            const int cnt_max = min(G, ri.max_deg - base);
            int c = 0, ev = 0;
            if (j < cnt) {
                c = column_indices[ri.beg + base + j];
                ev = eids[ri.beg + base + j];
            }
            for (int k = 0; k < cnt_max; k += U) {
                float gv[U][CHUNKS][VEC], ov[U][CHUNKS][VEC];
                float av[U][CHUNKS], sv[U][CHUNKS], erv[U][CHUNKS];
                int ek[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ck = gbcast_i<G>(c, kk & (G - 1));
                    ek[u] = gbcast_i<G>(ev, kk & (G - 1));
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if (kk < cnt && fok[ch]) {
                            av[u][ch] = ones ? 1.0f : A[(int64_t)ek[u] * H + hh[ch]];
                            sv[u][ch] = S[(int64_t)ck * H + hh[ch]];
                            erv[u][ch] = er[(int64_t)ck * H + hh[ch]];
                            vec_load<VEC>(gv[u][ch], g + (int64_t)ck * HD + foff[ch]);
                            vec_load<VEC>(ov[u][ch], outp + (int64_t)ck * HD + foff[ch]);
                        } else {
                            av[u][ch] = 0.f; sv[u][ch] = 1.f; erv[u][ch] = 0.f;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) { gv[u][ch][i] = 0.f; ov[u][ch][i] = 0.f; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = k + u < cnt;           // uniform within the row group
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        float p = 0.f;
                        if (on && fok[ch]) {
                            const float V3 = av[u][ch], V4 = sv[u][ch];
                            const float V5 = V3 / V4;
                            const float V0 = elu[ch] + erv[u][ch];
                            const float V1 = V0 - V0;
                            const float V14 = 1.0f / V4;
                            const float V24 = V1 > 0 ? 1.0f : slope;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                const float V8 = gv[u][ch][i];
                                a13[ch][i] = a13[ch][i] + V8 * V5;
                                const float V15 = (V8 * fu[ch][i]) * V14;
                                const float V17 = (V8 / V4) * ov[u][ch][i];
                                const float V22 = V15 + (-1.0f * V17);
                                const float V25 = (V22 * V3) * V24;
                                a29[ch][i] = a29[ch][i] + V25;
                                p = p + V25;
                            }
                        }
                        // every lane of the wave takes part (inactive lanes contribute 0)
                        const float tot = head_sum<POW2>(p, LH, lds_wave, lane, j, G);
                        if (on && lead[ch] && foff[ch] < HD_active) T[(int64_t)ek[u] * H + hh[ch]] = tot;
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) p = p + a29[ch][i];
            const float tot = head_sum<POW2>(p, LH, lds_wave, lane, j, G);
            if (ri.valid && lead[ch] && foff[ch] < HD_active) grad_el[(int64_t)ri.r * H + hh[ch]] = tot;
            if (ri.valid && fok[ch]) vec_store<VEC>(grad_feat + (int64_t)ri.r * HD + foff[ch], a13[ch]);
        }
    }
}

// ----------------------------------------------------------------- K2, factored form
// The per-lane term of K2 is  t_d = ((g_d f_d)(1/S) - (g_d/S) o_d) * A * slope  with g, o read at the
// edge's TARGET v and f at its source u.  Summed over the head:
//     sum_d t_d = A * slope * ( (1/S) * sum_d g_d f_d  -  P[v,h] ),   P[v,h] = sum_d (g_d/S) o_d
// P depends on the target only, so it is computed once per vertex (gat_bwd_prepass_kernel, reads g
// and out once) instead of once per edge -- the out[v] row gather (half of K2's traffic: 16.4 GB of
// 34.7 GB at |E| = 8M, H*D = 512) disappears.  The reference sums these terms with atomicAdd in an
// undefined order, so the regrouping stays inside its own run-to-run spread (tested to 1e-4).
// (z = s - s is 0 or NaN, so the LeakyReLU derivative `z > 0 ? 1 : slope` is always `slope`.)
// The same regrouping over a TARGET's in-edges gives grad_er without T:  sum_{e in in(v)} T[e,h]
//     = slope * ( (1/S) * sum_e A_e (g . f_e)  -  P * sum_e A_e )  =  slope * ( g . out  -  P * S )      per (v, h),
// because sum_e A_e f_e = S out (K1) and sum_e A_e = S (K0): a per-vertex quantity the prepass has in its registers.  In
// exact arithmetic it is 0 (the softmax gradient sums to zero under a constant LeakyReLU slope); the reference's
// atomicAdd sums leave rounding noise there (goldens: max |grad_er| 4.5e-7 next to grad_el of 4.7), and so does this
// form.  With `grad_er` given, T is not needed (pass NULL): the E*H scattered stores and the dst-major pass over T go.
// The prepass also leaves 1.0f / S[v,h] behind P (P is [2][N][H]): the per-edge kernel's T term then loads the
// quotient the emitted unit forms per edge (same value) -- an IEEE fp32 division is ~ 10 VALU instructions per wave,
// per edge and 256-float chunk.  alpha = A / S stays a division: grad_feat remains bit-identical to the emitted unit.
template <int VEC, int LOG2G, int CHUNKS, bool POW2>
__global__ __launch_bounds__(kBlock) void gat_bwd_prepass_kernel(
    const float *__restrict__ S, const float *__restrict__ outp, const float *__restrict__ g,
    float *__restrict__ P, int N, int H, int D, float *__restrict__ grad_er, float slope, float *__restrict__ g_pre)
{
    float *__restrict__ invS = P + (int64_t)N * H;
    constexpr int G = 1 << LOG2G;
    __shared__ float lds[POW2 ? 1 : kBlock];
    float *lds_wave = lds + (POW2 ? 0 : (threadIdx.x & ~(kWave - 1)));
    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & (G - 1);
    const int HD = H * D;
    const int LH = D / VEC;
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int v = wave_global * (kWave / G) + (lane >> LOG2G);
    const bool valid = v < N;
    for (int fbase = 0; fbase < HD; fbase += G * VEC * CHUNKS) {
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            const int foff = fbase + (ch * G + j) * VEC;
            const bool fok = valid && foff < HD;
            const int h = foff < HD ? foff / D : 0;
            float p = 0.f, q = 0.f;
            if (fok) {
                float gv[VEC], ov[VEC];
                vec_load<VEC>(gv, g + (int64_t)v * HD + foff);
                vec_load<VEC>(ov, outp + (int64_t)v * HD + foff);
                if (g_pre) {                          // kernel-uniform: g is the gradient of elu(out); see elu1_bwd
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gv[i] = elu1_bwd(gv[i], ov[i]);
                    vec_store<VEC>(g_pre + (int64_t)v * HD + foff, gv);
                }
                const float s = S[(int64_t)v * H + h];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    p = p + (gv[i] / s) * ov[i];
                    q = q + gv[i] * ov[i];
                }
            }
            const float tot = head_sum<POW2>(p, LH, lds_wave, lane, j, G);
            float go = 0.f;
            if (grad_er) go = head_sum<POW2>(q, LH, lds_wave, lane, j, G);          // kernel-uniform
            if (fok && (foff % D) == 0) {
                const float s = S[(int64_t)v * H + h];
                P[(int64_t)v * H + h] = tot;
                invS[(int64_t)v * H + h] = 1.0f / s;
                // = sum over v's in-edges of T (see below); S = 0 <=> no in-edge (A = exp(0) or NaN): an empty sum
                if (grad_er) grad_er[(int64_t)v * H + h] = s == 0.f ? 0.f : slope * (go - tot * s);
            }
        }
    }
}

// ROW16: a head is exactly one DPP row of 16 lanes (D = 16 * VEC, e.g. D = 64 at 16 B per lane): its per-edge dot
// product is four DPP adds instead of four ds_bpermute_b32 + adds.
template <int VEC, int LOG2G, int CHUNKS, int UNROLL, bool POW2, bool ROW16>
__global__ __launch_bounds__(kBlock) void gat_bwd_fact_kernel(
    const float *__restrict__ A, const float *__restrict__ S, const float *__restrict__ P,
    const float *__restrict__ g, const float *__restrict__ feat, float *__restrict__ grad_feat,
    float *__restrict__ grad_el, float *__restrict__ T, const int *__restrict__ row_offsets,
    const int *__restrict__ column_indices, const int *__restrict__ eids,
    const int *__restrict__ node_ids, int N, int H, int D, float slope, const int *__restrict__ flag)
{
    constexpr int G = 1 << LOG2G;
    constexpr int U = UNROLL < G ? UNROLL : G;
    __shared__ float lds[POW2 ? 1 : kBlock];
    float *lds_wave = lds + (POW2 ? 0 : (threadIdx.x & ~(kWave - 1)));
    const int lane = threadIdx.x & (kWave - 1);
    const int j = lane & (G - 1);
    const int HD = H * D;
    const int LH = D / VEC;
    const RowInfo ri = row_prologue<LOG2G>(row_offsets, node_ids, N);
    const bool ones = all_ones(flag);
    const float *__restrict__ invS = P + (int64_t)N * H;
    // Every A = 1.0f and no T wanted (grad_er comes from the per-vertex pass): alpha = 1.0f / S is the very quotient the T term
    // multiplies the dot product by, so  sum_e (g_e . f)(1 / S_e) = f . sum_e g_e alpha_e = f . grad_feat[u]  -- ONE dot product
    // per row after the loop instead of one (and its cross-lane sum) per edge; grad_el = slope (f . grad_feat - sum_e P_e).
    // Kernel-uniform.  (Narrow rows -- the 1-head output layer of the GAT model, 4 lanes per row -- are bound by these
    // per-edge instructions, not by their 64-byte gathers: 305 -> 2xx us at cfg3's second layer.)
    const bool lite = ones && T == nullptr;

    for (int fbase = 0; fbase < HD; fbase += G * VEC * CHUNKS) {
        float a13[CHUNKS][VEC], fu[CHUNKS][VEC], gel[CHUNKS];
        int foff[CHUNKS], hh[CHUNKS];
        bool fok[CHUNKS], lead[CHUNKS];
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            foff[ch] = fbase + (ch * G + j) * VEC;
            fok[ch] = foff[ch] < HD;
            hh[ch] = fok[ch] ? foff[ch] / D : 0;
            lead[ch] = fok[ch] && (foff[ch] % D) == 0;
            gel[ch] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { a13[ch][i] = 0.f; fu[ch][i] = 0.f; }
            if (ri.valid && fok[ch]) vec_load<VEC>(fu[ch], feat + (int64_t)ri.r * HD + foff[ch]);
        }
        for (int base = 0; base < ri.max_deg; base += G) {
            const int cnt = ri.deg - base;
            const int cnt_max = min(G, ri.max_deg - base);
            int c = 0, ev = 0;
            if (j < cnt) {
                c = column_indices[ri.beg + base + j];
                ev = eids[ri.beg + base + j];
            }
            for (int k = 0; k < cnt_max; k += U) {
                float gv[U][CHUNKS][VEC];
                float av[U][CHUNKS], sv[U][CHUNKS], iv[U][CHUNKS], pv[U][CHUNKS];
                int ek[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ck = gbcast_i<G>(c, kk & (G - 1));
                    ek[u] = gbcast_i<G>(ev, kk & (G - 1));
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        if (kk < cnt && fok[ch]) {
                            av[u][ch] = ones ? 1.0f : A[(int64_t)ek[u] * H + hh[ch]];
                            sv[u][ch] = S[(int64_t)ck * H + hh[ch]];
                            iv[u][ch] = lite ? 1.f : invS[(int64_t)ck * H + hh[ch]];          // 1.0f / S
                            pv[u][ch] = P[(int64_t)ck * H + hh[ch]];
                            vec_load<VEC>(gv[u][ch], g + (int64_t)ck * HD + foff[ch]);
                        } else {
                            av[u][ch] = 0.f; sv[u][ch] = 1.f; iv[u][ch] = 1.f; pv[u][ch] = 0.f;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) gv[u][ch][i] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = k + u < cnt;
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ++ch) {
                        float p = 0.f;
                        if (on && fok[ch]) {
                            const float alpha = av[u][ch] / sv[u][ch];
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                a13[ch][i] = a13[ch][i] + gv[u][ch][i] * alpha;
                                p = __builtin_fmaf(gv[u][ch][i], fu[ch][i], p);     // (a regrouped sum either way)
                            }
                        }
                        if (lite) {
                            if (on && lead[ch]) gel[ch] = gel[ch] + pv[u][ch];     // sum of P over the out-edges
                            continue;
                        }
                        float dot;
                        if constexpr (ROW16) dot = row16_sum_lane0(p);
                        else dot = head_sum<POW2>(p, LH, lds_wave, lane, j, G);
                        if (on && lead[ch]) {
                            const float tv = ((dot * iv[u][ch] - pv[u][ch]) * av[u][ch]) * slope;
                            if (T) T[(int64_t)ek[u] * H + hh[ch]] = tv;
                            gel[ch] = gel[ch] + tv;
                        }
                    }
                }
            }
        }
        if (lite) {
#pragma unroll
            for (int ch = 0; ch < CHUNKS; ++ch) {
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) p = __builtin_fmaf(a13[ch][i], fu[ch][i], p);
                float dot;
                if constexpr (ROW16) dot = row16_sum_lane0(p);
                else dot = head_sum<POW2>(p, LH, lds_wave, lane, j, G);
                gel[ch] = (dot - gel[ch]) * slope;
            }
        }
#pragma unroll
        for (int ch = 0; ch < CHUNKS; ++ch) {
            if (ri.valid && lead[ch]) grad_el[(int64_t)ri.r * H + hh[ch]] = gel[ch];
            if (ri.valid && fok[ch]) vec_store<VEC>(grad_feat + (int64_t)ri.r * HD + foff[ch], a13[ch]);
        }
    }
}

// ----------------------------------------------- K2, factored form, H = 8 heads of D = 64 (BASELINE configs[2])
// One wave per source row, 16 B per lane, two 256-float chunks = 4 heads each; a head is one DPP row of 16 lanes.
// In the general kernel above every lane loads its head's A, S, 1/S and P per edge AND chunk: eight vector loads = eight
// L1 misses per edge next to the eight 128-byte lines of the gathered g row (measured: 3.03 ms with them, 2.53 without;
// K1, which needs none of them, runs the same gather in 2.33).  Here the target's 24 scalars sit in ONE 128-byte line
// (pack[v] = S[8] | P[8], 64 bytes, written by the prepass) that lanes 0-15 load with one instruction, A[e, 0..7]
// is one more (lanes 0-7), and alpha = A / S, 1.0f / S (the same divisions) and the T term are formed ONCE per edge on lanes 0-7
// (lane = head); ds_bpermute_b32 hands alpha to the lanes of its head and brings the eight per-head dot products (DPP
// row sums, valid in each row's lane 0) back.  T and grad_el leave as 32 contiguous bytes from lanes 0-7.
__global__ __launch_bounds__(kBlock) void gat_bwd_prepass_h8d64_kernel(
    const float *__restrict__ S, const float *__restrict__ outp, const float *__restrict__ g,
    float *__restrict__ pack, int N, float *__restrict__ grad_er, float slope, float *__restrict__ g_pre)
{
    constexpr int H = 8, HD = 512;
    const int lane = threadIdx.x & (kWave - 1);
    const int v = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (v >= N) return;                              // whole wave
    float s = 1.f;
    if (lane < H) s = S[(int64_t)v * H + lane];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        float gv[4], ov[4];
        vec_load<4>(gv, g + (int64_t)v * HD + ch * 256 + lane * 4);
        vec_load<4>(ov, outp + (int64_t)v * HD + ch * 256 + lane * 4);
        if (g_pre) {                                  // kernel-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = elu1_bwd(gv[i], ov[i]);
            vec_store<4>(g_pre + (int64_t)v * HD + ch * 256 + lane * 4, gv);
        }
        // this lane's head: 4 ch + lane / 16; its S from lane (4 ch + lane / 16)
        const float sh = __int_as_float(__builtin_amdgcn_ds_bpermute((4 * ch + (lane >> 4)) * 4, __float_as_int(s)));
        float p = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p = p + (gv[i] / sh) * ov[i];
            q = q + gv[i] * ov[i];
        }
        const float tot = row16_sum_lane0(p);
        const float go = row16_sum_lane0(q);
        if ((lane & 15) == 0) {
            pack[(int64_t)v * 16 + 8 + 4 * ch + (lane >> 4)] = tot;
            if (grad_er) grad_er[(int64_t)v * H + 4 * ch + (lane >> 4)] = sh == 0.f ? 0.f : slope * (go - tot * sh);   // S = 0: no in-edge
        }
    }
    if (lane < H) pack[(int64_t)v * 16 + lane] = s;
}

template <int UNROLL>
__global__ __launch_bounds__(kBlock) void gat_bwd_fact_h8d64_kernel(
    const float *__restrict__ A, const float *__restrict__ pack, const float *__restrict__ g,
    const float *__restrict__ feat, float *__restrict__ grad_feat, float *__restrict__ grad_el,
    float *__restrict__ T, const int *__restrict__ row_offsets, const int *__restrict__ column_indices,
    const int *__restrict__ eids, const int *__restrict__ node_ids, int N, float slope, const int *__restrict__ flag,
    int unless_uniform, int nblocks)
{
    constexpr int H = 8, HD = 512, U = UNROLL;
    const int lane = threadIdx.x & (kWave - 1);
    const bool ones = all_ones(flag);
    // unless_uniform: the launch of the uniform-attention backward (gat_ubwd_* below) that stands in when some score is not
    // finite -- nothing to do otherwise; T (required then) carries the grad_el terms to the pass that sums them
    if (unless_uniform && ones) return;
    for (int vb = (int)blockIdx.x; vb < nblocks; vb += (int)gridDim.x) {     // (one row block each unless the grid is capped: row_prologue)
    const RowInfo ri = row_prologue<6>(row_offsets, node_ids, N, vb);
    float a13[2][4], fu[2][4];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a13[ch][i] = 0.f; fu[ch][i] = 0.f; }
        if (ri.valid) vec_load<4>(fu[ch], feat + (int64_t)ri.r * HD + ch * 256 + lane * 4);
    }
    float gel = 0.f;                                  // lanes 0-7: head = lane
    const int from_head0 = (lane >> 4) * 4, from_head1 = (4 + (lane >> 4)) * 4;   // bpermute byte addresses
    const int from_lead = ((lane & 3) * 16) * 4;     // lanes 0-7: lane 0 of the row that holds head (lane & 3) of a chunk
    const int from_p = ((lane & 7) + 8) * 4;
    for (int base = 0; base < ri.max_deg; base += kWave) {
        const int cnt = ri.deg - base;
        const int cnt_max = min(kWave, ri.max_deg - base);
        int c = 0, ev = 0;
        if (lane < cnt) {
            c = column_indices[ri.beg + base + lane];
            ev = eids[ri.beg + base + lane];
        }
        for (int k = 0; k < cnt_max; k += U) {
            float gv[U][2][4], X[U], Y[U];
            int ek[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k + u;
                const int ck = __builtin_amdgcn_readlane(c, kk & (kWave - 1));
                ek[u] = __builtin_amdgcn_readlane(ev, kk & (kWave - 1));
                if (kk < cnt) {                       // wave-uniform: one row per wave
                    X[u] = pack[(int64_t)ck * 16 + (lane & 15)];
                    Y[u] = ones ? 1.0f : A[(int64_t)ek[u] * H + (lane & 7)];
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) vec_load<4>(gv[u][ch], g + (int64_t)ck * HD + ch * 256 + lane * 4);
                } else {
                    X[u] = 1.f;
                    Y[u] = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                        for (int i = 0; i < 4; ++i) gv[u][ch][i] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + u < cnt) {
                    const float alpha8 = Y[u] / X[u];                       // lanes 0-7: A / S
                    float dot[2];
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const float alpha = __int_as_float(
                            __builtin_amdgcn_ds_bpermute(ch ? from_head1 : from_head0, __float_as_int(alpha8)));
                        float p = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a13[ch][i] = a13[ch][i] + gv[u][ch][i] * alpha;
                            p = __builtin_fmaf(gv[u][ch][i], fu[ch][i], p);     // (a regrouped sum either way)
                        }
                        dot[ch] = row16_sum_lane0(p);
                    }
                    const float d0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from_lead, __float_as_int(dot[0])));
                    const float d1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from_lead, __float_as_int(dot[1])));
                    const float inv = 1.0f / X[u];                          // lanes 0-7
                    const float pv = __int_as_float(__builtin_amdgcn_ds_bpermute(from_p, __float_as_int(X[u])));
                    const float dt = (lane & 4) ? d1 : d0;
                    const float tv = ((dt * inv - pv) * Y[u]) * slope;
                    if (lane < H) {
                        if (T) T[(int64_t)ek[u] * H + lane] = tv;
                        gel = gel + tv;
                    }
                }
            }
        }
    }
    if (ri.valid) {
        if (lane < H && !unless_uniform) grad_el[(int64_t)ri.r * H + lane] = gel;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) vec_store<4>(grad_feat + (int64_t)ri.r * HD + ch * 256 + lane * 4, a13[ch]);
    }
    }
}

// ----------------------------------------- K2 in the uniform-attention form, H = 8 heads of D = 64 over fin = 64 inputs
// With every A = 1.0f (all_ones) and feat = x W^T the backward unit needs no row of width H D per EDGE:
//   * the per-edge dot product of T is  g[v,h,:] . feat[u,h,:] = (W_h^T g[v,h,:]) . x[u,:] = gW[v,h,:] . x[u,:]  -- gW [H][N][fin]
//     is one block-diagonal product per vertex (the caller's batched GEMM), and an edge gathers x[u] (256 bytes) instead of
//     g[v] (2 KB): a pass over the FORWARD CSR (one wave per target v, gW[v] in registers) writes T[eid, 0..7];
//   * grad_feat is only ever used through  grad_feat W = A_hat^T (gs W)  and  grad_feat^T x = g^T xm  (xm = the forward's
//     mean of x over the in-edges, gs = g / S): gsW[v,:] = sum_h gW[v,h,:] / S[v,h] leaves the same pass, and a pass over
//     the BACKWARD CSR sums gsW over a source's out-edges (256 bytes per edge) and T into grad_el (32 bytes per edge).
// 16.4 GB of gathered g rows (2.53 ms) become 2 x 2 GB + 0.5 GB of T.  The regrouped dot product is the same sum in another
// order (the reference adds these terms with atomicAdd in no order at all: tests to 1e-4 as for the factored form).
__device__ __forceinline__ float row16_sum_all(float v)     // every lane of the DPP row gets the row's sum (row_ror 8, 4, 2, 1)
{
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
    return v;
}

// forward CSR: rows = targets v, columns = sources u.  A wave owns a target and takes its in-edges 16 at a time as ONE small
// product on the fp32 matrix instruction: D[edge][head] = X[16 edges][64] . gW[v][64][8 heads (of 16 columns)], sixteen
// v_mfma_f32_16x16x4_f32.  Lane (i = lane & 15, kq = lane >> 4) loads x[u_i][16 j + 4 kq .. + 3] (four 16-byte loads: the 256
// bytes of a row over its four kq lanes) and, as the B operand, gW[h = i][v] at the same columns -- the k index is permuted the
// same way on both sides, which a sum over k does not see.  (As vector FMAs with DPP row sums the same pass took 0.62 ms: 82
// instruction cycles per edge against ~ 37 here.)  If some score is not finite (flag set) the launch only zeroes gsW[v]: the
// general kernel (unless_uniform) writes T, and the row sums below then add nothing to gx.
using f32x4_t = __attribute__((ext_vector_type(4))) float;

template <int U>
__global__ __launch_bounds__(kBlock) void gat_ubwd_t_kernel(
    const float *__restrict__ gW, const float *__restrict__ pack, const float *__restrict__ x, float *__restrict__ T,
    float *__restrict__ gsW, const int *__restrict__ row_offsets, const int *__restrict__ column_indices,
    const int *__restrict__ eids, const int *__restrict__ node_ids, int N, float slope, const int *__restrict__ flag)
{
    constexpr int H = 8, FIN = 64;
    const int lane = threadIdx.x & (kWave - 1), i16 = lane & 15, kq = lane >> 4;
    const RowInfo ri = row_prologue<6>(row_offsets, node_ids, N);
    if (!all_ones(flag)) {
        if (ri.valid && lane < 16) *reinterpret_cast<float4 *>(gsW + (int64_t)ri.r * FIN + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float bw[4][4];                                  // B operand: head i16 (zeros for i16 >= 8), columns 16 j + 4 kq + comp
    float pv = 0.f, inv = 1.f, sv = 1.f;             // head i16 of this target
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[j][q] = 0.f;
    if (ri.valid && i16 < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vec_load<4>(bw[j], gW + ((int64_t)i16 * N + ri.r) * FIN + 16 * j + 4 * kq);
        sv = pack[(int64_t)ri.r * 16 + i16];
        pv = pack[(int64_t)ri.r * 16 + 8 + i16];
        inv = 1.0f / sv;
    }
    if (ri.valid) {
        // gsW[v, k] = sum_h gW[h, v, k] / S[v, h]  (alpha = A / S with A = 1: the unit's division): the heads are the lanes of a DPP row
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = row16_sum_all(bw[j][q] / sv);
            if (i16 == 0) vec_store<4>(gsW + (int64_t)ri.r * FIN + 16 * j + 4 * kq, o);
        }
    }
    for (int base = 0; base < ri.max_deg; base += kWave) {
        const int cnt = ri.deg - base;
        const int cnt_max = min(kWave, ri.max_deg - base);
        int cidx = 0, ev = 0;
        if (lane < cnt) {
            cidx = column_indices[ri.beg + base + lane];
            ev = eids[ri.beg + base + lane];
        }
        for (int k = 0; k < cnt_max; k += 16 * U) {
            float xv[U][4][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k + 16 * u + i16;                              // this lane's edge of the group (source 0 past the last: a valid row)
                const int ck = __builtin_amdgcn_ds_bpermute((kk & (kWave - 1)) * 4, cidx);
#pragma unroll
                for (int j = 0; j < 4; ++j) vec_load<4>(xv[u][j], x + (int64_t)ck * FIN + 16 * j + 4 * kq);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + 16 * u >= cnt_max) break;                             // wave-uniform
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][j][q], bw[j][q], acc, 0, 0, 0);
                // acc[r] = (edge 4 kq + r of the group) . (head i16)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = k + 16 * u + 4 * kq + r;
                    const int ek = __builtin_amdgcn_ds_bpermute((kk & (kWave - 1)) * 4, ev);
                    const float tv = ((acc[r] * inv - pv) * 1.0f) * slope;   // the unit's term with A = 1.0f
                    if (kk < cnt && i16 < H) T[(int64_t)ek * H + i16] = tv;
                }
            }
        }
    }
}

// backward CSR: rows = sources u, columns = targets v.  grad_el[u, h] = sum over u's out-edges of T[eid, h];
// gxa[u, :] = sum over them of gsW[v, :] (= grad_feat[u] W, see above).  K1's shape at width 64: a row per 16 lanes (16 bytes of
// gsW[v] each), four rows per wave, U edges of each in flight (one row per wave with 8 edges in flight: 0.42 ms, 5.4 TB/s;
// K1 itself runs the same 2 GB gather at 0.94 of the roofline).
template <int U>
__global__ __launch_bounds__(kBlock) void gat_ubwd_src_kernel(
    const float *__restrict__ T, const float *__restrict__ gsW, float *__restrict__ grad_el, float *__restrict__ gxa,
    const int *__restrict__ row_offsets, const int *__restrict__ column_indices, const int *__restrict__ eids,
    const int *__restrict__ node_ids, int N)
{
    constexpr int H = 8, FIN = 64, G = 16;
    const int c = threadIdx.x & (G - 1);
    const RowInfo ri = row_prologue<4>(row_offsets, node_ids, N);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float tacc = 0.f;                                // lanes c < 8: head c
    for (int base = 0; base < ri.max_deg; base += G) {
        const int cnt = ri.deg - base;
        const int cnt_max = min(G, ri.max_deg - base);
        int cidx = 0, ev = 0;
        if (c < cnt) {
            cidx = column_indices[ri.beg + base + c];
            ev = eids[ri.beg + base + c];
        }
        for (int k = 0; k < cnt_max; k += U) {
            float gv[U][4], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k + u;
                const int ck = gbcast_i<G>(cidx, kk & (G - 1));              // (edge 0 / target 0 past the row's last: valid addresses)
                const int ek = gbcast_i<G>(ev, kk & (G - 1));
                vec_load<4>(gv[u], gsW + (int64_t)ck * FIN + 4 * c);
                tv[u] = T[(int64_t)ek * H + (c & 7)];
                if (kk >= cnt) {
                    tv[u] = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[u][i] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                tacc = tacc + tv[u];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = acc[i] + gv[u][i];
            }
        }
    }
    if (ri.valid) {
        if (c < H) grad_el[(int64_t)ri.r * H + c] = tacc;
        vec_store<4>(gxa + (int64_t)ri.r * FIN + 4 * c, acc);
    }
}

// the stand-in's input gradient when some score is not finite: gx[u, :] += gf[u, :] W  (W [512][64]; a wave per row; rare and
// mostly NaN by then, so plain FMAs)
__global__ __launch_bounds__(kBlock) void gat_ubwd_gx_fallback_kernel(const float *__restrict__ gf, const float *__restrict__ W,
                                                                      float *__restrict__ gx, int N, const int *__restrict__ flag)
{
    if (all_ones(flag)) return;
    constexpr int HD = 512, FIN = 64;
    const int lane = threadIdx.x & (kWave - 1);
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= N) return;
    float acc = gx[(int64_t)r * FIN + lane];
    for (int j0 = 0; j0 < HD; j0 += kWave) {
        const float gvl = gf[(int64_t)r * HD + j0 + lane];
        for (int j = 0; j < kWave; ++j) {
            const float gj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gvl), j));
            acc = __builtin_fmaf(gj, W[(int64_t)(j0 + j) * FIN + lane], acc);
        }
    }
    gx[(int64_t)r * FIN + lane] = acc;
}

// -------------------------------------------------------------------------- bwd_er
template <int LOG2G>
__global__ __launch_bounds__(kBlock) void gat_bwd_er_kernel(
    const float *__restrict__ T, float *__restrict__ grad_er, const int *__restrict__ row_offsets,
    const int *__restrict__ eids, const int *__restrict__ node_ids, int N, int H, int H_active)
{
    constexpr int G = 1 << LOG2G;
    constexpr int U = G < 4 ? G : 4;
    const int j = threadIdx.x & (G - 1);
    const RowInfo ri = row_prologue<LOG2G>(row_offsets, node_ids, N);
    for (int hbase = 0; hbase < H_active; hbase += G) {
        const int h = hbase + j;
        const bool hok = h < H_active;
        float acc = 0.f;
        for (int base = 0; base < ri.max_deg; base += G) {
            const int cnt = ri.deg - base;
            const int cnt_max = min(G, ri.max_deg - base);
            int ev = 0;
            if (j < cnt) ev = eids[ri.beg + base + j];
            for (int k = 0; k < cnt_max; k += U) {
                float t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = k + u;
                    const int ek = gbcast_i<G>(ev, kk & (G - 1));
                    t[u] = (kk < cnt && hok) ? T[(int64_t)ek * H + h] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (k + u < cnt && hok) acc = acc + t[u];
            }
        }
        if (ri.valid && hok) grad_er[(int64_t)ri.r * H + h] = acc;
    }
}

// ------------------------------------------------------------------------ dispatch
namespace {

inline unsigned grid_for(int N, int log2g)
{
    const int rows_per_block = (kWave >> log2g) * kWavesPerBlock;
    return (unsigned)(((int64_t)N + rows_per_block - 1) / rows_per_block);
}

struct FeatPlan {
    int vec, log2g, chunks;
};

// Lane plan for a [H, D] feature row of which the first `active` floats are computed.
FeatPlan plan_features(int HD, int D, int active, uintptr_t align, bool need_head_lanes)
{
    int vec = 1;
    if (HD % 4 == 0 && active % 4 == 0 && D % 4 == 0 && align % 16 == 0) vec = 4;
    else if (HD % 2 == 0 && active % 2 == 0 && D % 2 == 0 && align % 8 == 0) vec = 2;
    (void)need_head_lanes;
    const int lanes = (active + vec - 1) / vec;
    int log2g = ilog2_ceil(lanes), chunks = 1;
    if (log2g > 6) {
        log2g = 6;
        chunks = ((lanes + kWave - 1) / kWave) >= 3 ? 4 : 2;
    }
    return {vec, log2g, chunks};
}

#define STG_SWITCH_LOG2G(L, ...)                       \
    switch (L) {                                       \
        case 0: { constexpr int LG = 0; __VA_ARGS__; break; } \
        case 1: { constexpr int LG = 1; __VA_ARGS__; break; } \
        case 2: { constexpr int LG = 2; __VA_ARGS__; break; } \
        case 3: { constexpr int LG = 3; __VA_ARGS__; break; } \
        case 4: { constexpr int LG = 4; __VA_ARGS__; break; } \
        case 5: { constexpr int LG = 5; __VA_ARGS__; break; } \
        default: { constexpr int LG = 6; __VA_ARGS__; break; } \
    }

}  // namespace
}  // namespace stg

extern "C" int stg_gat_fwd_k0(const float *el, const float *er, float *A, float *S,
                              const int32_t *row_offsets, const int32_t *column_indices,
                              const int32_t *eids, const int32_t *node_ids, int32_t N, int32_t H,
                              int32_t H_active, float slope, const int32_t *ones_flag, void *stream)
{
    using namespace stg;
    if (N < 0 || H <= 0 || H_active < 0 || H_active > H)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_fwd_k0: bad shape N=%d H=%d H_active=%d", N, H, H_active);
    if (N == 0 || H_active == 0) return 0;
    if (!el || !er || !S || !row_offsets)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_fwd_k0: NULL pointer argument");
    const int log2g = std::min(6, ilog2_ceil(H_active));
    hipStream_t st = static_cast<hipStream_t>(stream);
    STG_SWITCH_LOG2G(log2g, hipLaunchKernelGGL((gat_k0_kernel<LG>), dim3(grid_for(N, LG)), dim3(kBlock), 0, st,
                                               el, er, A, S, row_offsets, column_indices, eids, node_ids,
                                               N, H, H_active, slope, ones_flag));
    return check_launch("stg_gat_fwd_k0");
}

namespace stg {
namespace {
// One K1 launch: out[N, H*D] = sum_e (A / S[row * s_stride + head]) * feat[u]; `when` and `act_out` as in the kernel.
int launch_k1(const char *what, const float *A, const float *S, int s_stride, const float *feat, float *out, float *act_out,
              const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids, const int32_t *node_ids,
              int32_t N, int32_t H, int32_t D, int32_t HD_active, const int32_t *ones_flag, int when, void *stream)
{
    if (N < 0 || H <= 0 || D <= 0 || HD_active < 0 || (int64_t)HD_active > (int64_t)H * D)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad shape N=%d H=%d D=%d HD_active=%d", what, N, H, D, HD_active);
    if (N == 0 || HD_active == 0) return 0;
    if (!S || !feat || !out || !row_offsets || (when != kK1Always && !ones_flag))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", what);
    const uintptr_t align = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                            reinterpret_cast<uintptr_t>(act_out);
    const FeatPlan p = plan_features(H * D, D, HD_active, align, false);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define STG_K1(VEC, CH, UN)                                                                              \
    STG_SWITCH_LOG2G(p.chunks > 1 ? 6 : p.log2g,                                                         \
                     hipLaunchKernelGGL((gat_k1_kernel<VEC, (CH > 1 ? 6 : LG), CH, UN>),                 \
                                        dim3(when == kK1UnlessUniform ? std::min<unsigned>(grid_for(N, (CH > 1 ? 6 : LG)), kFallbackGrid) \
                                                                      : grid_for(N, (CH > 1 ? 6 : LG))), dim3(kBlock), 0, st, A, S, \
                                        feat, out, row_offsets, column_indices, eids, node_ids, N, H, D, \
                                        HD_active, ones_flag, s_stride, when, act_out, (int)grid_for(N, (CH > 1 ? 6 : LG))))
#define STG_K1_VEC(VEC)                          \
    if (p.chunks == 4) { STG_K1(VEC, 4, 2); }    \
    else if (p.chunks == 2) { STG_K1(VEC, 2, 4); } \
    else { STG_K1(VEC, 1, 8); }
    if (p.vec == 4) { STG_K1_VEC(4) } else if (p.vec == 2) { STG_K1_VEC(2) } else { STG_K1_VEC(1) }
#undef STG_K1_VEC
#undef STG_K1
    return check_launch(what);
}
}  // namespace
}  // namespace stg

extern "C" int stg_gat_fwd_k1(const float *A, const float *S, const float *feat, float *out,
                              const int32_t *row_offsets, const int32_t *column_indices,
                              const int32_t *eids, const int32_t *node_ids, int32_t N, int32_t H,
                              int32_t D, int32_t HD_active, const int32_t *ones_flag, void *stream)
{
    return stg::launch_k1("stg_gat_fwd_k1", A, S, H, feat, out, nullptr, row_offsets, column_indices, eids, node_ids, N,
                          H, D, HD_active, ones_flag, stg::kK1Always, stream);
}

extern "C" int stg_gat_fwd_k1_uniform(const float *S, int32_t H, const float *x, float *xm, const int32_t *row_offsets,
                                      const int32_t *column_indices, const int32_t *node_ids, int32_t N, int32_t F,
                                      const int32_t *ones_flag, void *stream)
{
    if (H <= 0) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_fwd_k1_uniform: bad H=%d", H);
    // one "head" of F columns whose S is head 0's of the H real ones (all equal: the in-degree)
    return stg::launch_k1("stg_gat_fwd_k1_uniform", nullptr, S, H, x, xm, nullptr, row_offsets, column_indices, nullptr,
                          node_ids, N, 1, F, F, ones_flag, stg::kK1IfUniform, stream);
}

extern "C" int stg_gat_fwd_k1_scored(const float *A, const float *S, const float *feat, float *out, float *act_out,
                                     const int32_t *row_offsets, const int32_t *column_indices, const int32_t *eids,
                                     const int32_t *node_ids, int32_t N, int32_t H, int32_t D,
                                     const int32_t *ones_flag, void *stream)
{
    if (!A || !eids) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_fwd_k1_scored: NULL pointer argument");
    return stg::launch_k1("stg_gat_fwd_k1_scored", A, S, H, feat, out, act_out, row_offsets, column_indices, eids, node_ids,
                          N, H, D, H * D, ones_flag, stg::kK1UnlessUniform, stream);
}

extern "C" int stg_gat_bwd(const float *A, const float *S, const float *out, const float *g,
                           const float *el, const float *er, const float *feat, float *grad_feat,
                           float *grad_el, float *T, const int32_t *row_offsets,
                           const int32_t *column_indices, const int32_t *eids,
                           const int32_t *node_ids, int32_t N, int32_t H, int32_t D,
                           int32_t HD_active, float slope, const int32_t *ones_flag, void *stream)
{
    using namespace stg;
    if (N < 0 || H <= 0 || D <= 0 || HD_active < 0 || (int64_t)HD_active > (int64_t)H * D)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd: bad shape N=%d H=%d D=%d HD_active=%d", N, H, D, HD_active);
    if (N == 0 || HD_active == 0) return 0;
    if (!S || !out || !g || !el || !er || !feat || !grad_feat || !grad_el || !row_offsets)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd: NULL pointer argument");
    const uintptr_t align = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                            reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(grad_feat);
    const FeatPlan p = plan_features(H * D, D, HD_active, align, true);
    const int LH = D / p.vec;
    // butterfly needs: power-of-two lanes per head that tile the row group exactly
    const bool pow2 = (LH & (LH - 1)) == 0 && LH <= (1 << p.log2g);
    // a head must live inside one row-group chunk, otherwise its per-edge sum would be split
    if (LH > kWave || (!pow2 && (HD_active + p.vec - 1) / p.vec > kWave))
        return fail(STG_ERR_UNSUPPORTED,
                    "stg_gat_bwd: head width D=%d (H=%d) is outside the supported range "
                    "(D/vec <= 64; non power-of-two D/vec needs H*D/vec <= 64)", D, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define STG_K2(VEC, CH, UN, P2)                                                                          \
    STG_SWITCH_LOG2G(p.chunks > 1 ? 6 : p.log2g,                                                         \
                     hipLaunchKernelGGL((gat_bwd_kernel<VEC, (CH > 1 ? 6 : LG), CH, UN, P2>),            \
                                        dim3(grid_for(N, (CH > 1 ? 6 : LG))), dim3(kBlock), 0, st, A, S, \
                                        out, g, el, er, feat, grad_feat, grad_el, T, row_offsets,        \
                                        column_indices, eids, node_ids, N, H, D, HD_active, slope, ones_flag))
#define STG_K2_VEC(VEC, P2)                          \
    if (p.chunks == 4) { STG_K2(VEC, 4, 2, P2); }    \
    else if (p.chunks == 2) { STG_K2(VEC, 2, 2, P2); } \
    else { STG_K2(VEC, 1, 4, P2); }
    if (pow2) {
        if (p.vec == 4) { STG_K2_VEC(4, true) } else if (p.vec == 2) { STG_K2_VEC(2, true) } else { STG_K2_VEC(1, true) }
    } else {
        if (p.vec == 4) { STG_K2_VEC(4, false) } else if (p.vec == 2) { STG_K2_VEC(2, false) } else { STG_K2_VEC(1, false) }
    }
#undef STG_K2_VEC
#undef STG_K2
    return check_launch("stg_gat_bwd");
}

namespace stg {
namespace {
// g_pre == nullptr: g is the gradient of `out`.  Otherwise g is the gradient of elu(out): the per-vertex pass turns it
// into the gradient of `out` on the way (it reads g and out anyway), leaves that in g_pre, and the edge pass gathers g_pre.
int bwd_factored(const char *what, const float *A, const float *S, const float *out, const float *g, float *g_pre,
                 const float *feat, float *grad_feat, float *grad_el, float *T, float *P, const int32_t *row_offsets,
                 const int32_t *column_indices, const int32_t *eids, const int32_t *node_ids, int32_t N, int32_t H,
                 int32_t D, float slope, float *grad_er, const int32_t *ones_flag, void *stream)
{
    if (N < 0 || H <= 0 || D <= 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad shape N=%d H=%d D=%d", what, N, H, D);
    if (N == 0) return 0;
    if (!S || !out || !g || !feat || !grad_feat || !grad_el || !P || !row_offsets || (!T && !grad_er))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument (T or grad_er must be given)", what);
    const uintptr_t align = reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                            reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(grad_feat) |
                            reinterpret_cast<uintptr_t>(g_pre);
    const FeatPlan p = plan_features(H * D, D, H * D, align, true);
    const int LH = D / p.vec;
    const bool pow2 = (LH & (LH - 1)) == 0 && LH <= (1 << p.log2g);
    if (LH > kWave || (!pow2 && (H * D + p.vec - 1) / p.vec > kWave))
        return fail(STG_ERR_UNSUPPORTED,
                    "%s: head width D=%d (H=%d) is outside the supported range "
                    "(D/vec <= 64; non power-of-two D/vec needs H*D/vec <= 64)", what, D, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *gq = g_pre ? g_pre : g;               // what the edge pass gathers
    if (H == 8 && D == 64 && p.vec == 4) {
        hipLaunchKernelGGL(gat_bwd_prepass_h8d64_kernel, dim3((unsigned)((N + kWavesPerBlock - 1) / kWavesPerBlock)),
                           dim3(kBlock), 0, st, S, out, g, P, N, grad_er, slope, g_pre);
        hipLaunchKernelGGL(gat_bwd_fact_h8d64_kernel<2>, dim3(grid_for(N, 6)), dim3(kBlock), 0, st, A, P, gq, feat,
                           grad_feat, grad_el, T, row_offsets, column_indices, eids, node_ids, N, slope, ones_flag, 0, (int)grid_for(N, 6));
        return check_launch(what);
    }
#define STG_K2F(VEC, CH, UN, P2)                                                                           \
    STG_SWITCH_LOG2G(p.chunks > 1 ? 6 : p.log2g, {                                                         \
        constexpr int LGE = (CH > 1 ? 6 : LG);                                                             \
        hipLaunchKernelGGL((gat_bwd_prepass_kernel<VEC, LGE, CH, P2>), dim3(grid_for(N, LGE)), dim3(kBlock), \
                           0, st, S, out, g, P, N, H, D, grad_er, slope, g_pre);                                  \
        if (P2 && LH == 16 && LGE >= 4)                                                                    \
            hipLaunchKernelGGL((gat_bwd_fact_kernel<VEC, LGE, CH, UN, P2, true>), dim3(grid_for(N, LGE)),  \
                               dim3(kBlock), 0, st, A, S, P, gq, feat, grad_feat, grad_el, T, row_offsets,  \
                               column_indices, eids, node_ids, N, H, D, slope, ones_flag);                 \
        else                                                                                               \
            hipLaunchKernelGGL((gat_bwd_fact_kernel<VEC, LGE, CH, UN, P2, false>), dim3(grid_for(N, LGE)), \
                               dim3(kBlock), 0, st, A, S, P, gq, feat, grad_feat, grad_el, T, row_offsets,  \
                               column_indices, eids, node_ids, N, H, D, slope, ones_flag);                 \
    })
    /* two-chunk rows (H*D = 512 at cfg3): unroll 2, not 4 -- 101 -> 80-odd VGPRs buys a fifth and sixth wave per   \
       SIMD, worth more than the deeper gather queue (measured 3.42 -> 2.90 ms; unroll 1: 3.1) */                  \
#define STG_K2F_VEC(VEC, P2)                           \
    if (p.chunks == 4) { STG_K2F(VEC, 4, 2, P2); }     \
    else if (p.chunks == 2) { STG_K2F(VEC, 2, 2, P2); } \
    else { STG_K2F(VEC, 1, 8, P2); }
    if (pow2) {
        if (p.vec == 4) { STG_K2F_VEC(4, true) } else if (p.vec == 2) { STG_K2F_VEC(2, true) } else { STG_K2F_VEC(1, true) }
    } else {
        if (p.vec == 4) { STG_K2F_VEC(4, false) } else if (p.vec == 2) { STG_K2F_VEC(2, false) } else { STG_K2F_VEC(1, false) }
    }
#undef STG_K2F_VEC
#undef STG_K2F
    return check_launch(what);
}

}  // namespace
}  // namespace stg

extern "C" int stg_gat_bwd_factored(const float *A, const float *S, const float *out, const float *g,
                                    const float *feat, float *grad_feat, float *grad_el, float *T,
                                    float *P, const int32_t *row_offsets,
                                    const int32_t *column_indices, const int32_t *eids,
                                    const int32_t *node_ids, int32_t N, int32_t H, int32_t D, float slope,
                                    float *grad_er, const int32_t *ones_flag, void *stream)
{
    return stg::bwd_factored("stg_gat_bwd_factored", A, S, out, g, nullptr, feat, grad_feat, grad_el, T, P, row_offsets,
                             column_indices, eids, node_ids, N, H, D, slope, grad_er, ones_flag, stream);
}

extern "C" int stg_gat_bwd_factored_elu(const float *A, const float *S, const float *out, const float *g_act, float *g_pre,
                                        const float *feat, float *grad_feat, float *grad_el, float *T,
                                        float *P, const int32_t *row_offsets,
                                        const int32_t *column_indices, const int32_t *eids,
                                        const int32_t *node_ids, int32_t N, int32_t H, int32_t D, float slope,
                                        float *grad_er, const int32_t *ones_flag, void *stream)
{
    if (!g_pre) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_factored_elu: NULL g_pre");
    return stg::bwd_factored("stg_gat_bwd_factored_elu", A, S, out, g_act, g_pre, feat, grad_feat, grad_el, T, P,
                             row_offsets, column_indices, eids, node_ids, N, H, D, slope, grad_er, ones_flag, stream);
}

// ---- K2 in the uniform-attention form (gat_ubwd_* kernels above): H = 8, D = 64, fin = 64 ------------------------------------------
extern "C" int stg_gat_bwd_uniform_supported(int32_t H, int32_t D, int32_t fin) { return H == 8 && D == 64 && fin == 64 ? 1 : 0; }

extern "C" int stg_gat_bwd_prepass(const float *S, const float *out, const float *g, float *g_pre, float *pack, int32_t N,
                                   int32_t H, int32_t D, float slope, float *grad_er, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_prepass: negative N");
    if (H != 8 || D != 64) return fail(STG_ERR_UNSUPPORTED, "stg_gat_bwd_prepass: H = 8, D = 64 only (got %d, %d)", H, D);
    if (N == 0) return 0;
    if (!S || !out || !g || !pack) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_prepass: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g_pre)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_prepass: out, g and g_pre must be 16-byte aligned");
    hipLaunchKernelGGL(gat_bwd_prepass_h8d64_kernel, dim3((unsigned)((N + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), S, out, g, pack, N, grad_er, slope, g_pre);
    return check_launch("stg_gat_bwd_prepass");
}

extern "C" int stg_gat_bwd_uniform_edges(const float *A, const float *pack, const float *gq, const float *feat, const float *x,
                                         const float *gW, float *T, float *gsW, float *grad_feat, float *grad_el, float *gxa,
                                         const int32_t *fwd_row_offsets, const int32_t *fwd_column_indices, const int32_t *fwd_eids,
                                         const int32_t *fwd_node_ids, const int32_t *bwd_row_offsets,
                                         const int32_t *bwd_column_indices, const int32_t *bwd_eids, const int32_t *bwd_node_ids,
                                         int32_t N, float slope, const int32_t *ones_flag, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_uniform_edges: negative N");
    if (N == 0) return 0;
    if (!A || !pack || !gq || !feat || !x || !gW || !T || !gsW || !grad_feat || !grad_el || !gxa || !fwd_row_offsets ||
        !fwd_column_indices || !fwd_eids || !bwd_row_offsets || !bwd_column_indices || !bwd_eids || !ones_flag)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_uniform_edges: NULL pointer argument");
    if ((reinterpret_cast<uintptr_t>(gq) | reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(x) |
         reinterpret_cast<uintptr_t>(gW) | reinterpret_cast<uintptr_t>(gsW) | reinterpret_cast<uintptr_t>(grad_feat) |
         reinterpret_cast<uintptr_t>(gxa)) % 16 != 0)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_uniform_edges: matrices must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // every score finite: T and gsW from the pass over the targets; otherwise gsW = 0 and the general unit writes T and grad_feat
    hipLaunchKernelGGL(gat_ubwd_t_kernel<2>, dim3(grid_for(N, 6)), dim3(kBlock), 0, st, gW, pack, x, T, gsW, fwd_row_offsets,
                       fwd_column_indices, fwd_eids, fwd_node_ids, N, slope, ones_flag);
    // (the general unit standing in: a capped grid walking the row blocks -- it almost always returns on the flag)
    hipLaunchKernelGGL(gat_bwd_fact_h8d64_kernel<2>, dim3(std::min<unsigned>(grid_for(N, 6), kFallbackGrid)), dim3(kBlock), 0, st, A, pack, gq,
                       feat, grad_feat, grad_el, T, bwd_row_offsets, bwd_column_indices, bwd_eids, bwd_node_ids, N, slope, ones_flag, 1,
                       (int)grid_for(N, 6));
    hipLaunchKernelGGL(gat_ubwd_src_kernel<8>, dim3(grid_for(N, 4)), dim3(kBlock), 0, st, T, gsW, grad_el, gxa, bwd_row_offsets,
                       bwd_column_indices, bwd_eids, bwd_node_ids, N);
    return check_launch("stg_gat_bwd_uniform_edges");
}

extern "C" int stg_gat_bwd_uniform_gx_fallback(const float *grad_feat, const float *W, float *gx, int32_t N, const int32_t *ones_flag,
                                               void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_uniform_gx_fallback: negative N");
    if (N == 0) return 0;
    if (!grad_feat || !W || !gx || !ones_flag) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_uniform_gx_fallback: NULL pointer argument");
    hipLaunchKernelGGL(gat_ubwd_gx_fallback_kernel, dim3((unsigned)((N + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), grad_feat, W, gx, N, ones_flag);
    return check_launch("stg_gat_bwd_uniform_gx_fallback");
}

extern "C" int stg_gat_bwd_er(const float *T, float *grad_er, const int32_t *row_offsets,
                              const int32_t *eids, const int32_t *node_ids, int32_t N, int32_t H,
                              int32_t H_active, void *stream)
{
    using namespace stg;
    if (N < 0 || H <= 0 || H_active < 0 || H_active > H)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_er: bad shape N=%d H=%d H_active=%d", N, H, H_active);
    if (N == 0 || H_active == 0) return 0;
    if (!grad_er || !row_offsets)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_bwd_er: NULL pointer argument");
    const int log2g = std::min(6, ilog2_ceil(H_active));
    hipStream_t st = static_cast<hipStream_t>(stream);
    STG_SWITCH_LOG2G(log2g, hipLaunchKernelGGL((gat_bwd_er_kernel<LG>), dim3(grid_for(N, LG)), dim3(kBlock), 0, st,
                                               T, grad_er, row_offsets, eids, node_ids, N, H, H_active));
    return check_launch("stg_gat_bwd_er");
}

extern "C" int stg_gat_score_flag(const float *el, const float *er, int64_t n, int32_t *flag, void *stream)
{
    using namespace stg;
    if (n < 0 || !flag) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_score_flag: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (const int rc = zero_async(flag, sizeof(int32_t), st)) return rc;
    if (n == 0) return 0;
    if (!el || !er) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_score_flag: NULL pointer argument");
    const unsigned grid = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL(gat_score_flag_kernel, dim3(grid), dim3(kBlock), 0, st, el, er, n, flag);
    return check_launch("stg_gat_score_flag");
}
