// One TGCN training step of the temporal harnesses as ONE launch forward and ONE launch backward.
//
// Reference: nn/pytorch/temporal/tgcn.py:21-55 (three GCNConv gates sharing graph and input, clamp, the three gate
// Linears, GRU blend) called from benchmarking/static-temporal-tgcn/seastar/model.py:6-18 (relu -> Linear(hidden, 32)
// -> Linear(32, 1)) and its train loop (`cost += mean((y_out - y[t]) ** 2)`); the dynamic-temporal model
// (dynamic-temporal-tgcn/seastar/model.py:5-21) uses the same step with the relu -> Linear head only.
//
// Forward, per 16-row tile (one wave):
//   P   = norm[r] * sum_e (nc[e] * x[col[e], :]) * w[e]                  in CSR order (arithmetic of gcn_agg_kernel)
//   x3  = P Wcat + b3        (= A_hat (x [Wcz|Wcr|Wch]) + b by linearity)     [16, 3C]
//   hg  = clamp(x3[:, g])    Z = sigmoid([hz | H] Wz^T + bz)   R = sigmoid([hr | H] Wr^T + br)
//   Ht  = tanh([hh | H*R] Wh^T + bh)     Hn = Z*H + (1 - Z)*Ht
//   y   = relu(Hn) W1^T + b1;   y_out = y W2^T + b2;   partial[tile] = sum_rows (y_out - target)^2
// Backward, per tile, given dHn (from the next step), the next step's input gradient BEFORE its aggregation
// (zn = da3 Wcat^T of step t+1) and d cost:
//   g_y = A_hat^T zn  (gather over the backward CSR)             dyo = 2 (y_out - t) / N * g_cost
//   dyt = g_y + dyo W2;   dHn += (Hn > 0) (dyt W1)               then the GRU / gate / clamp backward,
//   da3 (masked by lo <= x3 <= hi), dH, dzl / drl / dhl (pre-activation gradients), z = da3 Wcat^T
// The weight gradients are tall-skinny contractions over |V| taken once per window (gemm_tn.hip) from what the two
// kernels leave in HBM: P, x3, H, HR, Hn, y, dzl, drl, dhl, da3, dyt, dyo.
//
// Layout ("row pieces").  lane = (n16 = lane & 15, kq = lane >> 4) owns row n16 of the tile and, of every
// [16, F] matrix, the 16-byte pieces at columns 16 j + 4 kq (j = 0 .. F/16 - 1).  v_mfma_f32_16x16x4_f32 is used
// with the WEIGHT as its A operand and the activations as B:  D[m][n] = sum_k W[16 ct + m][k] X[n][k], so a lane's
// four accumulator values are columns 16 ct + 4 kq .. + 3 of ITS OWN row -- again a row piece.  Every product
// therefore consumes and produces the same layout: no LDS transposes between the chained GEMMs, elementwise
// stages act on matching pieces, all [N, *] tensors are loaded and stored as 16-byte row pieces, and the weight
// operand of four consecutive k steps is ONE ds_read_b128 (W kept in LDS in its torch Linear layout [out][in],
// rows padded by 4 floats).  The k order of a product is (j, i, kq); results agree with rocBLAS / the unfused
// kernels to fp32 rounding (tests: 1e-5 relative).
//
// The gather runs with four ADJACENT lanes per row (quad broadcasts of the edge indices on the DPP network, 32
// contiguous bytes of a neighbour row per lane), exactly gcn_agg_xw_kernel's arithmetic, and hands P to the piece
// layout with 16 ds_bpermute (no LDS memory: the ~140 KB of weights leave none to spare).
// Workgroup = WAVES waves, one per CU (~140 KB of weights in LDS); tiles are dealt wave-major over the grid.
#pragma once
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// v_rcp_f32 (1 ulp) instead of the correctly rounded reciprocal: `__frcp_rn` is a 10-instruction division sequence, 48 of them per
// tile were 40 % of the forward launch's vector instructions, all issued beside the MFMAs (MI355X_MICROARCH.md: every vector
// instruction of every wave shares the SIMD's issue port with them).  The gates move by at most one ulp (6e-8 relative).
__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
// clamp(x, lo, hi) of a finite x with lo <= hi as ONE instruction (v_med3_f32; fminf(fmaxf()) is three: a canonicalising max, max, min)
__device__ __forceinline__ float clamp3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

template <int CTRL>
__device__ __forceinline__ int dpp_quad(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
// value of lane `src` (0..3, a constant after unrolling) of the caller's quad
__device__ __forceinline__ int quad_bcast_i(int v, int src)
{
    switch (src) {
        case 0: return dpp_quad<0x00>(v);
        case 1: return dpp_quad<0x55>(v);
        case 2: return dpp_quad<0xAA>(v);
        default: return dpp_quad<0xFF>(v);
    }
}
__device__ __forceinline__ float quad_bcast_f(float v, int src) { return __int_as_float(quad_bcast_i(__float_as_int(v), src)); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 16 lanes of each DPP row; valid in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v)
{
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}

// copy a row-major [rows][cols] matrix into LDS with row stride ld (cols, ld multiples of 4)
template <int NT>
__device__ __forceinline__ void stage_rows(float *dst, int ld, const float *__restrict__ src, int rows, int cols)
{
    const int c4 = cols >> 2, total = rows * c4;
    for (int base = threadIdx.x; base < total; base += 4 * NT) {
        float4 v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = base + s * NT;
            v[s] = i < total ? *reinterpret_cast<const float4 *>(src + (int64_t)i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = base + s * NT;
            if (i < total) {
                const int r = i / c4, c = i - r * c4;
                *reinterpret_cast<float4 *>(dst + r * ld + 4 * c) = v[s];
            }
        }
    }
}

// The same copy in two halves, so that the global loads are in flight while the wave does something else (its first
// tile's gather): issue() fills PER float4 registers per thread from up to NSEG matrices, commit() writes them to LDS.
struct StageSeg {
    const float *src;
    float *dst;
    int rows, cols, ld;                    // rows = 0: unused
};

template <int NT, int NSEG, int PER>
struct Stager {
    float4 v[PER];

    __device__ __forceinline__ void issue(const StageSeg (&segs)[NSEG])
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * NT;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            int base = 0;
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                const int cnt = segs[s].rows * (segs[s].cols >> 2);
                if (i >= base && i < base + cnt) v[k] = *reinterpret_cast<const float4 *>(segs[s].src + (int64_t)(i - base) * 4);
                base += cnt;
            }
        }
    }

    __device__ __forceinline__ void commit(const StageSeg (&segs)[NSEG]) const
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * NT;
            int base = 0;
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                const int c4 = segs[s].cols >> 2, cnt = segs[s].rows * c4;
                if (i >= base && i < base + cnt) {
                    const int r = (i - base) / c4, c = (i - base) - r * c4;
                    *reinterpret_cast<float4 *>(segs[s].dst + r * segs[s].ld + 4 * c) = v[k];
                }
                base += cnt;
            }
        }
    }
};

// base (uniform: SGPR pair) + 32-bit BYTE offset (one VGPR) + immediate: the saddr form of global_load / global_store.  An
// element index scaled by the compiler (`p + idx`) becomes a 64-bit multiply-add and a VGPR PAIR per array instead.
__device__ __forceinline__ float4 ld_f4(const float *base, unsigned off_bytes, int imm_bytes)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + (size_t)off_bytes + imm_bytes);
}
__device__ __forceinline__ void st_f4(float *base, unsigned off_bytes, int imm_bytes, const float4 &v)
{
#ifdef STG_ABLATE_STORES        // diagnosis builds only (tools/diag/build_step_trace.sh): what the row-piece stores cost
    if (v.x == 123456.75f)
#endif
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + (size_t)off_bytes + imm_bytes) = v;
}
__device__ __forceinline__ float ld_f1(const float *base, unsigned off_bytes)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)off_bytes);
}
__device__ __forceinline__ void st_f1(float *base, unsigned off_bytes, float v)
{
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (size_t)off_bytes) = v;
}

// A per-lane LDS offset the compiler must keep as ONE register: left alone it folds the offset into a separate address
// VGPR per matrix / gate / bias block (a dozen loop invariants), spills them under the gather and reloads each right before
// its ds_read behind an s_waitcnt vmcnt(0).  Uses of `pinned(x) + constant` become ds_read immediates instead.
__device__ __forceinline__ unsigned pinned(unsigned v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Phase timestamps of every wave (tools/diag/step_trace.py builds a second library with -DSTG_STEP_TRACE; the product build has
// none of this): slot [global wave][k] of a host-provided buffer gets the 100 MHz wall clock, slots 14 / 15 the shader clock.
#ifdef STG_STEP_TRACE
#define STG_TRACE_SLOTS 16
__device__ unsigned long long *g_step_trace = nullptr;
#define STG_TRACE_MARK(k)                                                                                                  \
    do {                                                                                                                   \
        if (g_step_trace && (threadIdx.x & 63) == 0)                                                                       \
            g_step_trace[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * STG_TRACE_SLOTS + (k)] =           \
                (k) >= 14 ? (unsigned long long)clock64() : (unsigned long long)wall_clock64();                            \
    } while (0)
#else
#define STG_TRACE_MARK(k) ((void)0)
#endif

// "This value exists NOW": an IR-level sink otherwise moves its computation to the block that uses it, products later, and
// keeps the (larger) operands alive in its place.
__device__ __forceinline__ void materialize(float4 &v)
{
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

__device__ __forceinline__ f32x4 to_x4(const float4 &v) { return f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 to_f4(const f32x4 &v) { return make_float4(v[0], v[1], v[2], v[3]); }

// w[ct] = W[16 ct + n16][16 j + 4 kq .. + 3] (LDS, row stride ld; wrow = W + n16 ld + 4 kq): the weight operand of four k steps.
// Row stride: ds_read_b128 is served in four 16-lane groups ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS) over
// 64 banks; lane (n16, kq) starts at bank (ld n16 + 4 kq) mod 64, so ld = 4 mod 64 (the "+ 4 floats" pad of a 64- or
// 128-float row) puts lanes (11, 1) and (12, 0) of every group on one bank: one extra LDS cycle per group, 8 instead of
// 4 per instruction (measured: SQ_LDS_BANK_CONFLICT = 46 % of SQ_LDS_IDX_ACTIVE in both step kernels).  ld = 8 mod 16
// dwords (rows padded by 8 floats) is conflict-free for all four groups.
template <int CT>
__device__ __forceinline__ void load_w(float4 (&w)[CT], const float *__restrict__ wrow, int ld, int j)
{
#ifdef STG_ABLATE_LDS           // diagnosis builds only: what the weight reads from LDS cost (one fragment reused for every step)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) w[ct] = *reinterpret_cast<const float4 *>(wrow);
    return;
#endif
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) w[ct] = *reinterpret_cast<const float4 *>(wrow + ct * 16 * ld + 16 * j);
}

// acc[ct] += w[ct] (four k steps) x in (this lane's row piece):  4 MFMAs per ct, consecutive MFMAs on different accumulators
template <int CT>
__device__ __forceinline__ void mfma_w(f32x4 (&acc)[CT], const float4 (&w)[CT], const float4 &in)
{
    const float b[4] = {in.x, in.y, in.z, in.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float a = i == 0 ? w[ct].x : i == 1 ? w[ct].y : i == 2 ? w[ct].z : w[ct].w;
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], acc[ct], 0, 0, 0);
        }
    }
}

template <int CT>
__device__ __forceinline__ void mfma_piece(f32x4 (&acc)[CT], const float *__restrict__ wrow, int ld, int j, const float4 &in)
{
    float4 w[CT];
    load_w<CT>(w, wrow, ld, j);
    mfma_w<CT>(acc, w, in);
}

// acc[ct] += sum_{j < J} W[16 ct + n16][16 j + 4 kq ..] x in(j): the weights of step j + 1 are read from LDS while the
// MFMAs of step j issue, and no further ahead (left alone the scheduler hoists every LDS read of the unrolled chain
// to the top and spills: 200+ registers).
template <int CT, int J, bool PREFETCH = true, typename InFn>
__device__ __forceinline__ void gemm_pieces(f32x4 (&acc)[CT], const float *__restrict__ wrow, int ld, InFn in)
{
    if constexpr (!PREFETCH) {               // one weight buffer (16 registers less): the other waves of the SIMD cover the LDS latency
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float4 w[CT];
            load_w<CT>(w, wrow, ld, j);
            const float4 x = in(j);
            __builtin_amdgcn_sched_barrier(0);
            mfma_w<CT>(acc, w, x);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    float4 wn[CT];
    load_w<CT>(wn, wrow, ld, 0);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float4 w[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = wn[ct];
        if (j + 1 < J) load_w<CT>(wn, wrow, ld, j + 1);
        const float4 x = in(j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_w<CT>(acc, w, x);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// gemm_pieces with the step-0 weights handed in (`w0`: the caller read them from LDS a phase ahead -- load_w<CT>(w0, wrow, ld, 0) --
// so the product does not start by waiting for LDS) and `tail()` run where the last step has nothing of its own to prefetch:
// the caller reads the NEXT product's step-0 weights there.  With twelve waves on the CU's one LDS a read that an MFMA waits on
// costs several hundred cycles; the backward launch had 19 of them per tile (diagnosis build with the reads removed: -10 us).
template <int CT, int J, typename InFn, typename TailFn>
__device__ __forceinline__ void gemm_chain(f32x4 (&acc)[CT], const float *__restrict__ wrow, int ld, InFn in, const float4 (&w0)[CT],
                                           TailFn tail)
{
    float4 wn[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wn[ct] = w0[ct];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float4 w[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = wn[ct];
        if (j + 1 < J) load_w<CT>(wn, wrow, ld, j + 1);
        else tail();
        const float4 x = in(j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_w<CT>(acc, w, x);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- gather: out8 = norm[r] * sum_e (nc[e] * x[col[e], 8 q .. 8 q + 7]) * w[e], rows of FIN = 32 floats ------------
// lane = (grow = lane >> 2, q = lane & 3); arithmetic and order of gcn_agg_xw_kernel (bit-identical P).
// In three calls, one per dependent round of loads: begin() = the row's extent and norm, indices() = the column indices / per-edge
// scalars of its first kR edges, run() = the neighbour rows (and further index rounds for longer rows).
template <bool HAS_EW>
struct RowGather32 {
    static constexpr int FIN = 32, kR = 16, kI = kR / 4, kU = 8;
    int beg, deg;
    float nr;
    int c[kI];
    float nc[kI], w[kI];

    __device__ __forceinline__ void begin(const int *__restrict__ row_offsets, const float *__restrict__ norm, int r)
    {
        beg = row_offsets[r];
        deg = row_offsets[r + 1] - beg;
        nr = norm[r];
    }

    __device__ __forceinline__ void indices(const int *__restrict__ column_indices, const float *__restrict__ nc_edge,
                                            const float *__restrict__ ew_edge, int base, int q)
    {
        const int cnt = deg - base;
#pragma unroll
        for (int i = 0; i < kI; ++i) {
            c[i] = 0;
            nc[i] = 0.f;
            w[i] = 1.f;
            if (i * 4 + q < cnt) {
                const int e = beg + base + i * 4 + q;
                c[i] = column_indices[e];
                nc[i] = nc_edge[e];
                if constexpr (HAS_EW) w[i] = ew_edge[e];
            }
        }
    }

    // indices(.., 0, q) must have been called
    __device__ __forceinline__ void run(float (&out)[8], const float *__restrict__ x, const int *__restrict__ column_indices,
                                        const float *__restrict__ nc_edge, const float *__restrict__ ew_edge, int q)
    {
        const int max_deg = wave_max_nonneg(deg);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int base = 0; base < max_deg; base += kR) {
            if (base > 0) indices(column_indices, nc_edge, ew_edge, base, q);
            const int cnt = deg - base;
            const int cnt_max = min(kR, max_deg - base);
            // (round 5, measured and dropped: every load unconditional -- slots past the end of a row reading row 0 -- with the values
            //  selected to +0 and no guard around the adds: 37.2 / 40.8 us against 35.6 / 39.3 forward / backward.  With ~10 of 16 slots
            //  in use the exec-mask branches SKIP a third of the arithmetic; 8 selects per slot cost more than they save)
#pragma unroll
            for (int k = 0; k < kR; k += kU) {
                if (k < cnt_max) {
                    float4 v[kU][2];
                    float ncs[kU], ws[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int kk = k + u, el = kk >> 2, src = kk & 3;
                        const int ck = quad_bcast_i(c[el], src);
                        ncs[u] = quad_bcast_f(nc[el], src);
                        ws[u] = 1.f;
                        if constexpr (HAS_EW) ws[u] = quad_bcast_f(w[el], src);
                        if (kk < cnt) {
                            const unsigned off = (unsigned)ck * (FIN * 4u) + 32u * q;
                            v[u][0] = ld_f4(x, off, 0);
                            v[u][1] = ld_f4(x, off, 16);
                        } else {
                            v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        if (k + u < cnt) {
                            const float vv[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w,
                                                 v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                float t = ncs[u] * vv[i];
                                if constexpr (HAS_EW) t = t * ws[u];
                                acc[i] = acc[i] + t;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = acc[i] * nr;
    }
};

// From the gather layout (lane 4 g + q holds columns 8 q .. 8 q + 7 of row g) to row pieces (lane (n16, kq) holds
// columns 16 j + 4 kq .. + 3 of row n16, j = 0, 1) on the LDS crossbar (ds_bpermute: no LDS memory, no fence):
// piece j of lane (n16, kq) is half kq & 1 of what lane 4 n16 + 2 j + (kq >> 1) holds.
__device__ __forceinline__ void gather_to_pieces(const float (&p8)[8], float4 (&pc)[2], int n16, int kq)
{
    const bool hi = (kq & 1) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int addr = (4 * n16 + 2 * j + (kq >> 1)) * 4;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lo_ = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(p8[i]));
            const int hi_ = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(p8[4 + i]));
            v[i] = __int_as_float(hi ? hi_ : lo_);
        }
        pc[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
}


}  // namespace
}  // namespace stg
