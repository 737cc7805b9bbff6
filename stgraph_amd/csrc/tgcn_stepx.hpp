// The one-launch TGCN step on the MATRIX cores of gfx950: every product as a 3-term bf16 split with fp32 accumulation.
//
// Why (profiles/r04_coexec_f32mfma.jsonl, tools/diag/coexec.hip): v_mfma_f32_16x16x4_f32 runs ON the vector ALU's fp32 lanes --
// 32 cycles per instruction, every vector instruction of the wave adds its full 4 cycles on top (57.8 cycles for one MFMA + 4
// v_fma), for one, two or three waves per SIMD alike -- so the fp32 form of the step (tgcn_step_fwd.hip) is priced at
// 512 x 32 + ~740 x 4 cycles per 16-row tile whatever the schedule.  The bf16 matrix instruction v_mfma_f32_16x16x32_bf16 covers
// eight times the K in half the cycles on a pipe of its own (vector instructions issue in its shadow).  An fp32 value is the
// exact sum of three bf16 values (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8 significant bits), and a product
// x w is taken as the six terms h h + h m + m h + h l + l h + m m accumulated in fp32 -- what is dropped (m l, l m, l l) is below
// 2^-23 |x w|, the size of ONE fp32 rounding; each bf16 x bf16 product is exact in fp32.  Six 16-cycle instructions replace
// eight 32-cycle ones: 2.7 x less matrix time, off the vector lanes.
//
// Layout.  Three bf16 terms of the weights are 6 bytes per weight: 196 KB, more than a CU's LDS -- but the four SIMDs of a CU
// hold 512 KB of registers.  So the OUTPUT COLUMNS of every product are cut over the waves: wave (team, ct) owns columns
// 16 ct .. 16 ct + 15 of each gate (ct = 0 .. 3), keeps ITS rows of the three gate Linears in registers for the whole launch as
// MFMA A operands (36 fragments = 144 registers, split once per window by stg_tgcn_pack_weights_x3), and the activations -- the
// B operands, needed by all four waves -- travel through LDS as ready-made bf16 fragments: each wave splits the four values per
// lane it produced and drops them into the fragment image.  The small weights (Wcat, the head's W1: 48 KB as fragments) sit in LDS.
// A workgroup is two TEAMS of four waves (one wave of each team per SIMD) working on different tiles two phases apart, so that one
// team's matrix phase runs beside the other's gather / elementwise / store phase; a phase boundary is one workgroup barrier.
//
// Fragment conventions (v_mfma_f32_16x16x32_bf16: lane l holds A[row l & 15][k = 8 (l >> 4) + i], B[k][col l & 15], i = 0 .. 7;
// D[row 4 (l >> 4) + r][col l & 15]).  Weights are A (row = output column inside the wave's 16), activations are B (col = row
// n16 of the tile), so lane (n16 = l & 15, kq = l >> 4) receives output columns 16 ct + 4 kq .. + 3 of ITS row: a 16-byte row
// piece, as in tgcn_step.hpp.  The k index of K-block b is mapped to input column  c(b, kq, i) = 32 b + 16 (i >> 2) + 4 kq + (i & 3):
// a lane's eight k values are the row pieces j = 2 b and j = 2 b + 1 of the input, so the piece a wave PRODUCES (j = its ct) is
// half (ct & 1) of the fragment of K-block ct >> 1: one 8-byte LDS store per term.
#pragma once
#include "tgcn_step.hpp"

namespace stg {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kXC = 64, kXFin = 32, kXFh = 32;          // the shapes the split form is built for (stg_tgcn_step_supported)
constexpr int kXTerms = 3;
constexpr int kFragBytes = 64 * 16;                     // one fragment of one term: 16 bytes per lane

// ---- weight image (stg_tgcn_pack_weights_x3) -----------------------------------------------------------------------------
// forward:  [gate section: ct 0..3][g 0..2][b 0..3][t 0..2] | [Wcat section: ct][g][t] | [W1 section: ct' 0..1][b 0..1][t]
//           then fp32: b3 [3C] | bz br bh [3C] | b1 [Fh] | W2 [Fh] | b2 [1] (+ 3 pad)
// backward: [gate section: ct][g][half 0..1][b 0..1][t] | [W1T section: ct][t] | [Wcat section: ct' 0..1][b 0..5][t]
//           then fp32: W2 [Fh]
constexpr int kFwdGateFrags = 3 * 4 * kXTerms;          // per ct: 36
constexpr int kFwdCatFrags = 3 * kXTerms;               // per ct: 9
constexpr int kFwdHeadFrags = 2 * kXTerms;              // per ct': 6
constexpr int kFwdImgGate = 0;
constexpr int kFwdImgCat = kFwdImgGate + 4 * kFwdGateFrags * kFragBytes;
constexpr int kFwdImgHead = kFwdImgCat + 4 * kFwdCatFrags * kFragBytes;
constexpr int kFwdImgBias = kFwdImgHead + 2 * kFwdHeadFrags * kFragBytes;
constexpr int kFwdBiasFloats = 6 * kXC + 2 * kXFh + 4;
constexpr int kFwdImgBytes = kFwdImgBias + 4 * kFwdBiasFloats;

constexpr int kBwdGateFrags = 3 * 2 * 2 * kXTerms;      // per ct: 36
constexpr int kBwdHeadFrags = kXTerms;                  // per ct: 3  (W1T: K = Fh = 32 is one K-block)
constexpr int kBwdCatFrags = 6 * kXTerms;               // per ct': 18 (K = 3C = 192 is six K-blocks)
constexpr int kBwdImgGate = 0;
constexpr int kBwdImgHead = kBwdImgGate + 4 * kBwdGateFrags * kFragBytes;
constexpr int kBwdImgCat = kBwdImgHead + 4 * kBwdHeadFrags * kFragBytes;
constexpr int kBwdImgBias = kBwdImgCat + 2 * kBwdCatFrags * kFragBytes;
constexpr int kBwdBiasFloats = kXFh;
constexpr int kBwdImgBytes = kBwdImgBias + 4 * kBwdBiasFloats;

// input column of element i of lane group kq in K-block b
__host__ __device__ constexpr int xcol(int b, int kq, int i) { return 32 * b + 16 * (i >> 2) + 4 * kq + (i & 3); }

// ---- the split ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// four fp32 values -> three terms of four bf16 each (8 bytes per term): v = h + m + l to within 2^-25 |v|
struct Split4 {
    uint2 t[kXTerms];
};
__device__ __forceinline__ Split4 split4(const float4 &v)
{
    Split4 s;
    float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) {
        const unsigned p0 = pk_bf16(a, b), p1 = pk_bf16(c, d);
        s.t[k] = make_uint2(p0, p1);
        if (k + 1 < kXTerms) {
            a = a - bf16_lo(p0), b = b - bf16_hi(p0), c = c - bf16_lo(p1), d = d - bf16_hi(p1);      // exact in fp32
        }
    }
    return s;
}

// the fragment (eight k values of this lane) of each term from two split row pieces: elements 0..3 = piece 2 b, 4..7 = piece 2 b + 1
struct Frag3 {
    bf16x8 t[kXTerms];
};
__device__ __forceinline__ Frag3 frag_of(const float4 &lo, const float4 &hi)
{
    Frag3 f;
    float a = lo.x, b = lo.y, c = lo.z, d = lo.w, e = hi.x, g = hi.y, h = hi.z, i = hi.w;
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) {
        const unsigned p0 = pk_bf16(a, b), p1 = pk_bf16(c, d), p2 = pk_bf16(e, g), p3 = pk_bf16(h, i);
        f.t[k] = __builtin_bit_cast(bf16x8, make_uint4(p0, p1, p2, p3));
        if (k + 1 < kXTerms) {
            a = a - bf16_lo(p0), b = b - bf16_hi(p0), c = c - bf16_lo(p1), d = d - bf16_hi(p1);
            e = e - bf16_lo(p2), g = g - bf16_hi(p2), h = h - bf16_lo(p3), i = i - bf16_hi(p3);
        }
    }
    return f;
}

// acc += W x X over one K-block: the six kept terms, small ones first
__device__ __forceinline__ void mfma6(f32x4 &acc, const Frag3 &w, const Frag3 &x)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[0], x.t[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[2], x.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[1], x.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[0], x.t[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[1], x.t[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.t[0], x.t[0], acc, 0, 0, 0);
}

// ---- fragment images in LDS ----------------------------------------------------------------------------------------------------
// An activation image of KB K-blocks: [b][t][lane] 16 bytes.  The producer of row piece j writes half (j & 1) of K-block j >> 1.
__device__ __forceinline__ void frag_store_piece(char *img, int j, int lane, const Split4 &s)
{
    char *p = img + ((j >> 1) * kXTerms * kFragBytes) + lane * 16 + (j & 1) * 8;
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) *reinterpret_cast<uint2 *>(p + k * kFragBytes) = s.t[k];
}
__device__ __forceinline__ Frag3 frag_load(const char *img, int b, int lane)
{
    Frag3 f;
    const char *p = img + (b * kXTerms * kFragBytes) + lane * 16;
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) f.t[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p + k * kFragBytes));
    return f;
}
// a weight fragment triple out of an image section laid out [.. frag index ..][lane]
__device__ __forceinline__ Frag3 wfrag_load(const char *sec, int first_frag, int lane)
{
    Frag3 f;
    const char *p = sec + (size_t)first_frag * kFragBytes + lane * 16;
#pragma unroll
    for (int k = 0; k < kXTerms; ++k) f.t[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p + k * kFragBytes));
    return f;
}

// ---- gather: four rows per wave, sixteen lanes x two floats per row -----------------------------------------------------------
// P[r, :] = norm[r] * sum_e (nc[e] * x[col[e], :]) * w[e] in CSR order: the arithmetic (and order) of gcn_agg_kernel, bit-identical P.
// lane = (rl = lane >> 4: row 4 ct + rl of the tile, c2 = lane & 15: columns 2 c2, 2 c2 + 1).
//
// A gather is three dependent memory round trips (row extent -> edge records -> neighbour rows).  The step kernels spread the
// NEXT tile's gather over the phases of the current one, each round trip issued a phase ahead of its use:
//   extent()  : the row's extent and norm                                      -> 3 registers
//   indices() : (col, nc, w) of the row's first 32 edges, two edges per lane   -> 6 registers
//   stash()   : those records into the wave's own rows of an LDS table [16 rows][32 edges] x 16 bytes
//   run()     : the neighbour rows; edge records come from the table (all 16 lanes of a row read one address: a broadcast),
//               beyond 32 edges from global memory.
// No load sits behind a per-lane guard (a guarded load is a branch and a wait of its own -- the first version of this gather had
// 24 serialised round trips per batch): lanes past the end of their row re-read its last edge and drop the term (|E| >= 1).
#ifndef STGX_GATHER_U
#define STGX_GATHER_U 4                                  // neighbour rows in flight per lane: 4 keeps the step kernels free of spills
#endif
constexpr int kGatherTableEdges = 32;
constexpr int kGatherTableBytes = 16 * kGatherTableEdges * 16;         // per team

template <bool HAS_EW>
struct RowGatherX {
    int beg, end;
    float nr;
    int c0, c1;
    float n0, n1, w0, w1;

    __device__ __forceinline__ void extent(const int *__restrict__ row_offsets, const float *__restrict__ norm, int row)
    {
        beg = row_offsets[row];
        end = row_offsets[row + 1];
        nr = norm[row];
    }
    __device__ __forceinline__ void indices(const int *__restrict__ column_indices, const float *__restrict__ nc_edge,
                                            const float *__restrict__ ew_edge, int c2)
    {
        const int last = max(end - 1, 0);
        const int e0 = min(beg + c2, last), e1 = min(beg + 16 + c2, last);
        c0 = column_indices[e0], c1 = column_indices[e1];
        n0 = nc_edge[e0], n1 = nc_edge[e1];
        w0 = w1 = 1.f;
        if constexpr (HAS_EW) w0 = ew_edge[e0], w1 = ew_edge[e1];
    }
    // `table_row`: the 32 records of THIS lane's row (the 16 lanes of a row share it)
    __device__ __forceinline__ void stash(uint4 *table_row, int c2) const
    {
        table_row[c2] = make_uint4((unsigned)c0, __float_as_uint(n0), __float_as_uint(w0), 0u);
        table_row[16 + c2] = make_uint4((unsigned)c1, __float_as_uint(n1), __float_as_uint(w1), 0u);
    }
    __device__ __forceinline__ float2 run(const uint4 *table_row, const float *__restrict__ x, const int *__restrict__ column_indices,
                                          const float *__restrict__ nc_edge, const float *__restrict__ ew_edge, int c2) const
    {
        constexpr int U = STGX_GATHER_U;
        const int deg = end - beg, last = max(end - 1, 0);
        const int max_deg = wave_max_nonneg(deg);
        float a0 = 0.f, a1 = 0.f;
        for (int base = 0; base < max_deg; base += U) {
            int c[U];
            float nc[U], w[U];
            if (base < kGatherTableEdges) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint4 t = table_row[base + u];
                    c[u] = (int)t.x, nc[u] = __uint_as_float(t.y), w[u] = __uint_as_float(t.z);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = min(beg + base + u, last);
                    c[u] = column_indices[e];
                    nc[u] = nc_edge[e];
                    w[u] = 1.f;
                    if constexpr (HAS_EW) w[u] = ew_edge[e];
                }
            }
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned off = (unsigned)c[u] * (kXFin * 4u) + 8u * c2;
                v[u] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(x) + (size_t)off);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = base + u < deg;
                float t0 = nc[u] * v[u].x, t1 = nc[u] * v[u].y;
                if constexpr (HAS_EW) t0 = t0 * w[u], t1 = t1 * w[u];
                a0 = ok ? a0 + t0 : a0;
                a1 = ok ? a1 + t1 : a1;
            }
        }
        return make_float2(a0 * nr, a1 * nr);
    }
};

// Interval timestamps of every wave (tools/diag/stepx_trace.py builds a second library with -DSTG_STEPX_TRACE; the product build
// has none of this): slot [global wave][2 it] = the 100 MHz wall clock when the wave enters interval `it`, [2 it + 1] when it
// reaches the interval's barrier.
#ifdef STG_STEPX_TRACE
#define STGX_TRACE_SLOTS 128
__device__ unsigned long long *g_stepx_trace = nullptr;
#define STGX_MARK(k)                                                                                                        \
    do {                                                                                                                   \
        if (g_stepx_trace && (threadIdx.x & 63) == 0 && (k) < STGX_TRACE_SLOTS)                                             \
            g_stepx_trace[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * STGX_TRACE_SLOTS + (k)] =          \
                (unsigned long long)wall_clock64();                                                                        \
    } while (0)
#else
#define STGX_MARK(k) ((void)0)
#endif

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store of the wave (its fence
// drains vmcnt): the step kernels store ~15 row pieces per tile that nothing in the launch reads back, and eight waves waiting
// for those acknowledgements at every phase boundary is most of a phase.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace
}  // namespace stg
