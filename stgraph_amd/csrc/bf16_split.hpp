// The 3-term bf16 split of fp32 operands for v_mfma_f32_16x16x32_bf16 (used by rowgemm_x3.hip): x = h + m + l exactly to
// 2^-25 |x|, a product x w taken as the six terms hh + hm + mh + hl + lh + mm in fp32 (what is dropped is below 2^-23 |x w|).
// (Round 4 also built TGCN step kernels on it -- csrc/tgcn_stepx_*.hip, tgcn_stepf_fwd.hip: measured slower than the fp32
// forms at every shape, profiles/r04_stepx_*, r04_stepf_*; retired in round 5, sources in git history.)
#pragma once
#include "tgcn_step.hpp"

namespace stg {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kXTerms = 3;
constexpr int kFragBytes = 64 * 16;                     // one fragment of one term: 16 bytes per lane

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

// A workgroup barrier that orders LDS traffic only (__syncthreads() also drains vmcnt: every global store of the wave).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace
}  // namespace stg
