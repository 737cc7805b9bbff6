// Shared helpers for libstgraph_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/stgraph_hip.h"

namespace stg {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;

// thread-local message returned by stg_last_error_string()
char *last_error_buffer();
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);

struct Tuning {
    int gcn_lanes_per_row = 0;     // 0 = auto
    int gcn_unroll = 0;            // 0 = auto
    int gcn_long_threshold = 0;    // 0 = default (16 edges); rows above it take the wave-per-row path
};
Tuning &tuning();

template <int VEC>
__device__ __forceinline__ void vec_load(float (&dst)[VEC], const float *p)
{
    if constexpr (VEC == 1) {
        dst[0] = *p;
    } else if constexpr (VEC == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(p);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

template <int VEC>
__device__ __forceinline__ void vec_store(float *p, const float (&src)[VEC])
{
    if constexpr (VEC == 1) {
        *p = src[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(src[0], src[1]);
    } else {
        *reinterpret_cast<float4 *>(p) = make_float4(src[0], src[1], src[2], src[3]);
    }
}

// max over the 64 lanes of a wave (all lanes must be active)
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Order LDS traffic inside ONE wave (lanes exchange data through an LDS tile that no other wave touches): LDS
// instructions of a wave execute in issue order, so all that is needed is (a) the compiler not moving memory
// operations across this point and (b) outstanding LDS results having landed.  A wavefront-scope fence would do
// (a) and (b) too but also drains vmcnt, i.e. stalls on every global load / store in flight -- measured: the fused
// TGCN cell kernels call this 12-24 times per tile.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

inline int ilog2_ceil(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace stg
