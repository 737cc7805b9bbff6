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
// Zero-fill / device-to-device copy AS KERNELS.  hipMemsetAsync / hipMemcpyAsync become memset / copy NODES when the stream is being
// captured, and a replayed HIP graph did not always order such a node before the kernel node that follows it (round 5: the two
// floats stg_tgcn_fold_weights zeroes before its atomicMax kept the pool's old bits -- NaN -- in about one replayed window of 150;
// the step launch then reported a clamp that never happened).  A kernel node is ordered like every other launch of the library.
// bytes % 4 == 0, 4-byte aligned pointers.
int zero_async(void *dst, size_t bytes, hipStream_t stream);
int copy_async(void *dst, const void *src, size_t bytes, hipStream_t stream);

struct Tuning {
    int gcn_lanes_per_row = 0;     // 0 = auto
    int gcn_unroll = 0;            // 0 = auto
    int gcn_long_threshold = 0;    // 0 = default (16 edges); rows above it take the wave-per-row path
    int xw_rows = 0;               // gcn_agg_xw: 0 = auto, 32 / 64 rows per workgroup
    int xw_waves = 0;              // gcn_agg_xw: 0 = auto, 4 / 8 waves per workgroup
    int cell_rows = 0;             // fused TGCN forward cell: 0 = auto (32-row tiles), 16 = 16-row tiles, 32
    int gcn_tile = 0;              // edge-dealt narrow-row kernel: 0 = auto (large grids), 1 = never, 2 = whenever legal
    int gcn_tile_pipe = 0;         // its persistent software-pipelined form: 0 / 1 = off (measured: no gain), 2 = whenever the tile kernel runs
    int gcn_tile_rows = 0;         // its rows per workgroup: 0 = as many as lane groups, else 8 .. lane groups
    int gcn_block = 0;             // 0 = auto; 64 / 128 / 256 = threads per workgroup of the plain gcn_agg launch
    int gcn_addr32 = 0;            // 0 = auto (32-bit gather offsets when the matrix allows); 1 = always 64-bit
    int gcn_xcd_tile = 0;          // 0 = auto; 1 = workgroups round-robin over XCDs; T = runs of T workgroups per XCD
    int step_waves = 0;            // one-launch TGCN step: 0 = auto, 12 / 16 waves per workgroup (168 / 128 registers)
    int gcn_wide_long = 0;         // hubs of rows >= 64 lanes wide: 0 = feature-sliced workgroups (F % 4 == 0, F <= 256), 1 = never
    int step_coop = 0;             // one-launch TGCN step, tiles of the last partial round: 0 = shared by four waves (one per SIMD) where that helps, 1 = one wave each
    int step_spread = 0;           // one-launch TGCN step with fewer tiles than wave slots: 0 = one workgroup per CU, 1 = packed grid
    int store_rows = 0;            // stg_edgeset_step_device with row-offset hints: 0 = new row offsets derived from them inside the merge launch, 1 = searched in their own launch
    int build_lds_count = 0;       // per-snapshot CSR build: 0 = auto (histograms in LDS when |V| fits and the graph is dense enough), 1 = always when |V| fits, 2 = never
    int rowgemm16 = 0;             // stg_rowgemm_f32 at K, M in {64, 128}: 0 = the 16-row row-piece kernel, 1 = never
    int gemm_wide = 0;             // tall-skinny weight gradient: 0 = auto (16-byte-per-lane form where the widths allow), 1 = never
    int gemm_x3 = 0;               // tall-skinny contractions at M in {32, 64, 128}, N in {64, 96, 128}: 0 = 3-term bf16 split from 64 K rows, 1 = never, 2 = always
    int gemm_cyclic = 0;           // its K distribution inside a block: 0 = a contiguous quarter per wave, 1 = 4-row groups in turn
    int gemm_xcd_pair = 0;         // its workgroup order with several M / N groups: 0 = the groups of a K slice on one XCD (shared operand from L2), 1 = dealt in turn
    int rowgemm_x3 = 0;            // row products at K, M in {64, 128}: 0 = auto (3-term bf16 split on the matrix cores from 64 K rows), 1 = never, 2 = whenever legal
};
// rowgemm_x3.hip: the split form of the row product (X, W, Y 16-byte aligned, ldy % 4 == 0)
int rowgemm_x3_launch(int K, int M, const float *X, const float *W, const float *bias, float *Y, int64_t N, bool trans_w, bool relu,
                      void *stream, int ldy);
// the same with the ReLU sign pattern as bits (layout: rowgemm_x3.hip): bits_out with the ReLU forward (W [K][M]), bits_in with the
// input gradient of the layer above (W [M][K]); Y contiguous [N, M]
int rowgemm_x3_heads_launch(int K, int M, const float *X, const float *W, float *Y, int64_t N, int heads, void *stream);
// gat_heads_x3.hip: a GAT layer's per-head products (H % 4 == 0 heads of D = 64 over fin = 64) in the split form, four heads per launch
bool gat_heads_x3_wanted(int64_t N, int H, int D, int fin);
int gat_heads_fc_x3_launch(const char *what, const float *x, const float *W, const float *attn_l, const float *attn_r, float *out, float *act,
                           float *el, float *er, int64_t N, int H, void *stream);
int rowgemm_x3_bits_launch(int K, int M, const float *X, const float *W, const float *bias, float *Y, int64_t N, const uint32_t *bits_in,
                           uint32_t *bits_out, void *stream);
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

// Global-memory variants that only assume the 4-byte alignment of a float: gfx950 takes dwordx2 / dwordx4 global
// accesses at any dword address, which is what lets a row whose width is not a multiple of VEC be covered by
// OVERLAPPING windows (the last lane's window starts at F - VEC) instead of falling back to one float per lane.
typedef float stg_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float stg_f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int VEC>
__device__ __forceinline__ void vec_load_g(float (&dst)[VEC], const float *p)
{
    if constexpr (VEC == 1) {
        dst[0] = *p;
    } else if constexpr (VEC == 2) {
        const stg_f2u v = *reinterpret_cast<const stg_f2u *>(p);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        const stg_f4u v = *reinterpret_cast<const stg_f4u *>(p);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

template <int VEC>
__device__ __forceinline__ void vec_store_g(float *p, const float (&src)[VEC])
{
    if constexpr (VEC == 1) {
        *p = src[0];
    } else if constexpr (VEC == 2) {
        stg_f2u v; v.x = src[0]; v.y = src[1];
        *reinterpret_cast<stg_f2u *>(p) = v;
    } else {
        stg_f4u v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
        *reinterpret_cast<stg_f4u *>(p) = v;
    }
}

// The same accesses with the lane's VEC floats kept as ONE vector value (float for VEC = 1): a conditional gather
// then merges as a single 64/128-bit register tuple instead of VEC scalar values, and `vector * scalar`,
// `vector + vector` become packed fp32 instructions.
template <int VEC> struct GVec;
template <> struct GVec<1> { typedef float type; };
template <> struct GVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct GVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int VEC> using gvec_t = typename GVec<VEC>::type;

template <int VEC>
__device__ __forceinline__ gvec_t<VEC> gvec_load(const float *p)
{
    if constexpr (VEC == 1) return *p;
    else if constexpr (VEC == 2) return *reinterpret_cast<const stg_f2u *>(p);
    else return *reinterpret_cast<const stg_f4u *>(p);
}

template <int VEC>
__device__ __forceinline__ gvec_t<VEC> gvec_zero()
{
    if constexpr (VEC == 1) return 0.f;
    else return (gvec_t<VEC>)(0.f);
}

template <int VEC>
__device__ __forceinline__ float gvec_get(const gvec_t<VEC> &v, int i)
{
    if constexpr (VEC == 1) return v;
    else return v[i];
}

// max over the 64 lanes of a wave (all lanes must be active)
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Same for non-negative values, on the DPP network (no LDS crossbar trips): running max along each row of 16
// lanes (row_shr 1, 2, 4, 8), then row_bcast:15 / row_bcast:31 carry it across rows; lane 63 holds the result,
// returned wave-uniform.  Lanes whose DPP source is out of range keep their own value (old = v).
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
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


// Sum over the 16 lanes of each DPP row (row_shl 1, 2, 4, 8; lanes shifted in from outside the row contribute 0):
// the total is valid in lane 0 of the row.  Four VALU instructions, no LDS-pipe traffic (a __shfl_xor butterfly is four
// ds_bpermute_b32).
__device__ __forceinline__ float row16_sum_lane0(float v)
{
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x102, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x108, 0xf, 0xf, false));
    return v;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per kernel AND per device: a launcher keeps one of these as a
// function-local static and raises the limit the first time it launches on each device of the process.
struct PerDeviceOnce {
    bool done[64] = {};
    // returns the slot of the calling thread's current device (devices >= 64 share slot 63 and simply re-raise)
    bool *slot()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
        if (dev >= 63) { done[63] = false; return &done[63]; }
        return &done[dev];
    }
};
}  // namespace stg
