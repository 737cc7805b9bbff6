// Error reporting, ABI version and tuning knobs of libstgraph_hip.so.
#include <algorithm>

#include "stg_common.hpp"

#include <cstring>

namespace stg {

char *last_error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
    return 0;
}

Tuning &tuning()
{
    static Tuning t;
    return t;
}

}  // namespace stg

namespace stg {
namespace {
__global__ __launch_bounds__(kBlock) void zero_words_kernel(uint32_t *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = 0u;
}
__global__ __launch_bounds__(kBlock) void copy_words_kernel(uint32_t *__restrict__ d, const uint32_t *__restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) d[i] = s[i];
}
}  // namespace

int zero_async(void *dst, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    if (!dst || (bytes & 3) || (reinterpret_cast<uintptr_t>(dst) & 3)) return fail(STG_ERR_INVALID_ARGUMENT, "zero_async: bad destination");
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0, stream,
                       static_cast<uint32_t *>(dst), n);
    return check_launch("zero_async");
}

int copy_async(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    if (!dst || !src || (bytes & 3) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3))
        return fail(STG_ERR_INVALID_ARGUMENT, "copy_async: bad operands");
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0, stream,
                       static_cast<uint32_t *>(dst), static_cast<const uint32_t *>(src), n);
    return check_launch("copy_async");
}
}  // namespace stg

extern "C" int stg_abi_version(void) { return STG_ABI_VERSION; }

extern "C" const char *stg_last_error_string(void) { return stg::last_error_buffer(); }

extern "C" int stg_set_tuning(const char *key, int value)
{
    using namespace stg;
    if (!key) return fail(STG_ERR_INVALID_ARGUMENT, "stg_set_tuning: NULL key");
    if (!std::strcmp(key, "gcn_lanes_per_row")) {
        if (value != 0 && (value < 1 || value > 64 || (value & (value - 1))))
            return fail(STG_ERR_INVALID_ARGUMENT, "gcn_lanes_per_row must be 0 or a power of two <= 64");
        tuning().gcn_lanes_per_row = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_unroll")) {
        if (value != 0 && value != 2 && value != 4 && value != 8)
            return fail(STG_ERR_INVALID_ARGUMENT, "gcn_unroll must be 0, 2, 4 or 8");
        tuning().gcn_unroll = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_long_threshold")) {
        if (value < 0) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_long_threshold must be >= 0");
        tuning().gcn_long_threshold = value;
        return 0;
    }
    if (!std::strcmp(key, "xw_rows")) {
        if (value != 0 && value != 32 && value != 64) return fail(STG_ERR_INVALID_ARGUMENT, "xw_rows must be 0, 32 or 64");
        tuning().xw_rows = value;
        return 0;
    }
    if (!std::strcmp(key, "xw_waves")) {
        if (value != 0 && value != 4 && value != 8) return fail(STG_ERR_INVALID_ARGUMENT, "xw_waves must be 0, 4 or 8");
        tuning().xw_waves = value;
        return 0;
    }
    if (!std::strcmp(key, "cell_rows")) {
        if (value != 0 && value != 16 && value != 32) return fail(STG_ERR_INVALID_ARGUMENT, "cell_rows must be 0, 16 or 32");
        tuning().cell_rows = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_tile")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_tile must be 0, 1 or 2");
        tuning().gcn_tile = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_block")) {
        if (value != 0 && value != 64 && value != 128 && value != 256)
            return fail(STG_ERR_INVALID_ARGUMENT, "gcn_block must be 0, 64, 128 or 256");
        tuning().gcn_block = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_addr32")) {
        if (value != 0 && value != 1) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_addr32 must be 0 or 1");
        tuning().gcn_addr32 = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_tile_pipe")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_tile_pipe must be 0, 1 or 2");
        tuning().gcn_tile_pipe = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_tile_rows")) {
        if (value < 0 || value > 256 || (value > 0 && value < 8))
            return fail(STG_ERR_INVALID_ARGUMENT, "gcn_tile_rows must be 0 or in [8, 256]");
        tuning().gcn_tile_rows = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_xcd_tile")) {
        if (value < 0 || value > 4096) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_xcd_tile must be in [0, 4096]");
        tuning().gcn_xcd_tile = value;
        return 0;
    }
    if (!std::strcmp(key, "step_waves")) {
        if (value != 0 && value != 12 && value != 16) return fail(STG_ERR_INVALID_ARGUMENT, "step_waves must be 0, 12 or 16");
        tuning().step_waves = value;
        return 0;
    }
    if (!std::strcmp(key, "gcn_wide_long")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "gcn_wide_long must be 0 (auto), 1 (never) or 2 (behind the main launch, same stream)");
        tuning().gcn_wide_long = value;
        return 0;
    }
    if (!std::strcmp(key, "step_spread")) { tuning().step_spread = value; return 0; }
    if (!std::strcmp(key, "step_coop")) { tuning().step_coop = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "build_lds_count")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "build_lds_count must be 0 (auto), 1 (always when |V| fits) or 2 (never)");
        tuning().build_lds_count = value;
        return 0;
    }
    if (!std::strcmp(key, "store_rows")) { tuning().store_rows = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "rowgemm16")) { tuning().rowgemm16 = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "rowgemm_x3")) {
        if (value < 0 || value > 3) return fail(STG_ERR_INVALID_ARGUMENT, "rowgemm_x3 must be 0 .. 3");
        tuning().rowgemm_x3 = value;
        return 0;
    }
    if (!std::strcmp(key, "gemm_x3")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "gemm_x3 must be 0 (auto), 1 (never) or 2 (whenever covered)");
        tuning().gemm_x3 = value;
        return 0;
    }
    if (!std::strcmp(key, "gemm_wide")) {
        if (value != 0 && value != 1) return fail(STG_ERR_INVALID_ARGUMENT, "gemm_wide must be 0 (auto) or 1 (never)");
        tuning().gemm_wide = value;
        return 0;
    }
    if (!std::strcmp(key, "gemm_xcd_pair")) { tuning().gemm_xcd_pair = value ? 1 : 0; return 0; }
    if (!std::strcmp(key, "gemm_cyclic")) {
        if (value < 0 || value > 2) return fail(STG_ERR_INVALID_ARGUMENT, "gemm_cyclic must be 0, 1 or 2");
        tuning().gemm_cyclic = value;
        return 0;
    }
    return fail(STG_ERR_INVALID_ARGUMENT, "stg_set_tuning: unknown key '%s'", key);
}
