// Run-time compilation and launch of generated vertex-function kernels (stgraph_hip.h, "JIT").
// Replaces the reference's nvcc -> PTX -> cuModuleLoadData -> cuModuleGetFunction -> cuLaunchKernel chain
// (compiler/code_gen/compiler.py:14-44, compiler/execution_unit.py:241-269,359-372) with
// hiprtc -> code object -> hipModuleLoadData -> hipModuleGetFunction -> hipModuleLaunchKernel.
// Compilation needs no GPU (gfx950 is named explicitly), loading and launching do.
#include "stg_common.hpp"

#include <hip/hiprtc.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int stg_jit_compile(const char *source, const char *name, char **code_out, size_t *code_size_out,
                               char **log_out)
{
    using namespace stg;
    if (!source || !code_out || !code_size_out) return fail(STG_ERR_INVALID_ARGUMENT, "stg_jit_compile: NULL argument");
    *code_out = nullptr;
    *code_size_out = 0;
    if (log_out) *log_out = nullptr;
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, source, name ? name : "stg_generated.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return fail(STG_ERR_JIT, "stg_jit_compile: hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    // -ffp-contract=off: every a*b+c of a vertex function is two roundings, as in the hand-written kernels
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    r = hiprtcCompileProgram(prog, 4, opts);
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &log[0]);
    if (log_out && log_size > 1) {
        *log_out = static_cast<char *>(std::malloc(log_size + 1));
        std::memcpy(*log_out, log.c_str(), log_size);
        (*log_out)[log_size] = '\0';
    }
    if (r != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        return fail(STG_ERR_JIT, "stg_jit_compile: %s\n%.1800s", hiprtcGetErrorString(r), log.c_str());
    }
    size_t size = 0;
    hiprtcGetCodeSize(prog, &size);
    char *code = static_cast<char *>(std::malloc(size ? size : 1));
    r = hiprtcGetCode(prog, code);
    hiprtcDestroyProgram(&prog);
    if (r != HIPRTC_SUCCESS) {
        std::free(code);
        return fail(STG_ERR_JIT, "stg_jit_compile: hiprtcGetCode: %s", hiprtcGetErrorString(r));
    }
    *code_out = code;
    *code_size_out = size;
    return 0;
}

extern "C" void stg_jit_free(void *p) { std::free(p); }

extern "C" int stg_jit_load(const void *code, void **module_out)
{
    using namespace stg;
    if (!code || !module_out) return fail(STG_ERR_INVALID_ARGUMENT, "stg_jit_load: NULL argument");
    hipModule_t m;
    const hipError_t e = hipModuleLoadData(&m, code);
    if (e != hipSuccess) return fail((int)e, "stg_jit_load: hipModuleLoadData: %s", hipGetErrorString(e));
    *module_out = m;
    return 0;
}

extern "C" int stg_jit_get_function(void *module, const char *name, void **function_out)
{
    using namespace stg;
    if (!module || !name || !function_out) return fail(STG_ERR_INVALID_ARGUMENT, "stg_jit_get_function: NULL argument");
    hipFunction_t f;
    const hipError_t e = hipModuleGetFunction(&f, static_cast<hipModule_t>(module), name);
    if (e != hipSuccess) return fail((int)e, "stg_jit_get_function(%s): %s", name, hipGetErrorString(e));
    *function_out = f;
    return 0;
}

extern "C" int stg_jit_unload(void *module)
{
    using namespace stg;
    if (!module) return 0;
    const hipError_t e = hipModuleUnload(static_cast<hipModule_t>(module));
    if (e != hipSuccess) return fail((int)e, "stg_jit_unload: %s", hipGetErrorString(e));
    return 0;
}

// Kernel ABI of every generated kernel: n_ptr pointer arguments, then n_int int32 arguments.
extern "C" int stg_jit_launch(void *function, uint32_t grid, uint32_t block, const void *const *ptr_args, int32_t n_ptr,
                              const int32_t *int_args, int32_t n_int, void *stream)
{
    using namespace stg;
    if (!function || n_ptr < 0 || n_int < 0 || (n_ptr > 0 && !ptr_args) || (n_int > 0 && !int_args))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_jit_launch: bad argument");
    if (grid == 0) return 0;
    if (block == 0 || block > 1024) return fail(STG_ERR_INVALID_ARGUMENT, "stg_jit_launch: block size %u", block);
    std::vector<const void *> ptrs(ptr_args, ptr_args + n_ptr);
    std::vector<int32_t> ints(int_args, int_args + n_int);
    std::vector<void *> params;
    params.reserve((size_t)n_ptr + n_int);
    for (auto &p : ptrs) params.push_back(&p);
    for (auto &i : ints) params.push_back(&i);
    const hipError_t e = hipModuleLaunchKernel(static_cast<hipFunction_t>(function), grid, 1, 1, block, 1, 1, 0,
                                               static_cast<hipStream_t>(stream), params.data(), nullptr);
    if (e != hipSuccess) return fail((int)e, "stg_jit_launch: %s", hipGetErrorString(e));
    return 0;
}
