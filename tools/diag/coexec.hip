// Does a vector instruction issue in the shadow of an f32-input MFMA on gfx950?  (diagnosis tool, not product code)
//
// One wave per SIMD (or two / three, --waves), every CU busy.  Each wave runs ITER x 16 MFMAs, K independent filler
// instructions behind each MFMA, all in one asm block (nothing for the compiler to move), and stamps s_memtime around it.
// Printed: shader cycles per MFMA for K = 0, 1, 2, 4, 6, 8 and for each filler kind.  If the matrix instruction ran beside
// the vector ALU, cycles per MFMA would stay at its issue interval until the fillers' own issue cost fills the gap
// (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'); if it runs ON the vector ALU's lanes, every filler adds its full cost.
//
//   hipcc -O3 --offload-arch=gfx950 tools/diag/coexec.hip -o build/coexec && build/coexec
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// operands: %0-%3 accumulators, %4-%11 filler registers, %12 %13 MFMA inputs, %14 %15 filler inputs
#define M32(n) "v_mfma_f32_16x16x4_f32 %" #n ", %12, %13, %" #n "\n"
#define MBF(n) "v_mfma_f32_16x16x32_bf16 %" #n ", %16, %17, %" #n "\n"
#define F_FMA(n) "v_fma_f32 %" #n ", %14, %15, %" #n "\n"
#define F_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define F_MOV(n) "v_mov_b32 %" #n ", %14\n"
#define F_CNDMASK(n) "v_cndmask_b32 %" #n ", %14, %15, vcc\n"
#define F_NOP(n) "s_nop 0\n"

#define FILL0(F)
#define FILL1(F) F(4)
#define FILL2(F) F(4) F(5)
#define FILL3(F) F(4) F(5) F(6)
#define FILL4(F) F(4) F(5) F(6) F(7)
#define FILL6(F) F(4) F(5) F(6) F(7) F(8) F(9)
#define FILL8(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
#define FILL12(F) FILL8(F) FILL4(F)
#define FILL16(F) FILL8(F) FILL8(F)

#define BODY4(M, FILL, F) M(0) FILL(F) M(1) FILL(F) M(2) FILL(F) M(3) FILL(F)
#define BODY16(M, FILL, F) BODY4(M, FILL, F) BODY4(M, FILL, F) BODY4(M, FILL, F) BODY4(M, FILL, F)

#define MNONE(n)

template <int MODE>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&f)[8], float a, float b, float c, float d, bf16x8 pa, bf16x8 pb)
{
#define RUN(M, FILL, F)                                                                                                          \
    asm volatile(BODY16(M, FILL, F)                                                                                              \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]),       \
                   "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])                                                                \
                 : "v"(a), "v"(b), "v"(c), "v"(d), "v"(pa), "v"(pb))
    if constexpr (MODE == 0) RUN(M32, FILL0, F_FMA);
    if constexpr (MODE == 1) RUN(M32, FILL1, F_FMA);
    if constexpr (MODE == 2) RUN(M32, FILL2, F_FMA);
    if constexpr (MODE == 3) RUN(M32, FILL4, F_FMA);
    if constexpr (MODE == 4) RUN(M32, FILL6, F_FMA);
    if constexpr (MODE == 5) RUN(M32, FILL8, F_FMA);
    if constexpr (MODE == 6) RUN(M32, FILL2, F_EXP);
    if constexpr (MODE == 7) RUN(M32, FILL4, F_EXP);
    if constexpr (MODE == 8) RUN(M32, FILL4, F_MOV);
    if constexpr (MODE == 9) RUN(M32, FILL4, F_NOP);
    if constexpr (MODE == 10) RUN(MBF, FILL0, F_FMA);
    if constexpr (MODE == 11) RUN(MBF, FILL1, F_FMA);
    if constexpr (MODE == 12) RUN(MBF, FILL2, F_FMA);
    if constexpr (MODE == 13) RUN(MBF, FILL4, F_FMA);
    if constexpr (MODE == 14) RUN(MNONE, FILL4, F_FMA);       // 64 fillers and no MFMA: the fillers' own cost
    if constexpr (MODE == 15) RUN(MNONE, FILL4, F_EXP);
    if constexpr (MODE == 16) RUN(M32, FILL12, F_FMA);
    if constexpr (MODE == 17) RUN(M32, FILL16, F_FMA);
    if constexpr (MODE == 18) RUN(MBF, FILL8, F_FMA);
    if constexpr (MODE == 19) RUN(M32, FILL4, F_CNDMASK);
#undef RUN
}

// role: 0 = every wave runs MODE; 1 = waves >= 4 of the workgroup run MODE_B instead (partners on one SIMD with different streams)
template <int MODE, int MODE_B>
__global__ void k(unsigned long long *out, float *sink, int iters, int split)
{
    f32x4 acc[4];
    float f[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) f[i] = 0.001f * (float)(threadIdx.x + i);
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f, c = 0.999f, d = 1e-3f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) pa[i] = (short)(0x3f80 + i), pb[i] = (short)(0x3f00 + (threadIdx.x & 3));
    const int wave = threadIdx.x >> 6;
    const bool second = split && wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
        for (int it = 0; it < iters; ++it) body<MODE>(acc, f, a, b, c, d, pa, pb);
    } else {
        for (int it = 0; it < iters; ++it) body<MODE_B>(acc, f, a, b, c, d, pa, pb);
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE, int MODE_B>
int run(const char *what, int waves, int split, int n_mfma_first, int n_mfma_second)
{
    const int blocks = 256, iters = 2000;
    unsigned long long *out;
    float *sink;
    const int nw = blocks * waves;
    CK(hipMalloc(&out, nw * sizeof(*out)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, MODE_B>), dim3(blocks), dim3(64 * waves), 0, 0, out, sink, iters, split);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nw);
    CK(hipMemcpy(h.data(), out, nw * sizeof(*out), hipMemcpyDeviceToHost));
    // median over the waves of each role
    std::vector<double> first, second;
    for (int i = 0; i < nw; ++i) ((split && (i % waves) >= 4) ? second : first).push_back((double)h[i]);
    auto med = [](std::vector<double> &v) {
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    const double c1 = med(first) / ((double)iters * 16), c2 = med(second) / ((double)iters * 16);
    std::printf("{\"case\": \"%s\", \"waves_per_cu\": %d, \"cycles_per_slot_first\": %.2f, \"cycles_per_slot_second\": %.2f, "
                "\"launch_ms\": %.4f, \"mfma_per_slot\": [%d, %d]}\n",
                what, waves, c1, c2, ms, n_mfma_first, n_mfma_second);
    std::fflush(stdout);
    (void)hipFree(out);
    (void)hipFree(sink);
    return 0;
}

int main()
{
    // one wave per SIMD: a slot = one MFMA + K fillers
    if (run<0, 0>("f32mfma + 0 fma", 4, 0, 1, 0)) return 1;
    if (run<1, 0>("f32mfma + 1 fma", 4, 0, 1, 0)) return 1;
    if (run<2, 0>("f32mfma + 2 fma", 4, 0, 1, 0)) return 1;
    if (run<3, 0>("f32mfma + 4 fma", 4, 0, 1, 0)) return 1;
    if (run<4, 0>("f32mfma + 6 fma", 4, 0, 1, 0)) return 1;
    if (run<5, 0>("f32mfma + 8 fma", 4, 0, 1, 0)) return 1;
    if (run<16, 0>("f32mfma + 12 fma", 4, 0, 1, 0)) return 1;
    if (run<17, 0>("f32mfma + 16 fma", 4, 0, 1, 0)) return 1;
    if (run<6, 0>("f32mfma + 2 exp", 4, 0, 1, 0)) return 1;
    if (run<7, 0>("f32mfma + 4 exp", 4, 0, 1, 0)) return 1;
    if (run<8, 0>("f32mfma + 4 mov", 4, 0, 1, 0)) return 1;
    if (run<19, 0>("f32mfma + 4 cndmask", 4, 0, 1, 0)) return 1;
    if (run<9, 0>("f32mfma + 4 s_nop", 4, 0, 1, 0)) return 1;
    if (run<14, 0>("4 fma alone", 4, 0, 0, 0)) return 1;
    if (run<15, 0>("4 exp alone", 4, 0, 0, 0)) return 1;
    if (run<10, 0>("bf16mfma + 0 fma", 4, 0, 1, 0)) return 1;
    if (run<11, 0>("bf16mfma + 1 fma", 4, 0, 1, 0)) return 1;
    if (run<12, 0>("bf16mfma + 2 fma", 4, 0, 1, 0)) return 1;
    if (run<13, 0>("bf16mfma + 4 fma", 4, 0, 1, 0)) return 1;
    if (run<18, 0>("bf16mfma + 8 fma", 4, 0, 1, 0)) return 1;
    // two waves per SIMD, same stream
    if (run<0, 0>("2 waves/simd: f32mfma + 0", 8, 0, 1, 0)) return 1;
    if (run<3, 0>("2 waves/simd: f32mfma + 4 fma", 8, 0, 1, 0)) return 1;
    if (run<5, 0>("2 waves/simd: f32mfma + 8 fma", 8, 0, 1, 0)) return 1;
    if (run<3, 0>("3 waves/simd: f32mfma + 4 fma", 12, 0, 1, 0)) return 1;
    // two waves per SIMD, different streams: one only MFMAs, its partner only vector instructions (4 per slot)
    if (run<0, 14>("split: f32mfma only | 4 fma only", 8, 1, 1, 0)) return 1;
    if (run<10, 14>("split: bf16mfma only | 4 fma only", 8, 1, 1, 0)) return 1;
    if (run<0, 15>("split: f32mfma only | 4 exp only", 8, 1, 1, 0)) return 1;
    if (run<0, 10>("split: f32mfma only | bf16mfma only", 8, 1, 1, 1)) return 1;
    return 0;
}
