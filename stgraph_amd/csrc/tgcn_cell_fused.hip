// One TGCN step's row-local part as ONE kernel (reference nn/pytorch/temporal/tgcn.py:21-55):
//   h_g = clamp(a3[:, g] + b3[g])                       g in {z, r, h}      (GCNConv bias + clamp, :22,30,38)
//   Z = sigmoid([hz | H] Wz^T + bz)   R = sigmoid([hr | H] Wr^T + br)       (:24-27, :32-35)
//   Ht = tanh([hh | H*R] Wh^T + bh)   Hn = Z*H + (1 - Z)*Ht                 (:40-47)
// The unfused form runs 3 elementwise kernels around 3 rocBLAS GEMMs of 0.8 GFLOP each; at |V| = 50K each of
// those GEMMs is 17-30 us of mostly fixed cost (launch, weight staging, first-load latency), 3-6x their matrix
// time.  Here one wave owns a 32-row tile for the whole chain:
//   * the three weight matrices (torch Linear layout [C][2C]) are staged ONCE per workgroup into LDS, transposed
//     to [k][C + 1];
//   * A operands come straight from global memory in MFMA layout (lane (row, kh) holds the float4s
//     x[row][8 j + 4 kh ..], k permuted inside blocks of 8 -- see rowgemm.hip); bias + clamp are applied in
//     those registers, which are also what gets written out as the concatenated GEMM operands [h_g | H] the
//     backward pass (weight gradients) reads;
//   * Z, R, Ht come out of v_mfma_f32_32x32x2_f32 in accumulator layout and go through a per-wave LDS tile
//     (one 32x32 block at a time) to the A layout, where a lane holds pieces of its own row: H*R and the GRU
//     blend happen there against the H registers, and every [N,C] output leaves as 16-byte row pieces.
// Measured at |V| = 50K, C = 64.  With one wave per SIMD (4-wave workgroups, the 100 KB of weights allow one per CU)
// the phases simply add up (switched off one at a time: loads + elementwise 23 us, MFMA +22, operand stores +20,
// [N,C] stores +21, staging +4 = 86-100 us); with 8-wave workgroups -- two waves per SIMD, 256 registers each --
// one wave's memory phases overlap the other's MFMAs: 75-79 us, against ~135 us for the six launches it replaces.
// 180 MB of traffic, most of it the saved operands: the next step is to stop saving the concatenations
// (DESIGN.md section 8).
// fp32 in / fp32 accumulate.  Against the unfused stages of tgcn_cell.hip: the GEMM k-order differs from rocBLAS'
// and sigmoid / tanh use the hardware exp2 / rcp forms, so outputs agree to ~1e-6 (tested at 1e-5).
#include <algorithm>

#include "stg_common.hpp"

namespace stg {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// One wave per SIMD runs the whole chain, so the 192 transcendental evaluations per lane and tile are on the
// critical path next to the MFMAs: hardware exp2 / rcp forms (|error| < 2e-7 absolute on (0,1) / (-1,1) outputs).
__device__ __forceinline__ float sigmoid_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ void wave_lds_sync() { wave_lds_fence(); }

template <int C, int WAVES>
struct CellShape {
    static constexpr int K = 2 * C, KB = K / 8, KH = C / 8, MT = C / 32, LDW = C + 1, TLD = 33;
    static constexpr int kThreads = WAVES * kWave;
    static constexpr int kWeights = 3 * K * LDW;              // floats
    static constexpr int kBias = 6 * C;                       // b3 [3C], bz, br, bh
    static constexpr int kTile = 32 * TLD;                    // per wave: one 32x32 block of R at a time
    static constexpr size_t kLds = sizeof(float) * (size_t)(kWeights + kBias + WAVES * kTile);
};

// row of accumulator element i for lane half kh (v_mfma_f32_32x32x2_f32 C/D layout)
__device__ __forceinline__ int acc_row(int i, int kh) { return (i & 3) + 8 * (i >> 2) + 4 * kh; }

// WAVES = 8 (C = 64: the 100 KB of weights allow one workgroup per CU, so the workgroup itself brings two waves per
// SIMD -- one wave's loads and stores overlap the other's MFMAs) or 4 (C = 32: three workgroups per CU).
template <int C, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void cell_fused_fwd_kernel(
    const float *__restrict__ a3, const float *__restrict__ b3, const float *__restrict__ H,
    const float *__restrict__ Wz, const float *__restrict__ bz, const float *__restrict__ Wr,
    const float *__restrict__ br, const float *__restrict__ Wh, const float *__restrict__ bh,
    float *__restrict__ CZ, float *__restrict__ CR, float *__restrict__ CH, float *__restrict__ Z,
    float *__restrict__ R, float *__restrict__ Ht, float *__restrict__ Hn, int64_t N, float lo, float hi,
    int num_tiles)
{
    using S = CellShape<C, WAVES>;
    constexpr int K = S::K, KH = S::KH, MT = S::MT, LDW = S::LDW, TLD = S::TLD, NT = S::kThreads;
    extern __shared__ float lds[];
    float *Ws = lds;                                   // 3 x [K][LDW]
    float *bs = lds + S::kWeights;                     // b3 | bz | br | bh
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    float *T = bs + S::kBias + wave * S::kTile;        // [32][TLD] transpose tile of this wave

    {   // stage the weights (W is [C][K]: element i = (m, k), 4 consecutive k) and the biases
        const float *src[3] = {Wz, Wr, Wh};
        constexpr int total4 = C * K / 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float *dst = Ws + g * K * LDW;
            for (int base = 0; base < total4; base += 8 * NT) {
                float4 w4[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i4 = base + s * NT + threadIdx.x;
                    w4[s] = i4 < total4 ? *reinterpret_cast<const float4 *>(src[g] + (int64_t)i4 * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i = (base + s * NT + threadIdx.x) * 4;
                    if (i < C * K) {
                        const int m = i / K, k = i - m * K;
                        dst[(k + 0) * LDW + m] = w4[s].x;
                        dst[(k + 1) * LDW + m] = w4[s].y;
                        dst[(k + 2) * LDW + m] = w4[s].z;
                        dst[(k + 3) * LDW + m] = w4[s].w;
                    }
                }
            }
        }
        for (int i = threadIdx.x; i < 3 * C; i += NT) bs[i] = b3[i];
        for (int i = threadIdx.x; i < C; i += NT) {
            bs[3 * C + i] = bz[i];
            bs[4 * C + i] = br[i];
            bs[5 * C + i] = bh[i];
        }
    }
    __syncthreads();

    const int total = gridDim.x * WAVES;
    // tiles are dealt wave-major (wave w of workgroup b: tile w * grid + b), so every CU gets its share of a grid
    // smaller than WAVES * CUs (cfg4: 1563 tiles = 6.1 per CU on 256 CUs instead of 8 on 196)
    for (int tile = wave * (int)gridDim.x + (int)blockIdx.x; tile < num_tiles; tile += total) {
        const int64_t row_base = (int64_t)tile * 32;
        const int64_t row = row_base + l31;
        const bool rok = row < N;

        // ---- A operands: h_g = clamp(a3[:, g] + b3[g]) and H, in MFMA layout; written out as [h_g | H] ------
        // (the candidate's block g = 2 is fetched after the z / r GEMMs: 32 registers less while they run)
        float4 ag[3][KH], hh[KH];
        auto load_gate = [&](int g, int j) {
            const int c = 8 * j + 4 * kh;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) a = *reinterpret_cast<const float4 *>(a3 + row * 3 * C + g * C + c);
            const float4 b = *reinterpret_cast<const float4 *>(bs + g * C + c);
            a.x = fminf(fmaxf(a.x + b.x, lo), hi);
            a.y = fminf(fmaxf(a.y + b.y, lo), hi);
            a.z = fminf(fmaxf(a.z + b.z, lo), hi);
            a.w = fminf(fmaxf(a.w + b.w, lo), hi);
            return a;
        };
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const int c = 8 * j + 4 * kh;
            hh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) hh[j] = *reinterpret_cast<const float4 *>(H + row * C + c);
            ag[0][j] = load_gate(0, j);
            ag[1][j] = load_gate(1, j);
        }
        if (rok) {
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const int c = 8 * j + 4 * kh;
                *reinterpret_cast<float4 *>(CZ + row * K + c) = ag[0][j];
                *reinterpret_cast<float4 *>(CR + row * K + c) = ag[1][j];
                *reinterpret_cast<float4 *>(CZ + row * K + C + c) = hh[j];
                *reinterpret_cast<float4 *>(CR + row * K + C + c) = hh[j];
            }
        }

        // ---- zl = [hz | H] Wz^T + bz,  rl = [hr | H] Wr^T + br  ----------------------------------------------
        f32x16 accz[MT], accr[MT];
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
            const float vz = bs[3 * C + ct * 32 + l31], vr = bs[4 * C + ct * 32 + l31];
#pragma unroll
            for (int i = 0; i < 16; ++i) accz[ct][i] = vz, accr[ct][i] = vr;
        }
        const float *pz = Ws + (4 * kh) * LDW + l31, *pr = pz + K * LDW, *ph = pr + K * LDW;
#pragma unroll
        for (int j = 0; j < 2 * KH; ++j) {
            const float4 az4 = j < KH ? ag[0][j % KH] : hh[j % KH];
            const float4 ar4 = j < KH ? ag[1][j % KH] : hh[j % KH];
            const float az[4] = {az4.x, az4.y, az4.z, az4.w}, ar[4] = {ar4.x, ar4.y, ar4.z, ar4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ct = 0; ct < MT; ++ct) {
                    accz[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(az[i], pz[(8 * j + i) * LDW + ct * 32], accz[ct], 0, 0, 0);
                    accr[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i], pr[(8 * j + i) * LDW + ct * 32], accr[ct], 0, 0, 0);
                }
            }
        }

        // ---- Z, R: accumulator layout -> LDS tile (one 32x32 block at a time) -> A layout (a lane's own row), where
        //      H lives: H*R, and later the GRU blend; all [N,C] outputs leave as 16-byte row pieces -----------------
        float4 hr[KH], za[KH];
        auto to_rows = [&](const f32x16 &acc, int ct, float4 (&dst)[KH]) {      // 32x32 block ct of acc -> dst[4 ct ..]
#pragma unroll
            for (int i = 0; i < 16; ++i) T[acc_row(i, kh) * TLD + l31] = acc[i];
            wave_lds_sync();
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {                         // columns 32 ct .. 32 ct + 31 = j in [4 ct, 4 ct + 4)
                const float *t = T + l31 * TLD + 8 * jj + 4 * kh;
                dst[4 * ct + jj] = make_float4(t[0], t[1], t[2], t[3]);
            }
            wave_lds_sync();                                         // T is rewritten by the next block / tile
        };
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) accz[ct][i] = sigmoid_(accz[ct][i]), accr[ct][i] = sigmoid_(accr[ct][i]);
            to_rows(accz[ct], ct, za);
            to_rows(accr[ct], ct, hr);                               // hr holds R until multiplied below
        }
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const int c = 8 * j + 4 * kh;
            if (rok) {
                *reinterpret_cast<float4 *>(Z + row * C + c) = za[j];
                *reinterpret_cast<float4 *>(R + row * C + c) = hr[j];
            }
            hr[j] = make_float4(hh[j].x * hr[j].x, hh[j].y * hr[j].y, hh[j].z * hr[j].z, hh[j].w * hr[j].w);
        }
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            ag[2][j] = load_gate(2, j);
            if (rok) {
                *reinterpret_cast<float4 *>(CH + row * K + 8 * j + 4 * kh) = ag[2][j];
                *reinterpret_cast<float4 *>(CH + row * K + C + 8 * j + 4 * kh) = hr[j];
            }
        }

        // ---- hl = [hh | H*R] Wh^T + bh;  Ht = tanh(hl);  Hn = Z*H + (1 - Z)*Ht --------------------------------
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
            const float vh = bs[5 * C + ct * 32 + l31];
#pragma unroll
            for (int i = 0; i < 16; ++i) accr[ct][i] = vh;
        }
#pragma unroll
        for (int j = 0; j < 2 * KH; ++j) {
            const float4 a4 = j < KH ? ag[2][j % KH] : hr[j % KH];
            const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ct = 0; ct < MT; ++ct)
                    accr[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], ph[(8 * j + i) * LDW + ct * 32], accr[ct], 0, 0, 0);
            }
        }
        float4 ht[KH];
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) accr[ct][i] = tanh_(accr[ct][i]);
            to_rows(accr[ct], ct, ht);
        }
        if (rok) {
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const int c = 8 * j + 4 * kh;
                const float4 z = za[j], h = hh[j], t = ht[j];
                *reinterpret_cast<float4 *>(Ht + row * C + c) = t;
                *reinterpret_cast<float4 *>(Hn + row * C + c) =
                    make_float4(z.x * h.x + (1.0f - z.x) * t.x, z.y * h.y + (1.0f - z.y) * t.y,
                                z.z * h.z + (1.0f - z.z) * t.z, z.w * h.w + (1.0f - z.w) * t.w);
            }
        }
    }
}

template <int C, int WAVES>
int launch_fwd(const float *a3, const float *b3, const float *H, const float *Wz, const float *bz, const float *Wr,
               const float *br, const float *Wh, const float *bh, float *CZ, float *CR, float *CH, float *Z, float *R,
               float *Ht, float *Hn, int64_t N, float lo, float hi, hipStream_t stream)
{
    using S = CellShape<C, WAVES>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cell_fused_fwd_kernel<C, WAVES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_cell_fused_fwd: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int64_t tiles = (N + 31) / 32;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_fwd: too many rows");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (S::kLds + 512), 32 / WAVES));
    const unsigned blocks = (unsigned)std::min<int64_t>(tiles, 256 * per_cu);
    hipLaunchKernelGGL((cell_fused_fwd_kernel<C, WAVES>), dim3(blocks), dim3(S::kThreads), S::kLds, stream, a3, b3, H, Wz, bz,
                       Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, (int)tiles);
    return check_launch("stg_tgcn_cell_fused_fwd");
}


// ---------------------------------------------------------------------------------------------- forward, 16-row tiles
// The same chain on v_mfma_f32_16x16x4_f32 with one wave per 16-ROW tile: twice the tiles (cfg4: 3125 instead of
// 1563, i.e. 12 instead of 6 per CU) at half the work each, and 16-wave workgroups (four waves per SIMD at <= 128
// registers) so that more of one wave's load / LDS / store phases hide behind other waves' MFMAs.
// Layouts: lane = (n16 = lane & 15, kq = lane >> 4).  A operand of step i of k block j: element k = 16 j + 4 kq + i of
// row n16 -- a lane's 16-byte piece at column 16 j + 4 kq, as read from HBM; B operand: Ws[k][16 ct + n16]
// (LDW = C + 4 keeps the four k rows of a step on different LDS banks); accumulator element i of block ct:
// row 4 kq + i, column 16 ct + n16.
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int C, int WAVES>
struct CellShape16 {
    static constexpr int K = 2 * C, KQ = C / 16, CT = C / 16, LDW = C + 4, TLD = 17;
    static constexpr int kThreads = WAVES * kWave;
    static constexpr int kWeights = 3 * K * LDW;              // floats
    static constexpr int kBias = 6 * C;                       // b3 [3C], bz, br, bh
    static constexpr int kTile = 16 * TLD;                    // per wave: one 16x16 block at a time
    static constexpr size_t kLds = sizeof(float) * (size_t)(kWeights + kBias + WAVES * kTile);
};

template <int C, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void cell_fused_fwd16_kernel(
    const float *__restrict__ a3, const float *__restrict__ b3, const float *__restrict__ H,
    const float *__restrict__ Wz, const float *__restrict__ bz, const float *__restrict__ Wr,
    const float *__restrict__ br, const float *__restrict__ Wh, const float *__restrict__ bh,
    float *__restrict__ CZ, float *__restrict__ CR, float *__restrict__ CH, float *__restrict__ Z,
    float *__restrict__ R, float *__restrict__ Ht, float *__restrict__ Hn, int64_t N, float lo, float hi,
    int num_tiles)
{
    using S = CellShape16<C, WAVES>;
    constexpr int K = S::K, KQ = S::KQ, CT = S::CT, LDW = S::LDW, TLD = S::TLD, NT = S::kThreads;
    extern __shared__ float lds[];
    float *Ws = lds;                                   // 3 x [K][LDW]
    float *bs = lds + S::kWeights;                     // b3 | bz | br | bh
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    float *T = bs + S::kBias + wave * S::kTile;        // [16][TLD] transpose tile of this wave

    {   // stage the weights (W is [C][K]: element i = (m, k), 4 consecutive k) and the biases
        const float *src[3] = {Wz, Wr, Wh};
        constexpr int total4 = C * K / 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float *dst = Ws + g * K * LDW;
            for (int i4 = threadIdx.x; i4 < total4; i4 += NT) {
                const float4 w4 = *reinterpret_cast<const float4 *>(src[g] + (int64_t)i4 * 4);
                const int i = i4 * 4, m = i / K, k = i - m * K;
                dst[(k + 0) * LDW + m] = w4.x;
                dst[(k + 1) * LDW + m] = w4.y;
                dst[(k + 2) * LDW + m] = w4.z;
                dst[(k + 3) * LDW + m] = w4.w;
            }
        }
        for (int i = threadIdx.x; i < 3 * C; i += NT) bs[i] = b3[i];
        for (int i = threadIdx.x; i < C; i += NT) {
            bs[3 * C + i] = bz[i];
            bs[4 * C + i] = br[i];
            bs[5 * C + i] = bh[i];
        }
    }
    __syncthreads();

    const int total = gridDim.x * WAVES;
    for (int tile = wave * (int)gridDim.x + (int)blockIdx.x; tile < num_tiles; tile += total) {
        const int64_t row = (int64_t)tile * 16 + n16;
        const bool rok = row < N;

        float4 ag[3][KQ], hh[KQ];
        auto load_gate = [&](int g, int j) {
            const int c = 16 * j + 4 * kq;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) a = *reinterpret_cast<const float4 *>(a3 + row * 3 * C + g * C + c);
            const float4 b = *reinterpret_cast<const float4 *>(bs + g * C + c);
            a.x = fminf(fmaxf(a.x + b.x, lo), hi);
            a.y = fminf(fmaxf(a.y + b.y, lo), hi);
            a.z = fminf(fmaxf(a.z + b.z, lo), hi);
            a.w = fminf(fmaxf(a.w + b.w, lo), hi);
            return a;
        };
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int c = 16 * j + 4 * kq;
            hh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) hh[j] = *reinterpret_cast<const float4 *>(H + row * C + c);
            ag[0][j] = load_gate(0, j);
            ag[1][j] = load_gate(1, j);
        }
        if (rok) {
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const int c = 16 * j + 4 * kq;
                *reinterpret_cast<float4 *>(CZ + row * K + c) = ag[0][j];
                *reinterpret_cast<float4 *>(CR + row * K + c) = ag[1][j];
                *reinterpret_cast<float4 *>(CZ + row * K + C + c) = hh[j];
                *reinterpret_cast<float4 *>(CR + row * K + C + c) = hh[j];
            }
        }

        // ---- zl = [hz | H] Wz^T + bz,  rl = [hr | H] Wr^T + br
        f32x4 accz[CT], accr[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float vz = bs[3 * C + ct * 16 + n16], vr = bs[4 * C + ct * 16 + n16];
#pragma unroll
            for (int i = 0; i < 4; ++i) accz[ct][i] = vz, accr[ct][i] = vr;
        }
        const float *pz = Ws + (4 * kq) * LDW + n16, *pr = pz + K * LDW, *ph = pr + K * LDW;
#pragma unroll
        for (int j = 0; j < 2 * KQ; ++j) {
            const float4 az4 = j < KQ ? ag[0][j % KQ] : hh[j % KQ];
            const float4 ar4 = j < KQ ? ag[1][j % KQ] : hh[j % KQ];
            const float az[4] = {az4.x, az4.y, az4.z, az4.w}, ar[4] = {ar4.x, ar4.y, ar4.z, ar4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    accz[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[i], pz[(16 * j + i) * LDW + ct * 16], accz[ct], 0, 0, 0);
                    accr[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i], pr[(16 * j + i) * LDW + ct * 16], accr[ct], 0, 0, 0);
                }
            }
        }

        // ---- accumulator layout -> per-wave LDS tile (one 16x16 block at a time) -> A layout (a lane's own row)
        float4 hr[KQ], za[KQ];
        auto to_rows = [&](const f32x4 &acc, float4 &dst) {
#pragma unroll
            for (int i = 0; i < 4; ++i) T[(4 * kq + i) * TLD + n16] = acc[i];
            wave_lds_sync();
            const float *t = T + n16 * TLD + 4 * kq;
            dst = make_float4(t[0], t[1], t[2], t[3]);
            wave_lds_sync();                                         // T is rewritten by the next block / tile
        };
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accz[ct][i] = sigmoid_(accz[ct][i]), accr[ct][i] = sigmoid_(accr[ct][i]);
            to_rows(accz[ct], za[ct]);
            to_rows(accr[ct], hr[ct]);                               // hr holds R until multiplied below
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int c = 16 * j + 4 * kq;
            if (rok) {
                *reinterpret_cast<float4 *>(Z + row * C + c) = za[j];
                *reinterpret_cast<float4 *>(R + row * C + c) = hr[j];
            }
            hr[j] = make_float4(hh[j].x * hr[j].x, hh[j].y * hr[j].y, hh[j].z * hr[j].z, hh[j].w * hr[j].w);
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            ag[2][j] = load_gate(2, j);
            if (rok) {
                *reinterpret_cast<float4 *>(CH + row * K + 16 * j + 4 * kq) = ag[2][j];
                *reinterpret_cast<float4 *>(CH + row * K + C + 16 * j + 4 * kq) = hr[j];
            }
        }

        // ---- hl = [hh | H*R] Wh^T + bh;  Ht = tanh(hl);  Hn = Z*H + (1 - Z)*Ht
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float vh = bs[5 * C + ct * 16 + n16];
#pragma unroll
            for (int i = 0; i < 4; ++i) accr[ct][i] = vh;
        }
#pragma unroll
        for (int j = 0; j < 2 * KQ; ++j) {
            const float4 a4 = j < KQ ? ag[2][j % KQ] : hr[j % KQ];
            const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    accr[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], ph[(16 * j + i) * LDW + ct * 16], accr[ct], 0, 0, 0);
            }
        }
        float4 ht[KQ];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accr[ct][i] = tanh_(accr[ct][i]);
            to_rows(accr[ct], ht[ct]);
        }
        if (rok) {
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const int c = 16 * j + 4 * kq;
                const float4 z = za[j], h = hh[j], t = ht[j];
                *reinterpret_cast<float4 *>(Ht + row * C + c) = t;
                *reinterpret_cast<float4 *>(Hn + row * C + c) =
                    make_float4(z.x * h.x + (1.0f - z.x) * t.x, z.y * h.y + (1.0f - z.y) * t.y,
                                z.z * h.z + (1.0f - z.z) * t.z, z.w * h.w + (1.0f - z.w) * t.w);
            }
        }
    }
}

template <int C, int WAVES>
int launch_fwd16(const float *a3, const float *b3, const float *H, const float *Wz, const float *bz, const float *Wr,
                 const float *br, const float *Wh, const float *bh, float *CZ, float *CR, float *CH, float *Z, float *R,
                 float *Ht, float *Hn, int64_t N, float lo, float hi, hipStream_t stream)
{
    using S = CellShape16<C, WAVES>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cell_fused_fwd16_kernel<C, WAVES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_cell_fused_fwd: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int64_t tiles = (N + 15) / 16;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_fwd: too many rows");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (S::kLds + 512), 32 / WAVES));
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + WAVES - 1) / WAVES, 256 * per_cu);
    hipLaunchKernelGGL((cell_fused_fwd16_kernel<C, WAVES>), dim3(blocks), dim3(S::kThreads), S::kLds, stream, a3, b3, H, Wz,
                       bz, Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, (int)tiles);
    return check_launch("stg_tgcn_cell_fused_fwd");
}

// ---------------------------------------------------------------------------------------------- backward
// The backward row-local chain of one TGCN step in one launch (unfused: cell_update_bwd, cell_gates_bwd,
// cell_prep_bwd of tgcn_cell.hip around three rocBLAS input-gradient GEMMs, whose [N,2C] results dCH, dCZ, dCR
// make a 154 MB round trip through HBM per step):
//   dhl = (dHn (1-Z)) (1-Ht^2)     dzl = (dHn (H-Ht)) (Z (1-Z))     dH = dHn Z
//   dCH = dhl Wh  -> d(hh) = dCH[:, :C],  dHR = dCH[:, C:]
//   drl = (dHR H) (R (1-R))        dH += dHR R
//   dCZ = dzl Wz,  dCR = drl Wr    -> d(hz), d(hr) = [:, :C];   dH += dCZ[:, C:] + dCR[:, C:]
//   da3[:, g] = lo <= a3[:, g] + b3[g] <= hi ? d(h_g) : 0                              (clamp backward)
// Same machinery as the forward kernel: torch Linear weights [C][2C] are already [k][n] for these products and
// sit in LDS as [k][2C + 1]; dhl / dzl / drl are formed in MFMA A layout from row pieces loaded straight from HBM
// (and stored from there: the weight gradients read them later); each 32x32 block of a product goes through
// the per-wave LDS tile back to row pieces, where the mask, the dH sums and all stores happen.
template <int C, int WAVES>
struct CellBwdShape {
    static constexpr int K2 = 2 * C, KH = C / 8, NT = K2 / 32, LDB = K2 + 1, TLD = 33;
    static constexpr int kThreads = WAVES * kWave;
    static constexpr int kWeights = 3 * C * LDB;              // floats
    static constexpr int kBias = 3 * C;                       // b3
    static constexpr int kTile = 32 * TLD;
    static constexpr size_t kLds = sizeof(float) * (size_t)(kWeights + kBias + WAVES * kTile);
};

template <int C, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void cell_fused_bwd_kernel(
    const float *__restrict__ dHn, const float *__restrict__ Z, const float *__restrict__ H,
    const float *__restrict__ Ht, const float *__restrict__ R, const float *__restrict__ a3,
    const float *__restrict__ b3, const float *__restrict__ Wz, const float *__restrict__ Wr,
    const float *__restrict__ Wh, float *__restrict__ dhl_o, float *__restrict__ dzl_o, float *__restrict__ drl_o,
    float *__restrict__ da3, float *__restrict__ dH_o, int64_t N, float lo, float hi, int num_tiles)
{
    using S = CellBwdShape<C, WAVES>;
    constexpr int KH = S::KH, LDB = S::LDB, TLD = S::TLD, NTHR = S::kThreads, HB = C / 32;
    extern __shared__ float lds[];
    float *Ws = lds;                                   // Wz | Wr | Wh, each [C][LDB]
    float *bs = lds + S::kWeights;                     // b3
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    float *T = bs + S::kBias + wave * S::kTile;

    {
        const float *src[3] = {Wz, Wr, Wh};
        constexpr int total4 = C * 2 * C / 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float *dst = Ws + g * C * LDB;
            for (int base = 0; base < total4; base += 8 * NTHR) {
                float4 w4[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i4 = base + s * NTHR + threadIdx.x;
                    w4[s] = i4 < total4 ? *reinterpret_cast<const float4 *>(src[g] + (int64_t)i4 * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i = (base + s * NTHR + threadIdx.x) * 4;
                    if (i < C * 2 * C) {
                        const int k = i / (2 * C), n = i - k * 2 * C;       // W[k][n .. n+3]
                        dst[k * LDB + n + 0] = w4[s].x;
                        dst[k * LDB + n + 1] = w4[s].y;
                        dst[k * LDB + n + 2] = w4[s].z;
                        dst[k * LDB + n + 3] = w4[s].w;
                    }
                }
            }
        }
        for (int i = threadIdx.x; i < 3 * C; i += NTHR) bs[i] = b3[i];
    }
    __syncthreads();

    const int total = gridDim.x * WAVES;
    // tiles are dealt wave-major (wave w of workgroup b: tile w * grid + b), so every CU gets its share of a grid
    // smaller than WAVES * CUs (cfg4: 1563 tiles = 6.1 per CU on 256 CUs instead of 8 on 196)
    for (int tile = wave * (int)gridDim.x + (int)blockIdx.x; tile < num_tiles; tile += total) {
        const int64_t row = (int64_t)tile * 32 + l31;
        const bool rok = row < N;
        auto ldrow = [&](const float *p, int j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) v = *reinterpret_cast<const float4 *>(p + row * C + 8 * j + 4 * kh);
            return v;
        };
        // 32x32 block of an accumulator -> 4 row pieces (columns 32 blk + 8 jj + 4 kh .. + 3) of this lane's row
        auto to_rows = [&](const f32x16 &acc, float4 (&dst)[4]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) T[acc_row(i, kh) * TLD + l31] = acc[i];
            wave_lds_sync();
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float *t = T + l31 * TLD + 8 * jj + 4 * kh;
                dst[jj] = make_float4(t[0], t[1], t[2], t[3]);
            }
            wave_lds_sync();
        };
        // da3[:, g*C + cols of block blk] = clamp mask * piece
        auto store_da3 = [&](int g, int blk, const float4 (&piece)[4]) {
            if (!rok) return;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c = g * C + 32 * blk + 8 * jj + 4 * kh;
                const float4 a = *reinterpret_cast<const float4 *>(a3 + row * 3 * C + c);
                const float4 b = *reinterpret_cast<const float4 *>(bs + c);
                const float v0 = a.x + b.x, v1 = a.y + b.y, v2 = a.z + b.z, v3 = a.w + b.w;
                float4 o;
                o.x = (v0 >= lo && v0 <= hi) ? piece[jj].x : 0.f;
                o.y = (v1 >= lo && v1 <= hi) ? piece[jj].y : 0.f;
                o.z = (v2 >= lo && v2 <= hi) ? piece[jj].z : 0.f;
                o.w = (v3 >= lo && v3 <= hi) ? piece[jj].w : 0.f;
                *reinterpret_cast<float4 *>(da3 + row * 3 * C + c) = o;
            }
        };
        // acc[b] = A (row pieces, K = C) x W_g[:, half * C + 32 b ..]  (W_g is [C][2C] in LDS; one half of the 2C
        // output columns at a time: C / 32 accumulators live instead of 2C / 32)
        auto gemm = [&](const float4 (&A)[KH], int g, int half, f32x16 (&acc)[HB]) {
            const float *pw = Ws + g * C * LDB + (4 * kh) * LDB + half * C + l31;
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const float av[4] = {A[j].x, A[j].y, A[j].z, A[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int b = 0; b < HB; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], pw[(8 * j + i) * LDB + b * 32], acc[b], 0, 0, 0);
                }
            }
        };

        // ---- update backward (row pieces) --------------------------------------------------------------------
        // (register budget: 256 per wave at two waves per SIMD.  dzl is stored here and re-read by the same lane before
        // its GEMM, H is re-read where dHR needs it -- both hit in L2 -- instead of living through the first GEMM)
        float4 dhl[KH], dHa[KH];
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const float4 g = ldrow(dHn, j), z = ldrow(Z, j), t = ldrow(Ht, j);
            const float4 h = ldrow(H, j);
            float4 dzl[1];
            dhl[j] = make_float4((g.x * (1.0f - z.x)) * (1.0f - t.x * t.x), (g.y * (1.0f - z.y)) * (1.0f - t.y * t.y),
                                 (g.z * (1.0f - z.z)) * (1.0f - t.z * t.z), (g.w * (1.0f - z.w)) * (1.0f - t.w * t.w));
            dzl[0] = make_float4((g.x * (h.x - t.x)) * (z.x * (1.0f - z.x)), (g.y * (h.y - t.y)) * (z.y * (1.0f - z.y)),
                                 (g.z * (h.z - t.z)) * (z.z * (1.0f - z.z)), (g.w * (h.w - t.w)) * (z.w * (1.0f - z.w)));
            dHa[j] = make_float4(g.x * z.x, g.y * z.y, g.z * z.z, g.w * z.w);
            if (rok) {
                *reinterpret_cast<float4 *>(dhl_o + row * C + 8 * j + 4 * kh) = dhl[j];
                *reinterpret_cast<float4 *>(dzl_o + row * C + 8 * j + 4 * kh) = dzl[0];
            }
        }

        f32x16 acc[HB];
        float4 piece[4];
        // ---- dCH = dhl Wh: d(hh) -> da3[:, 2C..];  dHR -> drl, dH -------------------------------------------------
        gemm(dhl, 2, 0, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
            store_da3(2, blk, piece);
        }
        gemm(dhl, 2, 1, acc);
        float4 drl[KH];
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);                                            // dHR, columns 32 blk ..
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * blk + jj;
                const float4 r = ldrow(R, j), h = ldrow(H, j), d = piece[jj];
                drl[j] = make_float4((d.x * h.x) * (r.x * (1.0f - r.x)), (d.y * h.y) * (r.y * (1.0f - r.y)),
                                     (d.z * h.z) * (r.z * (1.0f - r.z)), (d.w * h.w) * (r.w * (1.0f - r.w)));
                dHa[j] = make_float4(dHa[j].x + d.x * r.x, dHa[j].y + d.y * r.y, dHa[j].z + d.z * r.z, dHa[j].w + d.w * r.w);
                if (rok) *reinterpret_cast<float4 *>(drl_o + row * C + 8 * j + 4 * kh) = drl[j];
            }
        }
        // ---- dCZ = dzl Wz,  dCR = drl Wr: d(hz), d(hr) -> da3;  second halves -> dH, dCZ's first, then dCR's --------
        // (the order cell_prep_bwd_kernel adds them in: (dH + x) + y)
        {
            float4 dzl[KH];
#pragma unroll
            for (int j = 0; j < KH; ++j) dzl[j] = ldrow(dzl_o, j);              // this lane's own stores, above
            gemm(dzl, 0, 0, acc);
#pragma unroll
            for (int blk = 0; blk < HB; ++blk) {
                to_rows(acc[blk], piece);
                store_da3(0, blk, piece);
            }
                gemm(dzl, 0, 1, acc);
#pragma unroll
            for (int blk = 0; blk < HB; ++blk) {
                to_rows(acc[blk], piece);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float4 &d = dHa[4 * blk + jj];
                    d = make_float4(d.x + piece[jj].x, d.y + piece[jj].y, d.z + piece[jj].z, d.w + piece[jj].w);
                }
            }
        }
        gemm(drl, 1, 0, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
            store_da3(1, blk, piece);
        }
        gemm(drl, 1, 1, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * blk + jj;
                const float4 y = piece[jj], d = dHa[j];
                if (rok)
                    *reinterpret_cast<float4 *>(dH_o + row * C + 8 * j + 4 * kh) =
                        make_float4(d.x + y.x, d.y + y.y, d.z + y.z, d.w + y.w);
            }
        }
    }
}

template <int C, int WAVES>
int launch_bwd(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R, const float *a3,
               const float *b3, const float *Wz, const float *Wr, const float *Wh, float *dhl, float *dzl, float *drl,
               float *da3, float *dH, int64_t N, float lo, float hi, hipStream_t stream)
{
    using S = CellBwdShape<C, WAVES>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cell_fused_bwd_kernel<C, WAVES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_cell_fused_bwd: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int64_t tiles = (N + 31) / 32;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_bwd: too many rows");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (S::kLds + 512), 32 / WAVES));
    const unsigned blocks = (unsigned)std::min<int64_t>(tiles, 256 * per_cu);
    hipLaunchKernelGGL((cell_fused_bwd_kernel<C, WAVES>), dim3(blocks), dim3(S::kThreads), S::kLds, stream, dHn, Z, H, Ht, R,
                       a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, (int)tiles);
    return check_launch("stg_tgcn_cell_fused_bwd");
}

// ---------------------------------------------------------------------------------------------- backward, 16-row tiles
// (see cell_fused_fwd16_kernel: v_mfma_f32_16x16x4_f32, lane = (n16, kq), pieces at column 16 j + 4 kq, one 16x16
// block of a product = one row piece per lane after the LDS transpose; LDB = 2C + 4 for conflict-free B reads)
// FIN > 0: the kernel also forms dx = da3 Wcat^T [N,FIN] (the gradient reaching the aggregated input: Wcat [FIN][3C]
// is the three GCN gate weights side by side) -- da3's row pieces are already MFMA A operands when they are stored,
// so the product costs 25 % more MFMAs here instead of a rocBLAS launch that re-reads da3.
template <int C, int WAVES, int FIN = 0>
struct CellBwdShape16 {
    static constexpr int K2 = 2 * C, KQ = C / 16, HB = C / 16, LDB = K2 + 4, TLD = 17, LDZ = FIN + 4, XT = FIN / 16;
    static constexpr int kThreads = WAVES * kWave;
    static constexpr int kWeights = 3 * C * LDB;              // floats
    static constexpr int kBias = 3 * C;                       // b3
    static constexpr int kTile = 16 * TLD;
    static constexpr int kDx = FIN > 0 ? 3 * C * LDZ : 0;     // Wcat^T [3C][LDZ]
    static constexpr size_t kLds = sizeof(float) * (size_t)(kWeights + kBias + WAVES * kTile + kDx);
};

template <int C, int WAVES, int FIN = 0>
__global__ __launch_bounds__(WAVES * kWave) void cell_fused_bwd16_kernel(
    const float *__restrict__ dHn, const float *__restrict__ Z, const float *__restrict__ H,
    const float *__restrict__ Ht, const float *__restrict__ R, const float *__restrict__ a3,
    const float *__restrict__ b3, const float *__restrict__ Wz, const float *__restrict__ Wr,
    const float *__restrict__ Wh, float *__restrict__ dhl_o, float *__restrict__ dzl_o, float *__restrict__ drl_o,
    float *__restrict__ da3, float *__restrict__ dH_o, int64_t N, float lo, float hi, int num_tiles,
    const float *__restrict__ Wcat, float *__restrict__ dx_o)
{
    using S = CellBwdShape16<C, WAVES, FIN>;
    constexpr int KQ = S::KQ, HB = S::HB, LDB = S::LDB, TLD = S::TLD, NTHR = S::kThreads, LDZ = S::LDZ, XT = S::XT;
    extern __shared__ float lds[];
    float *Ws = lds;                                   // Wz | Wr | Wh, each [C][LDB]
    float *bs = lds + S::kWeights;                     // b3
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    float *T = bs + S::kBias + wave * S::kTile;
    float *Xs = bs + S::kBias + WAVES * S::kTile;      // FIN > 0: Wcat^T [3C][LDZ]
    if constexpr (FIN > 0) {
        for (int i = threadIdx.x; i < FIN * 3 * C; i += NTHR) {
            const int k = i / (3 * C), c = i - k * 3 * C;           // Wcat[k][c]
            Xs[c * LDZ + k] = Wcat[i];
        }
    }

    {
        const float *src[3] = {Wz, Wr, Wh};
        constexpr int total4 = C * 2 * C / 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float *dst = Ws + g * C * LDB;
            for (int i4 = threadIdx.x; i4 < total4; i4 += NTHR) {
                const float4 w4 = *reinterpret_cast<const float4 *>(src[g] + (int64_t)i4 * 4);
                const int i = i4 * 4, k = i / (2 * C), n = i - k * 2 * C;       // W[k][n .. n+3]
                *reinterpret_cast<float4 *>(dst + k * LDB + n) = w4;            // LDB % 4 == 0: aligned
            }
        }
        for (int i = threadIdx.x; i < 3 * C; i += NTHR) bs[i] = b3[i];
    }
    __syncthreads();

    const int total = gridDim.x * WAVES;
    for (int tile = wave * (int)gridDim.x + (int)blockIdx.x; tile < num_tiles; tile += total) {
        const int64_t row = (int64_t)tile * 16 + n16;
        const bool rok = row < N;
        auto ldrow = [&](const float *p, int j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) v = *reinterpret_cast<const float4 *>(p + row * C + 16 * j + 4 * kq);
            return v;
        };
        // 16x16 block of an accumulator -> the row piece (columns 16 blk + 4 kq .. + 3) of this lane's row
        auto to_rows = [&](const f32x4 &acc, float4 &dst) {
#pragma unroll
            for (int i = 0; i < 4; ++i) T[(4 * kq + i) * TLD + n16] = acc[i];
            wave_lds_sync();
            const float *t = T + n16 * TLD + 4 * kq;
            dst = make_float4(t[0], t[1], t[2], t[3]);
            wave_lds_sync();
        };
        f32x4 accx[XT > 0 ? XT : 1];
        if constexpr (FIN > 0) {
#pragma unroll
            for (int ct = 0; ct < XT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) accx[ct][i] = 0.f;
        }
        auto store_da3 = [&](int g, int blk, const float4 &piece) {
            const int c = g * C + 16 * blk + 4 * kq;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) {
                const float4 a = *reinterpret_cast<const float4 *>(a3 + row * 3 * C + c);
                const float4 b = *reinterpret_cast<const float4 *>(bs + c);
                const float v0 = a.x + b.x, v1 = a.y + b.y, v2 = a.z + b.z, v3 = a.w + b.w;
                o.x = (v0 >= lo && v0 <= hi) ? piece.x : 0.f;
                o.y = (v1 >= lo && v1 <= hi) ? piece.y : 0.f;
                o.z = (v2 >= lo && v2 <= hi) ? piece.z : 0.f;
                o.w = (v3 >= lo && v3 <= hi) ? piece.w : 0.f;
                *reinterpret_cast<float4 *>(da3 + row * 3 * C + c) = o;
            }
            if constexpr (FIN > 0) {                                 // dx += da3 piece x Wcat^T rows c .. c + 3
                const float ov[4] = {o.x, o.y, o.z, o.w};
                const float *px = Xs + c * LDZ + n16;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ct = 0; ct < XT; ++ct)
                        accx[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ov[i], px[i * LDZ + ct * 16], accx[ct], 0, 0, 0);
            }
        };
        // acc[b] = A (row pieces, K = C) x W_g[:, half * C + 16 b ..]
        auto gemm = [&](const float4 (&A)[KQ], int g, int half, f32x4 (&acc)[HB]) {
            const float *pw = Ws + g * C * LDB + (4 * kq) * LDB + half * C + n16;
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[b][i] = 0.f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float av[4] = {A[j].x, A[j].y, A[j].z, A[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int b = 0; b < HB; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], pw[(16 * j + i) * LDB + b * 16], acc[b], 0, 0, 0);
                }
            }
        };

        // ---- update backward (row pieces)
        float4 dhl[KQ], dHa[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const float4 g = ldrow(dHn, j), z = ldrow(Z, j), t = ldrow(Ht, j);
            const float4 h = ldrow(H, j);
            dhl[j] = make_float4((g.x * (1.0f - z.x)) * (1.0f - t.x * t.x), (g.y * (1.0f - z.y)) * (1.0f - t.y * t.y),
                                 (g.z * (1.0f - z.z)) * (1.0f - t.z * t.z), (g.w * (1.0f - z.w)) * (1.0f - t.w * t.w));
            const float4 dz = make_float4((g.x * (h.x - t.x)) * (z.x * (1.0f - z.x)), (g.y * (h.y - t.y)) * (z.y * (1.0f - z.y)),
                                          (g.z * (h.z - t.z)) * (z.z * (1.0f - z.z)), (g.w * (h.w - t.w)) * (z.w * (1.0f - z.w)));
            dHa[j] = make_float4(g.x * z.x, g.y * z.y, g.z * z.z, g.w * z.w);
            if (rok) {
                *reinterpret_cast<float4 *>(dhl_o + row * C + 16 * j + 4 * kq) = dhl[j];
                *reinterpret_cast<float4 *>(dzl_o + row * C + 16 * j + 4 * kq) = dz;
            }
        }

        f32x4 acc[HB];
        float4 piece;
        // ---- dCH = dhl Wh: d(hh) -> da3[:, 2C..];  dHR -> drl, dH
        gemm(dhl, 2, 0, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
            store_da3(2, blk, piece);
        }
        gemm(dhl, 2, 1, acc);
        float4 drl[KQ];
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);                                            // dHR, columns 16 blk ..
            const float4 r = ldrow(R, blk), h = ldrow(H, blk), d = piece;
            drl[blk] = make_float4((d.x * h.x) * (r.x * (1.0f - r.x)), (d.y * h.y) * (r.y * (1.0f - r.y)),
                                   (d.z * h.z) * (r.z * (1.0f - r.z)), (d.w * h.w) * (r.w * (1.0f - r.w)));
            dHa[blk] = make_float4(dHa[blk].x + d.x * r.x, dHa[blk].y + d.y * r.y, dHa[blk].z + d.z * r.z,
                                   dHa[blk].w + d.w * r.w);
            if (rok) *reinterpret_cast<float4 *>(drl_o + row * C + 16 * blk + 4 * kq) = drl[blk];
        }
        // ---- dCZ = dzl Wz,  dCR = drl Wr: d(hz), d(hr) -> da3;  second halves -> dH, dCZ's first, then dCR's
        {
            float4 dzl[KQ];
#pragma unroll
            for (int j = 0; j < KQ; ++j) dzl[j] = ldrow(dzl_o, j);              // this lane's own stores, above
            gemm(dzl, 0, 0, acc);
#pragma unroll
            for (int blk = 0; blk < HB; ++blk) {
                to_rows(acc[blk], piece);
                store_da3(0, blk, piece);
            }
            gemm(dzl, 0, 1, acc);
#pragma unroll
            for (int blk = 0; blk < HB; ++blk) {
                to_rows(acc[blk], piece);
                float4 &d = dHa[blk];
                d = make_float4(d.x + piece.x, d.y + piece.y, d.z + piece.z, d.w + piece.w);
            }
        }
        gemm(drl, 1, 0, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
            store_da3(1, blk, piece);
        }
        gemm(drl, 1, 1, acc);
#pragma unroll
        for (int blk = 0; blk < HB; ++blk) {
            to_rows(acc[blk], piece);
            const float4 y = piece, d = dHa[blk];
            if (rok)
                *reinterpret_cast<float4 *>(dH_o + row * C + 16 * blk + 4 * kq) =
                    make_float4(d.x + y.x, d.y + y.y, d.z + y.z, d.w + y.w);
        }
        if constexpr (FIN > 0) {
#pragma unroll
            for (int ct = 0; ct < XT; ++ct) {
                to_rows(accx[ct], piece);
                if (rok) *reinterpret_cast<float4 *>(dx_o + row * FIN + 16 * ct + 4 * kq) = piece;
            }
        }
    }
}

template <int C, int WAVES, int FIN = 0>
int launch_bwd16(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R, const float *a3,
                 const float *b3, const float *Wz, const float *Wr, const float *Wh, float *dhl, float *dzl, float *drl,
                 float *da3, float *dH, int64_t N, float lo, float hi, hipStream_t stream, const float *Wcat = nullptr,
                 float *dx = nullptr)
{
    using S = CellBwdShape16<C, WAVES, FIN>;
    static PerDeviceOnce once;
    bool *raised = once.slot();
    if (S::kLds > 64 * 1024 && !*raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cell_fused_bwd16_kernel<C, WAVES, FIN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLds);
        if (e != hipSuccess) return fail((int)e, "stg_tgcn_cell_fused_bwd: %s", hipGetErrorString(e));
        *raised = true;
    }
    const int64_t tiles = (N + 15) / 16;
    if (tiles > INT32_MAX) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_bwd: too many rows");
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / (S::kLds + 512), 32 / WAVES));
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + WAVES - 1) / WAVES, 256 * per_cu);
    hipLaunchKernelGGL((cell_fused_bwd16_kernel<C, WAVES, FIN>), dim3(blocks), dim3(S::kThreads), S::kLds, stream, dHn, Z, H,
                       Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, (int)tiles, Wcat, dx);
    return check_launch("stg_tgcn_cell_fused_bwd");
}

}  // namespace
}  // namespace stg

extern "C" int stg_tgcn_cell_fused_supported(int32_t C) { return C == 32 || C == 64; }

extern "C" int stg_tgcn_cell_fused_fwd(const float *a3, const float *b3, const float *H, const float *Wz, const float *bz,
                                       const float *Wr, const float *br, const float *Wh, const float *bh, float *CZ,
                                       float *CR, float *CH, float *Z, float *R, float *Ht, float *Hn, int64_t N, int32_t C,
                                       float lo, float hi, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_fwd: negative N");
    if (!stg_tgcn_cell_fused_supported(C)) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_fwd: C must be 32 or 64 (got %d)", C);
    if (N == 0) return 0;
    if (!a3 || !b3 || !H || !Wz || !bz || !Wr || !br || !Wh || !bh || !CZ || !CR || !CH || !Z || !R || !Ht || !Hn)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_fwd: NULL pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // measured (tools/microbench_cell.py, C = 64, 12-wave workgroups): N = 50 K 83 -> 75 us, 400 K 421 -> 384 us, 25 K
    // 61 -> 62 us (with 16 waves: 78 / 436 / 73): 16-row tiles from 40 K rows on
    if (tuning().cell_rows == 16 || (tuning().cell_rows == 0 && N >= 40000)) {
        if (C == 64) return launch_fwd16<64, 12>(a3, b3, H, Wz, bz, Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, st);
        return launch_fwd16<32, 16>(a3, b3, H, Wz, bz, Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, st);
    }
    if (C == 64) return launch_fwd<64, 8>(a3, b3, H, Wz, bz, Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, st);
    return launch_fwd<32, 4>(a3, b3, H, Wz, bz, Wr, br, Wh, bh, CZ, CR, CH, Z, R, Ht, Hn, N, lo, hi, st);
}

extern "C" int stg_tgcn_cell_fused_bwd(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R,
                                       const float *a3, const float *b3, const float *Wz, const float *Wr, const float *Wh,
                                       float *dhl, float *dzl, float *drl, float *da3, float *dH, int64_t N, int32_t C,
                                       float lo, float hi, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_bwd: negative N");
    if (!stg_tgcn_cell_fused_supported(C)) return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_bwd: C must be 32 or 64 (got %d)", C);
    if (N == 0) return 0;
    if (!dHn || !Z || !H || !Ht || !R || !a3 || !b3 || !Wz || !Wr || !Wh || !dhl || !dzl || !drl || !da3 || !dH)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_bwd: NULL pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // measured (tools/microbench_cell.py): C = 64 with 12-wave workgroups (168 registers, no spills; 16 waves spill 59
    // and lose): N = 25 K 43 -> 39 us, 50 K 82 -> 67 us, 400 K 570 -> 404 us; C = 32 (16 waves): 50 K 34.8 -> 28.4 us
    if (tuning().cell_rows == 16 || (tuning().cell_rows == 0 && N >= 20000)) {
        if (C == 64) return launch_bwd16<64, 12>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st);
        return launch_bwd16<32, 16>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st);
    }
    if (C == 64) return launch_bwd<64, 8>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st);
    return launch_bwd<32, 4>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st);
}

extern "C" int stg_tgcn_cell_fused_bwd_dx_supported(int32_t C, int32_t Fin) { return (C == 32 || C == 64) && Fin == 32; }

extern "C" int stg_tgcn_cell_fused_bwd_dx(const float *dHn, const float *Z, const float *H, const float *Ht, const float *R,
                                          const float *a3, const float *b3, const float *Wz, const float *Wr,
                                          const float *Wh, const float *Wcat, float *dhl, float *dzl, float *drl,
                                          float *da3, float *dH, float *dx, int64_t N, int32_t C, int32_t Fin, float lo,
                                          float hi, void *stream)
{
    using namespace stg;
    if (N < 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_bwd_dx: negative N");
    if (!stg_tgcn_cell_fused_bwd_dx_supported(C, Fin))
        return fail(STG_ERR_UNSUPPORTED, "stg_tgcn_cell_fused_bwd_dx: C = %d, Fin = %d not supported", C, Fin);
    if (N == 0) return 0;
    if (!dHn || !Z || !H || !Ht || !R || !a3 || !b3 || !Wz || !Wr || !Wh || !Wcat || !dhl || !dzl || !drl || !da3 || !dH || !dx)
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_tgcn_cell_fused_bwd_dx: NULL pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C == 64)
        return launch_bwd16<64, 8, 32>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st, Wcat, dx);
    return launch_bwd16<32, 16, 32>(dHn, Z, H, Ht, R, a3, b3, Wz, Wr, Wh, dhl, dzl, drl, da3, dH, N, lo, hi, st, Wcat, dx);
}
