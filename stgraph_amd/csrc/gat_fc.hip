// GATConv's input side as ONE launch: feat = x W^T (the layer's bias-free `fc`) and, in its epilogue, the attention
// projections el[n,h] = sum_d feat[n,h,d] attn_l[h,d], er likewise.
//
// Reference: nn/pytorch/static/gat_conv.py:43-48 (`self.fc(h_src).view(-1, H, D)`, `(feat_src * self.attn_l).sum(-1)`,
// `(feat_dst * self.attn_r).sum(-1)`): three torch passes there; here rowgemm_wide (0.28 ms at |V| = 256 K, 64 -> 8 x 64)
// + stg_gat_proj_fwd (0.13 ms, re-reads all of feat) become one kernel that writes feat once.
//
// Layout: the "row pieces" scheme of tgcn_step.hpp -- v_mfma_f32_16x16x4_f32 with the WEIGHT as the A operand, so a
// lane's four accumulator values are four consecutive output columns of its own row: 16-byte stores, and the head's
// dot product with attn_l / attn_r is 16 multiply-adds per lane and two cross-lane adds (the four lanes that share
// a row).  W [H*D, FIN] stays in LDS in its torch Linear layout (rows padded by 8 floats: ~ 148 KB at 512 x 64), one
// workgroup of 16 waves per CU, 16-row tiles dealt wave-major; a tile is 128 output columns (two heads) at a time:
// 8 accumulators, 128 MFMAs, then the epilogue.  MFMA-bound: 512 MFMAs x 32 cycles per tile and SIMD wave.
// Results agree with rocBLAS / the unfused pair to fp32 rounding (k order (j, i, kq)); tests: 1e-5 relative.
//
// The same kernel without the projections (PROJ = false) is the layer's OUTPUT side in its uniform-attention form
// (stg_gat_fc_out, ABI 23): out = xm W^T with xm the in-neighbour mean of x (stg_gat_fwd_k1_uniform in gat.hip), and the
// layer's elu written beside it from the same accumulators.
#include "tgcn_step.hpp"

namespace stg {
namespace {

constexpr int kFcWaves = 16;
constexpr int kFcCT = 8;             // column tiles (of 16) per pass = 128 columns = two heads of 64

// PROJ: the attention projections in the epilogue (the layer's input side).  !PROJ: the plain product, optionally with
// elu(product) written beside it (act_out; the layer's OUTPUT side in its uniform-attention form, see stg_gat_fc_out).
template <int FIN, bool PROJ>
__global__ __launch_bounds__(kFcWaves * 64) void gat_fc_kernel(
    const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ attn_l,
    const float *__restrict__ attn_r, float *__restrict__ feat, float *__restrict__ el, float *__restrict__ er,
    int N, int H, float *__restrict__ act_out, const int *__restrict__ only_if)
{
    constexpr int NT = kFcWaves * 64, LD = FIN + 8, J = FIN / 16, D = 64;     // 8 mod 16 dwords: conflict-free ds_read_b128 (tgcn_step.hpp)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // only_if: the launch that materialises feat for the general units when some score is not finite (stg_gat_fc_feat_if): the
    // uniform-attention form never reads feat, so the input side leaves it unwritten (feat == nullptr below) -- kernel-uniform
    if (only_if && __builtin_amdgcn_readfirstlane(*only_if) == 0) return;
    const int HD = H * D;
    float *Wl = lds, *al = Wl + HD * LD, *ar = al + HD;
    stage_rows<NT>(Wl, LD, W, HD, FIN);
    if constexpr (PROJ) {
        for (int i = threadIdx.x; i < HD; i += NT) {
            al[i] = attn_l[i];
            ar[i] = attn_r[i];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const float *wrow = Wl + n16 * LD + 4 * kq;
    const int ntiles = (N + 15) >> 4;
    for (int tile = (int)blockIdx.x * kFcWaves + wave; tile < ntiles; tile += (int)gridDim.x * kFcWaves) {
        const int row = tile * 16 + n16;
        const bool ok = row < N;
        const float *xr = x + (int64_t)min(row, N - 1) * FIN + 4 * kq;
        float4 xin[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xin[j] = *reinterpret_cast<const float4 *>(xr + 16 * j);
        float *frow = feat + (int64_t)row * HD + 4 * kq;
        for (int g = 0; g < HD / 128; ++g) {
            f32x4 acc[kFcCT];
#pragma unroll
            for (int ct = 0; ct < kFcCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_pieces<kFcCT, J, false>(acc, wrow + g * 128 * LD, LD, [&](int j) { return xin[j]; });
            float pl[2] = {0.f, 0.f}, pr[2] = {0.f, 0.f};
#pragma unroll
            for (int ct = 0; ct < kFcCT; ++ct) {
                const int col = g * 128 + ct * 16;
                if (ok && feat) *reinterpret_cast<float4 *>(frow + col) = to_f4(acc[ct]);
                if constexpr (!PROJ) {
                    if (ok && act_out) {                      // kernel-uniform; torch's elu: x <= 0 ? exp(x) - 1 : x
                        float4 y;
                        y.x = acc[ct][0] <= 0.f ? expf(acc[ct][0]) - 1.0f : acc[ct][0];
                        y.y = acc[ct][1] <= 0.f ? expf(acc[ct][1]) - 1.0f : acc[ct][1];
                        y.z = acc[ct][2] <= 0.f ? expf(acc[ct][2]) - 1.0f : acc[ct][2];
                        y.w = acc[ct][3] <= 0.f ? expf(acc[ct][3]) - 1.0f : acc[ct][3];
                        *reinterpret_cast<float4 *>(act_out + (int64_t)row * HD + 4 * kq + col) = y;
                    }
                    continue;
                }
                const float4 a = *reinterpret_cast<const float4 *>(al + col + 4 * kq);
                const float4 b = *reinterpret_cast<const float4 *>(ar + col + 4 * kq);
                pl[ct >> 2] += acc[ct][0] * a.x + acc[ct][1] * a.y + acc[ct][2] * a.z + acc[ct][3] * a.w;
                pr[ct >> 2] += acc[ct][0] * b.x + acc[ct][1] * b.y + acc[ct][2] * b.z + acc[ct][3] * b.w;
            }
            if constexpr (!PROJ) continue;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                float l = pl[h2], r = pr[h2];
                l = l + __shfl_xor(l, 16, 64);
                r = r + __shfl_xor(r, 16, 64);
                l = l + __shfl_xor(l, 32, 64);
                r = r + __shfl_xor(r, 32, 64);
                if (ok && kq == 0) {
                    el[(int64_t)row * H + 2 * g + h2] = l;
                    er[(int64_t)row * H + 2 * g + h2] = r;
                }
            }
        }
    }
}

inline size_t fc_lds_bytes(int fin, int H) { return ((size_t)H * 64 * (fin + 8) + 2 * (size_t)H * 64) * sizeof(float); }

inline bool fc_shape_ok(int fin, int H, int D)
{
    return D == 64 && H >= 2 && H % 2 == 0 && (fin == 32 || fin == 64) && fc_lds_bytes(fin, H) <= 160 * 1024;
}

}  // namespace
}  // namespace stg

extern "C" int stg_gat_fc_supported(int32_t fin, int32_t H, int32_t D) { return stg::fc_shape_ok(fin, H, D) ? 1 : 0; }

namespace stg {
namespace {
int fc_launch(const char *what, bool proj, const float *x, const float *W, const float *attn_l, const float *attn_r,
              float *feat, float *el, float *er, float *act_out, int32_t N, int32_t fin, int32_t H, int32_t D, void *stream,
              const int32_t *only_if = nullptr)
{
    if (N < 0 || !fc_shape_ok(fin, H, D))
        return fail(STG_ERR_UNSUPPORTED, "%s: unsupported shape N=%d fin=%d H=%d D=%d", what, N, fin, H, D);
    if (N == 0) return 0;
    if (!x || !W || (!feat && !proj) || (proj && (!attn_l || !attn_r || !el || !er)))
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", what);
    if ((int64_t)N * H * D > 0x7fffffffll * 4)
        return fail(STG_ERR_UNSUPPORTED, "%s: N * H * D too large", what);
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        // the split form on the bf16 matrix cores (gat_heads_x3.hip): bound by its stores, not by 512 fp32 matrix instructions per tile
        const uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(feat) |
                                reinterpret_cast<uintptr_t>(act_out);
        if (!only_if && align % 16 == 0 && gat_heads_x3_wanted(N, H, D, fin))
            return gat_heads_fc_x3_launch(what, x, W, attn_l, attn_r, feat, act_out, proj ? el : nullptr, proj ? er : nullptr, N, H, stream);
    }
    const size_t lds = fc_lds_bytes(fin, H);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int ntiles = (N + 15) / 16;
    const int grid = std::max(1, std::min(cus, (ntiles + kFcWaves - 1) / kFcWaves));
    auto go = [&](auto kernel, PerDeviceOnce &once) {
        bool *done = once.slot();
        if (!*done) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
            *done = true;
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kFcWaves * 64), lds, st, x, W, attn_l, attn_r, feat, el,
                           er, N, H, act_out, only_if);
        return 0;
    };
    static PerDeviceOnce once32p, once64p, once32, once64;
    const int rc = proj ? (fin == 32 ? go(gat_fc_kernel<32, true>, once32p) : go(gat_fc_kernel<64, true>, once64p))
                        : (fin == 32 ? go(gat_fc_kernel<32, false>, once32) : go(gat_fc_kernel<64, false>, once64));
    if (rc) return rc;
    return check_launch(what);
}
}  // namespace
}  // namespace stg

extern "C" int stg_gat_fc_fwd(const float *x, const float *W, const float *attn_l, const float *attn_r, float *feat,
                              float *el, float *er, int32_t N, int32_t fin, int32_t H, int32_t D, void *stream)
{
    return stg::fc_launch("stg_gat_fc_fwd", true, x, W, attn_l, attn_r, feat, el, er, nullptr, N, fin, H, D, stream);
}

extern "C" int stg_gat_fc_out(const float *xm, const float *W, float *out, float *act_out, int32_t N, int32_t fin,
                              int32_t H, int32_t D, void *stream)
{
    return stg::fc_launch("stg_gat_fc_out", false, xm, W, nullptr, nullptr, out, nullptr, nullptr, act_out, N, fin, H, D,
                          stream);
}

// feat = x W^T only if *only_if != 0 (see the kernel): after stg_gat_fc_fwd with feat == NULL
extern "C" int stg_gat_fc_feat_if(const float *x, const float *W, float *feat, int32_t N, int32_t fin, int32_t H, int32_t D,
                                  const int32_t *only_if, void *stream)
{
    if (!only_if) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gat_fc_feat_if: NULL flag");
    return stg::fc_launch("stg_gat_fc_feat_if", false, x, W, nullptr, nullptr, feat, nullptr, nullptr, nullptr, N, fin, H, D, stream,
                          only_if);
}
