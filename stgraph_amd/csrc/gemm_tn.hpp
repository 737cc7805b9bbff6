// Operand description shared by the tall-skinny contraction kernels (gemm_tn.hip: fp32 matrix instruction; gemm_tn_x3.hip: 3-term
// bf16 split).
#pragma once
#include "stg_common.hpp"

namespace stg {

constexpr int kGemmMaxSeg = 32;
struct GemmSegs {
    const float *a[kGemmMaxSeg];
    const float *b[kGemmMaxSeg];
    const float *b2[kGemmMaxSeg];        // columns [nsplit, N) of B_t live in a second matrix (GemmForm::nsplit < N)
    const float *am[kGemmMaxSeg];        // GemmForm::a_mask: A_t[k][m] counts only where am_t[k][m] > 0 (same shape and stride)
};

// Operand forms.  A_t is [K, M] with row stride lda.  B_t is [K, N] given as ONE or TWO row-major matrices side by
// side: columns [0, nsplit) from b (row stride ldb), columns [nsplit, N) from b2 (row stride ldb2); nsplit is a multiple
// of 32 or N.  b_op transforms the values of b as they are loaded (b2 is taken as is): the weight gradients of the
// one-launch TGCN step contract dzl with [clamp(x3[:, gate]) | H] and dyt with relu(Hn) without those operands ever
// being written out.
struct GemmForm {
    int lda, ldb, ldb2, nsplit;
    int b_op;                            // STG_GEMM_B_NONE / _CLAMP / _RELU
    float lo, hi;
    int a_mask;                          // 1: A is taken as A * [am > 0] -- the backward of a ReLU applied while loading:
                                         // dW = (g * [out > 0])^T X and its column sums (the bias gradient) in one launch,
                                         // the masked gradient never written
    const int *gate;                     // gate_when 1: the launch (slabs and reduction) does nothing unless *gate == 0; 2: unless
    int gate_when;                       // *gate != 0.  Two products gated on the same word, one of each kind, write the same C:
                                         // which one is decided on the device (stg_gemm_tn_gated_f32)
};

__device__ __forceinline__ bool gemm_gated_off(const int *gate, int when)
{
    if (when == 0 || gate == nullptr) return false;
    const bool zero = __builtin_amdgcn_readfirstlane(*gate) == 0;
    return when == 1 ? !zero : zero;
}


// gemm_tn_x3.hip: C = sum_t A_t^T [op(b_t) | b2_t] as 3-term bf16 splits (slabs in `slab`, one per workgroup: [M x N] (+ M column
// sums)).  Returns the number of slabs written through *slabs, 0 if the shape is not covered (the caller takes the fp32 forms).
bool gemm_tn_x3_covers(int M, int N, const GemmForm &form, int64_t K, int T);
int gemm_tn_x3_slabs(int M, int N, int64_t K, int T);
int gemm_tn_x3_launch(const GemmSegs &segs, const GemmForm &form, float *slab, int64_t K, int M, int N, int T, bool colsum, int *slabs,
                      hipStream_t stream);

}  // namespace stg
