// C[M,N] = A[K,M]^T * B[K,N]  for tall-skinny operands (K = number of vertices, M,N = feature
// widths): the weight-gradient contraction  dW = X^T dY  of every dense layer next to the Seastar
// kernels (GCNConv's X W, TGCN's gate Linears).  rocBLAS/hipBLASLt pick a 32x32..32x64 macro-tile
// without split-K for these shapes, i.e. 4-16 workgroups on a 256-CU chip (measured: 194 us for
// 64x128xK=50K, 1.82 ms for 128x128xK=1M; profiles/r01).  This kernel splits K over the whole chip.
//
// fp32 in, fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf
// chain (no xf32/TF32 on gfx950), so the only difference from a sequential sum is the fixed split
// of K into slices, each summed in order, slices added in order (deterministic, no atomics).
//
// Layout trick: A is [K,M] row-major and the MFMA wants A^T fragments "lane l holds A^T[i=l&31]
// [k=l>>5]" = A[k][m0 + (l&31)]: for a fixed k that is 32 CONSECUTIVE floats, so both operands load
// straight from global memory in fragment layout as two coalesced 128-B segments per instruction;
// no LDS transpose, no bank conflicts.  Addresses are (wave-uniform row base in SGPRs) + (one
// loop-invariant 32-bit lane offset), so the two in-flight operand sets cost no address VGPRs.
// LDS is used only to add the 4 waves of a block (4 adjacent K sub-slices of the same output tile)
// before one slab write.
#include "gemm_tn.hpp"

namespace stg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kGemmKStep = 16;             // K slices are multiples of this (covers KU = 4 and 8)

// NT = 32-column tiles per wave, KU = k-pairs per operand set (KU * (1 + NT) dword loads in flight)
// CS: additionally emit the column sums of A (sum_k A[k][m]) -- the bias gradient that accompanies
// every weight gradient -- from one extra MFMA per k-pair against a constant-one B fragment.
// Operand segments: C = sum_t A_t^T B_t over up to kGemmMaxSeg (A_t, B_t) pairs of K rows each, passed
// BY VALUE in the kernel arguments (no device-side pointer table to build, HIP-graph capturable).
// This is how a BPTT window's weight gradient -- one contribution per timestep -- becomes ONE launch.
// MT = 32-row M tiles per wave: every B dword a wave loads feeds MT MFMAs and every A dword NT of them.  With MT = 1
// a k-pair costs 1 + NT dword loads for NT MFMAs; the kernel then runs at the rate the texture addresser issues
// those loads (16 cycles per wave instruction, 8 waves per CU: 262 us of address cycles against 210 us of MFMA at
// 1M x 128 x 128), and B is fetched once per M tile.  MT = 2: MT + NT loads for MT * NT MFMAs.
template <int NT, int KU, bool CS, int MT, bool AMASK = false>
__global__ __launch_bounds__(kBlock, 2) void gemm_tn_partial_kernel(
    const GemmSegs segs, const GemmForm form, float *__restrict__ slab, int64_t K, int M, int N, int64_t kslice_wave,
    int m_groups, int n_groups, int s_per_seg)
{
    if (gemm_gated_off(form.gate, form.gate_when)) return;
    extern __shared__ float lds[];                       // one wave's accumulators: (MT * NT) x 16 x 64 floats (+ 64 * MT)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mi = blockIdx.x % m_groups;                // group of MT M tiles
    const int t = blockIdx.x / m_groups;
    const int nj = t % n_groups;
    const int s = t / n_groups;                          // slab index = segment * s_per_seg + slice
    const int seg = s / s_per_seg;
    const int sl = s - seg * s_per_seg;
    const float *__restrict__ A = segs.a[seg];
    const float *__restrict__ B = segs.b[seg];
    const float *__restrict__ B2 = segs.b2[seg];
    const float *__restrict__ AM = segs.am[seg];
    constexpr bool a_mask = AMASK;      // (a template parameter: the mask registers must not cost the plain forms their occupancy)
    const int lda = form.lda, nsplit = form.nsplit;

    const int64_t k0 = ((int64_t)sl * kWavesPerBlock + wave) * kslice_wave;     // wave-uniform, inside the segment
    const int64_t k1 = min(K, k0 + kslice_wave);
    const int kh = lane >> 5;
    // Rows >= M / columns >= N of the tile are never written back, so out-of-range lanes read a
    // clamped (valid) address: no per-load m/n predicates.  Only the K tail needs masking.
    int m[MT];
    unsigned la[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        m[i] = (mi * MT + i) * 32 + (lane & 31);
        la[i] = (unsigned)(kh * lda + min(m[i], M - 1));                        // lane offset into an A row pair
    }
    int n[NT], ldj[NT];
    unsigned lb[NT];
    bool nok[NT], second[NT];            // second: the tile's 32 columns come from b2 (wave-uniform)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n0 = (nj * NT + j) * 32;
        n[j] = n0 + (lane & 31);
        nok[j] = n[j] < N;
        second[j] = n0 >= nsplit;
        ldj[j] = second[j] ? form.ldb2 : form.ldb;
        const int col = second[j] ? min(n[j], N - 1) - nsplit : min(n[j], nsplit - 1);
        lb[j] = (unsigned)(kh * ldj[j] + col);
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs_lane[MT];              // CS: this lane's share of sum_k A[k][m] (k of its parity); a VALU add per k-pair
#pragma unroll                      // instead of another MFMA on the pipe that bounds the kernel
    for (int i = 0; i < MT; ++i) cs_lane[i] = 0.f;
    const bool do_cs = CS && nj == 0;                                           // block-uniform

    constexpr int STEP = 2 * KU;
    // Buffer descriptors over THIS wave's K slice: the hardware range check returns 0 for rows
    // >= k1, which is exactly the zero padding the tail needs, and every load is
    // (descriptor in SGPRs) + (loop-invariant 32-bit lane offset) + (scalar row offset).
    const int64_t rows = k1 > k0 ? k1 - k0 : 0;
    // (records: the last row only as far as the columns in use, so that a column window of a wider matrix -- one gate
    // of x3 -- never reaches past the tensor)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + k0 * lda), 0,
                                                       rows > 0 ? (int)(((rows - 1) * lda + M) * (int64_t)sizeof(float)) : 0,
                                                       0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a_mask ? AM + k0 * lda : A + k0 * lda), 0,
                                                       rows > 0 ? (int)(((rows - 1) * lda + M) * (int64_t)sizeof(float)) : 0,
                                                       0x00020000);
    const auto rsB1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(B + k0 * form.ldb), 0,
        rows > 0 ? (int)(((rows - 1) * form.ldb + nsplit) * (int64_t)sizeof(float)) : 0, 0x00020000);
    const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(B2 ? B2 + k0 * form.ldb2 : B), 0,
        rows > 0 && B2 ? (int)(((rows - 1) * form.ldb2 + (N - nsplit)) * (int64_t)sizeof(float)) : 0, 0x00020000);
    const int b_op = form.b_op;
    const float blo = form.lo, bhi = form.hi;
    int voA[MT], voB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) voA[i] = (int)(la[i] * sizeof(float));
#pragma unroll
    for (int j = 0; j < NT; ++j) voB[j] = (int)(lb[j] * sizeof(float));
    auto load_set = [&](float (&a)[KU][MT], float (&b)[KU][NT], float (&am)[a_mask ? KU : 1][a_mask ? MT : 1], int64_t k) {
        const int r0 = (int)(k - k0);                                           // uniform
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int soA = (r0 + 2 * u) * lda * (int)sizeof(float);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                a[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, voA[i], soA, 0));
                if constexpr (a_mask) am[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsM, voA[i], soA, 0));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int soB = (r0 + 2 * u) * ldj[j] * (int)sizeof(float);
                b[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(second[j] ? rsB2 : rsB1, voB[j], soB, 0));
            }
        }
    };
    // the transform of b's values, applied when a set is consumed (its loads have landed by then anyway)
    auto transform = [&](float (&b)[KU][NT]) {
        if (b_op == STG_GEMM_B_NONE) return;                                    // wave-uniform
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (second[j]) continue;
                const float v = b[u][j];
                // clamp as ONE v_med3_f32 (equal to fminf(fmaxf(v, lo), hi) for every input incl. NaN -> lo): the loop is
                // close to issue-bound, and the two-instruction form cost the launch 17 %
                b[u][j] = b_op == STG_GEMM_B_RELU ? (v < 0.f ? 0.f : v) : __builtin_amdgcn_fmed3f(v, blo, bhi);
            }
    };
    auto mask_a = [&](float (&a)[KU][MT], const float (&am)[a_mask ? KU : 1][a_mask ? MT : 1]) {
        if constexpr (a_mask) {
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) a[u][i] = am[u][i] > 0.f ? a[u][i] : 0.f;
        }
    };
    auto mfma_set = [&](const float (&a)[KU][MT], const float (&b)[KU][NT]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
                if constexpr (CS) cs_lane[i] += a[u][i];
            }
        }
    };

    // two operand sets: the loads of the next STEP rows are in flight while the matrix pipe
    // consumes the current set (fp32 MFMA: 64 cycles each, KU*MT*NT of them per set)
    float a0[KU][MT], b0[KU][NT], a1[KU][MT], b1[KU][NT], m0[a_mask ? KU : 1][a_mask ? MT : 1], m1[a_mask ? KU : 1][a_mask ? MT : 1];
    if (k0 < k1) load_set(a0, b0, m0, k0);
    for (int64_t k = k0; k < k1; k += 2 * STEP) {
        const bool more1 = k + STEP < k1;
        if (more1) load_set(a1, b1, m1, k + STEP);
        transform(b0);
        mask_a(a0, m0);
        mfma_set(a0, b0);
        if (more1) {
            if (k + 2 * STEP < k1) load_set(a0, b0, m0, k + 2 * STEP);
            transform(b1);
            mask_a(a1, m1);
            mfma_set(a1, b1);
        }
    }

    // add the block's 4 K sub-slices in wave order (fixed => deterministic): waves 1..3 hand their accumulators to
    // wave 0 through ONE wave-sized LDS buffer, one after the other; wave 0 writes the slab
    constexpr int kAccFloats = MT * NT * 16 * kWave;
    for (int w = 1; w < kWavesPerBlock; ++w) {
        if (wave == w) {
            float *dst = lds + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((i * NT + j) * 16 + r) * kWave] = acc[i][j][r];
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < MT; ++i) lds[kAccFloats + i * kWave + lane] = cs_lane[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float *src = lds + lane;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * NT + j) * 16 + r) * kWave];
            if constexpr (CS) {
#pragma unroll
                for (int i = 0; i < MT; ++i) cs_lane[i] += lds[kAccFloats + i * kWave + lane];
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const int64_t slab_stride = (int64_t)M * N + (CS ? M : 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (CS) {
                const float both = cs_lane[i] + __shfl_xor(cs_lane[i], 32, kWave);   // even + odd k of column m
                if (do_cs && kh == 0 && m[i] < M) slab[(int64_t)s * slab_stride + (int64_t)M * N + m[i]] = both;
            }
            const int row0 = (mi * MT + i) * 32 + 4 * kh;
            float *out = slab + (int64_t)s * slab_stride + (int64_t)row0 * N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!nok[j]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (row0 + dr < M) out[dr * N + n[j]] = acc[i][j][r];
                }
            }
        }
    }
}

// ---- wide form (round 3) ---------------------------------------------------------------------------------------------
// The kernel above feeds v_mfma_f32_32x32x2_f32 from one dword per lane per operand tile: MT + NT buffer_load_dword per
// k-pair, i.e. 12 load instructions of 256 B per 4 rows of a 64 x 128 product.  This form loads 16 BYTES per lane: lane
// (n16, kq) of a wave reads columns 4 n16 .. 4 n16 + 3 of row k + kq, so ONE buffer_load_dwordx4 brings 4 rows x 64 columns
// (1 KiB, contiguous for a dense operand), and its four components are the operands of four v_mfma_f32_16x16x4_f32 whose
// tile index i = n16 stands for column 4 i + r -- the M (and N) dimension is visited in a PERMUTED order that costs nothing:
// the accumulator of tile (r_a, r_b) holds C[64 ca + 16 kq + 4 v + r_a][64 cb + 4 n16 + r_b], so a lane's r_b = 0..3
// values are again 16 contiguous bytes of an output row.  3 load instructions per 32 MFMAs instead of 12 per 16, same MFMA
// cycles (32 x 32 = 16 x 64), same exact-f32 k-ordered sums per K slice (a different, still fixed, split of K than the
// narrow form's: results differ from it by fp32 rounding of the slice sums only).  W = 2 / 1 (dwordx2 / dword) serve
// operands 32 / 16 columns wide.  D = 4 operand sets are in flight per wave (D - 1 sets ahead of the MFMAs).  The ReLU-masked
// form (a third operand stream) stays on the narrow kernel: here it spills 22 registers at D = 4 (390 us at cfg2's 1M x 128 x 128)
// and starves at D = 3 (454 us) against 331 us there.
template <int W>
__device__ __forceinline__ void load_vec(float (&dst)[W], __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    // (element by element through scalars: __builtin_bit_cast(float, v[i]) on a vector ELEMENT reads the vector's first
    // dword for every i with this clang -- all four values came back equal)
    if constexpr (W == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        const unsigned u0 = v[0], u1 = v[1], u2 = v[2], u3 = v[3];
        dst[0] = __uint_as_float(u0);
        dst[1] = __uint_as_float(u1);
        dst[2] = __uint_as_float(u2);
        dst[3] = __uint_as_float(u3);
    } else if constexpr (W == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        const unsigned u0 = v[0], u1 = v[1];
        dst[0] = __uint_as_float(u0);
        dst[1] = __uint_as_float(u1);
    } else {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    }
}

using f32x4w = __attribute__((ext_vector_type(4))) float;
constexpr int kWideDepth = 4;

// WA / WB: floats per lane of an A / B chunk (chunk = 16 WA / 16 WB columns); CA / CB: chunks per wave.
// CYC (experiment): the block's four waves take the 4-row groups of the block's K range in turn (wave w: groups w, w + 4,
// ...) instead of a contiguous quarter each, so the block reads 16 consecutive rows per step.
template <int WA, int CA, int WB, int CB, bool CS, bool AMASK, int D = kWideDepth, int MINW = 2>
__global__ __launch_bounds__(kBlock, MINW) void gemm_tn_wide_kernel(
    const GemmSegs segs, const GemmForm form, float *__restrict__ slab, int64_t K, int M, int N, int64_t kslice_wave,
    int m_groups, int n_groups, int s_per_seg, int cyclic_flags)
{
    if (gemm_gated_off(form.gate, form.gate_when)) return;
    constexpr int TA = WA * CA, TB = WB * CB;
    static_assert(TA * TB <= 32, "accumulator tiles per wave");
    extern __shared__ float lds[];                       // one wave's accumulators: TA x TB x 4 x 64 floats (+ TA x 64)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    // The m / n groups of one K slice read the same rows of B / A.  Workgroups go to the 8 XCDs in turn, each with an L2 of
    // its own: dealt (mi fastest) the two m groups of cfg2's 128 x 128 products sit on DIFFERENT XCDs and B comes from HBM twice
    // (1.5 GB for 1 GB of operands).  flags bit 3: the groups of slice s are workgroups xcd + 8 g of a run of 8 G, i.e. on the
    // same XCD and resident together, so the second reader hits in L2.
    int grp, s;                                          // s: slab index = segment * s_per_seg + slice
    {
        const int G = m_groups * n_groups, b = blockIdx.x, S_all = (int)gridDim.x / G, full = S_all & ~7;
        if ((cyclic_flags & 8) && b < full * G) {
            const int run = b / (8 * G), within = b - run * 8 * G;
            s = run * 8 + (within & 7);
            grp = within >> 3;
        } else {
            const int r = (cyclic_flags & 8) ? b - full * G : b;
            grp = r % G;
            s = ((cyclic_flags & 8) ? full : 0) + r / G;
        }
    }
    const int cyclic = cyclic_flags & 7;
    const int mi = grp % m_groups;
    const int nj = grp / m_groups;
    const int seg = s / s_per_seg;
    const int sl = s - seg * s_per_seg;
    const float *__restrict__ A = segs.a[seg];
    const float *__restrict__ B = segs.b[seg];
    const float *__restrict__ B2 = segs.b2[seg];
    const float *__restrict__ AM = segs.am[seg];
    const int lda = form.lda, nsplit = form.nsplit;
    // contiguous: wave-uniform K slice [k0, k1) of kslice_wave rows.  cyclic: the block's range starts at kb, this wave's
    // first group at kb + 4 wave, groups 16 rows apart.
    // cyclic == 2: ALL waves of a segment take its 4-row groups in turn (wave (sl, w): groups sl * 4 + w, + 4 S, ...), so the
    // chip reads each operand of a segment as ONE moving window instead of 4 S private streams.
    const int64_t kb = (int64_t)sl * kWavesPerBlock * kslice_wave;
    const int64_t k0 = cyclic == 2 ? (int64_t)(sl * kWavesPerBlock + wave) * 4
                                   : (cyclic ? kb + 4 * wave : kb + (int64_t)wave * kslice_wave);
    const int64_t kend = cyclic == 2 ? K : (cyclic ? min(K, kb + kWavesPerBlock * kslice_wave) : min(K, k0 + kslice_wave));
    const int64_t rows = kend > k0 ? kend - k0 : 0;      // rows from this wave's first row to the end of its range
    const int gstride = cyclic == 2 ? 16 * s_per_seg : (cyclic ? 16 : 4);   // rows between this wave's consecutive groups
    // rows >= kend read as zero through the descriptors' range check (see the narrow kernel)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + k0 * lda), 0,
                                                       rows > 0 ? (int)(((rows - 1) * lda + M) * (int64_t)sizeof(float)) : 0,
                                                       0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(AMASK ? AM + k0 * lda : A + k0 * lda), 0,
                                                       rows > 0 ? (int)(((rows - 1) * lda + M) * (int64_t)sizeof(float)) : 0,
                                                       0x00020000);
    const auto rsB1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(B + k0 * form.ldb), 0,
        rows > 0 ? (int)(((rows - 1) * form.ldb + nsplit) * (int64_t)sizeof(float)) : 0, 0x00020000);
    const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(B2 ? B2 + k0 * form.ldb2 : B), 0,
        rows > 0 && B2 ? (int)(((rows - 1) * form.ldb2 + (N - nsplit)) * (int64_t)sizeof(float)) : 0, 0x00020000);
    int voA[CA], voB[CB], ldj[CB];
    bool second[CB];
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) voA[ca] = (kq * lda + (mi * CA + ca) * 16 * WA + WA * n16) * (int)sizeof(float);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col0 = (nj * CB + cb) * 16 * WB;
        second[cb] = col0 >= nsplit;                                             // wave-uniform
        ldj[cb] = second[cb] ? form.ldb2 : form.ldb;
        voB[cb] = (kq * ldj[cb] + (second[cb] ? col0 - nsplit : col0) + WB * n16) * (int)sizeof(float);
    }
    const int b_op = form.b_op;
    const float blo = form.lo, bhi = form.hi;

    f32x4w acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float cs[TA];
#pragma unroll
    for (int i = 0; i < TA; ++i) cs[i] = 0.f;

    float a[D][CA][WA], b[D][CB][WB], am[AMASK ? D : 1][AMASK ? CA : 1][AMASK ? WA : 1];
    auto load_set = [&](int d, int step) {
        const int r0 = gstride * step;                                          // uniform
        const int soA = r0 * lda * (int)sizeof(float);
#pragma unroll
        for (int ca = 0; ca < CA; ++ca) {
            load_vec<WA>(a[d][ca], rsA, voA[ca], soA);
            if constexpr (AMASK) load_vec<WA>(am[d][ca], rsM, voA[ca], soA);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) load_vec<WB>(b[d][cb], second[cb] ? rsB2 : rsB1, voB[cb], r0 * ldj[cb] * (int)sizeof(float));
    };
    auto consume = [&](int d) {
        if (b_op != STG_GEMM_B_NONE) {                                          // wave-uniform
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if (second[cb]) continue;
#pragma unroll
                for (int r = 0; r < WB; ++r) {
                    const float v = b[d][cb][r];
                    b[d][cb][r] = b_op == STG_GEMM_B_RELU ? (v < 0.f ? 0.f : v) : __builtin_amdgcn_fmed3f(v, blo, bhi);
                }
            }
        }
        if constexpr (AMASK) {
#pragma unroll
            for (int ca = 0; ca < CA; ++ca)
#pragma unroll
                for (int r = 0; r < WA; ++r) a[d][ca][r] = am[d][ca][r] > 0.f ? a[d][ca][r] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TA; ++i) {
            const float av = a[d][i / WA][i % WA];
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[d][j / WB][j % WB], acc[i][j], 0, 0, 0);
            if constexpr (CS) cs[i] += av;
        }
    };

    const int nsteps = (int)((rows + gstride - 1) / gstride);
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nsteps) load_set(d, d);
    int i = 0;
    for (; i + 2 * D - 1 <= nsteps; i += D) {               // steady state: straight-line, D - 1 sets ahead of the MFMAs
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_set((d + D - 1) % D, i + d + D - 1);
            consume(d);
        }
    }
    for (; i < nsteps; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i + d < nsteps) {
                if (i + d + D - 1 < nsteps) load_set((d + D - 1) % D, i + d + D - 1);
                consume(d);
            }
        }
    }

    // the block's 4 K sub-slices, added in wave order through one wave-sized LDS buffer (as in the narrow kernel)
    constexpr int kAccFloats = TA * TB * 4 * kWave;
    for (int w = 1; w < kWavesPerBlock; ++w) {
        if (wave == w) {
            float *dst = lds + lane;
#pragma unroll
            for (int ii = 0; ii < TA; ++ii)
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((ii * TB + j) * 4 + r) * kWave] = acc[ii][j][r];
            if constexpr (CS) {
#pragma unroll
                for (int ii = 0; ii < TA; ++ii) lds[kAccFloats + ii * kWave + lane] = cs[ii];
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float *src = lds + lane;
#pragma unroll
            for (int ii = 0; ii < TA; ++ii)
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ii][j][r] += src[((ii * TB + j) * 4 + r) * kWave];
            if constexpr (CS) {
#pragma unroll
                for (int ii = 0; ii < TA; ++ii) cs[ii] += lds[kAccFloats + ii * kWave + lane];
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        // acc[ca WA + ra][cb WB + rb][v] = C[chunk_a(ca) + WA (4 kq + v) + ra][chunk_b(cb) + WB n16 + rb]
        const int64_t slab_stride = (int64_t)M * N + (CS ? M : 0);
        float *out = slab + (int64_t)s * slab_stride;
#pragma unroll
        for (int ii = 0; ii < TA; ++ii) {
            const int ca = ii / WA, ra = ii % WA;
            if constexpr (CS) {
                float tot = cs[ii] + __shfl_xor(cs[ii], 16, kWave);             // the four kq lanes hold rows k = kq mod 4
                tot = tot + __shfl_xor(tot, 32, kWave);
                if (nj == 0 && kq == 0) out[(int64_t)M * N + (mi * CA + ca) * 16 * WA + WA * n16 + ra] = tot;
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = (mi * CA + ca) * 16 * WA + WA * (4 * kq + v) + ra;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    float *o = out + (int64_t)m * N + (nj * CB + cb) * 16 * WB + WB * n16;
                    if constexpr (WB == 4)
                        *reinterpret_cast<float4 *>(o) = make_float4(acc[ii][cb * WB][v], acc[ii][cb * WB + 1][v], acc[ii][cb * WB + 2][v],
                                                                     acc[ii][cb * WB + 3][v]);
                    else if constexpr (WB == 2)
                        *reinterpret_cast<float2 *>(o) = make_float2(acc[ii][cb * WB][v], acc[ii][cb * WB + 1][v]);
                    else
                        o[0] = acc[ii][cb * WB][v];
                }
            }
        }
    }
}

// C[o] = sum_s slab[s][o]: 64 outputs per block, the 4 waves take s = w, w+4, ... (8 loads in
// flight each) and are combined in wave order through LDS -- fixed order, deterministic.
__global__ __launch_bounds__(kBlock) void gemm_tn_reduce_kernel(const float *__restrict__ slab,
                                                                float *__restrict__ C, float *__restrict__ CS,
                                                                int64_t MNc, int64_t MN, int S, const int *__restrict__ gate = nullptr,
                                                                int gate_when = 0)
{
    if (gemm_gated_off(gate, gate_when)) return;
    __shared__ float part[kWavesPerBlock][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x >> 6;
    // slab rows are MNc = MN (+ M column sums) long; outputs [0, MN) go to C, [MN, MNc) to CS
    const int64_t o = (int64_t)blockIdx.x * kWave + lane;
    float acc = 0.f;
    if (o < MNc) {
        int s = w;
        for (; s + 7 * kWavesPerBlock < S; s += 8 * kWavesPerBlock) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(s + u * kWavesPerBlock) * MNc + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s < S; s += kWavesPerBlock) acc += slab[(int64_t)s * MNc + o];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && o < MNc) {
        const float tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        if (o < MN) C[o] = tot;
        else CS[o - MN] = tot;
    }
}

// The same reduction for up to kReduceJobs products in ONE launch (blockIdx.y = job): a BPTT window's six weight-gradient
// contractions leave their slabs and are summed together (six 9 us launches otherwise, at the launch floor).
constexpr int kReduceJobs = 8;
constexpr int kReduceBlocks = STG_GEMM_REDUCE_BLOCKS;
struct ReduceJobs {
    const float *slab[kReduceJobs];
    float *C[kReduceJobs], *CS[kReduceJobs];
    int64_t MNc[kReduceJobs], MN[kReduceJobs];
    int S[kReduceJobs];
    // rows > 0: C [M, N] leaves as M / rows blocks, each TRANSPOSED into its own [N, rows] array Cb[.][b] (and its colsum slice
    // into CSb[.][b]) -- three layers' weight gradients computed as one stacked product land in their parameters' layout
    int rows[kReduceJobs], N[kReduceJobs];
    float *Cb[kReduceJobs][kReduceBlocks], *CSb[kReduceJobs][kReduceBlocks];
};
__global__ __launch_bounds__(kBlock) void gemm_tn_reduce_multi_kernel(const ReduceJobs jobs)
{
    __shared__ float part[kWavesPerBlock][kWave];
    const int job = blockIdx.y;
    const float *__restrict__ slab = jobs.slab[job];
    const int64_t MNc = jobs.MNc[job], MN = jobs.MN[job];
    const int S = jobs.S[job];
    if ((int64_t)blockIdx.x * kWave >= MNc) return;              // block-uniform: this job has fewer outputs than the widest
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x >> 6;
    const int64_t o = (int64_t)blockIdx.x * kWave + lane;
    float acc = 0.f;
    if (o < MNc) {                                               // (arithmetic and order of gemm_tn_reduce_kernel)
        int s = w;
        for (; s + 7 * kWavesPerBlock < S; s += 8 * kWavesPerBlock) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(s + u * kWavesPerBlock) * MNc + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s < S; s += kWavesPerBlock) acc += slab[(int64_t)s * MNc + o];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && o < MNc) {
        const float tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        const int rows = jobs.rows[job];
        if (rows == 0) {
            if (o < MN) jobs.C[job][o] = tot;
            else jobs.CS[job][o - MN] = tot;
        } else if (o < MN) {
            const int N = jobs.N[job];
            const int m = (int)(o / N), n = (int)(o - (int64_t)m * N), b = m / rows;
            jobs.Cb[job][b][(int64_t)n * rows + (m - b * rows)] = tot;
        } else {
            const int m = (int)(o - MN), b = m / rows;
            jobs.CSb[job][b][m - b * rows] = tot;
        }
    }
}

namespace {

struct GemmPlan {
    int nt, mt, m_tiles, n_groups, S;          // m_tiles = groups of mt 32-row tiles
    int64_t kslice_wave;
};

// The wide kernel's tile of one wave for (M, N), or wa = 0 where the narrow kernel serves (other widths).
struct WideShape {
    int wa, ca, wb, cb;
};

WideShape wide_shape(int M, int N, int nsplit)
{
    WideShape w{0, 0, 0, 0};
    if (M % 64 == 0) {
        w.wa = 4;
        if (N % 64 == 0) {
            w.wb = 4;
            if (N % 128 == 0) { w.ca = 1; w.cb = 2; }
            else if (M % 128 == 0) { w.ca = 2; w.cb = 1; }
            else { w.ca = 1; w.cb = 1; }
        } else if (N == 32) {
            w.wb = 2; w.cb = 1;
            w.ca = M % 192 == 0 ? 3 : (M % 128 == 0 ? 2 : 1);
        } else if (N == 96) {                                   // d_g^T [Hx | P] of a TGCN gate (C = 64, Fin = 32): 4 x 6 tiles per wave
            w.wb = 2; w.cb = 3; w.ca = 1;
        } else {
            w.wa = 0;
        }
    } else if (M == 32 && N % 64 == 0) {
        w.wa = 2; w.ca = 1; w.wb = 4; w.cb = N % 128 == 0 ? 2 : 1;
    }
    if (w.wa && nsplit < N && nsplit % (16 * w.wb) != 0) w.wa = 0;            // B's two matrices must meet on a chunk boundary
    return w;
}

GemmPlan plan_gemm_tn(int64_t K, int M, int N, int T = 1, int max_ld = 0, const WideShape *wide = nullptr)
{
    GemmPlan p{};
    p.nt = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
    p.mt = M > 32 ? 2 : 1;
    p.m_tiles = (M + 32 * p.mt - 1) / (32 * p.mt);
    p.n_groups = (N + 32 * p.nt - 1) / (32 * p.nt);
    if (wide && wide->wa) {
        p.m_tiles = M / (16 * wide->wa * wide->ca);
        p.n_groups = N / (16 * wide->wb * wide->cb);
    }
    const int64_t tiles = (int64_t)p.m_tiles * p.n_groups;
    // at most 512 blocks = ONE resident round (2 per CU): a 550-block grid (T = 25, 2 tiles: S rounded up to 11)
    // ran a second round for its last 38 blocks, i.e. at half speed; never slice below 64 rows per wave
    int64_t S = std::max<int64_t>(1, 512 / (tiles * T));        // slices PER SEGMENT
    const int64_t max_S = std::max<int64_t>(1, K / (64 * kWavesPerBlock));
    S = std::max<int64_t>(1, std::min(S, max_S));
    int64_t per_wave = (K + S * kWavesPerBlock - 1) / (S * kWavesPerBlock);
    per_wave = (per_wave + kGemmKStep - 1) / kGemmKStep * kGemmKStep;
    // a wave addresses its K slice through 32-bit buffer descriptors and scalar row offsets: rows * max(M, N) * 4
    // bytes must stay below 2^31 (more slices instead of longer ones; 0 = no slice length fits: unsupported)
    const int64_t cap = ((int64_t)INT32_MAX / (4 * (int64_t)std::max(std::max(M, N), max_ld))) / kGemmKStep * kGemmKStep;
    if (cap < kGemmKStep) { p.S = 0; return p; }
    per_wave = std::min(per_wave, cap);
    p.kslice_wave = std::max<int64_t>(per_wave, kGemmKStep);
    p.S = (int)((K + p.kslice_wave * kWavesPerBlock - 1) / (p.kslice_wave * kWavesPerBlock));
    if (p.S < 1) p.S = 1;
    return p;
}

// bytes of slab space that covers whichever form the launch picks (the wide form may cut K into more slices)
size_t workspace_need(int T, int64_t K, int M, int N, int max_ld)
{
    const GemmPlan p = plan_gemm_tn(K, M, N, T, max_ld);
    int S = p.S;
    const WideShape w = wide_shape(M, N, N);
    if (w.wa) S = std::max(S, plan_gemm_tn(K, M, N, T, max_ld, &w).S);
    if (S < 1) return 0;
    const size_t x3_slabs = (size_t)gemm_tn_x3_slabs(M, N, K, T);              // the split form's slab count (gemm_tn_x3.hip)
    return std::max((size_t)T * (size_t)S, x3_slabs) * ((size_t)M * (size_t)N + (size_t)M) * sizeof(float);
}

}  // namespace
}  // namespace stg

extern "C" size_t stg_gemm_tn_workspace_bytes(int64_t K, int32_t M, int32_t N)
{
    if (K <= 0 || M <= 0 || N <= 0) return 0;
    return stg::workspace_need(1, K, M, N, 0);
}

extern "C" size_t stg_gemm_tn_multi_workspace_bytes(int32_t T, int64_t K, int32_t M, int32_t N)
{
    if (T <= 0 || T > stg::kGemmMaxSeg || K <= 0 || M <= 0 || N <= 0) return 0;
    return stg::workspace_need(T, K, M, N, 0);
}

namespace stg {
namespace {
int gemm_tn_run(const float *const *As, const float *const *Bs, int T, float *C, float *colsum, int64_t K,
                int32_t M, int32_t N, void *workspace, size_t workspace_bytes, void *stream_, const char *what,
                const float *const *B2s = nullptr, const GemmForm *form_in = nullptr, const float *const *AMs = nullptr,
                int *defer_slabs = nullptr)
{
    // defer_slabs (non-NULL): leave the slabs in the workspace, write their count there and skip the reduction
    // (stg_gemm_tn_reduce_multi_f32 sums several products' slabs in one launch); C / colsum then only say what is wanted.
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (K < 0 || M <= 0 || N <= 0 || T <= 0 || T > kGemmMaxSeg)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad shape T=%d K=%lld M=%d N=%d (T <= %d)", what, T,
                    (long long)K, M, N, kGemmMaxSeg);
    if ((int64_t)2 * M * N > INT32_MAX || (int64_t)2 * M > INT32_MAX / 2)
        return fail(STG_ERR_UNSUPPORTED, "%s: output %d x %d too large for the tall-skinny kernel", what, M, N);
    if (!C) return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL output", what);
    if (K == 0 && defer_slabs) return fail(STG_ERR_INVALID_ARGUMENT, "%s: K = 0 has no slabs to defer", what);
    if (K == 0) {
        if (const int rc = zero_async(C, sizeof(float) * (size_t)M * N, stream)) return rc;
        return colsum ? zero_async(colsum, sizeof(float) * (size_t)M, stream) : 0;
    }
    if (!As || !Bs || !workspace) return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", what);
    GemmForm form{M, N, 0, N, STG_GEMM_B_NONE, 0.f, 0.f, 0};
    if (form_in) form = *form_in;
    if (form.a_mask && !AMs) return fail(STG_ERR_INVALID_ARGUMENT, "%s: a_mask without mask operands", what);
    if (form.lda < M || form.nsplit < 0 || form.nsplit > N || (form.nsplit < N && form.nsplit % 32 != 0) ||
        form.ldb < form.nsplit || (form.nsplit < N && (!B2s || form.ldb2 < N - form.nsplit)) ||
        form.b_op < STG_GEMM_B_NONE || form.b_op > STG_GEMM_B_RELU)
        return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad operand form (lda=%d ldb=%d ldb2=%d nsplit=%d b_op=%d)", what,
                    form.lda, form.ldb, form.ldb2, form.nsplit, form.b_op);
    if (form.nsplit == 0) return fail(STG_ERR_INVALID_ARGUMENT, "%s: nsplit must be > 0 (pass the matrix as B)", what);
    GemmSegs segs{};
    for (int t = 0; t < T; ++t) {
        if (!As[t] || !Bs[t] || (form.nsplit < N && !B2s[t]))
            return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL operand in segment %d", what, t);
        segs.a[t] = As[t];
        segs.b[t] = Bs[t];
        segs.b2[t] = form.nsplit < N ? B2s[t] : nullptr;
        segs.am[t] = form.a_mask ? AMs[t] : nullptr;
        if (form.a_mask && !AMs[t]) return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL mask in segment %d", what, t);
    }
    WideShape wide = wide_shape(M, N, form.nsplit);
    if (tuning().gemm_wide == 1 || form.a_mask || form.lda % 4 != 0 || form.ldb % 4 != 0 || (form.nsplit < N && form.ldb2 % 4 != 0)) wide.wa = 0;
    for (int t = 0; t < T && wide.wa; ++t) {                      // 16-byte lane loads: every operand base on a 16-byte boundary
        const uintptr_t bits = reinterpret_cast<uintptr_t>(segs.a[t]) | reinterpret_cast<uintptr_t>(segs.b[t]) |
                               reinterpret_cast<uintptr_t>(segs.b2[t]) | reinterpret_cast<uintptr_t>(segs.am[t]);
        if (bits & 15) wide.wa = 0;
    }
    // the 3-term bf16 split form (gemm_tn_x3.hip) where it covers the shape: from 64 K rows in all (knob "gemm_x3": 1 = never, 2 = always)
    {
        bool x3 = tuning().gemm_x3 != 1 && (tuning().gemm_x3 == 2 || (int64_t)K * T >= 65536) && gemm_tn_x3_covers(M, N, form, K, T);
        for (int t = 0; t < T && x3; ++t) {
            const uintptr_t bits = reinterpret_cast<uintptr_t>(segs.a[t]) | reinterpret_cast<uintptr_t>(segs.b[t]) | reinterpret_cast<uintptr_t>(segs.b2[t]);
            if (bits & 15) x3 = false;
        }
        const bool cs3 = colsum != nullptr;
        const int64_t MN3 = (int64_t)M * N, MNc3 = MN3 + (cs3 ? M : 0);
        if (x3 && workspace_bytes >= (size_t)gemm_tn_x3_slabs(M, N, K, T) * (size_t)MNc3 * sizeof(float)) {
            int slabs = 0;
            if (const int rc = gemm_tn_x3_launch(segs, form, static_cast<float *>(workspace), K, M, N, T, cs3, &slabs, stream)) return rc;
            if (defer_slabs) {
                *defer_slabs = slabs;
                return 0;
            }
            hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((MNc3 + kWave - 1) / kWave)), dim3(kBlock), 0, stream,
                               static_cast<float *>(workspace), C, colsum, MNc3, MN3, slabs, form.gate, form.gate_when);
            return check_launch(what);
        }
    }
    const GemmPlan p = plan_gemm_tn(K, M, N, T, std::max(form.lda, std::max(form.ldb, form.ldb2)), wide.wa ? &wide : nullptr);
    if (p.S < 1 || (int64_t)T * p.S > INT32_MAX / 2)
        return fail(STG_ERR_UNSUPPORTED, "%s: K=%lld x max(M, N)=%d is outside the 32-bit slice addressing", what,
                    (long long)K, std::max(M, N));
    const bool cs = colsum != nullptr;
    const int64_t MN = (int64_t)M * N, MNc = MN + (cs ? M : 0);
    const int S_total = T * p.S;
    const size_t need = (size_t)S_total * (size_t)MNc * sizeof(float);
    if (workspace_bytes < need)
        return fail(STG_ERR_WORKSPACE, "%s: workspace %zu < required %zu", what, workspace_bytes, need);
    float *slab = static_cast<float *>(workspace);
    const int64_t blocks = (int64_t)S_total * p.m_tiles * p.n_groups;
    if (wide.wa) {
        const size_t wlds = ((size_t)wide.wa * wide.ca * wide.wb * wide.cb * 4 + (size_t)wide.wa * wide.ca) * kWave * sizeof(float);
        // (cyclic: a wave's descriptor spans the whole block range, 4 x kslice_wave rows: only while that stays inside the
        // 32-bit addressing the plan sized for one slice)
        const int64_t max_ld_ = std::max(form.lda, std::max(form.ldb, form.ldb2));
        int cyclic = tuning().gemm_cyclic;
        if (cyclic == 1 && 4 * p.kslice_wave * max_ld_ * 4 >= (int64_t)INT32_MAX) cyclic = 0;
        if (cyclic == 2 && K * max_ld_ * 4 >= (int64_t)INT32_MAX) cyclic = 0;
        if (tuning().gemm_xcd_pair == 0 && p.m_tiles * p.n_groups > 1) cyclic |= 8;      // the groups of a K slice on one XCD
#define STG_WIDE_L(WA_, CA_, WB_, CB_, CS_, AM_)                                                                   \
    hipLaunchKernelGGL((gemm_tn_wide_kernel<WA_, CA_, WB_, CB_, CS_, AM_>), dim3((unsigned)blocks), dim3(kBlock), wlds, stream, segs, \
                       form, slab, K, M, N, p.kslice_wave, p.m_tiles, p.n_groups, p.S, cyclic)
#define STG_WIDE(WA_, CA_, WB_, CB_)                                                                               \
    if (wide.wa == WA_ && wide.ca == CA_ && wide.wb == WB_ && wide.cb == CB_) {                                    \
        if (form.a_mask) {                                                                                         \
            if (cs) STG_WIDE_L(WA_, CA_, WB_, CB_, true, true); else STG_WIDE_L(WA_, CA_, WB_, CB_, false, true);   \
        } else {                                                                                                   \
            if (cs) STG_WIDE_L(WA_, CA_, WB_, CB_, true, false); else STG_WIDE_L(WA_, CA_, WB_, CB_, false, false); \
        }                                                                                                          \
    }
        STG_WIDE(4, 1, 4, 2) else STG_WIDE(4, 2, 4, 1) else STG_WIDE(4, 1, 4, 1) else STG_WIDE(4, 3, 2, 1) else STG_WIDE(4, 2, 2, 1)
        else STG_WIDE(4, 1, 2, 1) else STG_WIDE(4, 1, 2, 3) else STG_WIDE(2, 1, 4, 2) else STG_WIDE(2, 1, 4, 1)
        else return fail(STG_ERR_UNSUPPORTED, "%s: no wide instantiation for this tile", what);
#undef STG_WIDE
#undef STG_WIDE_L
        if (defer_slabs) {
            *defer_slabs = S_total;
            return check_launch(what);
        }
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((MNc + kWave - 1) / kWave)), dim3(kBlock), 0, stream,
                           slab, C, colsum, MNc, MN, S_total, form.gate, form.gate_when);
        return check_launch(what);
    }
    const size_t lds = ((size_t)p.mt * p.nt * 16 + p.mt) * kWave * sizeof(float);
#define STG_GEMM_LAUNCH(NT_, KU_, CS_, MT_)                                                                       \
    do {                                                                                                          \
        if (form.a_mask)                                                                                          \
            hipLaunchKernelGGL((gemm_tn_partial_kernel<NT_, KU_, CS_, MT_, true>), dim3((unsigned)blocks),        \
                               dim3(kBlock), lds, stream, segs, form, slab, K, M, N, p.kslice_wave, p.m_tiles,    \
                               p.n_groups, p.S);                                                                  \
        else                                                                                                      \
            hipLaunchKernelGGL((gemm_tn_partial_kernel<NT_, KU_, CS_, MT_>), dim3((unsigned)blocks), dim3(kBlock), \
                               lds, stream, segs, form, slab, K, M, N, p.kslice_wave, p.m_tiles, p.n_groups, p.S); \
    } while (0)
#define STG_GEMM_NT(CS_, MT_)                                                                                     \
    if (p.nt == 1) STG_GEMM_LAUNCH(1, 8, CS_, MT_); else if (p.nt == 2) STG_GEMM_LAUNCH(2, 8, CS_, MT_);         \
    else STG_GEMM_LAUNCH(4, 4, CS_, MT_)
    if (cs) {
        if (p.mt == 2) { STG_GEMM_NT(true, 2); } else { STG_GEMM_NT(true, 1); }
    } else {
        if (p.mt == 2) { STG_GEMM_NT(false, 2); } else { STG_GEMM_NT(false, 1); }
    }
#undef STG_GEMM_NT
#undef STG_GEMM_LAUNCH
    if (defer_slabs) {
        *defer_slabs = S_total;
        return check_launch(what);
    }
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((MNc + kWave - 1) / kWave)), dim3(kBlock), 0, stream,
                       slab, C, colsum, MNc, MN, S_total, form.gate, form.gate_when);
    return check_launch(what);
}
}  // namespace
}  // namespace stg

extern "C" int stg_gemm_tn_f32(const float *A, const float *B, float *C, int64_t K, int32_t M, int32_t N,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    return stg::gemm_tn_run(&A, &B, 1, C, nullptr, K, M, N, workspace, workspace_bytes, stream, "stg_gemm_tn_f32");
}

extern "C" int stg_gemm_tn_gated_f32(const float *A, const float *B, float *C, int64_t K, int32_t M, int32_t N, void *workspace,
                                     size_t workspace_bytes, const int32_t *gate, int gate_when, void *stream)
{
    using namespace stg;
    if (!gate || (gate_when != 1 && gate_when != 2))
        return fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_gated_f32: gate [dev] and gate_when 1 (run if *gate == 0) or 2 (run if != 0)");
    if (K <= 0) return fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_gated_f32: K must be positive");
    GemmForm form{M, N, 0, N, STG_GEMM_B_NONE, 0.f, 0.f, 0, gate, gate_when};
    return gemm_tn_run(&A, &B, 1, C, nullptr, K, M, N, workspace, workspace_bytes, stream, "stg_gemm_tn_gated_f32", nullptr, &form);
}

extern "C" int stg_gemm_tn_colsum_f32(const float *A, const float *B, float *C, float *colsum_A, int64_t K,
                                      int32_t M, int32_t N, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!colsum_A) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_colsum_f32: NULL colsum output");
    return stg::gemm_tn_run(&A, &B, 1, C, colsum_A, K, M, N, workspace, workspace_bytes, stream,
                            "stg_gemm_tn_colsum_f32");
}

extern "C" int stg_gemm_tn_multi_f32(const float *const *A, const float *const *B, int32_t T, float *C,
                                     float *colsum_A, int64_t K, int32_t M, int32_t N, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    return stg::gemm_tn_run(A, B, T, C, colsum_A, K, M, N, workspace, workspace_bytes, stream,
                            "stg_gemm_tn_multi_f32");
}

extern "C" size_t stg_gemm_tn_form_workspace_bytes(int32_t T, int64_t K, int32_t M, int32_t N, int32_t max_ld)
{
    if (T <= 0 || T > stg::kGemmMaxSeg || K <= 0 || M <= 0 || N <= 0) return 0;
    return stg::workspace_need(T, K, M, N, max_ld);
}

extern "C" int stg_gemm_tn_form_f32(const float *const *A, int32_t lda, const float *const *B, int32_t ldb, int32_t nsplit,
                                    const float *const *B2, int32_t ldb2, int32_t b_op, float lo, float hi, int32_t T,
                                    float *C, float *colsum_A, int64_t K, int32_t M, int32_t N, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    const stg::GemmForm form{lda, ldb, ldb2, nsplit, b_op, lo, hi, 0};
    return stg::gemm_tn_run(A, B, T, C, colsum_A, K, M, N, workspace, workspace_bytes, stream, "stg_gemm_tn_form_f32", B2,
                            &form);
}

extern "C" int stg_gemm_tn_form_partial_f32(const float *const *A, int32_t lda, const float *const *B, int32_t ldb, int32_t nsplit,
                                            const float *const *B2, int32_t ldb2, int32_t b_op, float lo, float hi, int32_t T,
                                            int32_t want_colsum, int64_t K, int32_t M, int32_t N, void *workspace,
                                            size_t workspace_bytes, int32_t *slabs, void *stream)
{
    if (!slabs) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_form_partial_f32: NULL slab count");
    if (K <= 0) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_form_partial_f32: K must be positive");
    const stg::GemmForm form{lda, ldb, ldb2, nsplit, b_op, lo, hi, 0};
    float dummy = 0.f;                                      // gemm_tn_run only looks at which outputs are wanted here
    int S = 0;
    const int rc = stg::gemm_tn_run(A, B, T, &dummy, want_colsum ? &dummy : nullptr, K, M, N, workspace, workspace_bytes, stream,
                                    "stg_gemm_tn_form_partial_f32", B2, &form, nullptr, &S);
    *slabs = S;
    return rc;
}

extern "C" int stg_gemm_tn_reduce_multi_f32(int32_t count, const float *const *slabs, float *const *C, float *const *colsum,
                                            const int32_t *M, const int32_t *N, const int32_t *S, void *stream)
{
    return stg_gemm_tn_reduce_multi_blocks_f32(count, slabs, C, colsum, M, N, S, nullptr, nullptr, nullptr, stream);
}

extern "C" int stg_gemm_tn_reduce_multi_blocks_f32(int32_t count, const float *const *slabs, float *const *C, float *const *colsum,
                                                   const int32_t *M, const int32_t *N, const int32_t *S, const int32_t *block_rows,
                                                   float *const *C_blocks, float *const *colsum_blocks, void *stream)
{
    using namespace stg;
    const char *who = "stg_gemm_tn_reduce_multi_blocks_f32";
    if (count <= 0 || count > kReduceJobs) return fail(STG_ERR_INVALID_ARGUMENT, "%s: 1 .. %d products per call", who, kReduceJobs);
    if (!slabs || !C || !colsum || !M || !N || !S) return fail(STG_ERR_INVALID_ARGUMENT, "%s: NULL pointer argument", who);
    ReduceJobs jobs{};
    int64_t widest = 0;
    for (int i = 0; i < count; ++i) {
        const int rows = block_rows ? block_rows[i] : 0;
        if (!slabs[i] || M[i] <= 0 || N[i] <= 0 || S[i] <= 0 || rows < 0 || (rows == 0 && !C[i]))
            return fail(STG_ERR_INVALID_ARGUMENT, "%s: bad product %d", who, i);
        bool want_cs = colsum[i] != nullptr;
        if (rows > 0) {
            const int nb = M[i] / rows;
            if (M[i] % rows || nb > kReduceBlocks || !C_blocks || !colsum_blocks)
                return fail(STG_ERR_INVALID_ARGUMENT, "%s: product %d: M = %d is not 1 .. %d blocks of %d rows", who, i, M[i], kReduceBlocks, rows);
            want_cs = colsum_blocks[i * STG_GEMM_REDUCE_BLOCKS] != nullptr;
            for (int b = 0; b < nb; ++b) {
                jobs.Cb[i][b] = C_blocks[i * STG_GEMM_REDUCE_BLOCKS + b];
                jobs.CSb[i][b] = colsum_blocks[i * STG_GEMM_REDUCE_BLOCKS + b];
                if (!jobs.Cb[i][b] || (want_cs && !jobs.CSb[i][b]))
                    return fail(STG_ERR_INVALID_ARGUMENT, "%s: product %d: NULL block %d", who, i, b);
            }
        }
        jobs.rows[i] = rows;
        jobs.N[i] = N[i];
        jobs.slab[i] = slabs[i];
        jobs.C[i] = C[i];
        jobs.CS[i] = colsum[i];
        jobs.MN[i] = (int64_t)M[i] * N[i];
        jobs.MNc[i] = jobs.MN[i] + (want_cs ? M[i] : 0);
        jobs.S[i] = S[i];
        widest = std::max(widest, jobs.MNc[i]);
    }
    hipLaunchKernelGGL(gemm_tn_reduce_multi_kernel, dim3((unsigned)((widest + kWave - 1) / kWave), (unsigned)count), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), jobs);
    return check_launch(who);
}

extern "C" int stg_gemm_tn_relu_mask_f32(const float *A, const float *mask, const float *B, float *C, float *colsum_A,
                                         int64_t K, int32_t M, int32_t N, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    if (!mask) return stg::fail(STG_ERR_INVALID_ARGUMENT, "stg_gemm_tn_relu_mask_f32: NULL mask");
    const stg::GemmForm form{M, N, 0, N, STG_GEMM_B_NONE, 0.f, 0.f, 1};
    return stg::gemm_tn_run(&A, &B, 1, C, colsum_A, K, M, N, workspace, workspace_bytes, stream,
                            "stg_gemm_tn_relu_mask_f32", nullptr, &form, &mask);
}
