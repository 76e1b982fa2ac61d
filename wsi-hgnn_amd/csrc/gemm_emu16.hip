// fp32-class GEMMs on the 16-bit matrix cores: two emulations of an fp32 product, both accumulated in fp32.
//
// "bf16x6" (WSI_GEMM_BF16X6; gemm_bf16x6_kernel, all three ops): every fp32 operand x is split exactly into three bf16
// terms x = x0 + x1 + x2 (8+8+8 mantissa bits), and x*y is accumulated in fp32 from the six products whose magnitude
// is >= 2^-16 |x y|  (x0y0, x0y1, x1y0, x0y2, x2y0, x1y1); the dropped terms are <= 2^-23 |x y|, i.e. below one fp32
// rounding of the product.  bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so the result carries fp32-class error (tests/test_kernels_gpu.py::test_gemm_emulated_error_vs_fp32_mfma) at 6/16 of
// the fp32-MFMA matrix time (the fp32 matrix rate is 1/16 of the bf16 rate on gfx950; no xf32/TF32 exists).
// Same grouped launch / descriptor table / epilogues / split-K planning as gemm_f32.hip.  Differences:
//   * LDS holds bf16 planes [stage][A|B][3][128][16(+8)], k contiguous per row, so an A or B fragment of the 32x32x16
//     MFMA (8 consecutive k of one row per lane) is ONE conflict-free ds_read_b128 (row pitch 48 B);
//   * the split happens while staging (v_cvt_pk_bf16_f32 + exact residuals), K-contiguous operands store 8-byte
//     quads per plane, M/N-contiguous operands (dX's W, dW's dY and X) are transposed on the fly by packing the
//     (k, k+1) pair of each column into one 4-byte store;
//   * per 16-deep stage a wave issues 24 MFMAs (6 products x 2x2 tiles), term-major so consecutive MFMAs hit
//     different accumulators, small terms first, with the split of the NEXT stage interleaved between them.
//
// "fp16x3" (WSI_GEMM_FP16X3; gemm_fp16x3g_kernel / gemm_fp16x3w_kernel, NT and NN) does HALF the matrix work: x = x0 + x1 with
// two fp16 terms (11+11 significand bits, round-to-nearest at both levels: |x - x0| <= 2^-11 |x|, |x - x0 - x1| <= 2^-23 |x| -
// measured maximum over 4 M values 2^-23.0, i.e. ONE BIT short of fp32's 2^-24) and three products (x0y0, x0y1, x1y0; the
// dropped x1y1 is <= 2^-22 |x y|).  Per product that is <= 2^-23 + 2^-23 + 2^-22 = 2^-21 |x y| where exact fp32 has 0 (its
// error is the 2^-24 per accumulation step); for K >~ 100 the accumulation error dominates either way; for short dot products
// the bound allows up to 4x fp32's element-wise error, measured is 0.97x / 0.37x / 0.52x of fp32's maximum at K = 16 / 64 / 128
// (mean 1.3x at K = 16; tests/test_kernels_gpu.py::test_gemm_fp16x3_short_dot_products prints and bounds the factor).  fp16 has 5 exponent bits and the matrix cores flush fp16 denormals, so
//   * every operand is scaled by a power of two per index of the output it contributes to (row m of A / column n of B:
//     the scale leaves the contraction and is undone exactly with one v_ldexp_f32 in the epilogue): 2^-e with
//     e = exponent(absmax over the contraction axis) - 14 puts the largest element of each row in [2^14, 2^15);
//   * the second term is stored as 2^11 x1 and the two cross products go to a SECOND accumulator set that is folded in
//     with weight 2^-11 at the end: both planes are normal fp16 numbers for every element within 2^-28 of its row's
//     largest (smaller ones are flushed: an absolute error <= 2^-28 of the row's largest, the normwise fp32 class).  An
//     element 2^-d below its row's largest keeps all 22 bits for d <= 17 and 39 - d bits beyond (the low plane's tail goes
//     under the fp16 normal range): 2^-19 relative at d = 20, 2^-15 at d = 24 - invisible in a sum that contains the large
//     element, visible when that element meets an exact zero (::test_gemm_fp16x3_outlier_times_zero).
// The absmax bits come from a pre-pass (absmax_rows for A, inside pack_b_frag_kernel for B) into the call's workspace.  What was measured on
// the way (MI355X, kqv forward shape 80000 x 1536 x 512, TFLOP/s fp32-equivalent incl. the pre-pass; bf16x6 = 177):
//   both operands split in the kernel like bf16x6, one accumulator (flushes: 2^-13 errors on outlier rows)   246
//   the same with the two accumulator sets                                                                    228
//   ... with the residual as v_fma_mixlo/mixhi_f16 inline asm (5 instead of 9 VALU per pair)                  211
//   B packed in fragment order and loaded straight into registers (the kernel below)                          247
//   ... B fetched two stages ahead into three fragment sets (256 VGPRs, 2 spills)                             247
//   ... A fetched two stages ahead into three register sets (256 VGPRs, no spill): no gain, dropped
//   ... s_setprio(3) around the MFMAs (worth 1-2.5 % in gemm_f32.hip): 0.9 % slower here (same-box A/B, 3 alternations)
//   ... 256 x 128 tile, one wave per SIMD, 4 x 2 register blocking, all 256 AGPRs as accumulators             186
//   ablation of the kernel below: no A loads 289 / no B loads 282 / neither 331 / no split + LDS write 303 /
//   fragment reads + MFMAs + barrier only 342: the stage is balanced between the vector L1 (64 B/clk/CU: A tile + the B
//   fragments two waves fetch redundantly), the LDS (A planes) and the matrix pipe (0.44 busy), no single limiter.
// A scaled-fp16 TN (weight gradients: both operands are activations, scales per column over all nodes) costs more in
// absmax passes than it saves: in the fp16x3 mode the TN launches run as bf16x6.
#include "gemm_common.h"
#include "emu16.h"
#include <type_traits>

namespace wsi {

template <int MODE> struct Emu;
template <> struct Emu<0> {
    static constexpr int NP = 3, NT = 6;
    typedef bf16x8 frag;
    static constexpr int NACC = 1;
    static constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    static constexpr int TC[6] = {0, 0, 0, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Emu<1> {
    static constexpr int NP = 2, NT = 3;
    typedef f16x8 frag;
    static constexpr int NACC = 2;                  // [0]: x0 y0, [1]: 2^11 (x0 y1 + x1 y0)
    static constexpr int TA[3] = {0, 1, 0};          // x0 y1, x1 y0 (-> [1]), x0 y0 (-> [0]): the order of gemm_fp16x3g_kernel too
    static constexpr int TB[3] = {1, 0, 0};
    static constexpr int TC[3] = {1, 1, 0};
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float4 load4_guarded_b(const float* __restrict__ p, int i0, int n) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + 0 < n) r.x = p[0];
    if (i0 + 1 < n) r.y = p[1];
    if (i0 + 2 < n) r.z = p[2];
    if (i0 + 3 < n) r.w = p[3];
    return r;
}

// ---- epilogue (same contract as gemm_f32.hip); fsm = >= 32 KB of LDS no longer read by anyone
template <bool SPLITK>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& P, const GroupDesc& G, float* __restrict__ ws, float* fsm,
                                              f32x16 (&acc)[2][2], int row0, int n0, bool interior, int split, int wave, int lane) {
    // row0: first row of the 64 x 64 block this wave holds in acc (columns n0 + 64 (wave & 1) ...); interior: the whole
    // workgroup tile lies inside C (uniform over the workgroup: the vectorised path below has barriers)
    const int wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int epi = P.epilogue;
    WSI_DROP_SEED(G, epi);
    float gate_s = 1.f;
    if (!SPLITK && (epi & (WSI_EPI_SCALE_GATE | WSI_EPI_R_1MG)) && G.gate) gate_s = 1.f / (1.f + expf(-(*G.gate)));
    const float r_scale = (epi & WSI_EPI_R_1MG) ? (1.f - gate_s) : 1.f;

    if (interior && (G.flags & 4)) {
        float* wbuf = fsm + wave * (32 * 64);
        float* cbase;
        int64_t ldc;
        if (SPLITK) { cbase = ws + G.ws_off + (int64_t)split * G.M * G.N; ldc = G.N; }
        else { cbase = G.C; ldc = G.ldc; }
        const int rr0 = lane >> 4, c4 = (lane & 15) * 4;
        const int col = n0 + wn * 64 + c4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!SPLITK && (epi & WSI_EPI_BIAS) && G.bias) bv = make_float4(G.bias[col], G.bias[col + 1], G.bias[col + 2], G.bias[col + 3]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wbuf[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rr = q * 4 + rr0;
                const int row = row0 + i * 32 + rr;
                float4 x = *reinterpret_cast<const float4*>(wbuf + rr * 64 + c4);
                float* c = cbase + (int64_t)row * ldc + col;
                if (!SPLITK) {
                    x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w;
                    if (epi & WSI_EPI_GELU) { x.x = gelu_erf(x.x); x.y = gelu_erf(x.y); x.z = gelu_erf(x.z); x.w = gelu_erf(x.w); }
                    if (epi & WSI_EPI_MUL_M) {
                        const float4 mv = *reinterpret_cast<const float4*>(G.Mm + (int64_t)row * G.ldm + col);
                        x.x *= mv.x; x.y *= mv.y; x.z *= mv.z; x.w *= mv.w;
                    }
                    if (epi & WSI_EPI_DROPOUT) {
                        const float4 mv = WSI_DROP4(G, row, col);
                        x.x *= mv.x; x.y *= mv.y; x.z *= mv.z; x.w *= mv.w;
                    }
                    if (epi & WSI_EPI_SCALE_GATE) { x.x *= gate_s; x.y *= gate_s; x.z *= gate_s; x.w *= gate_s; }
                    if (epi & WSI_EPI_ADD_R) {
                        const float4 rv = *reinterpret_cast<const float4*>(G.R + (int64_t)row * G.ldr + col);
                        x.x = fmaf(r_scale, rv.x, x.x); x.y = fmaf(r_scale, rv.y, x.y);
                        x.z = fmaf(r_scale, rv.z, x.z); x.w = fmaf(r_scale, rv.w, x.w);
                    }
                    if (epi & WSI_EPI_ACCUMULATE) {
                        const float4 o = *reinterpret_cast<const float4*>(c);
                        x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
                    }
                    if (G.c_absmax) {   // absmax bits of this row over the wave's 64 columns -> the wave's own slot (plain store)
                        // (DPP within the 16 lanes that hold the row: VALU speed)
                        float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
                        m = fmaxf(m, dpp_mov<0xB1>(m));     // quad_perm [1,0,3,2]
                        m = fmaxf(m, dpp_mov<0x4E>(m));     // quad_perm [2,3,0,1]
                        m = fmaxf(m, dpp_mov<0x141>(m));    // row_half_mirror
                        m = fmaxf(m, dpp_mov<0x140>(m));    // row_mirror
                        if ((lane & 15) == 0) G.c_absmax[(int64_t)row * G.c_parts + G.c_first + 2 * (n0 / BN) + wn] = __float_as_uint(m);
                    }
                }
                *reinterpret_cast<float4*>(c) = x;
            }
            __syncthreads();
        }
        return;
    }
    if (SPLITK) {
        float* wsp = ws + G.ws_off + (int64_t)split * G.M * G.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < G.M && col < G.N) wsp[(int64_t)row * G.N + col] = acc[i][j][r];
                }
        }
        return;
    }
    float rmax[2][16];                   // edge tiles with c_absmax: |final value| per (i, r), max over this lane's two columns
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) rmax[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool colok = col < G.N;
        float bv = 0.f;
        if ((epi & WSI_EPI_BIAS) && G.bias && colok) bv = G.bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (!(colok && row < G.M)) continue;
                float x = acc[i][j][r] + bv;
                if (epi & WSI_EPI_GELU) x = gelu_erf(x);
                if (epi & WSI_EPI_MUL_M) x *= G.Mm[(int64_t)row * G.ldm + col];
                if (epi & WSI_EPI_DROPOUT) x *= WSI_DROP1(G, row, col);
                if (epi & WSI_EPI_SCALE_GATE) x *= gate_s;
                if (epi & WSI_EPI_ADD_R) x = fmaf(r_scale, G.R[(int64_t)row * G.ldr + col], x);
                float* c = G.C + (int64_t)row * G.ldc + col;
                if (epi & WSI_EPI_ACCUMULATE) x += *c;
                *c = x;
                rmax[i][r] = fmaxf(rmax[i][r], fabsf(x));
            }
        }
    }
    if (G.c_absmax) {                    // a row lives in the 32 lanes of one half-wave
        const int slot = G.c_first + 2 * (n0 / BN) + wn;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m = rmax[i][r];
                m = fmaxf(m, dpp_mov<0xB1>(m));
                m = fmaxf(m, dpp_mov<0x4E>(m));
                m = fmaxf(m, dpp_mov<0x141>(m));
                m = fmaxf(m, dpp_mov<0x140>(m));
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (l31 == 0 && row < G.M) G.c_absmax[(int64_t)row * G.c_parts + slot] = __float_as_uint(m);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Software pipeline: 16-deep stages, two LDS buffers.  While the matrix cores work on stage s from
// buffer s&1, the same wave splits the (already landed) registers of stage s+1 into buffer (s+1)&1 and the global
// loads of stage s+2 are in flight: the split's VALU work hides under the MFMAs of a stage instead of sitting
// between two barriers, and there is ONE barrier per stage.
constexpr int SK = 16;                      // k per stage
constexpr int LDS16 = SK + 8;               // 16-bit elements per LDS row: 48-byte pitch, conflict-free ds_read_b128
constexpr int PLANE16 = BM * LDS16;

// Thread map of an operand that is contiguous along its OUTPUT index (the TN operands, B of NN): a thread loads 2 reduction rows x 4 output
// columns of the 16 x 128 stage.  tid bits: [1:0] and [7:5] = column group mg (4 columns each), [3:2] = q (which 4 of the 16 rows),
// [4] = which 2 of those 4 - so lanes l and l + 16 hold the two halves of a run of 4 consecutive k for the same columns, trade them with one
// v_permlane16_swap per column pair and write 8 bytes (4 k) per row with ds_write_b64: half the LDS stores of the 2-k ds_write_b32 form, and
// no bank conflicts (that form hit every bank twice: 48-byte rows, 4 rows per thread).  The image keeps output row m at m ^ 2 when bit 3 of m
// is set (nk_row): with it the 16 lanes of a store group cover all 32 banks, and every group of a fragment ds_read_b128 still reads a
// permutation of the rows it read before.
__device__ __forceinline__ int nk_kp(int tid) { return ((tid >> 2) & 3) * 2 + ((tid >> 4) & 1); }
__device__ __forceinline__ int nk_mg(int tid) { return (tid & 3) | ((tid >> 5) << 2); }
__device__ __forceinline__ int nk_row(int m) { return m ^ (((m >> 3) & 1) << 1); }

template <int MODE, bool KCONTIG>
struct StageLoader {
    static_assert(MODE == 0 || KCONTIG, "the scaled-fp16 kernel splits only its K-contiguous A operand in the kernel");
    static constexpr int NP = Emu<MODE>::NP;
    float4 r[2];
    int e[2];                               // MODE 1: minus the scale exponent of the two rows this thread stages
    // o0: first row of the tile, o_end: rows of the operand; bits: `parts` partial absmax bit patterns per row
    __device__ __forceinline__ void load_scales(const uint32_t* __restrict__ bits, int parts, int o0, int o_end, int tid) {
        if constexpr (MODE == 1) {
            const int rr = tid >> 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) e[q] = -scale_exponent(row_absmax_bits(bits, parts, min(o0 + rr + 64 * q, o_end - 1)));
        }
    }
    __device__ __forceinline__ void copy_scales(const StageLoader& o) {
        if constexpr (MODE == 1) { e[0] = o.e[0]; e[1] = o.e[1]; }
    }
    __device__ __forceinline__ void load_fast(const float* __restrict__ base, int64_t ld, int o0, int k0, int o_end, int tid) {
        if constexpr (KCONTIG) {
            const int c = tid & 3, rr = tid >> 2;
            const float* p = base + k0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int o = min(o0 + rr + 64 * q, o_end - 1);
                r[q] = *reinterpret_cast<const float4*>(p + (int64_t)o * ld);
            }
        } else {
            const int kr = nk_kp(tid), mg = nk_mg(tid);
            const float* p = base + o0 + 4 * mg + (int64_t)(k0 + 2 * kr) * ld;
            r[0] = *reinterpret_cast<const float4*>(p);
            r[1] = *reinterpret_cast<const float4*>(p + ld);
        }
    }
    __device__ __forceinline__ void load_guarded(const float* __restrict__ base, int64_t ld, int o0, int k0, int o_end, int k_end, int tid) {
        if constexpr (KCONTIG) {
            const int c = tid & 3, rr = tid >> 2;
            const int k = k0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int o = o0 + rr + 64 * q;
                r[q] = (o < o_end && k < k_end) ? load4_guarded_b(base + (int64_t)o * ld + k, k, k_end) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int kr = nk_kp(tid), mg = nk_mg(tid);
            const int o = o0 + 4 * mg;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + 2 * kr + h;
                r[h] = (k < k_end && o < o_end) ? load4_guarded_b(base + (int64_t)k * ld + o, o, o_end) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // split the pair (a, b) [scaled by 2^e in MODE 1] into NP packed planes
    __device__ __forceinline__ void split_pair(float a, float b, int e, uint32_t (&p)[NP]) const {
        if constexpr (MODE == 0) split2(a, b, p[0], p[1], p[2]);
        else split2h(__builtin_ldexpf(a, e), __builtin_ldexpf(b, e), p[0], p[1]);
    }
    __device__ __forceinline__ void store(uint16_t* __restrict__ lds, int tid) const {
        if constexpr (KCONTIG) {
            const int c = tid & 3, rr = tid >> 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t a[NP], b[NP];
                const int eq = (MODE == 1) ? e[q] : 0;
                split_pair(r[q].x, r[q].y, eq, a);
                split_pair(r[q].z, r[q].w, eq, b);
                uint16_t* d = lds + (rr + 64 * q) * LDS16 + 4 * c;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * PLANE16) = make_uint2(a[pl], b[pl]);
            }
        } else {
            const int mg = nk_mg(tid), q = (tid >> 2) & 3, hb = (tid >> 4) & 1;
            const float x[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
            const float y[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
            uint32_t p[4][NP];               // p[i][plane]: the (k, k + 1) pair of column 4 mg + i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                split_pair(x[i], y[i], 0, p[i]);
            }
            // lanes l (hb = 0) and l + 16 (hb = 1) hold k = 4q, 4q + 1 and 4q + 2, 4q + 3 of the same four columns: after the swap the lower one
            // has all four k of columns 0 / 1, the upper one of columns 2 / 3, both in (first, second) register order
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint16_t* d = lds + nk_row(4 * mg + 2 * hb + j) * LDS16 + 4 * q;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const auto w = __builtin_amdgcn_permlane16_swap(p[j][pl], p[2 + j][pl], false, false);
                    *reinterpret_cast<uint2*>(d + pl * PLANE16) = make_uint2(w[0], w[1]);
                }
            }
        }
    }
    __device__ __forceinline__ void add_colsum(float (&cs)[4], float f) const {
        cs[0] = fmaf(f, r[0].x + r[1].x, cs[0]); cs[1] = fmaf(f, r[0].y + r[1].y, cs[1]);
        cs[2] = fmaf(f, r[0].z + r[1].z, cs[2]); cs[3] = fmaf(f, r[0].w + r[1].w, cs[3]);
    }
};

// bf16x6, all three ops (and the weight gradients of the fp16x3 mode)
template <bool A_KC, bool B_KC, bool SPLITK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16x6_kernel(const GemmParams P, float* __restrict__ ws) {
    constexpr int MODE = 0;
    typedef Emu<MODE> E;
    typedef typename E::frag frag;
    constexpr int NP = E::NP;
    constexpr int OPER16 = NP * PLANE16;    // 16-bit elements per operand stage (18,432 B / 12,288 B)
    __shared__ __attribute__((aligned(16))) uint16_t smem[4 * OPER16];    // [stage buffer][A | B][plane][row][k]

    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngroups; ++i) gi = (tile >= P.g[i].tile_start) ? i : gi;
    const GroupDesc& G = P.g[gi];
    int local = tile - G.tile_start;
    int split = 0;
    if (SPLITK) { split = local / G.tiles_mn; local -= split * G.tiles_mn; }
    const int tm = local / G.tiles_n, tn = local - tm * G.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kb = SPLITK ? split * G.kchunk : 0;
    const int ke = SPLITK ? min(G.K, kb + G.kchunk) : G.K;
    const bool avec = G.flags & 1, bvec = G.flags & 2;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 accs[E::NACC][2][2];
#pragma unroll
    for (int c = 0; c < E::NACC; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;
    f32x16 (&acc)[2][2] = accs[0];

    const bool do_colsum = SPLITK && !A_KC && (G.cs_off >= 0) && (tn == 0);
    const float csf = do_colsum ? 1.f : 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    // (an operand staged by the output-contiguous loader keeps its rows permuted: nk_row - bit 3 of the row is bit 3 of l31)
    const int fa = (wm * 64 + (A_KC ? l31 : nk_row(l31))) * LDS16 + 8 * hi;
    const int fb = OPER16 + (wn * 64 + (B_KC ? l31 : nk_row(l31))) * LDS16 + 8 * hi;
    frag fra[NP][2], frb[NP][2];
    auto read_frags = [&](const uint16_t* buf) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fra[pl][i] = *reinterpret_cast<const frag*>(buf + fa + pl * PLANE16 + i * 32 * LDS16);
                frb[pl][i] = *reinterpret_cast<const frag*>(buf + fb + pl * PLANE16 + i * 32 * LDS16);
            }
    };
    auto mfma_stage = [&]() {
#pragma unroll
        for (int t = 0; t < E::NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    accs[E::TC[t]][i][j] = E::mfma(fra[E::TA[t]][i], frb[E::TB[t]][j], accs[E::TC[t]][i][j]);
    };
    auto compute = [&](const uint16_t* buf) { read_frags(buf); mfma_stage(); };
    auto bsel = [&](int k0, int& kloc) -> const float* {
        if (G.bchunk <= 0) { kloc = k0; return G.B; }
        const int w = k0 / G.bchunk;
        kloc = k0 - w * G.bchunk;
        return w == 0 ? G.B : (w == 1 ? G.B1 : G.B2);
    };

    StageLoader<MODE, A_KC> a0;
    StageLoader<MODE, B_KC> b0;

    const bool fast = avec && bvec && (A_KC ? true : (m0 + BM <= G.M)) && (B_KC ? true : (n0 + BN <= G.N));
    const int nst = fast ? (ke - kb) / SK : 0;
    if (nst > 0) {
        StageLoader<MODE, A_KC> a1;
        StageLoader<MODE, B_KC> b1;
        auto fetch = [&](StageLoader<MODE, A_KC>& la, StageLoader<MODE, B_KC>& lb, int s) {
            const int k0 = kb + min(s, nst - 1) * SK;        // past the end: re-load the last stage (never consumed)
            int kl;
            const float* bb = bsel(k0, kl);
            la.load_fast(G.A, G.lda, m0, k0, G.M, tid);
            lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
        };
        // stage s: registers `c*` hold stage s+1 (landed), `n*` are free
        auto body = [&](StageLoader<MODE, A_KC>& ca, StageLoader<MODE, B_KC>& cb, StageLoader<MODE, A_KC>& na, StageLoader<MODE, B_KC>& nb, int s) {
            fetch(na, nb, s + 2);
            __builtin_amdgcn_sched_barrier(0);
            uint16_t* cur = smem + (s & 1) * 2 * OPER16;
            uint16_t* nxt = smem + ((s + 1) & 1) * 2 * OPER16;
            // source order matters: the fragment READS of `cur` come first so that the LDS WRITES into `nxt` (which the
            // compiler must assume may alias) can be scheduled late, between the MFMAs
            read_frags(cur);
            ca.add_colsum(cs, (s + 1 < nst) ? csf : 0.f);
            ca.store(nxt, tid);
            cb.store(nxt + OPER16, tid);
            mfma_stage();
            // issue order: fragment reads, a little split work while they land, then one MFMA per few VALU ops of the split
            // (the matrix core runs 8 passes per MFMA: the VALU work of the next stage rides in its shadow)
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        fetch(a0, b0, 0);
        fetch(a1, b1, 1);
        a0.add_colsum(cs, csf);
        a0.store(smem, tid);
        b0.store(smem + OPER16, tid);
        __syncthreads();
        int s = 0;
        for (; s + 1 < nst; s += 2) {
            body(a1, b1, a0, b0, s);
            body(a0, b0, a1, b1, s + 1);
        }
        if (s < nst) body(a1, b1, a0, b0, s);
    }
    {   // guarded stages (unaligned operands, partial edge tiles of an M/N-contiguous operand, K tail)
        for (int k0 = kb + nst * SK; k0 < ke; k0 += SK) {
            a0.load_guarded(G.A, G.lda, m0, k0, G.M, ke, tid);
            if (G.bchunk > 0) {
                const int w = k0 / G.bchunk, kloc = k0 - w * G.bchunk;
                b0.load_guarded(w == 0 ? G.B : (w == 1 ? G.B1 : G.B2), G.ldb, n0, kloc, G.N, min(G.bchunk, ke - w * G.bchunk), tid);
            } else
                b0.load_guarded(G.B, G.ldb, n0, k0, G.N, ke, tid);
            a0.add_colsum(cs, csf);
            a0.store(smem, tid);
            b0.store(smem + OPER16, tid);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    }

    float* fsm = reinterpret_cast<float*>(smem);
    if (do_colsum) {
        const int kr = nk_kp(tid), mg = nk_mg(tid);
#pragma unroll
        for (int i = 0; i < 4; ++i) fsm[(4 * mg + i) * 8 + kr] = cs[i];
        __syncthreads();
        if (tid < BM && m0 + tid < G.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += fsm[tid * 8 + q];
            ws[G.cs_off + (int64_t)split * G.M + m0 + tid] = t;
        }
        __syncthreads();
    }
    gemm_epilogue<SPLITK>(P, G, ws, fsm, acc, m0 + wm * 64, n0, (m0 + BM <= G.M) && (n0 + BN <= G.N), split, wave, lane);
}

// ---- fp16x3 pre-pass for A: bits of max |x| of every row
struct AbsmaxJob {
    const float* X; int64_t ld; uint32_t* out; int32_t rows, cols; int32_t block_start; int32_t vec;
};
struct AbsmaxParams {
    AbsmaxJob j[WSI_GEMM_MAX_GROUPS];
    int32_t njobs;
    int32_t total_blocks;
};

// out[r] = bits(max_c |X[r][c]|): one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void absmax_rows_kernel(const AbsmaxParams P) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < P.njobs; ++i) ji = ((int)blockIdx.x >= P.j[i].block_start) ? i : ji;
    const AbsmaxJob& J = P.j[ji];
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - J.block_start) * 4 + (threadIdx.x >> 6);
    if (row >= J.rows) return;
    const float* x = J.X + (int64_t)row * J.ld;
    float m = 0.f;
    if (J.vec) {
        for (int c = 4 * lane; c + 3 < J.cols; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(x + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (int c = (J.cols & ~3) + lane; c < J.cols; c += 64) m = fmaxf(m, fabsf(x[c]));
    } else {
        for (int c = lane; c < J.cols; c += 64) m = fmaxf(m, fabsf(x[c]));
    }
    uint32_t b = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, o, 64));
    // (fmaxf drops NaNs: a NaN element keeps its row's finite scale and propagates through the split as NaN)
    if (lane == 0) J.out[row] = b;
}

static inline bool vec_ok16(const void* p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0); }

// ------------------------------------------------------------------------------------------------
// fp16x3, NT / NN: the small operand (the weights) pre-packed in MFMA fragment order.
// With the matrix work halved, a 128x128 loop that splits BOTH operands in the kernel (the structure of gemm_bf16x6_kernel)
// is LDS-bound (1 KB of LDS traffic per 32-cycle MFMA at 128 B/clk per CU).  Here B never touches the LDS: its two fp16 planes are written once per call by pack_b_frag_kernel as
//     [N / 32][ceil(K / 16)][plane][lane 0..63][8 fp16]          (rows padded to the 128-row tile, K to 16, with zeros)
// i.e. the B fragment of a 32x32x16 MFMA is 1 KB of contiguous memory, and every lane loads its 16 bytes straight into the
// fragment registers one stage ahead (the weights are a few MB: they stay in L2).  A (the activations) is still split in
// the kernel and staged through the LDS: half the LDS traffic, half the split work per MFMA.
struct PackJob {
    const float* B[3]; int64_t ld; uint32_t* bits; uint16_t* out;
    int32_t N, K, bchunk, kc, KB, rows, block_start, vec;   // vec: every matrix 16-byte loadable
};
struct PackParams {
    PackJob j[WSI_GEMM_MAX_GROUPS];
    int32_t njobs, total_blocks;
};

// One workgroup per PR = 8 output columns n of B: absmax of each of them over the reduction
// (-> bits[n], the scale the GEMM undoes), then the two scaled fp16 planes in fragment order.  The weights are read twice
// (L2).  A thread owns groups of 8 consecutive k ("kk"):
//   B[N, K] (NT): thread (nl = t / 32, q = t % 32) walks kk = q, q + 32, ... of row nl          (two 16-byte loads per group)
//   B[K, N] (NN): thread (n4 = t % 2, q = t / 2) walks kk = q, q + 128, ... of columns 4 n4 .. 4 n4 + 3  (8 16-byte loads)
// so every load is a 16-byte access, a 1536 x 512 weight takes 2 - 6 trips per pass, and it is spread over 64 workgroups
// (a 32-column slab per workgroup left 16 of them crawling through it: 16 us per launch, on every projection's critical path).
constexpr int PR = 8;
template <bool KC>
__device__ __forceinline__ void pack_b_block(const PackJob& J, int nb, float (*sm)[32]) {
    const int t = (int)threadIdx.x;
    constexpr int NC = KC ? 1 : 4;                   // columns per thread
    constexpr int QS = KC ? 256 / PR : 256 / (PR / 4);   // threads along k
    const int nl = KC ? t / QS : 4 * (t % (PR / 4));
    const int q = KC ? t % QS : t / (PR / 4);
    const int n = nb * PR + nl;
    const int K8 = J.KB * 2;
    const bool whole = J.vec && (n + NC <= J.N);
    // w[c][i]: element k = kk*8 + i of column n + c
    auto load = [&](int kk, float (&w)[NC][8]) {
        const int k0 = kk * 8;
        const int ch = J.bchunk > 0 ? k0 / J.bchunk : 0;     // (a group of 8 never straddles chunks: b_chunk % 32 == 0)
        const float* b = J.B[ch];
        const int kl = k0 - ch * J.bchunk;
        if (whole && k0 + 8 <= J.K) {
            if constexpr (KC) {
                const float4 x = *reinterpret_cast<const float4*>(b + (int64_t)n * J.ld + kl);
                const float4 y = *reinterpret_cast<const float4*>(b + (int64_t)n * J.ld + kl + 4);
                w[0][0] = x.x; w[0][1] = x.y; w[0][2] = x.z; w[0][3] = x.w; w[0][4] = y.x; w[0][5] = y.y; w[0][6] = y.z; w[0][7] = y.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 x = *reinterpret_cast<const float4*>(b + (int64_t)(kl + i) * J.ld + n);
                    w[0][i] = x.x; w[1][i] = x.y; w[2][i] = x.z; w[3][i] = x.w;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = (n + c < J.N) && (k0 + i < J.K);
                    w[c][i] = ok ? (KC ? b[(int64_t)(n + c) * J.ld + kl + i] : b[(int64_t)(kl + i) * J.ld + n + c]) : 0.f;
                }
        }
    };
    float m[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) m[c] = 0.f;
#pragma unroll 2
    for (int kk = q; kk < K8; kk += QS) {
        float w[NC][8];
        load(kk, w);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) m[c] = fmaxf(m[c], fabsf(w[c][i]));
    }
    // the threads along k of a column combine through LDS: max-accumulate the (non-negative) float bits into 8 rows
    (&sm[0][0])[t] = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) atomicMax(reinterpret_cast<uint32_t*>(&sm[q & 7][nl + c]), __float_as_uint(m[c]));
    __syncthreads();
    int e[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) x = fmaxf(x, sm[i][nl + c]);
        const uint32_t bits = __float_as_uint(x);
        if (q == 0 && n + c < J.N) J.bits[n + c] = bits;
        e[c] = -scale_exponent(bits);
    }
#pragma unroll 2
    for (int kk = q; kk < K8; kk += QS) {
        float w[NC][8];
        load(kk, w);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split2h(__builtin_ldexpf(w[c][2 * i], e[c]), __builtin_ldexpf(w[c][2 * i + 1], e[c]), h[i], l[i]);
            const int lane = ((n + c) & 31) + 32 * (kk & 1);
            uint16_t* o = J.out + ((size_t)(((n + c) >> 5) * J.KB + (kk >> 1)) * 2) * 512 + lane * 8;
            *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(o + 512) = make_uint4(l[0], l[1], l[2], l[3]);
        }
    }
}

__global__ __launch_bounds__(256) void pack_b_frag_kernel(const PackParams P) {
    __shared__ float sm[8][32];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < P.njobs; ++i) ji = ((int)blockIdx.x >= P.j[i].block_start) ? i : ji;
    const PackJob& J = P.j[ji];
    const int nb = (int)blockIdx.x - J.block_start;
    if (J.kc) pack_b_block<true>(J, nb, sm);
    else pack_b_block<false>(J, nb, sm);
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_fp16x3w_kernel(const GemmParams P, float* __restrict__ ws) {
    typedef Emu<1> E;
    typedef f16x8 frag;
    constexpr int OPER16 = 2 * PLANE16;              // the two A planes of one stage (12,288 B)
    // two A stages (24,576 B); the epilogue stages 32 KB of C through it and keeps 1 KB of scale exponents behind that
    __shared__ __attribute__((aligned(16))) uint16_t smem[(4 * 32 * 64 * 4 + 1024) / 2];

    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngroups; ++i) gi = (tile >= P.g[i].tile_start) ? i : gi;
    const GroupDesc& G = P.g[gi];
    const int local = tile - G.tile_start;
    const int tm = local / G.tiles_n, tn = local - tm * G.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 accs[2][2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.f;

    const uint32_t* abits = G.a_absmax ? G.a_absmax : reinterpret_cast<const uint32_t*>(ws) + G.ea_off;
    const uint32_t* bbits = G.b_bits ? G.b_bits : reinterpret_cast<const uint32_t*>(ws) + G.eb_off;
    const int KB = (G.K + 15) >> 4;
    const uint16_t* bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        bj[j] = reinterpret_cast<const uint16_t*>(G.B) + ((size_t)((n0 >> 5) + wn * 2 + j) * KB * 2) * 512 + lane * 8;
    auto load_b = [&](frag (&fb)[2][2], int kb) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[pl][j] = *reinterpret_cast<const frag*>(bj[j] + (size_t)kb * 1024 + pl * 512);
    };
    const int fa = (wm * 64 + l31) * LDS16 + 8 * hi;
    frag fra[2][2];
    auto read_a = [&](const uint16_t* buf) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) fra[pl][i] = *reinterpret_cast<const frag*>(buf + fa + pl * PLANE16 + i * 32 * LDS16);
    };
    auto mfma_stage = [&](const frag (&fb)[2][2]) {
#pragma unroll
        for (int t = 0; t < E::NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    accs[E::TC[t]][i][j] = E::mfma(fra[E::TA[t]][i], fb[E::TB[t]][j], accs[E::TC[t]][i][j]);
    };

    StageLoader<1, true> a0;
    const int aparts = G.a_absmax ? G.a_parts : 1;
    a0.load_scales(abits, aparts, m0, G.M, tid);
    frag fb0[2][2], fb1[2][2];
    const int nst = (G.flags & 1) ? G.K / SK : 0;
    if (nst > 0) {
        StageLoader<1, true> a1;
        a1.copy_scales(a0);
        // stage s: `ca` holds A of stage s+1 (landed), `na` is free and receives stage s+2; `fbc` holds B of stage s, `fbn`
        // receives stage s+1.  Deeper prefetch was measured and dropped: a third A register set (two stages of lead, 256 VGPRs)
        // is no faster (A/B on separate boxes: within their 1.5 % spread), a third B fragment set spills.
        auto body = [&](StageLoader<1, true>& ca, StageLoader<1, true>& na, const frag (&fbc)[2][2], frag (&fbn)[2][2], int s) {
            na.load_fast(G.A, G.lda, m0, min(s + 2, nst - 1) * SK, G.M, tid);     // past the end: re-load the last stage (never consumed)
            load_b(fbn, min(s + 1, nst - 1));
            __builtin_amdgcn_sched_barrier(0);
            uint16_t* cur = smem + (s & 1) * OPER16;
            uint16_t* nxt = smem + ((s + 1) & 1) * OPER16;
            read_a(cur);
            ca.store(nxt, tid);
            mfma_stage(fbc);
            // (no sched_group_barrier interleave pattern here: 0 / 3 / 4 / 5 / 6 VALU per MFMA all measure within 0.3 % on one box)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        a0.load_fast(G.A, G.lda, m0, 0, G.M, tid);
        a1.load_fast(G.A, G.lda, m0, min(1, nst - 1) * SK, G.M, tid);
        load_b(fb0, 0);
        a0.store(smem, tid);
        __syncthreads();
        int s = 0;
        for (; s + 1 < nst; s += 2) {
            body(a1, a0, fb0, fb1, s);
            body(a0, a1, fb1, fb0, s + 1);
        }
        if (s < nst) body(a1, a0, fb0, fb1, s);
    }
    // guarded stages: an A that is not 16-byte loadable, the K tail (B's planes are zero-padded to a multiple of 16)
    for (int k0 = nst * SK; k0 < G.K; k0 += SK) {
        a0.load_guarded(G.A, G.lda, m0, k0, G.M, G.K, tid);
        load_b(fb0, k0 / SK);
        a0.store(smem, tid);
        __syncthreads();
        read_a(smem);
        mfma_stage(fb0);
        __syncthreads();
    }

    float* fsm = reinterpret_cast<float*>(smem);
    {   // undo the operand scales: 2^(e_a[row] + e_b[col]), exact
        int* se = reinterpret_cast<int*>(fsm + 4 * 32 * 64);
        se[tid] = (tid < BM) ? scale_exponent(row_absmax_bits(abits, aparts, min(m0 + tid, G.M - 1))) : scale_exponent(bbits[min(n0 + tid - BM, G.N - 1)]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ec = se[BM + wn * 64 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accs[0][i][j][r] = __builtin_ldexpf(fmaf(accs[1][i][j][r], 1.f / LO_SCALE, accs[0][i][j][r]),
                                                        ec + se[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);
        }
    }
    gemm_epilogue<false>(P, G, ws, fsm, accs[0], m0 + wm * 64, n0, (m0 + BM <= G.M) && (n0 + BN <= G.N), 0, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// fp16x3, NT / NN, both operands staged by LDS-DMA (gemm_fp16x3g_kernel; round 3).  Same arithmetic as gemm_fp16x3w_kernel, bit
// for bit (same split, same products in the same order per 16-deep k-step), different data path:
//   * wave layout 4 x 1: wave w owns rows [32 w, 32 w + 32) of the 128 x 128 tile and ALL 128 columns (acc 2 sets x 4 tiles).
//     A is needed by exactly one wave, so it goes to the LDS as RAW fp32 (global_load_lds_dwordx4: no staging registers, no
//     ds_write, no split before the barrier) and is split in registers after the fragment read - no redundancy, 8 values per
//     lane and k-step;
//   * B (the packed fp16 planes of pack_b_frag_kernel, 1 KB per fragment) is copied verbatim by LDS-DMA and read by all four
//     waves as conflict-free ds_read_b128 (256 B/clk/CU): the vector L1 sees every operand byte ONCE per workgroup (32 KB per
//     32-deep stage and 24 MFMAs per wave, where the register-fragment kernel above pulls 48 KB through it);
//   * the A image is lane-linear (DMA: base + 16 lane), rows of 128 B = one cache line per 8 lanes; the 16-byte column of a row
//     is XOR-swizzled with (row >> 1) & 7 on the SOURCE address and on the fragment read (same involution), which makes the
//     two ds_read_b128 of an A fragment conflict-free in every 16-lane group;
//   * two 32 KB stage buffers, one barrier per stage: the DMA of stage s+2 is issued right after the barrier that ends the
//     reads of stage s and has a whole stage of MFMAs to land; fragments are double-buffered in registers across the barrier.
// Requires K % 32 == 0, 16-byte loadable A and byte offsets below 2^31 within a group's A (the host falls back to the kernel
// above otherwise).
constexpr int GK = 32;                            // k per stage
constexpr int G_A_BYTES = BM * GK * 4;            // raw fp32 A tile: 16 KB
constexpr int G_B_BYTES = BN * GK * 2 * 2;        // two fp16 planes of the B tile: 16 KB
constexpr int G_STAGE = G_A_BYTES + G_B_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}

// epilogue of one 32 x 64 block (two accumulator tiles side by side) through the wave's own 8 KB of LDS: 16-byte rows out
// 16-byte load of data this launch reads exactly once (residual / mask / old C tiles): non-temporal, so that it does not displace the
// A panels and weights the other workgroups of the XCD stream from the L2 (same reasoning as the non-temporal C stores below)
__device__ __forceinline__ float4 ld_stream16(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

// STATS: also leave this wave's column statistics of the block (absmax and sum of the FINAL values over its 32 rows) in `st` (LDS: [2][64]
// floats of this wave: maxima, then sums, at the block's 64 columns) - the part of wsi_gemm_group_t.c_colmax / c_colsum one wave sees
template <bool STATS>
__device__ __forceinline__ void epilogue32x64_vec(const GemmParams& P, const GroupDesc& G, float* wbuf, const f32x16& t0, const f32x16& t1,
                                                  int row0, int col0, int slot, int lane, float gate_s, float r_scale, float* st, const float4 (&rv)[8]) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int epi = P.epilogue;
    WSI_DROP_SEED(G, epi);
    const int rr0 = lane >> 4, c4 = (lane & 15) * 4;
    const int col = col0 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((epi & WSI_EPI_BIAS) && G.bias) bv = make_float4(G.bias[col], G.bias[col + 1], G.bias[col + 2], G.bias[col + 3]);
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
    // (rv: the residual rows of the block - gated skip, dX + (1 - s) g_out - requested by the caller before anything else of the epilogue)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        wbuf[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + l31] = t0[r];
        wbuf[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + 32 + l31] = t1[r];
    }
    // (the buffer is this wave's own and a wave's DS operations execute in order: no barrier)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rr = q * 4 + rr0;
        const int row = row0 + rr;
        float4 x = *reinterpret_cast<const float4*>(wbuf + rr * 64 + c4);
        float* c = G.C + (int64_t)row * G.ldc + col;
        x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w;
        if (epi & WSI_EPI_GELU) { x.x = gelu_erf(x.x); x.y = gelu_erf(x.y); x.z = gelu_erf(x.z); x.w = gelu_erf(x.w); }
        if (epi & WSI_EPI_MUL_M) {
            const float4 mv = ld_stream16(G.Mm + (int64_t)row * G.ldm + col);
            x.x *= mv.x; x.y *= mv.y; x.z *= mv.z; x.w *= mv.w;
        }
        if (epi & WSI_EPI_DROPOUT) {
            const float4 mv = WSI_DROP4(G, row, col);
            x.x *= mv.x; x.y *= mv.y; x.z *= mv.z; x.w *= mv.w;
        }
        if (epi & WSI_EPI_SCALE_GATE) { x.x *= gate_s; x.y *= gate_s; x.z *= gate_s; x.w *= gate_s; }
        if (epi & WSI_EPI_ADD_R) {
            x.x = fmaf(r_scale, rv[q].x, x.x); x.y = fmaf(r_scale, rv[q].y, x.y);
            x.z = fmaf(r_scale, rv[q].z, x.z); x.w = fmaf(r_scale, rv[q].w, x.w);
        }
        if (epi & WSI_EPI_ACCUMULATE) {
            const float4 o = ld_stream16(c);
            x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
        }
        if (G.c_absmax) {
            float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
            m = fmaxf(m, dpp_mov<0xB1>(m));
            m = fmaxf(m, dpp_mov<0x4E>(m));
            m = fmaxf(m, dpp_mov<0x141>(m));
            m = fmaxf(m, dpp_mov<0x140>(m));
            if ((lane & 15) == 0) G.c_absmax[(int64_t)row * G.c_parts + slot] = __float_as_uint(m);
        }
        if constexpr (STATS) {
            cm.x = fmaxf(cm.x, fabsf(x.x)); cm.y = fmaxf(cm.y, fabsf(x.y)); cm.z = fmaxf(cm.z, fabsf(x.z)); cm.w = fmaxf(cm.w, fabsf(x.w));
            cs.x += x.x; cs.y += x.y; cs.z += x.z; cs.w += x.w;
        }
        // non-temporal: the tile is not read again by this launch, and written through the default policy it displaces the A panels
        // and the weights from the L2 the other workgroups of the XCD stream them from (+10 % on the K = 512 launches, same-box A/B)
        if (P.plain_stores) *reinterpret_cast<float4*>(c) = x;
        else { const f32x4 xv = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(xv, reinterpret_cast<f32x4*>(c)); }
    }
    if constexpr (STATS) {
        // the four lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same four columns (rows rr0, rr0 + 4, ...): a fixed butterfly
        float v[8] = {cm.x, cm.y, cm.z, cm.w, cs.x, cs.y, cs.z, cs.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(v[i], __shfl_xor(v[i], 16, 64));
            v[i] = fmaxf(v[i], __shfl_xor(v[i], 32, 64));
            v[4 + i] += __shfl_xor(v[4 + i], 16, 64);
            v[4 + i] += __shfl_xor(v[4 + i], 32, 64);
        }
        if (lane < 16) {
            *reinterpret_cast<float4*>(st + c4) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(st + 64 + c4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// the same block with every access guarded (edge tiles, C / R / Mm not 16-byte accessible)
template <bool STATS>
__device__ __forceinline__ void epilogue32x64_guarded(const GemmParams& P, const GroupDesc& G, const f32x16& t0, const f32x16& t1,
                                                      int row0, int col0, int slot, int lane, float gate_s, float r_scale, float* st) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int epi = P.epilogue;
    WSI_DROP_SEED(G, epi);
    float rmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rmax[r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int col = col0 + jj * 32 + l31;
        const bool colok = col < G.N;
        float bv = 0.f;
        if ((epi & WSI_EPI_BIAS) && G.bias && colok) bv = G.bias[col];
        float cmx = 0.f, csm = 0.f;           // this lane's column over its 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (!(colok && row < G.M)) continue;
            float x = (jj ? t1[r] : t0[r]) + bv;
            if (epi & WSI_EPI_GELU) x = gelu_erf(x);
            if (epi & WSI_EPI_MUL_M) x *= G.Mm[(int64_t)row * G.ldm + col];
            if (epi & WSI_EPI_DROPOUT) x *= WSI_DROP1(G, row, col);
            if (epi & WSI_EPI_SCALE_GATE) x *= gate_s;
            if (epi & WSI_EPI_ADD_R) x = fmaf(r_scale, G.R[(int64_t)row * G.ldr + col], x);
            float* c = G.C + (int64_t)row * G.ldc + col;
            if (epi & WSI_EPI_ACCUMULATE) x += *c;
            *c = x;
            rmax[r] = fmaxf(rmax[r], fabsf(x));
            if constexpr (STATS) { cmx = fmaxf(cmx, fabsf(x)); csm += x; }
        }
        if constexpr (STATS) {               // the two half-waves hold the same column (rows 4 hi + ...)
            cmx = fmaxf(cmx, __shfl_xor(cmx, 32, 64));
            csm += __shfl_xor(csm, 32, 64);
            if (lane < 32) { st[jj * 32 + l31] = cmx; st[64 + jj * 32 + l31] = csm; }
        }
    }
    if (G.c_absmax) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m = rmax[r];
            m = fmaxf(m, dpp_mov<0xB1>(m));
            m = fmaxf(m, dpp_mov<0x4E>(m));
            m = fmaxf(m, dpp_mov<0x141>(m));
            m = fmaxf(m, dpp_mov<0x140>(m));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (l31 == 0 && row < G.M) G.c_absmax[(int64_t)row * G.c_parts + slot] = __float_as_uint(m);
        }
    }
}

// ds_read_b128 as inline asm (see the kernel for why), byte offset as an immediate
template <int OFF, typename T>
__device__ __forceinline__ void lds_read16(T& d, uint32_t addr) {
    static_assert(sizeof(T) == 16 && OFF >= 0 && OFF < 65536, "one 16-byte LDS read, 16-bit offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// the fragments of one k-step: A split into its two fp16 planes (a0, a1), B's two planes for the four column blocks
struct FragSet {
    f16x8 a0, a1;
    f16x8 b0[4], b1[4];
};
// fragment reads: A raw (two 16-byte halves; the addresses carry buffer, k-step and swizzle), one B plane of k-step KS for the four
// column blocks of the B buffer at byte offset BOFF
__device__ __forceinline__ void read_a(uint32_t a0, uint32_t a1, f32x4& r0, f32x4& r1) {
    lds_read16<0>(r0, a0);
    lds_read16<0>(r1, a1);
}
template <int BOFF, int KS, int PL>
__device__ __forceinline__ void read_b(uint32_t b_addr, f16x8 (&b)[4]) {
    lds_read16<BOFF + 0 * 4096 + (KS * 2 + PL) * 1024>(b[0], b_addr);
    lds_read16<BOFF + 1 * 4096 + (KS * 2 + PL) * 1024>(b[1], b_addr);
    lds_read16<BOFF + 2 * 4096 + (KS * 2 + PL) * 1024>(b[2], b_addr);
    lds_read16<BOFF + 3 * 4096 + (KS * 2 + PL) * 1024>(b[3], b_addr);
}
// wait until at most N DS operations of this wave are outstanding; the registers pass THROUGH the wait (in/out operands),
// so nothing that consumes them can be scheduled above it
#define WSI_WAIT_B(N, b) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory")
#define WSI_WAIT_A(N, r0, r1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(r0), "+v"(r1) :: "memory")

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_fp16x3g_kernel(const GemmParams P, float* __restrict__ ws) {
    typedef f16x8 frag;
    // the two B buffers (two fp16 planes, 16 KB each: column block j at j * 4 KB as [k-step][plane][lane][8]; their read offsets are
    // instruction immediates), then the two A buffers (raw fp32, 16 KB each: wave w's rows at w * 4 KB).  The epilogue stages C
    // through the first 32 KB and keeps 1 KB of scale exponents behind that; ONE __shared__ object
    constexpr int B_BASE = 0, A_BASE = 2 * G_B_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[A_BASE + 2 * G_A_BYTES + 1024];
    int* se = reinterpret_cast<int*>(smem + A_BASE + 2 * G_A_BYTES);

    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngroups; ++i) gi = (tile >= P.g[i].tile_start) ? i : gi;
    const GroupDesc& G = P.g[gi];
    const int local = tile - G.tile_start;
    const int tm = local / G.tiles_n, tn = local - tm * G.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 acc0[4], acc1[4];            // [0]: x0 y0, [1]: 2^11 (x0 y1 + x1 y0), column blocks j = 0..3 of the wave's 32 rows
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }

    const uint32_t* abits = G.a_absmax ? G.a_absmax : reinterpret_cast<const uint32_t*>(ws) + G.ea_off;
    const uint32_t* bbits = G.b_bits ? G.b_bits : reinterpret_cast<const uint32_t*>(ws) + G.eb_off;
    const int aparts = G.a_absmax ? G.a_parts : 1;
    const int KB = G.K >> 4;
    const int nst = G.K / GK;

    // The scale words this workgroup needs (this lane's row for the split; all 128 rows and 128 columns of the tile for the
    // epilogue) are REQUESTED here, ahead of the first DMA, and consumed behind it: one memory latency for both, and no plain
    // load left for the epilogue to wait on (it used to open with these loads: ~3k cycles of every wave's ~54k).
    const uint32_t ea_bits = row_absmax_bits(abits, aparts, min(m0 + 32 * wave + l31, G.M - 1));
    const uint32_t se_bits = (tid < BM) ? row_absmax_bits(abits, aparts, min(m0 + tid, G.M - 1)) : bbits[min(n0 + tid - BM, G.N - 1)];
    int ea = 0;                          // minus the scale exponent of this lane's row
    // DMA sources.  A piece p of this wave: rows 32 w + 8 p + (lane >> 3), 16-byte column (lane & 7) ^ swizzle(row); the four
    // pointers walk along K.  B: this wave copies column block j = wave: 4 KB per stage, contiguous in the packed image (one
    // pointer, the pieces are instruction offsets on both sides of the copy)
    const char* a_src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = 32 * wave + 8 * p + (lane >> 3);
        const int kq = (lane & 7) ^ ((r >> 1) & 7);
        a_src[p] = reinterpret_cast<const char*>(G.A) + (size_t)min(m0 + r, G.M - 1) * (size_t)(G.lda * 4) + 16 * kq;
    }
    const char* b_src = reinterpret_cast<const char*>(G.B) + ((size_t)((n0 >> 5) + wave) * KB) * 2048 + 16 * lane;
    // pieces of the stage the pointers stand at -> buffer BUF (a piece = 1 KB = one wave-wide 16-byte DMA)
    auto dma_b = [&](int buf, auto qc) {
        constexpr int Q = decltype(qc)::value;
        __builtin_amdgcn_global_load_lds((glb_void*)b_src, (lds_void*)(smem + B_BASE + buf * G_B_BYTES + wave * 4096), 16, Q * 1024, 0);
    };
    auto dma_a = [&](int buf, auto pc) {
        constexpr int Pp = decltype(pc)::value;
        glds16(a_src[Pp], smem + A_BASE + buf * G_A_BYTES + wave * 4096 + Pp * 1024);
    };
    auto dma_next = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) a_src[p] += GK * 4;
        b_src += 4096;
    };
    std::integral_constant<int, 0> i0;
    std::integral_constant<int, 1> i1;
    std::integral_constant<int, 2> i2;
    std::integral_constant<int, 3> i3;
    // Fragment reads are inline asm on purpose: hipcc's wait-count pass treats an LDS-DMA in flight as a pending write to ANY LDS
    // address and puts s_waitcnt vmcnt(0) in front of the next ds_read it can see - the prefetch would be drained before the
    // first fragment of the next stage is read.  The asm reads are invisible to that pass; their results are tied to hand-placed
    // s_waitcnt lgkmcnt(N) by data dependence (WSI_WAIT_*: the fragment registers are in/out operands of the wait, so no
    // consumer can be scheduled above it).
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    const int sw = (l31 >> 1) & 7;
    uint32_t a_addr[2][2];               // [k-step][half]: this lane's two 16-byte columns of its row, in A buffer 0
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) a_addr[ks][h] = lds0 + A_BASE + (32 * wave + l31) * 128 + 16 * ((4 * ks + 2 * hi + h) ^ sw);
    const uint32_t b_addr = lds0 + 16 * lane;
    // split of one pair of a row's values (the arithmetic of split2h on the row-scaled values)
    float es = 1.f;                      // 2^ea
    auto split_pair = [&](float xa, float xb, uint32_t& h, uint32_t& l) {
        split_scaled<2>(xa, xb, es, es, h, l);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto pack4 = [](const uint32_t (&w)[4]) { const u32x4 v = {w[0], w[1], w[2], w[3]}; return __builtin_bit_cast(frag, v); };

    if (nst > 0) {
        dma_b(0, i0); dma_b(0, i1); dma_b(0, i2); dma_b(0, i3);
        dma_a(0, i0); dma_a(0, i1); dma_a(0, i2); dma_a(0, i3);
        ea = -scale_exponent(ea_bits);
        es = __builtin_ldexpf(1.f, ea);
        se[tid] = scale_exponent(se_bits);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage 0 has landed (explicit: see the stage barrier below)
        __syncthreads();
        if (nst > 1) {
            dma_next();
            dma_b(1, i0); dma_b(1, i1); dma_b(1, i2); dma_b(1, i3);
            dma_a(1, i0); dma_a(1, i1); dma_a(1, i2); dma_a(1, i3);
        }
        FragSet X, Y;
        {   // k-step 0 of stage 0
            f32x4 r0, r1;
            read_a(a_addr[0][0], a_addr[0][1], r0, r1);
            read_b<B_BASE, 0, 1>(b_addr, X.b1);
            read_b<B_BASE, 0, 0>(b_addr, X.b0);
            WSI_WAIT_A(8, r0, r1);
            uint32_t hv[4], lv[4];
            split_pair(r0[0], r0[1], hv[0], lv[0]); split_pair(r0[2], r0[3], hv[1], lv[1]);
            split_pair(r1[0], r1[1], hv[2], lv[2]); split_pair(r1[2], r1[3], hv[3], lv[3]);
            X.a0 = pack4(hv); X.a1 = pack4(lv);
            WSI_WAIT_B(4, X.b1);
        }
        // One k-step (16 deep) of stage s from the set `c`; the fragments of the NEXT k-step (ks 1 of the same buffers, or ks 0 of
        // the other ones) are requested as this one's registers fall free.  On entry: c.a0 / c.a1 / c.b1 ready, c.b0 requested last
        // (4 DS reads outstanding: the x0 y1 products cover their latency).  Products in the order of gemm_fp16x3w_kernel per
        // k-step: x0 y1, x1 y0 (-> acc1), x0 y0 (-> acc0).  The instruction stream is laid out by hand, ONE product per slot with
        // its share of the other work behind it (SLOT = sched_barrier: hipcc may order inside a slot, not across): a wave that
        // issues its twelve products back to back and then ~80 other instructions leaves the matrix pipe to its one co-resident
        // wave for hundreds of cycles at a time (measured: two workgroups per CU were only 1.36x one).
#define SLOT __builtin_amdgcn_sched_barrier(0)
        auto kstep = [&](FragSet& c, FragSet& n, int s, auto bufc, auto ksc) {
            constexpr int BUF = decltype(bufc)::value, KS = decltype(ksc)::value;
            constexpr int NBUF = KS ? (BUF ^ 1) : BUF;
            constexpr int NBOFF = B_BASE + NBUF * G_B_BYTES, NKS = KS ^ 1;
            constexpr uint32_t NAOFF = NBUF * G_A_BYTES;
            const bool more = KS ? (s + 1 < nst) : true;      // is there a next k-step
            const bool refill = KS && (s + 2 < nst);  // is there a stage s+2 to request into the buffers of stage s
            f32x4 r0, r1;
            uint32_t hv[4], lv[4];
            auto mf = [&](f32x16& acc, const frag& a, const frag& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); };
            auto rb = [&](frag& d, const frag&, auto jc, auto plc) {
                constexpr int J = decltype(jc)::value, PL = decltype(plc)::value;
                lds_read16<NBOFF + J * 4096 + (NKS * 2 + PL) * 1024>(d, b_addr);
            };
            if (!KS) {
                // ---- first k-step of a stage: everything it reads next is in the same buffers
                mf(acc1[0], c.a0, c.b1[0]); read_a(a_addr[1][0] + NAOFF, a_addr[1][1] + NAOFF, r0, r1); SLOT;
                mf(acc1[1], c.a0, c.b1[1]); SLOT;
                mf(acc1[2], c.a0, c.b1[2]); rb(n.b1[0], c.b1[0], i0, i1); rb(n.b1[1], c.b1[1], i1, i1); SLOT;
                mf(acc1[3], c.a0, c.b1[3]); rb(n.b1[2], c.b1[2], i2, i1); rb(n.b1[3], c.b1[3], i3, i1); SLOT;
                WSI_WAIT_B(4, c.b0);     // ten reads outstanding, the oldest six are c.b0 and the next A: both have landed
                WSI_WAIT_A(4, r0, r1);
                mf(acc1[0], c.a1, c.b0[0]); split_pair(r0[0], r0[1], hv[0], lv[0]); SLOT;
                mf(acc1[1], c.a1, c.b0[1]); split_pair(r0[2], r0[3], hv[1], lv[1]); SLOT;
                mf(acc1[2], c.a1, c.b0[2]); split_pair(r1[0], r1[1], hv[2], lv[2]); SLOT;
                mf(acc1[3], c.a1, c.b0[3]); split_pair(r1[2], r1[3], hv[3], lv[3]); SLOT;
                n.a0 = pack4(hv); n.a1 = pack4(lv);
                mf(acc0[0], c.a0, c.b0[0]); SLOT;
                mf(acc0[1], c.a0, c.b0[1]); SLOT;
                mf(acc0[2], c.a0, c.b0[2]); SLOT;
                mf(acc0[3], c.a0, c.b0[3]); SLOT;
                rb(n.b0[0], c.b0[0], i0, i0); rb(n.b0[1], c.b0[1], i1, i0); rb(n.b0[2], c.b0[2], i2, i0); rb(n.b0[3], c.b0[3], i3, i0);
                WSI_WAIT_B(4, n.b1);
            } else {
                // ---- second k-step: the stage's barrier sits behind its first four products; past it stage s+1 has landed for every
                // wave (its DMA was issued a stage ago: __syncthreads() waits vmcnt(0)) and nobody reads this stage's buffers any
                // more, so stage s+2 is requested into them, two pieces per slot
                mf(acc1[0], c.a0, c.b1[0]); SLOT;
                mf(acc1[1], c.a0, c.b1[1]); SLOT;
                mf(acc1[2], c.a0, c.b1[2]); SLOT;
                mf(acc1[3], c.a0, c.b1[3]); if (refill) dma_next(); SLOT;
                WSI_WAIT_B(0, c.b0);     // every read of this stage's buffers by this wave is complete
                if (more) {
                    // The DMA of stage s+1 (requested a stage ago) is waited for EXPLICITLY.  hipcc derives a vmcnt(0) for
                    // __syncthreads() from the LDS-DMA it sees in flight - but that is an inference of its wait-count pass, and it was
                    // seen to fail: with a tile loop wrapped around this kernel (a persistent variant, measured ~10 % slower and
                    // dropped) one of the two unrolled stage bodies got a bare lgkmcnt(0) + s_barrier, and rows of a tile were read
                    // before their 8-row piece had landed (~1 launch in 3).
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    read_a(a_addr[0][0] + NAOFF, a_addr[0][1] + NAOFF, r0, r1);
                }
                SLOT;
                mf(acc1[0], c.a1, c.b0[0]); if (refill) { dma_b(BUF, i0); dma_b(BUF, i1); } SLOT;
                mf(acc1[1], c.a1, c.b0[1]); if (refill) { dma_b(BUF, i2); dma_b(BUF, i3); } SLOT;
                mf(acc1[2], c.a1, c.b0[2]); if (more) { rb(n.b1[0], c.b1[0], i0, i1); rb(n.b1[1], c.b1[1], i1, i1); } SLOT;
                mf(acc1[3], c.a1, c.b0[3]); if (more) { rb(n.b1[2], c.b1[2], i2, i1); rb(n.b1[3], c.b1[3], i3, i1); } SLOT;
                if (more) WSI_WAIT_A(4, r0, r1);
                mf(acc0[0], c.a0, c.b0[0]); if (more) split_pair(r0[0], r0[1], hv[0], lv[0]); if (refill) dma_a(BUF, i0); SLOT;
                mf(acc0[1], c.a0, c.b0[1]); if (more) split_pair(r0[2], r0[3], hv[1], lv[1]); if (refill) dma_a(BUF, i1); SLOT;
                mf(acc0[2], c.a0, c.b0[2]); if (more) split_pair(r1[0], r1[1], hv[2], lv[2]); if (refill) dma_a(BUF, i2); SLOT;
                mf(acc0[3], c.a0, c.b0[3]); if (more) split_pair(r1[2], r1[3], hv[3], lv[3]); if (refill) dma_a(BUF, i3); SLOT;
                if (more) {
                    n.a0 = pack4(hv); n.a1 = pack4(lv);
                    rb(n.b0[0], c.b0[0], i0, i0); rb(n.b0[1], c.b0[1], i1, i0); rb(n.b0[2], c.b0[2], i2, i0); rb(n.b0[3], c.b0[3], i3, i0);
                    WSI_WAIT_B(4, n.b1);
                }
            }
        };
#undef SLOT
        int s = 0;
#pragma unroll 1
        for (; s + 1 < nst; s += 2) {
            kstep(X, Y, s, i0, i0);
            kstep(Y, X, s, i0, i1);
            kstep(X, Y, s + 1, i1, i0);
            kstep(Y, X, s + 1, i1, i1);
        }
        if (s < nst) {
            kstep(X, Y, s, i0, i0);
            kstep(Y, X, s, i0, i1);
        }
    }
    __syncthreads();                     // the stage buffers become the epilogue's staging area

    float* fsm = reinterpret_cast<float*>(smem);
    if (nst <= 0) { se[tid] = scale_exponent(se_bits); __syncthreads(); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ec = se[BM + j * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc0[j][r] = __builtin_ldexpf(fmaf(acc1[j][r], 1.f / LO_SCALE, acc0[j][r]), ec + se[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi]);
    }
    const int epi = P.epilogue;
    float gate_s = 1.f;
    if ((epi & (WSI_EPI_SCALE_GATE | WSI_EPI_R_1MG)) && G.gate) gate_s = 1.f / (1.f + expf(-(*G.gate)));
    const float r_scale = (epi & WSI_EPI_R_1MG) ? (1.f - gate_s) : 1.f;
    const int row0 = m0 + 32 * wave;
#ifdef WSI_ABLATE
    const bool vec = (m0 + BM <= G.M) && (n0 + BN <= G.N) && (G.flags & 4) && !P.ablate_guarded;
#else
    const bool vec = (m0 + BM <= G.M) && (n0 + BN <= G.N) && (G.flags & 4);
#endif
    float* wbuf = fsm + wave * (32 * 64);
    // column statistics of the tile (wsi_gemm_group_t.c_colmax / c_colsum): per wave behind the C staging area ([wave][hc][max | sum][64]),
    // combined over the four waves in wave order (rows ascending: a fixed order) by the first 128 threads
    float* stats = fsm + 4 * (32 * 64);
    const bool want_stats = G.c_colmax != nullptr;
    // (the two column halves written out by hand with constant accumulator indices: a loop the compiler declines to unroll would index acc0
    // dynamically and move the accumulators of the whole kernel to scratch memory)
    // The residual tile (gated skip, dX + (1 - s) g_out: 64 KB per tile, as much as C) is requested in ONE burst - sixteen 16-byte loads per lane, both
    // column halves - in front of the LDS transposition of the accumulators, instead of one row group at a time right in front of its use: the
    // launches with a residual ran at 0.74 of the plain ones (r04: 197 vs 265 TFLOP/s-equivalent on the output projection), 0.86 with the first
    // half requested up front, (tools/epi_probe.py) with both.  The main loop's fragment registers are dead here: 64 of them carry the burst.
    float4 rv[2][8];
    if (vec && (epi & WSI_EPI_ADD_R)) {
        const int rr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int hc = 0; hc < 2; ++hc)
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[hc][q] = ld_stream16(G.R + (int64_t)(row0 + q * 4 + rr0) * G.ldr + n0 + 64 * hc + c4);
    }
    auto half = [&](auto hcc) {
        constexpr int hc = decltype(hcc)::value;
        const int slot = G.c_first + 2 * (n0 / BN) + hc;
        float* st = stats + (wave * 2 + hc) * 128;
        if (want_stats) {
            if (vec) epilogue32x64_vec<true>(P, G, wbuf, acc0[2 * hc], acc0[2 * hc + 1], row0, n0 + 64 * hc, slot, lane, gate_s, r_scale, st, rv[hc]);
            else epilogue32x64_guarded<true>(P, G, acc0[2 * hc], acc0[2 * hc + 1], row0, n0 + 64 * hc, slot, lane, gate_s, r_scale, st);
        } else {
            if (vec) epilogue32x64_vec<false>(P, G, wbuf, acc0[2 * hc], acc0[2 * hc + 1], row0, n0 + 64 * hc, slot, lane, gate_s, r_scale, st, rv[hc]);
            else epilogue32x64_guarded<false>(P, G, acc0[2 * hc], acc0[2 * hc + 1], row0, n0 + 64 * hc, slot, lane, gate_s, r_scale, st);
        }
    };
    half(i0);
    half(i1);
    if (want_stats) {
        __syncthreads();
        if (tid < BN && n0 + tid < G.N) {
            const int hc = tid >> 6, cl = tid & 63;
            float m = 0.f, t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                m = fmaxf(m, stats[(w * 2 + hc) * 128 + cl]);
                t += stats[(w * 2 + hc) * 128 + 64 + cl];
            }
            G.c_colmax[(int64_t)tm * G.c_col_ld + n0 + tid] = __float_as_uint(m);
            if (G.c_colsum) G.c_colsum[(int64_t)tm * G.c_col_ld + n0 + tid] = t;
        }
    }
}

// (the 2 x 2 wave-grid form of this kernel, round 5's negative result - bit-identical, 5-7 % slower - lives in gemm_emu16_ablate.inc: measurement build only)
#ifdef WSI_ABLATE
#include "gemm_emu16_ablate.inc"
#endif

// The fp16x3 pre-pass of a launch: the absmax bits of A per output row (absmax_rows_kernel, unless the caller supplied
// them) and, per distinct B, ONE pack_b_frag_kernel workgroup row that finds the absmax of its 32 output columns and
// writes their planes - every scale word a group uses is written by exactly one of the two (no clearing, no atomics).
// Scale words sit at e_first.. in the workspace, the planes behind all of them; the groups' B pointers are redirected
// to the planes.  Operands shared by several groups (the K, Q and V projections read the same rows of h; one weight
// serves several row ranges) are reduced / packed once.
static void prepare_fp16x3(int op, GemmParams& P, float* ws, int64_t e_first, hipStream_t st) {
    AbsmaxParams R;
    PackParams K;
    R.njobs = K.njobs = 0; R.total_blocks = K.total_blocks = 0;
    struct Seen { const float* X[3]; int64_t ld; int o, k, bchunk; bool kc; int32_t off; uint16_t* planes; };
    Seen seen[2 * WSI_GEMM_MAX_GROUPS];
    int nseen = 0;
    int64_t next = e_first;
    int64_t pnext = e_first;
    for (int i = 0; i < P.ngroups; ++i) pnext += (int64_t)((P.g[i].M + 3) & ~3) + ((P.g[i].N + 3) & ~3);
    auto find = [&](const float* X0, const float* X1, const float* X2, int64_t ld, int o, int k, int bchunk, bool kc, bool planes) -> const Seen* {
        for (int q = 0; q < nseen; ++q) {
            const Seen& s = seen[q];
            if (s.X[0] == X0 && s.X[1] == X1 && s.X[2] == X2 && s.ld == ld && s.o == o && s.k == k && s.bchunk == bchunk && s.kc == kc &&
                (s.planes != nullptr) == planes) return &s;
        }
        return nullptr;
    };
    const bool b_kc = op == WSI_GEMM_NT;            // (TN never comes here: gemm_f32.hip runs it as bf16x6)
    for (int i = 0; i < P.ngroups; ++i) {
        GroupDesc& G = P.g[i];
        if (!G.a_absmax) {                           // rows of A: one wave per row
            const Seen* a = find(G.A, nullptr, nullptr, G.lda, G.M, G.K, 0, true, false);
            if (!a) {
                const int32_t off = (int32_t)next;
                next += (G.M + 3) & ~3;
                if (G.K > 0) {
                    AbsmaxJob& J = R.j[R.njobs++];
                    J.X = G.A; J.ld = G.lda; J.out = reinterpret_cast<uint32_t*>(ws) + off; J.vec = vec_ok16(G.A, G.lda) ? 1 : 0;
                    J.rows = G.M; J.cols = G.K; J.block_start = R.total_blocks;
                    R.total_blocks += (G.M + 3) / 4;
                }
                seen[nseen] = Seen{{G.A, nullptr, nullptr}, G.lda, G.M, G.K, 0, true, off, nullptr};
                a = &seen[nseen++];
            }
            G.ea_off = a->off;
        } else
            G.ea_off = 0;
        if (G.b_bits) {                              // the caller brought B packed (wsi_gemm_group_t.b_packed): scale words, then the planes
            G.eb_off = 0;
            G.B = reinterpret_cast<const float*>(G.b_bits + ((G.N + 3) & ~3));
            continue;
        }
        const float* B1 = G.bchunk > 0 ? G.B1 : nullptr;
        const float* B2 = G.bchunk > 0 ? G.B2 : nullptr;
        const Seen* b = find(G.B, B1, B2, G.ldb, G.N, G.K, G.bchunk, b_kc, true);
        if (!b) {
            const int32_t off = (int32_t)next;
            next += (G.N + 3) & ~3;
            PackJob& J = K.j[K.njobs++];
            J.B[0] = G.B; J.B[1] = B1; J.B[2] = B2; J.ld = G.ldb; J.bits = reinterpret_cast<uint32_t*>(ws) + off;
            J.out = reinterpret_cast<uint16_t*>(ws + pnext);
            J.N = G.N; J.K = G.K; J.bchunk = G.bchunk; J.kc = b_kc ? 1 : 0; J.KB = (G.K + 15) >> 4; J.rows = (G.N + 127) & ~127;
            J.vec = (vec_ok16(G.B, G.ldb) && (!B1 || vec_ok16(B1, G.ldb)) && (!B2 || vec_ok16(B2, G.ldb))) ? 1 : 0;
            J.block_start = K.total_blocks;
            K.total_blocks += J.rows / PR;
            pnext += (int64_t)J.rows * J.KB * 16;
            seen[nseen] = Seen{{G.B, B1, B2}, G.ldb, G.N, G.K, G.bchunk, b_kc, off, J.out};
            b = &seen[nseen++];
        }
        G.eb_off = b->off;
        G.B = reinterpret_cast<const float*>(b->planes);
    }
    if (R.njobs) hipLaunchKernelGGL(absmax_rows_kernel, dim3(R.total_blocks), dim3(256), 0, st, R);
    if (K.njobs) hipLaunchKernelGGL(pack_b_frag_kernel, dim3(K.total_blocks), dim3(256), 0, st, K);
}

// wsi_gemm_pack_b: the packed form of every group's B (b_packed: scale words, then the planes) in ONE launch - what prepare_fp16x3 does per call,
// for weights that change only in the optimizer step
int launch_pack_b(int op, const wsi_gemm_group_t* groups, int32_t ngroups, hipStream_t st) {
    PackParams K;
    K.njobs = 0; K.total_blocks = 0;
    const bool b_kc = op == WSI_GEMM_NT;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.N <= 0 || s.K <= 0) continue;
        PackJob& J = K.j[K.njobs++];
        const float* B1 = s.b_chunk > 0 ? s.B1 : nullptr;
        const float* B2 = s.b_chunk > 0 ? s.B2 : nullptr;
        J.B[0] = s.B; J.B[1] = B1; J.B[2] = B2; J.ld = s.ldb;
        J.bits = reinterpret_cast<uint32_t*>(s.b_packed);
        J.out = reinterpret_cast<uint16_t*>(J.bits + ((s.N + 3) & ~3));
        J.N = s.N; J.K = s.K; J.bchunk = s.b_chunk; J.kc = b_kc ? 1 : 0; J.KB = (s.K + 15) >> 4; J.rows = (s.N + 127) & ~127;
        J.vec = (vec_ok16(s.B, s.ldb) && (!B1 || vec_ok16(B1, s.ldb)) && (!B2 || vec_ok16(B2, s.ldb))) ? 1 : 0;
        J.block_start = K.total_blocks;
        K.total_blocks += J.rows / PR;
    }
    if (K.njobs) hipLaunchKernelGGL(pack_b_frag_kernel, dim3(K.total_blocks), dim3(256), 0, st, K);
    return check_launch("gemm_pack_b");
}

void launch_gemm_bf16x6(int op, const GemmParams& P, int tiles, unsigned lds_pad, float* ws, hipStream_t st) {
    const dim3 g(tiles), b(GEMM_THREADS);
    if (op == WSI_GEMM_TN) hipLaunchKernelGGL((gemm_bf16x6_kernel<false, false, true>), g, b, lds_pad, st, P, ws);
    else if (op == WSI_GEMM_NT) hipLaunchKernelGGL((gemm_bf16x6_kernel<true, true, false>), g, b, lds_pad, st, P, ws);
    else hipLaunchKernelGGL((gemm_bf16x6_kernel<true, false, false>), g, b, lds_pad, st, P, ws);
}

// the LDS-DMA kernel serves launches whose every group has K % 32 == 0, a 16-byte loadable A and 32-bit byte offsets into it
// (-DWSI_ABLATE builds only: WSI_GEMM_F16_KERNEL=w forces the register-fragment kernel, read per call for A/B runs in one process)
static bool fp16x3_dma_ok(const GemmParams& P) {
    const char* v = knob("WSI_GEMM_F16_KERNEL");
    if (v && v[0] == 'w') return false;
    for (int i = 0; i < P.ngroups; ++i) {
        const GroupDesc& G = P.g[i];
        if (G.K <= 0 || G.K % GK != 0 || !(G.flags & 1) || (int64_t)G.M * G.lda * 4 >= ((int64_t)1 << 31)) return false;
    }
    return true;
}

void launch_gemm_fp16x3(int op, GemmParams& P, int tiles, unsigned lds_pad, float* ws, int64_t e_first, int64_t e_words, hipStream_t st) {
    (void)e_words;
    prepare_fp16x3(op, P, ws, e_first, st);
    if (fp16x3_dma_ok(P)) {
        const dim3 g(tiles), b(GEMM_THREADS);
        P.plain_stores = 0;
#ifdef WSI_ABLATE
        { const char* e = knob("WSI_F16G_EPI"); P.ablate_guarded = (e && e[0] == 'g') ? 1 : 0; }
        const char* v = knob("WSI_GEMM_F16_KERNEL");
        if (v && v[0] == 'q') { hipLaunchKernelGGL(gemm_fp16x3q_kernel, g, b, lds_pad, st, P, ws); return; }
        if (v && v[0] == 'p') {              // persistent form: every group needs an even number of stages
            bool ok = true;
            for (int i = 0; i < P.ngroups; ++i) ok = ok && (P.g[i].K / GK) >= 2 && ((P.g[i].K / GK) % 2 == 0);
            if (ok) {
                const int slots = 512;       // 2 workgroups per CU x 256 CUs
                hipLaunchKernelGGL(gemm_fp16x3p_kernel, dim3(tiles < slots ? tiles : slots), b, lds_pad, st, P, ws);
                return;
            }
        }
#endif
        hipLaunchKernelGGL(gemm_fp16x3g_kernel, g, b, lds_pad, st, P, ws);
    }
    else hipLaunchKernelGGL(gemm_fp16x3w_kernel, dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, ws);
}

}  // namespace wsi

// absmax bits of every row of X (one part per row): the WSI_GEMM_FP16X3 scale of an operand that does not change from step to
// step (the input features of a resident graph), taken once by the caller and handed to every projection as a_absmax
extern "C" int wsi_row_absmax(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* out, void* stream) {
    if (rows < 0 || cols < 0) { wsi::set_error("row_absmax: bad shape %d x %d", rows, cols); return WSI_EINVAL; }
    if (rows == 0) return WSI_OK;
    if (!out || (cols > 0 && !x)) { wsi::set_error("row_absmax: null pointer"); return WSI_EINVAL; }
    wsi::AbsmaxParams R;
    R.njobs = 1; R.total_blocks = (rows + 3) / 4;
    R.j[0].X = x; R.j[0].ld = ld; R.j[0].out = out; R.j[0].rows = rows; R.j[0].cols = cols; R.j[0].block_start = 0;
    R.j[0].vec = wsi::vec_ok16(x, ld) ? 1 : 0;
    hipLaunchKernelGGL(wsi::absmax_rows_kernel, dim3(R.total_blocks), dim3(256), 0, (hipStream_t)stream, R);
    return wsi::check_launch("row_absmax");
}
