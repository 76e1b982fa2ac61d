// Pieces shared by the fp32-on-16-bit-matrix-core GEMM kernels (gemm_emu16.hip: NT / NN / bf16x6; gemm_tn16.hip: the scaled-fp16 weight gradients):
// packed conversions, the 3-way bf16 and 2-way fp16 splits, the scale exponent of a row / column from its absmax bits.
#pragma once
#include "gemm_common.h"

namespace wsi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_cvt_pk_bf16_f32 (round to nearest even); low half = first value
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// exact 3-way split of two floats into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void split2(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(sa, sb);
}

// round-to-nearest-even pair -> packed fp16 (low half = first value)
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// 2-way fp16 split of two (already scaled, |x| < 2^15) floats: p0 = RN16(x), p1 = RN16(2^11 (x - p0)).  The residual is
// <= 2^-11 |x|, so 2^11 times it has x's magnitude again: both planes are NORMAL fp16 numbers for every |x| >= 2^-13
// (the matrix cores flush fp16 denormals), i.e. for every element within 2^-28 of its row's largest.
constexpr float LO_SCALE = 2048.f;
__device__ __forceinline__ void split2h(float a, float b, uint32_t& p0, uint32_t& p1) {
    p0 = cvt_pk_f16(a, b);
    const f16x2 h = __builtin_bit_cast(f16x2, p0);
    const f32x2 r = {a - (float)h[0], b - (float)h[1]};     // exact
    const f32x2 k = {LO_SCALE, LO_SCALE};
    const f32x2 q = r * k;
    p1 = cvt_pk_f16(q[0], q[1]);
}
// (Round 2 measured a v_fma_mixlo/mixhi_f16 formulation of the residual SLOWER in the register-fragment kernel, where the compiler places the
// conversions; in the kernels that place them by hand - gemm_tn16_kernel, and since round 5 gemm_fp16x3g_kernel - it is the faster one: split_scaled.)

// Scale and split one pair (the arithmetic of split2h on x s, bit for bit; s = 2^e exactly): p0 = RN16(v), p1 = RN16(2^11 (v - p0)) with v = x s.
// The residual is one v_fma_mix per element - fma(float(p0), -2^11, 2^11 v), exact in fp32 (v - p0 has at most 13 significant bits), rounded once
// to fp16 into its half of the packed register - instead of converting p0 back, subtracting, scaling and converting again: 5 VALU instructions per
// pair where the compiler's own lowering of split2h takes 8.  The loop is instruction-issue bound (24 MFMAs, ~150 other vector instructions per
// stage and SIMD: profiles/r05_tn16_*), so the count is what matters; the same form in gemm_fp16x3g_kernel: profiles/r05_f16g_split.json.  NACC = 1: the low term as it is (no 2^11).
template <int NACC>
__device__ __forceinline__ void split_scaled(float x0, float x1, float s0, float s1, uint32_t& p0, uint32_t& p1) {
    const f32x2 x = {x0, x1}, sc = {s0, s1};
    const f32x2 v = x * sc;
    p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    uint32_t l;
    if constexpr (NACC == 2) {
        const f32x2 w = v * LO_SCALE;
        const float k = -LO_SCALE;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(p0), "v"(k), "v"(w[0]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(p0), "v"(k), "v"(w[1]));
    } else {
        const float k = -1.f;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(p0), "v"(k), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(p0), "v"(k), "v"(v[1]));
    }
    p1 = l;
}


// the per-row scale exponent from the absmax bits the pre-pass left: the largest element lands in [2^14, 2^15)
// (zeros / denormal rows clamp at -100 so that 2^-e stays finite; inf / nan rows give e = 114 and stay non-finite)
__device__ __forceinline__ int scale_exponent(uint32_t absmax_bits) {
    return max((int)((absmax_bits >> 23) & 0xffu) - 141, -100);
}

// a row's absmax bits from `parts` partial maxima (see wsi_gemm_group_t.a_absmax)
__device__ __forceinline__ uint32_t row_absmax_bits(const uint32_t* __restrict__ bits, int parts, int row) {
    const uint32_t* p = bits + (int64_t)row * parts;
    uint32_t b = p[0];
    for (int j = 1; j < parts; ++j) b = max(b, p[j]);
    return b;
}

}  // namespace wsi
