// Shared definitions of the grouped GEMM kernels (exact-fp32 MFMA in gemm_f32.hip, split-bf16 emulation in
// gemm_emu16.hip): tile shape, by-value descriptor tables, XCD-aware tile remap.
#pragma once
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int LD_T = BM + 1;   // k-major LDS row stride when the operand is transposed while staging
constexpr int LD_N = BM;       // ... when it is staged with 16-byte writes

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GroupDesc {
    const float* A; const float* B; float* C;
    const float* bias; const float* R; const float* gate;
    const float* B1; const float* B2;   // NN: further chunks of the reduction dimension
    const float* Mm;                    // MUL_M: element-wise multiplier [M,N] (dropout keep-mask / (1-p))
    const uint32_t* a_absmax;           // fp16x3: caller-provided absmax bits of A's rows, or NULL (then at ws + ea_off)
    uint32_t* c_absmax;                 // fp16x3: [M][c_parts] partial absmax bits of the rows of C (slots c_first + 2 tn + wn), or NULL
    int64_t lda, ldb, ldc, ldr, ldm;
    int64_t cs_off;        // TN with colsum_out: float offset of the [splits][M] column-sum partials in the workspace, else -1
    int64_t ws_off;        // TN: float offset of this group's slabs in the workspace
    int32_t M, N, K;
    int32_t tile_start;    // first logical tile id of the group
    int32_t tiles_n;       // tiles along N
    int32_t tiles_mn;      // tiles_m * tiles_n
    int32_t kchunk;        // TN: rows of the reduction per split (multiple of BK); else K
    int32_t flags;         // bit0: A vector-loadable, bit1: B vector-loadable, bit2: C (and R) take 16-byte accesses
    int32_t bchunk;        // NN with B1/B2: reduction rows per B matrix (multiple of BK), else 0
    int32_t ea_off, eb_off;   // fp16x3: word index in the workspace of the absmax bits of A's rows [M] / B's columns [N] of the output
    int32_t a_parts, c_parts, c_first;   // slots per row of a_absmax / c_absmax, first slot of this group
    int32_t tile_start_h;  // gemm_fp16x3h_kernel (256 x 128 tiles): first logical tile id of the group (set by its launcher)
};

struct GemmParams {
    GroupDesc g[WSI_GEMM_MAX_GROUPS];
    int32_t ngroups;
    int32_t total_tiles;
    int32_t epilogue;
    int32_t plain_stores;  // gemm_fp16x3g_kernel: 1 = default-policy C stores instead of non-temporal ones (WSI_F16G_NT=0, A/B runs)
    int32_t total_tiles_h; // tiles of a gemm_fp16x3h_kernel launch
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// split-K second stage (gemm_f32.hip), shared by every TN kernel: slabs summed in slab order -> deterministic
struct ReduceDesc {
    const float* ws; float* C; const float* gate; int64_t ldc; int32_t M, N, splits; int32_t pad; int64_t start;  // start: first flat element id
    const float* cs_ws; float* cs_out;   // column-sum partials [splits][M] -> cs_out[M] (or NULL)
};
struct ReduceParams {
    ReduceDesc g[WSI_GEMM_MAX_GROUPS];
    int32_t ngroups;
    int32_t epilogue;
    int64_t total;
};
void launch_splitk_reduce(const ReduceParams& RP, hipStream_t st);

// launchers of the split-bf16 kernels (gemm_emu16.hip)
void launch_gemm_bf16x6(int op, const GemmParams& P, int tiles, unsigned lds_pad, float* ws, hipStream_t st);
// fp16x3: absmax pre-pass (+ the small operand packed in MFMA fragment order for NT / NN) into the words
// [e_first, e_first + e_words) of the workspace, then the split-fp16 kernel.  fp16x3_words: what one group needs there.
inline int64_t fp16x3_words(int op, int M, int N, int K) {
    int64_t w = (int64_t)((M + 3) & ~3) + ((N + 3) & ~3);
    if (op != WSI_GEMM_TN) w += (int64_t)((N + 127) & ~127) * ((K + 15) & ~15);     // two fp16 planes of B: 4 bytes per element
    return w;
}
void launch_gemm_fp16x3(int op, GemmParams& P, int tiles, unsigned lds_pad, float* ws, int64_t e_first, int64_t e_words, hipStream_t st);

}  // namespace wsi
