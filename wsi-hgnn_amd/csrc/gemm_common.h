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
    uint32_t drop_seed, drop_thr;      // WSI_EPI_DROPOUT (include/wsi_hgnn.h): the draw, the 16-bit keep threshold
    float drop_scale;
    int32_t drop_row0, drop_col0;      // position of this group's C inside the masked tensor
    uint32_t drop_pairs;               // ceil(columns of the masked tensor / 2): pairs per row in the hash's index space
    const uint32_t* drop_seed_base;    // optional device word added to drop_seed when the kernel runs (hipGraph replays: new masks, same arguments)
    const uint32_t* b_bits;            // fp16x3 NT / NN: the scale words of B's packed planes when the caller brought them (wsi_gemm_group_t.b_packed), else NULL (ws + eb_off)
    uint32_t* c_colmax; float* c_colsum; int64_t c_col_ld;   // gemm_fp16x3g_kernel: per-tile-row partial column statistics of C (wsi_gemm_group_t), or NULL
};

struct GemmParams {
    GroupDesc g[WSI_GEMM_MAX_GROUPS];
    int32_t ngroups;
    int32_t total_tiles;
    int32_t epilogue;
    int32_t plain_stores;  // gemm_fp16x3g_kernel: 1 = default-policy C stores instead of non-temporal ones (WSI_F16G_NT=0, A/B runs)
#ifdef WSI_ABLATE
    int32_t ablate_guarded; // measurement build: force the element-wise epilogue (WSI_F16G_EPI=g)
#endif
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// ---- counter-based dropout (WSI_EPI_DROPOUT; the contract is in include/wsi_hgnn.h): one 32-bit hash decides a pair of columns
__host__ __device__ __forceinline__ uint32_t drop_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t drop_pair_hash(uint32_t row, uint32_t pair, uint32_t pairs_per_row, uint32_t seed) {
    return drop_fmix32((row * pairs_per_row + pair) * 0x9E3779B1u + seed);
}
// factor (scale or 0) of element (row, col) of the masked tensor
__device__ __forceinline__ float drop_factor1(uint32_t row, uint32_t col, uint32_t pairs, uint32_t seed, uint32_t thr, float scale) {
    const uint32_t h = drop_pair_hash(row, col >> 1, pairs, seed);
    return (((col & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr) ? scale : 0.f;
}
// the four factors of columns col .. col + 3 of one row (col % 4 == 0): two hashes
__device__ __forceinline__ float4 drop_factor4(uint32_t row, uint32_t col, uint32_t pairs, uint32_t seed, uint32_t thr, float scale) {
    const uint32_t h0 = drop_pair_hash(row, col >> 1, pairs, seed), h1 = drop_pair_hash(row, (col >> 1) + 1, pairs, seed);
    return make_float4((h0 & 0xffffu) >= thr ? scale : 0.f, (h0 >> 16) >= thr ? scale : 0.f, (h1 & 0xffffu) >= thr ? scale : 0.f, (h1 >> 16) >= thr ? scale : 0.f);
}
// the draw of this launch: the group's seed + the device word, read ONCE per epilogue into `wsi_drop_seed` (WSI_DROP_SEED), which the two macros use
#define WSI_DROP_SEED(G, epi) const uint32_t wsi_drop_seed = (((epi) & WSI_EPI_DROPOUT) && (G).drop_seed_base) ? (G).drop_seed + *(G).drop_seed_base : (G).drop_seed
#define WSI_DROP4(G, row_in_group, col_in_group) drop_factor4((uint32_t)((G).drop_row0 + (row_in_group)), (uint32_t)((G).drop_col0 + (col_in_group)), (G).drop_pairs, wsi_drop_seed, (G).drop_thr, (G).drop_scale)
#define WSI_DROP1(G, row_in_group, col_in_group) drop_factor1((uint32_t)((G).drop_row0 + (row_in_group)), (uint32_t)((G).drop_col0 + (col_in_group)), (G).drop_pairs, wsi_drop_seed, (G).drop_thr, (G).drop_scale)

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
// the packed form of one B operand (wsi_gemm_group_t.b_packed): [pad4(N) scale words | two fp16 planes in MFMA fragment order]
inline int64_t fp16x3_packed_b_bytes(int N, int K) { return ((int64_t)((N + 3) & ~3) + (int64_t)((N + 127) & ~127) * ((K + 15) & ~15)) * 4; }
int launch_pack_b(int op, const wsi_gemm_group_t* groups, int32_t ngroups, hipStream_t st);
// scaled-fp16 weight gradients (gemm_tn16.hip): own tiles, own split-K plan, own pre-pass (column maxima); the groups have been validated
int64_t tn16_workspace_floats(const wsi_gemm_group_t* groups, int32_t ngroups);
int launch_gemm_tn16(int32_t epilogue, const wsi_gemm_group_t* groups, int32_t ngroups, float* ws, int64_t ws_bytes, hipStream_t st);

}  // namespace wsi
