// Weight gradients dW = dY^T X (WSI_GEMM_TN) as a COLUMN-scaled 2-way fp16 split on the fp16 matrix cores: 3 matrix products per fp32 product
// instead of the 6 of the bf16x6 form these launches ran as until round 4 (autograd of every nn.Linear of models/HEATNet4.py:100-102,134,202).
//
// Arithmetic.  C[m, n] = sum_k A[k, m] B[k, n]: the contraction runs over the ROWS of both operands, so the power-of-two scale that brings an
// operand into fp16 range must be constant along k: one per COLUMN of A (output row m) and one per column of B (output column n),
// 2^-e with e = exponent(max_k |x[k, c]|) - 14: the largest element of every column lands in [2^14, 2^15); both scales leave the sum and are
// undone exactly with one v_ldexp_f32 per output element.  Each scaled element is split x = x0 + x1 (two fp16 terms, round to nearest: 22
// significand bits) and the product is summed from x0 y1, x1 y0, x0 y0 on v_mfma_f32_32x32x16_f16, fp32 accumulate.  Two forms of the low term:
//   NACC = 2 (ships): x1 is stored as 2^11 x1 (a normal fp16 number for every element within 2^-28 of its column's largest), the two cross
//             products go to a second accumulator set folded in with weight 2^-11: the arithmetic of the NT / NN kernels (gemm_emu16.hip);
//   NACC = 1 (measurement builds only): x1 stored as it is, all three products in one accumulator set - half the accumulator registers, so a
//             256 x 256 tile per workgroup and a third fewer operand bytes per product; 1.3x faster (profiles/r05_tn16_bench.*) and NOT fp32-class:
//             x1 is a normal fp16 number only for elements within 2^-17 of their column's largest, below that the matrix cores flush it and the
//             element keeps 11 bits - harmless against its own column's largest, but when such an element meets a LARGE element of the other
//             operand the product is as large as any in the sum and carries 2^-12 (tests/...::test_gemm_fp16x3_scaling_cases[TN-outlier] fails).
// Same error class as the row-scaled kernels relative to sum_k |a||b| (tests/test_kernels_gpu.py::test_gemm_fp16x3_scaling_cases[TN-*]).
//
// Data path.  Both operands are contiguous along their OUTPUT index, and an MFMA fragment wants 8 consecutive k of one output index per lane: a
// transpose.  The stage image in the LDS keeps the layout of memory - [k][m] fp16 planes, written with plain 8-byte stores (4 columns of one
// row per lane: no packing of (k, k + 1) pairs, no lane swaps) - and the fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS
// read (semantics established with tools/ubench/tr_b16_semantics.hip: lane s of a 16-lane group addresses row k0 + (s >> 2), columns
// 4 (s & 3) .. + 3 of a [4][16] block and receives column s, rows k0 .. k0 + 3).  Row pitch = tile width + 32 halves: consecutive k land 16 banks
// apart, so the 4 rows x 64 bytes a 32-lane half reads cover all 64 banks once (conflict-free), and the 16 lanes of a store group write 128
// contiguous bytes.  One workgroup = 512 threads = 8 waves as 4 (m) x 2 (n), one per CU (two waves per SIMD), a 256 x 128 tile; 16-deep stages,
// two LDS buffers, a register ring of three stages of 16-byte buffer loads (rows past the slab read as zeros: no tail code), the stage loop
// rotated so that the barrier sits between the second and third product group with the next stage's first fragments requested behind it (see the
// loop); split-K over the rows with per-group balanced slabs, one workgroup per CU, summed in slab order by the shared second stage
// (gemm_f32.hip::splitk_reduce_kernel): deterministic.  Where the time goes (profiles/r05_tn16_ablation.csv, average launch of the bench's
// four weight gradients): products + fragment reads + barriers alone 178 us (the matrix cores at their sustained rate), + LDS stores 212,
// + split arithmetic 234, + loads 251; the six-product bf16 kernel: 498.
//
// Column statistics.  Scales need max_k |x[k, c]| of every column of both operands, the bias gradient the column sums of A.  They come with the
// call (wsi_gemm_group_t.a_colmax / a_colsum / b_colmax: partial tables the operands' producers left - GEMM epilogues, constant features) or from
// a pass of the call's own (colstat_partial_kernel: 256-row x 256-column blocks, plain stores of partial maxima and sums; colstat_final_kernel
// combines the parts in order) - no atomics, no clearing.  Operands shared by several groups (the K, Q and V gradients read the same rows of h)
// are reduced once.
#include "gemm_common.h"
#include "emu16.h"

namespace wsi {

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h16x4;
typedef __attribute__((address_space(3))) h16x4 lds_h16x4;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int T_M = 256;            // tile rows (columns of A)
constexpr int T_THREADS = 512;
constexpr int T_KS = 16;            // k per stage
#ifndef T_NS_DEF
#define T_NS_DEF 3
#endif
constexpr int T_NS = T_NS_DEF;      // register ring: stages of global loads held per wave
constexpr int T_CH = 256;           // rows per chunk of the column-absmax pre-pass

struct TnGroup {
    const float* A; const float* B;
    const uint32_t* abits; const uint32_t* bbits;    // absmax bits of A's columns [M] / B's columns [N]
    int64_t lda, ldb;
    int64_t ws_off;                  // float offset of the group's slabs in the workspace
    int32_t M, N, K;
    int32_t tile_start, tiles_n, tiles_mn, kchunk, flags;   // flags bit0 / bit1: A / B take 16-byte loads, bit2: slabs take 16-byte stores
};
struct TnParams {
    TnGroup g[WSI_GEMM_MAX_GROUPS];
    int32_t ngroups, total_tiles;
};

__device__ __forceinline__ f16x8 tr_read8(const uint16_t* p0, const uint16_t* p1) {
    const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h16x4*)p0);
    const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h16x4*)p1);
    const f16x4 a = __builtin_bit_cast(f16x4, lo), b = __builtin_bit_cast(f16x4, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int TN_, int NACC>
__global__ __launch_bounds__(T_THREADS, 2) void gemm_tn16_kernel(const TnParams P, float* __restrict__ ws) {
    static_assert(TN_ == 128 || TN_ == 256, "tile width");
    constexpr int PA = T_M + 32, PB = TN_ + 32;          // halves per image row
    constexpr int PLANE_A = T_KS * PA, PLANE_B = T_KS * PB;
    constexpr int BUF = 2 * (PLANE_A + PLANE_B);         // halves per stage buffer: A hi | A lo | B hi | B lo
    constexpr int CB = TN_ / 4;                          // 16-byte column groups per row of the B tile
    constexpr int NBQ = TN_ / 128;                       // 16-byte loads per thread and stage of B (A: always 2)
    constexpr int RB = T_THREADS / CB;                   // rows of B one pass of the threads covers
    constexpr int WTN = TN_ / 64;                        // 32-column blocks per wave (two waves along n)
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];

    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngroups; ++i) gi = (tile >= P.g[i].tile_start) ? i : gi;
    const TnGroup& G = P.g[gi];
    int local = tile - G.tile_start;
    const int split = local / G.tiles_mn;
    local -= split * G.tiles_mn;
    const int tm = local / G.tiles_n, tn = local - tm * G.tiles_n;
    const int m0 = tm * T_M, n0 = tn * TN_;
    const int kb = split * G.kchunk, ke = min(G.K, kb + G.kchunk);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- loader coordinates: this thread's 4 columns of A / of B are the same in every stage, and so are their scale exponents
    const int c4a = tid & 63, ra = tid >> 6;             // A: rows ra, ra + 8 of the stage
    const int c4b = tid % CB, rb = tid / CB;             // B: rows rb + q RB
    float sa[4], sb[4];              // 2^-e of the thread's columns (e in [-100, 114]: a normal float)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sa[i] = __builtin_ldexpf(1.f, -scale_exponent(G.abits[min(m0 + 4 * c4a + i, G.M - 1)]));
        sb[i] = __builtin_ldexpf(1.f, -scale_exponent(G.bbits[min(n0 + 4 * c4b + i, G.N - 1)]));
    }
    // Fast path: 16-byte buffer loads through one descriptor per operand that covers exactly this workgroup's slab - rows [kb, ke), from the
    // tile's first column to the operand's last: a row past ke reads as zeros (the range check), so the K tail needs no other code; columns past
    // the operand's last (an edge tile) read the neighbouring columns or zeros and only ever reach output rows / columns that are not stored.
    // The lane offset is one 32-bit register per operand, the stage's offset a scalar.
    const int wA = min(T_M, G.M - m0), wB = min(TN_, G.N - n0);          // columns of the tile that exist
    const bool fast = (G.flags & 1) && (G.flags & 2) && (wA % 4 == 0) && (wB % 4 == 0) && ke > kb &&
                      (int64_t)(ke - kb) * G.lda * 4 < ((int64_t)1 << 31) && (int64_t)(ke - kb) * G.ldb * 4 < ((int64_t)1 << 31);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const float* baseA = G.A + (int64_t)kb * G.lda + m0;
    const float* baseB = G.B + (int64_t)kb * G.ldb + n0;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseA), 0, fast ? (int)(((int64_t)(ke - kb - 1) * G.lda + wA) * 4) : 0, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseB), 0, fast ? (int)(((int64_t)(ke - kb - 1) * G.ldb + wB) * 4) : 0, 0x00020000);
    const int voA = (int)((ra * G.lda + 4 * c4a) * 4), voB = (int)((rb * G.ldb + 4 * c4b) * 4);
    const int stepA = (int)(G.lda * 4), stepB = (int)(G.ldb * 4);      // bytes per row
    struct Stage { float4 a[2]; float4 b[NBQ]; };
    auto as_f4 = [](const u32x4& v) { return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])); };
    auto load_fast = [&](Stage& S, int s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) S.a[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsA, voA, (s * T_KS + 8 * q) * stepA, 0));
#pragma unroll
        for (int q = 0; q < NBQ; ++q) S.b[q] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rsB, voB, (s * T_KS + RB * q) * stepB, 0));
    };
    auto load4g = [](const float* __restrict__ row, int c0, int c_end) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 0 < c_end) r.x = row[c0 + 0];
        if (c0 + 1 < c_end) r.y = row[c0 + 1];
        if (c0 + 2 < c_end) r.z = row[c0 + 2];
        if (c0 + 3 < c_end) r.w = row[c0 + 3];
        return r;
    };
    auto load_guarded = [&](Stage& S, int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = k0 + ra + 8 * q;
            S.a[q] = (k < ke) ? load4g(G.A + (int64_t)k * G.lda, m0 + 4 * c4a, G.M) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int k = k0 + rb + RB * q;
            S.b[q] = (k < ke) ? load4g(G.B + (int64_t)k * G.ldb, n0 + 4 * c4b, G.N) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int nfast = fast ? (ke - kb + T_KS - 1) / T_KS : 0;       // stages of the fast path (all of them, or none)
    // scale, split and store one stage: 8 bytes (4 columns) per plane and loaded 16 bytes
    auto put4 = [&](const float4& v, const float (&sc)[4], uint16_t* hi_p, uint16_t* lo_p) {
        uint32_t h0, l0, h1, l1;
        split_scaled<NACC>(v.x, v.y, sc[0], sc[1], h0, l0);
        split_scaled<NACC>(v.z, v.w, sc[2], sc[3], h1, l1);
        *reinterpret_cast<uint2*>(hi_p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lo_p) = make_uint2(l0, l1);
    };
    auto store_stage = [&](const Stage& S, uint16_t* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint16_t* d = buf + (ra + 8 * q) * PA + 4 * c4a;
            put4(S.a[q], sa, d, d + PLANE_A);
        }
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            uint16_t* d = buf + 2 * PLANE_A + (rb + RB * q) * PB + 4 * c4b;
            put4(S.b[q], sb, d, d + PLANE_B);
        }
    };

    // ---- fragment coordinates (ds_read_b64_tr_b16): 16-lane group g = lane >> 4 reads the [4 k][16 columns] block of k-half g >> 1 and
    // column half g & 1; lane s of it addresses row (s >> 2), columns 4 (s & 3) ..
    const int fg = lane >> 4, fs = lane & 15;
    const int fk = 8 * (fg >> 1) + (fs >> 2), fc = 16 * (fg & 1) + 4 * (fs & 3);
    const int fa_off = fk * PA + wm * 64 + fc;                        // + i * 32 (row block) + 4 PA (second half of the 8 k) + plane
    const int fb_off = 2 * PLANE_A + fk * PB + wn * (TN_ / 2) + fc;    // + j * 32 + 4 PB + plane

    f32x16 acc[NACC][2][WTN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

    // fragments of one plane of the current stage: [32-row block] of A / [32-column block] of B
    auto read_a = [&](const uint16_t* buf, int pl, f16x8 (&f)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint16_t* p = buf + fa_off + pl * PLANE_A + i * 32;
            f[i] = tr_read8(p, p + 4 * PA);
        }
    };
    auto read_b = [&](const uint16_t* buf, int pl, f16x8 (&f)[WTN]) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const uint16_t* p = buf + fb_off + pl * PLANE_B + j * 32;
            f[j] = tr_read8(p, p + 4 * PB);
        }
    };
    auto products = [&](f32x16 (&c)[2][WTN], const f16x8 (&a)[2], const f16x8 (&b)[WTN]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], c[i][j], 0, 0, 0);
    };
    // x0 y1, x1 y0 (-> the cross accumulators), x0 y0: the order of the NT / NN kernels per accumulator.  The low plane of B is read first and
    // is dead after the first product group, the low plane of A after the second: at most three of the four fragment sets are live.
    constexpr int CX = NACC - 1;
    auto multiply_stage = [&](const uint16_t* buf) {
        f16x8 a0[2], a1[2], b0[WTN], b1[WTN];
        read_a(buf, 0, a0);
        read_b(buf, 1, b1);
        read_a(buf, 1, a1);
        read_b(buf, 0, b0);
        products(acc[CX], a0, b1);
        products(acc[CX], a1, b0);
        products(acc[0], a0, b0);
    };

    if (nfast > 0) {
        // Register ring of NS stages of global loads: stage t lives in R[t % NS]; iteration s requests stage s + NS into the set stage s left free an
        // iteration ago (past the end: the last stage again - never consumed - so that the body has no branch).
        //
        // One iteration = one 16-deep stage s, ROTATED so that no LDS latency sits between a barrier and the first product behind it:
        //     on entry   a0(s), b1(s) are in registers (requested behind the previous barrier)
        //     request    a1(s), b0(s)                       <- land under the first product group
        //     x0 y1      4 products                            the split of stage s + 1 (registers) rides between the products
        //     x1 y0      4 products                            its LDS stores go out here: buffer (s + 1) & 1 was last read for stage s - 1
        //     barrier    stage s + 1 is complete in the LDS, every wave's reads of stage s are done (its buffer is free for stage s + 2)
        //     request    a0(s + 1), b1(s + 1) into the registers a1(s) / b1(s) left free    <- land under the third group
        //     x0 y0      4 products
        // The barrier waits with eight products of each wave already issued: the matrix pipe drains them while the workgroup synchronises.  The same
        // products in the same order per accumulator as the NT / NN kernels (x0 y1, x1 y0 -> the cross accumulators; x0 y0).
        constexpr int NS = (TN_ == 128) ? T_NS : 2;
        Stage R[NS];
        struct Frags { f16x8 a0[2], a1[2], b0[WTN], b1[WTN]; };
        auto body = [&](Frags& c, Frags& n, Stage& C, Stage& F, int s) {
            load_fast(F, min(s + NS, nfast - 1));
            const uint16_t* cur = smem + (s & 1) * BUF;
            uint16_t* oth = smem + ((s + 1) & 1) * BUF;
            // waves still in front of the stage barrier outrank the ones already past it (-2 % on the bench shapes: whoever is late holds up all eight)
            __builtin_amdgcn_s_setprio(2);
            read_a(cur, 1, c.a1);
            read_b(cur, 0, c.b0);
            products(acc[CX], c.a0, c.b1);
            store_stage(C, oth);
            products(acc[CX], c.a1, c.b0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2 + NBQ, 0);         // the global loads
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + WTN), 0);   // the fragment reads
#pragma unroll
            for (int m = 0; m < 4 * WTN; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (m >= 4 * WTN - (2 + NBQ)) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            read_a(oth, 0, n.a0);
            read_b(oth, 1, n.b1);
            products(acc[0], c.a0, c.b0);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int q = 0; q < NS; ++q) load_fast(R[q], min(q, nfast - 1));
        store_stage(R[0], smem);
        __syncthreads();
        Frags X, Y;
        read_a(smem, 0, X.a0);
        read_b(smem, 1, X.b1);
        // unrolled by 2 NS: the fragment sets alternate, the ring rotates (NS = 3: six bodies)
        int s = 0;
        for (; s + 2 * NS <= nfast; s += 2 * NS) {
#pragma unroll
            for (int q = 0; q < 2 * NS; ++q) {
                if (q & 1) body(Y, X, R[(q + 1) % NS], R[q % NS], s + q);
                else body(X, Y, R[(q + 1) % NS], R[q % NS], s + q);
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * NS - 1; ++q) {
            if (s + q < nfast) {
                if (q & 1) body(Y, X, R[(q + 1) % NS], R[q % NS], s + q);
                else body(X, Y, R[(q + 1) % NS], R[q % NS], s + q);
            }
        }
    }
    // guarded stages: the K tail, edge tiles, operands that do not take 16-byte loads
    for (int k0 = kb + nfast * T_KS; k0 < ke; k0 += T_KS) {
        Stage g;
        load_guarded(g, k0);
        store_stage(g, smem);
        __syncthreads();
        multiply_stage(smem);
        __syncthreads();
    }

    // ---- epilogue: undo the column scales, write the slab of this split
    int* se = reinterpret_cast<int*>(smem);
    for (int i = tid; i < T_M + TN_; i += T_THREADS)
        se[i] = (i < T_M) ? scale_exponent(G.abits[min(m0 + i, G.M - 1)]) : scale_exponent(G.bbits[min(n0 + i - T_M, G.N - 1)]);
    __syncthreads();
    float* slab = ws + G.ws_off + (int64_t)split * G.M * G.N;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int cl = wn * (TN_ / 2) + j * 32 + l31;
        const int col = n0 + cl;
        const int ec = se[T_M + cl];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int row = m0 + rl;
                float x = acc[0][i][j][r];
                if constexpr (NACC == 2) x = fmaf(acc[1][i][j][r], 1.f / LO_SCALE, x);
                x = __builtin_ldexpf(x, ec + se[rl]);
                if (row < G.M && col < G.N) slab[(int64_t)row * G.N + col] = x;
            }
    }
}

// ------------------------------------------------------------------------------------------------ column statistics
// Per operand column: the absmax bits (the scale) and - for A, when the caller wants the bias gradient - the column SUM (dL/db = column sums of
// dY: the kernel above stages dY anyway, but eight additions per stage and thread cost it more issue slots than this pass, which reads the
// operand for its maxima in any case).  Two stages, plain stores, fixed order: deterministic.
struct ColJob {
    const float* X; int64_t ld;
    const uint32_t* part;            // [chunks][part_ld] partial maxima (bit patterns): the job's own (workspace) or the caller's (a producer left them)
    const float* psum;               // [chunks][part_ld] partial sums, or NULL
    int64_t part_ld;
    int32_t own;                     // 1: colstat_partial_kernel fills part / psum; 0: they came with the call
    uint32_t* out;                   // [cols] absmax bits
    float* sum_out;                  // [cols] column sums x gate (+ old value under ACCUMULATE), or NULL
    const float* gate;
    int32_t rows, cols, cols_pad, chunks, col_blocks;
    int32_t block_start;             // first workgroup of the job in the partial launch
    int32_t fblock_start;            // ... in the final launch
    int32_t vec;
};
struct ColParams {
    ColJob j[2 * WSI_GEMM_MAX_GROUPS];
    int32_t njobs, epilogue;
};

// partial[chunk][c] = bits(max over the chunk's rows of |X[r][c]|), psum[chunk][c] = their sum: one workgroup per (chunk of T_CH rows, block of
// 256 columns); thread = 4 columns x every 4th row (fmaxf drops NaNs: a NaN element keeps its column's finite scale and propagates through the split)
__global__ __launch_bounds__(256) void colstat_partial_kernel(const ColParams P) {
    __shared__ float4 sm[256], ss[256];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < P.njobs; ++i) ji = (P.j[i].own && (int)blockIdx.x >= P.j[i].block_start) ? i : ji;      // (jobs that came with their tables have no workgroup here)
    const ColJob& J = P.j[ji];
    if (!J.own) return;                                                                                           // (only job 0 can get here)
    const int b = (int)blockIdx.x - J.block_start;
    const int chunk = b / J.col_blocks, cb = b - chunk * J.col_blocks;
    const int c4 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c0 = cb * 256 + 4 * c4;
    const int r0 = chunk * T_CH, r1 = min(J.rows, r0 + T_CH);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.vec && c0 + 3 < J.cols) {
        const float* p = J.X + c0;
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)r * J.ld);
            m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
    } else if (c0 < J.cols) {
        for (int r = r0 + rl; r < r1; r += 4) {
            const float* p = J.X + (int64_t)r * J.ld + c0;
            m.x = fmaxf(m.x, fabsf(p[0])); t.x += p[0];
            if (c0 + 1 < J.cols) { m.y = fmaxf(m.y, fabsf(p[1])); t.y += p[1]; }
            if (c0 + 2 < J.cols) { m.z = fmaxf(m.z, fabsf(p[2])); t.z += p[2]; }
            if (c0 + 3 < J.cols) { m.w = fmaxf(m.w, fabsf(p[3])); t.w += p[3]; }
        }
    }
    sm[threadIdx.x] = m;
    ss[threadIdx.x] = t;
    __syncthreads();
    if (rl == 0 && c0 < J.cols_pad) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 o = sm[c4 + 64 * q], u = ss[c4 + 64 * q];
            m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<uint4*>(const_cast<uint32_t*>(J.part) + (int64_t)chunk * J.part_ld + c0) =
            make_uint4(__float_as_uint(m.x), __float_as_uint(m.y), __float_as_uint(m.z), __float_as_uint(m.w));
        if (J.psum) *reinterpret_cast<float4*>(const_cast<float*>(J.psum) + (int64_t)chunk * J.part_ld + c0) = t;
    }
}

// out[c] = max over the chunks of partial[chunk][c] (non-negative floats order like their bit patterns); sum_out[c] = the chunks' sums.  One
// workgroup per 16 columns: thread = one column x every 16th chunk, four independent loads in flight, combined through the LDS in lane order
// (a fixed order: deterministic).  (One column per thread over ALL chunks was 64 us of dependent loads for a producer's 313 parts.)
constexpr int CF_COLS = 16, CF_LANES = 256 / CF_COLS;
__global__ __launch_bounds__(256) void colstat_final_kernel(const ColParams P) {
    __shared__ uint32_t sm[256];
    __shared__ float ss[256];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < P.njobs; ++i) ji = ((int)blockIdx.x >= P.j[i].fblock_start) ? i : ji;
    const ColJob& J = P.j[ji];
    const int cl = threadIdx.x % CF_COLS, kl = threadIdx.x / CF_COLS;
    const int c = ((int)blockIdx.x - J.fblock_start) * CF_COLS + cl;
    uint32_t m = 0u;
    float t = 0.f;
    if (c < J.cols) {
        const uint32_t* pm = J.part + c;
        const float* ps = J.psum ? J.psum + c : nullptr;
        int k = kl;
        for (; k + 3 * CF_LANES < J.chunks; k += 4 * CF_LANES) {
            const uint32_t m0 = pm[(int64_t)k * J.part_ld], m1 = pm[(int64_t)(k + CF_LANES) * J.part_ld];
            const uint32_t m2 = pm[(int64_t)(k + 2 * CF_LANES) * J.part_ld], m3 = pm[(int64_t)(k + 3 * CF_LANES) * J.part_ld];
            m = max(max(m, max(m0, m1)), max(m2, m3));
            if (ps) {
                const float s0 = ps[(int64_t)k * J.part_ld], s1 = ps[(int64_t)(k + CF_LANES) * J.part_ld];
                const float s2 = ps[(int64_t)(k + 2 * CF_LANES) * J.part_ld], s3 = ps[(int64_t)(k + 3 * CF_LANES) * J.part_ld];
                t += (s0 + s1) + (s2 + s3);
            }
        }
        for (; k < J.chunks; k += CF_LANES) {
            m = max(m, pm[(int64_t)k * J.part_ld]);
            if (ps) t += ps[(int64_t)k * J.part_ld];
        }
    }
    sm[threadIdx.x] = m;
    ss[threadIdx.x] = t;
    __syncthreads();
    if (kl == 0 && c < J.cols) {
#pragma unroll
        for (int q = 1; q < CF_LANES; ++q) { m = max(m, sm[cl + CF_COLS * q]); t += ss[cl + CF_COLS * q]; }
        J.out[c] = m;
        if (J.sum_out) {
            if ((P.epilogue & WSI_EPI_SCALE_GATE) && J.gate) t *= 1.f / (1.f + expf(-(*J.gate)));
            if (P.epilogue & WSI_EPI_ACCUMULATE) t += J.sum_out[c];
            J.sum_out[c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static inline bool tn_vec_ok(const void* p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0); }

// the tile form: 256 x 128 with two accumulator sets; measurement builds: WSI_TN16_CFG=256 selects 256 x 256 with one
static int tn16_tile_n() {
    const char* s = knob("WSI_TN16_CFG");          // (a constant in the product build; read per call in measurement builds: A/B runs in one process)
    return (s && s[0] == '2') ? 256 : 128;
}

struct Tn16Plan {
    int32_t splits[WSI_GEMM_MAX_GROUPS];
    int32_t kchunk[WSI_GEMM_MAX_GROUPS];
    int32_t tiles_m[WSI_GEMM_MAX_GROUPS], tiles_n[WSI_GEMM_MAX_GROUPS];
    int32_t total_tiles;
};

// Split-K plan: one workgroup per CU (256), every workgroup about the same number of rows.  A common slab length decides how many slabs each
// group gets; inside a group the rows are then dealt out evenly (multiples of the stage depth), so the last slab of a group is not a stub.
static void tn16_plan(const wsi_gemm_group_t* g, int32_t ng, int tn, Tn16Plan& pl) {
    const int64_t target = 256;
    int64_t work = 0, maxk = 0;
    for (int i = 0; i < ng; ++i) {
        pl.tiles_m[i] = pl.tiles_n[i] = 0; pl.splits[i] = 1; pl.kchunk[i] = g[i].K > 0 ? g[i].K : T_KS;
        if (g[i].M <= 0 || g[i].N <= 0) continue;
        pl.tiles_m[i] = (g[i].M + T_M - 1) / T_M; pl.tiles_n[i] = (g[i].N + tn - 1) / tn;
        work += (int64_t)pl.tiles_m[i] * pl.tiles_n[i] * g[i].K;
        if (g[i].K > maxk) maxk = g[i].K;
    }
    int64_t kc = (work + target - 1) / target;
    kc = ((kc + T_KS - 1) / T_KS) * T_KS;
    if (kc < 8 * T_KS) kc = 8 * T_KS;
    for (;;) {
        int64_t blocks = 0;
        for (int i = 0; i < ng; ++i) {
            if (!pl.tiles_m[i]) continue;
            pl.splits[i] = g[i].K > 0 ? (int32_t)((g[i].K + kc - 1) / kc) : 1;
            blocks += (int64_t)pl.tiles_m[i] * pl.tiles_n[i] * pl.splits[i];
        }
        if (blocks <= target || kc >= maxk) break;
        kc += T_KS;
    }
    int32_t tiles = 0;
    for (int i = 0; i < ng; ++i) {
        if (!pl.tiles_m[i]) continue;
        if (g[i].K > 0) {
            int64_t per = (g[i].K + pl.splits[i] - 1) / pl.splits[i];
            per = ((per + T_KS - 1) / T_KS) * T_KS;
            pl.kchunk[i] = (int32_t)per;
            pl.splits[i] = (int32_t)((g[i].K + per - 1) / per);
        }
        tiles += pl.tiles_m[i] * pl.tiles_n[i] * pl.splits[i];
    }
    pl.total_tiles = tiles;
}

static inline int64_t pad4(int64_t x) { return (x + 3) & ~(int64_t)3; }

// workspace (floats): slabs + column-sum partials of every group, then per group the scale words and the chunk tables of the column maxima
int64_t tn16_workspace_floats(const wsi_gemm_group_t* groups, int32_t ngroups) {
    Tn16Plan pl;
    tn16_plan(groups, ngroups, tn16_tile_n(), pl);
    int64_t f = 0;
    for (int i = 0; i < ngroups; ++i) {
        if (groups[i].M <= 0 || groups[i].N <= 0) continue;
        f += (int64_t)pl.splits[i] * groups[i].M * groups[i].N;
    }
    f = pad4(f);
    for (int i = 0; i < ngroups; ++i) {
        if (groups[i].M <= 0 || groups[i].N <= 0) continue;
        const int64_t chunks = (groups[i].K + T_CH - 1) / T_CH;
        f += (pad4(groups[i].M) + pad4(groups[i].N)) * (1 + chunks) + (groups[i].colsum_out ? pad4(groups[i].M) * chunks : 0);
    }
    return f;
}

// groups: validated by wsi_gemm_grouped (non-negative shapes, non-null pointers where K > 0)
int launch_gemm_tn16(int32_t epilogue, const wsi_gemm_group_t* groups, int32_t ngroups, float* ws, int64_t ws_bytes, hipStream_t st) {
    const int tn = tn16_tile_n();
    Tn16Plan pl;
    tn16_plan(groups, ngroups, tn, pl);
    const int64_t need = tn16_workspace_floats(groups, ngroups);
    if (need >= ((int64_t)1 << 31)) { set_error("gemm TN fp16x3: workspace of %lld floats exceeds the 2^31 index range", (long long)need); return WSI_EINVAL; }
    if (!ws || ws_bytes < need * 4) { set_error("gemm TN fp16x3: workspace of %lld bytes needed, %lld given", (long long)(need * 4), (long long)ws_bytes); return WSI_ENOMEM; }
    if (reinterpret_cast<uintptr_t>(ws) & 15) { set_error("gemm TN fp16x3: the workspace must be 16-byte aligned"); return WSI_EINVAL; }
    TnParams P;
    ReduceParams RP;
    ColParams CP;
    P.ngroups = 0; RP.ngroups = 0; RP.epilogue = epilogue; CP.njobs = 0;
    int64_t f = 0, red_total = 0;
    int32_t tiles = 0;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.M <= 0 || s.N <= 0) continue;
        TnGroup& d = P.g[P.ngroups];
        d.A = s.A; d.B = s.B; d.lda = s.lda; d.ldb = s.ldb; d.M = s.M; d.N = s.N; d.K = s.K;
        d.tiles_n = pl.tiles_n[i]; d.tiles_mn = pl.tiles_m[i] * pl.tiles_n[i]; d.kchunk = pl.kchunk[i];
        d.tile_start = tiles;
        tiles += d.tiles_mn * pl.splits[i];
        d.flags = (tn_vec_ok(s.A, s.lda) ? 1 : 0) | (tn_vec_ok(s.B, s.ldb) ? 2 : 0);
        d.ws_off = f;
        ReduceDesc& r = RP.g[RP.ngroups++];
        r.ws = ws + f; r.C = s.C; r.gate = s.gate; r.ldc = s.ldc; r.M = s.M; r.N = s.N; r.pad = 0; r.splits = pl.splits[i]; r.start = red_total;
        red_total += (int64_t)s.M * s.N;
        f += (int64_t)pl.splits[i] * s.M * s.N;
        r.cs_ws = nullptr; r.cs_out = nullptr;      // (the bias gradient comes out of the column-statistics pass)
        P.ngroups++;
    }
    if (P.ngroups == 0) return WSI_OK;
    P.total_tiles = tiles;
    f = pad4(f);
    // column maxima: one job per DISTINCT operand (pointer, pitch, rows, columns)
    uint32_t* wu = reinterpret_cast<uint32_t*>(ws);
    int32_t pblocks = 0, fblocks = 0;
    CP.epilogue = epilogue;
    // given: partial tables that came with the call (gmax / gsum, gparts rows of pitch gld), or nullptr: the job makes its own pass
    auto job_for = [&](const float* X, int64_t ld, int32_t rows, int32_t cols, float* sum_out, const float* gate,
                       const uint32_t* gmax, const float* gsum, int64_t gld, int32_t gparts, int64_t& cursor) -> const uint32_t* {
        if (gmax && (gparts < 1 || (sum_out && !gsum))) gmax = nullptr;      // (sums wanted but not given: one own pass serves both)
        for (int q = 0; q < CP.njobs; ++q) {
            ColJob& J = CP.j[q];
            if (J.X == X && J.ld == ld && J.rows == rows && J.cols == cols && !sum_out && (J.own || J.part == gmax)) return J.out;
        }
        ColJob& J = CP.j[CP.njobs++];
        J.X = X; J.ld = ld; J.rows = rows; J.cols = cols; J.cols_pad = (int32_t)pad4(cols);
        J.col_blocks = (cols + 255) / 256;
        J.out = wu + cursor; cursor += J.cols_pad;
        J.sum_out = sum_out; J.gate = gate;
        if (gmax) {
            J.own = 0; J.part = gmax; J.psum = sum_out ? gsum : nullptr; J.part_ld = gld; J.chunks = gparts; J.vec = 0;
            J.block_start = pblocks;                                         // no workgroup of the partial launch
        } else {
            J.own = 1; J.part_ld = J.cols_pad; J.chunks = (rows + T_CH - 1) / T_CH;
            J.part = wu + cursor; cursor += (int64_t)J.chunks * J.cols_pad;
            J.psum = nullptr;
            if (sum_out) { J.psum = reinterpret_cast<float*>(wu + cursor); cursor += (int64_t)J.chunks * J.cols_pad; }
            J.vec = tn_vec_ok(X, ld) ? 1 : 0;
            J.block_start = pblocks; pblocks += J.chunks * J.col_blocks;
        }
        J.fblock_start = fblocks; fblocks += (cols + CF_COLS - 1) / CF_COLS;
        return J.out;
    };
    {
        int gidx = 0;
        for (int i = 0; i < ngroups; ++i) {
            const wsi_gemm_group_t& s = groups[i];
            if (s.M <= 0 || s.N <= 0) continue;
            TnGroup& d = P.g[gidx++];
            int64_t cursor = f;
            d.abits = job_for(s.A, s.lda, s.K, s.M, s.colsum_out, s.gate, s.a_colmax, s.a_colsum, s.a_col_ld, s.a_col_parts, cursor);
            d.bbits = job_for(s.B, s.ldb, s.K, s.N, nullptr, nullptr, s.b_colmax, nullptr, s.b_col_ld, s.b_col_parts, cursor);
            const int64_t chunks = (s.K + T_CH - 1) / T_CH;
            f += (pad4(s.M) + pad4(s.N)) * (1 + chunks) + (s.colsum_out ? pad4(s.M) * chunks : 0);      // the group's reservation (a shared operand leaves its share unused)
        }
    }
    if (pblocks > 0) hipLaunchKernelGGL(colstat_partial_kernel, dim3(pblocks), dim3(256), 0, st, CP);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(fblocks), dim3(256), 0, st, CP);
#ifdef WSI_ABLATE
    if (tn == 256) hipLaunchKernelGGL((gemm_tn16_kernel<256, 1>), dim3(tiles), dim3(T_THREADS), 0, st, P, ws);      // (not fp32-class: measurements only)
    else
#endif
    hipLaunchKernelGGL((gemm_tn16_kernel<128, 2>), dim3(tiles), dim3(T_THREADS), 0, st, P, ws);
    RP.total = red_total;
    launch_splitk_reduce(RP, st);
    return check_launch("gemm_tn16");
}

}  // namespace wsi

using namespace wsi;

extern "C" int32_t wsi_col_stats_parts(int32_t rows) { return rows > 0 ? (rows + T_CH - 1) / T_CH : 0; }

extern "C" int wsi_col_stats(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* part_max, float* part_sum, int64_t part_ld, void* stream) {
    if (rows < 0 || cols < 0) { set_error("col_stats: bad shape %d x %d", rows, cols); return WSI_EINVAL; }
    if (rows == 0 || cols == 0) return WSI_OK;
    if (!x || !part_max || part_ld < ((cols + 3) & ~3) || (part_ld & 3) || (reinterpret_cast<uintptr_t>(part_max) & 15) || (part_sum && (reinterpret_cast<uintptr_t>(part_sum) & 15))) {
        set_error("col_stats: null pointer, or tables not 16-byte aligned with a pitch >= the columns rounded up to 4"); return WSI_EINVAL; }
    ColParams CP;
    CP.njobs = 1; CP.epilogue = 0;
    ColJob& J = CP.j[0];
    J.X = x; J.ld = ld; J.rows = rows; J.cols = cols; J.cols_pad = (int32_t)pad4(cols); J.col_blocks = (cols + 255) / 256;
    J.out = nullptr; J.sum_out = nullptr; J.gate = nullptr; J.own = 1; J.part_ld = part_ld; J.chunks = (rows + T_CH - 1) / T_CH;
    J.part = part_max; J.psum = part_sum; J.vec = tn_vec_ok(x, ld) ? 1 : 0;
    J.block_start = 0; J.fblock_start = 0;
    hipLaunchKernelGGL(colstat_partial_kernel, dim3(J.chunks * J.col_blocks), dim3(256), 0, (hipStream_t)stream, CP);
    return check_launch("col_stats");
}

extern "C" int64_t wsi_col_absmax_workspace_bytes(int32_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (int64_t)((rows + T_CH - 1) / T_CH) * pad4(cols) * 4;
}

extern "C" int wsi_col_absmax(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (rows < 0 || cols < 0) { set_error("col_absmax: bad shape %d x %d", rows, cols); return WSI_EINVAL; }
    if (cols == 0) return WSI_OK;
    if (!out || (rows > 0 && !x)) { set_error("col_absmax: null pointer"); return WSI_EINVAL; }
    const int64_t need = wsi_col_absmax_workspace_bytes(rows, cols);
    if (need > 0 && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))) {
        set_error("col_absmax: a 16-byte aligned workspace of %lld bytes is needed", (long long)need); return WSI_ENOMEM; }
    ColParams CP;
    CP.njobs = 1; CP.epilogue = 0;
    ColJob& J = CP.j[0];
    J.X = x; J.ld = ld; J.rows = rows; J.cols = cols; J.cols_pad = (int32_t)pad4(cols); J.col_blocks = (cols + 255) / 256;
    J.out = out; J.sum_out = nullptr; J.gate = nullptr; J.own = 1; J.part_ld = J.cols_pad; J.chunks = (rows + T_CH - 1) / T_CH;
    J.part = reinterpret_cast<uint32_t*>(workspace); J.psum = nullptr; J.vec = tn_vec_ok(x, ld) ? 1 : 0;
    J.block_start = 0; J.fblock_start = 0;
    hipStream_t st = (hipStream_t)stream;
    if (J.chunks > 0) hipLaunchKernelGGL(colstat_partial_kernel, dim3(J.chunks * J.col_blocks), dim3(256), 0, st, CP);
    hipLaunchKernelGGL(colstat_final_kernel, dim3((cols + CF_COLS - 1) / CF_COLS), dim3(256), 0, st, CP);
    return check_launch("col_absmax");
}
