// Grouped fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) with fused epilogues.
// Contract + reference call sites (torch.nn.Linear at models/HEATNet4.py:100-102,134,202,219,243-245
// and their autograd): include/wsi_hgnn.h.
//
// Why exact-fp32 MFMA: the path must match the reference's fp32 arithmetic to 1e-4 on logits and
// gradients with K up to 1024, which rules out plain bf16/fp16 MFMA; gfx950 has no xf32/TF32.
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain at 64 FLOP/clk/SIMD = 157.3 TFLOP/s.
//
// Structure (one launch serves every node type / projection of a layer):
//   * a "group" = one independent GEMM (node type x projection); tiles of all groups are enumerated
//     in one grid, group lookup is a short scalar scan over the by-value descriptor table;
//   * 128x128x32 block tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles
//     (64 accumulator registers), one A/B fragment = ONE fp32 VGPR per lane;
//   * LDS tiles are stored k-major ([BK][BM+pad]) so a fragment read is a conflict-free ds_read_b32 of
//     32 consecutive floats per half wave for every operand layout; K-contiguous operands are
//     transposed while staging (pad 1 makes those ds_write_b32 conflict-free), M/N-contiguous operands
//     stage with ds_write_b128;
//   * global->register prefetch of tile t+1 is issued before the MFMA loop of tile t (the f32 matrix
//     pipe is slow enough - 64 cycles per MFMA - that one LDS buffer + register prefetch keeps it fed);
//   * blockIdx is remapped so each XCD (block b runs on XCD b%8) walks a contiguous range of tiles:
//     neighbouring tiles share their A row panel / the whole weight matrix in that XCD's 4 MiB L2;
//   * TN (dW = dY^T X, reduction over tens of thousands of rows, only a few output tiles) is split
//     along the reduction into slabs in a caller workspace and summed by a second kernel in a fixed
//     order: deterministic, no atomics.
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>

namespace wsi {

// 4 consecutive floats starting at p (logical index i0 of a dimension of extent n); zero beyond n. Scalar, any alignment.
__device__ __forceinline__ float4 load4_guarded(const float* __restrict__ p, int i0, int n) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + 0 < n) r.x = p[0];
    if (i0 + 1 < n) r.y = p[1];
    if (i0 + 2 < n) r.z = p[2];
    if (i0 + 3 < n) r.w = p[3];
    return r;
}

// Operand tile loader. "Outer" = the M (for A) or N (for B) dimension of the tile (128 wide),
// "k" = the reduction dimension (32 deep).  KCONTIG: element (o,k) at base[o*ld + k]; else base[k*ld + o].
//   load_fast   : unguarded 16-byte loads.  Requires a full k-tile (k0+BK <= k_end) and 16-byte alignment;
//                 KCONTIG rows beyond o_end are CLAMPED to the last row (their products land in C rows/cols
//                 that are never stored), non-KCONTIG needs the whole tile inside (o0+BM <= o_end).
//   load_guarded: per-element guarded scalar loads with zero fill (edge tiles, k tails, unaligned operands).
template <bool KCONTIG>
struct TileLoader {
    float4 r[4];
    __device__ __forceinline__ void load_fast(const float* __restrict__ base, int64_t ld, int o0, int k0, int o_end, int tid) {
        if constexpr (KCONTIG) {
            const int c = tid & 7, rr = tid >> 3;
            const float* p = base + k0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = min(o0 + rr + 32 * q, o_end - 1);
                r[q] = *reinterpret_cast<const float4*>(p + (int64_t)o * ld);
            }
        } else {
            const int c = tid & 31, kr = tid >> 5;
            const float* p = base + o0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                r[q] = *reinterpret_cast<const float4*>(p + (int64_t)(k0 + kr + 8 * q) * ld);
        }
    }
    // one quarter (a single float4) of load_fast / store, so staging can be interleaved between MFMA groups
    __device__ __forceinline__ void load_fast_piece(int q, const float* __restrict__ base, int64_t ld, int o0, int k0, int o_end, int tid) {
        if constexpr (KCONTIG) {
            const int c = tid & 7, rr = tid >> 3;
            const int o = min(o0 + rr + 32 * q, o_end - 1);
            r[q] = *reinterpret_cast<const float4*>(base + (int64_t)o * ld + k0 + 4 * c);
        } else {
            const int c = tid & 31, kr = tid >> 5;
            r[q] = *reinterpret_cast<const float4*>(base + (int64_t)(k0 + kr + 8 * q) * ld + o0 + 4 * c);
        }
    }
    __device__ __forceinline__ void store_piece(int q, float* __restrict__ lds, int tid) const {
        if constexpr (KCONTIG) {
            const int c = tid & 7, rr = tid >> 3;
            float* d = lds + (4 * c) * LD_T + rr + 32 * q;
            d[0] = r[q].x; d[LD_T] = r[q].y; d[2 * LD_T] = r[q].z; d[3 * LD_T] = r[q].w;
        } else {
            const int c = tid & 31, kr = tid >> 5;
            *reinterpret_cast<float4*>(lds + (kr + 8 * q) * LD_N + 4 * c) = r[q];
        }
    }
    __device__ __forceinline__ void load_guarded(const float* __restrict__ base, int64_t ld, int o0, int k0,
                                                 int o_end, int k_end, int tid) {
        if constexpr (KCONTIG) {
            const int c = tid & 7, rr = tid >> 3;
            const int k = k0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = o0 + rr + 32 * q;
                if (o < o_end && k < k_end) r[q] = load4_guarded(base + (int64_t)o * ld + k, k, k_end);
                else r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int c = tid & 31, kr = tid >> 5;
            const int o = o0 + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kr + 8 * q;
                if (k < k_end && o < o_end) r[q] = load4_guarded(base + (int64_t)k * ld + o, o, o_end);
                else r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
        if constexpr (KCONTIG) {
            const int c = tid & 7, rr = tid >> 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* d = lds + (4 * c) * LD_T + rr + 32 * q;
                d[0] = r[q].x; d[LD_T] = r[q].y; d[2 * LD_T] = r[q].z; d[3 * LD_T] = r[q].w;
            }
        } else {
            const int c = tid & 31, kr = tid >> 5;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(lds + (kr + 8 * q) * LD_N + 4 * c) = r[q];
        }
    }
};

// A_KC: A is [M,K] with K contiguous (else stored [K,M]).  B_KC: B is [N,K] with K contiguous (else [K,N]).
// PIPE: double-buffered LDS, ONE barrier per K-tile, and the staging of tile t+1 (ds_write) / prefetch of tile t+2
// (global_load) issued in the shadow of the 64-cycle MFMAs of tile t, so a single wave keeps its SIMD's matrix
// pipe continuously busy (2 workgroups per CU).  !PIPE: single LDS buffer, stage -> barrier -> MFMA -> barrier
// phases overlapped only across the 3 workgroups of a CU.
// RES = workgroups per CU the register budget is compiled for (!PIPE only): 3 (170 VGPRs) or 4 (128 VGPRs, no spills).
// Same code; which one is faster depends on how the launch's tile count quantises into residency rounds (see
// wsi_gemm_grouped): measured on the bench shapes, 4/CU wins for the 2500-tile launches (+5 %) and at 4096^3 (+23 %),
// 3/CU for the 7500-tile and the split-K (one planned round) launches.
template <bool A_KC, bool B_KC, bool SPLITK, bool PIPE, int RES = 3>
__global__ __launch_bounds__(GEMM_THREADS, PIPE ? 2 : RES) void gemm_f32_kernel(const GemmParams P, float* __restrict__ ws) {
    constexpr int LDA_S = A_KC ? LD_T : LD_N;
    constexpr int LDB_S = B_KC ? LD_T : LD_N;
    constexpr int STAGE = BK * LDA_S + BK * LDB_S;
    __shared__ __attribute__((aligned(16))) float smem[(PIPE ? 2 : 1) * STAGE];   // >= 4 waves x 32 x 64 floats (epilogue staging)
    float* As = smem;
    float* Bs = smem + BK * LDA_S;

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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<A_KC> la;
    TileLoader<B_KC> lb;
    const float* ap = As + hi * LDA_S + wm * 64 + l31;
    const float* bp = Bs + hi * LDB_S + wn * 64 + l31;

    // TN only: the workgroups of the first N-tile also reduce their A tiles over k (column sums of dY = bias gradient),
    // straight out of the staged LDS tile - the separate colsum pass over dY disappears.
    const bool do_colsum = SPLITK && !A_KC && (G.cs_off >= 0) && (tn == 0) && (tid < BM);
    float csum = 0.f;
    auto colsum_tile = [&](const float* Ard) {
        if (do_colsum) {
#pragma unroll
            for (int k = 0; k < BK; ++k) csum += Ard[k * LDA_S + tid];
        }
    };

    // One K-tile of MFMAs out of LDS; fragments of step kk+2 are read while the MFMAs of step kk run.
    auto compute_tile = [&]() {
        colsum_tile(As);
        float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < BK) {
                na0 = ap[(kk + 2) * LDA_S]; na1 = ap[(kk + 2) * LDA_S + 32];
                nb0 = bp[(kk + 2) * LDB_S]; nb1 = bp[(kk + 2) * LDB_S + 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads ahead of this step's MFMAs
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    };

    // wave-uniform choice: unguarded 16-byte loads for every full K-tile, guarded scalar loads otherwise
    const bool fast = avec && bvec && (A_KC ? true : (m0 + BM <= G.M)) && (B_KC ? true : (n0 + BN <= G.N));
    const int nfull = fast ? (ke - kb) / BK : 0;
    // NN with several B matrices: k-tile k0 lives in matrix k0 / bchunk at local row k0 % bchunk
    auto bsel = [&](int k0, int& kloc) -> const float* {
        if (G.bchunk <= 0) { kloc = k0; return G.B; }
        const int w = k0 / G.bchunk;
        kloc = k0 - w * G.bchunk;
        return w == 0 ? G.B : (w == 1 ? G.B1 : G.B2);
    };
    if constexpr (PIPE) {
        if (nfull > 0) {
            float* A0 = smem;
            float* B0 = A0 + BK * LDA_S;
            float* A1 = smem + STAGE;
            float* B1 = A1 + BK * LDA_S;
            const int klast = kb + (nfull - 1) * BK;
            int kl;
            const float* bb = bsel(kb, kl);
            la.load_fast(G.A, G.lda, m0, kb, G.M, tid);
            lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
            la.store(A0, tid);
            lb.store(B0, tid);
            __syncthreads();
            const int k1 = min(kb + BK, klast);
            bb = bsel(k1, kl);
            la.load_fast(G.A, G.lda, m0, k1, G.M, tid);      // registers now hold tile 1
            lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
            // one K-tile: MFMAs read (Ard,Brd); the registers (tile t+1) go to (Awr,Bwr) during steps 0-7, then tile k2
            // (= t+2) is prefetched into the same registers during steps 8-15 - all between MFMA groups.
            auto body = [&](const float* Ard, const float* Brd, float* Awr, float* Bwr, int k2) {
                const float* pa = Ard + hi * LDA_S + wm * 64 + l31;
                const float* pb = Brd + hi * LDB_S + wn * 64 + l31;
                int kl2;
                const float* bb2 = bsel(k2, kl2);
                colsum_tile(Ard);
                float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
#pragma unroll
                for (int st = 0; st < BK / 2; ++st) {
                    const int kk = 2 * st;
                    float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                    if (kk + 2 < BK) {
                        na0 = pa[(kk + 2) * LDA_S]; na1 = pa[(kk + 2) * LDA_S + 32];
                        nb0 = pb[(kk + 2) * LDB_S]; nb1 = pb[(kk + 2) * LDB_S + 32];
                    }
                    if (st < 4) la.store_piece(st, Awr, tid);
                    else if (st < 8) lb.store_piece(st - 4, Bwr, tid);
                    else if (st < 12) la.load_fast_piece(st - 8, G.A, G.lda, m0, k2, G.M, tid);
                    else lb.load_fast_piece(st - 12, bb2, G.ldb, n0, kl2, G.N, tid);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
                }
                __syncthreads();
            };
            for (int t = 0; t < nfull; t += 2) {
                body(A0, B0, A1, B1, kb + min(t + 2, nfull - 1) * BK);
                if (t + 1 < nfull) body(A1, B1, A0, B0, kb + min(t + 3, nfull - 1) * BK);
            }
        }
    } else if (nfull > 0) {
        // straight-line pipelined loop: no guards, no branches between the loads (the last iteration
        // re-loads the last full tile instead of branching around the prefetch)
        const int klast = kb + (nfull - 1) * BK;
        int kl;
        const float* bb = bsel(kb, kl);
        la.load_fast(G.A, G.lda, m0, kb, G.M, tid);
        lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
        for (int k0 = kb; k0 <= klast; k0 += BK) {
            la.store(As, tid);
            lb.store(Bs, tid);
            __syncthreads();
            const int kn = min(k0 + BK, klast);
            bb = bsel(kn, kl);
            la.load_fast(G.A, G.lda, m0, kn, G.M, tid);
            lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
            __builtin_amdgcn_sched_barrier(0);   // the prefetch must be in flight BEFORE the MFMA loop, not sunk below it
            __builtin_amdgcn_s_setprio(3);       // waves in their MFMA phase win issue arbitration over co-resident waves that are
            compute_tile();                       // staging / prefetching: +1-2.5 % on every shape (the reverse priority: -1.5 %)
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
    }
    // remaining K (tail of a fast tile, or everything on the guarded path)
    for (int k0 = kb + nfull * BK; k0 < ke; k0 += BK) {
        la.load_guarded(G.A, G.lda, m0, k0, G.M, ke, tid);
        if (G.bchunk > 0) {   // chunked B (bchunk is a multiple of BK, so a tile never straddles two matrices)
            const int w = k0 / G.bchunk, kloc = k0 - w * G.bchunk;
            lb.load_guarded(w == 0 ? G.B : (w == 1 ? G.B1 : G.B2), G.ldb, n0, kloc, G.N, min(G.bchunk, ke - w * G.bchunk), tid);
        } else
        lb.load_guarded(G.B, G.ldb, n0, k0, G.N, ke, tid);
        la.store(As, tid);
        lb.store(Bs, tid);
        __syncthreads();
        compute_tile();
        __syncthreads();
    }

    if (do_colsum && m0 + tid < G.M) ws[G.cs_off + (int64_t)split * G.M + m0 + tid] = csum;

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const int epi = P.epilogue;
    WSI_DROP_SEED(G, epi);
    const bool interior = (m0 + BM <= G.M) && (n0 + BN <= G.N);   // wave-uniform
    float gate_s = 1.f;
    if (!SPLITK && (epi & (WSI_EPI_SCALE_GATE | WSI_EPI_R_1MG)) && G.gate) gate_s = 1.f / (1.f + expf(-(*G.gate)));
    const float r_scale = (epi & WSI_EPI_R_1MG) ? (1.f - gate_s) : 1.f;

    if (interior && (G.flags & 4)) {
        // Fast path: stage each wave's 32x64 accumulator half through LDS (row-major) so that every lane
        // owns 4 consecutive columns: residual loads and C stores become 16-byte accesses, 256 B contiguous
        // per 16 lanes, instead of 64 scattered 4-byte accesses per lane.
        float* wbuf = smem + wave * (32 * 64);
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
                const int row = m0 + wm * 64 + i * 32 + rr;
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
                }
                *reinterpret_cast<float4*>(c) = x;
            }
            __syncthreads();
        }
        return;
    }

    // Guarded scalar path (edge tiles, unaligned C / R).
    if (SPLITK) {
        float* wsp = ws + G.ws_off + (int64_t)split * G.M * G.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < G.M && col < G.N) wsp[(int64_t)row * G.N + col] = acc[i][j][r];
                }
        }
        return;
    }
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
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
            }
        }
    }
}

// Sum the split-K slabs in slab order (deterministic) into C (ReduceParams: gemm_common.h).  VEC: every group's slabs, C and
// row pitch take 16-byte accesses (N % 4 == 0): one thread sums four adjacent columns - a quarter of the threads, each with
// four 16-byte loads in flight (the scalar form ran at 0.4 TB/s on the 200 x 200 gradients of the HGT configuration, where
// sixteen of these launches were 0.76 ms of a 15.6 ms step).
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceParams P) {
    // column-sum partials (tiny): block 0 .. handles them with a plain strided loop
    for (int gi = 0; gi < P.ngroups; ++gi) {
        const ReduceDesc& G = P.g[gi];
        if (!G.cs_out) continue;
        float gs = 1.f;
        if ((P.epilogue & WSI_EPI_SCALE_GATE) && G.gate) gs = 1.f / (1.f + expf(-(*G.gate)));
        for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < G.M; m += (int64_t)gridDim.x * 256) {
            float s = 0.f;
            for (int sp = 0; sp < G.splits; ++sp) s += G.cs_ws[(int64_t)sp * G.M + m];
            s *= gs;
            if (P.epilogue & WSI_EPI_ACCUMULATE) s += G.cs_out[m];
            G.cs_out[m] = s;
        }
    }
    constexpr int W = VEC ? 4 : 1;
    const int64_t total = P.total / W;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        int gi = 0;
        for (int i = 1; i < P.ngroups; ++i) gi = (id * W >= P.g[i].start) ? i : gi;
        const ReduceDesc& G = P.g[gi];
        const int64_t loc = id * W - G.start;
        const int64_t mn = (int64_t)G.M * G.N;
        const int row = (int)(loc / G.N), col = (int)(loc - (int64_t)row * G.N);
        float gs = 1.f;
        if ((P.epilogue & WSI_EPI_SCALE_GATE) && G.gate) gs = 1.f / (1.f + expf(-(*G.gate)));
        float* c = G.C + (int64_t)row * G.ldc + col;
        if constexpr (VEC) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int sp = 0;
            for (; sp + 4 <= G.splits; sp += 4) {       // four slabs in flight; still summed in slab order
                const float4 a = *reinterpret_cast<const float4*>(G.ws + (int64_t)sp * mn + loc);
                const float4 b = *reinterpret_cast<const float4*>(G.ws + (int64_t)(sp + 1) * mn + loc);
                const float4 c2 = *reinterpret_cast<const float4*>(G.ws + (int64_t)(sp + 2) * mn + loc);
                const float4 d = *reinterpret_cast<const float4*>(G.ws + (int64_t)(sp + 3) * mn + loc);
                s.x = (((s.x + a.x) + b.x) + c2.x) + d.x; s.y = (((s.y + a.y) + b.y) + c2.y) + d.y;
                s.z = (((s.z + a.z) + b.z) + c2.z) + d.z; s.w = (((s.w + a.w) + b.w) + c2.w) + d.w;
            }
            for (; sp < G.splits; ++sp) {
                const float4 a = *reinterpret_cast<const float4*>(G.ws + (int64_t)sp * mn + loc);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            if ((P.epilogue & WSI_EPI_SCALE_GATE) && G.gate) { s.x *= gs; s.y *= gs; s.z *= gs; s.w *= gs; }
            if (P.epilogue & WSI_EPI_ACCUMULATE) {
                const float4 o = *reinterpret_cast<const float4*>(c);
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(c) = s;
        } else {
            float s = 0.f;
            int sp = 0;
            for (; sp + 4 <= G.splits; sp += 4) {
                const float a = G.ws[(int64_t)sp * mn + loc], b = G.ws[(int64_t)(sp + 1) * mn + loc];
                const float c2 = G.ws[(int64_t)(sp + 2) * mn + loc], d = G.ws[(int64_t)(sp + 3) * mn + loc];
                s = (((s + a) + b) + c2) + d;
            }
            for (; sp < G.splits; ++sp) s += G.ws[(int64_t)sp * mn + loc];
            if ((P.epilogue & WSI_EPI_SCALE_GATE) && G.gate) s *= gs;
            if (P.epilogue & WSI_EPI_ACCUMULATE) s += *c;
            *c = s;
        }
    }
}

void launch_splitk_reduce(const ReduceParams& RP, hipStream_t st) {
    bool vec = true;
    for (int i = 0; i < RP.ngroups; ++i) {
        const ReduceDesc& G = RP.g[i];
        vec = vec && (G.N % 4 == 0) && (G.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(G.C) & 15) == 0) &&
              ((reinterpret_cast<uintptr_t>(G.ws) & 15) == 0) && (G.start % 4 == 0);
    }
    int rb = (int)((RP.total / (vec ? 4 : 1) + 255) / 256);
    if (rb > 2048) rb = 2048;
    if (rb < 1) rb = 1;
    if (vec) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(rb), dim3(256), 0, st, RP);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(rb), dim3(256), 0, st, RP);
}

// ------------------------------------------------------------------------------------------------
// Skinny launches: the classifier head works on one row per GRAPH (M = batch size = 8, or 24 for the per-type Linear(512, 256);
// its weight gradients reduce over those 8 rows).  On the 128 x 128 MFMA tile such a GEMM is ONE workgroup walking K alone -
// 12-39 us each, thirteen of them per step (0.27 ms of an 8.4 ms step).  Here every output element gets its own thread (or its
// own wave, for the K-contiguous NT form) and the whole launch is a few microseconds; plain IEEE fp32 FMA chains in a fixed
// order, whatever the precision argument (exact fp32 is within every mode's error class); deterministic.
//   NT  C[m, n] = sum_k A[m, k] B[n, k]     one wave per n (64 lanes along k, coalesced in both operands), all m <= 32 at once
//   NN  C[m, n] = sum_k A[m, k] B[k, n]     the same with B walked down a column
//   TN  C[m, n] = sum_k A[k, m] B[k, n]     one thread per (m, n), K <= 32; column sums of A on the side (bias gradients)
// Epilogues: BIAS (NT / NN), ACCUMULATE, SCALE_GATE; anything else, or a request for row scales, takes the tiled kernels.
constexpr int SKINNY_M = 32, SKINNY_K = 32;
struct SkinnyDesc { const float* A; const float* B; float* C; const float* bias; const float* gate; float* cs_out; const float* R; int64_t lda, ldb, ldc, ldr; int32_t M, N, K, units; };
struct SkinnyParams { SkinnyDesc g[WSI_GEMM_MAX_GROUPS]; int32_t ngroups, epilogue; };

// the TN form: one thread per (m, n)
__device__ __forceinline__ void skinny_tn(const SkinnyDesc& G, int epilogue, int bx) {
    const int loc = bx * 256 + (int)threadIdx.x;
    if (loc >= G.units) return;
    float gs = 1.f;
    if ((epilogue & WSI_EPI_SCALE_GATE) && G.gate) gs = 1.f / (1.f + expf(-(*G.gate)));
    const int m = loc / G.N, n = loc - m * G.N;
    float s = 0.f, cs = 0.f;
    for (int k = 0; k < G.K; ++k) {
        const float a = G.A[(int64_t)k * G.lda + m];
        s = fmaf(a, G.B[(int64_t)k * G.ldb + n], s);
        cs += a;
    }
    s *= gs;
    float* c = G.C + (int64_t)m * G.ldc + n;
    if (epilogue & WSI_EPI_ACCUMULATE) s += *c;
    *c = s;
    if (G.cs_out && n == 0) {
        cs *= gs;
        if (epilogue & WSI_EPI_ACCUMULATE) cs += G.cs_out[m];
        G.cs_out[m] = cs;
    }
}

// NT / NN: one wave per output column n, its 64 lanes along k (NT: both operands coalesced; NN: B[k, n] is a strided
// column, the rows it touches are shared with the neighbouring columns' waves through the L1 / L2)
template <int OP, int MB>      // MB: accumulators per thread (8 / 16 / 32 >= the group's M)
__device__ __forceinline__ void skinny_rows(const SkinnyDesc& G, int epilogue, int bx) {
    const int loc = bx * 4 + (int)(threadIdx.x >> 6);
    if (loc >= G.units) return;
    float gate_s = 1.f;
    if ((epilogue & (WSI_EPI_SCALE_GATE | WSI_EPI_R_1MG)) && G.gate) gate_s = 1.f / (1.f + expf(-(*G.gate)));
    const float gs = (epilogue & WSI_EPI_SCALE_GATE) ? gate_s : 1.f;
    const float r_scale = (epilogue & WSI_EPI_R_1MG) ? 1.f - gate_s : 1.f;
    const int n = loc, lane = threadIdx.x & 63;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    const float* b = (OP == WSI_GEMM_NT) ? G.B + (int64_t)n * G.ldb : G.B + n;
    const int64_t bstep = (OP == WSI_GEMM_NT) ? 1 : G.ldb;
    // rows beyond G.M re-read the last row (their sums are dropped below): no branch inside the loop, so that all the loads
    // of an unrolled batch are in flight together (a conditional load + fma per row compiles to load, wait, fma - serial)
    const float* arow[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) arow[m] = G.A + (int64_t)(m < G.M ? m : G.M - 1) * G.lda;
#pragma unroll 4
    for (int k = lane; k < G.K; k += 64) {
        const float bk = b[(int64_t)k * bstep];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = fmaf(arow[m][k], bk, acc[m]);
    }
    const float bv = ((epilogue & WSI_EPI_BIAS) && G.bias) ? G.bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
        if (m < G.M) {                       // (G.M is uniform: every lane takes part in the butterfly)
            float x = acc[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if (lane == 0) {
                x = (x + bv) * gs;
                if ((epilogue & WSI_EPI_ADD_R) && G.R) x = fmaf(G.R[(int64_t)m * G.ldr + n], r_scale, x);
                float* c = G.C + (int64_t)m * G.ldc + n;
                if (epilogue & WSI_EPI_ACCUMULATE) x += *c;
                *c = x;
            }
        }
}

// grid.y = group (the descriptor is workgroup-uniform: A[m, k] becomes scalar loads), grid.x covers the units of the largest group
template <int OP, int MB>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkinnyParams P) {
    const SkinnyDesc& G = P.g[blockIdx.y];
    if constexpr (OP == WSI_GEMM_TN) skinny_tn(G, P.epilogue, (int)blockIdx.x);
    else skinny_rows<OP, MB>(G, P.epilogue, (int)blockIdx.x);
}

// A small Linear's two gradients in ONE launch (wsi_gemm_small_pair): groups [0, n_rows) are dX = dY W (the NN form above), the others
// dW = dY^T X with the bias gradient on the side (the TN form) - both read dY, neither reads the other's result.
constexpr int SKINNY_PAIR_GROUPS = 16;
struct SkinnyPairParams { SkinnyDesc g[SKINNY_PAIR_GROUPS]; int32_t n_rows, n_tn, epi_rows, epi_tn; };

template <int MB>
__global__ __launch_bounds__(256) void gemm_skinny_pair_kernel(const SkinnyPairParams P) {
    const SkinnyDesc& G = P.g[blockIdx.y];
    if ((int)blockIdx.y < P.n_rows) skinny_rows<WSI_GEMM_NN, MB>(G, P.epi_rows, (int)blockIdx.x);
    else skinny_tn(G, P.epi_tn, (int)blockIdx.x);
}

// true (and the launch done) when every group of the call is skinny and asks for nothing the kernel above does not do
static bool launch_skinny(int32_t op, int32_t epilogue, const wsi_gemm_group_t* groups, int32_t ngroups, hipStream_t st) {
    if (epilogue & ~(WSI_EPI_BIAS | WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE | (op == WSI_GEMM_TN ? 0 : (WSI_EPI_ADD_R | WSI_EPI_R_1MG)))) return false;
    SkinnyParams P;
    P.ngroups = 0; P.epilogue = epilogue;
    int32_t maxu = 0;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.M <= 0 || s.N <= 0) continue;
        if (s.K <= 0 || !s.A || !s.B || !s.C || s.c_absmax || s.b_chunk) return false;
        if (op == WSI_GEMM_TN ? (s.K > SKINNY_K) : (s.M > SKINNY_M)) return false;
        if (op != WSI_GEMM_TN && s.colsum_out) return false;
        const int64_t units = (op == WSI_GEMM_TN) ? (int64_t)s.M * s.N : s.N;
        if (units > (1 << 24)) return false;
        SkinnyDesc& d = P.g[P.ngroups++];
        d.A = s.A; d.B = s.B; d.C = s.C; d.bias = s.bias; d.gate = s.gate; d.cs_out = s.colsum_out; d.R = s.R;
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.ldr = s.ldr; d.M = s.M; d.N = s.N; d.K = s.K; d.units = (int32_t)units;
        if (d.units > maxu) maxu = d.units;
    }
    if (P.ngroups == 0) return false;
    int maxm = 0;
    for (int i = 0; i < P.ngroups; ++i) maxm = P.g[i].M > maxm ? P.g[i].M : maxm;
    const dim3 gw((maxu + 3) / 4, P.ngroups), b(256);
    if (op == WSI_GEMM_TN) hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_TN, 1>), dim3((maxu + 255) / 256, P.ngroups), b, 0, st, P);
    else if (op == WSI_GEMM_NT) {
        if (maxm <= 8) hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NT, 8>), gw, b, 0, st, P);
        else if (maxm <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NT, 16>), gw, b, 0, st, P);
        else hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NT, 32>), gw, b, 0, st, P);
    } else {
        if (maxm <= 8) hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NN, 8>), gw, b, 0, st, P);
        else if (maxm <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NN, 16>), gw, b, 0, st, P);
        else hipLaunchKernelGGL((gemm_skinny_kernel<WSI_GEMM_NN, 32>), gw, b, 0, st, P);
    }
    return true;
}

static inline bool vec_ok(const void* p, int64_t ld) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0);
}

// TN split planning shared by workspace query and launch: one chunk length (multiple of BK) for all
// groups, the smallest for which the whole launch is at most one residency round (3 workgroups per CU).
// WSI_GEMM_PIPE=1 selects the single-barrier software-pipelined kernel (2 workgroups/CU).  Interleaved A/B runs on
// one MI355X: within +-3 % of the default phase-structured kernel (3 workgroups/CU) on the bench shapes, +20 % at
// 4096^3 - kept selectable for measurements, not the default.
static bool gemm_pipe() {
    static const bool on = [] { const char* pv = knob("WSI_GEMM_PIPE"); return pv && pv[0] == '1'; }();
    return on;
}

static int32_t plan_kchunk(const wsi_gemm_group_t* g, int32_t ng, int32_t precision) {
    const int64_t target_blocks = ((gemm_pipe() || precision != WSI_GEMM_FP32) ? 2 : 3) * 256;   // one residency round
    int64_t work = 0, maxk = 0;
    for (int i = 0; i < ng; ++i) {
        if (g[i].M <= 0 || g[i].N <= 0) continue;
        const int64_t tmn = (int64_t)((g[i].M + BM - 1) / BM) * ((g[i].N + BN - 1) / BN);
        work += tmn * g[i].K;
        if (g[i].K > maxk) maxk = g[i].K;
    }
    int64_t kc = (work + target_blocks - 1) / target_blocks;
    kc = ((kc + BK - 1) / BK) * BK;
    if (kc < 8 * BK) kc = 8 * BK;
    for (;;) {   // per-group ceil() can overshoot the target: grow the chunk until it fits
        int64_t blocks = 0;
        for (int i = 0; i < ng; ++i) {
            if (g[i].M <= 0 || g[i].N <= 0) continue;
            const int64_t tmn = (int64_t)((g[i].M + BM - 1) / BM) * ((g[i].N + BN - 1) / BN);
            blocks += tmn * (g[i].K > 0 ? (g[i].K + kc - 1) / kc : 1);
        }
        if (blocks <= target_blocks || kc >= maxk) break;
        kc += BK;
    }
    return (int32_t)kc;
}

}  // namespace wsi

using namespace wsi;

// fp16x3: words of absmax bits (and packed B planes) appended to the workspace (after the TN slabs)
static int64_t scale_words(int32_t op, const wsi_gemm_group_t* groups, int32_t ngroups) {
    int64_t w = 0;
    for (int i = 0; i < ngroups; ++i)
        if (groups[i].M > 0 && groups[i].N > 0) w += fp16x3_words(op, groups[i].M, groups[i].N, groups[i].K);
    return w;
}

// the kernel family a launch runs on: WSI_GEMM_AUTO picks the scaled-fp16 kernel where it pays.  NT / NN, measured on one MI355X with the
// pre-pass inside the launch (no row scales, no packed weights supplied: the worst case; tools/auto_threshold_probe.py, profiles/r05_auto_threshold.json):
// K = 128 never; K = 256 from 4 GFLOP (N = 768; N = 256: break-even at 5-10 GFLOP); K = 384 / 512 from 3-5 GFLOP; K = 1024 always - and inside a model,
// where the producers leave the row scales and the optimizer step packs the weights, earlier (HGT at hidden 200 -> 256-wide: projection time 2.83 -> 2.39 ms
// per step, HGT + ASAP 6.78 -> 5.78, HEATNet2 at hidden 256 1.40 -> 1.21 with every launch on the scaled kernel).  Hence, for LARGE batches (below), >= 5 GFLOP
// and every K >= 256; otherwise the rule of rounds 2-4: 12 GFLOP and K >= 384.
static int32_t kernel_precision(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups) {
    if (precision == WSI_GEMM_FP32 || precision == WSI_GEMM_BF16X6) return precision;
    if (precision != WSI_GEMM_FP16X3 && precision != WSI_GEMM_AUTO) return -1;
    if (precision == WSI_GEMM_FP16X3) return WSI_GEMM_FP16X3;
    double flops = 0.0;
    int32_t kmin = INT32_MAX, kmax = 0, mmax = 0, wmin = INT32_MAX;
    for (int i = 0; groups && i < ngroups; ++i) {
        const wsi_gemm_group_t& g = groups[i];
        if (g.M <= 0 || g.N <= 0) continue;
        flops += 2.0 * g.M * g.N * g.K;
        kmin = g.K < kmin ? g.K : kmin;
        kmax = g.K > kmax ? g.K : kmax;
        mmax = g.M > mmax ? g.M : mmax;
        wmin = g.M < wmin ? g.M : wmin;
        wmin = g.N < wmin ? g.N : wmin;
    }
    // A LARGE batch (some group of >= 24576 rows, i.e. ~50 k nodes and more per step) runs GPU-bound and takes the scaled kernels from the sizes where
    // they win on the GPU; a small one is bound by the host's launches, and every launch the scale exchange adds (row-scale fills, statistics passes) costs
    // it more than the kernel saves: there the thresholds of rounds 2-4 stay.  Measured, same box, old / extended rule for every batch:
    // HGT 4 x 20 k nodes 4.57 -> 4.2 ms per step, HGT + ASAP 11.9 -> 11.1; HEATNet2 8 x 5 k nodes 2.0 -> 2.4-2.9; two 10 k-node slides per step loader-fed 3.2 -> 4.3.
    const bool large = (op == WSI_GEMM_TN ? kmax : mmax) >= 24576;
    if (op == WSI_GEMM_TN) {
        // weight gradients (K = the rows of the operands): the column-scaled kernel of gemm_tn16.hip where its statistics pass and its 256 x 128 tiles pay.
        // With the pass inside the launch (the worst case; tools/tn_threshold_probe.py, profiles/r05_tn_threshold.json): 512- and 1024-wide outputs from
        // 8 GFLOP, six 256 x 256 groups from 12, 128-wide outputs NEVER (1.5-1.7 x slower: half-empty tiles); inside a model, where the producers leave the
        // statistics, from 3-4 GFLOP (HGT: weight gradients 0.77 -> 0.60 ms per step, HGT + ASAP 1.90 -> 1.43).
        // Two ways in: the rule of rounds 2-4 (>= 30 GFLOP per launch, every group >= 2048 rows) for ANY batch, and - on a large batch only - from 4 GFLOP
        // when every group has >= 8192 rows.  (Round 5 applied the 8192-row condition to every launch of a large batch: the reference's real schema - six
        // node types of 27 200 ... 3 200 rows, 126 GFLOP per K|Q|V weight-gradient launch - fell back to bf16x6: 2.26 instead of 1.19 ms per step.)
        double tn_min = 4e9, tn_any = 3e10;
#ifdef WSI_ABLATE
        if (const char* v = knob("WSI_TN_AUTO_GFLOP")) tn_min = tn_any = atof(v) * 1e9;    // measurement build: where should auto switch the weight gradients
#endif
        const bool by_size = flops >= tn_any && kmin >= 2048;
        const bool by_batch = large && flops >= tn_min && kmin >= 8192;
        return (wmin >= 192 && (by_size || by_batch)) ? WSI_GEMM_FP16X3 : WSI_GEMM_BF16X6;
    }
    if (large) return (flops >= 5e9 && kmin >= 256) ? WSI_GEMM_FP16X3 : WSI_GEMM_BF16X6;
    return (flops >= 12e9 && kmin >= 384) ? WSI_GEMM_FP16X3 : WSI_GEMM_BF16X6;
}

extern "C" int32_t wsi_gemm_kernel_precision(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups) {
    return kernel_precision(op, precision, groups, ngroups);
}

extern "C" int64_t wsi_gemm_packed_b_bytes(int32_t N, int32_t K) { return (N > 0 && K > 0) ? fp16x3_packed_b_bytes(N, K) : 0; }

extern "C" int wsi_gemm_pack_b(int32_t op, const wsi_gemm_group_t* groups, int32_t ngroups, void* stream) {
    if (op != WSI_GEMM_NT && op != WSI_GEMM_NN) { set_error("gemm_pack_b: NT or NN"); return WSI_EINVAL; }
    if (ngroups < 0 || ngroups > WSI_GEMM_MAX_GROUPS || (ngroups > 0 && !groups)) { set_error("gemm_pack_b: bad group table"); return WSI_EINVAL; }
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.N < 0 || s.K < 0) { set_error("gemm_pack_b: negative dimension in group %d", i); return WSI_EINVAL; }
        if (s.N == 0 || s.K == 0) continue;
        if (!s.B || !s.b_packed || (reinterpret_cast<uintptr_t>(s.b_packed) & 15)) { set_error("gemm_pack_b: group %d needs B and a 16-byte aligned b_packed", i); return WSI_EINVAL; }
        if (s.b_chunk != 0 && (op != WSI_GEMM_NN || s.b_chunk < 0 || s.b_chunk % BK != 0 || (int64_t)3 * s.b_chunk < s.K ||
                               (s.K > s.b_chunk && !s.B1) || (s.K > 2 * (int64_t)s.b_chunk && !s.B2))) {
            set_error("gemm_pack_b: bad b_chunk/B1/B2 in group %d", i); return WSI_EINVAL; }
    }
    return launch_pack_b(op, groups, ngroups, (hipStream_t)stream);
}

// does the launch run on gemm_fp16x3g_kernel (the one NT / NN kernel that leaves column statistics)?  Mirrors gemm_emu16.hip::fp16x3_dma_ok
extern "C" int32_t wsi_gemm_writes_colstats(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups) {
    if (op == WSI_GEMM_TN || !groups || ngroups <= 0 || ngroups > WSI_GEMM_MAX_GROUPS) return 0;
    if (kernel_precision(op, precision, groups, ngroups) != WSI_GEMM_FP16X3) return 0;
    const char* v = knob("WSI_GEMM_F16_KERNEL");
    if (v && v[0] == 'w') return 0;
    bool any = false;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.M <= 0 || s.N <= 0) continue;
        if (s.K <= 0 || s.K % 32 != 0 || !vec_ok(s.A, s.lda) || (int64_t)s.M * s.lda * 4 >= ((int64_t)1 << 31)) return 0;
        any = true;
    }
    if (!any) return 0;
    {   // (skinny launches take a kernel of their own: same test as wsi_gemm_grouped)
        bool small = true;
        for (int i = 0; i < ngroups; ++i) small = small && groups[i].M <= 32;
        if (small) return 0;
    }
    return 1;
}

extern "C" int64_t wsi_gemm_workspace_bytes(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups) {
    if (!groups || ngroups <= 0) return 0;
    const int32_t kp = kernel_precision(op, precision, groups, ngroups);
    int64_t floats = 0;
    if (op == WSI_GEMM_TN && kp == WSI_GEMM_FP16X3) return tn16_workspace_floats(groups, ngroups) * 4;
    if (op == WSI_GEMM_TN) {
        const int32_t kc = plan_kchunk(groups, ngroups, kp);
        for (int i = 0; i < ngroups; ++i) {
            if (groups[i].M <= 0 || groups[i].N <= 0) continue;
            const int64_t splits = groups[i].K > 0 ? (groups[i].K + kc - 1) / kc : 1;
            floats += splits * (int64_t)groups[i].M * groups[i].N;
            if (groups[i].colsum_out) floats += splits * (int64_t)((groups[i].M + 3) / 4 * 4);
        }
    }
    if (kp == WSI_GEMM_FP16X3) floats = ((floats + 3) & ~(int64_t)3) + scale_words(op, groups, ngroups);
    return floats * 4;
}

extern "C" int wsi_gemm_grouped(int32_t op, int32_t epilogue, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    const int32_t kp = kernel_precision(op, precision, groups, ngroups);
    if (kp < 0) { set_error("gemm: unknown precision mode %d", precision); return WSI_EINVAL; }
    if (ngroups < 0 || (ngroups > 0 && !groups)) { set_error("gemm: bad group table"); return WSI_EINVAL; }
    if (ngroups > WSI_GEMM_MAX_GROUPS) { set_error("gemm: %d groups > WSI_GEMM_MAX_GROUPS", ngroups); return WSI_EINVAL; }
    if (op < 0 || op > 2) { set_error("gemm: unknown op %d", op); return WSI_EINVAL; }
    if (epilogue & ~(WSI_EPI_BIAS | WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE | WSI_EPI_GELU | WSI_EPI_ADD_R | WSI_EPI_R_1MG | WSI_EPI_MUL_M | WSI_EPI_BACKGROUND | WSI_EPI_DROPOUT)) {
        set_error("gemm: unknown epilogue bits 0x%x", epilogue); return WSI_EINVAL; }
    if ((epilogue & WSI_EPI_DROPOUT) && (op == WSI_GEMM_TN || (epilogue & WSI_EPI_MUL_M))) { set_error("gemm: WSI_EPI_DROPOUT is for NT / NN launches without MUL_M"); return WSI_EINVAL; }
    if (op == WSI_GEMM_TN && (epilogue & ~(WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE | WSI_EPI_BACKGROUND))) { set_error("gemm: TN accepts only ACCUMULATE and SCALE_GATE"); return WSI_EINVAL; }
    if (op != WSI_GEMM_TN && (epilogue & WSI_EPI_BACKGROUND)) { set_error("gemm: WSI_EPI_BACKGROUND is a hint for TN launches"); return WSI_EINVAL; }
    const bool background = (epilogue & WSI_EPI_BACKGROUND) != 0;
    epilogue &= ~WSI_EPI_BACKGROUND;
    hipStream_t st = (hipStream_t)stream;
    {   // skinny launches (classifier head): every group validated exactly as below, then the dedicated kernel
        bool ok = ngroups > 0;
        for (int i = 0; ok && i < ngroups; ++i) {
            const wsi_gemm_group_t& s = groups[i];
            ok = s.M >= 0 && s.N >= 0 && s.K > 0 && (s.M == 0 || s.N == 0 || (s.A && s.B && s.C));
        }
        static const bool skinny_on = [] { const char* v = knob("WSI_GEMM_SKINNY"); return !(v && v[0] == '0'); }();
        if (ok && skinny_on && launch_skinny(op, epilogue, groups, ngroups, st)) return check_launch("gemm_skinny");
    }
    if (op == WSI_GEMM_TN && kp == WSI_GEMM_FP16X3) {        // the column-scaled weight-gradient kernel: its own tiles, plan and pre-pass
        for (int i = 0; i < ngroups; ++i) {
            const wsi_gemm_group_t& s = groups[i];
            if (s.M < 0 || s.N < 0 || s.K < 0) { set_error("gemm: negative dimension in group %d", i); return WSI_EINVAL; }
            if (s.M == 0 || s.N == 0) continue;
            if (!s.C || (s.K > 0 && (!s.A || !s.B))) { set_error("gemm: null pointer in group %d", i); return WSI_EINVAL; }
            if (s.b_chunk != 0) { set_error("gemm: b_chunk is for NN launches (group %d)", i); return WSI_EINVAL; }
        }
        return launch_gemm_tn16(epilogue, groups, ngroups, (float*)workspace, workspace_bytes, st);
    }
    const bool pipe = gemm_pipe();
    // FP16X3 covers NT / NN (the weights are the packed operand); the weight gradients (TN: both operands are activations,
    // scales would be per column over all nodes) run as bf16x6 in that mode: measured faster than a scaled fp16 TN
    const bool emu = kp == WSI_GEMM_BF16X6;
    const bool f16 = kp == WSI_GEMM_FP16X3;
    const bool scales = precision == WSI_GEMM_FP16X3 || precision == WSI_GEMM_AUTO;    // c_absmax is written by either kernel
    // experiment knob (read once): extra dynamic LDS bytes per workgroup, to cap residency in A/B runs
    static const unsigned lds_knob = [] { const char* v = knob("WSI_GEMM_LDS_PAD"); return v ? (unsigned)atoi(v) : 0u; }();
    // WSI_EPI_BACKGROUND: dynamic LDS nobody touches, sized so that a second workgroup of the launch no longer fits on a CU (the emulation kernels
    // hold 72 KB of the 160, the exact-fp32 one 33 KB)
    const unsigned lds_pad = background ? (kp == WSI_GEMM_FP32 ? 64u * 1024u : 16u * 1024u) : lds_knob;

    GemmParams P;
    ReduceParams RP;
    P.ngroups = 0; P.epilogue = epilogue; P.plain_stores = 0;
    RP.ngroups = 0; RP.epilogue = epilogue;
    const int32_t kc = (op == WSI_GEMM_TN) ? plan_kchunk(groups, ngroups, kp) : 0;
    int32_t tiles = 0;
    int64_t ws_floats = 0, red_total = 0;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_group_t& s = groups[i];
        if (s.M < 0 || s.N < 0 || s.K < 0) { set_error("gemm: negative dimension in group %d", i); return WSI_EINVAL; }
        if (s.M == 0 || s.N == 0) continue;
        if (!s.C || (s.K > 0 && (!s.A || !s.B))) { set_error("gemm: null pointer in group %d", i); return WSI_EINVAL; }
        if ((epilogue & WSI_EPI_ADD_R) && !s.R) { set_error("gemm: ADD_R needs R (group %d)", i); return WSI_EINVAL; }
        if ((epilogue & WSI_EPI_MUL_M) && !s.Mm) { set_error("gemm: MUL_M needs Mm (group %d)", i); return WSI_EINVAL; }
        if (s.b_chunk != 0) {
            if (op != WSI_GEMM_NN || s.b_chunk < 0 || s.b_chunk % BK != 0 || (int64_t)3 * s.b_chunk < s.K ||
                (s.K > s.b_chunk && !s.B1) || (s.K > 2 * (int64_t)s.b_chunk && !s.B2)) {
                set_error("gemm: bad b_chunk/B1/B2 in group %d (NN only, multiple of %d, at most 3 chunks)", i, BK); return WSI_EINVAL; }
        }
        GroupDesc& d = P.g[P.ngroups];
        d.A = s.A; d.B = s.B; d.C = s.C; d.bias = s.bias; d.R = s.R; d.gate = s.gate;
        d.B1 = s.B1; d.B2 = s.B2; d.bchunk = s.b_chunk; d.ea_off = d.eb_off = -1;
        d.Mm = s.Mm; d.ldm = s.ldm;
        d.drop_seed = s.drop_seed; d.drop_thr = s.drop_threshold; d.drop_scale = s.drop_scale; d.drop_row0 = s.drop_row0; d.drop_col0 = s.drop_col0;
        d.drop_pairs = (uint32_t)((s.drop_cols + 1) / 2);
        d.drop_seed_base = s.drop_seed_base;
        d.b_bits = (f16 && op != WSI_GEMM_TN) ? reinterpret_cast<const uint32_t*>(s.b_packed) : nullptr;
        if (d.b_bits && (reinterpret_cast<uintptr_t>(d.b_bits) & 15)) { set_error("gemm: b_packed of group %d must be 16-byte aligned", i); return WSI_EINVAL; }
        d.c_colmax = (f16 && op != WSI_GEMM_TN) ? s.c_colmax : nullptr; d.c_colsum = d.c_colmax ? s.c_colsum : nullptr; d.c_col_ld = s.c_col_ld;
        if ((epilogue & WSI_EPI_DROPOUT) && (s.drop_threshold > 65536u || s.drop_cols <= 0 || s.drop_row0 < 0 || s.drop_col0 < 0 || s.drop_col0 % 4 != 0 || s.drop_col0 + s.N > s.drop_cols)) {
            set_error("gemm: bad dropout fields in group %d (threshold <= 65536, the group's columns inside the masked tensor's, drop_col0 %% 4 == 0)", i); return WSI_EINVAL; }
        d.a_absmax = f16 ? s.a_absmax : nullptr; d.c_absmax = (scales && op != WSI_GEMM_TN) ? s.c_absmax : nullptr;
        d.a_parts = d.a_absmax ? s.a_absmax_parts : 1; d.c_parts = s.c_absmax_parts; d.c_first = s.c_absmax_first;
        if (d.a_absmax && (s.a_absmax_parts < 1 || s.a_absmax_parts > 64)) { set_error("gemm: a_absmax_parts = %d of group %d (1..64)", s.a_absmax_parts, i); return WSI_EINVAL; }
        if (d.c_absmax && (s.c_absmax_first < 0 || s.c_absmax_first + 2 * ((s.N + BN - 1) / BN) > s.c_absmax_parts)) {
            set_error("gemm: c_absmax slots [%d, %d) of group %d exceed c_absmax_parts = %d", s.c_absmax_first, s.c_absmax_first + 2 * ((s.N + BN - 1) / BN), i, s.c_absmax_parts);
            return WSI_EINVAL; }
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.ldr = s.ldr;
        d.M = s.M; d.N = s.N; d.K = s.K;
        const int tmm = (s.M + BM - 1) / BM, tnn = (s.N + BN - 1) / BN;
        d.tiles_n = tnn; d.tiles_mn = tmm * tnn;
        d.tile_start = tiles;
        const bool bv = vec_ok(s.B, s.ldb) && (!s.b_chunk || ((!s.B1 || vec_ok(s.B1, s.ldb)) && (!s.B2 || vec_ok(s.B2, s.ldb))));
        bool cv;
        if (op == WSI_GEMM_TN) cv = (s.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(workspace) & 15) == 0) && (ws_floats % 4 == 0);
        else cv = vec_ok(s.C, s.ldc) && (!(epilogue & WSI_EPI_ADD_R) || vec_ok(s.R, s.ldr)) && (!(epilogue & WSI_EPI_MUL_M) || vec_ok(s.Mm, s.ldm));
        d.flags = (vec_ok(s.A, s.lda) ? 1 : 0) | (bv ? 2 : 0) | (cv ? 4 : 0);
        d.ws_off = 0; d.kchunk = s.K; d.cs_off = -1;
        if (op == WSI_GEMM_TN) {
            const int32_t splits = s.K > 0 ? (s.K + kc - 1) / kc : 1;
            d.kchunk = kc; d.ws_off = ws_floats;
            tiles += d.tiles_mn * splits;
            ReduceDesc& r = RP.g[RP.ngroups++];
            r.ws = (const float*)workspace + ws_floats; r.C = s.C; r.gate = s.gate; r.ldc = s.ldc; r.M = s.M; r.N = s.N; r.pad = 0;
            r.splits = splits; r.start = red_total;
            red_total += (int64_t)s.M * s.N;
            ws_floats += (int64_t)splits * s.M * s.N;
            r.cs_ws = nullptr; r.cs_out = nullptr;
            if (s.colsum_out) {
                d.cs_off = ws_floats;
                r.cs_ws = (const float*)workspace + ws_floats; r.cs_out = s.colsum_out;
                ws_floats += (int64_t)splits * ((s.M + 3) / 4 * 4);
            }
        } else {
            tiles += d.tiles_mn;
        }
        P.ngroups++;
    }
    if (P.ngroups == 0) return WSI_OK;
    P.total_tiles = tiles;
    int64_t e_first = 0, e_words = 0;
    if (f16) {   // absmax bits of every group's A rows / B columns behind the slabs
        e_first = (ws_floats + 3) & ~(int64_t)3;
        for (int i = 0; i < P.ngroups; ++i) e_words += fp16x3_words(op, P.g[i].M, P.g[i].N, P.g[i].K);
        ws_floats = e_first + e_words;
        if (ws_floats >= ((int64_t)1 << 31)) { set_error("gemm fp16x3: workspace of %lld floats exceeds the 2^31 index range", (long long)ws_floats); return WSI_EINVAL; }
        if (!workspace || workspace_bytes < ws_floats * 4) {
            set_error("gemm fp16x3: workspace of %lld bytes needed, %lld given", (long long)(ws_floats * 4), (long long)workspace_bytes);
            return WSI_ENOMEM;
        }
        if (reinterpret_cast<uintptr_t>(workspace) & 15) { set_error("gemm fp16x3: the workspace must be 16-byte aligned"); return WSI_EINVAL; }
    }
    // residency the kernel is compiled for (see gemm_f32_kernel): 4 workgroups/CU when the launch is at most ~5 rounds of
    // them, 3 otherwise (and always for the split-K launches, which are planned as exactly one round of 3/CU)
    static const int res_env = [] { const char* v = knob("WSI_GEMM_RES"); return v ? atoi(v) : 0; }();
    const bool res4 = res_env ? res_env == 4 : (tiles <= 5 * 1024);
    if (op == WSI_GEMM_TN) {
        if (!workspace || workspace_bytes < ws_floats * 4) {
            set_error("gemm TN: workspace of %lld bytes needed, %lld given", (long long)(ws_floats * 4), (long long)workspace_bytes);
            return WSI_ENOMEM;
        }
        if (emu) launch_gemm_bf16x6(op, P, tiles, lds_pad, (float*)workspace, st);
        else if (f16) launch_gemm_fp16x3(op, P, tiles, lds_pad, (float*)workspace, e_first, e_words, st);
        else if (pipe) hipLaunchKernelGGL((gemm_f32_kernel<false, false, true, true>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)workspace);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false, true, false>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)workspace);
        RP.total = red_total;
        launch_splitk_reduce(RP, st);
    } else if (emu) {
        launch_gemm_bf16x6(op, P, tiles, lds_pad, nullptr, st);
    } else if (f16) {
        launch_gemm_fp16x3(op, P, tiles, lds_pad, (float*)workspace, e_first, e_words, st);
    } else if (op == WSI_GEMM_NT) {
        if (pipe) hipLaunchKernelGGL((gemm_f32_kernel<true, true, false, true>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
        else if (res4) hipLaunchKernelGGL((gemm_f32_kernel<true, true, false, false, 4>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, true, false, false>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
    } else {
        if (pipe) hipLaunchKernelGGL((gemm_f32_kernel<true, false, false, true>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
        else if (res4) hipLaunchKernelGGL((gemm_f32_kernel<true, false, false, false, 4>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, false, false, false>), dim3(tiles), dim3(GEMM_THREADS), lds_pad, st, P, (float*)nullptr);
    }
    return check_launch("gemm_f32");
}

// dX and dW (+ db) of small Linear layers in one launch - see include/wsi_hgnn.h
extern "C" int wsi_gemm_small_pair(const wsi_gemm_group_t* dx, int32_t n_dx, int32_t dx_epilogue,
                                   const wsi_gemm_group_t* dw, int32_t n_dw, int32_t dw_epilogue, void* stream) {
    if (n_dx < 0 || n_dw < 0 || n_dx + n_dw > SKINNY_PAIR_GROUPS || (n_dx && !dx) || (n_dw && !dw)) { set_error("gemm_small_pair: bad group count"); return WSI_EINVAL; }
    if ((dx_epilogue & ~(WSI_EPI_BIAS | WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE | WSI_EPI_ADD_R | WSI_EPI_R_1MG)) ||
        (dw_epilogue & ~(WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE))) { set_error("gemm_small_pair: unsupported epilogue"); return WSI_ENOSYS; }
    SkinnyPairParams P;
    P.n_rows = 0; P.n_tn = 0; P.epi_rows = dx_epilogue; P.epi_tn = dw_epilogue;
    int32_t ngroups = 0, maxblocks = 0, maxm = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const wsi_gemm_group_t* gs = pass ? dw : dx;
        const int32_t n = pass ? n_dw : n_dx;
        for (int i = 0; i < n; ++i) {
            const wsi_gemm_group_t& s = gs[i];
            if (s.M <= 0 || s.N <= 0) continue;
            if (s.K <= 0 || !s.A || !s.B || !s.C || s.c_absmax || s.b_chunk || (!pass && s.colsum_out)) { set_error("gemm_small_pair: group %d: unsupported field", i); return WSI_EINVAL; }
            if (pass ? (s.K > SKINNY_K) : (s.M > SKINNY_M)) { set_error("gemm_small_pair: group %d is not small (dX: M <= %d rows, dW: K <= %d rows)", i, SKINNY_M, SKINNY_K); return WSI_ENOSYS; }
            const int64_t units = pass ? (int64_t)s.M * s.N : s.N;
            if (units > (1 << 24)) { set_error("gemm_small_pair: group too wide"); return WSI_ENOSYS; }
            SkinnyDesc& d = P.g[ngroups++];
            d.A = s.A; d.B = s.B; d.C = s.C; d.bias = s.bias; d.gate = s.gate; d.cs_out = s.colsum_out; d.R = s.R;
            d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.ldr = s.ldr; d.M = s.M; d.N = s.N; d.K = s.K; d.units = (int32_t)units;
            const int32_t blocks = pass ? (int32_t)((units + 255) / 256) : (int32_t)((units + 3) / 4);
            if (blocks > maxblocks) maxblocks = blocks;
            if (!pass && s.M > maxm) maxm = s.M;
            if (pass) ++P.n_tn; else ++P.n_rows;
        }
    }
    if (ngroups == 0) return WSI_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(maxblocks, ngroups), b(256);
    if (maxm <= 8) hipLaunchKernelGGL((gemm_skinny_pair_kernel<8>), grid, b, 0, st, P);
    else if (maxm <= 16) hipLaunchKernelGGL((gemm_skinny_pair_kernel<16>), grid, b, 0, st, P);
    else hipLaunchKernelGGL((gemm_skinny_pair_kernel<32>), grid, b, 0, st, P);
    return check_launch("gemm_small_pair");
}
