// fp32 GEMM emulated on the bf16 matrix cores ("bf16x6"): every fp32 operand x is split exactly into three bf16
// terms x = x0 + x1 + x2 (8+8+8 mantissa bits), and x*y is accumulated in fp32 from the six products whose magnitude
// is >= 2^-16 |x y|  (x0y0, x0y1, x1y0, x0y2, x2y0, x1y1); the dropped terms are <= 2^-23 |x y|, i.e. below one fp32
// rounding of the product.  bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so the result carries fp32-class error (tests/test_kernels_gpu.py::test_gemm_bf16x6_error_vs_fp32_mfma) at 6/16 of
// the fp32-MFMA matrix time (the fp32 matrix rate is 1/16 of the bf16 rate on gfx950; no xf32/TF32 exists).
// OPT-IN (wsi_gemm_set_precision(1)); the exact v_mfma_f32_32x32x2_f32 path in gemm_f32.hip stays the default.
//
// Same grouped launch / descriptor table / epilogues / split-K planning as gemm_f32.hip.  Differences:
//   * LDS holds bf16 planes [stage][A|B][3][128][16(+8)], k contiguous per row, so an A or B fragment of the 32x32x16
//     MFMA (8 consecutive k of one row per lane) is ONE conflict-free ds_read_b128 (row pitch 48 B);
//   * the split happens while staging (v_cvt_pk_bf16_f32 + exact residuals), K-contiguous operands store 8-byte
//     quads per plane, M/N-contiguous operands (dX's W, dW's dY and X) are transposed on the fly by packing the
//     (k, k+1) pair of each column into one 4-byte store;
//   * per 16-deep stage a wave issues 24 MFMAs (6 products x 2x2 tiles), term-major so consecutive MFMAs hit
//     different accumulators, small terms first, with the split of the NEXT stage interleaved between them.
#include "gemm_common.h"
#include <stdlib.h>

namespace wsi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
                                              f32x16 (&acc)[2][2], int m0, int n0, int split, int wave, int lane) {
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int epi = P.epilogue;
    const bool interior = (m0 + BM <= G.M) && (n0 + BN <= G.N);
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
                if (epi & WSI_EPI_SCALE_GATE) x *= gate_s;
                if (epi & WSI_EPI_ADD_R) x = fmaf(r_scale, G.R[(int64_t)row * G.ldr + col], x);
                float* c = G.C + (int64_t)row * G.ldc + col;
                if (epi & WSI_EPI_ACCUMULATE) x += *c;
                *c = x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Software pipeline: 16-deep stages, two LDS buffers.  While the matrix cores work on stage s from
// buffer s&1, the same wave splits the (already landed) registers of stage s+1 into buffer (s+1)&1 and the global
// loads of stage s+2 are in flight: the split's VALU work hides under the 24 MFMAs of a stage instead of sitting
// between two barriers, and there is ONE barrier per stage.
constexpr int SK = 16;                      // k per stage
constexpr int LDS16 = SK + 8;               // bf16 per LDS row: 48-byte pitch, conflict-free ds_read_b128
constexpr int PLANE16 = BM * LDS16;
constexpr int OPER16 = 3 * PLANE16;         // bf16 elements per operand stage (18,432 B)

template <bool KCONTIG>
struct StageLoader {
    float4 r[2];
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
            const int kr = tid & 7, mg = tid >> 3;
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
            const int kr = tid & 7, mg = tid >> 3;
            const int o = o0 + 4 * mg;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + 2 * kr + h;
                r[h] = (k < k_end && o < o_end) ? load4_guarded_b(base + (int64_t)k * ld + o, o, o_end) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ lds, int tid) const {
        if constexpr (KCONTIG) {
            const int c = tid & 3, rr = tid >> 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t a0, a1, a2, b0, b1, b2;
                split2(r[q].x, r[q].y, a0, a1, a2);
                split2(r[q].z, r[q].w, b0, b1, b2);
                __bf16* d = lds + (rr + 64 * q) * LDS16 + 4 * c;
                *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(d + PLANE16) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(d + 2 * PLANE16) = make_uint2(a2, b2);
            }
        } else {
            const int kr = tid & 7, mg = tid >> 3;
            const float x[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
            const float y[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t p0, p1, p2;
                split2(x[i], y[i], p0, p1, p2);
                __bf16* d = lds + (4 * mg + i) * LDS16 + 2 * kr;
                *reinterpret_cast<uint32_t*>(d) = p0;
                *reinterpret_cast<uint32_t*>(d + PLANE16) = p1;
                *reinterpret_cast<uint32_t*>(d + 2 * PLANE16) = p2;
            }
        }
    }
    __device__ __forceinline__ void add_colsum(float (&cs)[4], float f) const {
        cs[0] = fmaf(f, r[0].x + r[1].x, cs[0]); cs[1] = fmaf(f, r[0].y + r[1].y, cs[1]);
        cs[2] = fmaf(f, r[0].z + r[1].z, cs[2]); cs[3] = fmaf(f, r[0].w + r[1].w, cs[3]);
    }
};

template <bool A_KC, bool B_KC, bool SPLITK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16x6_pipe_kernel(const GemmParams P, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[4 * OPER16];    // 73,728 B: [stage buffer][A | B][plane][row][k]

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

    const bool do_colsum = SPLITK && !A_KC && (G.cs_off >= 0) && (tn == 0);
    const float csf = do_colsum ? 1.f : 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    const int fa = (wm * 64 + l31) * LDS16 + 8 * hi;
    const int fb = OPER16 + (wn * 64 + l31) * LDS16 + 8 * hi;
    bf16x8 fra[3][2], frb[3][2];
    auto read_frags = [&](const __bf16* buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fra[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fa + pl * PLANE16 + i * 32 * LDS16);
                frb[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fb + pl * PLANE16 + i * 32 * LDS16);
            }
    };
    auto mfma_stage = [&]() {
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[TA[t]][i], frb[TB[t]][j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](const __bf16* buf) { read_frags(buf); mfma_stage(); };
    auto bsel = [&](int k0, int& kloc) -> const float* {
        if (G.bchunk <= 0) { kloc = k0; return G.B; }
        const int w = k0 / G.bchunk;
        kloc = k0 - w * G.bchunk;
        return w == 0 ? G.B : (w == 1 ? G.B1 : G.B2);
    };

    const bool fast = avec && bvec && (A_KC ? true : (m0 + BM <= G.M)) && (B_KC ? true : (n0 + BN <= G.N));
    const int nst = fast ? (ke - kb) / SK : 0;
    if (nst > 0) {
        StageLoader<A_KC> a0, a1;
        StageLoader<B_KC> b0, b1;
        auto fetch = [&](StageLoader<A_KC>& la, StageLoader<B_KC>& lb, int s) {
            const int k0 = kb + min(s, nst - 1) * SK;        // past the end: re-load the last stage (never consumed)
            int kl;
            const float* bb = bsel(k0, kl);
            la.load_fast(G.A, G.lda, m0, k0, G.M, tid);
            lb.load_fast(bb, G.ldb, n0, kl, G.N, tid);
        };
        // stage s: registers `c*` hold stage s+1 (landed), `n*` are free
        auto body = [&](StageLoader<A_KC>& ca, StageLoader<B_KC>& cb, StageLoader<A_KC>& na, StageLoader<B_KC>& nb, int s) {
            fetch(na, nb, s + 2);
            __builtin_amdgcn_sched_barrier(0);
            __bf16* cur = smem + (s & 1) * 2 * OPER16;
            __bf16* nxt = smem + ((s + 1) & 1) * 2 * OPER16;
            // source order matters: the fragment READS of `cur` come first so that the LDS WRITES into `nxt` (which the
            // compiler must assume may alias) can be scheduled late, between the MFMAs
            read_frags(cur);
            ca.add_colsum(cs, (s + 1 < nst) ? csf : 0.f);
            ca.store(nxt, tid);
            cb.store(nxt + OPER16, tid);
            mfma_stage();
            // issue order: fragment reads, a little split work while they land, then one MFMA per ~4 VALU ops of the split
            // (the matrix core runs 8 passes per MFMA: the VALU work of the next stage rides in its shadow)
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
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
        StageLoader<A_KC> la;
        StageLoader<B_KC> lb;
        for (int k0 = kb + nst * SK; k0 < ke; k0 += SK) {
            la.load_guarded(G.A, G.lda, m0, k0, G.M, ke, tid);
            if (G.bchunk > 0) {
                const int w = k0 / G.bchunk, kloc = k0 - w * G.bchunk;
                lb.load_guarded(w == 0 ? G.B : (w == 1 ? G.B1 : G.B2), G.ldb, n0, kloc, G.N, min(G.bchunk, ke - w * G.bchunk), tid);
            } else
                lb.load_guarded(G.B, G.ldb, n0, k0, G.N, ke, tid);
            la.add_colsum(cs, csf);
            la.store(smem, tid);
            lb.store(smem + OPER16, tid);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    }

    float* fsm = reinterpret_cast<float*>(smem);
    if (do_colsum) {
        const int kr = tid & 7, mg = tid >> 3;
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
    gemm_epilogue<SPLITK>(P, G, ws, fsm, acc, m0, n0, split, wave, lane);
}

void launch_gemm_bf16x6(int op, const GemmParams& P, int tiles, unsigned lds_pad, float* ws, hipStream_t st) {
    const dim3 g(tiles), b(GEMM_THREADS);
    if (op == WSI_GEMM_TN) hipLaunchKernelGGL((gemm_bf16x6_pipe_kernel<false, false, true>), g, b, lds_pad, st, P, ws);
    else if (op == WSI_GEMM_NT) hipLaunchKernelGGL((gemm_bf16x6_pipe_kernel<true, true, false>), g, b, lds_pad, st, P, ws);
    else hipLaunchKernelGGL((gemm_bf16x6_pipe_kernel<true, false, false>), g, b, lds_pad, st, P, ws);
}

}  // namespace wsi
