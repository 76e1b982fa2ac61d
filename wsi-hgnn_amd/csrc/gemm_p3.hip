// Grouped fp32-class GEMM on PRE-SPLIT operands ("P3" = blocked bf16x3 planes), gfx950 only.
//
// gemm_emu16.hip emulates an fp32 product with 6 bf16 MFMA products of the exact 3-way bf16 split of both operands, but
// splits every operand tile again in every workgroup that touches it (an A tile N/128 times, a weight tile M/128 = 625
// times at the bench shapes): ~95 of its ~120 non-MFMA instructions per 16-deep stage.  Here every tensor is split ONCE,
// by whoever produces it (wsi_split_planes for inputs that arrive as fp32, the epilogues of this GEMM and of the attention
// kernels for everything computed on the way), and the GEMM main loop is a pure bf16 pipeline: 16-byte global loads ->
// ds_write_b128 -> ds_read_b128 fragments -> 24 x v_mfma_f32_32x32x16_bf16 per stage and wave, no VALU work at all.
//
// Plane format of a logical fp32 matrix X[R, C] ("P3"):  bf16 array  [R][ceil(C/16)][3][16]
//   element (r, c), term t (x = x0 + x1 + x2, 8 + 8 + 8 significand bits, exact) at  r*ld + (c/16)*48 + t*16 + (c%16);
//   columns beyond C inside the last block are ZERO, so every reduction over C can run whole 16-deep stages;
//   one (row, 16-column block) = 96 contiguous bytes holding all three terms -> a K-contiguous operand stage is 96 B per
//   row, and one k-row of a TN operand tile (128 columns) is 768 contiguous bytes.
//
//   NT  C[M,N] = A[M,K] B[N,K]^T   both operands K-contiguous planes (forward Y = X W^T; dX = dY (W^T)^T with planes of W^T)
//   TN  C[M,N] = A[K,M]^T B[K,N]   reduction over ROWS of both plane sets (dW = dY^T X); the 4x8 transposition each operand
//                                  needs (a lane loads 8 columns of one k-row, an MFMA lane wants 8 k of one column) is done
//                                  with v_perm_b32 on packed bf16 pairs while staging: 16 VALU ops per 64 staged values
//                                  instead of the ~190 the fp32 split cost.  Split-K slabs + fixed-order reduce as gemm_f32.hip;
//                                  the bias gradient colsum(dY) comes out of 6 extra MFMAs per sub-step against a ones fragment.
// Numerics are those of gemm_emu16.hip (same six products, same order).
#include "gemm_common.h"
#include <stdlib.h>

namespace wsi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: arrays of it copy through memcpy on private memory)

__device__ __forceinline__ uint32_t p3_cvt_pk(float lo, float hi) {
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}
// exact 3-way split of two floats into packed bf16 pairs (low half = first value)
__device__ __forceinline__ void p3_split2(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = p3_cvt_pk(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = p3_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = p3_cvt_pk(sa, sb);
}

struct P3Group {
    const uint16_t* Ap; const uint16_t* Bp; float* C; uint16_t* Cp;
    const float* bias; const float* R; const float* gate; const float* Mm;
    int64_t ldap, ldbp, ldc, ldcp, ldr, ldm;
    int64_t cs_off;        // TN with colsum_out: float offset of the [splits][M] column-sum partials in the workspace, else -1
    int64_t ws_off;        // TN: float offset of this group's slabs in the workspace
    int32_t M, N, K;
    int32_t tile_start, tiles_n, tiles_mn;
    int32_t kchunk;        // TN: rows of the reduction per split (multiple of 32)
    int32_t flags;         // bit2: C (and R, Mm) take 16-byte accesses
};
struct P3Params {
    P3Group g[WSI_GEMM_MAX_GROUPS];
    int32_t ngroups, total_tiles, epilogue, pad;
};

// ------------------------------------------------------------------------------------------------ epilogue
// fsm: >= 32 KB of LDS nobody reads any more.  SPLITK: raw accumulators into this split's slab of the workspace.
template <bool SPLITK>
__device__ __forceinline__ void p3_epilogue(const P3Params& P, const P3Group& G, float* __restrict__ ws, float* fsm,
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
                        const float4 o = *reinterpret_cast<const float4*>(cbase + (int64_t)row * ldc + col);
                        x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
                    }
                    if (G.Cp) {     // the consumer GEMMs read this tensor as planes: split it here, once
                        uint32_t a0, a1, a2, b0, b1, b2;
                        p3_split2(x.x, x.y, a0, a1, a2);
                        p3_split2(x.z, x.w, b0, b1, b2);
                        uint16_t* d = G.Cp + (int64_t)row * G.ldcp + (col >> 4) * 48 + (col & 15);
                        *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
                        *reinterpret_cast<uint2*>(d + 16) = make_uint2(a1, b1);
                        *reinterpret_cast<uint2*>(d + 32) = make_uint2(a2, b2);
                    }
                }
                if (SPLITK || G.C) *reinterpret_cast<float4*>(cbase + (int64_t)row * ldc + col) = x;
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
                if ((epi & WSI_EPI_ACCUMULATE) && G.C) x += G.C[(int64_t)row * G.ldc + col];
                if (G.C) G.C[(int64_t)row * G.ldc + col] = x;
                if (G.Cp) {
                    uint32_t q0, q1, q2;
                    p3_split2(x, 0.f, q0, q1, q2);
                    uint16_t* d = G.Cp + (int64_t)row * G.ldcp + (col >> 4) * 48 + (col & 15);
                    d[0] = (uint16_t)(q0 & 0xffffu); d[16] = (uint16_t)(q1 & 0xffffu); d[32] = (uint16_t)(q2 & 0xffffu);
                }
            }
        }
    }
}

// NT tile order inside a group: the N dimension is cut into chunks of `w` tile columns whose B planes fit the XCD's L2 (host:
// p3_nchunk); tiles run chunk by chunk, inside a chunk row panel by row panel, inside a panel along N.  An XCD walks a contiguous
// range of that order (xcd_remap), so consecutive tiles share their A panel and the chunk's B slice stays L2-resident across
// panels — with the plain row-major order a B of 4.7 MB (kqv planes) is re-streamed from the Infinity Cache for every panel.
__device__ __forceinline__ void p3_tile_coords(int local, int tiles_m, int tiles_n, int w, int& tm, int& tn) {
    const int per = tiles_m * w;
    int c = local / per;
    const int nc = tiles_n / w;
    if (c > nc) c = nc;
    const int rem = local - c * per;
    const int wc = (c < nc) ? w : (tiles_n - nc * w);
    tm = rem / wc;
    tn = c * w + (rem - tm * wc);
}

__device__ __forceinline__ int p3_find_group(const P3Params& P, int tile) {
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngroups; ++i) gi = (tile >= P.g[i].tile_start) ? i : gi;
    return gi;
}

// six cross products of one 16-deep stage, small terms first, term-major so consecutive MFMAs hit different accumulators
__device__ __forceinline__ void p3_mfma_stage(f32x16 (&acc)[2][2], const bf16x8 (&fa)[3][2], const bf16x8 (&fb)[3][2]) {
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]][i], fb[TB[t]][j], acc[i][j], 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ NT
constexpr int NT_ROWB = 96;                 // LDS bytes per tile row: six 16-byte slots (3 planes x 2 k-halves), no padding
constexpr int NT_OPER = BM * NT_ROWB;       // 12,288 B per operand stage -> 49,152 B per workgroup: THREE workgroups per CU
// A 96-byte pitch (24 banks) repeats its bank pattern every 8 rows, and a ds_read_b128 lane group reads 16 rows: rows with
// bit 3 set rotate their six slots by one (slot' = (slot + 1) mod 6), which moves them 4 banks off every unrotated row
// (those sit on multiples of 8 banks) -> conflict-free without the padding that would cost the third workgroup.
__device__ __forceinline__ int nt_slot(int row, int slot) {
    const int s = slot + ((row >> 3) & 1);
    return s >= 6 ? s - 6 : s;
}

template <int RES>
__global__ __launch_bounds__(GEMM_THREADS, RES) void gemm_p3_nt_kernel(const P3Params P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * NT_OPER];    // 49,152 B: [buffer][A | B][row][6 slots]
    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    const P3Group& G = P.g[p3_find_group(P, tile)];
    const int local = tile - G.tile_start;
    int tm, tn;
    p3_tile_coords(local, G.tiles_mn / G.tiles_n, G.tiles_n, G.kchunk, tm, tn);      // NT: kchunk holds the N chunk width in tiles
    const int m0 = tm * BM, n0 = tn * BN;
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

    // staging: an operand stage = 128 rows x 96 B = 768 16-byte chunks, three per thread; chunk c = (row c/6, part c%6).
    // Rows beyond the matrix are CLAMPED to its last row (their products land in C rows/columns that are never stored).
    const unsigned char* ag[3];
    const unsigned char* bg[3];
    int lo[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = tid + 256 * q;
        const int row = c / 6, part = c - row * 6;
        ag[q] = reinterpret_cast<const unsigned char*>(G.Ap) + (int64_t)min(m0 + row, G.M - 1) * G.ldap * 2 + part * 16;
        bg[q] = reinterpret_cast<const unsigned char*>(G.Bp) + (int64_t)min(n0 + row, G.N - 1) * G.ldbp * 2 + part * 16;
        lo[q] = row * NT_ROWB + nt_slot(row, part) * 16;
    }
    const int nst = (G.K + 15) >> 4;
    // fragment of plane pl, k-half hi = slot 2*pl + hi of its row (rows +32 share bit 3: one offset serves both i)
    int fa_off[3], fb_off[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        fa_off[pl] = (wm * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
        fb_off[pl] = NT_OPER + (wn * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
    }

    u32x4 ra[3], rb[3];
    auto fetch = [&](int s) __attribute__((always_inline)) {
        const int off = min(s, nst - 1) * 96;              // past the end: re-load the last stage (never consumed)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ra[q] = *reinterpret_cast<const u32x4*>(ag[q] + off);
            rb[q] = *reinterpret_cast<const u32x4*>(bg[q] + off);
        }
    };
    auto store = [&](unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            *reinterpret_cast<u32x4*>(buf + lo[q]) = ra[q];
            *reinterpret_cast<u32x4*>(buf + NT_OPER + lo[q]) = rb[q];
        }
    };
    bf16x8 fa[3][2], fb[3][2];
    auto read_frags = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fa_off[pl] + i * 32 * NT_ROWB);
                fb[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fb_off[pl] + i * 32 * NT_ROWB);
            }
    };
    // One register set, two LDS buffers, one barrier per stage: the loads of stage s+1 are issued before the fragment reads of
    // stage s, fly under its first 18 MFMAs and are written to the other buffer between the last six.
    fetch(0);
    store(smem);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        unsigned char* cur = smem + (s & 1) * 2 * NT_OPER;
        unsigned char* nxt = smem + ((s + 1) & 1) * 2 * NT_OPER;
        fetch(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur);
        p3_mfma_stage(acc, fa, fb);
        store(nxt);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);     // fragment reads first
        __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);     // MFMAs while the global loads land
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one LDS write of the next stage ...
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // ... per remaining MFMA
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    p3_epilogue<false>(P, G, nullptr, reinterpret_cast<float*>(smem), acc, m0, n0, 0, wave, lane);
}

// Same tile, operands brought in by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass).  An LDS-DMA
// instruction writes wave-uniform base + lane*16, so the LDS image of an operand stage is the LINEAR chunk order c = row*6 +
// physical slot and the slot rotation of nt_slot() is applied on the SOURCE side: lane c fetches the logical slot that
// belongs at its physical position.  Two LDS buffers; the loads of stage s+1 are issued before the fragment reads of stage s
// and the barrier at the end of the stage (which carries the vmcnt(0) wait) publishes them.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int RES>
__global__ __launch_bounds__(GEMM_THREADS, RES) void gemm_p3_nt_glds_kernel(const P3Params P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * NT_OPER];
    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    const P3Group& G = P.g[p3_find_group(P, tile)];
    const int local = tile - G.tile_start;
    int tm, tn;
    p3_tile_coords(local, G.tiles_mn / G.tiles_n, G.tiles_n, G.kchunk, tm, tn);      // NT: kchunk holds the N chunk width in tiles
    const int m0 = tm * BM, n0 = tn * BN;
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

    const unsigned char* ag[3];
    const unsigned char* bg[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = (wave * 3 + q) * 64 + lane;          // linear LDS chunk this lane fills
        const int row = c / 6, ps = c - row * 6;
        int ls = ps - ((row >> 3) & 1);                    // inverse of nt_slot: the logical slot stored at physical slot ps
        ls = ls < 0 ? ls + 6 : ls;
        ag[q] = reinterpret_cast<const unsigned char*>(G.Ap) + (int64_t)min(m0 + row, G.M - 1) * G.ldap * 2 + ls * 16;
        bg[q] = reinterpret_cast<const unsigned char*>(G.Bp) + (int64_t)min(n0 + row, G.N - 1) * G.ldbp * 2 + ls * 16;
    }
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 3 * 1024);     // bytes: three 1-KB pieces per wave and operand
    const int nst = (G.K + 15) >> 4;
    int fa_off[3], fb_off[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        fa_off[pl] = (wm * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
        fb_off[pl] = NT_OPER + (wn * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
    }
    auto dma = [&](unsigned char* buf, int s) __attribute__((always_inline)) {
        const int off = min(s, nst - 1) * 96;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(ag[q] + off), (lds_void_t*)(buf + wbase + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(bg[q] + off), (lds_void_t*)(buf + NT_OPER + wbase + q * 1024), 16, 0, 0);
        }
    };
    bf16x8 fa[3][2], fb[3][2];
    dma(smem, 0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        unsigned char* cur = smem + (s & 1) * 2 * NT_OPER;
        unsigned char* nxt = smem + ((s + 1) & 1) * 2 * NT_OPER;
        dma(nxt, s + 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[pl][i] = *reinterpret_cast<const bf16x8*>(cur + fa_off[pl] + i * 32 * NT_ROWB);
                fb[pl][i] = *reinterpret_cast<const bf16x8*>(cur + fb_off[pl] + i * 32 * NT_ROWB);
            }
        __builtin_amdgcn_s_setprio(1);
        p3_mfma_stage(acc, fa, fb);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    p3_epilogue<false>(P, G, nullptr, reinterpret_cast<float*>(smem), acc, m0, n0, 0, wave, lane);
}

// LDS-DMA with THREE buffers and register-prefetched fragments.  The two-buffer kernels above put each wave through
// [issue loads][12 fragment reads, wait for them][24 MFMAs][wait for the loads, barrier] per stage; co-resident workgroups run the
// same timeline, fall into lock-step, and the matrix pipe idles while all of them sit in the wait phases (measured: 0.6 busy with 2
// or 3 workgroups per CU alike).  Here a wave's own loads overlap its own MFMAs:
//   stage s:  DMA of stage s+3 -> buffer s%3 (free: its fragments went to registers during stage s-1)
//             fragment reads of stage s+1 <- buffer (s+1)%3 into the OTHER register set (complete since the last barrier)
//             24 MFMAs on the fragments of stage s (already in registers: they start right after the barrier)
//             s_waitcnt vmcnt(6) [stage s+2 has landed; the 6 pieces of stage s+3 stay in flight], lgkmcnt(0); s_barrier
// so every load has two stages to land and no MFMA waits for LDS latency.  Raw s_barrier + counted vmcnt: a __syncthreads()
// would drain the LDS-DMA queue (vmcnt(0)) at every stage.
template <int RES>
__global__ __launch_bounds__(GEMM_THREADS, RES) void gemm_p3_nt_pf_kernel(const P3Params P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * NT_OPER];    // 73,728 B: [3 buffers][A | B][row][6 slots]
    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    const P3Group& G = P.g[p3_find_group(P, tile)];
    const int local = tile - G.tile_start;
    int tm, tn;
    p3_tile_coords(local, G.tiles_mn / G.tiles_n, G.tiles_n, G.kchunk, tm, tn);      // NT: kchunk holds the N chunk width in tiles
    const int m0 = tm * BM, n0 = tn * BN;
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

    const unsigned char* ag[3];
    const unsigned char* bg[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c = (wave * 3 + q) * 64 + lane;
        const int row = c / 6, ps = c - row * 6;
        int ls = ps - ((row >> 3) & 1);
        ls = ls < 0 ? ls + 6 : ls;
        ag[q] = reinterpret_cast<const unsigned char*>(G.Ap) + (int64_t)min(m0 + row, G.M - 1) * G.ldap * 2 + ls * 16;
        bg[q] = reinterpret_cast<const unsigned char*>(G.Bp) + (int64_t)min(n0 + row, G.N - 1) * G.ldbp * 2 + ls * 16;
    }
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 3 * 1024);
    const int nst = (G.K + 15) >> 4;
    int fa_off[3], fb_off[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        fa_off[pl] = (wm * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
        fb_off[pl] = NT_OPER + (wn * 64 + l31) * NT_ROWB + nt_slot(l31, 2 * pl + hi) * 16;
    }
    auto dma = [&](int bufi, int s) __attribute__((always_inline)) {
        unsigned char* buf = smem + bufi * 2 * NT_OPER;
        const int off = min(s, nst - 1) * 96;              // past the end: re-load the last stage (never consumed)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(ag[q] + off), (lds_void_t*)(buf + wbase + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(bg[q] + off), (lds_void_t*)(buf + NT_OPER + wbase + q * 1024), 16, 0, 0);
        }
    };
    auto frags = [&](bf16x8 (&fa)[3][2], bf16x8 (&fb)[3][2], int bufi) __attribute__((always_inline)) {
        const unsigned char* buf = smem + bufi * 2 * NT_OPER;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fa_off[pl] + i * 32 * NT_ROWB);
                fb[pl][i] = *reinterpret_cast<const bf16x8*>(buf + fb_off[pl] + i * 32 * NT_ROWB);
            }
    };
    bf16x8 fa0[3][2], fb0[3][2], fa1[3][2], fb1[3][2];
    // one stage: cf = fragments of stage s (in registers), nf = where stage s+1's fragments go
    auto body = [&](bf16x8 (&cfa)[3][2], bf16x8 (&cfb)[3][2], bf16x8 (&nfa)[3][2], bf16x8 (&nfb)[3][2], int s, int b0, int b1) __attribute__((always_inline)) {
        dma(b0, s + 3);                                     // buffer s%3
        const unsigned char* nb = smem + b1 * 2 * NT_OPER;  // buffer (s+1)%3
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
        // source order = issue order: two MFMAs of this stage, then one fragment read of the next; the first MFMAs come BEFORE the
        // first read so that the wait hipcc places for this stage's fragments (it cannot see the asm wait below) is the trivial one
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfa[TA[t]][i], cfb[TB[t]][j], acc[i][j], 0, 0, 0);
                const int g = t * 2 + i;                    // 0..11: which fragment to fetch now: A(pl, ii) for g < 6, B(pl, ii) after
                const int pl = (g % 6) >> 1, ii = g & 1;
                if (g < 6) nfa[pl][ii] = *reinterpret_cast<const bf16x8*>(nb + fa_off[pl] + ii * 32 * NT_ROWB);
                else nfb[pl][ii] = *reinterpret_cast<const bf16x8*>(nb + fb_off[pl] + ii * 32 * NT_ROWB);
            }
        __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);      // the 6 LDS-DMA issues
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // two MFMAs of this stage ...
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... one fragment read of the next
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    dma(0, 0);
    dma(1, 1);
    dma(2, 2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // stages 0 and 1 have landed (stage 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    frags(fa0, fb0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // every wave holds stage 0's fragments: buffer 0 may be refilled
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (; s + 5 < nst; s += 6) {                           // 6 = lcm(2 register sets, 3 buffers): static indices throughout
        body(fa0, fb0, fa1, fb1, s + 0, 0, 1);
        body(fa1, fb1, fa0, fb0, s + 1, 1, 2);
        body(fa0, fb0, fa1, fb1, s + 2, 2, 0);
        body(fa1, fb1, fa0, fb0, s + 3, 0, 1);
        body(fa0, fb0, fa1, fb1, s + 4, 1, 2);
        body(fa1, fb1, fa0, fb0, s + 5, 2, 0);
    }
    // tail (< 6 stages left): same bodies, stopping when the reduction ends (the loop above leaves s % 6 == 0)
    if (s < nst) { body(fa0, fb0, fa1, fb1, s, 0, 1); ++s; }
    if (s < nst) { body(fa1, fb1, fa0, fb0, s, 1, 2); ++s; }
    if (s < nst) { body(fa0, fb0, fa1, fb1, s, 2, 0); ++s; }
    if (s < nst) { body(fa1, fb1, fa0, fb0, s, 0, 1); ++s; }
    if (s < nst) { body(fa0, fb0, fa1, fb1, s, 1, 2); ++s; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain the over-issued loads before LDS is reused by the epilogue
    __syncthreads();
    p3_epilogue<false>(P, G, nullptr, reinterpret_cast<float*>(smem), acc, m0, n0, 0, wave, lane);
}

// ------------------------------------------------------------------------------------------------ TN
constexpr int TN_K = 32;                    // reduction rows per stage
constexpr int TN_ROWB = 256;                // LDS bytes per tile column m: 16 slots of 16 B; logical slot = plane*4 + k/8 (12 used)
constexpr int TN_OPER = BM * TN_ROWB;       // 32,768 B per operand
// Physical slot = logical slot ^ (m & 15): with a 256-byte pitch every column starts on bank 0, and the XOR sends the 16
// columns a ds_read_b128 lane group reads (same plane, same k) to 16 different slots = all 64 banks, conflict-free; because
// the pitch has no low address bits, "^ (plane << 6)" and "^ (sub-step << 5)" on a lane's base address select plane and k
// half without further address arithmetic.

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_p3_tn_kernel(const P3Params P, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TN_OPER];    // 65,536 B: [A | B][column][16 swizzled slots]
    const int tid = threadIdx.x;
    const int tile = xcd_remap((int)blockIdx.x, P.total_tiles);
    const P3Group& G = P.g[p3_find_group(P, tile)];
    int local = tile - G.tile_start;
    const int split = local / G.tiles_mn;
    local -= split * G.tiles_mn;
    const int tm = local / G.tiles_n, tn = local - tm * G.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kb = split * G.kchunk;
    const int ke = min(G.K, kb + G.kchunk);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    f32x16 acc[2][2];
    f32x16 accs[2];                                           // column sums of A (bias gradient) via MFMAs against ones
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const bool do_colsum = (G.cs_off >= 0) && (tn == 0) && (wn == 0);     // wave-uniform

    // staging items: (operand, k-quad kq = 4 rows, 16-byte chunk c of the 768-byte tile row): 2 x 8 x 48 = 768 items, three
    // per thread, numbered so that the 48 chunks of a k-row are read by 48 consecutive lanes (768 contiguous bytes).
    // Each item: four 16-byte loads (rows 4kq..4kq+3, 8 columns of one plane) -> 4x8 transposition with v_perm_b32 ->
    // eight ds_write_b64 (4 k of one column each).
    const unsigned char* gp[3];
    int64_t gld[3];
    int lbase[3];          // LDS byte address of the item's first column + the 8-byte half of its k-quad inside a slot
    int lslot[3];          // logical slot (plane*4 + kq/2) ^ (first column & 15); column +jj: ^ jj
    int krow[3];           // first reduction row of the item relative to the stage
    bool cvalid[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int it = tid + 256 * q;
        const int oper = it >= 384 ? 1 : 0;                  // wave-uniform (384 = 6 waves)
        const int rem = it - oper * 384;
        // k-quad in the low 3 lane bits: the 16 lanes of a ds_write_b64 group are 8 k-quads (8 different 8-byte positions =
        // 16 banks) x 2 planes (+16 banks), instead of 16 columns that all alias (8-way conflicts measured: SQ_LDS_BANK_CONFLICT
        // = 0.6 of the LDS cycles); the 8 chunk indices a wave covers per k-row are still 128 contiguous bytes
        const int kq = rem & 7, c = rem >> 3;
        const int cb = c / 6, pp = c - cb * 6;
        const int half = pp / 3, p = pp - half * 3;
        const int part = 2 * p + half;
        const int col0 = oper ? n0 : m0, dim = oper ? G.N : G.M;
        const int gcb = (col0 >> 4) + cb;
        cvalid[q] = gcb * 16 < dim;
        gld[q] = (oper ? G.ldbp : G.ldap) * 2;
        gp[q] = reinterpret_cast<const unsigned char*>(oper ? G.Bp : G.Ap) + (int64_t)gcb * 96 + part * 16;
        krow[q] = 4 * kq;
        const int mloc = cb * 16 + half * 8;
        lbase[q] = oper * TN_OPER + mloc * TN_ROWB + (kq & 1) * 8;
        lslot[q] = (p * 4 + (kq >> 1)) ^ (half * 8);
    }
    u32x4 rg[3][4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + krow[q] + j;
                rg[q][j] = (cvalid[q] && k < ke) ? *reinterpret_cast<const u32x4*>(gp[q] + (int64_t)k * gld[q]) : u32x4{0u, 0u, 0u, 0u};
            }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint32_t r0[4] = {rg[q][0].x, rg[q][0].y, rg[q][0].z, rg[q][0].w};
            const uint32_t r1[4] = {rg[q][1].x, rg[q][1].y, rg[q][1].z, rg[q][1].w};
            const uint32_t r2[4] = {rg[q][2].x, rg[q][2].y, rg[q][2].z, rg[q][2].w};
            const uint32_t r3[4] = {rg[q][3].x, rg[q][3].y, rg[q][3].z, rg[q][3].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // dword i holds columns 2i (low half) and 2i+1 (high half)
                const uint32_t e_lo = __builtin_amdgcn_perm(r1[i], r0[i], 0x05040100u), e_hi = __builtin_amdgcn_perm(r3[i], r2[i], 0x05040100u);
                const uint32_t o_lo = __builtin_amdgcn_perm(r1[i], r0[i], 0x07060302u), o_hi = __builtin_amdgcn_perm(r3[i], r2[i], 0x07060302u);
                *reinterpret_cast<uint2*>(smem + lbase[q] + (2 * i) * TN_ROWB + ((lslot[q] ^ (2 * i)) << 4)) = make_uint2(e_lo, e_hi);
                *reinterpret_cast<uint2*>(smem + lbase[q] + (2 * i + 1) * TN_ROWB + ((lslot[q] ^ (2 * i + 1)) << 4)) = make_uint2(o_lo, o_hi);
            }
        }
    };
    // fragment addresses: column m = w*64 + i*32 + l31 (m & 15 = l31 & 15), logical slot = plane*4 + 2*s2 + hi
    const int fa_base = (wm * 64 + l31) * TN_ROWB + ((hi ^ (l31 & 15)) << 4);
    const int fb_base = TN_OPER + (wn * 64 + l31) * TN_ROWB + ((hi ^ (l31 & 15)) << 4);
    const uint32_t one2 = 0x3f803f80u;                         // two bf16 ones
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(one2, one2, one2, one2));

    const int nst = (ke - kb + TN_K - 1) / TN_K;
    if (nst > 0) fetch(kb);
    for (int s = 0; s < nst; ++s) {
        stage();
        __syncthreads();
        if (s + 1 < nst) fetch(kb + (s + 1) * TN_K);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[pl][i] = *reinterpret_cast<const bf16x8*>(smem + ((fa_base ^ (pl << 6) ^ (s2 << 5)) + i * 32 * TN_ROWB));
                    fb[pl][i] = *reinterpret_cast<const bf16x8*>(smem + ((fb_base ^ (pl << 6) ^ (s2 << 5)) + i * 32 * TN_ROWB));
                }
            p3_mfma_stage(acc, fa, fb);
            if (do_colsum) {
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pl][i], ones, accs[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    if (do_colsum && l31 == 0) {      // every column of accs holds the same sums; lane 0 / 32 write their 16 rows
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < G.M) ws[G.cs_off + (int64_t)split * G.M + m] = accs[i][r];
            }
    }
    p3_epilogue<true>(P, G, ws, reinterpret_cast<float*>(smem), acc, m0, n0, split, wave, lane);
}

// ------------------------------------------------------------------------------------------------ fp32 -> planes
// One thread per (row, 8-column half block): two 16-byte loads, three 16-byte stores.  HBM-bound (4 B in, 6 B out per value).
__global__ __launch_bounds__(256) void p3_split_rows_kernel(const float* __restrict__ x, int64_t ldx, int32_t rows, int32_t cols,
                                                            uint16_t* __restrict__ pl, int64_t ldp, int32_t nhalf, int vec) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = idx / nhalf;
    if (r >= rows) return;
    const int h = (int)(idx - r * nhalf);
    const int c0 = h * 8;
    float v[8];
    const float* src = x + r * ldx + c0;
    if (vec && c0 + 8 <= cols) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c0 + i < cols) ? src[i] : 0.f;
    }
    uint32_t q0[4], q1[4], q2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p3_split2(v[2 * i], v[2 * i + 1], q0[i], q1[i], q2[i]);
    uint16_t* d = pl + r * ldp + (h >> 1) * 48 + (h & 1) * 8;
    *reinterpret_cast<uint4*>(d) = make_uint4(q0[0], q0[1], q0[2], q0[3]);
    *reinterpret_cast<uint4*>(d + 16) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    *reinterpret_cast<uint4*>(d + 32) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}

// planes of X^T (weights: a few hundred KB): thread (c, h) gathers x[8h .. 8h+7][c] (adjacent threads = adjacent c: coalesced
// reads) and writes one 8-column half block of output row c.
__global__ __launch_bounds__(256) void p3_split_transpose_kernel(const float* __restrict__ x, int64_t ldx, int32_t rows, int32_t cols,
                                                                 uint16_t* __restrict__ pl, int64_t ldp, int32_t nhalf) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t h = idx / cols;
    if (h >= nhalf) return;
    const int c = (int)(idx - h * cols);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (h * 8 + i < rows) ? x[(h * 8 + i) * ldx + c] : 0.f;
    uint32_t q0[4], q1[4], q2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p3_split2(v[2 * i], v[2 * i + 1], q0[i], q1[i], q2[i]);
    uint16_t* d = pl + (int64_t)c * ldp + (h >> 1) * 48 + (h & 1) * 8;
    *reinterpret_cast<uint4*>(d) = make_uint4(q0[0], q0[1], q0[2], q0[3]);
    *reinterpret_cast<uint4*>(d + 16) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    *reinterpret_cast<uint4*>(d + 32) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}

static int32_t p3_plan_kchunk(const wsi_gemm_p3_group_t* g, int32_t ng) {
    const int64_t target_blocks = 2 * 256;       // one residency round at 2 workgroups per CU
    int64_t work = 0, maxk = 0;
    for (int i = 0; i < ng; ++i) {
        if (g[i].M <= 0 || g[i].N <= 0) continue;
        const int64_t tmn = (int64_t)((g[i].M + BM - 1) / BM) * ((g[i].N + BN - 1) / BN);
        work += tmn * g[i].K;
        if (g[i].K > maxk) maxk = g[i].K;
    }
    int64_t kc = (work + target_blocks - 1) / target_blocks;
    kc = ((kc + TN_K - 1) / TN_K) * TN_K;
    if (kc < 8 * TN_K) kc = 8 * TN_K;
    for (;;) {
        int64_t blocks = 0;
        for (int i = 0; i < ng; ++i) {
            if (g[i].M <= 0 || g[i].N <= 0) continue;
            const int64_t tmn = (int64_t)((g[i].M + BM - 1) / BM) * ((g[i].N + BN - 1) / BN);
            blocks += tmn * (g[i].K > 0 ? (g[i].K + kc - 1) / kc : 1);
        }
        if (blocks <= target_blocks || kc >= maxk) break;
        kc += TN_K;
    }
    return (int32_t)kc;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace wsi

using namespace wsi;

extern "C" int64_t wsi_planes_ld(int32_t cols) { return (int64_t)((cols + 15) / 16) * 48; }

extern "C" int wsi_split_planes(const float* x, int64_t ldx, int32_t rows, int32_t cols, uint16_t* planes, int64_t ldp,
                                int32_t transpose, void* stream) {
    if (rows < 0 || cols < 0 || ldp % 8 != 0) { set_error("split_planes: bad shape rows=%d cols=%d ldp=%lld", rows, cols, (long long)ldp); return WSI_EINVAL; }
    if (rows == 0 || cols == 0) return WSI_OK;
    if (!x || !planes || !al16(planes)) { set_error("split_planes: null or unaligned pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (!transpose) {
        if (ldp < wsi_planes_ld(cols)) { set_error("split_planes: ldp %lld < %lld", (long long)ldp, (long long)wsi_planes_ld(cols)); return WSI_EINVAL; }
        const int nhalf = ((cols + 15) / 16) * 2;
        const int64_t n = (int64_t)rows * nhalf;
        const int vec = al16(x) && (ldx % 4 == 0);
        hipLaunchKernelGGL(p3_split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, rows, cols, planes, ldp, nhalf, vec);
    } else {
        if (ldp < wsi_planes_ld(rows)) { set_error("split_planes(T): ldp %lld < %lld", (long long)ldp, (long long)wsi_planes_ld(rows)); return WSI_EINVAL; }
        const int nhalf = ((rows + 15) / 16) * 2;
        const int64_t n = (int64_t)cols * nhalf;
        hipLaunchKernelGGL(p3_split_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, rows, cols, planes, ldp, nhalf);
    }
    return check_launch("split_planes");
}

extern "C" int64_t wsi_gemm_p3_workspace_bytes(int32_t op, const wsi_gemm_p3_group_t* groups, int32_t ngroups) {
    if (op != WSI_GEMM_TN || !groups || ngroups <= 0) return 0;
    const int32_t kc = p3_plan_kchunk(groups, ngroups);
    int64_t floats = 0;
    for (int i = 0; i < ngroups; ++i) {
        if (groups[i].M <= 0 || groups[i].N <= 0) continue;
        const int64_t splits = groups[i].K > 0 ? (groups[i].K + kc - 1) / kc : 1;
        floats += splits * (int64_t)groups[i].M * groups[i].N;
        if (groups[i].colsum_out) floats += splits * (int64_t)((groups[i].M + 3) / 4 * 4);
    }
    return floats * 4;
}

extern "C" int wsi_gemm_p3(int32_t op, int32_t epilogue, const wsi_gemm_p3_group_t* groups, int32_t ngroups,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (ngroups < 0 || (ngroups > 0 && !groups)) { set_error("gemm_p3: bad group table"); return WSI_EINVAL; }
    if (ngroups > WSI_GEMM_MAX_GROUPS) { set_error("gemm_p3: %d groups > WSI_GEMM_MAX_GROUPS", ngroups); return WSI_EINVAL; }
    if (op != WSI_GEMM_NT && op != WSI_GEMM_TN) { set_error("gemm_p3: op must be NT or TN (NN = NT on the planes of B^T)"); return WSI_EINVAL; }
    if (epilogue & ~(WSI_EPI_BIAS | WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE | WSI_EPI_GELU | WSI_EPI_ADD_R | WSI_EPI_R_1MG | WSI_EPI_MUL_M)) {
        set_error("gemm_p3: unknown epilogue bits 0x%x", epilogue); return WSI_EINVAL; }
    if (op == WSI_GEMM_TN && (epilogue & ~(WSI_EPI_ACCUMULATE | WSI_EPI_SCALE_GATE))) { set_error("gemm_p3: TN accepts only ACCUMULATE and SCALE_GATE"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    P3Params P;
    ReduceParams RP;
    P.ngroups = 0; P.epilogue = epilogue; P.pad = 0;
    RP.ngroups = 0; RP.epilogue = epilogue;
    const int32_t kc = (op == WSI_GEMM_TN) ? p3_plan_kchunk(groups, ngroups) : 0;
    int32_t tiles = 0;
    int64_t ws_floats = 0, red_total = 0;
    for (int i = 0; i < ngroups; ++i) {
        const wsi_gemm_p3_group_t& s = groups[i];
        if (s.M < 0 || s.N < 0 || s.K < 0) { set_error("gemm_p3: negative dimension in group %d", i); return WSI_EINVAL; }
        if (s.M == 0 || s.N == 0) continue;
        if (s.K > 0 && (!s.Ap || !s.Bp)) { set_error("gemm_p3: null operand planes in group %d", i); return WSI_EINVAL; }
        if (!al16(s.Ap) || !al16(s.Bp) || s.ldap % 8 != 0 || s.ldbp % 8 != 0) { set_error("gemm_p3: planes must be 16-byte aligned with ld %% 8 == 0 (group %d)", i); return WSI_EINVAL; }
        if (op == WSI_GEMM_NT) {
            const int64_t need = wsi_planes_ld(s.K);
            if (s.ldap < need || s.ldbp < need) { set_error("gemm_p3 NT: plane ld smaller than ceil(K/16)*48 in group %d", i); return WSI_EINVAL; }
            if (!s.C && !s.Cp) { set_error("gemm_p3: group %d has neither C nor Cp", i); return WSI_EINVAL; }
            if ((epilogue & WSI_EPI_ACCUMULATE) && !s.C) { set_error("gemm_p3: ACCUMULATE needs C (group %d)", i); return WSI_EINVAL; }
            if (s.Cp && (!al16(s.Cp) || s.ldcp % 8 != 0 || s.ldcp < wsi_planes_ld(s.N))) { set_error("gemm_p3: bad output planes in group %d", i); return WSI_EINVAL; }
        } else {
            if (s.ldap < wsi_planes_ld(s.M) || s.ldbp < wsi_planes_ld(s.N)) { set_error("gemm_p3 TN: plane ld too small in group %d", i); return WSI_EINVAL; }
            if (!s.C || s.Cp) { set_error("gemm_p3 TN: needs C, no Cp (group %d)", i); return WSI_EINVAL; }
        }
        if ((epilogue & WSI_EPI_ADD_R) && !s.R) { set_error("gemm_p3: ADD_R needs R (group %d)", i); return WSI_EINVAL; }
        if ((epilogue & WSI_EPI_MUL_M) && !s.Mm) { set_error("gemm_p3: MUL_M needs Mm (group %d)", i); return WSI_EINVAL; }
        P3Group& d = P.g[P.ngroups];
        d.Ap = s.Ap; d.Bp = s.Bp; d.C = s.C; d.Cp = (op == WSI_GEMM_NT) ? s.Cp : nullptr;
        d.bias = s.bias; d.R = s.R; d.gate = s.gate; d.Mm = s.Mm;
        d.ldap = s.ldap; d.ldbp = s.ldbp; d.ldc = s.ldc; d.ldcp = s.ldcp; d.ldr = s.ldr; d.ldm = s.ldm;
        d.M = s.M; d.N = s.N; d.K = s.K;
        const int tmm = (s.M + BM - 1) / BM, tnn = (s.N + BN - 1) / BN;
        d.tiles_n = tnn; d.tiles_mn = tmm * tnn;
        d.tile_start = tiles;
        bool cv;
        if (op == WSI_GEMM_TN) cv = (s.N % 4 == 0) && al16(workspace) && (ws_floats % 4 == 0);
        else cv = (!s.C || (al16(s.C) && s.ldc % 4 == 0)) && (!(epilogue & WSI_EPI_ADD_R) || (al16(s.R) && s.ldr % 4 == 0)) &&
                  (!(epilogue & WSI_EPI_MUL_M) || (al16(s.Mm) && s.ldm % 4 == 0));
        d.flags = cv ? 4 : 0;
        d.ws_off = 0; d.cs_off = -1;
        {   // NT: N chunk width (tiles).  Default = the whole N (plain row-panel-major order); WSI_P3_BCHUNK_KB=<KB> cuts N so that a
            // chunk's B planes stay within that many KB of L2 — measured neutral to slightly negative (1/2/3 MB) on the bench shapes:
            // re-streaming B is not what bounds the kernel (profiles/r02_gemm_pmc.md)
            static const int budget_kb = [] { const char* v = getenv("WSI_P3_BCHUNK_KB"); return v ? atoi(v) : 0; }();
            const int64_t col_bytes = (int64_t)BN * wsi_planes_ld(s.K) * 2;
            int w = budget_kb > 0 ? (int)(((int64_t)budget_kb * 1024) / col_bytes) : tnn;
            d.kchunk = w < 1 ? 1 : (w > tnn ? tnn : w);
        }
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
    if (op == WSI_GEMM_TN) {
        if (!workspace || workspace_bytes < ws_floats * 4) {
            set_error("gemm_p3 TN: workspace of %lld bytes needed, %lld given", (long long)(ws_floats * 4), (long long)workspace_bytes);
            return WSI_ENOMEM;
        }
        hipLaunchKernelGGL(gemm_p3_tn_kernel, dim3(tiles), dim3(GEMM_THREADS), 0, st, P, (float*)workspace);
        RP.total = red_total;
        launch_splitk_reduce(RP, st);
    } else {
        static const int dma = [] { const char* v = getenv("WSI_P3_DMA"); return v ? atoi(v) : 2; }();   // A/B knob, read once: 0 register staging, 1 LDS-DMA x2 buffers, 2 LDS-DMA x3 + prefetch
        if (dma == 2) hipLaunchKernelGGL(gemm_p3_nt_pf_kernel<2>, dim3(tiles), dim3(GEMM_THREADS), 0, st, P);
        else if (dma) hipLaunchKernelGGL(gemm_p3_nt_glds_kernel<3>, dim3(tiles), dim3(GEMM_THREADS), 0, st, P);
        else hipLaunchKernelGGL(gemm_p3_nt_kernel<3>, dim3(tiles), dim3(GEMM_THREADS), 0, st, P);
    }
    return check_launch("gemm_p3");
}
