// Segmented row reductions for gfx950: per-(graph, node type) readout (mean/sum/max) and per-type
// column sums (bias gradients).  Contract + reference call sites (dgl.readout.*_nodes behind
// pooling/avg_pooling.py:15-17, sum_pooling.py:14-16, max_pooling.py:15-17): include/wsi_hgnn.h.
//
// HBM-bound streaming: every row is read exactly once with 16-byte lane accesses; the reduction is
// two-stage over caller-supplied chunk tables (chunks never straddle segments), so the summation
// order is fixed by the tables and the result is bit-reproducible - no atomics.
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int SEG_THREADS = 256;

// stage 1: block = (chunk, 256-column tile); 4 waves take rows round-robin, lanes take 4 columns each.
template <int OP>
__global__ __launch_bounds__(SEG_THREADS) void seg_stage1(const float* __restrict__ x, int64_t ldx, int32_t D, bool vec,
                                                           const int32_t* __restrict__ chunk_row,
                                                           float* __restrict__ partial, int32_t* __restrict__ partial_arg) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 256 + lane * 4;
    const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
    float a[4];
    int arg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (OP == WSI_RED_MAX) ? -INFINITY : 0.f; arg[i] = -1; }
    if (col < D) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const float* p = x + (int64_t)r * ldx + col;
            float v[4];
            if (vec && col + 3 < D) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (col + i < D) ? p[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (OP == WSI_RED_MAX) { if (v[i] > a[i]) { a[i] = v[i]; arg[i] = r; } }
                else a[i] += v[i];
            }
        }
    }
    __shared__ float sh[4][256];
    __shared__ int shi[4][256];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh[wave][lane * 4 + i] = a[i]; if (OP == WSI_RED_MAX) shi[wave][lane * 4 + i] = arg[i]; }
    __syncthreads();
    const int t = threadIdx.x;
    const int oc = blockIdx.y * 256 + t;
    if (oc < D) {
        float s = sh[0][t];
        int sa = (OP == WSI_RED_MAX) ? shi[0][t] : 0;
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            if (OP == WSI_RED_MAX) {
                const float o = sh[w][t]; const int oa = shi[w][t];
                // first (lowest row) maximum wins, as a sequential scan would give
                if (o > s || (o == s && oa >= 0 && (sa < 0 || oa < sa))) { s = o; sa = oa; }
            } else s += sh[w][t];
        }
        partial[(int64_t)c * D + oc] = s;
        if (OP == WSI_RED_MAX) partial_arg[(int64_t)c * D + oc] = sa;
    }
}

// stage 2: thread = (segment, column); sums the segment's chunk partials in chunk order.
template <int OP>
__global__ __launch_bounds__(SEG_THREADS) void seg_stage2(const float* __restrict__ partial, const int32_t* __restrict__ partial_arg,
                                                           int32_t D, const int32_t* __restrict__ chunk_row,
                                                           const int32_t* __restrict__ seg_chunk,
                                                           float* __restrict__ out, int64_t ldo, int32_t* __restrict__ argmax) {
    const int s = blockIdx.x;
    const int col = blockIdx.y * SEG_THREADS + threadIdx.x;
    if (col >= D) return;
    const int c0 = seg_chunk[s], c1 = seg_chunk[s + 1];
    float acc = (OP == WSI_RED_MAX) ? -INFINITY : 0.f;
    int arg = -1;
    for (int c = c0; c < c1; ++c) {
        const float v = partial[(int64_t)c * D + col];
        if (OP == WSI_RED_MAX) {
            if (v > acc) { acc = v; arg = partial_arg[(int64_t)c * D + col]; }
        } else acc += v;
    }
    if (OP == WSI_RED_MEAN) {
        const int cnt = (c1 > c0) ? chunk_row[c1] - chunk_row[c0] : 0;
        acc = acc / (float)(cnt > 0 ? cnt : 1);
    }
    if (OP == WSI_RED_MAX) {
        if (arg < 0) acc = 0.f;   // empty segment -> 0 (DGL replaces -inf by 0)
        argmax[(int64_t)s * D + col] = arg;
    }
    out[(int64_t)s * ldo + col] = acc;
}

// weighted sums, stage 1: block = (chunk, 256-column tile, group of 16 weights); partial[c, j, :] = sum over the chunk's rows of w[r, j] * x[r, :].
// 4 waves take rows round-robin (the row, and with it the 16 weights, is wave-uniform: scalar loads), lanes take 4 columns each.
constexpr int WSUM_J = 16;
__global__ __launch_bounds__(SEG_THREADS) void seg_wsum_stage1(const float* __restrict__ x, int64_t ldx, int32_t D, bool vec,
                                                                const float* __restrict__ w, int64_t ldw, int32_t JW, int32_t J,
                                                                const int32_t* __restrict__ chunk_row, float* __restrict__ partial,
                                                                float* __restrict__ wpartial) {     // optional [chunks, J]: sums of the weights alone
    // JW = columns of w; J = JW or JW + 1: one more column of implicit weight 1 (the plain sums of the rows, for the segment means)
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = blockIdx.y * 256 + lane * 4;
    const int jb = blockIdx.z * WSUM_J;
    const int nj = min(WSUM_J, J - jb);
    const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
    float a[WSUM_J][4];
    float ws[WSUM_J];                         // (wave-uniform: the weights of a row are scalar loads)
#pragma unroll
    for (int j = 0; j < WSUM_J; ++j) {
        ws[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[j][i] = 0.f;
    }
    if (col < D) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const float* p = x + (int64_t)r * ldx + col;
            float v[4];
            if (vec && col + 3 < D) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (col + i < D) ? p[i] : 0.f;
            }
            const float* wr = w + (int64_t)r * ldw + jb;
#pragma unroll
            for (int j = 0; j < WSUM_J; ++j) {
                const float wj = (jb + j < JW) ? wr[j] : ((j < nj) ? 1.f : 0.f);
                ws[j] += wj;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[j][i] = fmaf(wj, v[i], a[j][i]);
            }
        }
    }
    if (wpartial && blockIdx.y == 0) {         // (block-uniform; column tile 0 always has col < D)
        __shared__ float shw[4][WSUM_J];
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < WSUM_J; ++j) shw[wave][j] = ws[j];
        }
        __syncthreads();
        if ((int)threadIdx.x < nj) {
            const int t = threadIdx.x;
            wpartial[(int64_t)c * J + jb + t] = (shw[0][t] + shw[1][t]) + (shw[2][t] + shw[3][t]);
        }
    }
    __shared__ float sh[4][256];
    for (int j = 0; j < nj; ++j) {           // (nj is uniform)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vj = 0.f;
#pragma unroll
            for (int jj = 0; jj < WSUM_J; ++jj) vj = (jj == j) ? a[jj][i] : vj;      // static register indices only
            sh[wave][lane * 4 + i] = vj;
        }
        __syncthreads();
        const int t = threadIdx.x;
        const int oc = blockIdx.y * 256 + t;
        if (oc < D) partial[((int64_t)c * J + jb + j) * D + oc] = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
    }
}

// backward sum/mean: block = (chunk, 256-column tile); gx[r, :] = gout[seg, :] * scale
__global__ __launch_bounds__(SEG_THREADS) void seg_bwd_bcast(const float* __restrict__ gout, int64_t ldgo, int32_t D, int32_t op,
                                                              const int32_t* __restrict__ chunk_row, const int32_t* __restrict__ chunk_seg,
                                                              const int32_t* __restrict__ seg_chunk, bool vec,
                                                              float* __restrict__ gx, int64_t ldgx) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 256 + lane * 4;
    if (col >= D) return;
    const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
    const int s = chunk_seg[c];
    float scale = 1.f;
    if (op == WSI_RED_MEAN) {
        const int cnt = chunk_row[seg_chunk[s + 1]] - chunk_row[seg_chunk[s]];
        scale = 1.f / (float)(cnt > 0 ? cnt : 1);
    }
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = (col + i < D) ? gout[(int64_t)s * ldgo + col + i] * scale : 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        float* p = gx + (int64_t)r * ldgx + col;
        if (vec && col + 3 < D) *reinterpret_cast<float4*>(p) = make_float4(g[0], g[1], g[2], g[3]);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (col + i < D) p[i] = g[i];
        }
    }
}

// backward max: thread = (segment, column): gx[argmax, col] = gout[seg, col]   (gx pre-zeroed)
__global__ __launch_bounds__(SEG_THREADS) void seg_bwd_max(const float* __restrict__ gout, int64_t ldgo, int32_t D,
                                                            const int32_t* __restrict__ argmax, float* __restrict__ gx, int64_t ldgx) {
    const int s = blockIdx.x;
    const int col = blockIdx.y * SEG_THREADS + threadIdx.x;
    if (col >= D) return;
    const int r = argmax[(int64_t)s * D + col];
    if (r >= 0) gx[(int64_t)r * ldgx + col] = gout[(int64_t)s * ldgo + col];
}


// ------------------------------------------------------------------------------------------------ segment dot
// out[s] = sum over rows r of segment s, columns c:  g[r,c] * (a[r,c] - b[r,c])
// (gradient of the HEAT skip gate: d/d alpha of alpha*y + (1-alpha)*h, models/HEATNet4.py:128,135).
__global__ __launch_bounds__(SEG_THREADS) void seg_dot_stage1(const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ a, int64_t lda,
                                                               const float* __restrict__ b, int64_t ldb,
                                                               int32_t D, bool vec, const int32_t* __restrict__ chunk_row,
                                                               float* __restrict__ partial) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 256 + lane * 4;
    const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
    float acc = 0.f;
    if (col < D) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const float* pg = g + (int64_t)r * ldg + col;
            const float* pa = a + (int64_t)r * lda + col;
            const float* pb = b + (int64_t)r * ldb + col;
            if (vec && col + 3 < D) {
                const float4 x = *reinterpret_cast<const float4*>(pg);
                const float4 y = *reinterpret_cast<const float4*>(pa);
                const float4 z = *reinterpret_cast<const float4*>(pb);
                acc = fmaf(x.x, y.x - z.x, acc); acc = fmaf(x.y, y.y - z.y, acc);
                acc = fmaf(x.z, y.z - z.z, acc); acc = fmaf(x.w, y.w - z.w, acc);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (col + i < D) acc = fmaf(pg[i], pa[i] - pb[i], acc);
            }
        }
    }
    acc = wave_sum(acc);
    __shared__ float sh[4];
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)c * gridDim.y + blockIdx.y] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(64) void seg_dot_stage2(const float* __restrict__ partial, int32_t ncoltiles,
                                                      const int32_t* __restrict__ seg_chunk, float* __restrict__ out) {
    const int s = blockIdx.x;
    const int64_t p0 = (int64_t)seg_chunk[s] * ncoltiles, p1 = (int64_t)seg_chunk[s + 1] * ncoltiles;
    float acc = 0.f;
    for (int64_t i = p0 + threadIdx.x; i < p1; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[s] = acc;
}

// The weighted sums of the layer under a readout (wsi_pool_factors), stage 2: source segment s = tau * Bg + g, weight column j = b * H + hh;
// the sums land where their consumers read them - hp[seg = b * Bg + g][hh][tau][:] (the T source types of a destination segment side by
// side: the [S]-deep products with W_v) and csum[tau][seg][hh] - instead of in a [s][j] table that is then permuted by copies.
// grid = (source segments, column tiles of J * D, + 1 block for the sums of the weights).
__global__ __launch_bounds__(SEG_THREADS) void pool_factors_stage2(const float* __restrict__ partial, const float* __restrict__ wpartial,
                                                                    int32_t D, int32_t T, int32_t H, int32_t Bg, int32_t JT,
                                                                    const int32_t* __restrict__ chunk_row, const int32_t* __restrict__ seg_chunk,
                                                                    float* __restrict__ hp, float* __restrict__ csum, float* __restrict__ x_mean) {
    // JT = T * H (+ 1 with x_mean: the column of implicit weight 1 -> x_mean[s][:] = the mean of the segment's rows, as seg_stage2<MEAN> takes it)
    const int s = blockIdx.x, J = T * H;
    const int tau = s / Bg, g = s - tau * Bg;
    const int c0 = seg_chunk[s], c1 = seg_chunk[s + 1];
    if (blockIdx.y + 1 == gridDim.y) {
        for (int j = threadIdx.x; j < J; j += SEG_THREADS) {
            float acc = 0.f;
            for (int c = c0; c < c1; ++c) acc += wpartial[(int64_t)c * JT + j];
            const int b = j / H, hh = j - b * H;
            csum[((int64_t)tau * (T * Bg) + (b * Bg + g)) * H + hh] = acc;
        }
        return;
    }
    const int64_t idx = (int64_t)blockIdx.y * SEG_THREADS + threadIdx.x;
    if (idx >= (int64_t)JT * D) return;
    const int j = (int)(idx / D), col = (int)(idx - (int64_t)j * D);
    float acc = 0.f;
    for (int c = c0; c < c1; ++c) acc += partial[((int64_t)c * JT + j) * D + col];
    if (j == J) {
        const int cnt = (c1 > c0) ? chunk_row[c1] - chunk_row[c0] : 0;
        x_mean[(int64_t)s * D + col] = acc / (float)(cnt > 0 ? cnt : 1);
        return;
    }
    const int b = j / H, hh = j - b * H;
    hp[((((int64_t)(b * Bg + g)) * H + hh) * T + tau) * D + col] = acc;
}

// Skip-gate gradient, stage 2 (wsi_gate_grad): dots[s] = the segment's partials summed in order; g_skip[gate] = (1 - sigmoid(skip[gate])) * the dots
// of the segments that belong to that gate, in segment order.  One workgroup.
__global__ __launch_bounds__(256) void gate_grad_stage2(const float* __restrict__ partial, int32_t ncoltiles, const int32_t* __restrict__ seg_chunk,
                                                        int32_t num_segs, const int32_t* __restrict__ seg_gate, const float* __restrict__ skip,
                                                        int32_t n_gates, float* __restrict__ g_skip) {
    extern __shared__ float dots[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < num_segs; s += 4) {
        const int64_t p0 = (int64_t)seg_chunk[s] * ncoltiles, p1 = (int64_t)seg_chunk[s + 1] * ncoltiles;
        float acc = 0.f;
        for (int64_t i = p0 + lane; i < p1; i += 64) acc += partial[i];
        acc = wave_sum(acc);
        if (lane == 0) dots[s] = acc;
    }
    __syncthreads();
    for (int gt = threadIdx.x; gt < n_gates; gt += 256) {
        float acc = 0.f;
        for (int s = 0; s < num_segs; ++s) acc += (seg_gate[s] == gt) ? dots[s] : 0.f;
        g_skip[gt] = acc * (1.f - 1.f / (1.f + expf(-skip[gt])));
    }
}

static inline bool vec_ok(const void* p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0); }

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_segment_reduce_fwd(const float* x, int64_t ldx, int32_t D, int32_t op,
                                      const int32_t* chunk_row, int32_t num_chunks,
                                      const int32_t* seg_chunk, int32_t num_segs,
                                      float* partial, float* out, int64_t ldo, int32_t* argmax, void* stream) {
    if (D <= 0 || num_chunks < 0 || num_segs < 0 || op < 0 || op > 2) { set_error("segment_reduce_fwd: bad argument"); return WSI_EINVAL; }
    if (num_segs == 0) return WSI_OK;
    if (!chunk_row || !seg_chunk || !out || (num_chunks > 0 && (!x || !partial))) { set_error("segment_reduce_fwd: null pointer"); return WSI_EINVAL; }
    if (op == WSI_RED_MAX && !argmax) { set_error("segment_reduce_fwd: max needs argmax"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(x, ldx);
    int32_t* parg = reinterpret_cast<int32_t*>(partial + (int64_t)num_chunks * D);
    const dim3 g1(num_chunks, (D + 255) / 256), g2(num_segs, (D + SEG_THREADS - 1) / SEG_THREADS);
    if (op == WSI_RED_SUM) {
        if (num_chunks) hipLaunchKernelGGL(seg_stage1<WSI_RED_SUM>, g1, dim3(SEG_THREADS), 0, st, x, ldx, D, vec, chunk_row, partial, (int32_t*)nullptr);
        hipLaunchKernelGGL(seg_stage2<WSI_RED_SUM>, g2, dim3(SEG_THREADS), 0, st, (const float*)partial, (const int32_t*)nullptr, D, chunk_row, seg_chunk, out, ldo, (int32_t*)nullptr);
    } else if (op == WSI_RED_MEAN) {
        if (num_chunks) hipLaunchKernelGGL(seg_stage1<WSI_RED_SUM>, g1, dim3(SEG_THREADS), 0, st, x, ldx, D, vec, chunk_row, partial, (int32_t*)nullptr);
        hipLaunchKernelGGL(seg_stage2<WSI_RED_MEAN>, g2, dim3(SEG_THREADS), 0, st, (const float*)partial, (const int32_t*)nullptr, D, chunk_row, seg_chunk, out, ldo, (int32_t*)nullptr);
    } else {
        if (num_chunks) hipLaunchKernelGGL(seg_stage1<WSI_RED_MAX>, g1, dim3(SEG_THREADS), 0, st, x, ldx, D, vec, chunk_row, partial, parg);
        hipLaunchKernelGGL(seg_stage2<WSI_RED_MAX>, g2, dim3(SEG_THREADS), 0, st, (const float*)partial, (const int32_t*)parg, D, chunk_row, seg_chunk, out, ldo, argmax);
    }
    return check_launch("segment_reduce_fwd");
}

extern "C" int wsi_segment_reduce_bwd(const float* gout, int64_t ldgo, int32_t D, int32_t op,
                                      const int32_t* chunk_row, const int32_t* chunk_seg, int32_t num_chunks,
                                      const int32_t* seg_chunk, int32_t num_segs,
                                      const int32_t* argmax, float* gx, int64_t ldgx, void* stream) {
    if (D <= 0 || num_chunks < 0 || num_segs < 0 || op < 0 || op > 2) { set_error("segment_reduce_bwd: bad argument"); return WSI_EINVAL; }
    if (num_segs == 0 || num_chunks == 0) return WSI_OK;
    if (!gout || !gx || !chunk_row || !chunk_seg || !seg_chunk) { set_error("segment_reduce_bwd: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (op == WSI_RED_MAX) {
        if (!argmax) { set_error("segment_reduce_bwd: max needs argmax"); return WSI_EINVAL; }
        hipLaunchKernelGGL(seg_bwd_max, dim3(num_segs, (D + SEG_THREADS - 1) / SEG_THREADS), dim3(SEG_THREADS), 0, st, gout, ldgo, D, argmax, gx, ldgx);
    } else {
        hipLaunchKernelGGL(seg_bwd_bcast, dim3(num_chunks, (D + 255) / 256), dim3(SEG_THREADS), 0, st, gout, ldgo, D, op,
                           chunk_row, chunk_seg, seg_chunk, vec_ok(gx, ldgx), gx, ldgx);
    }
    return check_launch("segment_reduce_bwd");
}

extern "C" int wsi_segment_dot_diff(const float* g, int64_t ldg, const float* a, int64_t lda, const float* b, int64_t ldb,
                                    int32_t D, const int32_t* chunk_row, int32_t num_chunks,
                                    const int32_t* seg_chunk, int32_t num_segs,
                                    float* partial, float* out, void* stream) {
    if (D <= 0 || num_chunks < 0 || num_segs < 0) { set_error("segment_dot_diff: bad argument"); return WSI_EINVAL; }
    if (num_segs == 0) return WSI_OK;
    if (!chunk_row || !seg_chunk || !out || (num_chunks > 0 && (!g || !a || !b || !partial))) { set_error("segment_dot_diff: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nct = (D + 255) / 256;
    const bool vec = vec_ok(g, ldg) && vec_ok(a, lda) && vec_ok(b, ldb);
    if (num_chunks) hipLaunchKernelGGL(seg_dot_stage1, dim3(num_chunks, nct), dim3(SEG_THREADS), 0, st, g, ldg, a, lda, b, ldb, D, vec, chunk_row, partial);
    hipLaunchKernelGGL(seg_dot_stage2, dim3(num_segs), dim3(64), 0, st, (const float*)partial, nct, seg_chunk, out);
    return check_launch("segment_dot_diff");
}

// out[s, j, :] = sum over the rows r of segment s of w[r, j] * x[r, :]  -  see include/wsi_hgnn.h
extern "C" int wsi_segment_weighted_sums(const float* x, int64_t ldx, int32_t D, const float* w, int64_t ldw, int32_t J,
                                         const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk, int32_t num_segs,
                                         float* partial, float* out, void* stream) {
    if (D <= 0 || J <= 0 || J > 1024 || num_chunks < 0 || num_segs < 0) { set_error("segment_weighted_sums: bad argument"); return WSI_EINVAL; }
    if (num_segs == 0) return WSI_OK;
    if (!chunk_row || !seg_chunk || !out || (num_chunks > 0 && (!x || !w || !partial))) { set_error("segment_weighted_sums: null pointer"); return WSI_EINVAL; }
    if ((int64_t)J * D > INT32_MAX) { set_error("segment_weighted_sums: J * D too large"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(x, ldx);
    const int32_t JD = J * D;
    const dim3 g1(num_chunks, (D + 255) / 256, (J + WSUM_J - 1) / WSUM_J), g2(num_segs, (JD + SEG_THREADS - 1) / SEG_THREADS);
    if (num_chunks) hipLaunchKernelGGL(seg_wsum_stage1, g1, dim3(SEG_THREADS), 0, st, x, ldx, D, vec, w, ldw, J, J, chunk_row, partial, (float*)nullptr);
    hipLaunchKernelGGL(seg_stage2<WSI_RED_SUM>, g2, dim3(SEG_THREADS), 0, st, (const float*)partial, (const int32_t*)nullptr, JD, chunk_row, seg_chunk,
                       out, (int64_t)JD, (int32_t*)nullptr);
    return check_launch("segment_weighted_sums");
}

// hp / csum of the layer under a readout in two launches - see include/wsi_hgnn.h
extern "C" int wsi_pool_factors(const float* x, int64_t ldx, int32_t D, const float* w, int64_t ldw, int32_t T, int32_t H, int32_t Bg,
                                const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk,
                                float* partial, float* hp, float* csum, float* x_mean, void* stream) {
    if (D <= 0 || T <= 0 || H <= 0 || Bg <= 0 || num_chunks < 0 || (int64_t)T * H > 1024) { set_error("pool_factors: bad argument"); return WSI_EINVAL; }
    if (!chunk_row || !seg_chunk || !hp || !csum || (num_chunks > 0 && (!x || !w || !partial))) { set_error("pool_factors: null pointer"); return WSI_EINVAL; }
    const int32_t J = T * H, JT = J + (x_mean ? 1 : 0);
    if ((int64_t)JT * D > INT32_MAX) { set_error("pool_factors: T * H * D too large"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    float* wpartial = partial + (int64_t)num_chunks * JT * D;
    const dim3 g1(num_chunks, (D + 255) / 256, (JT + WSUM_J - 1) / WSUM_J);
    const dim3 g2(T * Bg, (uint32_t)(((int64_t)JT * D + SEG_THREADS - 1) / SEG_THREADS) + 1);
    if (num_chunks) hipLaunchKernelGGL(seg_wsum_stage1, g1, dim3(SEG_THREADS), 0, st, x, ldx, D, vec_ok(x, ldx), w, ldw, J, JT, chunk_row, partial, wpartial);
    hipLaunchKernelGGL(pool_factors_stage2, g2, dim3(SEG_THREADS), 0, st, (const float*)partial, (const float*)wpartial, D, T, H, Bg, JT, chunk_row, seg_chunk,
                       hp, csum, x_mean);
    return check_launch("pool_factors");
}

// g_skip of a HEAT layer: segment dots of g * (a - b) + the gate map + (1 - sigmoid) in two launches - see include/wsi_hgnn.h
extern "C" int wsi_gate_grad(const float* g, int64_t ldg, const float* a, int64_t lda, const float* b, int64_t ldb,
                             int32_t D, const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk, int32_t num_segs,
                             const int32_t* seg_gate, const float* skip, int32_t n_gates, float* partial, float* g_skip, void* stream) {
    if (D <= 0 || num_chunks < 0 || num_segs < 0 || n_gates <= 0 || num_segs > 8192) { set_error("gate_grad: bad argument"); return WSI_EINVAL; }
    if (!chunk_row || !seg_chunk || !seg_gate || !skip || !g_skip || (num_chunks > 0 && (!g || !a || !b || !partial))) { set_error("gate_grad: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nct = (D + 255) / 256;
    const bool vec = vec_ok(g, ldg) && vec_ok(a, lda) && vec_ok(b, ldb);
    if (num_chunks) hipLaunchKernelGGL(seg_dot_stage1, dim3(num_chunks, nct), dim3(SEG_THREADS), 0, st, g, ldg, a, lda, b, ldb, D, vec, chunk_row, partial);
    hipLaunchKernelGGL(gate_grad_stage2, dim3(1), dim3(256), (size_t)(num_segs > 0 ? num_segs : 1) * sizeof(float), st, (const float*)partial, nct, seg_chunk,
                       num_segs, seg_gate, skip, n_gates, g_skip);
    return check_launch("gate_grad");
}
