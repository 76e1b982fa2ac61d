// The S-row algebra of a HEAT layer under a sum / mean readout: what is left of the last layer's output stage, V projection and
// their backward once they act on S = graphs x node-types rows instead of N.  Each of these is a handful of [S, D]-sized products and sums;
// as framework tensor operations they were ~30 launches of 4-6 us per step, here they are three.  Contracts: include/wsi_hgnn.h.
// Plain fp32 in a fixed order (deterministic).  T = node types, H = heads, dk = D / H, segment s = type * Bg + graph.
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int POOL_MAX_TYPES = 16;
struct PtrList { const float* p[POOL_MAX_TYPES]; };

// t_mean[s, c] (+)= ( sum_tau tpart[tau, s, c] + sum_tau csum[tau, s, c / dk] * bv[tau][c] ) ;  * scale[s] when `finish`
// grid = (S, column tiles of 256)
__global__ __launch_bounds__(256) void pool_tmean_kernel(const float* __restrict__ tpart, int tau0, int nt, int T, int S, int D, int H,
                                                         const float* __restrict__ csum, const PtrList bv, const float* __restrict__ scale,
                                                         int accumulate, int finish, float* __restrict__ t_mean) {
    const int s = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    const int hh = c / (D / H);
    float acc = accumulate ? t_mean[(int64_t)s * D + c] : 0.f;
    for (int i = 0; i < nt; ++i) {
        const int tau = tau0 + i;
        acc += tpart[((int64_t)tau * S + s) * D + c];
        if (bv.p[i]) acc = fmaf(csum[((int64_t)tau * S + s) * H + hh], bv.p[i][c], acc);
    }
    if (finish && scale) acc *= scale[s];
    t_mean[(int64_t)s * D + c] = acc;
}

// The head of the pooled layer's backward.  One workgroup (S * D is a few thousand elements):
//   g_row[s, :] = g_pool[s, :] * (mean ? 1 / count[s] (0 for an empty segment) : 1)        the gradient of every output row of the segment
//   g_sum[s, :] = g_pool[s, :] * (mean ? [count[s] > 0] : count[s])                         ... summed over the segment's rows
//   g_skip[gate] = (1 - sigmoid(skip[gate])) * sum over the segments s of that gate of  g_sum[s, :] . (z_mean[s, :] - h_mean[s, :])
//   omg[i] = 1 - sigmoid(skip[type_gate[i]])   (1 for a type the layer passes through: type_gate[i] < 0)
__global__ __launch_bounds__(256) void pool_bwd_prep_kernel(const float* __restrict__ g_pool, int S, int D, int mean, const float* __restrict__ counts,
                                                            const float* __restrict__ z_mean, const float* __restrict__ h_mean,
                                                            const int32_t* __restrict__ seg_gate, const float* __restrict__ skip, int n_gates,
                                                            const int32_t* __restrict__ type_gate, int T,
                                                            float* __restrict__ g_row, float* __restrict__ g_sum, float* __restrict__ g_skip,
                                                            float* __restrict__ omg) {
    extern __shared__ float dots[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < S; s += 4) {
        const float cnt = counts[s];
        const float fr = mean ? (cnt > 0.f ? 1.f / cnt : 0.f) : 1.f;
        const float fs = mean ? (cnt > 0.f ? 1.f : 0.f) : cnt;
        float acc = 0.f;
        for (int c = lane; c < D; c += 64) {
            const int64_t i = (int64_t)s * D + c;
            const float g = g_pool[i];
            const float gs = g * fs;
            g_row[i] = g * fr;
            g_sum[i] = gs;
            acc = fmaf(gs, z_mean[i] - h_mean[i], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) dots[s] = acc;
    }
    __syncthreads();
    for (int gt = threadIdx.x; gt < n_gates; gt += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += (seg_gate[s] == gt) ? dots[s] : 0.f;
        g_skip[gt] = acc * (1.f - 1.f / (1.f + expf(-skip[gt])));
    }
    for (int i = threadIdx.x; i < T; i += 256) {
        const int gt = type_gate[i];
        omg[i] = gt >= 0 ? 1.f - 1.f / (1.f + expf(-skip[gt])) : 1.f;
    }
}

// beta[tau, s, hh] = sum over the columns c of head hh of gt_seg[s, c] * bv[tau][c]        blocks [0, nt * S): one per (tau, s), one wave per head
// gbv[tau, c]      = sum_s csum[tau, s, c / dk] * gt_seg[s, c]                              blocks [nt * S, + nt * column tiles)
__global__ __launch_bounds__(256) void pool_bwd_bias_kernel(const float* __restrict__ gt_seg, int tau0, int nt, int S, int D, int H, const PtrList bv,
                                                            const float* __restrict__ csum, float* __restrict__ beta, float* __restrict__ gbv) {
    const int dk = D / H;
    int blk = blockIdx.x;
    if (beta) {
        if (blk < nt * S) {
            const int i = blk / S, s = blk - i * S, tau = tau0 + i;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            for (int hh = wave; hh < H; hh += 4) {
                float acc = 0.f;
                if (bv.p[i])
                    for (int c = lane; c < dk; c += 64) acc = fmaf(gt_seg[(int64_t)s * D + hh * dk + c], bv.p[i][hh * dk + c], acc);
                acc = wave_sum(acc);
                if (lane == 0) beta[((int64_t)tau * S + s) * H + hh] = acc;
            }
            return;
        }
        blk -= nt * S;
    }
    if (!gbv) return;
    const int tiles = (D + 255) / 256;
    const int i = blk / tiles, c = (blk - i * tiles) * 256 + threadIdx.x, tau = tau0 + i;
    if (i >= nt || c >= D) return;
    const int hh = c / dk;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc = fmaf(csum[((int64_t)tau * S + s) * H + hh], gt_seg[(int64_t)s * D + c], acc);
    gbv[(int64_t)tau * D + c] = acc;
}

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_pool_tmean(const float* tpart, int32_t T, int32_t S, int32_t D, int32_t H, const float* csum, const float* const* bv,
                              const float* scale, float* t_mean, void* stream) {
    if (T <= 0 || S <= 0 || D <= 0 || H <= 0 || D % H) { set_error("pool_tmean: bad argument"); return WSI_EINVAL; }
    if (!tpart || !csum || !bv || !t_mean) { set_error("pool_tmean: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < T; t0 += POOL_MAX_TYPES) {
        const int nt = (T - t0 < POOL_MAX_TYPES) ? T - t0 : POOL_MAX_TYPES;
        PtrList L;
        for (int i = 0; i < POOL_MAX_TYPES; ++i) L.p[i] = (i < nt) ? bv[t0 + i] : nullptr;
        hipLaunchKernelGGL(pool_tmean_kernel, dim3(S, (D + 255) / 256), dim3(256), 0, st, tpart, t0, nt, (int)T, (int)S, (int)D, (int)H, csum, L, scale,
                           t0 > 0 ? 1 : 0, t0 + nt >= T ? 1 : 0, t_mean);
    }
    return check_launch("pool_tmean");
}

extern "C" int wsi_pool_bwd_prep(const float* g_pool, int32_t S, int32_t D, int32_t op, const float* counts, const float* z_mean, const float* h_mean,
                                 const int32_t* seg_gate, const float* skip, int32_t n_gates, const int32_t* type_gate, int32_t T,
                                 float* g_row, float* g_sum, float* g_skip, float* omg, void* stream) {
    if (S <= 0 || S > 8192 || D <= 0 || n_gates <= 0 || T <= 0 || (op != WSI_RED_SUM && op != WSI_RED_MEAN)) { set_error("pool_bwd_prep: bad argument"); return WSI_EINVAL; }
    if (!g_pool || !counts || !z_mean || !h_mean || !seg_gate || !skip || !type_gate || !g_row || !g_sum || !g_skip || !omg) {
        set_error("pool_bwd_prep: null pointer"); return WSI_EINVAL;
    }
    hipLaunchKernelGGL(pool_bwd_prep_kernel, dim3(1), dim3(256), (size_t)S * sizeof(float), (hipStream_t)stream, g_pool, (int)S, (int)D,
                       op == WSI_RED_MEAN ? 1 : 0, counts, z_mean, h_mean, seg_gate, skip, (int)n_gates, type_gate, (int)T, g_row, g_sum, g_skip, omg);
    return check_launch("pool_bwd_prep");
}

extern "C" int wsi_pool_bwd_bias(const float* gt_seg, int32_t T, int32_t S, int32_t D, int32_t H, const float* const* bv, const float* csum,
                                 float* beta, float* gbv, void* stream) {
    if (T <= 0 || S <= 0 || D <= 0 || H <= 0 || D % H) { set_error("pool_bwd_bias: bad argument"); return WSI_EINVAL; }
    if (!gt_seg || (beta && !bv) || (gbv && !csum)) { set_error("pool_bwd_bias: null pointer"); return WSI_EINVAL; }
    if (!beta && !gbv) return WSI_OK;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (D + 255) / 256;
    for (int t0 = 0; t0 < T; t0 += POOL_MAX_TYPES) {
        const int nt = (T - t0 < POOL_MAX_TYPES) ? T - t0 : POOL_MAX_TYPES;
        PtrList L;
        for (int i = 0; i < POOL_MAX_TYPES; ++i) L.p[i] = (beta && i < nt) ? bv[t0 + i] : nullptr;
        const int blocks = (beta ? nt * S : 0) + (gbv ? nt * tiles : 0);
        hipLaunchKernelGGL(pool_bwd_bias_kernel, dim3(blocks), dim3(256), 0, st, gt_seg, t0, nt, (int)S, (int)D, (int)H, L, csum, beta, gbv);
    }
    return check_launch("pool_bwd_bias");
}
