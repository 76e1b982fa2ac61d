// Row-wise kernels for the HGT / GCN siblings of the HEAT path (gfx950): LayerNorm, GELU, and the
// degree-normalised neighbour sum of DGL's GraphConv.  Contracts + reference call sites: include/wsi_hgnn.h.
// All are HBM-bound streaming / gather kernels: one 64-lane wave per row, lane l owns elements
// l, l+64, ... (256-byte coalesced wave accesses for any feature width), wave reductions by DPP/bpermute.
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int RW_BLOCK = 256;
constexpr int RW_WAVES = RW_BLOCK / 64;

__device__ __forceinline__ int row_of_wave(int n) {
    int w = (int)blockIdx.x * RW_WAVES + (int)(threadIdx.x >> 6);
    w = __builtin_amdgcn_readfirstlane(w);
    return w < n ? w : -1;
}

// ------------------------------------------------------------------------------------------ LayerNorm
// y = (x - mean) * rstd * gamma + beta over the last dim (eps inside the sqrt, biased variance: torch.nn.LayerNorm,
// models/HGT.py:57,124).  gamma/beta are selected per row through row_param[row] (node type -> norms[n_id]).
template <int NV>
__global__ __launch_bounds__(RW_BLOCK) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx, int n, int D, float eps,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const int32_t* __restrict__ row_param,
                                                                  float* __restrict__ y, int64_t ldy, float* __restrict__ stats) {
    const int r = row_of_wave(n);
    if (r < 0) return;
    const int lane = threadIdx.x & 63;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; v[i] = c < D ? x[(int64_t)r * ldx + c] : 0.f; s += v[i]; }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; const float d = c < D ? v[i] - mean : 0.f; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const int pi = row_param ? row_param[r] : 0;
    const float* g = gamma + (int64_t)pi * D;
    const float* b = beta + (int64_t)pi * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) y[(int64_t)r * ldy + c] = (v[i] - mean) * rstd * g[c] + b[c];
    }
    if (lane == 0) { stats[2 * (int64_t)r] = mean; stats[2 * (int64_t)r + 1] = rstd; }
}

// gx = rstd * (gg - mean(gg) - xhat * mean(gg*xhat)),  gg = gy*gamma;  xhat_gy = gy * xhat (for d gamma = colsum)
template <int NV>
__global__ __launch_bounds__(RW_BLOCK) void layernorm_bwd_kernel(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ x, int64_t ldx,
                                                                  int n, int D, const float* __restrict__ gamma, const int32_t* __restrict__ row_param,
                                                                  const float* __restrict__ stats, float* __restrict__ gx, int64_t ldgx,
                                                                  float* __restrict__ xhat_gy, int64_t ldp) {
    const int r = row_of_wave(n);
    if (r < 0) return;
    const int lane = threadIdx.x & 63;
    const float mean = stats[2 * (int64_t)r], rstd = stats[2 * (int64_t)r + 1];
    const float* g = gamma + (int64_t)(row_param ? row_param[r] : 0) * D;
    float xh[NV], gg[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            const float go = gy[(int64_t)r * ldgy + c];
            xh[i] = (x[(int64_t)r * ldx + c] - mean) * rstd;
            gg[i] = go * g[c];
            xhat_gy[(int64_t)r * ldp + c] = go * xh[i];
            s1 += gg[i];
            s2 = fmaf(gg[i], xh[i], s2);
        } else { xh[i] = 0.f; gg[i] = 0.f; }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) gx[(int64_t)r * ldgx + c] = rstd * (gg[i] - m1 - xh[i] * m2);
    }
}

// ------------------------------------------------------------------------------------------ GELU (exact erf form)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        gx[i] = gy[i] * (cdf + v * pdf);
    }
}

// ------------------------------------------------------------------------------------------ neighbour sum (GraphConv)
// out[w] = act( oscale[w] * sum_{e in [ptr[w],ptr[w+1])} iscale[idx[e]] * x[idx[e]]  (+ bias) )
// One wave per output row; used with the CSR-by-dst for forward and the CSC-by-src for backward
// (DGL GraphConv norm='both': iscale = outdeg^-1/2, oscale = indeg^-1/2; models/GCN.py:30-33).
// relu_mask != NULL (backward): x rows are first multiplied by the mask row (y > 0) of the gathered node.
template <int NV>
__global__ __launch_bounds__(RW_BLOCK) void spmm_sum_kernel(const float* __restrict__ x, int64_t ldx, int n_out, int D,
                                                             const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ edge_w,
                                                             const float* __restrict__ iscale, const float* __restrict__ oscale,
                                                             const float* __restrict__ bias, int relu,
                                                             const float* __restrict__ relu_ref, int64_t ldref,
                                                             float* __restrict__ out, int64_t ldo) {
    const int w = row_of_wave(n_out);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int e0 = ptr[w], e1 = ptr[w + 1];
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
        const float sc = (iscale ? iscale[u] : 1.f) * (edge_w ? edge_w[e] : 1.f);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                float v = x[(int64_t)u * ldx + c];
                if (relu_ref) v = relu_ref[(int64_t)u * ldref + c] > 0.f ? v : 0.f;
                acc[i] = fmaf(sc, v, acc[i]);
            }
        }
    }
    const float os = oscale ? oscale[w] : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            float v = acc[i] * os + (bias ? bias[c] : 0.f);
            if (relu) v = fmaxf(v, 0.f);
            out[(int64_t)w * ldo + c] = v;
        }
    }
}

#define WSI_NV_DISPATCH(D, CALL)                 \
    {                                            \
        const int nv_ = ((D) + 63) / 64;         \
        if (nv_ <= 1) { CALL(1); }               \
        else if (nv_ <= 2) { CALL(2); }          \
        else if (nv_ <= 4) { CALL(4); }          \
        else if (nv_ <= 8) { CALL(8); }          \
        else if (nv_ <= 16) { CALL(16); }        \
        else { set_error("feature width %d > 1024 unsupported", (D)); return WSI_ENOSYS; } \
    }

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_layernorm_fwd(const float* x, int64_t ldx, int32_t n, int32_t D, float eps,
                                 const float* gamma, const float* beta, const int32_t* row_param,
                                 float* y, int64_t ldy, float* stats, void* stream) {
    if (n < 0 || D <= 0) { set_error("layernorm_fwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!x || !gamma || !beta || !y || !stats) { set_error("layernorm_fwd: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n + RW_WAVES - 1) / RW_WAVES;
#define CALL(NV) hipLaunchKernelGGL((layernorm_fwd_kernel<NV>), dim3(blocks), dim3(RW_BLOCK), 0, st, x, ldx, n, D, eps, gamma, beta, row_param, y, ldy, stats)
    WSI_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("layernorm_fwd");
}

extern "C" int wsi_layernorm_bwd(const float* gy, int64_t ldgy, const float* x, int64_t ldx, int32_t n, int32_t D,
                                 const float* gamma, const int32_t* row_param, const float* stats,
                                 float* gx, int64_t ldgx, float* xhat_gy, int64_t ldp, void* stream) {
    if (n < 0 || D <= 0) { set_error("layernorm_bwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!gy || !x || !gamma || !stats || !gx || !xhat_gy) { set_error("layernorm_bwd: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n + RW_WAVES - 1) / RW_WAVES;
#define CALL(NV) hipLaunchKernelGGL((layernorm_bwd_kernel<NV>), dim3(blocks), dim3(RW_BLOCK), 0, st, gy, ldgy, x, ldx, n, D, gamma, row_param, stats, gx, ldgx, xhat_gy, ldp)
    WSI_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("layernorm_bwd");
}

extern "C" int wsi_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!x || !y))) { set_error("gelu_fwd: bad argument"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("gelu_fwd");
}

extern "C" int wsi_gelu_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!x || !gy || !gx))) { set_error("gelu_bwd: bad argument"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n);
    return check_launch("gelu_bwd");
}

extern "C" int wsi_spmm_sum(const float* x, int64_t ldx, int32_t n_out, int32_t D,
                            const int32_t* ptr, const int32_t* idx, const float* edge_w, const float* iscale, const float* oscale,
                            const float* bias, int32_t relu, const float* relu_ref, int64_t ldref,
                            float* out, int64_t ldo, void* stream) {
    if (n_out < 0 || D <= 0) { set_error("spmm_sum: bad shape"); return WSI_EINVAL; }
    if (n_out == 0) return WSI_OK;
    if (!x || !ptr || !out) { set_error("spmm_sum: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n_out + RW_WAVES - 1) / RW_WAVES;
#define CALL(NV) hipLaunchKernelGGL((spmm_sum_kernel<NV>), dim3(blocks), dim3(RW_BLOCK), 0, st, x, ldx, n_out, D, ptr, idx, edge_w, iscale, oscale, bias, relu, relu_ref, ldref, out, ldo)
    WSI_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("spmm_sum");
}

// ------------------------------------------------------------------------------------------------
// wsi_dropout_apply: out = keep(seed, row, col) ? x * scale : 0 with the counter-based mask of WSI_EPI_DROPOUT (gemm_common.h): the backward of the
// nn.Dropout of models/HEATNet4.py:135 (g_y = g_out * mask) without a stored mask.  HBM-bound streaming: 16-byte lane accesses, two hashes per four columns.
#include "gemm_common.h"
namespace wsi {
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo, int rows, int cols,
                                                            int row0, uint32_t pairs, int col0, uint32_t seed0, const uint32_t* __restrict__ seed_base, uint32_t thr, float scale, int vec) {
    const uint32_t seed = seed0 + (seed_base ? *seed_base : 0u);
    const int c4n = (cols + 3) >> 2;
    const int64_t total = (int64_t)rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / c4n), c = (int)(i - (int64_t)r * c4n) * 4;
        if (vec && c + 3 < cols) {
            const float4 m = drop_factor4((uint32_t)(row0 + r), (uint32_t)(col0 + c), pairs, seed, thr, scale);
            float4 v = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) = v;
        } else {
            for (int k = 0; k < 4 && c + k < cols; ++k)
                out[(int64_t)r * ldo + c + k] = x[(int64_t)r * ldx + c + k] * drop_factor1((uint32_t)(row0 + r), (uint32_t)(col0 + c + k), pairs, seed, thr, scale);
        }
    }
}
}  // namespace wsi

extern "C" int wsi_dropout_apply(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t rows, int32_t cols, int32_t row0, int32_t tensor_cols, int32_t col0,
                                 uint32_t seed, const uint32_t* seed_base, uint32_t threshold, float scale, void* stream) {
    using namespace wsi;
    if (rows < 0 || cols < 0 || row0 < 0 || col0 < 0 || tensor_cols <= 0 || col0 + cols > tensor_cols || threshold > 65536u) { set_error("dropout_apply: bad argument"); return WSI_EINVAL; }
    if (rows == 0 || cols == 0) return WSI_OK;
    if (!x || !out) { set_error("dropout_apply: null pointer"); return WSI_EINVAL; }
    const int vec = (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && ldx % 4 == 0 && ldo % 4 == 0 && col0 % 4 == 0) ? 1 : 0;
    const int64_t total = (int64_t)rows * ((cols + 3) / 4);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, (int)rows, (int)cols, (int)row0,
                       (uint32_t)((tensor_cols + 1) / 2), (int)col0, seed, seed_base, threshold, scale, vec);
    return check_launch("dropout_apply");
}
