// Edge kernels of ASAPPooling (pooling/ASAP.py:142-199) for gfx950.  Contracts: include/wsi_hgnn.h.
// Same shape as rowwise.hip: one 64-lane wave per graph node, lane l owns feature columns l, l+64, ... (256-byte
// coalesced wave accesses for any width <= 1024); edges arrive grouped by the node that aggregates them (CSR) and, for the
// backward passes that reduce onto the gathered node, by that node (CSC) — no atomics, fixed summation order.
//   pooling/ASAP.py:158,163  x_pool_j = x_pool[j]; X_q = scatter_max(x_pool_j, i)           -> wsi_csr_gather_max_fwd/bwd
//   pooling/ASAP.py:167-179  score = softmax_i(leaky_relu(gat_att([M_q[i], x_pool[j]])));    -> wsi_asap_attend_fwd/bwd
//                            out = scatter_add(x[j] * score, i)
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int AS_BLOCK = 256;
constexpr int AS_WAVES = AS_BLOCK / 64;

__device__ __forceinline__ int as_row(int n) {
    int w = (int)blockIdx.x * AS_WAVES + (int)(threadIdx.x >> 6);
    w = __builtin_amdgcn_readfirstlane(w);
    return w < n ? w : -1;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x = fmaxf(x, __shfl_xor(x, m));
    return x;
}

// out[w,c] = max over the segment of x[idx[e],c]; arg[w,c] = CSR position e of the FIRST maximum; empty segment: 0 / -1
// (torch_scatter's scatter(..., reduce='max') leaves 0 where nothing was scattered).
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void gather_max_fwd_kernel(const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                  const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                                  float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    float best[NV];
    int where[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { best[i] = -INFINITY; where[i] = -1; }
    const int e0 = ptr[w], e1 = ptr[w + 1];
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float v = x[(int64_t)u * ldx + c];
                if (v > best[i] || where[i] < 0) { best[i] = v; where[i] = e; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            out[(int64_t)w * ldo + c] = where[i] < 0 ? 0.f : best[i];
            arg[(int64_t)w * D + c] = where[i];
        }
    }
}

// gx[u,c] = sum over the out-edges (u -> w, CSR position eid) of g[w,c] where arg[w,c] == eid      (CSC by gathered node u)
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void gather_max_bwd_kernel(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ arg,
                                                                  int n_src, int D, const int32_t* __restrict__ colptr,
                                                                  const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
                                                                  float* __restrict__ gx, int64_t ldgx) {
    const int u = as_row(n_src);
    if (u < 0) return;
    const int lane = threadIdx.x & 63;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int eid = csc_eid[j], w = csc_dst[j];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D && arg[(int64_t)w * D + c] == eid) acc[i] += g[(int64_t)w * ldg + c];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) gx[(int64_t)u * ldgx + c] = acc[i];
    }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// score[e] = exp(s_e - max) / (sum exp + 1e-16), s_e = leaky_relu(a[w] + b[idx[e]])  (torch_geometric.utils.softmax);
// out[w,:] = sum_e score[e] * x[idx[e],:]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                   const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                   const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, float slope,
                                                                   float* __restrict__ score, float* __restrict__ out, int64_t ldo) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int e0 = ptr[w], e1 = ptr[w + 1];
    const float aw = a[w];
    float m = -INFINITY;
    for (int e = e0 + lane; e < e1; e += 64) m = fmaxf(m, leaky(aw + b[idx[e]], slope));
    m = wave_max(m);
    float l = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) l += expf(leaky(aw + b[idx[e]], slope) - m);
    l = wave_sum(l);
    const float inv = 1.f / (l + 1e-16f);
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
        const float p = expf(leaky(aw + b[u], slope) - m) * inv;
        if (lane == 0) score[e] = p;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) acc[i] = fmaf(p, x[(int64_t)u * ldx + c], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) out[(int64_t)w * ldo + c] = acc[i];
    }
}

// dst-major backward: gp_e = g_out[w,:] . x[idx[e],:];  g_s_e = p_e (gp_e - sum_k p_k gp_k);
//                     gpre[e] = g_s_e * (pre_e > 0 ? 1 : slope);  g_a[w] = sum_e gpre[e]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_bwd_dst_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                       const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, float slope,
                                                                       const float* __restrict__ score, const float* __restrict__ g_out, int64_t ldg,
                                                                       float* __restrict__ gpre, float* __restrict__ g_a) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int e0 = ptr[w], e1 = ptr[w + 1];
    float go[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; go[i] = c < D ? g_out[(int64_t)w * ldg + c] : 0.f; }
    float delta = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; if (c < D) d = fmaf(go[i], x[(int64_t)u * ldx + c], d); }
        d = wave_sum(d);
        if (lane == 0) gpre[e] = d;                        // gp_e, finished below
        delta = fmaf(score[e], d, delta);
    }
    __threadfence_block();                                 // lane 0's gp_e stores must be visible to the other lanes of this wave
    const float aw = a[w];
    float ga = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) {
        const float pre = aw + b[idx[e]];
        const float gs = score[e] * (gpre[e] - delta) * (pre > 0.f ? 1.f : slope);
        gpre[e] = gs;
        ga += gs;
    }
    ga = wave_sum(ga);
    if (lane == 0) g_a[w] = ga;
}

// src-major backward (CSC by gathered node u):  g_x[u,:] = sum score[eid] * g_out[dst,:];  g_b[u] = sum gpre[eid]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_bwd_src_kernel(const float* __restrict__ g_out, int64_t ldg, int n_src, int D,
                                                                       const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid,
                                                                       const int32_t* __restrict__ csc_dst, const float* __restrict__ score,
                                                                       const float* __restrict__ gpre, float* __restrict__ gx, int64_t ldgx,
                                                                       float* __restrict__ g_b) {
    const int u = as_row(n_src);
    if (u < 0) return;
    const int lane = threadIdx.x & 63;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    float gb = 0.f;
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int eid = csc_eid[j], w = csc_dst[j];
        const float p = score[eid];
        gb += gpre[eid];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) acc[i] = fmaf(p, g_out[(int64_t)w * ldg + c], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) gx[(int64_t)u * ldgx + c] = acc[i];
    }
    if (lane == 0) g_b[u] = gb;
}

#define AS_NV_DISPATCH(D, CALL)                  \
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

extern "C" int wsi_csr_gather_max_fwd(const float* x, int64_t ldx, int32_t n, int32_t D, const int32_t* ptr, const int32_t* idx,
                                      float* out, int64_t ldo, int32_t* arg, void* stream) {
    if (n < 0 || D <= 0) { set_error("csr_gather_max_fwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!x || !ptr || !out || !arg) { set_error("csr_gather_max_fwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), b(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((gather_max_fwd_kernel<NV>), g, b, 0, st, x, ldx, n, D, ptr, idx, out, ldo, arg)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("csr_gather_max_fwd");
}

extern "C" int wsi_csr_gather_max_bwd(const float* g_out, int64_t ldg, const int32_t* arg, int32_t n_src, int32_t D,
                                      const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                                      float* gx, int64_t ldgx, void* stream) {
    if (n_src < 0 || D <= 0) { set_error("csr_gather_max_bwd: bad shape"); return WSI_EINVAL; }
    if (n_src == 0) return WSI_OK;
    if (!g_out || !arg || !colptr || !gx) { set_error("csr_gather_max_bwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n_src + AS_WAVES - 1) / AS_WAVES), b(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((gather_max_bwd_kernel<NV>), g, b, 0, st, g_out, ldg, arg, n_src, D, colptr, csc_eid, csc_dst, gx, ldgx)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("csr_gather_max_bwd");
}

extern "C" int wsi_asap_attend_fwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                                   const int32_t* ptr, const int32_t* idx, float negative_slope,
                                   float* score, float* out, int64_t ldo, void* stream) {
    if (n < 0 || D <= 0) { set_error("asap_attend_fwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!a || !b || !x || !ptr || !score || !out) { set_error("asap_attend_fwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), bl(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((asap_attend_fwd_kernel<NV>), g, bl, 0, st, a, b, x, ldx, n, D, ptr, idx, negative_slope, score, out, ldo)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("asap_attend_fwd");
}

extern "C" int wsi_asap_attend_bwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                                   const int32_t* ptr, const int32_t* idx,
                                   const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, float negative_slope,
                                   const float* score, const float* g_out, int64_t ldg,
                                   float* gpre, float* g_a, float* g_b, float* gx, int64_t ldgx, void* stream) {
    if (n < 0 || D <= 0) { set_error("asap_attend_bwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!a || !b || !x || !ptr || !colptr || !score || !g_out || !gpre || !g_a || !g_b || !gx) { set_error("asap_attend_bwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), bl(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) { hipLaunchKernelGGL((asap_attend_bwd_dst_kernel<NV>), g, bl, 0, st, a, b, x, ldx, n, D, ptr, idx, negative_slope, score, g_out, ldg, gpre, g_a); \
                   hipLaunchKernelGGL((asap_attend_bwd_src_kernel<NV>), g, bl, 0, st, g_out, ldg, n, D, colptr, csc_eid, csc_dst, score, (const float*)gpre, gx, ldgx, g_b); }
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("asap_attend_bwd");
}
