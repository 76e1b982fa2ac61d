// Multi-tensor Adam step for gfx950: ONE launch over all parameters of the model (the optimizer step the reference's trainer takes after
// loss.backward(): torch.optim.Adam(lr, weight_decay) of parser.py:33-38, trainer/train_gnn.py:72).  Contract: include/wsi_hgnn.h.
//
// HBM-bound streaming: per element 16 bytes read (p, g, m, v), 12 written; 16-byte lane accesses where the four pointers allow it.  The
// tensor table travels in the kernel arguments (no upload); a workgroup finds its tensor by a scan of the block prefix (<= 64 entries).
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int ADAM_MAX = 128;           // tensors per launch (HEATNet4 with 3 node types has 75)
constexpr int ADAM_BLOCK_ELEMS = 4096;  // elements per workgroup (256 threads x 4 vectors of 4)

struct AdamTable {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int32_t block_start[ADAM_MAX + 1];  // prefix of workgroups per tensor
    int64_t n[ADAM_MAX];
    int32_t count;
    float step_size, beta1, beta2, omb1, omb2, eps, weight_decay, bc2_sqrt;   // step_size = lr / (1 - beta1^t), omb = 1 - beta (taken in double), bc2_sqrt = sqrt(1 - beta2^t)
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamTable& T) {
    g = fmaf(T.weight_decay, p, g);                       // torch.optim.Adam: L2 penalty added to the gradient (not AdamW)
    m = fmaf(T.omb1, g - m, m);                           // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(T.beta2, v, T.omb2 * g * g);                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / T.bc2_sqrt + T.eps;
    p -= T.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTable T) {
    int t = 0;
    const int b = (int)blockIdx.x;
    while (t + 1 < T.count && T.block_start[t + 1] <= b) ++t;          // (block-uniform; <= 64 steps)
    const int64_t base = (int64_t)(b - T.block_start[t]) * ADAM_BLOCK_ELEMS;
    const int64_t n = T.n[t];
    float* __restrict__ p = T.p[t];
    const float* __restrict__ g = T.g[t];
    float* __restrict__ m = T.m[t];
    float* __restrict__ v = T.v[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int q = 0; q < ADAM_BLOCK_ELEMS / 1024; ++q) {
        const int64_t i = base + q * 1024 + (int64_t)threadIdx.x * 4;
        if (i >= n) break;
        if (vec && i + 3 < n) {
            float4 pv = *reinterpret_cast<const float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<const float4*>(m + i);
            float4 vv = *reinterpret_cast<const float4*>(v + i);
            adam_one(pv.x, gv.x, mv.x, vv.x, T); adam_one(pv.y, gv.y, mv.y, vv.y, T);
            adam_one(pv.z, gv.z, mv.z, vv.z, T); adam_one(pv.w, gv.w, mv.w, vv.w, T);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k) {
                float pk = p[i + k], mk = m[i + k], vk = v[i + k];
                adam_one(pk, g[i + k], mk, vk, T);
                p[i + k] = pk; m[i + k] = mk; v[i + k] = vk;
            }
        }
    }
}

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_adam_step(const wsi_adam_tensor_t* tensors, int32_t count, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int64_t step, void* stream) {
    if (count < 0 || step <= 0 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) { set_error("adam_step: bad argument"); return WSI_EINVAL; }
    if (count == 0) return WSI_OK;
    if (!tensors) { set_error("adam_step: null tensor table"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    // the scalar factors in double, as torch's host code takes them (1 - 0.999f in float is off by 1.3e-5 relative)
    const float step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    for (int32_t first = 0; first < count;) {          // (`first` advances by what the table CONSUMED: empty tensors are skipped without taking a slot)
        AdamTable T;
        T.count = 0; T.step_size = step_size; T.beta1 = (float)beta1; T.beta2 = (float)beta2; T.omb1 = (float)(1.0 - beta1); T.omb2 = (float)(1.0 - beta2);
        T.eps = (float)eps; T.weight_decay = (float)weight_decay; T.bc2_sqrt = bc2_sqrt;
        int64_t blocks = 0;
        int32_t i = first;
        for (; i < count && T.count < ADAM_MAX; ++i) {
            const wsi_adam_tensor_t& a = tensors[i];
            if (a.n < 0 || (a.n > 0 && (!a.p || !a.g || !a.m || !a.v))) { set_error("adam_step: tensor %d: null pointer or negative size", i); return WSI_EINVAL; }
            if (a.n == 0) continue;
            const int k = T.count++;
            T.p[k] = a.p; T.g[k] = a.g; T.m[k] = a.m; T.v[k] = a.v; T.n[k] = a.n;
            T.block_start[k] = (int32_t)blocks;
            blocks += (a.n + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS;
            if (blocks > INT32_MAX) { set_error("adam_step: too many elements in one launch"); return WSI_EINVAL; }
        }
        T.block_start[T.count] = (int32_t)blocks;
        if (T.count) hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, T);
        first = i;
    }
    return check_launch("adam_step");
}
