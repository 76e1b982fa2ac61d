// The trainer's classification loss in one launch.  Contract: include/wsi_hgnn.h.
//
// (Round 4 also built HEATNet4's whole prediction head - models/HEATNet4.py:216-245 behind the readout - as fused kernels, three launches each
// way instead of one per Linear and per gradient.  Measured on one MI355X in alternating runs it was SLOWER than the chain of skinny GEMM
// launches it replaced (6.87 vs 6.79 ms per step; 134 vs 92 us of kernel time): the chain's levels are dependent, each level is a few
// hundred independent dot products that a launch of its own spreads over the whole chip, and a kernel that keeps several levels inside one
// workgroup walks the weights at the latency of one CU.  Removed again; profiles/README.md keeps the numbers.)
#include "common.h"
#include <math.h>

namespace wsi {

// mean cross entropy of `logits` [B, C] against integer labels, forward and the gradient factor in ONE launch:
//   loss = mean_{b valid} ( logsumexp(logits[b, :]) - logits[b, y_b] ) ;  dlogits[b, c] = (softmax(logits[b, :])[c] - [c == y_b]) / #valid
// (torch.nn.CrossEntropyLoss() with its defaults - parser.py:182-183 - as trainer/train_gnn.py:67 applies it).  A label equal to torch's default
// ignore_index (-100) is IGNORED as torch ignores it: zero gradient row, not counted in the mean (no valid row at all: loss = 0 / 0 = NaN, as torch).
// Any other label outside [0, C) - torch's device assert - sets *bad (if given), zeroes its gradient row and turns the loss into NaN: the failure is
// visible in the result itself, without a host read.  One workgroup; B * C <= 65536.
constexpr int64_t CE_IGNORE_INDEX = -100;
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B, int C,
                                                            float* __restrict__ loss, float* __restrict__ dlogits, int* __restrict__ bad) {
    __shared__ float part[256];
    __shared__ int cnt[256];
    __shared__ int any_bad;
    if (threadIdx.x == 0) any_bad = 0;
    int valid = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int64_t y = labels[b];
        valid += (y >= 0 && y < C) ? 1 : 0;
    }
    cnt[threadIdx.x] = valid;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) cnt[threadIdx.x] += cnt[threadIdx.x + o];
        __syncthreads();
    }
    const int nvalid = cnt[0];
    const float invB = nvalid > 0 ? 1.f / (float)nvalid : 0.f;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* row = logits + (int64_t)b * C;
        const int64_t y = labels[b];
        if (y < 0 || y >= C) {
            if (y != CE_IGNORE_INDEX) { any_bad = 1; if (bad) *bad = 1; }
            for (int c = 0; c < C; ++c) dlogits[(int64_t)b * C + c] = 0.f;
            continue;
        }
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(row[c] - mx);
        acc += (logf(den) + mx) - row[y];
        const float inv = 1.f / den;
        for (int c = 0; c < C; ++c) dlogits[(int64_t)b * C + c] = (expf(row[c] - mx) * inv - (c == y ? 1.f : 0.f)) * invB;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = any_bad ? NAN : part[0] / (float)nvalid;
}

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_cross_entropy(const float* logits, const int64_t* labels, int32_t B, int32_t C, float* loss, float* dlogits, int32_t* bad_label,
                                 void* stream) {
    if (B <= 0 || C <= 0 || (int64_t)B * C > 65536) { set_error("cross_entropy: unsupported shape %d x %d", B, C); return WSI_ENOSYS; }
    if (!logits || !labels || !loss || !dlogits) { set_error("cross_entropy: null pointer"); return WSI_EINVAL; }
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, (int)B, (int)C, loss, dlogits, (int*)bad_label);
    return check_launch("cross_entropy");
}
