// Kernel-plan assembly of a block-diagonal batch (graph.assemble_plan): every table of the batch's plan is a concatenation of the stored
// graphs' pieces with per-piece offsets (and, for node / edge ids, a small table lookup) - ~90 framework launches per new batch as tensor
// operations, ONE launch here.  Contract: include/wsi_hgnn.h (wsi_plan_assemble).
#include "common.h"

namespace wsi {

constexpr int PLAN_ROW = 10;              // int64 words per segment descriptor
constexpr int PLAN_BLOCK = 1024;          // elements per workgroup

// desc (device, int64): per segment  [out, in1, in2, tab_off, key, add, stride, n, mode, block_start], then the lookup tables the tab_off's point into
__global__ __launch_bounds__(256) void plan_assemble_kernel(const int64_t* __restrict__ desc, int nsegs) {
    const int b = blockIdx.x;
    int lo = 0, hi = nsegs - 1;           // last segment whose block_start <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(int64_t)mid * PLAN_ROW + 9] <= b) lo = mid; else hi = mid - 1;
    }
    const int64_t* d = desc + (int64_t)lo * PLAN_ROW;
    const int64_t n = d[7];
    const int mode = (int)d[8];
    const int64_t i0 = ((int64_t)b - d[9]) * PLAN_BLOCK;
    const int64_t* in1 = reinterpret_cast<const int64_t*>(d[1]);
    const int64_t* in2 = reinterpret_cast<const int64_t*>(d[2]);
    const int64_t* tab = d[3] >= 0 ? desc + d[3] + d[4] : nullptr;
    const int64_t add = d[5], stride = d[6];
#pragma unroll
    for (int r = 0; r < PLAN_BLOCK / 256; ++r) {
        const int64_t i = i0 + r * 256 + threadIdx.x;
        if (i >= n) break;
        if (mode == 0) {
            int64_t v = add + i * stride;
            if (in1) v += in1[i];
            if (tab) v += tab[in2 ? in2[i] : 0];
            reinterpret_cast<int32_t*>(d[0])[i] = (int32_t)v;
        } else if (mode == 1) {
            reinterpret_cast<float*>(d[0])[i] = reinterpret_cast<const float*>(d[1])[i];
        } else {
            reinterpret_cast<float*>(d[0])[i] = __int_as_float((int)add);
        }
    }
}

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_plan_assemble(const int64_t* desc, int32_t nsegs, int32_t total_blocks, void* stream) {
    if (nsegs < 0 || total_blocks < 0) { set_error("plan_assemble: bad argument"); return WSI_EINVAL; }
    if (nsegs == 0 || total_blocks == 0) return WSI_OK;
    if (!desc) { set_error("plan_assemble: null pointer"); return WSI_EINVAL; }
    hipLaunchKernelGGL(plan_assemble_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, desc, (int)nsegs);
    return check_launch("plan_assemble");
}
