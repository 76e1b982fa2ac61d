// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate with NO memory traffic in the loop, for the accumulator pattern of
// the bf16x6 GEMMs (4 accumulators round-robin, 24 MFMAs per "stage"), operands random / zero.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint4* __restrict__ src, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[p][i] = __builtin_bit_cast(bf16x8, src[(tid * 12 + p * 4 + i * 2) & 0xfffff]);
            b[p][i] = __builtin_bit_cast(bf16x8, src[(tid * 12 + p * 4 + i * 2 + 1) & 0xfffff]);
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[tid] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<uint32_t> h(n * 4);
    uint4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 256 * 256 * 8 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 3; ++fill) {
        for (size_t i = 0; i < h.size(); ++i) {
            if (fill == 0) h[i] = 0;
            else if (fill == 1) {   // random bf16 in [-1,1): sign, exponent 0x70..0x7e, random mantissa
                uint32_t lo = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
                uint32_t hi = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
                h[i] = lo | (hi << 16);
            } else {                // like the split planes: plane magnitudes 1, 2^-8, 2^-16 do not matter for toggling; positive only
                uint32_t lo = ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
                uint32_t hi = ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
                h[i] = lo | (hi << 16);
            }
        }
        hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 3; ++wps) {
            const int blocks = 256 * wps;   // one block of 4 waves per CU per wave-per-SIMD
            const int iters = 4000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (wps == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else if (wps == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
            printf("fill=%s waves/SIMD=%d: %.1f TFLOP/s bf16 (%.3f ms)\n", fill == 0 ? "zero" : fill == 1 ? "random" : "random+", wps, fl / ms / 1e9, ms);
        }
    }
    return 0;
}
