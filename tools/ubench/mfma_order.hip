// Does the ORDER of the 24 MFMAs of a bf16x6 stage matter for the power-limited rate?  Same 24 (A fragment, B fragment, accumulator)
// triples, different sequences: (0) production order: product-major, (i,j) = 00,01,10,11; (1) snake inside a product: 00,01,11,10
// (one operand changes per step); (2) snake + product sequence that keeps one plane fixed across product boundaries;
// (3) accumulator-major: all 6 products of one (i,j) back to back (dependent chain per accumulator).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ORDER, int WPS>
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
    constexpr int TA0[6] = {2, 0, 1, 1, 0, 0}, TB0[6] = {0, 2, 1, 0, 1, 0};
    constexpr int TA2[6] = {2, 1, 0, 0, 1, 0}, TB2[6] = {0, 0, 0, 1, 1, 2};
    constexpr int SI[4] = {0, 0, 1, 1}, SJ[4] = {0, 1, 1, 0};
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA0[t]][i], b[TB0[t]][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = ORDER == 0 ? (q >> 1) : ((t & 1) ? SI[3 - q] : SI[q]);
                    const int j = ORDER == 0 ? (q & 1) : ((t & 1) ? SJ[3 - q] : SJ[q]);
                    const int ta = ORDER == 2 ? TA2[t] : TA0[t], tb = ORDER == 2 ? TB2[t] : TB0[t];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta][i], b[tb][j], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[tid] = s;
}

template <int ORDER>
void run(const uint4* d, float* o) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    float ms2 = 0, ms3 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<ORDER, 2>), dim3(512), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms2, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL((k<ORDER, 3>), dim3(768), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms3, e0, e1);
    }
    const double f = 4.0 * iters * 24 * 2.0 * 32 * 32 * 16 / 1e9;
    printf("order %d: %.0f / %.0f TFLOP/s (2 / 3 waves per SIMD)\n", ORDER, 512 * f / ms2, 768 * f / ms3);
}

int main() {
    const int n = 1 << 20;
    std::vector<uint32_t> h(n * 4);
    for (size_t i = 0; i < h.size(); ++i) {
        uint32_t lo = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
        uint32_t hi = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
        h[i] = lo | (hi << 16);
    }
    uint4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 768 * 256 * 4);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    run<0>(d, o); run<1>(d, o); run<2>(d, o); run<3>(d, o);
    return 0;
}
