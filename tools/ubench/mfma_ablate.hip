// Ablation: what takes a 24-MFMA "stage" loop from the pure-MFMA rate to the GEMM's?  Variants add, one at a time, the other
// ingredients of the GEMM main loop: a workgroup barrier per stage, the 12 fragment reads (ds_read_b128) per stage, the 6
// LDS-DMA loads per stage.  Random bf16 operands.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// MODE bits: 1 = barrier per stage, 2 = fragment reads from LDS per stage, 4 = LDS-DMA loads per stage (6 x 1 KB per wave),
//            8 = the DMA source is the GEMM's real pattern: 96-byte segments of 128 + 128 rows at a 3072-byte pitch, streaming through a
//                246 MB A operand and a 4.7 MB B operand (instead of one small L2-resident block read in contiguous 1-KB pieces)
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint4* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[49152];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gt = blockIdx.x * 256 + tid;
    // fill LDS with random data
    for (int i = tid; i < 49152 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[(gt * 7 + i) & 0xfffff];
    __syncthreads();
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[p][i] = __builtin_bit_cast(bf16x8, src[(gt * 12 + p * 4 + i * 2) & 0xfffff]);
            b[p][i] = __builtin_bit_cast(bf16x8, src[(gt * 12 + p * 4 + i * 2 + 1) & 0xfffff]);
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
    const int foff = ((wave >> 1) * 64 + (lane & 31)) * 96 + (lane >> 5) * 16;
    const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(src) + (size_t)(blockIdx.x % 4096) * 49152 + wave * 6144 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 4) {
            unsigned char* nb = smem + ((it & 1) ? 0 : 24576);
            if (MODE & 8) {
                // tile (tm, tn) of a [80000 x 512] x [1536 x 512] product: A rows tm*128.., B rows tn*128.., stage it%32
                const int tile = blockIdx.x, tm = (tile / 12) % 625, tn = tile % 12;
                const unsigned char* base = reinterpret_cast<const unsigned char*>(src);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = (wave * 3 + q) * 64 + lane, row = c / 6, part = c - row * 6;
                    const unsigned char* pa = base + (size_t)(tm * 128 + row) * 3072 + (it & 31) * 96 + part * 16;
                    const unsigned char* pb = base + (size_t)80000 * 3072 + (size_t)(tn * 128 + row) * 3072 + (it & 31) * 96 + part * 16;
                    __builtin_amdgcn_global_load_lds((glb_void_t*)pa, (lds_void_t*)(nb + wave * 3072 + q * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((glb_void_t*)pb, (lds_void_t*)(nb + 12288 + wave * 3072 + q * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc + q * 1024 + (size_t)(it & 63) * 96), (lds_void_t*)(nb + wave * 6144 + q * 1024), 16, 0, 0);
            }
        }
        if (MODE & 2) {
            const unsigned char* cb = smem + ((it & 1) ? 24576 : 0);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const bf16x8*>(cb + foff + p * 32 + i * 32 * 96);
                    b[p][i] = *reinterpret_cast<const bf16x8*>(cb + 12288 + foff + p * 32 + i * 32 * 96);
                }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
        if (MODE & 1) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[gt] = s;
}

template <int MODE, int WPS>
double run(const uint4* d, float* o, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * WPS;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16 / ms / 1e9;
}

int main() {
    const size_t n = ((size_t)81536 * 3072 + 4096) / 16 + (1 << 20);      // A planes [80000 x 512] + B planes [1536 x 512]
    std::vector<uint32_t> h(n * 4);
    for (size_t i = 0; i < h.size(); ++i) {
        uint32_t lo = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
        uint32_t hi = ((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f);
        h[i] = lo | (hi << 16);
    }
    uint4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 256 * 256 * 8 * 4 * 4);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    const int it = 3000;
    printf("random operands, TFLOP/s bf16, 2 / 3 waves per SIMD\n");
    printf("mfma only                 : %.0f / %.0f\n", run<0, 2>(d, o, it), run<0, 3>(d, o, it));
    printf("+ barrier per stage       : %.0f / %.0f\n", run<1, 2>(d, o, it), run<1, 3>(d, o, it));
    printf("+ 12 fragment reads       : %.0f / %.0f\n", run<2, 2>(d, o, it), run<2, 3>(d, o, it));
    printf("+ reads + barrier         : %.0f / %.0f\n", run<3, 2>(d, o, it), run<3, 3>(d, o, it));
    printf("+ LDS-DMA loads + barrier : %.0f / %.0f\n", run<5, 2>(d, o, it), run<5, 3>(d, o, it));
    printf("all (GEMM main loop)      : %.0f / %.0f\n", run<7, 2>(d, o, it), run<7, 3>(d, o, it));
    printf("all, real address pattern : %.0f / %.0f\n", run<15, 2>(d, o, it), run<15, 3>(d, o, it));
    return 0;
}
