// What does a CU's vector-memory path deliver for the attention kernels' access shapes?  Every wave issues `ITERS` rounds of 8 independent
// loads; a load instruction covers 64 / LPS segments of LPS lanes x VW floats (contiguous inside a segment), segments chosen by a hash
// (random) or consecutively (stream) out of a table of ROWS rows with a pitch of PITCH floats that is private to the XCD (block b -> XCD b % 8).
// Reports bytes per clock per CU at the measured time (2.1 GHz nominal used only for the B/clk column) and TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };
__device__ __forceinline__ float sum(float v) { return v; }
__device__ __forceinline__ float sum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float sum(float4 v) { return v.x + v.y + v.z + v.w; }

template <int LPS, int VW, bool RANDOM>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ tab, int rows_mask, int pitch, int iters, float* __restrict__ out) {
    using V = typename Vec<VW>::T;
    const int lane = threadIdx.x & 63;
    const int seg = lane / LPS, sl = lane % LPS;
    const int xcd = blockIdx.x & 7;
    const float* base = tab + (size_t)xcd * (size_t)(rows_mask + 1) * pitch + sl * VW;
    const uint32_t wid = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        V v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t key = (wid + it * 8 + j) * (64 / LPS) + seg;
            const uint32_t r = (RANDOM ? mix(key) : key) & (uint32_t)rows_mask;
            v[j] = *reinterpret_cast<const V*>(base + (size_t)r * pitch);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += sum(v[j]);
    }
    if (acc == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LPS, int VW, bool RANDOM>
void run(const char* name, const float* tab, int rows, int pitch, float* out) {
    const int blocks = 256 * 8, iters = 200;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_kernel<LPS, VW, RANDOM>), dim3(blocks), dim3(256), 0, 0, tab, rows - 1, pitch, iters, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((gather_kernel<LPS, VW, RANDOM>), dim3(blocks), dim3(256), 0, 0, tab, rows - 1, pitch, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 256 * iters * 8 * VW * 4;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-44s rows/XCD %7d pitch %5d B  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  %6.1f G segments/s\n", name, rows, pitch * 4, ms * 1e3, tbs,
           tbs * 1e12 / 256 / 2.1e9, bytes / (LPS * VW * 4) / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t floats = (size_t)8 * 262144 * 64;       // 8 XCD-private tables of up to 262144 rows x 256 B = 64 MB each
    float *tab, *out;
    hipMalloc(&tab, floats * 4);
    hipMemset(tab, 0, floats * 4);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int rows : {8192, 262144}) {       // 2 MB per XCD (L2-resident) / 64 MB per XCD (512 MB total: HBM / Infinity Cache)
        printf("== %d rows of 256 B per XCD\n", rows);
        run<64, 4, false>("stream: 1 KB per instruction (x4)", tab, rows / 4, 256, out);
        run<64, 1, false>("stream: 256 B per instruction (x1)", tab, rows, 64, out);
        run<64, 4, true>("gather 1 KB segments (64 lanes x 16 B)", tab, rows / 4, 256, out);
        run<16, 4, true>("gather 256 B segments (16 lanes x 16 B)", tab, rows, 64, out);
        run<8, 4, true>("gather 128 B segments (8 lanes x 16 B)", tab, rows * 2, 32, out);
        run<32, 2, true>("gather 256 B segments (32 lanes x 8 B)", tab, rows, 64, out);
        run<64, 1, true>("gather 256 B segments (64 lanes x 4 B)", tab, rows, 64, out);
        run<32, 1, true>("gather 128 B segments (32 lanes x 4 B)", tab, rows * 2, 32, out);
        run<16, 1, true>("gather 64 B segments (16 lanes x 4 B)", tab, rows * 4, 16, out);
        run<4, 4, true>("gather 64 B segments (4 lanes x 16 B)", tab, rows * 4, 16, out);
    }
    return 0;
}
