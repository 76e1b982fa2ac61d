// What ds_read_b64_tr_b16 returns, established on the GPU (gfx950): the LDS holds element index e at 16-bit slot e; every lane supplies its own
// 8-byte-aligned address and receives four 16-bit values.  Printed: for every lane, which LDS slots its four result elements came from, for (a) the
// lane-linear address pattern addr = 8 * lane and (b) the pattern the weight-gradient kernel uses (a [k][m] image with a row pitch: lane s of a
// 16-lane group addresses row k0 + (s >> 2), columns 4 (s & 3) .. + 3 and is expected to receive column s, rows k0 .. k0 + 3).
//   hipcc --offload-arch=gfx950 -O2 tr_b16_semantics.hip -o tr_b16_semantics && ./tr_b16_semantics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 f16x4;
typedef __attribute__((address_space(3))) f16x4 lds_f16x4;

__global__ void probe(const int* __restrict__ addr_bytes, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    const f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)((__attribute__((address_space(3))) char*)lds + a));
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(bits >> (16 * j));
}

int main() {
    int h_addr[64];
    uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pattern = 0; pattern < 2; ++pattern) {
        const int pitch = 160;        // 16-bit elements per image row in pattern (b)
        for (int l = 0; l < 64; ++l) {
            if (pattern == 0) h_addr[l] = 8 * l;
            else {
                const int g = l >> 4, s = l & 15;
                const int k = 8 * (g >> 1) + (s >> 2), col = 16 * (g & 1) + 4 * (s & 3);
                h_addr[l] = (k * pitch + col) * 2;
            }
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pattern);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr %5d ->", l, h_addr[l]);
            for (int j = 0; j < 4; ++j) {
                const int e = h_out[l * 4 + j];
                if (pattern == 0) printf(" %4d", e);
                else {
                    printf(" (k %2d, m %2d)", e / pitch, e % pitch);
                    const int g = l >> 4, s = l & 15;
                    if (e / pitch != 8 * (g >> 1) + j || e % pitch != 16 * (g & 1) + s) ++bad;
                }
            }
            printf("\n");
        }
        if (pattern == 1) printf("pattern 1: %d elements differ from the expected (column = lane, rows k0..k0+3) result\n", bad);
    }
    return 0;
}
