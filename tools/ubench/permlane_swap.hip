// What v_permlane16_swap_b32 does on gfx950 (the builtin returns the two registers after the swap): expectation
// r0[lane] = (lane & 16) ? b[lane - 16] : a[lane],  r1[lane] = (lane & 16) ? b[lane] : a[lane + 16]
// i.e. the upper 16-lane rows of the first operand trade places with the lower rows of the second.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[threadIdx.x + 64] = r[1];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    unsigned h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; ++i) {
        const unsigned e0 = (i & 16) ? 100 + i - 16 : i, e1 = (i & 16) ? 100 + i : i + 16;
        if (h[i] != e0 || h[i + 64] != e1) ok = 0;
        printf("%2d: %3u %3u\n", i, h[i], h[i + 64]);
    }
    printf(ok ? "AS EXPECTED\n" : "DIFFERENT\n");
}
