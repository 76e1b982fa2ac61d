// What buffer_load_dwordx4 ... lds does with its three offsets (gfx950): where the data comes from and where it lands.
// One wave; source = 64 KB of dwords holding their own index; the LDS (16 KB, filled with 0xdeadbeef) is dumped afterwards.
// usage: hipcc --offload-arch=gfx950 -O3 lds_dma_semantics.hip -o lds_dma_semantics && ./lds_dma_semantics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int IMM>
__global__ void k(const unsigned* src, unsigned* dump, int soff, int lds_off, int nrec) {
    __shared__ __attribute__((aligned(1024))) unsigned sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = 0xdeadbeefu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)((char*)sm + lds_off), 16, threadIdx.x * 16, soff, IMM, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) dump[i] = sm[i];
}
int main() {
    const int N = 16384;
    std::vector<unsigned> h(N), d(4096);
    for (int i = 0; i < N; ++i) h[i] = i;
    unsigned *src, *dump;
    hipMalloc(&src, N * 4); hipMalloc(&dump, 4096 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    auto run = [&](const char* name, int imm, int soff, int lds_off, int nrec) {
        if (imm == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, src, dump, soff, lds_off, nrec);
        else hipLaunchKernelGGL(k<1024>, dim3(1), dim3(64), 0, 0, src, dump, soff, lds_off, nrec);
        hipMemcpy(d.data(), dump, 4096 * 4, hipMemcpyDeviceToHost);
        int first = -1, last = -1;
        for (int i = 0; i < 4096; ++i) if (d[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
        printf("%-46s imm=%4d soff=%5d lds_off=%5d nrec=%6d -> LDS dwords [%d..%d] written, first value %u (= source byte %u), lane1 value %u\n",
               name, imm, soff, lds_off, nrec, first, last, first >= 0 ? d[first] : 0, first >= 0 ? d[first] * 4 : 0, first >= 0 ? d[first + 4] : 0);
    };
    run("plain", 0, 0, 0, N * 4);
    run("lds pointer + 2048", 0, 0, 2048, N * 4);
    run("imm 1024", 1024, 0, 0, N * 4);
    run("imm 1024, lds pointer + 1024", 1024, 0, 1024, N * 4);
    run("soffset 4096", 0, 4096, 0, N * 4);
    run("soffset 4096, lds pointer + 2048", 0, 4096, 2048, N * 4);
    run("soffset 4096, num_records 2048 (voffset in range)", 0, 4096, 0, 2048);
    run("soffset 0, num_records 512 (half the lanes out)", 0, 0, 0, 512);
    return 0;
}
