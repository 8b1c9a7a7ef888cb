// Does the LDS-DMA destination (M0) reach beyond 64 KiB on gfx950?  Each test: DMA 1 KiB from global to LDS byte
// address A, read it back with ds_read, compare.  hipcc --offload-arch=gfx950 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__global__ void probe(const uint32_t *src, uint32_t *out, uint32_t addr) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    // clear the whole LDS window
    for (uint32_t i = threadIdx.x; i < 160 * 256; i += 64) ((uint32_t *)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    glds16(src + threadIdx.x * 4, __builtin_amdgcn_readfirstlane(base + addr));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // where did the data land?  scan LDS for the first word of the pattern
    const uint32_t v = *(uint32_t *)(smem + addr + threadIdx.x * 16);
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) {
        uint32_t found = 0xffffffffu;
        for (uint32_t i = 0; i < 160 * 256; ++i)
            if (((uint32_t *)smem)[i] == src[0]) { found = i * 4; break; }
        out[64] = found;
    }
}
int main() {
    uint32_t h[256], *d, *o, ho[65];
    for (int i = 0; i < 256; ++i) h[i] = 0x1000000u + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t addr : {0u, 32768u, 65536u - 1024u, 65536u, 65536u + 16384u, 98304u, 131072u, 159u * 1024u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, o, addr);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 64; ++i) ok &= ho[i] == h[i * 4];
        printf("dst %6u: read-back %s, pattern found at LDS byte %d\n", addr, ok ? "OK" : "MISMATCH", (int)ho[64]);
    }
    return 0;
}
