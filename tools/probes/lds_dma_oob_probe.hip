// What does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA) leave in LDS on gfx950: zeros, or the old bytes?
// And does the soffset operand take part in the range check?   hipcc --offload-arch=gfx950 lds_dma_oob_probe.hip -o lds_dma_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint32_t *src, uint32_t *out, uint32_t nrec, uint32_t soff) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    for (uint32_t i = threadIdx.x; i < 1024; i += 64) ((uint32_t *)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t base = (uint64_t)(uintptr_t)src;
    const uint32_t d0 = __builtin_amdgcn_readfirstlane((uint32_t)base), d1 = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint32_t voff = threadIdx.x * 16;
    const uint32_t nr = __builtin_amdgcn_readfirstlane(nrec), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 s40, %1\n\ts_mov_b32 s41, %2\n\ts_mov_b32 s42, %3\n\ts_mov_b32 s43, 0x00020000\n\ts_mov_b32 s44, %4\n\t"
                 "s_mov_b32 m0, 0\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, s[40:43], s44 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(d0), "s"(d1), "s"(nr), "s"(so) : "s40", "s41", "s42", "s43", "s44", "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) out[threadIdx.x * 4 + k] = ((uint32_t *)smem)[threadIdx.x * 4 + k];
}
int main() {
    uint32_t h[1024], *d, *o, ho[256];
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000000u + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    struct { uint32_t nrec, soff; } cases[] = {{1024, 0}, {512, 0}, {0, 0}, {1024, 512}, {1536, 512}, {520, 0}};
    for (auto c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, o, c.nrec, c.soff);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        int first_bad = -1, zeros = 0, old = 0, data = 0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t v = ho[l * 4];
            if (v == 0) zeros++; else if (v == 0xdeadbeefu) old++; else data++;
            if (first_bad < 0 && v != h[(c.soff / 4) + l * 4]) first_bad = l;
        }
        printf("num_records %4u soffset %3u: lanes with data %2d, zeros %2d, old LDS bytes %2d; first lane without its data: %d (lane 33 words: %08x %08x %08x %08x)\n",
               c.nrec, c.soff, data, zeros, old, first_bad, ho[33 * 4], ho[33 * 4 + 1], ho[33 * 4 + 2], ho[33 * 4 + 3]);
    }
    return 0;
}
