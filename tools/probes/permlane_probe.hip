// v_permlane32_swap semantics on gfx950: with both operands = x, lane l ends up holding x[l % 32] and x[32 + l % 32].
// hipcc --offload-arch=gfx950 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const unsigned x = 1000 + threadIdx.x;
    const u2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const unsigned a = h[l], b = h[64 + l];
        const unsigned lo = 1000 + l % 32, hi = 1000 + 32 + l % 32;
        ok &= (a == lo && b == hi) || (a == hi && b == lo);
    }
    printf("lane 0: %u %u, lane 40: %u %u -> every lane sees both members of its pair: %s\n", h[0], h[64], h[40], h[104], ok ? "yes" : "NO");
    return !ok;
}
