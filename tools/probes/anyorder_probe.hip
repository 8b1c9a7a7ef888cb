// Can consecutive kernels of ONE stream overlap their tail / ramp on gfx950?  hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch)
// asks the runtime for an AQL packet without the barrier bit (hip_ext.h says "not supported on GFX9xx" for the module-launch variant).
// Measured here: (1) the gap between the end of kernel A and the start of kernel B, ordered launch vs any-order launch; (2) the DISPATCH
// ORDER within the queue -- do all workgroups of A start before any workgroup of B when A needs two rounds (the property that makes a
// device-side "B waits for A's flag" hand-off deadlock-free)?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void busy(long long *start, long long *end, int us, int lds_touch) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    if (lds_touch) smem[threadIdx.x] = 1;
    while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) end[blockIdx.x] = wall_clock64();
}

int main() {
    const int lds = 150 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&busy), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 4096;
    long long *sa, *ea, *sb, *eb;
    CK(hipMalloc(&sa, N * 8)); CK(hipMalloc(&ea, N * 8)); CK(hipMalloc(&sb, N * 8)); CK(hipMalloc(&eb, N * 8));
    std::vector<long long> hsa(N), hea(N), hsb(N), heb(N);
    for (int flags = 0; flags <= 1; ++flags)
        for (int ga : {256, 512}) {
            const int gb = 256;
            double gap_sum = 0, early = 0;
            int reps = 5;
            for (int r = 0; r < reps + 1; ++r) {
                hipExtLaunchKernelGGL(busy, dim3(ga), dim3(256), lds, s, nullptr, nullptr, 0, sa, ea, 20, 1);
                hipExtLaunchKernelGGL(busy, dim3(gb), dim3(256), lds, s, nullptr, nullptr, flags, sb, eb, 20, 1);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(hsa.data(), sa, ga * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hea.data(), ea, ga * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hsb.data(), sb, gb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(heb.data(), eb, gb * 8, hipMemcpyDeviceToHost));
                if (r == 0) continue;
                const long long a_end = *std::max_element(hea.begin(), hea.begin() + ga), a_last_start = *std::max_element(hsa.begin(), hsa.begin() + ga);
                const long long b_first = *std::min_element(hsb.begin(), hsb.begin() + gb);
                gap_sum += (b_first - a_end) / 100.0;
                int n_early = 0;
                for (int i = 0; i < gb; ++i) n_early += hsb[i] < a_last_start;
                early += n_early;
            }
            printf("flags=%d  A=%d workgroups (150 KiB LDS each, 20 us), B=256: first B start - last A end = %+.2f us;  B workgroups started before A's LAST workgroup started: %.1f\n",
                   flags, ga, gap_sum / reps, early / reps);
        }
    return 0;
}
