// Probe: how does the MI355X dispatcher place workgroups on CUs?  Each workgroup records its XCC id,
// HW_ID (SE / CU), start and end time, and busy-waits for dur[blockIdx.x] ticks of wall_clock64.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_probe.hip -o /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>
struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };
template <int LDSB>
__global__ void probe(const int* dur, Rec* out) {
    extern __shared__ char hog[];   // dynamic LDS: the launch decides how many workgroups fit on a CU
    hog[threadIdx.x] = 1;
    unsigned long long t0 = (unsigned long long)wall_clock64();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long until = t0 + (unsigned long long)dur[blockIdx.x];
    while ((unsigned long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x] = Rec{xcc & 0xf, hwid, t0, (unsigned long long)wall_clock64()}; }
    if (hog[(threadIdx.x + 1) % 64] == 77) out[0].xcc = 99;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;   // 0: 1 WG/CU (LDS hog, 512 thr), 1: 64-thread WGs
    std::vector<int> dur(n);
    for (int i = 0; i < n; ++i) dur[i] = 2000 + 1000 * (((i / 8) * 5) % 8);   // 100 MHz clock: 20..90 us, varies with the index inside the XCD
    int* d_dur; Rec* d_out;
    hipMalloc(&d_dur, n * sizeof(int)); hipMalloc(&d_out, n * sizeof(Rec));
    hipMemcpy(d_dur, dur.data(), n * sizeof(int), hipMemcpyHostToDevice);
    if (mode == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        hipLaunchKernelGGL((probe<0>), dim3(n), dim3(512), 98304, 0, d_dur, d_out);   // 96 KiB: one workgroup per CU
    } else hipLaunchKernelGGL((probe<0>), dim3(n), dim3(64), 64, 0, d_dur, d_out);
    hipDeviceSynchronize();
    std::vector<Rec> r(n);
    hipMemcpy(r.data(), d_out, n * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto& x : r) tmin = std::min(tmin, x.t0);
    // CU identity: (xcc, se, sh, cu) from HW_ID: cu_id bits [11:8], sh_id [12], se_id [15:13] on gfx9
    std::map<unsigned, std::vector<int>> by_cu;
    for (int i = 0; i < n; ++i) {
        unsigned cu = (r[i].hwid >> 8) & 0xf, sh = (r[i].hwid >> 12) & 1, se = (r[i].hwid >> 13) & 7;
        unsigned key = (r[i].xcc << 12) | (se << 8) | (sh << 4) | cu;
        by_cu[key].push_back(i);
        if (i < 0) printf("wg %4d xcc %u se %u sh %u cu %2u start %6llu end %6llu\n", i, r[i].xcc, se, sh, cu, r[i].t0 - tmin, r[i].t1 - tmin);
    }
    printf("distinct CUs used: %zu\n", by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 10) break;
        printf("cu key %05x:", kv.first);
        for (int i : kv.second) printf(" %d(%llu-%llu)", i, r[i].t0 - tmin, r[i].t1 - tmin);
        printf("\n");
    }
    unsigned long long tmax = 0; for (auto& x : r) tmax = std::max(tmax, x.t1);
    printf("total ticks %llu\n", tmax - tmin);
    return 0;
}
