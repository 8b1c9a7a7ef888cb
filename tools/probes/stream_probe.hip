// What can the chip deliver for the decode kernel's ACCESS PATTERN with no arithmetic at all?
// One wavefront per (sequence, kv head): per 16-token tile it reads 16 rows x 256 B of K and of V (row stride
// 2 KiB = 8 kv heads x 256 B, pages of 16 rows scattered by a random block table), 1 KiB per wave instruction,
// TILES_IN_FLIGHT tiles ahead, non-temporal buffer loads -- the loader of paged_decode_kernel -- and only XORs
// the data into a register.  B = 256 sequences x 8 kv heads x 4096 tokens = 4.29 GB, the C2a workload.
//   hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int P = 3;
template <bool NT>
__global__ void __launch_bounds__(64) probe(const char *k, const char *v, const int *bt, int pages_per_seq, int hk_n, unsigned *sink) {
    const int lane = threadIdx.x, sub = lane >> 4, dc = lane & 15;
    const int hk = blockIdx.x % hk_n, b = blockIdx.x / hk_n;
    const int *row = bt + (int64_t)b * pages_per_seq;
    const int64_t row_bytes = (int64_t)hk_n * 256, page_bytes = 16 * row_bytes;
    const uint32_t lane_off = (uint32_t)(sub * row_bytes + dc * 16);
    constexpr int AUX = NT ? 2 : 0;
    u32x4 kb[P][4], vb[P][4];
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int s, int tile) {
        const int pid = row[tile];
        const char *kt = k + (int64_t)pid * page_bytes + hk * 256, *vt = v + (int64_t)pid * page_bytes + hk * 256;
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(kt), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vt), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) kb[s][r] = __builtin_amdgcn_raw_buffer_load_b128(kr, lane_off, (int)(r * 4 * row_bytes), AUX);
#pragma unroll
        for (int r = 0; r < 4; ++r) vb[s][r] = __builtin_amdgcn_raw_buffer_load_b128(vr, lane_off, (int)(r * 4 * row_bytes), AUX);
    };
#pragma unroll
    for (int s = 0; s < P; ++s) issue(s, s);
    int t = 0;
    for (; t + 2 * P <= pages_per_seq; t += P) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc ^= kb[s][r] ^ vb[s][r];
            issue(s, t + s + P);
        }
    }
    for (; t < pages_per_seq; ++t) {   // drain (the last tiles were loaded above; re-reads of the final tiles are negligible)
#pragma unroll
        for (int s = 0; s < P; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc ^= kb[s][r] ^ vb[s][r];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;   // keeps the loads alive
}
int main() {
    const int B = 256, hk = 8, S = 4096, pps = S / 16;
    const int64_t n_pages = (int64_t)B * pps * 9 / 8, page_bytes = 16 * hk * 256;
    char *k, *v;
    int *bt;
    unsigned *sink;
    hipMalloc(&k, n_pages * page_bytes); hipMalloc(&v, n_pages * page_bytes); hipMalloc(&bt, B * pps * 4); hipMalloc(&sink, 4);
    hipMemset(k, 1, n_pages * page_bytes); hipMemset(v, 2, n_pages * page_bytes);
    std::vector<int> perm(n_pages);
    for (int64_t i = 0; i < n_pages; ++i) perm[i] = (int)i;
    srand(1);
    for (int64_t i = n_pages - 1; i > 0; --i) { const int64_t j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
    hipMemcpy(bt, perm.data(), B * pps * 4, hipMemcpyHostToDevice);
    hipEvent_t a, e;
    hipEventCreate(&a); hipEventCreate(&e);
    const double bytes = 2.0 * B * S * hk * 256;
    for (int nt = 1; nt >= 0; --nt) {
        for (int it = 0; it < 3; ++it)
            if (nt) hipLaunchKernelGGL(probe<true>, dim3(B * hk), dim3(64), 0, 0, k, v, bt, pps, hk, sink);
            else hipLaunchKernelGGL(probe<false>, dim3(B * hk), dim3(64), 0, 0, k, v, bt, pps, hk, sink);
        hipEventRecord(a, 0);
        const int iters = 20;
        for (int it = 0; it < iters; ++it)
            if (nt) hipLaunchKernelGGL(probe<true>, dim3(B * hk), dim3(64), 0, 0, k, v, bt, pps, hk, sink);
            else hipLaunchKernelGGL(probe<false>, dim3(B * hk), dim3(64), 0, 0, k, v, bt, pps, hk, sink);
        hipEventRecord(e, 0);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, a, e);
        ms /= iters;
        printf("decode access pattern, no arithmetic, %s loads: %.4f ms per pass, %.0f GB/s (%.1f %% of 8 TB/s)\n", nt ? "non-temporal" : "default",
               ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12 * 100);
    }
    return 0;
}
