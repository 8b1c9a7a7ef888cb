// What does the LOAD PATTERN of the weight-streaming kernels cost?  lm_head of Llama-3.1-8B ([128256 x 4096] bf16, 1.05 GB),
// one wavefront per 16 consecutive rows (= one contiguous 128 KiB block), 16 KiB in flight per wavefront, non-temporal
// buffer loads, no arithmetic (XOR).  Patterns, per load instruction of 1 KiB:
//   0  "mfma":   16 rows x 64 B   (lane = 16.grp + row: the A-operand layout of v_mfma_f32_16x16x32, linear_decode.hip)
//   1  "row256":  4 rows x 256 B
//   2  "row1k":   1 row  x 1 KiB
//   3  "linear": the 128 KiB block front to back (1 KiB per instruction, rows one after the other)
//   hipcc -O3 --offload-arch=gfx950 weight_stream_probe.hip -o weight_stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 128256, K = 4096, ROWB = K * 2;
template <int PATTERN>
__global__ void __launch_bounds__(64) probe(const char *w, unsigned *sink) {
    const int lane = threadIdx.x;
    const char *base = w + (int64_t)blockIdx.x * 16 * ROWB;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 16 * ROWB, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    constexpr int STAGE = 8;                 // 8 KiB per stage, two stages in flight
    u32x4 buf[2][STAGE];
    // byte offset of this lane's 16 bytes for the i-th KiB of the block (i = 0 .. 127), per pattern
    auto off = [&](int i) -> uint32_t {
        if (PATTERN == 0) { const int chunk = i >> 2, q = i & 3; return (uint32_t)((lane & 15) * ROWB + chunk * 256 + q * 64 + (lane >> 4) * 16); }
        if (PATTERN == 1) { const int chunk = i >> 2, rg = i & 3; return (uint32_t)((rg * 4 + (lane >> 4)) * ROWB + chunk * 256 + (lane & 15) * 16); }
        if (PATTERN == 2) { const int seg = i >> 4, row = i & 15; return (uint32_t)(row * ROWB + seg * 1024 + lane * 16); }
        return (uint32_t)(i * 1024 + lane * 16);
    };
    auto issue = [&](int s, int st) {
#pragma unroll
        for (int j = 0; j < STAGE; ++j) buf[s][j] = __builtin_amdgcn_raw_buffer_load_b128(r, off(st * STAGE + j), 0, 2);
    };
    issue(0, 0);
    issue(1, 1);
    constexpr int NST = 128 / STAGE;
    for (int st = 0; st < NST; st += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < STAGE; ++j) acc ^= buf[s][j];
            if (st + s + 2 < NST) issue(s, st + s + 2);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
template <int PATTERN> static void run(const char *w, unsigned *sink, const char *name) {
    hipEvent_t a, e;
    hipEventCreate(&a); hipEventCreate(&e);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe<PATTERN>, dim3(N / 16), dim3(64), 0, 0, w, sink);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(probe<PATTERN>, dim3(N / 16), dim3(64), 0, 0, w, sink);
    hipEventRecord(e, 0);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, a, e);
    ms /= iters;
    const double bytes = (double)N * ROWB;
    printf("%-8s %.4f ms  %.0f GB/s (%.1f %% of 8 TB/s)\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12 * 100);
}
int main() {
    char *w;
    unsigned *sink;
    hipMalloc(&w, (size_t)N * ROWB); hipMalloc(&sink, 4);
    hipMemset(w, 1, (size_t)N * ROWB);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(w, sink, "mfma");
        run<1>(w, sink, "row256");
        run<2>(w, sink, "row1k");
        run<3>(w, sink, "linear");
    }
    return 0;
}
