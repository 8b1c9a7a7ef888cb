// What bounds a projection at 17..64 rows (the tensor-parallel rank of configs[3]: 16-117 MB of weights per product)?
// Stand-alone test bed for the LDS-DMA tile kernel behind atoma_linear_decode at these sizes (csrc/linear_tile.hip):
//   y[M x N] = x[M x K] . W[N x K]^T, bf16, M <= 64; a workgroup owns NW weight rows x all 64 batch rows over a K range;
//   both operands arrive in LDS by the global->LDS DMA in 4 rows x 256 B pieces (full 128-byte lines, image [row][slot ^ (row & 15)]),
//   ring of NSLOT chunks of 128 inputs, one barrier per chunk, v_mfma_f32_16x16x32 on fragments read with ds_read_b128.
// VAR bits switch stages off:  1 = no LDS reads / MFMAs   2 = no W DMA   4 = no x DMA
// Weights rotate through several copies so that no launch finds its matrix in the 256 MB cache.
//   hipcc -O3 --offload-arch=gfx950 gemm64_probe.hip -o gemm64_probe ; ./gemm64_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16_saddr(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void glds16_saddr_nt(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__host__ __device__ __forceinline__ float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ __forceinline__ uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <int NW, int WAVES, int NSLOT, int VAR, bool NT>
__global__ void __launch_bounds__(64 * WAVES, 1) gemm64(const uint16_t *x, const uint16_t *w, float *part, uint16_t *y, int M, int N, int K, int cps) {
    constexpr int WT = NW * 256, XT = 64 * 256, SLOT = WT + XT;
    constexpr int PW = NW / 4, P = PW + 16, PPW = P / WAVES;     // DMA pieces (1 KiB = 4 rows x 256 B) per chunk / per wavefront
    static_assert(P % WAVES == 0, "pieces per wavefront");
    constexpr int G = NW / 16, GPW = G * 4 / WAVES;               // 16-row groups of W; groups per wavefront (one batch tile per wavefront)
    static_assert(GPW >= 1 && (G * 4) % WAVES == 0, "units per wavefront");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = lane >> 4, col = lane & 15;
    const int tiles = N / NW;
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    const int n0 = tile * NW;
    const int chunks_all = K >> 7;
    const int c0 = split * cps, c1 = min(c0 + cps, chunks_all), chunks = c1 - c0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // piece q of a chunk: q < PW: W rows 4q..4q+3, else x rows 4(q - PW)..; lane fills (row, slot) = (4q' + (lane >> 4), lane & 15)
    // from the de-swizzled source chunk (lane & 15) ^ (row & 15)
    uint32_t voff[PPW];
    bool isw[PPW];
    uint32_t dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        isw[i] = q < PW;
        const int row = 4 * (isw[i] ? q : q - PW) + (lane >> 4);
        const int src = isw[i] ? row : min(row, M - 1);
        voff[i] = (uint32_t)(src * K * 2 + ((lane & 15) ^ (row & 15)) * 16);
        dst[i] = (isw[i] ? 0 : WT) + 4 * (isw[i] ? q : q - PW) * 256;
    }
    const uint64_t wb = uniform64((uint64_t)(w + (int64_t)n0 * K)) + (uint64_t)c0 * 256, xb = uniform64((uint64_t)x) + (uint64_t)c0 * 256;
    auto issue = [&](int chunk, int slot) {
        const uint32_t sl = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (isw[i]) { if (!(VAR & 2)) { if (NT) glds16_saddr_nt(wb + (uint64_t)chunk * 256, voff[i], sl + dst[i]); else glds16_saddr(wb + (uint64_t)chunk * 256, voff[i], sl + dst[i]); } }
            else if (!(VAR & 4)) glds16_saddr(xb + (uint64_t)chunk * 256, voff[i], sl + dst[i]);
        }
    };
    // compute mapping: wavefront -> batch tile ct and GPW row groups
    const int ct = WAVES == 4 ? wave : wave >> 1, g0 = WAVES == 4 ? 0 : (wave & 1) * GPW;
    f32x4 acc[GPW];
#pragma unroll
    for (int a = 0; a < GPW; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    int a_off[GPW];
#pragma unroll
    for (int a = 0; a < GPW; ++a) a_off[a] = (16 * (g0 + a) + col) * 256;
    const int b_off = WT + (16 * ct + col) * 256;
    auto compute = [&](int slot) {
        if (VAR & 1) return;
        const char *base = smem + slot * SLOT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sw = ((4 * s + grp) ^ col) * 16;
            const u32x4 b = *reinterpret_cast<const u32x4 *>(base + b_off + sw);
#pragma unroll
            for (int a = 0; a < GPW; ++a) {
                const u32x4 av = *reinterpret_cast<const u32x4 *>(base + a_off[a] + sw);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b), acc[a], 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) if (s < chunks) issue(s, s);
    int slot = 0;
    for (int c = 0; c < chunks; ++c) {
        if (c + NSLOT - 2 < chunks) vm_wait<PPW * (NSLOT - 2)>(); else vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        if (c + NSLOT - 1 < chunks) issue(c + NSLOT - 1, pslot);
        compute(slot);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    const int brow = 16 * ct + col;
    if (brow >= M) return;
#pragma unroll
    for (int a = 0; a < GPW; ++a) {
        const int n = n0 + 16 * (g0 + a) + 4 * grp;
        if (part) {
            *reinterpret_cast<float4 *>(part + ((int64_t)split * M + brow) * N + n) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        } else {
            uint2 o;
            o.x = f2bf(acc[a][0]) | ((uint32_t)f2bf(acc[a][1]) << 16);
            o.y = f2bf(acc[a][2]) | ((uint32_t)f2bf(acc[a][3]) << 16);
            *reinterpret_cast<uint2 *>(y + (int64_t)brow * N + n) = o;
        }
    }
}

#define CHECK(e) do { hipError_t err_ = (e); if (err_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

struct Cfg { int nw, waves, nslot, var, nt; };

template <int NW, int WAVES, int NSLOT, int VAR, bool NT>
static void launch(const uint16_t *x, const uint16_t *w, float *part, uint16_t *y, int M, int N, int K, int splits) {
    const int chunks = K / 128, cps = (chunks + splits - 1) / splits;
    const size_t lds = (size_t)NSLOT * (NW * 256 + 64 * 256);
    static bool once = false;
    if (!once) { CHECK(hipFuncSetAttribute((const void *)gemm64<NW, WAVES, NSLOT, VAR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    hipLaunchKernelGGL((gemm64<NW, WAVES, NSLOT, VAR, NT>), dim3((N / NW) * splits), dim3(64 * WAVES), lds, 0, x, w, splits > 1 ? part : nullptr, y, M, N, K, cps);
}

typedef void (*LaunchFn)(const uint16_t *, const uint16_t *, float *, uint16_t *, int, int, int, int);
struct Variant { const char *name; LaunchFn fn; int nw; };
#define V(NW, WAVES, NSLOT, VAR, NT) {#NW "rows " #WAVES "w ring" #NSLOT " var" #VAR " nt" #NT, launch<NW, WAVES, NSLOT, VAR, NT>, NW}
static Variant variants[] = {
    V(64, 8, 4, 0, true), V(64, 8, 4, 0, false), V(64, 8, 3, 0, true), V(64, 4, 4, 0, true), V(64, 8, 5, 0, true),
    V(32, 8, 4, 0, true), V(128, 8, 3, 0, true), V(128, 8, 2, 0, true),
    V(64, 8, 4, 1, true), V(64, 8, 4, 5, true), V(64, 8, 4, 3, true), V(64, 8, 4, 4, true),
    V(128, 8, 3, 5, true), V(32, 8, 4, 5, true),
};

int main(int argc, char **argv) {
    const int M = 64;
    // correctness on a small problem, every full variant
    {
        const int N = 512, K = 1024;
        std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
        srand(1);
        for (auto &v : hx) v = f2bf((float)(rand() % 2001 - 1000) / 1000.f);
        for (auto &v : hw) v = f2bf((float)(rand() % 2001 - 1000) / 1000.f);
        uint16_t *dx, *dw, *dy; float *dp;
        CHECK(hipMalloc(&dx, hx.size() * 2)); CHECK(hipMalloc(&dw, hw.size() * 2)); CHECK(hipMalloc(&dy, (size_t)M * N * 2)); CHECK(hipMalloc(&dp, (size_t)4 * M * N * 4));
        CHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        std::vector<float> ref((size_t)M * N);
        for (int b = 0; b < M; ++b) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)bf2f(hx[(size_t)b * K + k]) * bf2f(hw[(size_t)n * K + k]); ref[(size_t)b * N + n] = (float)s; }
        for (auto &v : variants) {
            if (strstr(v.name, "var0") == nullptr) continue;
            for (int splits = 1; splits <= 2; ++splits) {
                CHECK(hipMemset(dy, 0, (size_t)M * N * 2)); CHECK(hipMemset(dp, 0, (size_t)4 * M * N * 4));
                v.fn(dx, dw, dp, dy, M, N, K, splits);
                CHECK(hipDeviceSynchronize());
                double worst = 0;
                if (splits == 1) {
                    std::vector<uint16_t> hy((size_t)M * N);
                    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < hy.size(); ++i) worst = std::max(worst, (double)fabsf(bf2f(hy[i]) - ref[i]) / (fabs(ref[i]) * 0.0079 + 1e-2));
                } else {
                    std::vector<float> hp((size_t)2 * M * N);
                    CHECK(hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < (size_t)M * N; ++i) worst = std::max(worst, (double)fabsf(hp[i] + hp[(size_t)M * N + i] - ref[i]) / 1e-3);
                }
                printf("check %-28s splits %d: %s (worst %.3f of the bound)\n", v.name, splits, worst <= 1.0 ? "ok" : "WRONG", worst);
            }
        }
        (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(dp);
    }
    struct Shape { const char *name; int N, K; } shapes[] = {{"gate_up", 7168, 8192}, {"down", 8192, 3584}, {"o", 8192, 1024}, {"qkv", 1280, 8192}};
    for (auto &sh : shapes) {
        const int N = sh.N, K = sh.K;
        const size_t wbytes = (size_t)N * K * 2;
        const int nbuf = std::max(2, (int)(600000000 / wbytes));
        std::vector<uint16_t *> dw(nbuf);
        std::vector<uint16_t> h((size_t)1 << 22);
        for (auto &v : h) v = f2bf((float)(rand() % 2001 - 1000) / 1000.f);
        for (auto &p : dw) { CHECK(hipMalloc(&p, wbytes)); for (size_t o = 0; o < wbytes; o += h.size() * 2) CHECK(hipMemcpy((char *)p + o, h.data(), std::min(h.size() * 2, wbytes - o), hipMemcpyHostToDevice)); }
        uint16_t *dx, *dy; float *dp;
        CHECK(hipMalloc(&dx, (size_t)M * K * 2)); CHECK(hipMalloc(&dy, (size_t)M * N * 2)); CHECK(hipMalloc(&dp, (size_t)16 * M * N * 4));
        CHECK(hipMemcpy(dx, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int rounds = 5, iters = 10;
        for (auto &v : variants) {
            if (N % v.nw) continue;
            for (int splits : {1, 2, 4, 8, 16}) {
                const int wgs = N / v.nw * splits;
                if (wgs < 100 || wgs > 640 || K / 128 / splits < 4) continue;
                std::vector<float> t;
                for (int r = 0; r < rounds; ++r) {
                    v.fn(dx, dw[0], dp, dy, M, N, K, splits);
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < iters; ++i) v.fn(dx, dw[i % nbuf], dp, dy, M, N, K, splits);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms / iters * 1e3f);
                }
                std::sort(t.begin(), t.end());
                printf("%-8s [%5d x %5d] %-28s splits %2d wgs %3d : median %6.1f us  min %6.1f us  %5.0f GB/s of W\n", sh.name, N, K, v.name, splits, wgs, t[rounds / 2], t[0], wbytes / t[rounds / 2] / 1e3);
                fflush(stdout);
            }
        }
        for (auto &p : dw) (void)hipFree(p);
        (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dp);
    }
    return 0;
}
