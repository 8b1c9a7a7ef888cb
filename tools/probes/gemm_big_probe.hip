// Where do the cycles of the 65..256-row GEMM tile go?  Stand-alone test bed for linear_big_kernel (csrc/linear_decode.hip):
// y[M x N] = x[M x K] . W[N x K]^T in bf16, 128 weight rows x 256 batch rows per workgroup (4 wavefronts, one per SIMD),
// K in chunks of 64 inputs, operands brought into LDS by the global->LDS DMA (no staging registers, no ds_write), ring of
// NSLOT chunks.  VAR bits switch single pipeline stages off to see what bounds the loop:
//   1 = no MFMA   2 = no DMA inside the loop   4 = no LDS fragment reads   8 = W only through the DMA (x tile loaded once)
//   hipcc -O3 --offload-arch=gfx950 gemm_big_probe.hip -o gemm_big_probe ; ./gemm_big_probe [N K M]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16_saddr(uint64_t base_uniform, uint32_t voff, uint32_t lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <int BM, int NSLOT, int VAR>
__global__ void __launch_bounds__(256, 1) gemm_big(const uint16_t *x, const uint16_t *w, uint16_t *y, int M, int N, int K) {
    constexpr int WT = 128 * 128, XT = BM * 128, SLOT = WT + XT;
    constexpr int NB = BM / 64;                      // B fragments per wavefront
    constexpr int XP = BM / 32;                      // x DMA pieces per wavefront and chunk
    constexpr int PIECES = 4 + XP;                   // DMA instructions per wavefront and chunk
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * 128;
    const int chunks = K >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // DMA pieces: 1 KiB = 8 rows x 128 B of the LDS image [row][slot], slot = chunk ^ ((row >> 1) & 7): the lane fills
    // (row, slot) = (8 u + (lane >> 3), lane & 7) from the de-swizzled source chunk
    uint32_t woff[4], xoff[XP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        woff[i] = (uint32_t)(row * K * 2 + (((lane & 7) ^ ((row >> 1) & 7)) & 7) * 16);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int row = 8 * (XP * wave + i) + (lane >> 3);
        xoff[i] = (uint32_t)(min(row, M - 1) * K * 2 + (((lane & 7) ^ ((row >> 1) & 7)) & 7) * 16);
    }
    const uint64_t wb = uniform64((uint64_t)(w + (int64_t)n0 * K)), xb = uniform64((uint64_t)x);
    auto issue = [&](int chunk, int slot) {
        const uint64_t wbc = wb + (uint64_t)chunk * 128, xbc = xb + (uint64_t)chunk * 128;
        const uint32_t dst = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_saddr(wbc, woff[i], dst + (4 * wave + i) * 1024);
        if (!(VAR & 8) || chunk == 0) {
#pragma unroll
            for (int i = 0; i < XP; ++i) glds16_saddr(xbc, xoff[i], dst + WT + (XP * wave + i) * 1024);
        }
    };
    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    int a_off[2], b_off[NB], a_sw[2], b_sw[NB];
#pragma unroll
    for (int a = 0; a < 2; ++a) { const int r = 64 * wn + 32 * a + l32; a_off[a] = r * 128; a_sw[a] = (r >> 1) & 7; }
#pragma unroll
    for (int b = 0; b < NB; ++b) { const int r = (BM / 2) * wm + 32 * b + l32; b_off[b] = WT + r * 128; b_sw[b] = (r >> 1) & 7; }
    struct Frag { u32x4 a[2], b[NB]; };
    auto read_frags = [&](int slot, int s4, Frag &f) {
        if (VAR & 4) return;
        const char *base = smem + ((VAR & 8) ? 0 : slot * SLOT);
        const char *wbase = smem + slot * SLOT;
#pragma unroll
        for (int a = 0; a < 2; ++a) f.a[a] = *reinterpret_cast<const u32x4 *>(wbase + a_off[a] + (((2 * s4 + kh) ^ a_sw[a]) & 7) * 16);
#pragma unroll
        for (int b = 0; b < NB; ++b) f.b[b] = *reinterpret_cast<const u32x4 *>(base + b_off[b] + (((2 * s4 + kh) ^ b_sw[b]) & 7) * 16);
    };
    auto mfmas = [&](const Frag &f) {
        if (VAR & 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a][b][0] += __uint_as_float(f.a[a][0] ^ f.b[b][1]);
            return;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[a]), __builtin_bit_cast(bf16x8, f.b[b]), acc[a][b], 0, 0, 0);
    };
    Frag fA, fB;
    if (VAR & 4) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fA.a[a] = fB.a[a] = u32x4{(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
        for (int b = 0; b < NB; ++b) fA.b[b] = fB.b[b] = u32x4{(unsigned)lane, 5u, 6u, 7u};
    }
#define SB __builtin_amdgcn_sched_barrier(0)
    // prologue: chunks 0 .. NSLOT-2 on their way
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) if (s < chunks) issue(s, s);
    if (chunks >= NSLOT - 1) vm_wait<PIECES * (NSLOT - 2)>(); else vm_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(0, 0, fA);
    int slot = 0;
    for (int c = 0; c < chunks; ++c) {
        const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        const bool steady = c + NSLOT - 1 < chunks;
        if (steady && (!(VAR & 2))) issue(c + NSLOT - 1, pslot);
        SB;
        read_frags(slot, 1, fB); SB; mfmas(fA); SB;
        read_frags(slot, 2, fA); SB; mfmas(fB); SB;
        read_frags(slot, 3, fB); SB; mfmas(fA); SB;
        if (steady) vm_wait<PIECES * (NSLOT - 2)>(); else vm_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < chunks) read_frags(nslot, 0, fA);
        SB; mfmas(fB); SB;
        slot = nslot;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int brow = (BM / 2) * wm + 32 * b + l32;
        if (brow >= M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 64 * wn + 32 * a + 8 * j + 4 * kh;
                uint2 o;
                o.x = f2bf(acc[a][b][4 * j]) | ((uint32_t)f2bf(acc[a][b][4 * j + 1]) << 16);
                o.y = f2bf(acc[a][b][4 * j + 2]) | ((uint32_t)f2bf(acc[a][b][4 * j + 3]) << 16);
                *reinterpret_cast<uint2 *>(y + (int64_t)brow * N + n) = o;
            }
    }
}


// v2: split rings (W: NWS slots of 16 KiB, x: NXS slots of BM x 128 B), the DMA issue specialised by operand (the first half
// of the wavefronts brings W, the second half x: vmcnt retires in order, so a wave that waits for its x piece of the NEXT chunk
// would also wait for every W piece it issued earlier -- with the operands on different waves the W stream can run NWS-1
// chunks ahead), DMA issues / fragment reads interleaved with the MFMAs, and WAVES = 4 (one per SIMD) or 8 (two per SIMD: a
// wave stuck in the VMEM issue queue no longer idles its matrix core).
template <int BM, int WAVES, int NWS, int NXS, int VAR>
__global__ void __launch_bounds__(64 * WAVES, 1) gemm_big2(const uint16_t *x, const uint16_t *w, uint16_t *y, int M, int N, int K) {
    constexpr int WT = 128 * 128, XT = BM * 128;
    constexpr int R = WAVES / 2;                                    // wavefronts per DMA role = wave rows of the compute grid
    constexpr int BMW = BM / R, NB = BMW / 32;                      // batch rows / B fragments per wavefront
    constexpr int PW = 16 / R, PX = BM / 8 / R, PMAX = PW > PX ? PW : PX;   // DMA pieces per wavefront and chunk, by role
    constexpr int DW = NWS - 1, DX = NXS - 1;                       // chunks ahead
    constexpr int NR = 2 + NB, NM = 2 * NB;                         // fragment reads / MFMAs per k16 step
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [W ring][x ring]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * 128;
    const int chunks = K >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const bool wrole = wave < R;
    const int np = wrole ? PW : PX, rw = wrole ? wave : wave - R;

    uint32_t voff[PMAX];
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int row = 8 * (np * rw + i) + (lane >> 3);
        const int src = wrole ? row : min(row, M - 1);
        voff[i] = (uint32_t)(src * K * 2 + (((lane & 7) ^ ((row >> 1) & 7)) & 7) * 16);
    }
    const uint64_t gbase = wrole ? uniform64((uint64_t)(w + (int64_t)n0 * K)) : uniform64((uint64_t)x);
    const uint32_t ring = (wrole ? lds0 : lds0 + NWS * WT) + np * rw * 1024;
    const uint32_t slotb = wrole ? WT : XT;
    const int nslots = wrole ? NWS : NXS, dist = wrole ? DW : DX;
    auto piece = [&](auto I, int chunk) {
        constexpr int i = decltype(I)::value;
        if (i < np) glds16_saddr(gbase + (uint64_t)chunk * 128, voff[i], ring + (chunk % nslots) * slotb + i * 1024);
    };
    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    int f_off[NR], f_sw[NR];                                        // fragment reads: 0,1 = A (W rows), 2.. = B (x rows)
#pragma unroll
    for (int a = 0; a < 2; ++a) { const int r = 64 * wn + 32 * a + l32; f_off[a] = r * 128; f_sw[a] = (r >> 1) & 7; }
#pragma unroll
    for (int b = 0; b < NB; ++b) { const int r = BMW * wm + 32 * b + l32; f_off[2 + b] = NWS * WT + r * 128; f_sw[2 + b] = (r >> 1) & 7; }
    struct Frag { u32x4 f[NR]; };
    Frag fA, fB;
    if (VAR & 4) {
#pragma unroll
        for (int i = 0; i < NR; ++i) fA.f[i] = fB.f[i] = u32x4{(unsigned)lane, 1u + i, 2u, 3u};
    }
    // one k16 step: the MFMAs on `cur`; behind MFMA k: fragment reads 2k, 2k+1 of step `s4` of chunk `rc` into `nxt` (if rd) and DMA piece lo + k of chunk `dc`
    auto group = [&](const Frag &cur, Frag &nxt, bool rd, int rc, int s4, auto LO, auto HI, bool dma, int dc) {
        constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
        const int wsl = (rc % NWS) * WT, xsl = (rc % NXS) * XT;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            const int a = k & 1, b = k >> 1;
            if (VAR & 1) acc[a][b][0] += __uint_as_float(cur.f[a][0] ^ cur.f[2 + b][1]);
            else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.f[a]), __builtin_bit_cast(bf16x8, cur.f[2 + b]), acc[a][b], 0, 0, 0);
            if (!(VAR & 4) && rd) {
#pragma unroll
                for (int i = 2 * k; i < 2 * k + 2 && i < NR; ++i)
                    nxt.f[i] = *reinterpret_cast<const u32x4 *>(smem + (i < 2 ? wsl : xsl) + f_off[i] + (((2 * s4 + kh) ^ f_sw[i]) & 7) * 16);
            }
            SB;
            if (dma && !(VAR & 2)) {
                if (k == 0) { if constexpr (lo + 0 < hi) piece(std::integral_constant<int, lo + 0>{}, dc); }
                if (k == 1) { if constexpr (lo + 1 < hi) piece(std::integral_constant<int, lo + 1>{}, dc); }
                if (k == 2) { if constexpr (lo + 2 < hi) piece(std::integral_constant<int, lo + 2>{}, dc); }
                if (k == 3) { if constexpr (lo + 3 < hi) piece(std::integral_constant<int, lo + 3>{}, dc); }
                if (k == 4) { if constexpr (lo + 4 < hi) piece(std::integral_constant<int, lo + 4>{}, dc); }
                if (k == 5) { if constexpr (lo + 5 < hi) piece(std::integral_constant<int, lo + 5>{}, dc); }
                if (k == NM - 1) {    // what did not fit behind an MFMA
                    if constexpr (lo + NM < hi) piece(std::integral_constant<int, lo + NM>{}, dc);
                    if constexpr (lo + NM + 1 < hi) piece(std::integral_constant<int, lo + NM + 1>{}, dc);
                }
                SB;
            }
        }
    };
    static_assert(PMAX <= 16 && (PMAX + 2) / 3 <= NM + 2, "piece schedule");
    // prologue
    for (int c = 0; c < dist && c < chunks; ++c) {
        piece(std::integral_constant<int, 0>{}, c); piece(std::integral_constant<int, 1>{}, c); piece(std::integral_constant<int, 2>{}, c); piece(std::integral_constant<int, 3>{}, c);
        if constexpr (PMAX > 4) {
            piece(std::integral_constant<int, 4>{}, c); piece(std::integral_constant<int, 5>{}, c); piece(std::integral_constant<int, 6>{}, c); piece(std::integral_constant<int, 7>{}, c);
        }
        if constexpr (PMAX > 8) {
            piece(std::integral_constant<int, 8>{}, c); piece(std::integral_constant<int, 9>{}, c); piece(std::integral_constant<int, 10>{}, c); piece(std::integral_constant<int, 11>{}, c);
            piece(std::integral_constant<int, 12>{}, c); piece(std::integral_constant<int, 13>{}, c); piece(std::integral_constant<int, 14>{}, c); piece(std::integral_constant<int, 15>{}, c);
        }
    }
    vm_wait<0>();
    __syncthreads();
    if (!(VAR & 4)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) fA.f[i] = *reinterpret_cast<const u32x4 *>(smem + f_off[i] + (((0 + kh) ^ f_sw[i]) & 7) * 16);
    }
    typedef std::integral_constant<int, 0> I0;
    for (int c = 0; c < chunks; ++c) {
        const int dc = c + dist;
        const bool dma = dc < chunks;
        constexpr int g1 = (PMAX + 2) / 3, g2 = (2 * PMAX + 2) / 3;
        group(fA, fB, true, c, 1, I0{}, std::integral_constant<int, g1>{}, dma, dc);
        group(fB, fA, true, c, 2, std::integral_constant<int, g1>{}, std::integral_constant<int, g2>{}, dma, dc);
        group(fA, fB, true, c, 3, std::integral_constant<int, g2>{}, std::integral_constant<int, PMAX>{}, dma, dc);
        if (!dma) vm_wait<0>();
        else if (wrole) vm_wait<PW * (DW - 1)>();
        else vm_wait<PX * (DX - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        group(fB, fA, c + 1 < chunks, c + 1, 0, I0{}, I0{}, false, 0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int brow = BMW * wm + 32 * b + l32;
        if (brow >= M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 64 * wn + 32 * a + 8 * j + 4 * kh;
                uint2 o;
                o.x = f2bf(acc[a][b][4 * j]) | ((uint32_t)f2bf(acc[a][b][4 * j + 1]) << 16);
                o.y = f2bf(acc[a][b][4 * j + 2]) | ((uint32_t)f2bf(acc[a][b][4 * j + 3]) << 16);
                *reinterpret_cast<uint2 *>(y + (int64_t)brow * N + n) = o;
            }
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
static float h_bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <typename KF>
static void run_k(const char *name, KF kern, int LDS, int threads, int BM, int s1, int s2, int VAR, const uint16_t *x, uint16_t *const *w, int nw, uint16_t *y, int M, int N, int K, const std::vector<uint16_t> &hx, const std::vector<uint16_t> &hw) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(N / 128), dim3(threads), LDS, 0, x, w[i % nw], y, M, N, K);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(N / 128), dim3(threads), LDS, 0, x, w[i % nw], y, M, N, K);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    double err = 0;
    if (VAR == 0) {
        CK(hipMemset(y, 0, (size_t)M * N * 2));
        hipLaunchKernelGGL(kern, dim3(N / 128), dim3(threads), LDS, 0, x, w[0], y, M, N, K);
        std::vector<uint16_t> hy((size_t)M * N);
        CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
        for (int t = 0; t < 4000; ++t) {
            const int m = (t * 37 + t / 7) % M, n = (int)(((int64_t)t * 7919 + 13) % N);
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)h_bf2f(hx[(size_t)m * K + k]) * h_bf2f(hw[(size_t)n * K + k]);
            const double d = fabs(s - h_bf2f(hy[(size_t)m * N + n])) / (fabs(s) + 1.0);
            if (d > err) err = d;
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 2.0 * ((double)N * K + (double)M * K + (double)M * N);
    printf("%-30s BM=%d rings=%d/%d var=%2d  %8.1f us  %7.1f TFLOP/s  %7.1f GB/s  max rel err %.2e\n", name, BM, s1, s2, VAR,
           ms * 1e3, flops / ms * 1e-9, bytes / ms * 1e-6, err);
}
template <int BM, int WAVES, int NWS, int NXS, int VAR>
static void run2(const char *name, const uint16_t *x, uint16_t *const *w, int nw, uint16_t *y, int M, int N, int K, const std::vector<uint16_t> &hx, const std::vector<uint16_t> &hw) {
    run_k(name, &gemm_big2<BM, WAVES, NWS, NXS, VAR>, NWS * 16384 + NXS * BM * 128, 64 * WAVES, BM, NWS, NXS, VAR, x, w, nw, y, M, N, K, hx, hw);
}
template <int BM, int NSLOT, int VAR>
static void run(const char *name, const uint16_t *x, uint16_t *const *w, int nw, uint16_t *y, int M, int N, int K, const std::vector<uint16_t> &hx, const std::vector<uint16_t> &hw) {
    constexpr int LDS = NSLOT * (128 * 128 + BM * 128);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_big<BM, NSLOT, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_big<BM, NSLOT, VAR>), dim3(N / 128), dim3(256), LDS, 0, x, w[i % nw], y, M, N, K);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_big<BM, NSLOT, VAR>), dim3(N / 128), dim3(256), LDS, 0, x, w[i % nw], y, M, N, K);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    double err = 0;
    if (VAR == 0) {   // check some outputs of a launch on w[0]
        hipLaunchKernelGGL((gemm_big<BM, NSLOT, VAR>), dim3(N / 128), dim3(256), LDS, 0, x, w[0], y, M, N, K);
        std::vector<uint16_t> hy((size_t)M * N);
        CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
        for (int t = 0; t < 4000; ++t) {
            const int m = (t * 37 + t / 7) % M, n = (int)(((int64_t)t * 7919 + 13) % N);
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)h_bf2f(hx[(size_t)m * K + k]) * h_bf2f(hw[(size_t)n * K + k]);
            const double d = fabs(s - h_bf2f(hy[(size_t)m * N + n])) / (fabs(s) + 1.0);
            if (d > err) err = d;
        }
    }
    const double flops = 2.0 * M * N * K, bytes = 2.0 * ((double)N * K + (double)M * K + (double)M * N);
    printf("%-34s BM=%d slots=%d var=%2d  %8.1f us  %7.1f TFLOP/s  %7.1f GB/s  cycles/chunk=%6.0f  max rel err %.2e\n", name, BM, NSLOT, VAR,
           ms * 1e3, flops / ms * 1e-9, bytes / ms * 1e-6, ms * 1e-3 * 2.4e9 / (K / 64) / ((N / 128 + 255) / 256), err);
}

// W stream alone, the access pattern of the GEMM tile: a workgroup walks its 128 weight rows along K, RB bytes of every row
// per step, DEPTH steps in flight (global->LDS DMA into a ring, nothing reads it).  What does the memory system give?
template <int RB, int DEPTH>
__global__ void __launch_bounds__(256, 1) wstream(const uint16_t *w, int N, int K) {
    constexpr int STEPB = 128 * RB, PPW = STEPB / 1024 / 4, RPP = 1024 / RB;   // pieces per wave and step, rows per piece
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint64_t wb = uniform64((uint64_t)(w + (int64_t)blockIdx.x * 128 * K));
    uint32_t voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) voff[i] = (uint32_t)(((wave * PPW + i) * RPP + lane / (RB / 16)) * K * 2 + (lane % (RB / 16)) * 16);
    const int steps = K * 2 / RB;
    auto issue = [&](int s) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16_saddr(wb + (uint64_t)s * RB, voff[i], lds0 + (s % (DEPTH + 1)) * STEPB + (wave * PPW + i) * 1024);
    };
    for (int s = 0; s < DEPTH && s < steps; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        if (s + DEPTH < steps) { issue(s + DEPTH); vm_wait<PPW * DEPTH>(); } else vm_wait<0>();
        __syncthreads();
    }
}
template <int RB, int DEPTH>
static void run_ws(uint16_t *const *w, int nw, int N, int K) {
    constexpr int LDS = (DEPTH + 1) * 128 * RB;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wstream<RB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wstream<RB, DEPTH>), dim3(N / 128), dim3(256), LDS, 0, w[i % nw], N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((wstream<RB, DEPTH>), dim3(N / 128), dim3(256), LDS, 0, w[i % nw], N, K);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 30;
    printf("W stream: %4d B of each row per step, %d steps (%3d KiB) in flight per workgroup: %7.1f us  %7.1f GB/s\n", RB, DEPTH, DEPTH * 128 * RB / 1024, ms * 1e3, (double)N * K * 2 / ms * 1e-6);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 28672, K = argc > 2 ? atoi(argv[2]) : 4096, M = argc > 3 ? atoi(argv[3]) : 256;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 8388608.f - 1.f); };
    for (auto &v : hx) v = h_f2bf(rnd());
    for (auto &v : hw) v = h_f2bf(rnd() * 0.05f);
    uint16_t *x, *y, *w[3];
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&y, (size_t)M * N * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&w[i], hw.size() * 2)); CK(hipMemcpy(w[i], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
    printf("N=%d K=%d M=%d: weights %.1f MB, HBM floor at 6.5 TB/s %.1f us, MFMA floor %.1f us\n", N, K, M, N * (double)K * 2e-6, N * (double)K * 2 / 6.5e6,
           2.0 * M * N * K / 2.5e9 * 256.0 / ((N / 128) < 256 ? (N / 128) : 256));
    run_ws<128, 2>(w, 3, N, K); run_ws<128, 4>(w, 3, N, K); run_ws<128, 8>(w, 3, N, K);
    run_ws<256, 1>(w, 3, N, K); run_ws<256, 2>(w, 3, N, K); run_ws<256, 4>(w, 3, N, K);
    run_ws<512, 1>(w, 3, N, K); run_ws<512, 2>(w, 3, N, K);
    run_ws<1024, 1>(w, 3, N, K);
    run<256, 3, 0>("full", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 4, 4, 3, 0>("v2 4 waves full", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 8, 4, 3, 0>("v2 8 waves full", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 8, 6, 2, 0>("v2 8 waves full", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 8, 4, 3, 1>("v2 8 waves no MFMA", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 8, 4, 3, 2>("v2 8 waves no DMA in loop", x, w, 3, y, M, N, K, hx, hw);
    run2<256, 8, 4, 3, 5>("v2 8 waves DMA only", x, w, 3, y, M, N, K, hx, hw);
    run2<128, 8, 6, 4, 0>("v2 8 waves full", x, w, 3, y, M < 128 ? M : 128, N, K, hx, hw);
    run2<128, 4, 6, 4, 0>("v2 4 waves full", x, w, 3, y, M < 128 ? M : 128, N, K, hx, hw);
    run<256, 3, 1>("no MFMA", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 2>("no DMA in loop", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 4>("no LDS reads", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 6>("MFMA only", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 5>("DMA only", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 8>("W DMA only, x resident", x, w, 3, y, M, N, K, hx, hw);
    run<256, 3, 13>("W DMA only, nothing else", x, w, 3, y, M, N, K, hx, hw);
    run<128, 4, 0>("full", x, w, 3, y, M < 128 ? M : 128, N, K, hx, hw);
    return 0;
}
