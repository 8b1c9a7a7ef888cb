// Linear layer of a decode step at small batch: y[b][n] = sum_k x[b][k] * w[n][k], batch <= 64.
//
// Replaces candle_nn::Linear::forward on [B, hidden] activations (/root/reference/models/src/llama.rs:269-271,311,364-365:
// q/k/v/o and the MLP projections), i.e. a cuBLAS GEMM with 1..64 rows.  At these sizes the op is a stream over the
// weights -- 2.N.K bytes read once, 2.B flop per byte -- so it is laid out around the weight stream like the decode
// attention kernel, and the arithmetic rides along on the matrix cores: a 16 x 32 slab of W (16 output features,
// 32 inputs, 1 KiB = one 16-byte load per lane) is the A operand of one v_mfma_f32_16x16x32, x^T (zero-padded to 16
// columns) the B operand, and the 16 x 16 result tile holds y^T for 16 features x 16 batch rows.
//   * one wavefront per (16 output features, K split); 4 MFMAs per 128 inputs; P chunks of 4 KiB in flight per wave;
//   * W: non-temporal buffer loads (read once); x: ordinary loads (128 KiB at most, L2-resident, shared by every wave);
//   * the K range is split until ~8 wavefronts per CU exist; partial sums go to an fp32 workspace [split][B][N] and a
//     second small kernel adds them and rounds once -- fp32 accumulation throughout, one rounding to the storage dtype.
// Batches of 1..4 rows (the ones atoma_linear sends here) run linear_gemv_kernel below: loads of 256 contiguous bytes per row and
// v_dot2c instead of the MFMA operand layout, one launch; 5..16 rows run linear_wg_kernel<RELAY> (the same loads, re-laid into
// the MFMA operand order through LDS, one launch); the two-launch MFMA-layout kernel at the top serves 17..64 rows.
// Parity: unpinned (Candle / cuBLAS are not in the tree); the oracle is the f64-accumulated product rounded once,
// the kernel differs from it by at most one unit in the last place (accumulation order).
#include "common.h"
#include "norm_shared.h"
#include "linear_params.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace atoma {

// lane = 16.grp + col.  A operand: W[n0 + 16r + col][k0 + 32s + 8.grp ..+7] for the RT row tiles r of the wave;
// B operand: x[16c + col][same k] for the CT column tiles c (batch rows 16c .. 16c+15), each shared by the RT row tiles;
// result: lane holds y^T[n0 + 16r + 4.grp + i][batch 16c + col], i = 0..3.
// RT = 4 when more than a couple of batch rows are live: every wave re-reads x (from L2), and with one row tile per
// wave that is as many load instructions as the weight stream itself.  CT > 1 (batch 17..64): RT = 2.
template <typename T, int RT, int CT, int P, int CH>   // CH: 128-input chunks per pipeline stage (bytes per row visit = 256.CH)
__global__ void __launch_bounds__(64) linear_decode_kernel(const LinearParams p) {
    const int lane = threadIdx.x, grp = lane >> 4, col = lane & 15;
    const int tiles_n = p.n / (16 * RT);
    const int tile = blockIdx.x % tiles_n, split = blockIdx.x / tiles_n;
    const int n0 = tile * 16 * RT;
    const int c0 = split * p.chunks_per_split / CH, c1 = min(c0 + p.chunks_per_split / CH, (p.k >> 7) / CH);   // in stages of CH chunks

    const char *wrow = reinterpret_cast<const char *>(p.w + (int64_t)n0 * p.w_row_stride);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wrow), 0, 0x7fffffff, 0x00020000);
    uint32_t w_lane[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) w_lane[r] = (uint32_t)((16 * r + col) * p.w_row_stride * 2 + grp * 16);
    bool has_x[CT];
    const uint16_t *xrow[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        has_x[c] = 16 * c + col < p.batch;
        xrow[c] = p.x + (int64_t)(has_x[c] ? 16 * c + col : 0) * p.x_row_stride + grp * 8;
    }

    lu32x4 wb[P][RT][4 * CH], xb[P][CT][4 * CH];
    auto issue = [&](int s, int chunk) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4 * CH; ++q) wb[s][r][q] = __builtin_amdgcn_raw_buffer_load_b128(wr, w_lane[r], chunk * 256 * CH + q * 64, 2 /* nt */);
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 4 * CH; ++q)
                xb[s][c][q] = has_x[c] ? *reinterpret_cast<const lu32x4 *>(xrow[c] + chunk * 128 * CH + q * 32) : lu32x4{0, 0, 0, 0};
    };
    lf32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = lf32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4 * CH; ++q)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[r][c] = lin_mfma<T>(wb[s][r][q], xb[s][c][q], acc[r][c]);
    };
    // software pipeline: P chunks in flight; unconditional loads in the steady state keep the vmcnt waits exact
    int c = c0;
    if (c0 + 2 * P <= c1) {
#pragma unroll
        for (int s = 0; s < P; ++s) issue(s, c0 + s);
        for (; c + 2 * P <= c1; c += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                compute(s);
                issue(s, c + s + P);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (c0 + s < c1) issue(s, c0 + s);
    }
    for (; c < c1; c += P) {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (c + s < c1) {
                compute(s);
                if (c + s + P < c1) issue(s, c + s + P);
            }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if (!has_x[ct]) continue;
        const int brow = 16 * ct + col;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int n = n0 + 16 * r + 4 * grp;
            const lf32x4 a = acc[r][ct];
            if (p.partial) {
                *reinterpret_cast<float4 *>(p.partial + ((int64_t)split * p.batch + brow) * p.n + n) = make_float4(a[0], a[1], a[2], a[3]);
            } else {
                uint2 o;
                o.x = pack2<T>(a[0], a[1]);
                o.y = pack2<T>(a[2], a[3]);
                *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n) = o;
            }
        }
    }
}

// y[b][n] = round(sum over splits of the fp32 partials); 4 outputs per thread.  The epilogues keep the reference's
// rounding points: the projection's output is rounded to the storage dtype first, then the next op is applied and rounds again.
template <typename T> __device__ __forceinline__ float4 linear_sum_splits(const LinearParams &p, int64_t i, int64_t total) {
    float4 a = *reinterpret_cast<const float4 *>(p.partial + i);
    for (int s = 1; s < p.splits; ++s) {
        const float4 b = *reinterpret_cast<const float4 *>(p.partial + (int64_t)s * total + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    return make_float4(round_through<T>(a.x), round_through<T>(a.y), round_through<T>(a.z), round_through<T>(a.w));
}
template <typename T> __global__ void __launch_bounds__(256) linear_reduce_kernel(const LinearParams p) {
    const int64_t total = (int64_t)p.batch * p.n;
    const int out_n = p.epilogue == 2 ? p.n / 2 : p.n;
    const int64_t o = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;      // index into the [batch, out_n] result
    if (o >= (int64_t)p.batch * out_n) return;
    const int64_t row = o / out_n, n = o - row * out_n;
    float4 a = linear_sum_splits<T>(p, row * p.n + n, total);
    if (p.epilogue == 1) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p.aux + row * p.aux_row_stride + n);
        a.x += lo_to_f32<T>(r.x); a.y += hi_to_f32<T>(r.x); a.z += lo_to_f32<T>(r.y); a.w += hi_to_f32<T>(r.y);
    } else if (p.epilogue == 2) {
        const float4 u = linear_sum_splits<T>(p, row * p.n + out_n + n, total);
        auto silu = [](float g) { return round_through<T>(g / (1.f + __expf(-g))); };
        a.x = silu(a.x) * u.x; a.y = silu(a.y) * u.y; a.z = silu(a.z) * u.z; a.w = silu(a.w) * u.w;
    }
    uint2 w;
    w.x = pack2<T>(a.x, a.y);
    w.y = pack2<T>(a.z, a.w);
    *reinterpret_cast<uint2 *>(p.y + row * p.y_row_stride + n) = w;
}

// Batches of 1..4 rows, one launch: a workgroup of NW wavefronts owns 16 output features (PAIR: 16 gate + the 16 matching
// up features of a stacked gate / up matrix); wavefront w streams the w-th part of K, the NW accumulator tiles are added
// through LDS and the epilogue (none / + residual / silu(gate).up) runs in the workgroup -- no fp32 partials in HBM and no
// second kernel, which on the 33-120 MB projections of one layer is 3-5 us of a 11-30 us op.
// RELAY: the weights are loaded 4 rows x 256 contiguous bytes per instruction (the pattern that streams at 6.7 TB/s instead
// of 5.7, see linear_gemv_kernel) and re-laid into the MFMA A-operand order through a 4 KiB LDS tile per wavefront: lane
// 16.g + c writes its 16 bytes of row 4.rg + g to [row][chunk c ^ row], lane 16.kg + i reads [row i][chunk (4q + kg) ^ i] --
// both conflict-free; LDS serves one wavefront's accesses in order, the compiler is kept from reordering them by fences.
template <typename T, int NW, bool PAIR, int P, bool RELAY = false>
__global__ void __launch_bounds__(64 * NW) linear_wg_kernel(const LinearParams p) {
    constexpr int RT = PAIR ? 2 : 1;
    __shared__ float red[NW > 1 ? NW - 1 : 1][RT][64][4];
    __shared__ __attribute__((aligned(16))) char relay[RELAY ? NW : 1][RELAY ? RT : 1][RELAY ? 4096 : 16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), grp = lane >> 4, col = lane & 15;
    const int n0 = blockIdx.x * 16;
    const int out_n = PAIR ? p.n / 2 : p.n;
    const int chunks = p.k >> 7, per = (chunks + NW - 1) / NW;
    const int c0 = wave * per, c1 = min(c0 + per, chunks);

    const char *wrow = reinterpret_cast<const char *>(p.w + (int64_t)n0 * p.w_row_stride);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wrow), 0, 0x7fffffff, 0x00020000);
    uint32_t w_lane[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
        w_lane[r] = RELAY ? (uint32_t)(((int64_t)r * out_n + grp) * p.w_row_stride * 2 + col * 16)     // row 4.q + grp, chunk col
                          : (uint32_t)(((int64_t)r * out_n + col) * p.w_row_stride * 2 + grp * 16);
    const int64_t row_bytes = p.w_row_stride * 2;
    const bool has_x = col < p.batch;
    const uint16_t *xrow = p.x + (int64_t)(has_x ? col : 0) * p.x_row_stride + grp * 8;

    lu32x4 wb[P][RT][4], xb[P][4];
    auto issue = [&](int s, int chunk) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wb[s][r][q] = RELAY ? __builtin_amdgcn_raw_buffer_load_b128(wr, w_lane[r], (int)(q * 4 * row_bytes + chunk * 256), 2 /* nt */)
                                    : __builtin_amdgcn_raw_buffer_load_b128(wr, w_lane[r], chunk * 256 + q * 64, 2 /* nt */);
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[s][q] = has_x ? *reinterpret_cast<const lu32x4 *>(xrow + chunk * 128 + q * 32) : lu32x4{0, 0, 0, 0};
    };
    lf32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = lf32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int s) {
        if constexpr (RELAY) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                char *tile = relay[wave][r];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + grp;
                    *reinterpret_cast<lu32x4 *>(tile + row * 256 + ((col ^ row) & 15) * 16) = wb[s][r][q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const char *tile = relay[wave][r];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const lu32x4 a = *reinterpret_cast<const lu32x4 *>(tile + col * 256 + (((4 * q + grp) ^ col) & 15) * 16);
                    acc[r] = lin_mfma<T>(a, xb[s][q], acc[r]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = lin_mfma<T>(wb[s][r][q], xb[s][q], acc[r]);
        }
    };
    int c = c0;
    if (c0 + 2 * P <= c1) {
#pragma unroll
        for (int s = 0; s < P; ++s) issue(s, c0 + s);
        for (; c + 2 * P <= c1; c += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                compute(s);
                issue(s, c + s + P);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (c0 + s < c1) issue(s, c0 + s);
    }
    for (; c < c1; c += P) {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (c + s < c1) {
                compute(s);
                if (c + s + P < c1) issue(s, c + s + P);
            }
    }
    // add the NW accumulator tiles: wavefronts 1.. write, wavefront 0 sums (fixed order: deterministic)
    if constexpr (NW > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < RT; ++r) *reinterpret_cast<lf32x4 *>(red[wave - 1][r][lane]) = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const lf32x4 o = *reinterpret_cast<const lf32x4 *>(red[w][r][lane]);
                acc[r] += o;
            }
    }
    if (!has_x) return;
    // lane holds y^T[n0 + 4.grp + i][batch row col]; rounding points as in linear_reduce_kernel
    const int n = n0 + 4 * grp;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = round_through<T>(acc[0][i]);
    if (p.epilogue == 1) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p.aux + (int64_t)col * p.aux_row_stride + n);
        v[0] += lo_to_f32<T>(r.x); v[1] += hi_to_f32<T>(r.x); v[2] += lo_to_f32<T>(r.y); v[3] += hi_to_f32<T>(r.y);
    }
    if constexpr (PAIR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = round_through<T>(v[i] / (1.f + __expf(-v[i]))) * round_through<T>(acc[1][i]);
    }
    uint2 o;
    o.x = pack2<T>(v[0], v[1]);
    o.y = pack2<T>(v[2], v[3]);
    *reinterpret_cast<uint2 *>(p.y + (int64_t)col * p.y_row_stride + n) = o;
}

// Batches of 17..64 rows (the continuous-batching and tensor-parallel decode regimes: config[2] tails, config[3] at bs = 64):
// one workgroup of 4 wavefronts owns 64 output features (PAIR: 64 gate + the 64 matching up features) x ALL batch rows over a
// K range.  What the two kernels above get wrong at these sizes is the x operand: every wavefront re-read its own K slice of
// x from L2 in fragment-shaped 64-byte pieces, 4 x the weight traffic at 64 rows.  Here
//   * x is staged ONCE per workgroup and 128-input chunk, in full 256-byte row pieces (a wave instruction = 4 rows x 256 B),
//     into an XOR-swizzled LDS tile [row][chunk ^ row] that all four wavefronts read as MFMA B operands (conflict-free);
//     x traffic = (N / 64) x |x| -- about the size of W itself at 64 rows;
//   * W keeps the 4 rows x 256 B streaming pattern (6.7 TB/s in the probe) and is re-laid into the A-operand order through
//     a private 4 KiB LDS tile per wavefront, as in linear_wg_kernel<RELAY>;
//   * global loads of chunk c + 1 are issued before the MFMAs of chunk c and written to LDS after them (x double-buffered,
//     one workgroup barrier per chunk); 4.CT MFMAs (16x16x32) per wavefront and chunk against 4 + 4.CT LDS reads;
//   * K is split over workgroups only as far as needed to put ~2 workgroups on every CU; the fp32 partials and the
//     epilogues then go through linear_reduce_kernel, otherwise the epilogue runs here (one launch).
template <typename T, int CT, bool PAIR>
__global__ void __launch_bounds__(256, 2) linear_mid_kernel(const LinearParams p) {
    constexpr int RT = PAIR ? 2 : 1;
    constexpr int XT = CT * 4096;                              // one x tile: 16.CT rows x 256 B
    __shared__ __attribute__((aligned(16))) char smem[2 * XT + 4 * RT * 4096];   // ONE LDS object: x tiles [2], then the relay tiles [wave][r]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = lane >> 4, col = lane & 15;
    const int out_n = PAIR ? p.n / 2 : p.n;
    const int tiles_n = out_n >> 6;
    const int tile = blockIdx.x % tiles_n, split = blockIdx.x / tiles_n;
    const int n0 = tile * 64 + wave * 16;
    const int chunks = p.k >> 7;
    const int c0 = split * p.chunks_per_split, c1 = min(c0 + p.chunks_per_split, chunks);
    char *relay = smem + 2 * XT + wave * RT * 4096;

    const int64_t row_bytes = p.w_row_stride * 2;
    const char *wrow = reinterpret_cast<const char *>(p.w + (int64_t)n0 * p.w_row_stride);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wrow), 0, 0x7fffffff, 0x00020000);
    uint32_t w_lane[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) w_lane[r] = (uint32_t)(((int64_t)r * out_n + grp) * row_bytes + col * 16);   // row 4.q + grp, chunk col
    // x staging: thread -> (row xr of a 16-row slab, 16-byte chunk xc); rows beyond the batch re-read the last row (never stored)
    const int xr = tid >> 4, xc = tid & 15;
    const uint16_t *xsrc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) xsrc[c] = p.x + (int64_t)min(16 * c + xr, p.batch - 1) * p.x_row_stride + xc * 8;
    const int x_st = xr * 256 + ((xc ^ xr) & 15) * 16;         // + c * 4096 inside a tile

    // two register stages: chunk c + 2 is requested while chunk c is multiplied and chunk c + 1 (requested one step earlier)
    // moves from its stage into LDS -- two chunks (2 x 16 KiB of W per workgroup) are always in flight, which is what the HBM
    // latency needs at two workgroups per CU (one chunk in flight: 3.9-4.8 TB/s, the loop ran at the latency of a load)
    lu32x4 wst[2][RT][4], xst[2][CT];
    auto gload = [&](auto ST, int chunk) {
        constexpr int st = decltype(ST)::value;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) wst[st][r][q] = __builtin_amdgcn_raw_buffer_load_b128(wr, w_lane[r], (int)(q * 4 * row_bytes + chunk * 256), 2 /* nt */);
#pragma unroll
        for (int c = 0; c < CT; ++c) xst[st][c] = *reinterpret_cast<const lu32x4 *>(xsrc[c] + chunk * 128);
    };
    auto lds_store = [&](auto ST, int buf) {
        constexpr int st = decltype(ST)::value;
        char *xt = smem + buf * XT;
#pragma unroll
        for (int c = 0; c < CT; ++c) *reinterpret_cast<lu32x4 *>(xt + c * 4096 + x_st) = xst[st][c];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * q + grp;
                *reinterpret_cast<lu32x4 *>(relay + r * 4096 + row * 256 + ((col ^ row) & 15) * 16) = wst[st][r][q];
            }
    };
    lf32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = lf32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const char *xt = smem + buf * XT;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sw = (((4 * q + grp) ^ col) & 15) * 16;
            lu32x4 a[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = *reinterpret_cast<const lu32x4 *>(relay + r * 4096 + col * 256 + sw);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const lu32x4 b = *reinterpret_cast<const lu32x4 *>(xt + c * 4096 + col * 256 + sw);
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][c] = lin_mfma<T>(a[r], b, acc[r][c]);
            }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    // one chunk: PAR = parity of (chunk - c0) = LDS buffer and register stage of this chunk; FULL = no bounds checks
    auto step = [&](auto PAR, auto FULL, int c) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool full = decltype(FULL)::value;
        if (full || c + 2 < c1) gload(std::integral_constant<int, par>{}, c + 2);       // this chunk's stage is free: it sits in LDS already
        compute(par);
        if (full || c + 1 < c1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this wavefront's relay reads above stay ahead of the stores below
            __builtin_amdgcn_wave_barrier();
            lds_store(std::integral_constant<int, par ^ 1>{}, par ^ 1);
        }
        __syncthreads();
    };
    if (c0 < c1) {
        gload(S0{}, c0);
        if (c0 + 1 < c1) gload(S1{}, c0 + 1);
        lds_store(S0{}, 0);
        __syncthreads();
        int c = c0;
        for (; c + 3 < c1; c += 2) {
            step(S0{}, std::true_type{}, c);
            step(S1{}, std::true_type{}, c + 1);
        }
        for (; c < c1; c += 2) {
            step(S0{}, std::false_type{}, c);
            if (c + 1 < c1) step(S1{}, std::false_type{}, c + 1);
        }
    }
    // lane holds y^T[n0 + 4.grp + i][batch row 16.c + col]
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int brow = 16 * c + col;
        if (brow >= p.batch) continue;
        const int n = n0 + 4 * grp;
        if (p.partial) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const lf32x4 a = acc[r][c];
                *reinterpret_cast<float4 *>(p.partial + ((int64_t)split * p.batch + brow) * p.n + (int64_t)r * out_n + n) = make_float4(a[0], a[1], a[2], a[3]);
            }
        } else {                                               // rounding points as in linear_reduce_kernel
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = round_through<T>(acc[0][c][i]);
            if (p.epilogue == 1) {
                const uint2 rr = *reinterpret_cast<const uint2 *>(p.aux + (int64_t)brow * p.aux_row_stride + n);
                v[0] += lo_to_f32<T>(rr.x); v[1] += hi_to_f32<T>(rr.x); v[2] += lo_to_f32<T>(rr.y); v[3] += hi_to_f32<T>(rr.y);
            }
            if constexpr (PAIR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = round_through<T>(v[i] / (1.f + __expf(-v[i]))) * round_through<T>(acc[1][c][i]);
            }
            uint2 o;
            o.x = pack2<T>(v[0], v[1]);
            o.y = pack2<T>(v[2], v[3]);
            *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n) = o;
        }
    }
}


// Batches of 65..256 rows (continuous batching at bs <= 256, config[2]): 2.B flop per weight byte reaches the machine balance
// here, so this is a real GEMM tile -- one workgroup of 4 wavefronts (one per SIMD, 512 registers) owns 128 weight rows
// (PAIR: 64 gate + the 64 matching up rows) x BM = 128 or 256 batch rows over a K range, v_mfma_f32_32x32x16:
//   * wavefront (wm, wn) of a 2 x 2 grid: BM / 2 batch rows x 64 weight rows = 2 A fragments x BM / 64 B fragments; every
//     fragment read from LDS feeds 2-4 MFMAs (6 KiB of LDS reads per 8 MFMAs = 75 % of the LDS pipe at the MFMA rate);
//   * K advances in chunks of 64 inputs: W tile 128 x 128 B and x tile BM x 128 B, staged through registers in full
//     128-byte lines (a wave instruction = 8 rows x 128 B), two chunks in flight, LDS double-buffered, one barrier per chunk;
//     LDS image [row][chunk ^ ((row >> 1) & 7)]: a 16-lane group reading 16 consecutive rows at one k-chunk hits all 64 banks;
//   * weights are read once per workgroup (x is re-read N / 128 times from L2: 2 MB at 256 rows, K = 4096);
//   * K is split over workgroups for the layers with few row tiles (q/k/v, o, down); partials + epilogues then go through
//     linear_reduce_kernel; otherwise the epilogue runs here with the reference's rounding points.
typedef __attribute__((ext_vector_type(16))) float lf32x16;
template <typename T> __device__ __forceinline__ lf32x16 lin_mfma32(const lu32x4 &a, const lu32x4 &b, lf32x16 c);
template <> __device__ __forceinline__ lf32x16 lin_mfma32<bf16_t>(const lu32x4 &a, const lu32x4 &b, lf32x16 c) {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ lf32x16 lin_mfma32<f16_t>(const lu32x4 &a, const lu32x4 &b, lf32x16 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 v8;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
}

template <typename T, int BM, bool PAIR>
__global__ void __launch_bounds__(256, 1) linear_big_kernel(const LinearParams p) {
    constexpr int WT = 128 * 128;                              // W tile bytes: 128 rows x 64 inputs
    constexpr int XT = BM * 128;                               // x tile bytes
    constexpr int NB = BM / 64;                                // B fragments (32 batch rows each) per wavefront
    constexpr int XL = BM / 32;                                // x staging loads per thread and chunk
    __shared__ __attribute__((aligned(16))) char smem[2 * (WT + XT)];   // [buffer][W tile | x tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, kh = lane >> 5;
    const int out_n = PAIR ? p.n / 2 : p.n;
    const int tiles_n = PAIR ? out_n >> 6 : out_n >> 7;
    const int tile = blockIdx.x % tiles_n, split = blockIdx.x / tiles_n;
    const int n0 = PAIR ? tile * 64 : tile * 128;              // first output feature of the workgroup
    const int chunks = p.k >> 6;                               // 64-input chunks
    const int cps = p.chunks_per_split * 2;                    // chunks_per_split counts 128-input chunks
    const int c0 = split * cps, c1 = min(c0 + cps, chunks);

    // staging: thread -> row (tid >> 3) of a 32-row slab, 16-byte chunk (tid & 7)
    const int sr = tid >> 3, sc = tid & 7;
    const uint16_t *wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * i + sr;                             // W tile row: PAIR rows 0..63 gate, 64..127 up
        const int64_t wrow = PAIR ? (r < 64 ? n0 + r : out_n + n0 + (r - 64)) : n0 + r;
        wsrc[i] = p.w + wrow * p.w_row_stride + sc * 8;
    }
    const uint16_t *xsrc[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) xsrc[i] = p.x + (int64_t)min(32 * i + sr, p.batch - 1) * p.x_row_stride + sc * 8;
    const int st_off = sr * 128 + ((sc ^ ((sr >> 1) & 7)) & 7) * 16;    // + 32-row slab * 4096 (rows 32 i + sr: (row >> 1) & 7 == (sr >> 1) & 7)

    lu32x4 wst[2][4], xst[2][XL];
    auto gload = [&](auto ST, int chunk) {
        constexpr int st = decltype(ST)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) wst[st][i] = __builtin_nontemporal_load(reinterpret_cast<const lu32x4 *>(wsrc[i] + chunk * 64));
#pragma unroll
        for (int i = 0; i < XL; ++i) xst[st][i] = *reinterpret_cast<const lu32x4 *>(xsrc[i] + chunk * 64);
    };
    auto lds_store = [&](auto ST, int buf) {
        constexpr int st = decltype(ST)::value;
        char *wt = smem + buf * (WT + XT), *xt = wt + WT;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<lu32x4 *>(wt + i * 4096 + st_off) = wst[st][i];
#pragma unroll
        for (int i = 0; i < XL; ++i) *reinterpret_cast<lu32x4 *>(xt + i * 4096 + st_off) = xst[st][i];
    };
    lf32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    // fragment rows of this lane inside the tiles
    int a_row[2], b_row[NB];
#pragma unroll
    for (int a = 0; a < 2; ++a) a_row[a] = (PAIR ? 64 * a + 32 * wn : 64 * wn + 32 * a) + l32;
#pragma unroll
    for (int b = 0; b < NB; ++b) b_row[b] = (BM / 2) * wm + 32 * b + l32;
    // Fragments are read from LDS ONE k16 step ahead of the MFMAs that use them (two register sets), also across the chunk
    // boundary: the barrier sits in front of the LAST step's MFMAs, the first fragments of the next chunk are requested right
    // behind it and their latency is covered by those 8 MFMAs; the ds_writes of the next chunk go out in the middle of the
    // current one.  The order is pinned with sched_barrier(0): left alone, hipcc issues every ds_read right in front of its
    // MFMA (seen in the ISA: two exposed LDS latencies per k16 step, ~3500 cycles per chunk for 1024 cycles of MFMA).
    struct Frag { lu32x4 a[2], b[NB]; };
    auto read_frags = [&](int buf, int s4, Frag &f) {          // k16 step s4 of the chunk in `buf`: lane's 16-byte chunk 2 s4 + kh
        const char *wt = smem + buf * (WT + XT), *xt = wt + WT;
#pragma unroll
        for (int a = 0; a < 2; ++a) f.a[a] = *reinterpret_cast<const lu32x4 *>(wt + a_row[a] * 128 + (((2 * s4 + kh) ^ ((a_row[a] >> 1) & 7)) & 7) * 16);
#pragma unroll
        for (int b = 0; b < NB; ++b) f.b[b] = *reinterpret_cast<const lu32x4 *>(xt + b_row[b] * 128 + (((2 * s4 + kh) ^ ((b_row[b] >> 1) & 7)) & 7) * 16);
    };
    auto mfmas = [&](const Frag &f) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a][b] = lin_mfma32<T>(f.a[a], f.b[b], acc[a][b]);
    };
    Frag fA, fB;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    // one chunk; on entry fA holds step 0 of chunk c (LDS buffer par); on exit fA holds step 0 of chunk c + 1 (if any)
    auto step = [&](auto PAR, auto FULL, int c) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool full = decltype(FULL)::value;
        const bool more1 = full || c + 1 < c1;
        if (full || c + 2 < c1) gload(std::integral_constant<int, par>{}, c + 2);
        read_frags(par, 1, fB);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fA);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(par, 2, fA);
        if (more1) lds_store(std::integral_constant<int, par ^ 1>{}, par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fB);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(par, 3, fB);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fA);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                       // everybody's chunk c + 1 is in LDS; nobody reads chunk c from LDS any more
        if (more1) read_frags(par ^ 1, 0, fA);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fB);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (c0 < c1) {
        gload(S0{}, c0);
        if (c0 + 1 < c1) gload(S1{}, c0 + 1);
        lds_store(S0{}, 0);
        __syncthreads();
        read_frags(0, 0, fA);
        int c = c0;
        for (; c + 3 < c1; c += 2) {
            step(S0{}, std::true_type{}, c);
            step(S1{}, std::true_type{}, c + 1);
        }
        for (; c < c1; c += 2) {
            step(S0{}, std::false_type{}, c);
            if (c + 1 < c1) step(S1{}, std::false_type{}, c + 1);
        }
    }
    // C layout of 32x32: lane holds, for batch row (lane & 31) of the B fragment, weight rows 8 j + 4 kh + i of the A fragment (reg 4 j + i)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int brow = (BM / 2) * wm + 32 * b + l32;
        if (brow >= p.batch) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (p.partial) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int64_t col = PAIR ? (int64_t)a * out_n + n0 + 32 * wn + 8 * j + 4 * kh : (int64_t)n0 + 64 * wn + 32 * a + 8 * j + 4 * kh;
                    *reinterpret_cast<float4 *>(p.partial + ((int64_t)split * p.batch + brow) * p.n + col) =
                        make_float4(acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]);
                }
            } else if constexpr (PAIR) {
                const int n = n0 + 32 * wn + 8 * j + 4 * kh;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g = round_through<T>(acc[0][b][4 * j + i]);
                    v[i] = round_through<T>(g / (1.f + __expf(-g))) * round_through<T>(acc[1][b][4 * j + i]);
                }
                uint2 o;
                o.x = pack2<T>(v[0], v[1]);
                o.y = pack2<T>(v[2], v[3]);
                *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n) = o;
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int n = n0 + 64 * wn + 32 * a + 8 * j + 4 * kh;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = round_through<T>(acc[a][b][4 * j + i]);
                    if (p.epilogue == 1) {
                        const uint2 rr = *reinterpret_cast<const uint2 *>(p.aux + (int64_t)brow * p.aux_row_stride + n);
                        v[0] += lo_to_f32<T>(rr.x); v[1] += hi_to_f32<T>(rr.x); v[2] += lo_to_f32<T>(rr.y); v[3] += hi_to_f32<T>(rr.y);
                    }
                    uint2 o;
                    o.x = pack2<T>(v[0], v[1]);
                    o.y = pack2<T>(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(p.y + (int64_t)brow * p.y_row_stride + n) = o;
                }
            }
        }
    }
}

// Batches of 1..8 rows, loads laid out for the memory system instead of the matrix cores.  tools/probes/weight_stream_probe.hip:
// with no arithmetic at all, 16 rows x 64 bytes per load instruction (the MFMA A-operand layout used above) streams lm_head at
// 5.7 TB/s, 4 rows x 256 bytes per instruction at 6.7 TB/s (the whole 128 KiB block front to back: 6.85).  At 1..4 batch rows
// the arithmetic is tiny (2.B flop per byte), so it moves to the VALU: lane = 16.g + c loads 16 bytes (8 inputs, chunk c of a
// 128-input slab) of row 4.rg + g, multiplies them with the matching 8 inputs of every batch row (v_dot2c, fp32) and the 16
// lanes of a row are summed once at the end (DPP).  Workgroup structure, split of K over NW wavefronts, LDS merge and the
// epilogues are those of linear_wg_kernel.
__device__ __forceinline__ float lin_row16_sum(float x) {   // sum over the 16 lanes of a DPP row
    x += __builtin_amdgcn_update_dpp(0.f, x, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0.f, x, 0x140, 0xf, 0xf, true);   // row_mirror
    return x;
}
// NORM: y = rms_norm(x; norm_w, eps) . W^T -- the RMSNorm in front of the q/k/v and the gate/up projections (llama.rs:402,408)
// folded into the projection: every wavefront derives the row's scale itself, with the norm kernel's own arithmetic
// (norm_shared.h: the lane plays the four wavefronts of that kernel's workgroup in turn; 8 KiB of x per batch row from L2, while
// the first weight loads are in flight), and normalises + rounds its 16-byte pieces of x as they arrive -- the operand of the dot
// products is bit for bit what the separate kernel would have written, so the results are identical and one launch per norm goes.
template <typename T, int NW, bool PAIR, int NB, int P, bool NORM = false>
__global__ void __launch_bounds__(64 * NW) linear_gemv_kernel(const LinearParams p) {
    constexpr int RT = PAIR ? 2 : 1;
    __shared__ float red[NW][RT][16][NB < 4 ? 4 : NB];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
    const int n0 = blockIdx.x * 16;
    const int out_n = PAIR ? p.n / 2 : p.n;
    const int chunks = p.k >> 7, per = (chunks + NW - 1) / NW;
    const int c0 = wave * per, c1 = min(c0 + per, chunks);
    const int64_t row_bytes = p.w_row_stride * 2;

    const char *wrow = reinterpret_cast<const char *>(p.w + (int64_t)n0 * p.w_row_stride);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wrow), 0, 0x7fffffff, 0x00020000);
    uint32_t w_lane[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) w_lane[r] = (uint32_t)(((int64_t)r * out_n + g) * row_bytes + c * 16);
    const uint16_t *xrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) xrow[b] = p.x + (int64_t)min(b, p.batch - 1) * p.x_row_stride + c * 8;   // rows beyond the batch: recomputed, never stored

    lu32x4 wb[P][RT][4], xb[P][NB], gb[P];
    auto issue = [&](int s, int chunk) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                wb[s][r][rg] = __builtin_amdgcn_raw_buffer_load_b128(wr, w_lane[r], (int)(rg * 4 * row_bytes + chunk * 256), 2 /* nt */);
#pragma unroll
        for (int b = 0; b < NB; ++b) xb[s][b] = *reinterpret_cast<const lu32x4 *>(xrow[b] + chunk * 128);
        if constexpr (NORM) gb[s] = *reinterpret_cast<const lu32x4 *>(p.norm_w + c * 8 + chunk * 128);
    };
    float acc[RT][4][NB];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[r][rg][b] = 0.f;
    float nscale[NB];
    auto compute = [&](int s) {
        if constexpr (NORM) {
            const uint4 gv = make_uint4(gb[s][0], gb[s][1], gb[s][2], gb[s][3]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint4 xn = norm_apply8<T>(make_uint4(xb[s][b][0], xb[s][b][1], xb[s][b][2], xb[s][b][3]), gv, nscale[b]);
                xb[s][b] = lu32x4{xn.x, xn.y, xn.z, xn.w};
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float a = acc[r][rg][b];
                    a = dot2<T>(wb[s][r][rg][0], xb[s][b][0], a);
                    a = dot2<T>(wb[s][r][rg][1], xb[s][b][1], a);
                    a = dot2<T>(wb[s][r][rg][2], xb[s][b][2], a);
                    a = dot2<T>(wb[s][r][rg][3], xb[s][b][3], a);
                    acc[r][rg][b] = a;
                }
    };
    // NORM: the scales are derived between the first issue and the first compute (the weight loads cover it)
    auto scales = [&]() {
        if constexpr (NORM) norm_scales_by_one_wave<T, NB>(p.x, p.x_row_stride, p.batch, p.k, p.norm_eps, lane, nscale);
    };
    int ch = c0;
    if (c0 + 2 * P <= c1) {
#pragma unroll
        for (int s = 0; s < P; ++s) issue(s, c0 + s);
        scales();
        for (; ch + 2 * P <= c1; ch += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                compute(s);
                issue(s, ch + s + P);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (c0 + s < c1) issue(s, c0 + s);
        scales();
    }
    for (; ch < c1; ch += P) {
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (ch + s < c1) {
                compute(s);
                if (ch + s + P < c1) issue(s, ch + s + P);
            }
    }
    // sum the 16 lanes of every row; lane c == 0 of group g then holds row 4.rg + g
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float v = lin_row16_sum(acc[r][rg][b]);
                if (c == 0) red[wave][r][rg * 4 + g][b] = v;
            }
    __syncthreads();
    if (wave > 0) return;
    const int row = lane & 15, n = n0 + row;          // 16 rows x 4 batch rows per pass over the 64 lanes of wavefront 0
    for (int b = lane >> 4; b < NB && b < p.batch; b += 4) {
        float v[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float t = red[0][r][row][b];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += red[w][r][row][b];   // fixed order: deterministic
            v[r] = round_through<T>(t);
        }
        float out = v[0];
        if (p.epilogue == 1) out += lo_to_f32<T>((uint32_t)p.aux[(int64_t)b * p.aux_row_stride + n]);
        if constexpr (PAIR) out = round_through<T>(out / (1.f + __expf(-out))) * v[1];
        p.y[(int64_t)b * p.y_row_stride + n] = (uint16_t)f32_to_bits<T>(out);
    }
}

constexpr int LINEAR_NORM_MAX_BATCH = 2;
static const int linear_gemv = getenv("ATOMA_LINEAR_GEMV") ? atoi(getenv("ATOMA_LINEAR_GEMV")) : 1;
// Measured inside the 8B decode step: the VALU kernel wins at ONE row (3.33 against 3.49 ms); from 2 rows up the workgroup kernel
// with the matrix cores is faster (batch 2: 3.75 -> 3.64 ms, 3: 3.85 -> 3.79, 4: 4.00 -> 3.92), although the single-kernel benchmark
// still favoured this one up to 4 rows.
static const int linear_gemv_max_batch = getenv("ATOMA_LINEAR_GEMV_MAX_BATCH") ? atoi(getenv("ATOMA_LINEAR_GEMV_MAX_BATCH")) : 1;
template <typename T> static int launch_linear_gemv(LinearParams &p, hipStream_t stream) {
    static const int gemv_wpc = getenv("ATOMA_LINEAR_GEMV_WAVES_PER_CU") ? atoi(getenv("ATOMA_LINEAR_GEMV_WAVES_PER_CU")) : 32;
    const bool pair = p.epilogue == 2;
    const int64_t tiles = (pair ? p.n / 2 : p.n) / 16, chunks = p.k / 128;
    // wavefronts per workgroup (the K split inside it) from the W ROWS, not from the workgroups: the stacked gate / up launch then
    // splits K exactly like the plain projection of the same matrix and stays bit-identical to projection + atoma_silu_mul
    const int64_t row_tiles = p.n / 16;
    int nw = 1;
    while (nw < 8 && row_tiles * nw * 2 <= (int64_t)device_num_cus() * gemv_wpc && chunks / (nw * 2) >= 4) nw *= 2;
    const dim3 grid((unsigned)tiles), block(64 * nw);
    const bool norm = p.norm_w != nullptr;
#define ATOMA_GV3(NW_, NB_) do { if (pair) hipLaunchKernelGGL((linear_gemv_kernel<T, NW_, true, NB_, 2>), grid, block, 0, stream, p); \
                                 else hipLaunchKernelGGL((linear_gemv_kernel<T, NW_, false, NB_, (NB_ > 4 ? 2 : 3)>), grid, block, 0, stream, p); } while (0)
#define ATOMA_GV3N(NW_, NB_) do { if (!norm) ATOMA_GV3(NW_, NB_); \
                                  else if (pair) hipLaunchKernelGGL((linear_gemv_kernel<T, NW_, true, NB_, 2, true>), grid, block, 0, stream, p); \
                                  else hipLaunchKernelGGL((linear_gemv_kernel<T, NW_, false, NB_, 3, true>), grid, block, 0, stream, p); } while (0)
#define ATOMA_GV2(NW_) do { if (p.batch == 1) ATOMA_GV3N(NW_, 1); else if (p.batch == 2) ATOMA_GV3N(NW_, 2); else if (p.batch <= 4) ATOMA_GV3(NW_, 4); else ATOMA_GV3(NW_, 8); } while (0)
    switch (nw) {
        case 1: ATOMA_GV2(1); break;
        case 2: ATOMA_GV2(2); break;
        case 4: ATOMA_GV2(4); break;
        default: ATOMA_GV2(8); break;
    }
#undef ATOMA_GV2
#undef ATOMA_GV3N
#undef ATOMA_GV3
    return ATOMA_CHECK_LAUNCH("linear_gemv_kernel") ? 0 : -1;
}

static const int linear_wg = getenv("ATOMA_LINEAR_WG") ? atoi(getenv("ATOMA_LINEAR_WG")) : 1;
static const int linear_wg_max_batch = getenv("ATOMA_LINEAR_WG_MAX_BATCH") ? atoi(getenv("ATOMA_LINEAR_WG_MAX_BATCH")) : 16;
// wavefronts per workgroup: as many as keep ~8 wavefronts per CU streaming, each with at least 4 chunks (512 inputs)
template <typename T> static int launch_linear_wg(LinearParams &p, hipStream_t stream) {
    const bool pair = p.epilogue == 2;
    const int64_t tiles = (pair ? p.n / 2 : p.n) / 16, chunks = p.k / 128, row_tiles = p.n / 16;   // nw from the W rows: see launch_linear_gemv
    int nw = 1;
    // Measured in the whole 8B step (5..16 rows): 4 wavefronts per workgroup for q/k/v, o and down, ONE for gate/up beats the
    // earlier 4 / 8 / 8 / 2 by 7-10 % (fewer, longer streams per CU; the LDS merge of 8 partial sums is not free).
    static const int wg_wpc = getenv("ATOMA_LINEAR_WG_WAVES_PER_CU") ? atoi(getenv("ATOMA_LINEAR_WG_WAVES_PER_CU")) : 6;
    while (nw < 8 && row_tiles * nw * 2 <= (int64_t)device_num_cus() * wg_wpc && chunks / (nw * 2) >= 4) nw *= 2;
    static const int relay_on = getenv("ATOMA_LINEAR_RELAY") ? atoi(getenv("ATOMA_LINEAR_RELAY")) : 1;
    if (!relay_on && nw == 1 && p.epilogue == 0) return 1;   // MFMA-layout loads: the split-free path visits 512 bytes per row and is 3-5 % faster
    const dim3 grid((unsigned)tiles), block(64 * nw);
    static const int relay = getenv("ATOMA_LINEAR_RELAY") ? atoi(getenv("ATOMA_LINEAR_RELAY")) : 1;
#define ATOMA_LWG(NW_) do { if (relay) { if (pair) hipLaunchKernelGGL((linear_wg_kernel<T, NW_, true, 2, true>), grid, block, 0, stream, p); \
                                         else hipLaunchKernelGGL((linear_wg_kernel<T, NW_, false, 3, true>), grid, block, 0, stream, p); } \
                            else { if (pair) hipLaunchKernelGGL((linear_wg_kernel<T, NW_, true, 2>), grid, block, 0, stream, p); \
                                   else hipLaunchKernelGGL((linear_wg_kernel<T, NW_, false, 3>), grid, block, 0, stream, p); } } while (0)
    switch (nw) {
        case 1: ATOMA_LWG(1); break;
        case 2: ATOMA_LWG(2); break;
        case 4: ATOMA_LWG(4); break;
        default: ATOMA_LWG(8); break;
    }
#undef ATOMA_LWG
    return ATOMA_CHECK_LAUNCH("linear_wg_kernel") ? 0 : -1;
}

// 17..64 rows: linear_mid_kernel (x staged through LDS once per workgroup).  Needs 64 output features per workgroup.
static const int linear_mid = getenv("ATOMA_LINEAR_MID") ? atoi(getenv("ATOMA_LINEAR_MID")) : 1;
// measured in the whole step: ONE round of workgroups beats two (fewer partials to merge), no split at all loses
static const float linear_mid_wg_per_cu = getenv("ATOMA_LINEAR_MID_WG_PER_CU") ? (float)atof(getenv("ATOMA_LINEAR_MID_WG_PER_CU")) : 1.f;
template <typename T> static int launch_linear_mid(LinearParams &p, hipStream_t stream) {
    const bool pair = p.epilogue == 2;
    const int out_n = pair ? p.n / 2 : p.n;
    if (out_n % 64) return 1;                                  // not served: the caller falls through to linear_decode_kernel
    const int64_t tiles_n = out_n / 64, chunks = p.k / 128;
    const int ct = (p.batch + 15) / 16;
    // Split K so that the workgroups fill one round of the CUs (linear_mid_wg_per_cu per CU): the largest split count whose
    // workgroups still fit one round, never below 4 chunks (512 inputs) per split.  The count is derived from the W ROWS
    // (p.n / 64), not from the workgroups, so that the stacked gate / up launch with its SiLU.up epilogue splits K exactly like
    // the plain projection of the same matrix and stays bit-identical to projection + atoma_silu_mul.
    const int64_t target = (int64_t)((float)device_num_cus() * linear_mid_wg_per_cu), row_tiles = p.n / 64;
    int64_t splits = std::max<int64_t>(1, std::min<int64_t>(target / std::max<int64_t>(row_tiles, 1), chunks / 4));
    p.chunks_per_split = (int)cdiv(chunks, splits);
    p.splits = (int)cdiv(chunks, p.chunks_per_split);
    p.partial = nullptr;
    if (p.splits > 1) {
        p.partial = static_cast<float *>(workspace(stream, (size_t)p.splits * p.batch * p.n * sizeof(float)));
        if (!p.partial) return -1;
    }
    const dim3 grid((unsigned)(tiles_n * p.splits)), block(256);
#define ATOMA_MID(CT_) do { if (pair) hipLaunchKernelGGL((linear_mid_kernel<T, CT_, true>), grid, block, 0, stream, p); \
                            else hipLaunchKernelGGL((linear_mid_kernel<T, CT_, false>), grid, block, 0, stream, p); } while (0)
    switch (ct) {
        case 1: ATOMA_MID(1); break;
        case 2: ATOMA_MID(2); break;
        case 3: ATOMA_MID(3); break;
        default: ATOMA_MID(4); break;
    }
#undef ATOMA_MID
    if (!ATOMA_CHECK_LAUNCH("linear_mid_kernel")) return -1;
    if (p.partial) {
        hipLaunchKernelGGL((linear_reduce_kernel<T>), dim3((unsigned)cdiv((int64_t)p.batch * out_n, 1024)), dim3(256), 0, stream, p);
        if (!ATOMA_CHECK_LAUNCH("linear_reduce_kernel")) return -1;
    }
    return 0;
}


// 65..256 rows: linear_big_kernel.  128 weight rows per workgroup (PAIR: 64 gate + 64 up), one workgroup per CU.
template <typename T> static int launch_linear_big(LinearParams &p, hipStream_t stream) {
    const bool pair = p.epilogue == 2;
    const int out_n = pair ? p.n / 2 : p.n;
    if (p.n % 128 || p.k % 128) return 1;
    const int64_t row_tiles = p.n / 128, chunks = p.k / 128;
    // split K (in units of 128 inputs) until the workgroups fill one round of the CUs; derived from the W rows, so that the
    // stacked gate / up launch with its epilogue splits exactly like the plain projection (bit-identical results)
    static const float big_wg_per_cu = getenv("ATOMA_LINEAR_BIG_WG_PER_CU") ? (float)atof(getenv("ATOMA_LINEAR_BIG_WG_PER_CU")) : 1.f;
    const int64_t target = (int64_t)((float)device_num_cus() * big_wg_per_cu);
    int64_t splits = std::max<int64_t>(1, std::min<int64_t>(target / std::max<int64_t>(row_tiles, 1), chunks / 4));
    p.chunks_per_split = (int)cdiv(chunks, splits);
    p.splits = (int)cdiv(chunks, p.chunks_per_split);
    p.partial = nullptr;
    if (p.splits > 1) {
        p.partial = static_cast<float *>(workspace(stream, (size_t)p.splits * p.batch * p.n * sizeof(float)));
        if (!p.partial) return -1;
    }
    const int64_t tiles_n = pair ? out_n / 64 : out_n / 128;
    const dim3 grid((unsigned)(tiles_n * p.splits)), block(256);
#define ATOMA_BIG(BM_) do { if (pair) hipLaunchKernelGGL((linear_big_kernel<T, BM_, true>), grid, block, 0, stream, p); \
                            else hipLaunchKernelGGL((linear_big_kernel<T, BM_, false>), grid, block, 0, stream, p); } while (0)
    if (p.batch <= 128) ATOMA_BIG(128); else ATOMA_BIG(256);
#undef ATOMA_BIG
    if (!ATOMA_CHECK_LAUNCH("linear_big_kernel")) return -1;
    if (p.partial) {
        hipLaunchKernelGGL((linear_reduce_kernel<T>), dim3((unsigned)cdiv((int64_t)p.batch * out_n, 1024)), dim3(256), 0, stream, p);
        if (!ATOMA_CHECK_LAUNCH("linear_reduce_kernel")) return -1;
    }
    return 0;
}

// 512 bytes per weight-row visit (two 128-input chunks per pipeline stage) when the split allows: +2-5 % over 256
static const int linear_ch = getenv("ATOMA_LINEAR_CH") ? atoi(getenv("ATOMA_LINEAR_CH")) : 2;
template <typename T> static int launch_linear(LinearParams &p, hipStream_t stream) {
    // the self-normalising variant serves 1 and 2 rows: at 4 rows the normalisation of x (45 VALU instructions per row and 128-input
    // chunk, repeated by every wavefront) makes the kernel VALU-bound -- measured 4.82 ms against 4.06 ms for the 8B step
    if (linear_gemv && p.batch <= std::min(linear_gemv_max_batch, 8) && !(p.norm_w && p.batch > LINEAR_NORM_MAX_BATCH)) return launch_linear_gemv<T>(p, stream);
    if (p.norm_w) return 2;                                                      // only the kernel above normalises its own input: the entry point runs the norm first
    if (linear_wg && p.batch <= std::min(linear_wg_max_batch, 16)) {
        const int rc = launch_linear_wg<T>(p, stream);
        if (rc <= 0) return rc;
    }
    if (p.batch > 64) {
        int rc = launch_linear_wide(p, std::is_same<T, bf16_t>::value ? ATOMA_BF16 : ATOMA_F16, stream);   // round 6: LDS-DMA tile, 8 wavefronts
        if (rc <= 0) return rc;
        rc = launch_linear_big<T>(p, stream);
        if (rc <= 0) return rc;
        // a shape neither tile kernel takes (out_features not a multiple of 64, ...): the batch in slices of 64 rows through the kernels for
        // 17..64 rows -- every epilogue is row-wise, so slices are independent (correct for any shape the small-batch path accepts; not fast)
        for (int b0 = 0; b0 < p.batch; b0 += 64) {
            LinearParams q = p;
            q.batch = std::min(64, p.batch - b0);
            q.x = p.x + (int64_t)b0 * p.x_row_stride;
            q.y = p.y + (int64_t)b0 * p.y_row_stride;
            if (p.aux) q.aux = p.aux + (int64_t)b0 * p.aux_row_stride;
            q.partial = nullptr;
            rc = launch_linear<T>(q, stream);
            if (rc != 0) return rc;
        }
        return 0;
    }
    if (p.batch > 16) {
        int rc = launch_linear_tile(p, std::is_same<T, bf16_t>::value ? ATOMA_BF16 : ATOMA_F16, stream);
        if (rc < 0) return rc;
        if (rc == 0) {
            if (p.partial) {
                const int out_n = p.epilogue == 2 ? p.n / 2 : p.n;
                hipLaunchKernelGGL((linear_reduce_kernel<T>), dim3((unsigned)cdiv((int64_t)p.batch * out_n, 1024)), dim3(256), 0, stream, p);
                if (!ATOMA_CHECK_LAUNCH("linear_reduce_kernel")) return -1;
            }
            return 0;
        }
    }
    if (linear_mid && p.batch > 16) {
        const int rc = launch_linear_mid<T>(p, stream);
        if (rc <= 0) return rc;
    }
    const int64_t chunks = p.k / 128;
    const int ct = (p.batch + 15) / 16;                                          // column tiles of 16 batch rows
    // row tiles per wave: bounded by the 256 registers of two wavefronts per SIMD (x fragments: 16 registers per column tile and chunk in flight)
    const int rt = ct > 1 ? ((ct < 4 && p.n % 32 == 0) ? 2 : 1) : ((p.batch > 2 && p.n % 64 == 0) ? 4 : 1);
    const int64_t tiles_n = p.n / (16 * rt);
    // split K until about 8 (one row tile) / 4 (more bytes in flight per wave) wavefronts per CU stream the weights,
    // never below 4 chunks (512 inputs) per split
    const int64_t target = (int64_t)device_num_cus() * (rt > 1 ? 4 : 8);
    int64_t splits = std::max<int64_t>(1, std::min<int64_t>(target / std::max<int64_t>(tiles_n, 1), chunks / 4));
    p.chunks_per_split = (int)cdiv(chunks, splits);
    p.splits = (int)cdiv(chunks, p.chunks_per_split);
    p.partial = nullptr;
    if (p.splits > 1 || p.epilogue != 0) {   // an epilogue always runs in the reduce kernel (it needs whole rows of y)
        p.partial = static_cast<float *>(workspace(stream, (size_t)p.splits * p.batch * p.n * sizeof(float)));
        if (!p.partial) return -1;
    }
    const dim3 grid((unsigned)(tiles_n * p.splits));
    const int ch = (rt == 1 && ct == 1 && p.chunks_per_split % 2 == 0 && chunks % 2 == 0) ? linear_ch : 1;
#define ATOMA_LIN(RT_, CT_, P_, CH_) hipLaunchKernelGGL((linear_decode_kernel<T, RT_, CT_, P_, CH_>), grid, dim3(64), 0, stream, p)
    if (ct == 1) {
        if (rt == 4) ATOMA_LIN(4, 1, 2, 1);
        else if (ch == 2) ATOMA_LIN(1, 1, 2, 2);
        else ATOMA_LIN(1, 1, 3, 1);
    } else if (rt == 2) {
        if (ct == 2) ATOMA_LIN(2, 2, 2, 1);
        else ATOMA_LIN(2, 3, 2, 1);
    } else {
        if (ct == 2) ATOMA_LIN(1, 2, 2, 1);
        else if (ct == 3) ATOMA_LIN(1, 3, 2, 1);
        else ATOMA_LIN(1, 4, 2, 1);
    }
#undef ATOMA_LIN
    if (!ATOMA_CHECK_LAUNCH("linear_decode_kernel")) return -1;
    if (p.partial) {
        hipLaunchKernelGGL((linear_reduce_kernel<T>), dim3((unsigned)cdiv((int64_t)p.batch * p.n, 1024)), dim3(256), 0, stream, p);
        if (!ATOMA_CHECK_LAUNCH("linear_reduce_kernel")) return -1;
    }
    return 0;
}

}  // namespace atoma

extern "C" int atoma_rms_norm(const void *x, const void *weight, void *y, int64_t rows, int64_t hidden, int64_t x_row_stride,
                              int64_t y_row_stride, float eps, int dtype, void *stream);

static int linear_decode_entry(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                               int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int epilogue, const void *aux,
                               int64_t aux_row_stride, int dtype, void *stream, const void *norm_w = nullptr, float norm_eps = 0.f,
                               void *xn_scratch = nullptr) {
    using namespace atoma;
    clear_error();
    if (norm_w && (reinterpret_cast<uintptr_t>(norm_w) & 15u)) { set_error("linear_decode: the norm weight must be 16-byte aligned"); return -1; }
    if (norm_w && in_features > 8 * 8 * NORM_THREADS) { set_error("linear_decode: the fused RMSNorm supports in_features <= 16384"); return -1; }
    if (dtype != ATOMA_F16 && dtype != ATOMA_BF16) { set_error("linear_decode: dtype must be f16 or bf16"); return -1; }
    if (batch < 0 || batch > 256) { set_error("linear_decode: batch must be in [0, 256]"); return -1; }
    if (in_features <= 0 || in_features % 128 != 0) { set_error("linear_decode: in_features must be a positive multiple of 128"); return -1; }
    if (out_features <= 0 || out_features % 16 != 0) { set_error("linear_decode: out_features must be a positive multiple of 16"); return -1; }
    const int64_t y_width = epilogue == 2 ? out_features / 2 : out_features;
    if (epilogue < 0 || epilogue > 2) { set_error("linear_decode: unknown epilogue"); return -1; }
    if (epilogue == 2 && out_features % 32 != 0) { set_error("linear_decode: the silu.up epilogue needs out_features to be a multiple of 32"); return -1; }
    if (epilogue == 1 && (!aux || aux_row_stride < out_features || aux_row_stride % 4 || (reinterpret_cast<uintptr_t>(aux) & 7u))) {
        set_error("linear_decode: the residual epilogue needs an 8-byte aligned [batch, out_features] tensor");
        return -1;
    }
    if (x_row_stride < in_features || w_row_stride < in_features || y_row_stride < y_width) {
        set_error("linear_decode: row strides must cover a row");
        return -1;
    }
    if (x_row_stride % 8 || w_row_stride % 8 || y_row_stride % 4) { set_error("linear_decode: strides must keep rows 16-byte (x, w) / 8-byte (y) aligned"); return -1; }
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u || reinterpret_cast<uintptr_t>(y) & 7u) {
        set_error("linear_decode: x and w must be 16-byte aligned, y 8-byte aligned");
        return -1;
    }
    // the stacked gate / up kernels reach the up rows through a 32-bit buffer offset from the gate tile's base
    if (epilogue == 2 && (out_features / 2 + 16) * w_row_stride * 2 >= ((int64_t)1 << 31)) {
        set_error("linear_decode: the silu.up epilogue needs the gate block [intermediate, in_features] to stay below 2 GiB (32-bit buffer offsets)");
        return -1;
    }
    if (batch == 0) return 0;
    LinearParams p{};
    p.x = static_cast<const uint16_t *>(x);
    p.w = static_cast<const uint16_t *>(w);
    p.y = static_cast<uint16_t *>(y);
    p.x_row_stride = x_row_stride; p.w_row_stride = w_row_stride; p.y_row_stride = y_row_stride;
    p.batch = (int)batch; p.n = (int)out_features; p.k = (int)in_features;
    p.epilogue = epilogue; p.aux = static_cast<const uint16_t *>(aux); p.aux_row_stride = aux_row_stride;
    p.norm_w = static_cast<const uint16_t *>(norm_w); p.norm_eps = norm_eps;
    const auto s = static_cast<hipStream_t>(stream);
    int rc = dtype == ATOMA_BF16 ? launch_linear<bf16_t>(p, s) : launch_linear<f16_t>(p, s);
    if (rc == 2) {   // a batch the self-normalising kernel does not take: the two ops, through the caller's scratch rows
        if (!xn_scratch || (reinterpret_cast<uintptr_t>(xn_scratch) & 15u)) {
            set_error("linear_decode: this batch size needs a 16-byte aligned xn_scratch [batch, in_features] for the normalised rows");
            return -1;
        }
        if (atoma_rms_norm(x, norm_w, xn_scratch, batch, in_features, x_row_stride, in_features, norm_eps, dtype, stream) != 0) return -1;
        p.x = static_cast<const uint16_t *>(xn_scratch); p.x_row_stride = in_features; p.norm_w = nullptr;
        rc = dtype == ATOMA_BF16 ? launch_linear<bf16_t>(p, s) : launch_linear<f16_t>(p, s);
    }
    return rc;
}

extern "C" int atoma_linear_decode(const void *x, const void *w, void *y, int64_t batch, int64_t in_features, int64_t out_features,
                                   int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream) {
    return linear_decode_entry(x, w, y, batch, in_features, out_features, x_row_stride, w_row_stride, y_row_stride, 0, nullptr, 0, dtype, stream);
}
extern "C" int atoma_linear_decode_residual(const void *x, const void *w, const void *residual, void *y, int64_t batch, int64_t in_features,
                                            int64_t out_features, int64_t x_row_stride, int64_t w_row_stride, int64_t residual_row_stride,
                                            int64_t y_row_stride, int dtype, void *stream) {
    return linear_decode_entry(x, w, y, batch, in_features, out_features, x_row_stride, w_row_stride, y_row_stride, 1, residual,
                               residual_row_stride, dtype, stream);
}
extern "C" int atoma_linear_decode_rmsnorm(const void *x, const void *norm_weight, float eps, const void *w, void *y, void *xn_scratch, int64_t batch,
                                           int64_t in_features, int64_t out_features, int64_t x_row_stride, int64_t w_row_stride,
                                           int64_t y_row_stride, int dtype, void *stream) {
    if (!norm_weight) { atoma::clear_error(); atoma::set_error("linear_decode_rmsnorm: norm_weight is required"); return -1; }
    return linear_decode_entry(x, w, y, batch, in_features, out_features, x_row_stride, w_row_stride, y_row_stride, 0, nullptr, 0, dtype, stream,
                               norm_weight, eps, xn_scratch);
}
extern "C" int atoma_linear_decode_rmsnorm_silu_mul(const void *x, const void *norm_weight, float eps, const void *w_gate_up, void *y,
                                                    void *xn_scratch, int64_t batch, int64_t in_features, int64_t intermediate,
                                                    int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype, void *stream) {
    if (!norm_weight) { atoma::clear_error(); atoma::set_error("linear_decode_rmsnorm_silu_mul: norm_weight is required"); return -1; }
    return linear_decode_entry(x, w_gate_up, y, batch, in_features, 2 * intermediate, x_row_stride, w_row_stride, y_row_stride, 2, nullptr, 0,
                               dtype, stream, norm_weight, eps, xn_scratch);
}
extern "C" int atoma_linear_decode_silu_mul(const void *x, const void *w_gate_up, void *y, int64_t batch, int64_t in_features,
                                            int64_t intermediate, int64_t x_row_stride, int64_t w_row_stride, int64_t y_row_stride, int dtype,
                                            void *stream) {
    return linear_decode_entry(x, w_gate_up, y, batch, in_features, 2 * intermediate, x_row_stride, w_row_stride, y_row_stride, 2, nullptr, 0,
                               dtype, stream);
}
