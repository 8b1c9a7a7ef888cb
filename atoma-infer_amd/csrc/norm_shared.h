// The arithmetic of rms_norm_kernel (norm_rope.hip), in one place, so that a kernel which normalises its own input
// (linear_gemv_kernel<.., NORM>, linear_decode.hip) produces the bits the separate kernel would have produced:
// same products, same order of the sums, same reduction tree, same rounding points.
#pragma once
#include "common.h"
#include <math.h>

namespace atoma {

constexpr int NORM_THREADS = 256;   // rms_norm_kernel: one workgroup of 4 wavefronts per row, thread t owns the 16-byte vectors t, t + 256, ..

template <typename T> __device__ __forceinline__ void unpack8(const uint4 &v, float (&f)[8]) {
    f[0] = lo_to_f32<T>(v.x); f[1] = hi_to_f32<T>(v.x);
    f[2] = lo_to_f32<T>(v.y); f[3] = hi_to_f32<T>(v.y);
    f[4] = lo_to_f32<T>(v.z); f[5] = hi_to_f32<T>(v.z);
    f[6] = lo_to_f32<T>(v.w); f[7] = hi_to_f32<T>(v.w);
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack2<T>(f[0], f[1]); v.y = pack2<T>(f[2], f[3]);
    v.z = pack2<T>(f[4], f[5]); v.w = pack2<T>(f[6], f[7]);
    return v;
}

// ss += the 8 squares of one vector, in element order (explicit fma: the same instruction in every translation unit)
template <typename T> __device__ __forceinline__ float norm_sumsq8(const uint4 &v, float ss) {
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(f[e], f[e], ss);
    return ss;
}
// the butterfly over the 64 lanes of a wavefront (every lane ends with the same bits: each step adds the same two numbers)
__device__ __forceinline__ float norm_wave_sum(float ss) {
#pragma unroll
    for (int off = 32; off; off >>= 1) ss += __shfl_xor(ss, off, 64);
    return ss;
}
// the workgroup total from the 4 wavefront sums, in wavefront order, and the scale
__device__ __forceinline__ float norm_scale(const float (&red)[NORM_THREADS / 64], int hidden, float eps) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_THREADS / 64; ++i) tot += red[i];
    return 1.0f / sqrtf(tot / (float)hidden + eps);
}
// y = round((scale * x) * w), candle-kernels' rmsnorm order
template <typename T> __device__ __forceinline__ uint4 norm_apply8(const uint4 &xv, const uint4 &wv, float scale) {
    float f[8], g[8];
    unpack8<T>(xv, f);
    unpack8<T>(wv, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (scale * f[e]) * g[e];
    return pack8<T>(f);
}

// What rms_norm_kernel computes as `scale`, for NB rows at once, evaluated by ONE wavefront (every lane gets all of them): the lane
// plays thread 64 g + lane of the norm workgroup for g = 0..3.  The NB x 4 chains are independent, so their loads go out together
// and their butterflies advance in step (one round trip per 2048 elements of a row instead of one per chain).  hidden % 8 == 0,
// 16-byte aligned rows; rows beyond `batch` repeat the last one.
template <typename T, int NB>
__device__ __forceinline__ void norm_scales_by_one_wave(const uint16_t *x, int64_t x_row_stride, int batch, int hidden, float eps, int lane,
                                                        float (&scale)[NB]) {
    constexpr int G = NORM_THREADS / 64;
    const int nvec = hidden >> 3;
    float ss[NB][G];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < G; ++g) ss[b][g] = 0.f;
    for (int i0 = 0; i0 < nvec; i0 += NORM_THREADS) {
        uint4 v[NB][G];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i = i0 + g * 64 + lane;
                v[b][g] = i < nvec ? reinterpret_cast<const uint4 *>(x + (int64_t)(b < batch ? b : batch - 1) * x_row_stride)[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < G; ++g) ss[b][g] = norm_sumsq8<T>(v[b][g], ss[b][g]);   // a zero vector adds +0.0: the sum is unchanged
    }
#pragma unroll
    for (int off = 32; off; off >>= 1)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < G; ++g) ss[b][g] += __shfl_xor(ss[b][g], off, 64);
#pragma unroll
    for (int b = 0; b < NB; ++b) scale[b] = norm_scale(ss[b], hidden, eps);
}

}  // namespace atoma
