// Shared helpers of libatoma_hip (gfx950 only): error slot, launch check, dtype bit tricks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/atoma_hip.h"

namespace atoma {

// Thread-local error slot behind atoma_last_error() (the reference calls exit() instead:
// /root/reference/csrc/kernels/flash_fwd_launch_template.h:25-31).
void set_error(const std::string &msg);
void clear_error();
bool has_error();

inline bool check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    fprintf(stderr, "[atoma_hip] %s: %s\n", what, hipGetErrorString(e));
    return false;
}
#define ATOMA_CHECK_LAUNCH(what) ::atoma::check_hip(hipGetLastError(), what)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

int device_num_cus();  // CUs of the current device (cached per device)

// ---- device-side 16-bit float helpers --------------------------------------------------
struct bf16_t {};  // tags: storage is always uint16_t bit patterns
struct f16_t {};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_v;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_v;

template <typename T> __device__ __forceinline__ float lo_to_f32(uint32_t packed);
template <typename T> __device__ __forceinline__ float hi_to_f32(uint32_t packed);
template <> __device__ __forceinline__ float lo_to_f32<bf16_t>(uint32_t p) { return __uint_as_float(p << 16); }
template <> __device__ __forceinline__ float hi_to_f32<bf16_t>(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
template <> __device__ __forceinline__ float lo_to_f32<f16_t>(uint32_t p) {
    return (float)__builtin_bit_cast(f16x2_v, p)[0];
}
template <> __device__ __forceinline__ float hi_to_f32<f16_t>(uint32_t p) {
    return (float)__builtin_bit_cast(f16x2_v, p)[1];
}

// round-to-nearest-even f32 -> 16-bit storage
template <typename T> __device__ __forceinline__ uint32_t f32_to_bits(float f);
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round to nearest even; a NaN stays a quiet NaN): one instruction for two values where the
// integer sequence (NaN test, bias, carry, shift) took six per value -- round 6: the RoPE epilogue of the 256-row q/k/v projection spent
// ~4 us of a 38 us launch in those sequences (10 roundings per rotated pair)
typedef __attribute__((ext_vector_type(2))) float f32x2_v;
template <> __device__ __forceinline__ uint32_t f32_to_bits<bf16_t>(float f) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_v{f, 0.f}, bf16x2_v)) & 0xffffu;
}
template <> __device__ __forceinline__ uint32_t f32_to_bits<f16_t>(float f) {
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32, RNE
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return f32_to_bits<T>(lo) | (f32_to_bits<T>(hi) << 16);
}
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_v{lo, hi}, bf16x2_v));
}
template <typename T> __device__ __forceinline__ float round_through(float f) {
    return lo_to_f32<T>(f32_to_bits<T>(f));
}
template <> __device__ __forceinline__ float round_through<bf16_t>(float f) {
    return __uint_as_float(__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_v{f, 0.f}, bf16x2_v)) << 16);
}

// acc += a.lo*b.lo + a.hi*b.hi on packed 16-bit pairs: v_dot2c_f32_bf16 / v_dot2c_f32_f16
template <typename T> __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc);
template <> __device__ __forceinline__ float dot2<bf16_t>(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, a), __builtin_bit_cast(bf16x2_v, b), acc, false);
}
template <> __device__ __forceinline__ float dot2<f16_t>(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_v, a), __builtin_bit_cast(f16x2_v, b), acc, false);
}

}  // namespace atoma
