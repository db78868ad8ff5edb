// 8-element packed, non-temporal stores from fp32 registers for the noise generators (one rounding, RNE).
// 16-bit halves are packed as scalars: hipcc 7.2 mis-compiles a dword -> _Float16x2 vector bit_cast.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skr {

typedef uint32_t pk_u32x4 __attribute__((ext_vector_type(4)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));
typedef double pk_f64x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ uint32_t pack_pair(float a, float b) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) {
    pk_f32x2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, pk_bf16x2));  // v_cvt_pk_bf16_f32
  } else {
    const uint32_t lo = __builtin_bit_cast(uint16_t, (_Float16)a);
    const uint32_t hi = __builtin_bit_cast(uint16_t, (_Float16)b);
    return lo | (hi << 16);
  }
}

// element index of the first value = 8 * vec; `base` must be 16-byte aligned at that element
template <typename T>
__device__ __forceinline__ void store8_from_f32(T* base, int64_t vec, const float v[8]) {
  if constexpr (sizeof(T) == 2) {
    pk_u32x4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_pair<T>(v[2 * i], v[2 * i + 1]);
    __builtin_nontemporal_store(q, reinterpret_cast<pk_u32x4*>(base) + vec);
  } else if constexpr (sizeof(T) == 4) {
    pk_f32x4* p = reinterpret_cast<pk_f32x4*>(base) + vec * 2;
    pk_f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    __builtin_nontemporal_store(a, p);
    __builtin_nontemporal_store(b, p + 1);
  } else {
    pk_f64x2* p = reinterpret_cast<pk_f64x2*>(base) + vec * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pk_f64x2 a = {(double)v[2 * i], (double)v[2 * i + 1]};
      __builtin_nontemporal_store(a, p + i);
    }
  }
}

}  // namespace skr
