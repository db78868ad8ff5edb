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

// 16-byte streaming store for 16-bit outputs (one full 1 KiB run per wave instruction; wider outputs take two
// or four instructions that each cover half / a quarter of every line, and measured slower with write-through, so
// they stay non-temporal).  Output tensors are written once and never re-read by the launch; on gfx950 a
// write-through store (`sc0 sc1`) drains to HBM as it is issued and measured 4-5 % faster for the whole
// 4-read/1-write step kernel than a non-temporal (`nt`) or plain (write-back) store, which leave the lines to be
// evicted from L2 in bursts (tools/tune/tune_policy.hip: 27.4 vs 28.8 us at 256x4x128x128 bf16).  The compiler has
// no spelling for that policy on a plain vector store, hence the inline assembly.  Two things the compiler would
// otherwise do for us: (1) gfx940+ needs 2 wait states between a >8-byte VMEM store and a VALU write to its data
// VGPRs (LLVM's "12-dword store" hazard; the hazard recogniser cannot see into inline asm) -- the `s_nop 1` travels
// with the store; (2) vmcnt bookkeeping -- an outstanding store the compiler does not know about only makes its
// waits for later loads more conservative, never less (loads return in order).
template <typename V>
__device__ __forceinline__ void store16_stream(V* p, V v) {
  static_assert(sizeof(V) == 16, "one dwordx4");
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <typename V>
__device__ __forceinline__ void store8_stream(V* p, V v) {  // 8-byte form (no wide-store hazard below 12 bytes)
  static_assert(sizeof(V) == 8, "one dwordx2");
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

template <typename T>
__device__ __forceinline__ uint32_t pack_pair(float a, float b) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) {
    pk_f32x2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, pk_bf16x2));  // v_cvt_pk_bf16_f32
  } else {
    asm("" : "+v"(a), "+v"(b));  // convert the rounded fp32 value (no fused multiply-convert)
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
    store16_stream(reinterpret_cast<pk_u32x4*>(base) + vec, q);
  } else if constexpr (sizeof(T) == 4) {
    pk_f32x4* p = reinterpret_cast<pk_f32x4*>(base) + vec * 2;
    pk_f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    p[0] = a;  // 32 B per lane = two half-covered lines per instruction: plain write-back stores, L2 merges the halves
    p[1] = b;
  } else {
    pk_f64x2* p = reinterpret_cast<pk_f64x2*>(base) + vec * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pk_f64x2 a = {(double)v[2 * i], (double)v[2 * i + 1]};
      __builtin_nontemporal_store(a, p + i);
    }
  }
}

// 4 consecutive values at `dst` (8-byte aligned for 16-bit T, 16-byte for float): one streaming store
template <typename T>
__device__ __forceinline__ void store4_from_f32(T* dst, float a, float b, float c, float d) {
  if constexpr (sizeof(T) == 2) {
    typedef uint32_t pk_u32x2 __attribute__((ext_vector_type(2)));
    store8_stream(reinterpret_cast<pk_u32x2*>(dst), pk_u32x2{pack_pair<T>(a, b), pack_pair<T>(c, d)});
  } else if constexpr (sizeof(T) == 4) {
    store16_stream(reinterpret_cast<pk_f32x4*>(dst), pk_f32x4{a, b, c, d});
  } else {
    dst[0] = (T)a; dst[1] = (T)b; dst[2] = (T)c; dst[3] = (T)d;
  }
}

}  // namespace skr
