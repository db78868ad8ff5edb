// Shared device helpers of the fused step kernels (skr_step.hip: general + grid-stride kernels; skr_step_fast.hip:
// one-trip compile-time kernels): lane ownership, loads / stores, rounded conversion, tuning switches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "../../include/skrample_hip.h"
#include "skr_philox.h"
#include "skr_pack.h"

namespace skr {

// tuning switches (defaults = the measured best; initialised from the environment, changed with skr_set_tuning)
struct Tuning {
  int one_trip, xmap, tile, rk_uv, two_out, pace, rk_blk, two_nt, tape_words;
  Tuning() {
    const char* e;
    one_trip = !((e = getenv("SKR_ONE_TRIP")) && e[0] == '0');
    xmap = (e = getenv("SKR_XMAP")) ? atoi(e) : 7;
    tile = getenv("SKR_NO_TILE") == nullptr;
    rk_uv = (e = getenv("SKR_RK_UV")) ? atoi(e) : 0;
    two_out = getenv("SKR_NO_TWO_OUT") == nullptr;
    pace = getenv("SKR_NO_PACE") == nullptr;
    two_nt = (e = getenv("SKR_TWO_NT")) ? atoi(e) : -1;  // two-output launches without in-kernel noise: non-temporal stores (1), write-through (0), by operand count (-1)
    rk_blk = (e = getenv("SKR_RK_BLK")) ? atoi(e) : 0;  // threads per workgroup of the one-trip Runge-Kutta stage kernel: 0 = by operand count, 128, 256
    tape_words = (e = getenv("SKR_TAPE_WORDS")) ? atoi(e) : 0;  // 16-byte words per lane and trip of the op-tape kernel: 0 = by tensor size, 1, 2
  }
};
extern Tuning g_tune;  // defined in skr_step.hip

constexpr int VEC = 8;       // elements per lane per trip
constexpr int BLOCK = 256;   // 4 waves
// vectors per lane per trip (spaced BLOCK apart so every wave access stays 1 KiB contiguous).  Measured on
// MI355X (tools/tune/tune_step.hip, B=256 DPM-2): without Philox more bytes in flight per lane win
// (UV 1/2/4 -> 29.3/28.2/27.4 us with non-temporal stores; 26.9/26.4/26.7 with the write-through stores used now);
// with Philox one vector per lane and one trip per lane is best (UV 1/2/4 -> 26.4/27.1/28.0 us): the VALU work then
// overlaps other waves' loads instead of its own.
constexpr int uv_for(bool noise, bool has1) { return noise ? 1 : (has1 ? 2 : 4); }
constexpr int MAXK = SKR_MAX_TERMS;

struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

template <typename Acc>
struct StepArgs {
  const void* in[MAXK];
  Acc c0[MAXK];
  Acc c1[MAXK];
  void* out0;
  void* out1;
  const uint64_t* seeds;
  Acc chain, zeta0, zeta1;
  uint64_t stream0, stream1;
  int64_t numel;
  int64_t sample_numel;
  double inv_sample_numel;
  int64_t vps;          // vectors per sample (per-sample grid only)
  int32_t n_a, n_terms;
  int32_t grid_mode;    // 0 flat grid, 1 per-sample grid (noise kernels, sample_numel % 8 == 0)
  int32_t conv_to, conv_from;  // rounded pair conversion (CONV kernels): see convert_rounded()
  double ck[4];
  // skr_step_launch_indexed: the scalars above are read from rows[index[0] + row_offset] by the kernel (one-trip kernels only)
  const skr_step_row* rows;
  const int32_t* index;
  int32_t row_offset;
};

// device-resident scalars of a launch (skr_step_launch_indexed); rows == nullptr: use the kernarg values
struct RowRef {
  const skr_step_row* rows;
  const int32_t* index;
  int32_t row_offset;
};
__device__ __forceinline__ const skr_step_row* row_of(const RowRef& r) {  // (only called by the TAB instantiations: rows != nullptr)
  return r.rows + ((r.index != nullptr ? r.index[0] : 0) + r.row_offset);
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- 8-element loads (widening to Acc) ---------------------------------------------------------
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

template <typename T> struct Raw;  // raw register image of 8 elements
template <> struct Raw<bf16_t> { u32x4_t q; };
template <> struct Raw<f16_t> { u32x4_t q; };
template <> struct Raw<float> { f32x4_t q[2]; };
template <> struct Raw<double> { f64x2_t q[4]; };

// Which 8 elements lane-slot `v` owns, as two groups of 4 consecutive elements (group index = element / 4):
//   TILE = false  the 8 consecutive elements 8v .. 8v+7            -> groups 2v, 2v+1
//   TILE = true   within the wave's 512-element tile, elements 4l..4l+3 and 256+4l..256+4l+3 (l = lane)
//                 -> groups 128*(v>>6) + l and that + 64
// With 8 consecutive elements a 16-bit operand is one 16-byte access per lane (a full KiB per wave instruction), but a
// 32-bit operand is two 16-byte accesses to the lane's own 32 bytes: each wave instruction then covers only HALF of
// every line it touches, which costs ~20 % on loads and far more on write-through stores.  The tile layout makes
// every wave instruction cover whole lines for 16- and 32-bit operands alike (16-bit: two 8-byte accesses, 512 B
// each; 32-bit: two 16-byte accesses, 1 KiB each).  Used whenever a 32-bit tensor takes part and the launch is made of
// whole tiles (tools/tune/tune_policy.hip: 2 bf16 + 3 fp32 in, fp32 + bf16 out: 66.9 -> 54.7 us).
template <bool TILE> __device__ __forceinline__ int64_t group0(int64_t v) {
  if constexpr (TILE) return ((v >> 6) << 7) + (v & 63);
  else return 2 * v;
}
template <bool TILE> __device__ __forceinline__ int64_t group1(int64_t v) {
  if constexpr (TILE) return ((v >> 6) << 7) + (v & 63) + 64;
  else return 2 * v + 1;
}
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

template <typename T, bool TILE = false>
__device__ __forceinline__ Raw<T> load_raw(const void* base, int64_t vec) {
  Raw<T> r;
  if constexpr (sizeof(T) == 2) {
    if constexpr (TILE) {
      const u32x2_t* p = reinterpret_cast<const u32x2_t*>(base);
      const u32x2_t lo = __builtin_nontemporal_load(p + group0<true>(vec)), hi = __builtin_nontemporal_load(p + group1<true>(vec));
      r.q = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
    } else {
      r.q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(base) + vec);
    }
  } else if constexpr (sizeof(T) == 4) {
    const f32x4_t* p = reinterpret_cast<const f32x4_t*>(base);
    r.q[0] = __builtin_nontemporal_load(p + group0<TILE>(vec));
    r.q[1] = __builtin_nontemporal_load(p + group1<TILE>(vec));
  } else {
    const f64x2_t* p = reinterpret_cast<const f64x2_t*>(base);
    const int64_t g0 = group0<TILE>(vec) * 2, g1 = group1<TILE>(vec) * 2;
    r.q[0] = __builtin_nontemporal_load(p + g0);
    r.q[1] = __builtin_nontemporal_load(p + g0 + 1);
    r.q[2] = __builtin_nontemporal_load(p + g1);
    r.q[3] = __builtin_nontemporal_load(p + g1 + 1);
  }
  return r;
}

template <typename T>
__device__ __forceinline__ void pin_raw(Raw<T>& r) {  // "the loaded registers are consumed here": nothing that reads them moves above
  if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(r.q));
  else if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]));
  else asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]));
}

template <typename T, typename Acc>
__device__ __forceinline__ void widen(const Raw<T>& r, Acc v[VEC]) {
  if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = (Acc)__uint_as_float(r.q[i] << 16);
      v[2 * i + 1] = (Acc)__uint_as_float(r.q[i] & 0xFFFF0000u);
    }
  } else if constexpr (std::is_same<T, f16_t>::value) {
    // (bit_cast of a dword to a _Float16x2 vector is mis-compiled by ROCm 7.2 hipcc for lanes 1..3 of a
    //  dwordx4: go through scalar 16-bit halves instead)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = r.q[i];
      v[2 * i] = (Acc)(float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu));
      v[2 * i + 1] = (Acc)(float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    }
  } else if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (Acc)r.q[i >> 2][i & 3];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (Acc)r.q[i >> 1][i & 1];
  }
}

template <typename T>
__device__ __forceinline__ float load_scalar(const void* base, int64_t i) {
  if constexpr (std::is_same<T, bf16_t>::value)
    return __uint_as_float((uint32_t) reinterpret_cast<const uint16_t*>(base)[i] << 16);
  else if constexpr (std::is_same<T, f16_t>::value)
    return (float)reinterpret_cast<const _Float16*>(base)[i];
  else
    return (float)reinterpret_cast<const T*>(base)[i];
}
template <typename T>
__device__ __forceinline__ double load_scalar_d(const void* base, int64_t i) {
  if constexpr (std::is_same<T, double>::value) return reinterpret_cast<const double*>(base)[i];
  else return (double)load_scalar<T>(base, i);
}

// ---- stores (single rounding from Acc) --------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2_t f = {a, b};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  // round the fp32 accumulator, as the reference does (compute in fp32, then .to(half)): without the pin the compiler
  // may fuse the last FMA with the conversion (v_fma_mixlo_f16), which rounds the exact sum once
  asm("" : "+v"(a), "+v"(b));
  const uint32_t lo = __builtin_bit_cast(uint16_t, (_Float16)a);
  const uint32_t hi = __builtin_bit_cast(uint16_t, (_Float16)b);
  return lo | (hi << 16);
}

// NT (tile layout only): non-temporal instead of write-through stores -- what tools/tune/tune_r3.hip measured 1.5 % faster for the
// two-output launch with 10 + 1 operands and no arithmetic besides (km<..spb1,spf1>: 336.1 vs 341.2 us), the one traffic mix
// where write-through did not win or tie
template <typename T, typename Acc, bool TILE = false, bool NT = false>
__device__ __forceinline__ void store8(void* base, int64_t vec, const Acc v[VEC]) {
  if constexpr (sizeof(T) == 2) {
    u32x4_t q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (std::is_same<T, bf16_t>::value) q[i] = pack_bf16((float)v[2 * i], (float)v[2 * i + 1]);
      else q[i] = pack_f16((float)v[2 * i], (float)v[2 * i + 1]);
    }
    if constexpr (TILE && NT) {
      u32x2_t* p = reinterpret_cast<u32x2_t*>(base);
      __builtin_nontemporal_store(u32x2_t{q[0], q[1]}, p + group0<true>(vec));
      __builtin_nontemporal_store(u32x2_t{q[2], q[3]}, p + group1<true>(vec));
    } else if constexpr (TILE) {
      u32x2_t* p = reinterpret_cast<u32x2_t*>(base);
      store8_stream(p + group0<true>(vec), u32x2_t{q[0], q[1]});
      store8_stream(p + group1<true>(vec), u32x2_t{q[2], q[3]});
    } else {
      store16_stream(reinterpret_cast<u32x4_t*>(base) + vec, q);
    }
  } else if constexpr (sizeof(T) == 4) {
    f32x4_t* p = reinterpret_cast<f32x4_t*>(base);
    f32x4_t a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    f32x4_t b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    if constexpr (TILE && NT) {
      __builtin_nontemporal_store(a, p + group0<true>(vec));
      __builtin_nontemporal_store(b, p + group1<true>(vec));
    } else if constexpr (TILE) {  // whole lines per wave instruction: write through
      store16_stream(p + group0<true>(vec), a);
      store16_stream(p + group1<true>(vec), b);
    } else {  // 32 B per lane = two half-covered lines per instruction: plain write-back stores, so L2 merges the halves
      p[2 * vec] = a;  // (measured: plain 33.5 us, non-temporal 41.0 us, write-through 47.2 us for 4 bf16 in -> fp32 out)
      p[2 * vec + 1] = b;
    }
  } else {
    f64x2_t* p = reinterpret_cast<f64x2_t*>(base);
    const int64_t g[2] = {group0<TILE>(vec) * 2, group1<TILE>(vec) * 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f64x2_t a = {(double)v[2 * i], (double)v[2 * i + 1]};
      __builtin_nontemporal_store(a, p + g[i >> 1] + (i & 1));
    }
  }
}

template <typename T, typename Acc>
__device__ __forceinline__ void store_scalar(void* base, int64_t i, Acc v) {
  if constexpr (std::is_same<T, bf16_t>::value) reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)(pack_bf16((float)v, 0.f) & 0xFFFFu);
  else if constexpr (std::is_same<T, f16_t>::value) reinterpret_cast<_Float16*>(base)[i] = (_Float16)(float)v;
  else reinterpret_cast<T*>(base)[i] = (T)v;
}

// ---- rounded pair conversion (Runge-Kutta wrapper) ---------------------------------------------------
// The reference's RK wrapper converts the network output to derivative space in the INPUT dtype,
// one rounded tensor op at a time, before any cast to compute_scale (skrample/diffusers.py:819-834 with
// models.py:92-224).  out0 = from_x(s, to_x(s, o)) is reproduced here op for op with the same roundings,
// so the stored derivative tensor is bit-identical to the reference's.
//   to_x   kinds: 0 o | 1 ((s - k0*o) / k1) | 2 (k1*s - k0*o) | 3 (o * k0)
//   from_x kinds: 0 x | 1 ((s - k2*x) / k3) | 2 ((k2*s - x) / k3) | 3 (x / k2)
template <typename T> struct OpMath { using type = float; };
template <> struct OpMath<double> { using type = double; };

template <typename T> __device__ __forceinline__ float rnd(float v) {
  if constexpr (std::is_same<T, bf16_t>::value) return __uint_as_float(pack_bf16(v, 0.f) << 16);
  else if constexpr (std::is_same<T, f16_t>::value) {
    // The reference rounds twice (fp32 op result, then to half).  Left alone, the compiler folds `half(k * float(h))`
    // into v_fma_mixlo_f16, which rounds the exact product ONCE and differs from torch on fp32 ties (seen as 1-ulp
    // derivative mismatches in ~4 % of elements).  The empty asm pins the fp32 result in a VGPR first.
    asm("" : "+v"(v));
    return (float)(_Float16)v;
  } else return v;
}
__device__ __forceinline__ double rnd_d(double v) { return v; }

// individually rounded ops: hip's __fmul_rn / __fsub_rn are plain operators that the compiler may still contract
// into an FMA with a neighbour, so contraction is switched off inside these helpers
__device__ __forceinline__ float mul_(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float sub_(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float div_(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ double mul_(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double sub_(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double div_(double a, double b) { return __ddiv_rn(a, b); }

template <typename T, typename M>
__device__ __forceinline__ M convert_rounded(M s, M o, int to_kind, int from_kind, const M k[4]) {
  auto R = [](M v) -> M { if constexpr (std::is_same<M, double>::value) return v; else return rnd<T>(v); };
  M x;
  switch (to_kind) {
    case 1: x = R(div_(R(sub_(s, R(mul_(k[0], o)))), k[1])); break;
    case 2: x = R(sub_(R(mul_(k[1], s)), R(mul_(k[0], o)))); break;
    case 3: x = R(mul_(o, k[0])); break;
    default: x = o; break;
  }
  switch (from_kind) {
    case 1: return R(div_(R(sub_(s, R(mul_(k[2], x)))), k[3]));
    case 2: return R(div_(R(sub_(R(mul_(k[2], s)), x)), k[3]));
    case 3: return R(div_(x, k[2]));
    default: return x;
  }
}

// ---- noise ---------------------------------------------------------------------------------------
// element e of the whole tensor -> sample s = e / sample_numel, r = e % sample_numel,
// Philox block r >> 2, lane r & 3 (oracle/skr_oracle/noise.py::philox_normal).
template <typename Acc>
__device__ __forceinline__ void fma_noise8(Acc zeta, const float z[VEC], Acc s[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = fma_(zeta, (Acc)z[i], s[i]);
}

// kernarg of the Runge-Kutta stage kernels (grid-stride and one-trip)
struct RkArgs {
  const void* in[8];
  float c1[8];
  void* out0;
  void* out1;
  float chain;
  float ck[4];
  int32_t conv_to, conv_from;
  int32_t xmap_lr;
  int64_t numel;
  RowRef tab;
  // stochastic final stage (one-trip kernel only): out1 += zeta1 * N(stream1)
  const uint64_t* seeds;
  float zeta1;
  int32_t bps_shift;
  uint64_t stream1;
};

extern thread_local int g_last_hip_error;
int finish_launch();

// one-trip launches (skr_step_fast.hip); `taken` = false when the plan is outside what they cover
template <typename T> int launch_one_trip_k(const StepArgs<float>& args, bool noise, hipStream_t stream, bool& taken);
template <typename T> int launch_one_trip_rk(const StepArgs<float>& args, bool noise, hipStream_t stream, bool& taken);
template <typename TA> int launch_one_trip_two(const StepArgs<float>& args, bool noise, bool group_b_f32, hipStream_t stream, bool& taken);

}  // namespace skr
