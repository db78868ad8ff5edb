// Op-tape kernel: a solver step evaluated ONE ROUNDED TENSOR OPERATION AT A TIME, in one pass over HBM.
//
// The reference's samplers are generic over the sample type; called directly on bf16 / fp16 tensors (no scheduler wrapper, or a wrapper
// with compute_scale=None) every `*`, `+`, `-`, `/` of the step is its own torch op in the TENSOR dtype: the operands are widened to the
// op-math type (fp32 for 16- and 32-bit tensors, fp64 for fp64; a Python scalar is converted to that type first), combined once, and the
// result is rounded back to the tensor dtype (skrample/sampling/structured.py:167-497, models.py:53-224).  The fused step kernel
// (skr_step.hip) evaluates the collapsed linear form in fp32 and rounds once, which is closer to the exact result but NOT what the
// reference returns (4-47 last-place units apart on 16-bit tensors, tests/golden/native16.npz).  This kernel replays the reference's own
// sequence instead: the host (skrample_amd/sampling/native.py) records the step's operations on a tape, and every lane runs the tape
// over its four elements with the values held in registers -- each input tensor is read once, each result written once, nothing else
// touches memory, and the bits are the reference's.
//
// Registers: the tape's SKR_TAPE_REGS values live in LDS, one column per thread ([register][thread]: consecutive lanes, consecutive words --
// no bank conflict, no barrier: a thread only ever touches its own column), IN THE TENSOR DTYPE.  That is exact, not a shortcut: every value
// a tape defines is the result of an op rounded to the tensor dtype, so a 16-bit register is two bytes -- a thread's eight elements are one
// 16-byte word, LOAD and STORE are plain copies between memory and the file, and an op is two 16-byte LDS reads, eight lane operations,
// four rounding packs and one LDS write.  (Round 5's first version kept fp32 registers in VGPRs behind s_set_gpr_idx: the compiler copied the
// 64-register file at every indexed write, 192 v_mov_b64 per trip, and waited for every load on its own -- 414 us for a 21-op DPM-2 tape
// over 100 MB, see profiles/r05_bench_tape.txt.)  The tape itself sits in the kernel argument block (scalar loads).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skr_device.h"
#include "skr_step_common.h"

namespace skr {

// elements per lane and trip: eight 16-bit values (16-byte accesses, two independent chains of four per op decode), four fp32 (16 bytes), four fp64 (32)
template <typename T> struct TapeElems { static constexpr int value = sizeof(T) == 2 ? 8 : 4; };
constexpr int TAPE_THREADS = 256;

struct TapeArgs {
  skr_tape tape;
  const void* in[SKR_TAPE_MAX_INPUTS];
  void* out[SKR_TAPE_MAX_OUTPUTS];
  int64_t numel;
};

// what one register holds for a thread's four elements, and its memory image
template <typename T> struct TapeWord;
template <> struct TapeWord<bf16_t> { typedef u32x4_t type; };
template <> struct TapeWord<f16_t> { typedef u32x4_t type; };
template <> struct TapeWord<float> { typedef f32x4_t type; };
struct f64x4_pack { f64x2_t lo, hi; };
template <> struct TapeWord<double> { typedef f64x4_pack type; };

template <typename T, typename M> __device__ __forceinline__ void tape_unpack(const typename TapeWord<T>::type& w, M (&x)[TapeElems<T>::value]) {
  if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[2 * j] = __uint_as_float(w[j] << 16); x[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
  } else if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[2 * j] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] & 0xFFFFu));
      x[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] >> 16));
    }
  } else if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = w[i];
  } else {
    x[0] = w.lo[0]; x[1] = w.lo[1]; x[2] = w.hi[0]; x[3] = w.hi[1];
  }
}
// the op results (op-math type) rounded to the tensor dtype: the register's new contents
template <typename T, typename M> __device__ __forceinline__ typename TapeWord<T>::type tape_pack(const M (&y)[TapeElems<T>::value]) {
  typename TapeWord<T>::type w;
  if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = pack_bf16(y[2 * j], y[2 * j + 1]);
  } else if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = pack_f16(y[2 * j], y[2 * j + 1]);  // (pack_f16 pins the fp32 result first: two roundings, as torch's)
  } else if constexpr (std::is_same<T, float>::value) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = y[i];
  } else { w.lo = f64x2_t{y[0], y[1]}; w.hi = f64x2_t{y[2], y[3]}; }
  return w;
}

// The arithmetic of one op.  Contraction is off inside mul_ / sub_ (skr_step_common.h); every op stands alone here, so a plain `+`
// under `fp contract(off)` is enough for the additions.
template <typename M> __device__ __forceinline__ M tape_add(M a, M b) {
#pragma clang fp contract(off)
  return a + b;
}

// an op result on its way into another op of the same visit: rounded to the tensor dtype, kept in the op-math type
template <typename T, typename M> __device__ __forceinline__ M tape_round(M v) {
  if constexpr (std::is_same<M, double>::value || std::is_same<T, float>::value) return v;
  else return rnd<T>(v);
}

// one op over a thread's NW words; each case of the dispatch below is a complete read - compute - round - write of its own, so that no value has to be
// merged across cases (the merged form cost the compiler a register copy per element and a flag per case, and this kernel lives on instruction issue)
template <typename T, typename M, int NW, typename F>
__device__ __forceinline__ void tape_binary(typename TapeWord<T>::type* file, int dst, int ia, int ib, F f) {
  constexpr int WE = TapeElems<T>::value;
#pragma unroll
  for (int h = 0; h < NW; ++h) {
    M x[WE], z[WE], y[WE];
    tape_unpack<T, M>(file[(ia * NW + h) * TAPE_THREADS], x);
    tape_unpack<T, M>(file[(ib * NW + h) * TAPE_THREADS], z);
#pragma unroll
    for (int i = 0; i < WE; ++i) y[i] = f(x[i], z[i]);
    file[(dst * NW + h) * TAPE_THREADS] = tape_pack<T, M>(y);
  }
}
template <typename T, typename M, int NW, typename F>
__device__ __forceinline__ void tape_scalar(typename TapeWord<T>::type* file, int dst, int ia, M k, F f) {
  constexpr int WE = TapeElems<T>::value;
#pragma unroll
  for (int h = 0; h < NW; ++h) {
    M x[WE], y[WE];
    tape_unpack<T, M>(file[(ia * NW + h) * TAPE_THREADS], x);
#pragma unroll
    for (int i = 0; i < WE; ++i) y[i] = f(x[i], k);
    file[(dst * NW + h) * TAPE_THREADS] = tape_pack<T, M>(y);
  }
}

// NW 16-byte (fp64: 32-byte) words per lane and trip: the decode of an op -- scalar loads, compares, branches, on the CU's one scalar unit -- is paid
// per wave and op whatever the wave then computes, and with one word per lane it was what bounded the kernel.  Word h of a trip's chunk sits a whole
// block apart from word h - 1 (consecutive lanes, consecutive words: every access coalesced).
template <typename T, typename M, int NW>
__global__ __launch_bounds__(TAPE_THREADS) void tape_kernel(const TapeArgs a) {
  typedef typename TapeWord<T>::type Word;
  constexpr int WE = TapeElems<T>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char tape_lds[];
  Word* const file = reinterpret_cast<Word*>(tape_lds) + threadIdx.x;  // word h of register g of this thread: file[(g * NW + h) * TAPE_THREADS]
  const int n_ops = a.tape.n_ops;
  const int64_t n_words = (a.numel + WE - 1) / WE;
  const int64_t n_chunks = (n_words + (int64_t)TAPE_THREADS * NW - 1) / ((int64_t)TAPE_THREADS * NW);
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    int64_t w[NW];
    bool whole[NW], some[NW];
#pragma unroll
    for (int h = 0; h < NW; ++h) {
      w[h] = (c * NW + h) * TAPE_THREADS + threadIdx.x;
      whole[h] = (w[h] + 1) * WE <= a.numel;
      some[h] = w[h] * WE < a.numel;
    }
    skr_tape_op ahead = a.tape.ops[0];  // uniform: scalar loads from the kernel argument block, one op ahead of its use (the load's latency
    for (int o = 0; o < n_ops; ++o) {   // then runs under the current op's LDS round trip instead of in front of it)
      const skr_tape_op op = ahead;
      ahead = a.tape.ops[o + 1 < SKR_TAPE_MAX_OPS ? o + 1 : o];
      const int code = op.code, dst = op.dst, ia = op.a, ib = op.b;
      switch (code) {
      case SKR_TAPE_LOAD: {
        // A run of consecutive LOADs (the recorder opens a tape with its leaves) is issued as ONE batch: every global load first, the LDS
        // writes behind them -- one memory latency per run instead of one per input.
        constexpr int RUN = (sizeof(T) == 8 ? 2 : (sizeof(T) == 4 ? 4 : 6)) / (NW > 1 ? 2 : 1) + (NW > 1 && sizeof(T) != 8 ? 1 : 0);  // (batch registers: 8 / 16 / 24 with one word, 16 / 24 / 32 with two)
        int run = 1;
        while (run < RUN && o + run < n_ops && a.tape.ops[o + run].code == SKR_TAPE_LOAD) ++run;
        Word q[RUN][NW];
#pragma unroll
        for (int j = 0; j < RUN; ++j) {
          if (j < run) {
            const void* src = a.in[a.tape.ops[o + j].a];
#pragma unroll
            for (int h = 0; h < NW; ++h) {
              if (whole[h]) q[j][h] = *(reinterpret_cast<const Word*>(src) + w[h]);
              else {
                M x[WE];
#pragma unroll
                for (int i = 0; i < WE; ++i) {
                  const int64_t e = w[h] * WE + i;
                  if constexpr (std::is_same<M, double>::value) x[i] = e < a.numel ? load_scalar_d<T>(src, e) : 0.0;
                  else x[i] = e < a.numel ? load_scalar<T>(src, e) : 0.f;
                }
                q[j][h] = tape_pack<T, M>(x);  // (exact: the values come from the tensor dtype)
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < RUN; ++j) {
          if (j < run) {
            const int d = a.tape.ops[o + j].dst;
#pragma unroll
            for (int h = 0; h < NW; ++h) file[(d * NW + h) * TAPE_THREADS] = q[j][h];
          }
        }
        if (run > 1) { o += run - 1; ahead = a.tape.ops[o + 1 < SKR_TAPE_MAX_OPS ? o + 1 : o]; }
        break;
      }
      case SKR_TAPE_STORE: {
        void* dstp = a.out[ib];
#pragma unroll
        for (int h = 0; h < NW; ++h) {
          const Word v = file[(ia * NW + h) * TAPE_THREADS];
          if (whole[h]) {
            if constexpr (sizeof(T) == 8) { f64x2_t* p = reinterpret_cast<f64x2_t*>(dstp) + 2 * w[h]; p[0] = v.lo; p[1] = v.hi; }
            else __builtin_nontemporal_store(v, reinterpret_cast<Word*>(dstp) + w[h]);
          } else if (some[h]) {
            M x[WE];
            tape_unpack<T, M>(v, x);
#pragma unroll
            for (int i = 0; i < WE; ++i) if (w[h] * WE + i < a.numel) store_scalar<T, M>(dstp, w[h] * WE + i, x[i]);
          }
        }
        break;
      }
      default: {
        // both operand numbers are valid for every code (the host checks them).  The Python scalar: converted to the op-math type for x k, / k, k /
        // (torch's mul / div keep it there) -- but torch's add / sub / rsub round it to the TENSOR dtype first (`bf16_tensor + 7.7` adds 7.6875; found
        // by the tape fuzz of tests/test_step_gpu.py)
        M k = (M)op.k;
        if constexpr (!std::is_same<M, double>::value) {
          if (code == SKR_TAPE_ADD_S || code == SKR_TAPE_RSUB_S) k = rnd<T>(k);
        }
        switch (code) {
          case SKR_TAPE_MUL_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return mul_(x, kk); }); break;
          case SKR_TAPE_DIV_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return div_(x, kk); }); break;
          case SKR_TAPE_ADD_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return tape_add(x, kk); }); break;
          case SKR_TAPE_RSUB_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return sub_(kk, x); }); break;
          case SKR_TAPE_RDIV_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return div_(kk, x); }); break;
          case SKR_TAPE_ADD: tape_binary<T, M, NW>(file, dst, ia, ib, [](M x, M z) { return tape_add(x, z); }); break;
          case SKR_TAPE_SUB: tape_binary<T, M, NW>(file, dst, ia, ib, [](M x, M z) { return sub_(x, z); }); break;
          case SKR_TAPE_MUL: tape_binary<T, M, NW>(file, dst, ia, ib, [](M x, M z) { return mul_(x, z); }); break;
          case SKR_TAPE_DIV: tape_binary<T, M, NW>(file, dst, ia, ib, [](M x, M z) { return div_(x, z); }); break;
          // a product by a scalar read by one sum or difference only, in the same visit (the product rounded to the tensor dtype on its way, as the op it stands for)
          case SKR_TAPE_ADD_MS: tape_binary<T, M, NW>(file, dst, ia, ib, [k](M x, M z) { return tape_add(x, tape_round<T, M>(mul_(z, k))); }); break;
          case SKR_TAPE_SUB_MS: tape_binary<T, M, NW>(file, dst, ia, ib, [k](M x, M z) { return sub_(x, tape_round<T, M>(mul_(z, k))); }); break;
          case SKR_TAPE_RSUB_MS: tape_binary<T, M, NW>(file, dst, ia, ib, [k](M x, M z) { return sub_(tape_round<T, M>(mul_(z, k)), x); }); break;
          case SKR_TAPE_MULZ_S: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M kk) { return tape_add(tape_round<T, M>(mul_(x, kk)), (M)0); }); break;
          default: tape_scalar<T, M, NW>(file, dst, ia, k, [](M x, M) { return -x; }); break;  // SKR_TAPE_NEG
        }
        break;
      }
      }
    }
  }
}

template <typename T, typename M, int NW>
static int launch_tape_words(const TapeArgs& a, int regs, hipStream_t s) {
  constexpr int WE = TapeElems<T>::value;
  const int64_t n_words = (a.numel + WE - 1) / WE;
  int64_t blocks = (n_words + (int64_t)TAPE_THREADS * NW - 1) / ((int64_t)TAPE_THREADS * NW);
  if (blocks > 256 * 32) blocks = 256 * 32;  // chunk-stride beyond 32 blocks per CU
  const size_t lds = sizeof(typename TapeWord<T>::type) * (size_t)regs * NW * TAPE_THREADS;  // per register and word: 4 KiB (16-bit, fp32), 8 KiB (fp64)
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(tape_kernel<T, M, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return SKR_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((tape_kernel<T, M, NW>), dim3((unsigned)blocks), dim3(TAPE_THREADS), lds, s, a);
  return finish_launch();
}

template <typename T, typename M>
static int launch_tape(const TapeArgs& a, hipStream_t s) {
  // the file holds the registers the tape names, not all SKR_TAPE_REGS: the elements in flight on a CU are what its LDS can give a register
  // file to, and the loads of those elements are the memory parallelism of this kernel (a 4-register Euler tape: 8 KiB per block and word)
  int regs = 1;
  for (int o = 0; o < a.tape.n_ops; ++o) {
    const skr_tape_op& op = a.tape.ops[o];
    const bool two = op.code >= SKR_TAPE_ADD && op.code != SKR_TAPE_NEG && op.code != SKR_TAPE_MULZ_S;  // (the second operand number means a register for the tensor-tensor ops only)
    int hi = op.code == SKR_TAPE_LOAD ? op.dst : (op.code == SKR_TAPE_STORE ? op.a : (op.dst > op.a ? op.dst : op.a));
    if (two && op.b > hi) hi = op.b;
    if (hi + 1 > regs) regs = hi + 1;
  }
  const int forced = g_tune.tape_words;  // tuning switch (skr_set_tuning "tape_words"): 1 or 2 words per lane
  constexpr int WE = TapeElems<T>::value;
  // two words per lane halve the decode work per element, and pay while four blocks of the doubled file still fit a CU (measured at 256 x 4 x 128 x 128
  // bf16, kernel time under rocprofv3: Euler, 2 registers, 26.2 -> 22.4 us; DPM-2, 5 registers, 61.6 -> 58.0; UniPC-3, 8 registers, 137.6 -> 148.6);
  // one word where the tensor is too small to give every CU two blocks that way
  const size_t doubled = sizeof(typename TapeWord<T>::type) * (size_t)regs * 2 * TAPE_THREADS;
  const bool two_words = forced ? (forced == 2 && doubled <= 64 * 1024) : (doubled <= 40 * 1024 && a.numel >= (int64_t)WE * 2 * TAPE_THREADS * 512);
  return two_words ? launch_tape_words<T, M, 2>(a, regs, s) : launch_tape_words<T, M, 1>(a, regs, s);
}

}  // namespace skr

extern "C" int skr_tape_launch(const skr_tape* tape, const void* const* inputs, void* const* outputs, int64_t numel, void* stream) {
  if (!tape || !inputs || !outputs) return SKR_ERR_NULL;
  const skr_tape& t = *tape;
  if (t.n_ops < 0 || t.n_ops > SKR_TAPE_MAX_OPS || t.n_inputs < 1 || t.n_inputs > SKR_TAPE_MAX_INPUTS || t.n_outputs < 1 || t.n_outputs > SKR_TAPE_MAX_OUTPUTS) return SKR_ERR_TERMS;
  if (numel < 0) return SKR_ERR_SHAPE;
  if (numel == 0) return SKR_OK;
  skr::DeviceGuard device_guard(outputs[0]);
  skr::TapeArgs a;
  a.tape = t;
  a.numel = numel;
  for (int i = 0; i < SKR_TAPE_MAX_INPUTS; ++i) a.in[i] = nullptr;
  for (int i = 0; i < SKR_TAPE_MAX_OUTPUTS; ++i) a.out[i] = nullptr;
  for (int i = 0; i < t.n_inputs; ++i) {
    if (!inputs[i]) return SKR_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(inputs[i]) & 15u) return SKR_ERR_ALIGN;
    a.in[i] = inputs[i];
  }
  for (int i = 0; i < t.n_outputs; ++i) {
    if (!outputs[i]) return SKR_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(outputs[i]) & 15u) return SKR_ERR_ALIGN;
    a.out[i] = outputs[i];
  }
  // every operand number is checked here, once: the kernel indexes registers and pointers with them
  for (int o = 0; o < t.n_ops; ++o) {
    const skr_tape_op& op = t.ops[o];
    const bool reg_a = op.a >= 0 && op.a < SKR_TAPE_REGS, reg_b = op.b >= 0 && op.b < SKR_TAPE_REGS, reg_d = op.dst >= 0 && op.dst < SKR_TAPE_REGS;
    switch (op.code) {
      case SKR_TAPE_LOAD: if (!reg_d || op.a < 0 || op.a >= t.n_inputs) return SKR_ERR_TERMS; break;
      case SKR_TAPE_STORE: if (!reg_a || op.b < 0 || op.b >= t.n_outputs) return SKR_ERR_TERMS; break;
      case SKR_TAPE_MUL_S: case SKR_TAPE_DIV_S: case SKR_TAPE_ADD_S: case SKR_TAPE_RSUB_S: case SKR_TAPE_RDIV_S: case SKR_TAPE_NEG: case SKR_TAPE_MULZ_S:
        if (!reg_a || !reg_d) return SKR_ERR_TERMS; break;
      case SKR_TAPE_ADD: case SKR_TAPE_SUB: case SKR_TAPE_MUL: case SKR_TAPE_DIV: case SKR_TAPE_ADD_MS: case SKR_TAPE_SUB_MS: case SKR_TAPE_RSUB_MS:
        if (!reg_a || !reg_b || !reg_d) return SKR_ERR_TERMS; break;
      default: return SKR_ERR_UNSUPPORTED;
    }
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (t.dtype) {
    case SKR_BF16: return skr::launch_tape<skr::bf16_t, float>(a, s);
    case SKR_F16: return skr::launch_tape<skr::f16_t, float>(a, s);
    case SKR_F32: return skr::launch_tape<float, float>(a, s);
    case SKR_F64: return skr::launch_tape<double, double>(a, s);
    default: return SKR_ERR_DTYPE;
  }
}
