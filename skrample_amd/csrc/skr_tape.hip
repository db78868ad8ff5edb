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

template <typename T, typename M>
__global__ __launch_bounds__(TAPE_THREADS) void tape_kernel(const TapeArgs a) {
  typedef typename TapeWord<T>::type Word;
  constexpr int TAPE_ELEMS = TapeElems<T>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char tape_lds[];
  Word* const file = reinterpret_cast<Word*>(tape_lds) + threadIdx.x;  // register g of this thread: file[g * TAPE_THREADS]
  const int n_ops = a.tape.n_ops;
  const int64_t n_vec = (a.numel + TAPE_ELEMS - 1) / TAPE_ELEMS;
  for (int64_t v = (int64_t)blockIdx.x * TAPE_THREADS + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * TAPE_THREADS) {
    const int64_t e0 = v * TAPE_ELEMS;
    const bool whole = e0 + TAPE_ELEMS <= a.numel;
    skr_tape_op ahead = a.tape.ops[0];  // uniform: scalar loads from the kernel argument block, one op ahead of its use (the load's latency
    for (int o = 0; o < n_ops; ++o) {   // then runs under the current op's LDS round trip instead of in front of it)
      const skr_tape_op op = ahead;
      ahead = a.tape.ops[o + 1 < SKR_TAPE_MAX_OPS ? o + 1 : o];
      const int code = op.code, dst = op.dst, ia = op.a, ib = op.b;
      if (code == SKR_TAPE_LOAD) {
        // A run of consecutive LOADs (the recorder opens a tape with its leaves) is issued as ONE batch: every global load first, the LDS
        // writes behind them -- one memory latency per run instead of one per input.
        constexpr int RUN = sizeof(T) == 8 ? 2 : (sizeof(T) == 4 ? 4 : 6);  // (batch registers: 8, 16, 24)
        int run = 1;
        while (run < RUN && o + run < n_ops && a.tape.ops[o + run].code == SKR_TAPE_LOAD) ++run;
        Word q[RUN];
        if (whole) {
#pragma unroll
          for (int j = 0; j < RUN; ++j)
            if (j < run) q[j] = *(reinterpret_cast<const Word*>(a.in[a.tape.ops[o + j].a]) + v);
        } else {
          for (int j = 0; j < run; ++j) {
            const void* src = a.in[a.tape.ops[o + j].a];
            M x[TAPE_ELEMS];
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) {
              if constexpr (std::is_same<M, double>::value) x[i] = e0 + i < a.numel ? load_scalar_d<T>(src, e0 + i) : 0.0;
              else x[i] = e0 + i < a.numel ? load_scalar<T>(src, e0 + i) : 0.f;
            }
            const Word w = tape_pack<T, M>(x);  // (exact: the values come from the tensor dtype)
#pragma unroll
            for (int jj = 0; jj < RUN; ++jj)
              if (jj == j) q[jj] = w;
          }
        }
#pragma unroll
        for (int j = 0; j < RUN; ++j)
          if (j < run) file[a.tape.ops[o + j].dst * TAPE_THREADS] = q[j];
        if (run > 1) { o += run - 1; ahead = a.tape.ops[o + 1 < SKR_TAPE_MAX_OPS ? o + 1 : o]; }
      } else if (code == SKR_TAPE_STORE) {
        const Word w = file[ia * TAPE_THREADS];
        void* dstp = a.out[ib];
        if (whole) {
          if constexpr (sizeof(T) == 8) { f64x2_t* p = reinterpret_cast<f64x2_t*>(dstp) + 2 * v; p[0] = w.lo; p[1] = w.hi; }
          else __builtin_nontemporal_store(w, reinterpret_cast<Word*>(dstp) + v);
        } else {
          M x[TAPE_ELEMS];
          tape_unpack<T, M>(w, x);
#pragma unroll
          for (int i = 0; i < TAPE_ELEMS; ++i) if (e0 + i < a.numel) store_scalar<T, M>(dstp, e0 + i, x[i]);
        }
      } else {
        // both operand numbers are valid for every code (the host checks them); a scalar op replaces the second operand by the Python
        // scalar, converted to the op-math type as torch does
        M x[TAPE_ELEMS], z[TAPE_ELEMS], y[TAPE_ELEMS];
        tape_unpack<T, M>(file[ia * TAPE_THREADS], x);
        if (code >= SKR_TAPE_ADD && code != SKR_TAPE_NEG) tape_unpack<T, M>(file[ib * TAPE_THREADS], z);
        else {
          // the Python scalar: converted to the op-math type for x k, / k, k / (torch's mul / div keep it there) -- but torch's add / sub /
          // rsub round it to the TENSOR dtype first (`bf16_tensor + 7.7` adds 7.6875; found by the tape fuzz of tests/test_step_gpu.py)
          M k = (M)op.k;
          if constexpr (!std::is_same<M, double>::value) {
            if (code == SKR_TAPE_ADD_S || code == SKR_TAPE_RSUB_S) k = rnd<T>(k);
          }
#pragma unroll
          for (int i = 0; i < TAPE_ELEMS; ++i) z[i] = k;
        }
        switch (code) {
          case SKR_TAPE_MUL_S: case SKR_TAPE_MUL:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = mul_(x[i], z[i]);
            break;
          case SKR_TAPE_DIV_S: case SKR_TAPE_DIV:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = div_(x[i], z[i]);
            break;
          case SKR_TAPE_ADD_S: case SKR_TAPE_ADD:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = tape_add(x[i], z[i]);
            break;
          case SKR_TAPE_RSUB_S:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = sub_(z[i], x[i]);
            break;
          case SKR_TAPE_SUB:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = sub_(x[i], z[i]);
            break;
          case SKR_TAPE_RDIV_S:
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = div_(z[i], x[i]);
            break;
          default:  // SKR_TAPE_NEG
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) y[i] = -x[i];
            break;
        }
        file[dst * TAPE_THREADS] = tape_pack<T, M>(y);
      }
    }
  }
}

template <typename T, typename M>
static int launch_tape(const TapeArgs& a, hipStream_t s) {
  constexpr int TAPE_ELEMS = TapeElems<T>::value;
  const int64_t n_vec = (a.numel + TAPE_ELEMS - 1) / TAPE_ELEMS;
  int64_t blocks = (n_vec + TAPE_THREADS - 1) / TAPE_THREADS;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
  // the file holds the registers the tape names, not all SKR_TAPE_REGS: the elements in flight on a CU are what its LDS can give a register
  // file to, and the loads of those elements are the memory parallelism of this kernel (a 4-register Euler tape: 8 KiB per block, 32 waves per CU)
  int regs = 1;
  for (int o = 0; o < a.tape.n_ops; ++o) {
    const skr_tape_op& op = a.tape.ops[o];
    const bool two = op.code >= SKR_TAPE_ADD && op.code != SKR_TAPE_NEG;  // (the second operand number means a register for the tensor-tensor ops only)
    int hi = op.code == SKR_TAPE_LOAD ? op.dst : (op.code == SKR_TAPE_STORE ? op.a : (op.dst > op.a ? op.dst : op.a));
    if (two && op.b > hi) hi = op.b;
    if (hi + 1 > regs) regs = hi + 1;
  }
  const size_t lds = sizeof(typename TapeWord<T>::type) * (size_t)regs * TAPE_THREADS;  // per register: 4 KiB (16-bit, fp32), 8 KiB (fp64)
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(tape_kernel<T, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return SKR_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((tape_kernel<T, M>), dim3((unsigned)blocks), dim3(TAPE_THREADS), lds, s, a);
  return finish_launch();
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
      case SKR_TAPE_MUL_S: case SKR_TAPE_DIV_S: case SKR_TAPE_ADD_S: case SKR_TAPE_RSUB_S: case SKR_TAPE_RDIV_S: case SKR_TAPE_NEG:
        if (!reg_a || !reg_d) return SKR_ERR_TERMS; break;
      case SKR_TAPE_ADD: case SKR_TAPE_SUB: case SKR_TAPE_MUL: case SKR_TAPE_DIV:
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
