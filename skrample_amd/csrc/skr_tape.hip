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
// Registers: SKR_TAPE_REGS values per element, as ext-vector registers indexed by the (wave-uniform) operand numbers of the current op --
// the compiler turns that into VGPR-indexed moves, not scratch.  The tape itself sits in the kernel argument block (scalar loads).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skr_device.h"
#include "skr_step_common.h"

namespace skr {

constexpr int TAPE_ELEMS = 4;  // elements per lane and trip: 8-byte accesses on 16-bit tensors, 16-byte on fp32, 32-byte on fp64

struct TapeArgs {
  skr_tape tape;
  const void* in[SKR_TAPE_MAX_INPUTS];
  void* out[SKR_TAPE_MAX_OUTPUTS];
  int64_t numel;
};

template <typename M> struct TapeRegs;
template <> struct TapeRegs<float> { typedef float type __attribute__((ext_vector_type(SKR_TAPE_REGS))); };
template <> struct TapeRegs<double> { typedef double type __attribute__((ext_vector_type(SKR_TAPE_REGS))); };

// one value rounded to the tensor dtype, kept in the op-math type
template <typename T, typename M> __device__ __forceinline__ M tape_round(M v) {
  if constexpr (std::is_same<M, double>::value) return v;
  else return rnd<T>(v);
}

template <typename T, typename M>
__device__ __forceinline__ M tape_load(const void* base, int64_t e) {
  if constexpr (std::is_same<M, double>::value) return load_scalar_d<T>(base, e);
  else return load_scalar<T>(base, e);
}

// The arithmetic of one op.  Contraction is off inside mul_ / sub_ (skr_step_common.h); additions are spelled through sub_ of the
// negated operand only where a neighbouring multiply could be contracted into them -- here every op stands alone, so plain operators
// under `fp contract(off)` are enough.
template <typename M> __device__ __forceinline__ M tape_add(M a, M b) {
#pragma clang fp contract(off)
  return a + b;
}

template <typename T, typename M>
__global__ __launch_bounds__(256) void tape_kernel(const TapeArgs a) {
  typedef typename TapeRegs<M>::type Regs;
  const int n_ops = a.tape.n_ops;
  const int64_t n_vec = (a.numel + TAPE_ELEMS - 1) / TAPE_ELEMS;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * 256) {
    const int64_t e0 = v * TAPE_ELEMS;
    const bool whole = e0 + TAPE_ELEMS <= a.numel;
    Regs r[TAPE_ELEMS];
#pragma unroll
    for (int i = 0; i < TAPE_ELEMS; ++i) r[i] = (M)0;
    for (int o = 0; o < n_ops; ++o) {
      const skr_tape_op op = a.tape.ops[o];  // uniform: scalar loads from the kernel argument block
      const int code = op.code, dst = op.dst, ia = op.a, ib = op.b;
      const M k = (M)op.k;  // the Python scalar, converted to the op-math type as torch does
      if (code == SKR_TAPE_LOAD) {
        const void* src = a.in[ia];
        if (whole) {
          if constexpr (sizeof(T) == 2) {
            const u32x2_t q = *(reinterpret_cast<const u32x2_t*>(src) + v);
            if constexpr (std::is_same<T, bf16_t>::value) {
              r[0][dst] = __uint_as_float(q[0] << 16); r[1][dst] = __uint_as_float(q[0] & 0xFFFF0000u);
              r[2][dst] = __uint_as_float(q[1] << 16); r[3][dst] = __uint_as_float(q[1] & 0xFFFF0000u);
            } else {
              r[0][dst] = (float)__builtin_bit_cast(_Float16, (uint16_t)(q[0] & 0xFFFFu)); r[1][dst] = (float)__builtin_bit_cast(_Float16, (uint16_t)(q[0] >> 16));
              r[2][dst] = (float)__builtin_bit_cast(_Float16, (uint16_t)(q[1] & 0xFFFFu)); r[3][dst] = (float)__builtin_bit_cast(_Float16, (uint16_t)(q[1] >> 16));
            }
          } else if constexpr (sizeof(T) == 4) {
            const f32x4_t q = *(reinterpret_cast<const f32x4_t*>(src) + v);
#pragma unroll
            for (int i = 0; i < TAPE_ELEMS; ++i) r[i][dst] = q[i];
          } else {
            const f64x2_t q0 = *(reinterpret_cast<const f64x2_t*>(src) + 2 * v), q1 = *(reinterpret_cast<const f64x2_t*>(src) + 2 * v + 1);
            r[0][dst] = q0[0]; r[1][dst] = q0[1]; r[2][dst] = q1[0]; r[3][dst] = q1[1];
          }
        } else {
#pragma unroll
          for (int i = 0; i < TAPE_ELEMS; ++i) r[i][dst] = e0 + i < a.numel ? tape_load<T, M>(src, e0 + i) : (M)0;
        }
      } else if (code == SKR_TAPE_STORE) {
        void* dstp = a.out[ib];
        if (whole && sizeof(T) == 2) {
          u32x2_t q;
          if constexpr (std::is_same<T, bf16_t>::value) { q[0] = pack_bf16((float)r[0][ia], (float)r[1][ia]); q[1] = pack_bf16((float)r[2][ia], (float)r[3][ia]); }
          else { q[0] = pack_f16((float)r[0][ia], (float)r[1][ia]); q[1] = pack_f16((float)r[2][ia], (float)r[3][ia]); }
          __builtin_nontemporal_store(q, reinterpret_cast<u32x2_t*>(dstp) + v);
        } else if (whole && sizeof(T) == 4) {
          __builtin_nontemporal_store(f32x4_t{(float)r[0][ia], (float)r[1][ia], (float)r[2][ia], (float)r[3][ia]}, reinterpret_cast<f32x4_t*>(dstp) + v);
        } else {
#pragma unroll
          for (int i = 0; i < TAPE_ELEMS; ++i) if (e0 + i < a.numel) store_scalar<T, M>(dstp, e0 + i, r[i][ia]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < TAPE_ELEMS; ++i) {
          const M x = r[i][ia];
          M y;
          switch (code) {
            case SKR_TAPE_MUL_S: y = mul_(x, k); break;
            case SKR_TAPE_DIV_S: y = div_(x, k); break;
            case SKR_TAPE_ADD_S: y = tape_add(x, k); break;
            case SKR_TAPE_RSUB_S: y = sub_(k, x); break;
            case SKR_TAPE_RDIV_S: y = div_(k, x); break;
            case SKR_TAPE_ADD: y = tape_add(x, (M)r[i][ib]); break;
            case SKR_TAPE_SUB: y = sub_(x, (M)r[i][ib]); break;
            case SKR_TAPE_MUL: y = mul_(x, (M)r[i][ib]); break;
            case SKR_TAPE_DIV: y = div_(x, (M)r[i][ib]); break;
            default: y = -x; break;  // SKR_TAPE_NEG
          }
          r[i][dst] = tape_round<T, M>(y);
        }
      }
    }
  }
}

template <typename T, typename M>
static int launch_tape(const TapeArgs& a, hipStream_t s) {
  const int64_t n_vec = (a.numel + TAPE_ELEMS - 1) / TAPE_ELEMS;
  int64_t blocks = (n_vec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
  hipLaunchKernelGGL((tape_kernel<T, M>), dim3((unsigned)blocks), dim3(256), 0, s, a);
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
