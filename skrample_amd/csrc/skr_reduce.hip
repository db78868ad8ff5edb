// Error norms for adaptive step control (reference skrample/sampling/functional.py:197-214: FunctionalAdaptive.mae /
// .mse = mean(|a - b|^p)).  Two-stage, fixed-order reduction in double: bit-reproducible, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skr_device.h"
#include "../../include/skrample_hip.h"

namespace {

constexpr int RED_BLOCKS = 1024;

template <typename T> __device__ __forceinline__ double ld(const void* p, int64_t i) { return (double)reinterpret_cast<const T*>(p)[i]; }
template <> __device__ __forceinline__ double ld<__bf16>(const void* p, int64_t i) { return (double)(float)reinterpret_cast<const __bf16*>(p)[i]; }
template <> __device__ __forceinline__ double ld<_Float16>(const void* p, int64_t i) { return (double)(float)reinterpret_cast<const _Float16*>(p)[i]; }

template <typename T>
__global__ __launch_bounds__(256) void norm_partials(const void* a, const void* b, int64_t numel, int power, double* partials) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const double d = fabs((a ? ld<T>(a, i) : 0.0) - ld<T>(b, i));
    s += power == 2 ? d * d : d;
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void norm_final(const double* partials, int n, int64_t numel, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partials[i];
    out[0] = s / (double)numel;
  }
}

}  // namespace

extern "C" int skr_error_mean(const void* a_or_null, const void* b, int32_t dtype, int64_t numel, int32_t power,
                              double* out_dev, double* partials_dev /* [1024] */, void* stream) {
  skr::DeviceGuard device_guard(b);
  if (!b || !out_dev || !partials_dev) return SKR_ERR_NULL;
  if (numel <= 0) return SKR_ERR_SHAPE;
  if (power != 1 && power != 2) return SKR_ERR_UNSUPPORTED;
  int64_t blocks = (numel + 255) / 256;
  if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (dtype) {
    case SKR_BF16: hipLaunchKernelGGL(norm_partials<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, a_or_null, b, numel, power, partials_dev); break;
    case SKR_F16: hipLaunchKernelGGL(norm_partials<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, a_or_null, b, numel, power, partials_dev); break;
    case SKR_F32: hipLaunchKernelGGL(norm_partials<float>, dim3((unsigned)blocks), dim3(256), 0, s, a_or_null, b, numel, power, partials_dev); break;
    case SKR_F64: hipLaunchKernelGGL(norm_partials<double>, dim3((unsigned)blocks), dim3(256), 0, s, a_or_null, b, numel, power, partials_dev); break;
    default: return SKR_ERR_DTYPE;
  }
  hipLaunchKernelGGL(norm_final, dim3(1), dim3(64), 0, s, partials_dev, (int)blocks, numel, out_dev);
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

// ---- signed-power blend (SPC with power != 1; reference structured.py:568-572, common.py:187-190) --------------
// out = spowf(p * spowf(a, P) + c * spowf(b, P), 1/P),  spowf(x, f) = |x|^f * sign(x)   (sign(0) = +1 as in the reference)
// fp32 results: |x|^f on the raw log2/exp2 units (|x| = 0 gives 0 for f > 0); fp64 results (compute_scale = float64):
// double-precision pow.  Elementwise, any numel.
namespace skr {
template <typename T, typename M> __device__ __forceinline__ M blend_load(const void* p, int64_t i) {
  if constexpr (__is_same(T, __bf16)) return (M)__uint_as_float((uint32_t) reinterpret_cast<const uint16_t*>(p)[i] << 16);
  else return (M) reinterpret_cast<const T*>(p)[i];
}

__device__ __forceinline__ float spow_dev(float x, float f) {
  const float m = __builtin_amdgcn_exp2f(f * __builtin_amdgcn_logf(__builtin_fabsf(x)));
  return x < 0.f ? -m : m;
}
__device__ __forceinline__ double spow_dev(double x, double f) {
  const double m = pow(fabs(x), f);
  return x < 0.0 ? -m : m;
}

template <typename TA, typename TB, typename M>
__global__ __launch_bounds__(256) void power_blend_kernel(M* out, const void* a, const void* b, M p, M c, M power, M inv_power, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const M mix = spow_dev(blend_load<TA, M>(a, i), power) * p + spow_dev(blend_load<TB, M>(b, i), power) * c;
    out[i] = spow_dev(mix, inv_power);
  }
}

template <typename TA, typename M>
static int power_blend_b(M* out, const void* a, const void* b, int32_t b_dtype, double p, double c, double power, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const M mp = (M)p, mc = (M)c, mw = (M)power, ip = (M)(1.0 / power);
  switch (b_dtype) {
    case SKR_BF16: hipLaunchKernelGGL((power_blend_kernel<TA, __bf16, M>), dim3((unsigned)blocks), dim3(256), 0, s, out, a, b, mp, mc, mw, ip, n); break;
    case SKR_F16: hipLaunchKernelGGL((power_blend_kernel<TA, _Float16, M>), dim3((unsigned)blocks), dim3(256), 0, s, out, a, b, mp, mc, mw, ip, n); break;
    case SKR_F32: hipLaunchKernelGGL((power_blend_kernel<TA, float, M>), dim3((unsigned)blocks), dim3(256), 0, s, out, a, b, mp, mc, mw, ip, n); break;
    case SKR_F64: hipLaunchKernelGGL((power_blend_kernel<TA, double, M>), dim3((unsigned)blocks), dim3(256), 0, s, out, a, b, mp, mc, mw, ip, n); break;
    default: return SKR_ERR_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

template <typename M>
static int power_blend_a(M* out, const void* a, int32_t a_dtype, const void* b, int32_t b_dtype, double p, double c, double power, int64_t n, hipStream_t s) {
  switch (a_dtype) {
    case SKR_BF16: return power_blend_b<__bf16, M>(out, a, b, b_dtype, p, c, power, n, s);
    case SKR_F16: return power_blend_b<_Float16, M>(out, a, b, b_dtype, p, c, power, n, s);
    case SKR_F32: return power_blend_b<float, M>(out, a, b, b_dtype, p, c, power, n, s);
    case SKR_F64: return power_blend_b<double, M>(out, a, b, b_dtype, p, c, power, n, s);
    default: return SKR_ERR_DTYPE;
  }
}
}  // namespace skr

extern "C" int skr_power_blend(void* out, int32_t out_dtype, const void* a, int32_t a_dtype, const void* b, int32_t b_dtype, double p, double c,
                               double power, int64_t numel, void* stream) {
  skr::DeviceGuard device_guard(out);
  if (numel < 0) return SKR_ERR_SHAPE;
  if (numel == 0) return SKR_OK;
  if (!out || !a || !b) return SKR_ERR_NULL;
  if (power == 0.0) return SKR_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (out_dtype == SKR_F32) return skr::power_blend_a<float>((float*)out, a, a_dtype, b, b_dtype, p, c, power, numel, s);
  if (out_dtype == SKR_F64) return skr::power_blend_a<double>((double*)out, a, a_dtype, b, b_dtype, p, c, power, numel, s);
  return SKR_ERR_DTYPE;
}
