// Error norms for adaptive step control (reference skrample/sampling/functional.py:197-214: FunctionalAdaptive.mae /
// .mse = mean(|a - b|^p)).  Two-stage, fixed-order reduction in double: bit-reproducible, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
