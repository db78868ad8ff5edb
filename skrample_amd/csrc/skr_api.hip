// C-ABI odds and ends: version / error strings, the stand-alone Random generator and the raw
// Philox dump used by the parity tests.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/skrample_hip.h"
#include "skr_philox.h"
#include "skr_pack.h"

extern "C" int skr_abi_version(void) { return SKR_ABI_VERSION; }

extern "C" const char* skr_build_info(void) { return "skrample_hip gfx950 (wave64), built " __DATE__ " " __TIME__; }

extern "C" const char* skr_strerror(int status) {
  switch (status) {
    case SKR_OK: return "ok";
    case SKR_ERR_NULL: return "required pointer is NULL";
    case SKR_ERR_DTYPE: return "dtype combination has no kernel";
    case SKR_ERR_TERMS: return "term count out of range or terms not grouped by dtype";
    case SKR_ERR_ALIGN: return "buffer not 16-byte aligned";
    case SKR_ERR_SHAPE: return "inconsistent numel / sample_numel / shape";
    case SKR_ERR_LAUNCH: return "hip kernel launch failed";
    case SKR_ERR_UNSUPPORTED: return "request outside kernel coverage";
    default: return "unknown status";
  }
}

// Random.generate (reference noise.py:58-74) for ANY per-sample size: one lane per Philox block of
// a sample (4 elements), so ragged sizes need no special casing.  Shapes with sample_numel % 8 == 0
// normally never get here -- their noise is drawn inside the fused step kernel.
template <typename T>
__global__ __launch_bounds__(256) void random_kernel(T* out, const uint64_t* seeds, uint64_t stream_id, int64_t sample_numel, int64_t blocks_per_sample, int64_t total_blocks) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_blocks; i += (int64_t)gridDim.x * 256) {
    const int64_t smp = i / blocks_per_sample;
    const int64_t blk = i - smp * blocks_per_sample;
    float z[4];
    skr::normal4(seeds[smp], stream_id, (uint64_t)blk, z);
    const int64_t r0 = blk * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r0 + j < sample_numel) out[smp * sample_numel + r0 + j] = (T)z[j];
    }
  }
}

// Aligned fast path (sample_numel % 8 == 0 and a 16-byte aligned base): blockIdx.y = sample, every thread draws two
// Philox blocks and writes its 8 values with one packed non-temporal store.  Same (seed, stream, block) -> value
// mapping as the kernel above, so both produce identical bits.
template <typename T>
__global__ __launch_bounds__(256) void random_kernel_v8(T* out, const uint64_t* seeds, uint64_t stream_id, int64_t sample_numel) {
  const int64_t smp = blockIdx.y, vps = sample_numel >> 3;
  const uint64_t seed = seeds[smp];
  T* dst = out + smp * sample_numel;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < vps; v += (int64_t)gridDim.x * 256) {
    float z[8];
    skr::normal4(seed, stream_id, (uint64_t)(2 * v), z);
    skr::normal4(seed, stream_id, (uint64_t)(2 * v + 1), z + 4);
    skr::store8_from_f32<T>(dst, v, z);
  }
}

template <typename T>
static void launch_random(void* out, const uint64_t* seeds, uint64_t stream_id, int64_t batch, int64_t sample_numel, hipStream_t s) {
  const bool fast = sample_numel % 8 == 0 && batch <= 65535 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (fast) {
    const int64_t vps = sample_numel / 8;
    int64_t bx = (vps + 255) / 256;
    const int64_t cap = (256 * 16 + batch - 1) / batch;  // ~16 blocks per CU over the whole grid
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(random_kernel_v8<T>, dim3((unsigned)bx, (unsigned)batch), dim3(256), 0, s, (T*)out, seeds, stream_id, sample_numel);
    return;
  }
  const int64_t bps = (sample_numel + 3) / 4, total = bps * batch;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(random_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, (T*)out, seeds, stream_id, sample_numel, bps, total);
}

extern "C" int skr_noise_random(void* out, int32_t out_dtype, const uint64_t* seeds_dev, uint64_t stream_id,
                                int64_t batch, int64_t sample_numel, void* stream) {
  if (batch < 0 || sample_numel < 0) return SKR_ERR_SHAPE;
  if (batch == 0 || sample_numel == 0) return SKR_OK;
  if (!out || !seeds_dev) return SKR_ERR_NULL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SKR_BF16: launch_random<__bf16>(out, seeds_dev, stream_id, batch, sample_numel, s); break;
    case SKR_F16: launch_random<_Float16>(out, seeds_dev, stream_id, batch, sample_numel, s); break;
    case SKR_F32: launch_random<float>(out, seeds_dev, stream_id, batch, sample_numel, s); break;
    case SKR_F64: launch_random<double>(out, seeds_dev, stream_id, batch, sample_numel, s); break;
    default: return SKR_ERR_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

// ---- weighted sum of Philox normal streams (Brownian increments) -----------------------------------------
struct WeightedArgs {
  uint64_t stream_ids[SKR_MAX_WEIGHTED_STREAMS];
  float weights[SKR_MAX_WEIGHTED_STREAMS];
  int32_t n;
};

// ALIGNED: sample_numel % 8 == 0 and a 16-byte aligned base -> blockIdx.y = sample, 8 elements per thread, packed
// store.  Otherwise one Philox block (4 elements) per thread with bounds checks.  Stream ids and weights are
// wave-uniform (scalar loads from the kernel arguments); fp32 FMA accumulation in stream order.
template <typename T, bool ALIGNED>
__global__ __launch_bounds__(256) void weighted_kernel(T* out, const uint64_t* seeds, const WeightedArgs a, int64_t sample_numel, int64_t blocks_per_sample, int64_t total_blocks) {
  if constexpr (ALIGNED) {
    const int64_t smp = blockIdx.y, vps = sample_numel >> 3;
    const uint64_t seed = seeds[smp];
    T* dst = out + smp * sample_numel;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < vps; v += (int64_t)gridDim.x * 256) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < a.n; ++k) {
        float z[8];
        skr::normal4(seed, a.stream_ids[k], (uint64_t)(2 * v), z);
        skr::normal4(seed, a.stream_ids[k], (uint64_t)(2 * v + 1), z + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(a.weights[k], z[j], acc[j]);
      }
      skr::store8_from_f32<T>(dst, v, acc);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_blocks; i += (int64_t)gridDim.x * 256) {
      const int64_t smp = i / blocks_per_sample, blk = i - smp * blocks_per_sample;
      const uint64_t seed = seeds[smp];
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < a.n; ++k) {
        float z[4];
        skr::normal4(seed, a.stream_ids[k], (uint64_t)blk, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(a.weights[k], z[j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (blk * 4 + j < sample_numel) out[smp * sample_numel + blk * 4 + j] = (T)acc[j];
      }
    }
  }
}

template <typename T>
static void launch_weighted(void* out, const uint64_t* seeds, const WeightedArgs& a, int64_t batch, int64_t sample_numel, hipStream_t s) {
  const bool fast = sample_numel % 8 == 0 && batch <= 65535 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const int64_t bps = (sample_numel + 3) / 4, total = bps * batch;
  if (fast) {
    int64_t bx = (sample_numel / 8 + 255) / 256;
    const int64_t cap = (256 * 16 + batch - 1) / batch;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL((weighted_kernel<T, true>), dim3((unsigned)bx, (unsigned)batch), dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
    return;
  }
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL((weighted_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
}

extern "C" int skr_noise_weighted(void* out, int32_t out_dtype, const uint64_t* seeds_dev, const uint64_t* stream_ids, const double* weights,
                                  int32_t n_streams, int64_t batch, int64_t sample_numel, void* stream) {
  if (batch < 0 || sample_numel < 0 || n_streams < 0) return SKR_ERR_SHAPE;
  if (n_streams > SKR_MAX_WEIGHTED_STREAMS) return SKR_ERR_UNSUPPORTED;
  if (batch == 0 || sample_numel == 0) return SKR_OK;
  if (!out || !seeds_dev || (n_streams > 0 && (!stream_ids || !weights))) return SKR_ERR_NULL;
  WeightedArgs a;
  for (int k = 0; k < SKR_MAX_WEIGHTED_STREAMS; ++k) {
    a.stream_ids[k] = k < n_streams ? stream_ids[k] : 0;
    a.weights[k] = k < n_streams ? (float)weights[k] : 0.f;
  }
  a.n = n_streams;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SKR_BF16: launch_weighted<__bf16>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F16: launch_weighted<_Float16>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F32: launch_weighted<float>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F64: launch_weighted<double>(out, seeds_dev, a, batch, sample_numel, s); break;
    default: return SKR_ERR_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

__global__ void philox_dump_kernel(uint32_t* out, uint64_t seed, uint64_t stream_id, uint64_t first, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t blk = first + (uint64_t)i;
  skr::u32x4 c{(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  c = skr::philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  out[4 * i + 0] = c.x;
  out[4 * i + 1] = c.y;
  out[4 * i + 2] = c.z;
  out[4 * i + 3] = c.w;
}

extern "C" int skr_philox_u32(uint32_t* out, uint64_t seed, uint64_t stream_id, uint64_t first_block, int64_t n_blocks, void* stream) {
  if (!out) return SKR_ERR_NULL;
  if (n_blocks < 0) return SKR_ERR_SHAPE;
  if (n_blocks == 0) return SKR_OK;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n_blocks + threads - 1) / threads);
  hipLaunchKernelGGL(philox_dump_kernel, dim3(blocks), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), out, seed, stream_id, first_block, n_blocks);
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}
