// C-ABI odds and ends: version / error strings, the stand-alone Random generator and the raw
// Philox dump used by the parity tests.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skr_device.h"
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
    case SKR_ERR_CAPTURE: return "first use of this shape needs tables / plans that are built outside stream capture: run the shape once eagerly, then capture";
    case SKR_ERR_LIBRARY: return "hipFFT returned a wrong transform in its self-check (rocFFT plan defect): not used";
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
  skr::DeviceGuard device_guard(out);
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

// ---- Brownian increments: difference of two weighted sums of Philox normal streams ---------------------------
// out = scale * (S_to - S_from),  S_x = sum_k w_x[k] * N(stream_ids[k])   (fp32 FMA chain in node order)
// S_from is either computed in the same pass or, when the previous query ended where this one starts, read from the
// per-generator cache of S_to (fp32, updated in place).  A zero weight leaves an FMA chain untouched, so S(t) has the
// same bits whether it was accumulated alongside another path or on its own: cache hits and misses agree bit for bit.
struct BrownianArgs {
  uint64_t stream_ids[SKR_MAX_WEIGHTED_STREAMS];
  float w_to[SKR_MAX_WEIGHTED_STREAMS];
  float w_from[SKR_MAX_WEIGHTED_STREAMS];
  float* cache;  // [batch * sample_numel] fp32 or null
  float scale;
  int32_t n;
  int32_t from_cache;  // S_from = cache (else accumulate w_from)
};

// ALIGNED: sample_numel % 8 == 0 and 16-byte aligned bases -> blockIdx.y = sample, 8 elements per thread, packed
// store.  Otherwise one Philox block (4 elements) per thread with bounds checks.  Stream ids and weights are
// wave-uniform (scalar loads from the kernel arguments).
template <typename T, bool ALIGNED, bool FROM_CACHE>
__global__ __launch_bounds__(256) void brownian_kernel(T* out, const uint64_t* seeds, const BrownianArgs a, int64_t sample_numel, int64_t blocks_per_sample, int64_t total_blocks) {
  constexpr int W = ALIGNED ? 8 : 4;
  auto body = [&](uint64_t seed, uint64_t blk0, int64_t e0, int64_t limit) {
    float to[W], from[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { to[j] = 0.f; from[j] = 0.f; }
    for (int k = 0; k < a.n; ++k) {
      float z[W];
      skr::normal4(seed, a.stream_ids[k], blk0, z);
      if constexpr (ALIGNED) skr::normal4(seed, a.stream_ids[k], blk0 + 1, z + 4);
      const float wt = a.w_to[k];
#pragma unroll
      for (int j = 0; j < W; ++j) to[j] = __builtin_fmaf(wt, z[j], to[j]);
      if constexpr (!FROM_CACHE) {
        const float wf = a.w_from[k];
#pragma unroll
        for (int j = 0; j < W; ++j) from[j] = __builtin_fmaf(wf, z[j], from[j]);
      }
    }
    if constexpr (ALIGNED) {  // 32 contiguous bytes per lane: explicit 16-byte accesses
      float4* c4 = reinterpret_cast<float4*>(a.cache + e0);
      if constexpr (FROM_CACHE) {
        const float4 lo = c4[0], hi = c4[1];
        from[0] = lo.x; from[1] = lo.y; from[2] = lo.z; from[3] = lo.w; from[4] = hi.x; from[5] = hi.y; from[6] = hi.z; from[7] = hi.w;
      }
      if (a.cache) { c4[0] = make_float4(to[0], to[1], to[2], to[3]); c4[1] = make_float4(to[4], to[5], to[6], to[7]); }
    } else {
      if constexpr (FROM_CACHE) {
#pragma unroll
        for (int j = 0; j < W; ++j) if (j < limit) from[j] = a.cache[e0 + j];
      }
      if (a.cache) {
#pragma unroll
        for (int j = 0; j < W; ++j) if (j < limit) a.cache[e0 + j] = to[j];
      }
    }
    float r[W];
#pragma unroll
    for (int j = 0; j < W; ++j) r[j] = a.scale * (to[j] - from[j]);
    if constexpr (ALIGNED) {
      skr::store8_from_f32<T>(out, e0 >> 3, r);
    } else {
#pragma unroll
      for (int j = 0; j < W; ++j) if (j < limit) out[e0 + j] = (T)r[j];
    }
  };
  if constexpr (ALIGNED) {
    const int64_t smp = blockIdx.y, vps = sample_numel >> 3;
    const uint64_t seed = seeds[smp];
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < vps; v += (int64_t)gridDim.x * 256) body(seed, (uint64_t)(2 * v), smp * sample_numel + 8 * v, 8);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_blocks; i += (int64_t)gridDim.x * 256) {
      const int64_t smp = i / blocks_per_sample, blk = i - smp * blocks_per_sample;
      const int64_t left = sample_numel - blk * 4;
      body(seeds[smp], (uint64_t)blk, smp * sample_numel + blk * 4, left < 4 ? left : 4);
    }
  }
}

template <typename T>
static void launch_brownian(void* out, const uint64_t* seeds, const BrownianArgs& a, int64_t batch, int64_t sample_numel, hipStream_t s) {
  const bool fast = sample_numel % 8 == 0 && batch <= 65535 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.cache) & 15) == 0;
  const int64_t bps = (sample_numel + 3) / 4, total = bps * batch;
  if (fast) {
    int64_t bx = (sample_numel / 8 + 255) / 256;
    const int64_t cap = (256 * 16 + batch - 1) / batch;
    if (bx > cap) bx = cap;
    dim3 grid((unsigned)bx, (unsigned)batch);
    if (a.from_cache) hipLaunchKernelGGL((brownian_kernel<T, true, true>), grid, dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
    else hipLaunchKernelGGL((brownian_kernel<T, true, false>), grid, dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
    return;
  }
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (a.from_cache) hipLaunchKernelGGL((brownian_kernel<T, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
  else hipLaunchKernelGGL((brownian_kernel<T, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)out, seeds, a, sample_numel, bps, total);
}

extern "C" int skr_noise_brownian(void* out, int32_t out_dtype, const uint64_t* seeds_dev, const uint64_t* stream_ids, const double* weights_to,
                                  const double* weights_from, int32_t n_streams, double scale, float* cache_f32, int32_t from_cache,
                                  int64_t batch, int64_t sample_numel, void* stream) {
  skr::DeviceGuard device_guard(out);
  if (batch < 0 || sample_numel < 0 || n_streams < 0) return SKR_ERR_SHAPE;
  if (n_streams > SKR_MAX_WEIGHTED_STREAMS) return SKR_ERR_UNSUPPORTED;
  if (batch == 0 || sample_numel == 0) return SKR_OK;
  if (!out || !seeds_dev || (n_streams > 0 && (!stream_ids || !weights_to))) return SKR_ERR_NULL;
  if (from_cache ? !cache_f32 : (n_streams > 0 && !weights_from)) return SKR_ERR_NULL;
  BrownianArgs a;
  for (int k = 0; k < SKR_MAX_WEIGHTED_STREAMS; ++k) {
    a.stream_ids[k] = k < n_streams ? stream_ids[k] : 0;
    a.w_to[k] = k < n_streams ? (float)weights_to[k] : 0.f;
    a.w_from[k] = (k < n_streams && weights_from) ? (float)weights_from[k] : 0.f;
  }
  a.cache = cache_f32; a.scale = (float)scale; a.n = n_streams; a.from_cache = from_cache ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SKR_BF16: launch_brownian<__bf16>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F16: launch_brownian<_Float16>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F32: launch_brownian<float>(out, seeds_dev, a, batch, sample_numel, s); break;
    case SKR_F64: launch_brownian<double>(out, seeds_dev, a, batch, sample_numel, s); break;
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
  skr::DeviceGuard device_guard(out);
  if (!out) return SKR_ERR_NULL;
  if (n_blocks < 0) return SKR_ERR_SHAPE;
  if (n_blocks == 0) return SKR_OK;
  const int threads = 256;
  const unsigned blocks = (unsigned)((n_blocks + threads - 1) / threads);
  hipLaunchKernelGGL(philox_dump_kernel, dim3(blocks), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), out, seed, stream_id, first_block, n_blocks);
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}
