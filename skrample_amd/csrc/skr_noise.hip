// Structured noise generators for MI355X: Offset and Pyramid (reference skrample/pytorch/noise.py:77-207).
// All randomness is Philox4x32-10 keyed per sample (skr_philox.h); one draw of a generator owns 256
// consecutive stream ids: +0 base normal, +1.. auxiliary normals (offset / pyramid levels), +255 uniforms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/skrample_hip.h"
#include "skr_philox.h"

namespace skr {

template <typename T> __device__ __forceinline__ void put(T* p, int64_t i, float v) { p[i] = (T)v; }

__device__ __forceinline__ float normal1(uint64_t seed, uint64_t stream, uint64_t idx) {
  float z[4];
  normal4(seed, stream, idx >> 2, z);
  return z[idx & 3];
}

// ---- Offset: out = N(base) + strength^2 * N(offset)[reduced index] -------------------------------------
// unit shape up to 4 dims (d0,d1,d2,d3 ; missing leading dims = 1).  `mask` bit k set => dim k keeps its size in
// the reduced (offset) tensor, otherwise it is broadcast.
struct OffsetArgs {
  void* out;
  const uint64_t* seeds;
  uint64_t stream_base, stream_offset;
  int64_t batch, d0, d1, d2, d3;
  uint32_t mask;
  float gain;  // strength^2
};

template <typename T>
__global__ __launch_bounds__(256) void offset_kernel(const OffsetArgs a) {
  const int64_t unit = a.d0 * a.d1 * a.d2 * a.d3;
  const int64_t bps = (unit + 3) / 4, total = bps * a.batch;
  // sizes of the reduced tensor
  const int64_t r0 = (a.mask & 1) ? a.d0 : 1, r1 = (a.mask & 2) ? a.d1 : 1, r2 = (a.mask & 4) ? a.d2 : 1, r3 = (a.mask & 8) ? a.d3 : 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t smp = i / bps, blk = i - smp * bps;
    const uint64_t seed = a.seeds[smp];
    float z[4];
    normal4(seed, a.stream_base, (uint64_t)blk, z);
    int64_t last_ridx = -1;
    float off = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = blk * 4 + j;
      if (e >= unit) break;
      int64_t rem = e;
      const int64_t i3 = rem % a.d3; rem /= a.d3;
      const int64_t i2 = rem % a.d2; rem /= a.d2;
      const int64_t i1 = rem % a.d1; rem /= a.d1;
      const int64_t i0 = rem;
      const int64_t ridx = ((((a.mask & 1) ? i0 : 0) * r1 + ((a.mask & 2) ? i1 : 0)) * r2 + ((a.mask & 4) ? i2 : 0)) * r3 + ((a.mask & 8) ? i3 : 0);
      if (ridx != last_ridx) { off = normal1(seed, a.stream_offset, (uint64_t)ridx); last_ridx = ridx; }
      put<T>((T*)a.out, smp * unit + e, z[j] + off * a.gain);
    }
  }
}

// ---- Pyramid ---------------------------------------------------------------------------------------------
// Per (sample, leading slice) block.  Levels >= 1 are tiny (<= 64x64): their normals are generated straight
// into LDS and bilinearly sampled from there (torch upsample_bilinear2d, align_corners=False); level 0 is
// full resolution, i.e. an identity "interpolation" of a second normal tensor.  Pass 1 writes the
// un-normalised sum in fp32 plus per-block (sum, sum of squares) in double; pass 2 divides by the per-sample
// unbiased std (fixed summation order => bit-reproducible) and rounds to the output dtype.
constexpr int PYR_MAX_LEVELS = 8;
constexpr int PYR_LDS_FLOATS = 38 * 1024;  // up to 152 KiB of level storage per block (dynamic LDS, 160 KiB per CU)

struct PyramidArgs {
  float* scratch;          // [batch][lead][h][w] fp32
  double* partials;        // [batch][lead][2]
  const uint64_t* seeds;
  const int32_t* level_hw; // [batch][PYR_MAX_LEVELS][2] (h_l, w_l); level 0 is (h, w)
  const int32_t* n_levels; // [batch]
  uint64_t stream_base;
  int64_t batch, lead, h, w;
  float weight[PYR_MAX_LEVELS];  // strength^l, 0 for skipped levels
  int32_t with_base;       // 1: add the base normal (stream_base + 0)
};

__device__ __forceinline__ void src_index(int dst, int in_size, int out_size, int& i0, int& i1, float& l1) {
  // area_pixel_compute_source_index(scale = in/out, align_corners = false, cubic = false)
  const float scale = (float)in_size / (float)out_size;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void pyramid_pass1(const PyramidArgs a) {
  extern __shared__ float lds[];  // sum over levels >= 1 of h_l*w_l floats (sized by the host per launch)
  __shared__ double red[2][4];
  const int64_t slice = blockIdx.x;  // smp * lead + c
  const int64_t smp = slice / a.lead, c = slice - smp * a.lead;
  const uint64_t seed = a.seeds[smp];
  const int nl = a.n_levels[smp];
  const int32_t* hw = a.level_hw + smp * PYR_MAX_LEVELS * 2;

  // stage levels >= 1 in LDS
  int base_off[PYR_MAX_LEVELS];
  int off = 0;
  for (int l = 1; l < nl; ++l) {
    base_off[l] = off;
    const int lh = hw[2 * l], lw = hw[2 * l + 1];
    const int n = lh * lw;
    if (a.weight[l] != 0.f) {
      for (int i = threadIdx.x; i < n; i += 256)
        lds[off + i] = normal1(seed, a.stream_base + 1 + l, (uint64_t)(c * n + i));  // level tensor is [lead][lh][lw]
    }
    off += n;
  }
  __syncthreads();

  const int64_t hwn = a.h * a.w;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t p4 = threadIdx.x; p4 * 4 < hwn; p4 += 256) {
    const int64_t e0 = c * hwn + p4 * 4;  // element index inside the sample (hwn % 4 == 0 is checked on the host)
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.with_base) normal4(seed, a.stream_base, (uint64_t)e0 >> 2, v);
    if (a.weight[0] != 0.f) {
      float z[4];
      normal4(seed, a.stream_base + 1, (uint64_t)e0 >> 2, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += z[j] * a.weight[0];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p4 * 4 + j;
      const int y = (int)(p / a.w), x = (int)(p - (int64_t)y * a.w);
      for (int l = 1; l < nl; ++l) {
        if (a.weight[l] == 0.f) continue;
        const int lh = hw[2 * l], lw = hw[2 * l + 1];
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(y, lh, (int)a.h, y0, y1, ly);
        src_index(x, lw, (int)a.w, x0, x1, lx);
        const float* g = lds + base_off[l];
        const float top = (1.f - lx) * g[y0 * lw + x0] + lx * g[y0 * lw + x1];
        const float bot = (1.f - lx) * g[y1 * lw + x0] + lx * g[y1 * lw + x1];
        v[j] += a.weight[l] * ((1.f - ly) * top + ly * bot);
      }
      s1 += (double)v[j];
      s2 += (double)v[j] * (double)v[j];
    }
    *reinterpret_cast<float4*>(a.scratch + (smp * a.lead + c) * hwn + p4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  // block reduction in a fixed order
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partials[slice * 2 + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    a.partials[slice * 2 + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void normalise_pass2(T* out, const float* scratch, const double* partials, int64_t lead, int64_t unit, int64_t batch, double target /* <0: unit std */) {
  const int64_t smp = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t c = 0; c < lead; ++c) { s1 += partials[(smp * lead + c) * 2]; s2 += partials[(smp * lead + c) * 2 + 1]; }
  const double n = (double)unit;
  const double var = (s2 - s1 * s1 / n) / (n - 1.0);  // unbiased, as torch.std
  const float inv = (float)(1.0 / sqrt(var));
  const float* src = scratch + smp * unit;
  T* dst = out + smp * unit;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < unit; i += (int64_t)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    dst[i] = (T)(v.x * inv); dst[i + 1] = (T)(v.y * inv); dst[i + 2] = (T)(v.z * inv); dst[i + 3] = (T)(v.w * inv);
  }
}

}  // namespace skr

static int status_of_launch() { return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH; }

extern "C" int skr_noise_offset(void* out, int32_t out_dtype, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_offset,
                                int64_t batch, const int64_t* unit_shape, int32_t ndim, uint32_t keep_mask, double strength, void* stream) {
  if (batch < 0 || ndim < 1 || ndim > 4 || !unit_shape) return SKR_ERR_SHAPE;
  skr::OffsetArgs a;
  int64_t d[4] = {1, 1, 1, 1};
  for (int i = 0; i < ndim; ++i) { if (unit_shape[i] < 0) return SKR_ERR_SHAPE; d[4 - ndim + i] = unit_shape[i]; }
  a.d0 = d[0]; a.d1 = d[1]; a.d2 = d[2]; a.d3 = d[3];
  const int64_t unit = d[0] * d[1] * d[2] * d[3];
  if (batch == 0 || unit == 0) return SKR_OK;
  if (!out || !seeds_dev) return SKR_ERR_NULL;
  a.out = out; a.seeds = seeds_dev; a.stream_base = stream_base; a.stream_offset = stream_offset; a.batch = batch;
  a.mask = (keep_mask & ((1u << ndim) - 1u)) << (4 - ndim);
  a.gain = (float)(strength * strength);
  const int64_t total = ((unit + 3) / 4) * batch;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(skr::offset_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    case SKR_F16: hipLaunchKernelGGL(skr::offset_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    case SKR_F32: hipLaunchKernelGGL(skr::offset_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    case SKR_F64: hipLaunchKernelGGL(skr::offset_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    default: return SKR_ERR_DTYPE;
  }
  return status_of_launch();
}

extern "C" int skr_noise_pyramid(void* out, int32_t out_dtype, float* scratch_f32, double* partials_f64, const uint64_t* seeds_dev,
                                 uint64_t stream_base, int64_t batch, int64_t lead, int64_t h, int64_t w,
                                 const int32_t* level_hw_dev, const int32_t* n_levels_dev, const double* level_weight /* host [8] */,
                                 int32_t max_level_elems, int32_t with_base, int32_t normalise, void* stream) {
  if (batch < 0 || lead < 1 || h < 1 || w < 1) return SKR_ERR_SHAPE;
  if (batch == 0) return SKR_OK;
  if (!out || !scratch_f32 || !partials_f64 || !seeds_dev || !level_hw_dev || !n_levels_dev || !level_weight) return SKR_ERR_NULL;
  if ((h * w) % 4 != 0) return SKR_ERR_UNSUPPORTED;
  if (max_level_elems > skr::PYR_LDS_FLOATS) return SKR_ERR_UNSUPPORTED;  // sum of level >= 1 sizes must fit the LDS stage
  if (batch * lead > 0x7fffffffll || batch > 65535) return SKR_ERR_UNSUPPORTED;
  skr::PyramidArgs a;
  a.scratch = scratch_f32; a.partials = partials_f64; a.seeds = seeds_dev; a.level_hw = level_hw_dev; a.n_levels = n_levels_dev;
  a.stream_base = stream_base; a.batch = batch; a.lead = lead; a.h = h; a.w = w; a.with_base = with_base;
  for (int l = 0; l < skr::PYR_MAX_LEVELS; ++l) a.weight[l] = (float)level_weight[l];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds_bytes = sizeof(float) * (size_t)(max_level_elems > 0 ? max_level_elems : 1);
  if (lds_bytes > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(skr::pyramid_pass1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return SKR_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(skr::pyramid_pass1, dim3((unsigned)(batch * lead)), dim3(256), lds_bytes, s, a);
  if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
  const int64_t unit = lead * h * w;
  int64_t bx = (unit / 4 + 255) / 256;
  if (bx > 64) bx = 64;
  const double target = normalise ? -1.0 : 0.0;
  (void)target;
  dim3 grid((unsigned)bx, (unsigned)batch);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(skr::normalise_pass2<__bf16>, grid, dim3(256), 0, s, (__bf16*)out, scratch_f32, partials_f64, lead, unit, batch, -1.0); break;
    case SKR_F16: hipLaunchKernelGGL(skr::normalise_pass2<_Float16>, grid, dim3(256), 0, s, (_Float16*)out, scratch_f32, partials_f64, lead, unit, batch, -1.0); break;
    case SKR_F32: hipLaunchKernelGGL(skr::normalise_pass2<float>, grid, dim3(256), 0, s, (float*)out, scratch_f32, partials_f64, lead, unit, batch, -1.0); break;
    case SKR_F64: hipLaunchKernelGGL(skr::normalise_pass2<double>, grid, dim3(256), 0, s, (double*)out, scratch_f32, partials_f64, lead, unit, batch, -1.0); break;
    default: return SKR_ERR_DTYPE;
  }
  return status_of_launch();
}
