// Structured noise generators for MI355X: Offset and Pyramid (reference skrample/pytorch/noise.py:77-207).
// All randomness is Philox4x32-10 keyed per sample (skr_philox.h); one draw of a generator owns 256
// consecutive stream ids: +0 base normal, +1.. auxiliary normals (offset / pyramid levels), +255 uniforms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "skr_device.h"
#include "../../include/skrample_hip.h"
#include "skr_philox.h"
#include "skr_pack.h"

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

// torch evaluates `noise + offset * strength**2` as a rounded multiply, then a rounded add: no FMA contraction here
// (hip's __fadd_rn is a plain `+` and would still be contracted with a neighbouring multiply)
__device__ __forceinline__ float add_offset(float z, float off, float gain) {
#pragma clang fp contract(off)
  const float scaled = off * gain;
  return z + scaled;
}

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
      put<T>((T*)a.out, smp * unit + e, add_offset(z[j], off, a.gain));
    }
  }
}

// Aligned fast path: d3 % 8 == 0, unit < 2^31, 16-byte aligned base.  blockIdx.y = sample; a thread owns 8
// consecutive elements of one innermost row, so the index decomposition is done once (32-bit) and the offset
// normal is drawn once per thread unless the innermost axis itself is kept, in which case the 8 reduced indices
// are consecutive and cost two Philox blocks.  Same per-element arithmetic as the kernel above.
template <typename T>
__global__ __launch_bounds__(256) void offset_kernel_v8(const OffsetArgs a) {
  const uint32_t d1 = (uint32_t)a.d1, d2 = (uint32_t)a.d2, d3 = (uint32_t)a.d3;
  const uint32_t unit = (uint32_t)(a.d0 * a.d1 * a.d2 * a.d3), vps = unit >> 3;
  const uint32_t r1 = (a.mask & 2) ? d1 : 1, r2 = (a.mask & 4) ? d2 : 1, r3 = (a.mask & 8) ? d3 : 1;
  const int64_t smp = blockIdx.y;
  const uint64_t seed = a.seeds[smp];
  T* dst = (T*)a.out + smp * (int64_t)unit;
  // few distinct offsets per sample (the default: one per channel): the block draws them once into LDS instead of one Philox
  // block per thread and trip -- a third of the kernel's Philox work (same values: the same function of the same index)
  __shared__ float offs[256];
  const uint32_t n_off = ((a.mask & 1) ? (uint32_t)a.d0 : 1u) * r1 * r2 * r3;
  const bool staged = !(a.mask & 8) && n_off <= 256u;
  if (staged) {
    if (threadIdx.x < n_off) offs[threadIdx.x] = normal1(seed, a.stream_offset, (uint64_t)threadIdx.x);
    __syncthreads();
  }
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < vps; v += gridDim.x * 256) {
    float z[8];
    normal4(seed, a.stream_base, (uint64_t)(2 * v), z);
    normal4(seed, a.stream_base, (uint64_t)(2 * v) + 1, z + 4);
    uint32_t rem = v * 8;
    const uint32_t i3 = rem % d3; rem /= d3;
    const uint32_t i2 = rem % d2; rem /= d2;
    const uint32_t i1 = rem % d1;
    const uint32_t i0 = rem / d1;
    const uint64_t ridx = (uint64_t)((((a.mask & 1) ? i0 : 0) * r1 + ((a.mask & 2) ? i1 : 0)) * r2 + ((a.mask & 4) ? i2 : 0)) * r3 + ((a.mask & 8) ? i3 : 0);
    if (a.mask & 8) {
      // i3 % 8 == 0 and r3 = d3 % 8 == 0 => ridx % 8 == 0: two whole Philox blocks
      float o[8];
      normal4(seed, a.stream_offset, ridx >> 2, o);
      normal4(seed, a.stream_offset, (ridx >> 2) + 1, o + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = add_offset(z[j], o[j], a.gain);
    } else {
      const float off = staged ? offs[(uint32_t)ridx] : normal1(seed, a.stream_offset, ridx);
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = add_offset(z[j], off, a.gain);
    }
    store8_from_f32<T>(dst, (int64_t)v, z);
  }
}

template <typename T>
static void launch_offset(const OffsetArgs& a, hipStream_t s) {
  const int64_t unit = a.d0 * a.d1 * a.d2 * a.d3;
  const bool fast = a.d3 % 8 == 0 && unit < (1ll << 31) && a.batch <= 65535 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
  if (fast) {
    int64_t bx = (unit / 8 + 255) / 256;
    const int64_t cap = (256 * 16 + a.batch - 1) / a.batch;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(offset_kernel_v8<T>, dim3((unsigned)bx, (unsigned)a.batch), dim3(256), 0, s, a);
    return;
  }
  const int64_t total = ((unit + 3) / 4) * a.batch;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(offset_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ---- Pyramid ---------------------------------------------------------------------------------------------
// Per (sample, leading slice) block.  Levels >= 1 are small (<= half size per axis): their normals are generated
// straight into LDS and bilinearly sampled from there (torch upsample_bilinear2d, align_corners=False);
// level 0 is full resolution, i.e. an identity "interpolation" of a second normal tensor.  Pass 0 draws the
// per-sample level geometry on the device (no host round trip); pass 1 writes the un-normalised sum in fp32
// plus per-block (sum, sum of squares) in double; pass 2 divides by the per-sample unbiased std (fixed
// summation order => bit-reproducible) and rounds to the output dtype.
constexpr int PYR_MAX_LEVELS = 8;
constexpr int PYR_LDS_FLOATS = 38 * 1024;  // up to 152 KiB of level storage per block (dynamic LDS, 160 KiB per CU)

struct PyramidArgs {
  float* scratch;          // [batch][lead][h][w] fp32
  double* partials;        // [batch][lead][2]
  const uint64_t* seeds;
  int32_t* level_hw;       // [batch][PYR_MAX_LEVELS][2] (h_l, w_l); level 0 is (h, w)
  int32_t* n_levels;       // [batch]
  uint64_t stream_base;    // base normal: stream_base + 0
  uint64_t stream_levels;  // level l normal: stream_levels + 1 + l, geometry uniforms: stream_levels + 255
  int64_t batch;
  int32_t lead, h, w;
  int32_t resize_h, depth, with_base;
  float strength;
  // any-dims path (pyramid_pass1_any): the unit as up to four axes, h = dim[axis_a] (axis_a = -1: one resized axis),
  // w = dim[axis_b], lead = product of the other axes.  Level tensors keep the unit's axis ORDER (reference
  // noise.py:162-165 draws `randn(running_shape)` before permuting), so level l, element (i0..i3) is normal number
  // sum_j i_j * stride_l[j] of that level's stream.
  int32_t nd, dim[4], axis_a, axis_b;
  int32_t ytab_off;  // UNI kernels: float offset of the vertical tap table inside the dynamic LDS (behind the level planes)
};

// level geometry (reference noise.py:157-162,195-196): level i shrinks the RUNNING size by r_i**i,
// r_i = 2 + 2*u_i, u_i = (philox word i of stream_base+255 >> 8) * 2^-24; stop at the first level with a
// resized dimension of 1.  Same arithmetic as skrample_amd/pytorch/noise.py::pyramid_level_tables.
// shrink factor of level i: r_i ** i, r_i = 2 + 2 u_i from word i of the geometry stream (Philox block i / 4)
__device__ __forceinline__ double pyramid_level_shrink(const PyramidArgs& a, uint64_t seed, int i) {
  u32x4 c{(uint32_t)(i >> 2), 0u, (uint32_t)(a.stream_levels + 255), (uint32_t)((a.stream_levels + 255) >> 32)};
  c = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t word = (i & 3) == 0 ? c.x : ((i & 3) == 1 ? c.y : ((i & 3) == 2 ? c.z : c.w));
  const double r = (double)(word >> 8) * 5.9604644775390625e-08 * 2.0 + 2.0;
  // r ** i, i < 8, by squaring in double-double arithmetic (error 2^-100 before the final rounding: the correctly rounded power, which
  // is what the C library's pow returns to Python); the general pow() was 1 us of every block's start for an integer exponent below 8
  double h = 1.0, l = 0.0, bh = r, bl = 0.0;
  auto mul = [](double& xh, double& xl, double yh, double yl) {
    const double p = xh * yh;
    double e = fma(xh, yh, -p);
    e = fma(xh, yl, e);
    e = fma(xl, yh, e);
    const double t = p + e;
    xl = e - (t - p);
    xh = t;
  };
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (i & (1 << k)) mul(h, l, bh, bl);
    mul(bh, bl, bh, bl);
  }
  return h;
}
// the running sizes from the eight shrink factors: (h_l, w_l) into hw[2 l], hw[2 l + 1]; returns the number of levels
template <typename HW>
__device__ __forceinline__ int pyramid_level_walk(const PyramidArgs& a, const double* shrink, HW&& put) {
  int64_t h = a.h, w = a.w;
  int n = 0;
  for (int i = 0; i < PYR_MAX_LEVELS; ++i) {
    if (a.resize_h) { h = (int64_t)((double)h / shrink[i]); if (h < 1) h = 1; }
    w = (int64_t)((double)w / shrink[i]); if (w < 1) w = 1;
    put(i, (int32_t)h, (int32_t)w);
    n = i + 1;
    if (w <= 1 || (a.resize_h && h <= 1)) break;
  }
  return n;
}

__global__ void pyramid_geometry(const PyramidArgs a) {  // (the any-shape kernels: their launches are sized from nothing but need the table)
  const int64_t smp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (smp >= a.batch) return;
  const uint64_t seed = a.seeds[smp];
  double shrink[PYR_MAX_LEVELS];
  for (int i = 0; i < PYR_MAX_LEVELS; ++i) shrink[i] = pyramid_level_shrink(a, seed, i);
  a.n_levels[smp] = pyramid_level_walk(a, shrink, [&](int i, int32_t h, int32_t w) {
    a.level_hw[(smp * PYR_MAX_LEVELS + i) * 2] = h;
    a.level_hw[(smp * PYR_MAX_LEVELS + i) * 2 + 1] = w;
  });
}

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  // area_pixel_compute_source_index(scale = in/out, align_corners = false, cubic = false)
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

constexpr int PYR_UNROLLED = 5;  // levels 1..4 are unrolled (and cached per column strip); deeper levels are rare and tiny
constexpr int PYR_THREADS = 512;  // 8 waves per block (the strip kernel's 165 registers allow one such block per CU; 128 registers with spills were no faster)

// STRIP: the block width divides THREADS, so a thread keeps its 4 columns for a RUN of consecutive rows: the
// horizontally interpolated level rows (top / bottom) stay in registers while the coarse source row does not change
// and slide (bottom -> top) when it advances by one.
// THREADS: 512 for large planes; 256 for small ones, where it doubles the rows of a run.
// UNI (round 5; strips whose rows are a whole number of waves wide, w % 256 == 0): the row a wave works on, and with it every vertical tap
// (source rows y0 / y1 and the blend weight ly of each level), is the same in all 64 lanes.  Those taps come from a table in LDS, filled once
// per block (one src_index per thread), as one broadcast read per level and row, and live in SGPRs from there: the row-cache tests are scalar
// compares, ly is a scalar operand.  The cached source rows are kept as T = wl top and D = wl (bottom - top), so a pixel costs an add and
// an FMA per level (was three FMA-class operations after nine VALU operations of tap arithmetic per level and row), and the per-row sums
// stay in fp32 for eight rows before they are widened.  Measured on 64 x (4, 256, 256): see profiles/r05_prof_pyramid.txt.
template <bool STRIP, int THREADS, bool UNI = false>
__global__ __launch_bounds__(THREADS, THREADS >= 1024 ? 1 : (STRIP && THREADS > 256 ? 2 : 4)) void pyramid_pass1(const PyramidArgs a) {
  extern __shared__ float lds[];  // levels >= 1, back to back
  __shared__ double red[2][THREADS / 64];
  __shared__ int s_lh[PYR_MAX_LEVELS], s_lw[PYR_MAX_LEVELS], s_off[PYR_MAX_LEVELS];
  __shared__ float s_wgt[PYR_MAX_LEVELS], s_sy[PYR_MAX_LEVELS], s_sx[PYR_MAX_LEVELS];
  const int slice = blockIdx.x;  // smp * lead + c
  const int smp = slice / a.lead, c = slice - smp * a.lead;
  const uint64_t seed = a.seeds[smp];
  // Level geometry in the block itself (round 3; a launch of its own before: 5 us of a 76 us draw for eight numbers per sample):
  // lane i draws word i and raises r_i to the i-th power, lane 0 walks the running sizes.  Same functions, same values.
  __shared__ double s_shrink[PYR_MAX_LEVELS];
  __shared__ int s_nl;
  // Strip kernels: the base and level-0 normals of a thread's first PYR_AHEAD rows are drawn NOW -- they need nothing of the level
  // geometry, and while eight lanes work that out (3.7 us of a 51 us block, then 3.8 us of level planes) every other lane is idle.
  constexpr int PYR_AHEAD = STRIP ? 2 : 0;  // (4 rows: 130 spilled registers in the 1024-lane kernel, 79-82 us per cfg5 draw against 69)
  float ahead_v[PYR_AHEAD ? PYR_AHEAD : 1][4], ahead_z[PYR_AHEAD ? PYR_AHEAD : 1][4];
  if constexpr (STRIP) {
    const int w4s = a.w >> 2, xs = (threadIdx.x % w4s) * 4, gs = THREADS / w4s, rs = (a.h + gs - 1) / gs, ys = (threadIdx.x / w4s) * rs;
#pragma unroll
    for (int r = 0; r < PYR_AHEAD; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ahead_v[r][j] = 0.f; ahead_z[r][j] = 0.f; }
      if (ys + r < a.h && r < rs) {
        const int64_t e0 = ((int64_t)c * a.h + ys + r) * a.w + xs;
        if (a.with_base) normal4(seed, a.stream_base, (uint64_t)e0 >> 2, ahead_v[r]);
        normal4(seed, a.stream_levels + 1, (uint64_t)e0 >> 2, ahead_z[r]);
      }
    }
  }
  if (threadIdx.x < PYR_MAX_LEVELS) s_shrink[threadIdx.x] = pyramid_level_shrink(a, seed, (int)threadIdx.x);
  __syncthreads();
  if (threadIdx.x < 2) {  // the running sizes (pyramid_level_walk's arithmetic), widths in lane 0 and heights in lane 1
    const bool heights = threadIdx.x == 1;
    int64_t d = heights ? a.h : a.w;
    int* sizes = heights ? s_lh : s_lw;
    for (int i = 0; i < PYR_MAX_LEVELS; ++i) {
      if (!heights || a.resize_h) { d = (int64_t)((double)d / s_shrink[i]); if (d < 1) d = 1; }
      sizes[i] = (int)d;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int nl = 0;
    for (int i = 0; i < PYR_MAX_LEVELS; ++i) { nl = i + 1; if (s_lw[i] <= 1 || (a.resize_h && s_lh[i] <= 1)) break; }
    s_nl = nl;
    if (c == 0) {  // (the table stays readable on the host side: tests compare it with the specification)
      for (int l = 0; l < nl; ++l) { a.level_hw[((int64_t)smp * PYR_MAX_LEVELS + l) * 2] = s_lh[l]; a.level_hw[((int64_t)smp * PYR_MAX_LEVELS + l) * 2 + 1] = s_lw[l]; }
      a.n_levels[smp] = nl;
    }
    const int skip = (nl - 1) - a.depth > 0 ? (nl - 1) - a.depth : 0;  // keep the depth+1 coarsest levels (noise.py:198-200)
    int off = 0;
    float wgt = 1.f;
    for (int l = 0; l < nl; ++l) {
      s_off[l] = off;
      s_wgt[l] = l >= skip ? wgt : 0.f;
      s_sy[l] = (float)s_lh[l] / (float)a.h;
      s_sx[l] = (float)s_lw[l] / (float)a.w;
      if (l >= 1) off += s_lh[l] * s_lw[l];
      wgt *= a.strength;
    }
  }
  __syncthreads();
  const int nl = s_nl;
  for (int l = 1; l < nl; ++l) {
    if (s_wgt[l] == 0.f) continue;
    const int n = s_lh[l] * s_lw[l];
    float* g = lds + s_off[l];
    // level tensor is [lead][lh][lw]: this plane owns elements [c n, (c + 1) n) of the level's stream, element e = normal e % 4 of Philox
    // block e / 4.  One thread per BLOCK (not per 4 elements of the plane: a plane that starts inside a block would draw every block twice)
    const int64_t e_lo = (int64_t)c * n, b_hi = (e_lo + n + 3) >> 2;
    for (int64_t b = (e_lo >> 2) + threadIdx.x; b < b_hi; b += THREADS) {
      float z[4];
      normal4(seed, a.stream_levels + 1 + l, (uint64_t)b, z);
      const int at = (int)((b << 2) - e_lo);  // -3 .. n - 1
#pragma unroll
      for (int j = 0; j < 4; ++j) if (at + j >= 0 && at + j < n) g[at + j] = z[j];
    }
  }
  if constexpr (UNI) {  // vertical taps of the cached levels for every row: (ly, y0 | (y1 - y0) << 16)
    float2* ytab = reinterpret_cast<float2*>(lds + a.ytab_off);
    for (int idx = threadIdx.x; idx < (PYR_UNROLLED - 1) * a.h; idx += THREADS) {
      const int l = 1 + idx / a.h, y = idx - (l - 1) * a.h;
      int y0 = 0, y1 = 0;
      float ly = 0.f;
      if (l < nl) src_index(y, s_sy[l], s_lh[l], y0, y1, ly);
      ytab[idx] = make_float2(ly, __int_as_float(y0 | ((y1 - y0) << 16)));
    }
  }
  __syncthreads();

  const int w4 = a.w >> 2;  // w % 4 == 0 (host-checked): 4 consecutive pixels never straddle a row
  const int n4 = a.h * w4;
  const float w0 = s_wgt[0];
  double s1 = 0.0, s2 = 0.0;
  float* const dst = a.scratch + (int64_t)slice * a.h * a.w;

  // one group of 4 pixels at row y, columns x0..x0+3.  X(l, j, xa, xb, lx) yields the horizontal source taps.
  auto pixel_group = [&](int y, int x0, int q, auto&& taps) {
    const int64_t e0 = ((int64_t)c * a.h + y) * a.w + x0;  // element index inside the sample
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.with_base) normal4(seed, a.stream_base, (uint64_t)e0 >> 2, v);
    if (w0 != 0.f) {
      float z[4];
      normal4(seed, a.stream_levels + 1, (uint64_t)e0 >> 2, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(z[j], w0, v[j]);
    }
    auto add_level = [&](int l) {
      const float wl = s_wgt[l];
      if (wl == 0.f) return;
      const int lh = s_lh[l], lw = s_lw[l];
      const float* g = lds + s_off[l];
      int y0, y1;
      float ly;
      src_index(y, s_sy[l], lh, y0, y1, ly);
      const float* r0 = g + y0 * lw;
      const float* r1 = g + y1 * lw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int xa, xb;
        float lx;
        taps(l, j, xa, xb, lx);
        const float top = (1.f - lx) * r0[xa] + lx * r0[xb];
        const float bot = (1.f - lx) * r1[xa] + lx * r1[xb];
        v[j] += wl * ((1.f - ly) * top + ly * bot);
      }
    };
    // the first levels unrolled (their taps may sit in registers: compile-time l), deeper ones in a rolled loop
#pragma unroll
    for (int l = 1; l < PYR_UNROLLED; ++l)
      if (l < nl) add_level(l);
#pragma unroll 1
    for (int l = PYR_UNROLLED; l < nl; ++l) add_level(l);
    const float p1 = (v[0] + v[1]) + (v[2] + v[3]);  // 4 values in fp32, then one widening add: the per-slice totals stay double
    const float p2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
    s1 += (double)p1; s2 += (double)p2;
    *reinterpret_cast<float4*>(dst + (int64_t)q * 4) = make_float4(v[0], v[1], v[2], v[3]);
  };

  if constexpr (STRIP && UNI) {
    constexpr int NC = PYR_UNROLLED;  // levels 1 .. NC - 1 are cached
    const int xg = threadIdx.x % w4, x0 = xg * 4, groups = THREADS / w4;
    const int run = (a.h + groups - 1) / groups;
    const int ya = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / w4) * run), yz = ya + run < a.h ? ya + run : a.h;
    const float2* ytab = reinterpret_cast<const float2*>(lds + a.ytab_off);
    int tap_idx[NC][4];
    float tap_lx[NC][4], T[NC][4], D[NC][4];
    int cy0[NC], cy1[NC];
#pragma unroll
    for (int l = 1; l < NC; ++l) {
      cy0[l] = -1; cy1[l] = -1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int xa = 0, xb = 0;
        float lx = 0.f;
        if (l < nl) src_index(x0 + j, s_sx[l], s_lw[l], xa, xb, lx);
        tap_idx[l][j] = xa | (xb << 16);
        tap_lx[l][j] = lx;
        T[l][j] = 0.f; D[l][j] = 0.f;
      }
    }
    float f1 = 0.f, f2 = 0.f;
    auto row = [&](const int y, const float* drawn_v, const float* drawn_z) {
      const int64_t e0 = ((int64_t)c * a.h + y) * a.w + x0;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (drawn_v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drawn_v[j];
      } else if (a.with_base) normal4(seed, a.stream_base, (uint64_t)e0 >> 2, v);
      if (w0 != 0.f) {
        float z[4];
        if (drawn_z) {
#pragma unroll
          for (int j = 0; j < 4; ++j) z[j] = drawn_z[j];
        } else normal4(seed, a.stream_levels + 1, (uint64_t)e0 >> 2, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(z[j], w0, v[j]);
      }
#pragma unroll
      for (int l = 1; l < NC; ++l) {
        if (l >= nl) break;
        const float wl = s_wgt[l];
        if (wl == 0.f) continue;
        const float2 e = ytab[(l - 1) * a.h + y];  // one address for the whole wave: a broadcast
        const float ly = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e.x)));
        const int pk = __builtin_amdgcn_readfirstlane(__float_as_int(e.y));
        const int y0 = pk & 0xFFFF, y1 = y0 + (pk >> 16);
        const float* g = lds + s_off[l];
        const int lw = s_lw[l];
        auto row_taps = [&](int r_, float* dstv) {
          const float* r = g + r_ * lw;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lx = tap_lx[l][j];
            dstv[j] = (1.f - lx) * r[tap_idx[l][j] & 0xFFFF] + lx * r[tap_idx[l][j] >> 16];
          }
        };
        bool fresh = false;
        if (y0 != cy0[l]) {
          if (y0 == cy1[l]) {  // the bottom row becomes the top row: wl top' = wl top + wl (bottom - top), one rounding away from wl * bottom
#pragma unroll
            for (int j = 0; j < 4; ++j) T[l][j] += D[l][j];
          } else {
            float t[4];
            row_taps(y0, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) T[l][j] = wl * t[j];
          }
          cy0[l] = y0;
          fresh = true;
        }
        if (fresh || y1 != cy1[l]) {
          if (y1 == y0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) D[l][j] = 0.f;
          } else {
            float b[4];
            row_taps(y1, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) D[l][j] = fmaf(wl, b[j], -T[l][j]);
          }
          cy1[l] = y1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(ly, D[l][j], v[j] + T[l][j]);
      }
#pragma unroll 1
      for (int l = NC; l < nl; ++l) {  // deeper levels: rare and tiny, sampled directly
        const float wl = s_wgt[l];
        if (wl == 0.f) continue;
        const int lh = s_lh[l], lw = s_lw[l];
        const float* g = lds + s_off[l];
        int y0, y1;
        float ly;
        src_index(y, s_sy[l], lh, y0, y1, ly);
        const float* r0 = g + y0 * lw;
        const float* r1 = g + y1 * lw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int xa, xb;
          float lx;
          src_index(x0 + j, s_sx[l], lw, xa, xb, lx);
          const float tp = (1.f - lx) * r0[xa] + lx * r0[xb];
          const float bt = (1.f - lx) * r1[xa] + lx * r1[xb];
          v[j] += wl * ((1.f - ly) * tp + ly * bt);
        }
      }
      f1 += (v[0] + v[1]) + (v[2] + v[3]);
      f2 += fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
      *reinterpret_cast<float4*>(dst + ((int64_t)y * w4 + xg) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    };
    int y = ya;
#pragma unroll
    for (int r = 0; r < PYR_AHEAD; ++r, ++y)
      if (y < yz) row(y, ahead_v[r], ahead_z[r]);
    for (; y < yz; ++y) {
      row(y, nullptr, nullptr);
      if (((y - ya) & 7) == 7) { s1 += (double)f1; s2 += (double)f2; f1 = 0.f; f2 = 0.f; }  // 32 values per fp32 partial sum
    }
    s1 += (double)f1; s2 += (double)f2;
  } else if constexpr (STRIP) {
    // column strips over a run of consecutive rows: the horizontal taps of the first PYR_CACHED levels (packed
    // xa | xb << 16 and the blend weight) are computed once per thread, and the horizontally interpolated source rows
    // (top / bottom) once per coarse row instead of once per pixel
    // 256-lane blocks serve small planes, whose level 4 is a point or two: without its cache the kernel fits 128 registers and a CU holds
    // 16 waves instead of 12 (cfg3 shape 262 -> 251 us per draw); the 1024-lane kernel was faster WITH the cache and 19 spilled registers
    constexpr int PYR_CACHED = THREADS <= 256 ? PYR_UNROLLED - 1 : PYR_UNROLLED;
    const int xg = threadIdx.x % w4, x0 = xg * 4, groups = THREADS / w4;
    const int run = (a.h + groups - 1) / groups, ya = (threadIdx.x / w4) * run, yz = ya + run < a.h ? ya + run : a.h;
    int tap_idx[PYR_CACHED][4], cy0[PYR_CACHED], cy1[PYR_CACHED];
    float tap_lx[PYR_CACHED][4], top[PYR_CACHED][4], bot[PYR_CACHED][4];
#pragma unroll
    for (int l = 1; l < PYR_CACHED; ++l) {
      cy0[l] = -1; cy1[l] = -1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int xa = 0, xb = 0;
        float lx = 0.f;
        if (l < nl) src_index(x0 + j, s_sx[l], s_lw[l], xa, xb, lx);
        tap_idx[l][j] = xa | (xb << 16);
        tap_lx[l][j] = lx;
        top[l][j] = 0.f; bot[l][j] = 0.f;
      }
    }
    auto row = [&](const int y, const float* drawn_v, const float* drawn_z) {  // drawn_*: this row's normals, drawn ahead (or nullptr)
      const int64_t e0 = ((int64_t)c * a.h + y) * a.w + x0;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (drawn_v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drawn_v[j];
      } else if (a.with_base) normal4(seed, a.stream_base, (uint64_t)e0 >> 2, v);
      if (w0 != 0.f) {
        float z[4];
        if (drawn_z) {
#pragma unroll
          for (int j = 0; j < 4; ++j) z[j] = drawn_z[j];
        } else normal4(seed, a.stream_levels + 1, (uint64_t)e0 >> 2, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(z[j], w0, v[j]);
      }
#pragma unroll
      for (int l = 1; l < PYR_CACHED; ++l) {
        if (l >= nl) break;
        const float wl = s_wgt[l];
        if (wl == 0.f) continue;
        const int lw = s_lw[l];
        const float* g = lds + s_off[l];
        int y0, y1;
        float ly;
        src_index(y, s_sy[l], s_lh[l], y0, y1, ly);
        auto row_taps = [&](int row, float* dstv) {
          const float* r = g + row * lw;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lx = tap_lx[l][j];
            dstv[j] = (1.f - lx) * r[tap_idx[l][j] & 0xFFFF] + lx * r[tap_idx[l][j] >> 16];
          }
        };
        if (y0 != cy0[l]) {
          if (y0 == cy1[l]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) top[l][j] = bot[l][j];
          } else row_taps(y0, top[l]);
          cy0[l] = y0;
        }
        if (y1 != cy1[l]) {
          if (y1 == y0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bot[l][j] = top[l][j];
          } else row_taps(y1, bot[l]);
          cy1[l] = y1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += wl * ((1.f - ly) * top[l][j] + ly * bot[l][j]);
      }
#pragma unroll 1
      for (int l = PYR_CACHED; l < nl; ++l) {  // deeper levels: rare and tiny, sampled directly
        const float wl = s_wgt[l];
        if (wl == 0.f) continue;
        const int lh = s_lh[l], lw = s_lw[l];
        const float* g = lds + s_off[l];
        int y0, y1;
        float ly;
        src_index(y, s_sy[l], lh, y0, y1, ly);
        const float* r0 = g + y0 * lw;
        const float* r1 = g + y1 * lw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int xa, xb;
          float lx;
          src_index(x0 + j, s_sx[l], lw, xa, xb, lx);
          const float tp = (1.f - lx) * r0[xa] + lx * r0[xb];
          const float bt = (1.f - lx) * r1[xa] + lx * r1[xb];
          v[j] += wl * ((1.f - ly) * tp + ly * bt);
        }
      }
      const float p1 = (v[0] + v[1]) + (v[2] + v[3]);
      const float p2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
      s1 += (double)p1; s2 += (double)p2;
      *reinterpret_cast<float4*>(dst + ((int64_t)y * w4 + xg) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    };
    int y = ya;
#pragma unroll
    for (int r = 0; r < PYR_AHEAD; ++r, ++y)
      if (y < yz) row(y, ahead_v[r], ahead_z[r]);
    for (; y < yz; ++y) row(y, nullptr, nullptr);
  } else {
    for (int q = threadIdx.x; q < n4; q += THREADS) {
      const int y = q / w4, x0 = (q - y * w4) * 4;
      pixel_group(y, x0, q, [&](int l, int j, int& xa, int& xb, float& lx) { src_index(x0 + j, s_sx[l], s_lw[l], xa, xb, lx); });
    }
  }
  // block reduction in a fixed order
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int wv = 0; wv < THREADS / 64; ++wv) { t1 += red[0][wv]; t2 += red[1][wv]; }
    a.partials[(int64_t)slice * 2 + 0] = t1;
    a.partials[(int64_t)slice * 2 + 1] = t2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void normalise_pass2(T* out, const float* scratch, const double* partials, int64_t lead, int64_t unit, int64_t batch, double target /* <0: unit std, 0: no rescale */) {
  const int64_t smp = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t c = 0; c < lead; ++c) { s1 += partials[(smp * lead + c) * 2]; s2 += partials[(smp * lead + c) * 2 + 1]; }
  const double n = (double)unit;
  const double var = (s2 - s1 * s1 / n) / (n - 1.0);  // unbiased, as torch.std
  const float inv = target == 0.0 ? 1.0f : (float)(1.0 / sqrt(var));  // target 0: the raw sum (Pyramid.pyramid(), no base)
  const float* src = scratch + smp * unit;
  T* dst = out + smp * unit;
  if ((unit & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < unit; i += (int64_t)gridDim.x * 1024) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      store4_from_f32<T>(dst + i, v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
  } else {  // ragged units (any-shape path)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < unit; i += (int64_t)gridDim.x * 256) put<T>(dst, i, src[i] * inv);
  }
}

// ---- any-shape fallback: levels in global memory ----------------------------------------------------------------
// For planes whose coarse levels do not fit LDS (about h*w > 380^2) or whose width is not a multiple of 4.  Same
// arithmetic per pixel as pyramid_pass1; the level normals are generated once into `levels` ([batch][cap] fp32,
// cap = lead*h*w >= the sum of all level sizes) and sampled from there (L2-resident for the coarse levels).
struct PyramidAnyArgs {
  PyramidArgs p;
  float* levels;
  int64_t cap;
  int32_t n_slots;
};

__device__ __forceinline__ void pyramid_tables(const PyramidArgs& a, int smp, int nl, int* s_lh, int* s_lw, int64_t* s_off, float* s_wgt, float* s_sy, float* s_sx) {
  const int32_t* hw = a.level_hw + (int64_t)smp * PYR_MAX_LEVELS * 2;
  const int skip = (nl - 1) - a.depth > 0 ? (nl - 1) - a.depth : 0;
  int64_t off = 0;
  float wgt = 1.f;
  for (int l = 0; l < nl; ++l) {
    s_lh[l] = hw[2 * l]; s_lw[l] = hw[2 * l + 1];
    s_off[l] = off;
    s_wgt[l] = l >= skip ? wgt : 0.f;
    s_sy[l] = (float)s_lh[l] / (float)a.h;
    s_sx[l] = (float)s_lw[l] / (float)a.w;
    if (l >= 1) off += (int64_t)a.lead * s_lh[l] * s_lw[l];
    wgt *= a.strength;
  }
}

__global__ __launch_bounds__(256) void pyramid_levels_any(const PyramidAnyArgs q) {
  const PyramidArgs& a = q.p;
  __shared__ int s_lh[PYR_MAX_LEVELS], s_lw[PYR_MAX_LEVELS];
  __shared__ int64_t s_off[PYR_MAX_LEVELS];
  __shared__ float s_wgt[PYR_MAX_LEVELS], s_sy[PYR_MAX_LEVELS], s_sx[PYR_MAX_LEVELS];
  const int smp = blockIdx.y;
  const int nl = a.n_levels[smp];
  if (threadIdx.x == 0) pyramid_tables(a, smp, nl, s_lh, s_lw, s_off, s_wgt, s_sy, s_sx);
  __syncthreads();
  const uint64_t seed = a.seeds[smp];
  float* dst = q.levels + (int64_t)smp * q.cap;
  for (int l = 1; l < nl; ++l) {
    if (s_wgt[l] == 0.f) continue;
    const int64_t n = (int64_t)a.lead * s_lh[l] * s_lw[l];  // level tensor [lead][lh][lw], element e = Philox block e/4 lane e%4
    for (int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x; blk * 4 < n; blk += (int64_t)gridDim.x * 256) {
      float z[4];
      normal4(seed, a.stream_levels + 1 + l, (uint64_t)blk, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) if (blk * 4 + j < n) dst[s_off[l] + blk * 4 + j] = z[j];
    }
  }
}

__global__ __launch_bounds__(256) void pyramid_pass1_any(const PyramidAnyArgs q) {
  const PyramidArgs& a = q.p;
  __shared__ double red[2][4];
  __shared__ int s_lh[PYR_MAX_LEVELS], s_lw[PYR_MAX_LEVELS];
  __shared__ int64_t s_off[PYR_MAX_LEVELS];
  __shared__ float s_wgt[PYR_MAX_LEVELS], s_sy[PYR_MAX_LEVELS], s_sx[PYR_MAX_LEVELS];
  const int smp = blockIdx.y;
  const int nl = a.n_levels[smp];
  if (threadIdx.x == 0) pyramid_tables(a, smp, nl, s_lh, s_lw, s_off, s_wgt, s_sy, s_sx);
  __syncthreads();
  const uint64_t seed = a.seeds[smp];
  const int64_t plane = (int64_t)a.h * a.w, unit = (int64_t)a.lead * plane;
  const float* lv = q.levels + (int64_t)smp * q.cap;
  float* dst = a.scratch + (int64_t)smp * unit;
  const float w0 = s_wgt[0];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g * 4 < unit; g += (int64_t)gridDim.x * 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.with_base) normal4(seed, a.stream_base, (uint64_t)g, v);
    if (w0 != 0.f) {
      float z[4];
      normal4(seed, a.stream_levels + 1, (uint64_t)g, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(z[j], w0, v[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = g * 4 + j;
      if (e >= unit) break;
      // coordinates of e in the unit's own axis order; (y, x) = the resized axes, the rest select the slice
      int idx[4];
      {
        int64_t r = e;
#pragma unroll
        for (int ax = 3; ax >= 0; --ax) { const int d = a.dim[ax]; idx[ax] = (int)(r % d); r /= d; }
      }
      int y = 0, x = 0;
#pragma unroll
      for (int ax = 0; ax < 4; ++ax) { if (ax == a.axis_a) y = idx[ax]; if (ax == a.axis_b) x = idx[ax]; }
      for (int l = 1; l < nl; ++l) {
        const float wl = s_wgt[l];
        if (wl == 0.f) continue;
        const int lh = s_lh[l], lw = s_lw[l];
        // row-major strides of this level's tensor (axis_a -> lh, axis_b -> lw, other axes unchanged)
        int64_t stride = 1, base_off = 0, str_a = 0, str_b = 0;
#pragma unroll
        for (int ax = 3; ax >= 0; --ax) {
          if (ax == a.axis_b) { str_b = stride; stride *= lw; }
          else if (ax == a.axis_a) { str_a = stride; stride *= lh; }
          else { base_off += idx[ax] * stride; stride *= a.dim[ax]; }
        }
        const float* gl = lv + s_off[l] + base_off;
        int y0, y1, xa, xb;
        float ly, lx;
        src_index(y, s_sy[l], lh, y0, y1, ly);
        src_index(x, s_sx[l], lw, xa, xb, lx);
        const float* r0 = gl + (int64_t)y0 * str_a;
        const float* r1 = gl + (int64_t)y1 * str_a;
        const float top = (1.f - lx) * r0[xa * str_b] + lx * r0[xb * str_b];
        const float bot = (1.f - lx) * r1[xa * str_b] + lx * r1[xb * str_b];
        v[j] += wl * ((1.f - ly) * top + ly * bot);
      }
      dst[e] = v[j];
      s1 += (double)v[j]; s2 += (double)v[j] * (double)v[j];
    }
  }
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int wv = 0; wv < 4; ++wv) { t1 += red[0][wv]; t2 += red[1][wv]; }
    a.partials[((int64_t)smp * q.n_slots + blockIdx.x) * 2 + 0] = t1;
    a.partials[((int64_t)smp * q.n_slots + blockIdx.x) * 2 + 1] = t2;
  }
}

}  // namespace skr

static int status_of_launch() { return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH; }

extern "C" int skr_noise_offset(void* out, int32_t out_dtype, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_offset,
                                int64_t batch, const int64_t* unit_shape, int32_t ndim, uint32_t keep_mask, double strength, void* stream) {
  skr::DeviceGuard device_guard(out);
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
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SKR_BF16: skr::launch_offset<__bf16>(a, s); break;
    case SKR_F16: skr::launch_offset<_Float16>(a, s); break;
    case SKR_F32: skr::launch_offset<float>(a, s); break;
    case SKR_F64: skr::launch_offset<double>(a, s); break;
    default: return SKR_ERR_DTYPE;
  }
  return status_of_launch();
}

extern "C" int skr_noise_pyramid(void* out, int32_t out_dtype, float* scratch_f32, double* partials_f64, int32_t* level_ws /* [batch*17] */,
                                 const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch, int64_t lead, int64_t h, int64_t w,
                                 int32_t resize_h, double strength, int32_t depth, int32_t with_base, void* stream) {
  skr::DeviceGuard device_guard(out);
  if (batch < 0 || lead < 1 || h < 1 || w < 1 || depth < 0) return SKR_ERR_SHAPE;
  if (batch == 0) return SKR_OK;
  if (!out || !scratch_f32 || !partials_f64 || !seeds_dev || !level_ws) return SKR_ERR_NULL;
  if (w % 4 != 0 || h > 32767 || w > 32767) return SKR_ERR_UNSUPPORTED;
  if (!resize_h && h != 1) return SKR_ERR_SHAPE;
  if (batch * lead > 0x7fffffffll || batch > 65535) return SKR_ERR_UNSUPPORTED;
  // LDS stage for levels >= 1: every level is at most half the previous size per resized axis (r >= 2, and
  // level i >= 2 shrinks by r^i >= 4), so sum_{l>=1} h_l*w_l <= h*w/4 * (1 + 1/16 + ...) (1-D: w/2 * (1 + 1/4 + ...))
  const int64_t bound = resize_h ? (h / 2) * (w / 2) + (h / 8) * (w / 8) + (h / 32) * (w / 32) + 64 : w / 2 + w / 8 + w / 32 + 64;
  if (bound > skr::PYR_LDS_FLOATS) return SKR_ERR_UNSUPPORTED;
  skr::PyramidArgs a;
  a.scratch = scratch_f32; a.partials = partials_f64; a.seeds = seeds_dev;
  a.level_hw = level_ws; a.n_levels = level_ws + batch * skr::PYR_MAX_LEVELS * 2;
  a.stream_base = stream_base; a.stream_levels = stream_levels; a.batch = batch; a.lead = (int32_t)lead; a.h = (int32_t)h; a.w = (int32_t)w;
  a.resize_h = resize_h; a.depth = depth; a.with_base = with_base; a.strength = (float)strength;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  size_t lds_bytes = sizeof(float) * (size_t)bound;  // (the level geometry is worked out inside pass 1)
  // rows a whole number of waves wide (w % 256 == 0) under 1024-lane strips: the vertical taps of the cached levels in a table behind the level planes
  static const bool no_uni = getenv("SKR_PYR_NO_UNI") != nullptr;
  const size_t ytab_bytes = sizeof(float2) * (size_t)(skr::PYR_UNROLLED - 1) * (size_t)h;
  const bool uni = !no_uni && resize_h && (w / 4) % 64 == 0 && 1024 % (w / 4) == 0 && h / (1024 / (w / 4)) >= 12 && lds_bytes + ytab_bytes <= 156 * 1024;
  a.ytab_off = uni ? (int32_t)((bound + 1) & ~(int64_t)1) : -1;  // (8-byte entries)
  if (uni) lds_bytes = sizeof(float) * (size_t)a.ytab_off + ytab_bytes;
  if (lds_bytes > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(skr::pyramid_pass1<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return SKR_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(skr::pyramid_pass1<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return SKR_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(skr::pyramid_pass1<true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return SKR_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(skr::pyramid_pass1<true, 1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return SKR_ERR_UNSUPPORTED;
  }
  const int64_t w4 = w / 4;
  const dim3 grid1((unsigned)(batch * lead));
  static const int forced = [] { const char* e = getenv("SKR_PYR_MODE"); return e ? atoi(e) : 0; }();  // tuning switch: 1 generic, 2 strip/512, 3 strip/256, 4 strip/1024
  // strips pay off when a thread visits enough rows to amortise its tap table; small planes get there with 256-lane blocks
  const bool strip512 = skr::PYR_THREADS % w4 == 0 && h / (skr::PYR_THREADS / w4) >= 12;
  const bool strip256 = lds_bytes <= 48 * 1024 && 256 % w4 == 0 && h / (256 / w4) >= 12;
  // the strip kernel holds 165 registers, so a CU runs ONE 512-lane block (2 waves per SIMD) whatever the LDS would allow: where the
  // runs stay long enough, 1024 lanes (4 waves per SIMD at 128 registers, 19 of them spilled) hide more of the Philox / Box-Muller
  // dependency chains -- 72.9 against 75.9 us per draw at 64 x (4, 256, 256)
  const bool strip1024 = 1024 % w4 == 0 && h / (1024 / w4) >= 12;
  const int mode = forced ? forced : (strip1024 ? 4 : (strip512 ? 2 : (strip256 ? 3 : 1)));
  if (mode == 4 && strip1024 && uni) hipLaunchKernelGGL((skr::pyramid_pass1<true, 1024, true>), grid1, dim3(1024), lds_bytes, s, a);
  else if (mode == 4 && strip1024) hipLaunchKernelGGL((skr::pyramid_pass1<true, 1024>), grid1, dim3(1024), lds_bytes, s, a);
  else if (mode == 2 && strip512) hipLaunchKernelGGL((skr::pyramid_pass1<true, 512>), grid1, dim3(512), lds_bytes, s, a);
  else if (mode == 3 && strip256) hipLaunchKernelGGL((skr::pyramid_pass1<true, 256>), grid1, dim3(256), lds_bytes, s, a);
  else hipLaunchKernelGGL((skr::pyramid_pass1<false, 512>), grid1, dim3(512), lds_bytes, s, a);
  if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
  const int64_t unit = lead * h * w;
  int64_t bx = (unit / 4 + 255) / 256;
  if (bx > 64) bx = 64;
  dim3 grid((unsigned)bx, (unsigned)batch);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(skr::normalise_pass2<__bf16>, grid, dim3(256), 0, s, (__bf16*)out, scratch_f32, partials_f64, lead, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F16: hipLaunchKernelGGL(skr::normalise_pass2<_Float16>, grid, dim3(256), 0, s, (_Float16*)out, scratch_f32, partials_f64, lead, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F32: hipLaunchKernelGGL(skr::normalise_pass2<float>, grid, dim3(256), 0, s, (float*)out, scratch_f32, partials_f64, lead, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F64: hipLaunchKernelGGL(skr::normalise_pass2<double>, grid, dim3(256), 0, s, (double*)out, scratch_f32, partials_f64, lead, unit, batch, with_base ? -1.0 : 0.0); break;
    default: return SKR_ERR_DTYPE;
  }
  return status_of_launch();
}

static int pyramid_nd_impl(void* out, int32_t out_dtype, float* scratch_f32, float* levels_f32, double* partials_f64, int32_t n_slots,
                           int32_t* level_ws, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch,
                           int32_t nd, const int64_t* shape, int32_t axis_a, int32_t axis_b, double strength, int32_t depth, int32_t with_base, void* stream) {
  skr::DeviceGuard device_guard(out);
  if (batch < 0 || depth < 0 || n_slots < 1 || nd < 1 || nd > 4 || !shape) return SKR_ERR_SHAPE;
  if (axis_b < 0 || axis_b >= nd || axis_a >= axis_b || axis_a < -1) return SKR_ERR_SHAPE;
  for (int i = 0; i < nd; ++i) if (shape[i] < 1 || shape[i] > 0x7fffffffll) return SKR_ERR_SHAPE;
  if (batch == 0) return SKR_OK;
  if (!out || !scratch_f32 || !levels_f32 || !partials_f64 || !seeds_dev || !level_ws) return SKR_ERR_NULL;
  skr::PyramidAnyArgs q;
  skr::PyramidArgs& a = q.p;
  // right-align the axes in the 4-slot descriptor
  const int pad = 4 - nd;
  for (int i = 0; i < 4; ++i) a.dim[i] = i < pad ? 1 : (int32_t)shape[i - pad];
  a.nd = nd; a.axis_a = axis_a < 0 ? -1 : axis_a + pad; a.axis_b = axis_b + pad; a.ytab_off = -1;
  const int64_t h = axis_a < 0 ? 1 : shape[axis_a], w = shape[axis_b];
  int64_t unit = 1;
  for (int i = 0; i < nd; ++i) unit *= shape[i];
  const int64_t lead = unit / (h * w);
  if (h > 32767 || w > 32767 || batch > 65535 || n_slots > 65535 || lead > 0x7fffffffll) return SKR_ERR_UNSUPPORTED;
  a.scratch = scratch_f32; a.partials = partials_f64; a.seeds = seeds_dev;
  a.level_hw = level_ws; a.n_levels = level_ws + batch * skr::PYR_MAX_LEVELS * 2;
  a.stream_base = stream_base; a.stream_levels = stream_levels; a.batch = batch; a.lead = (int32_t)lead; a.h = (int32_t)h; a.w = (int32_t)w;
  a.resize_h = axis_a >= 0 ? 1 : 0; a.depth = depth; a.with_base = with_base; a.strength = (float)strength;
  q.levels = levels_f32; q.cap = unit; q.n_slots = n_slots;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(skr::pyramid_geometry, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, a);
  if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
  int64_t lb = (unit / 16 + 255) / 256;  // the levels >= 1 together hold fewer than `unit` values
  if (lb < 1) lb = 1;
  if (lb > 1024) lb = 1024;
  hipLaunchKernelGGL(skr::pyramid_levels_any, dim3((unsigned)lb, (unsigned)batch), dim3(256), 0, s, q);
  if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
  hipLaunchKernelGGL(skr::pyramid_pass1_any, dim3((unsigned)n_slots, (unsigned)batch), dim3(256), 0, s, q);
  if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
  int64_t bx = (unit / 4 + 255) / 256;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  dim3 grid((unsigned)bx, (unsigned)batch);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(skr::normalise_pass2<__bf16>, grid, dim3(256), 0, s, (__bf16*)out, scratch_f32, partials_f64, (int64_t)n_slots, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F16: hipLaunchKernelGGL(skr::normalise_pass2<_Float16>, grid, dim3(256), 0, s, (_Float16*)out, scratch_f32, partials_f64, (int64_t)n_slots, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F32: hipLaunchKernelGGL(skr::normalise_pass2<float>, grid, dim3(256), 0, s, (float*)out, scratch_f32, partials_f64, (int64_t)n_slots, unit, batch, with_base ? -1.0 : 0.0); break;
    case SKR_F64: hipLaunchKernelGGL(skr::normalise_pass2<double>, grid, dim3(256), 0, s, (double*)out, scratch_f32, partials_f64, (int64_t)n_slots, unit, batch, with_base ? -1.0 : 0.0); break;
    default: return SKR_ERR_DTYPE;
  }
  return status_of_launch();
}

extern "C" int skr_noise_pyramid_any(void* out, int32_t out_dtype, float* scratch_f32, float* levels_f32, double* partials_f64, int32_t n_slots,
                                     int32_t* level_ws, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch,
                                     int64_t lead, int64_t h, int64_t w, int32_t resize_h, double strength, int32_t depth, int32_t with_base, void* stream) {
  if (lead < 1 || h < 1 || w < 1) return SKR_ERR_SHAPE;
  if (!resize_h && h != 1) return SKR_ERR_SHAPE;
  const int64_t shape[3] = {lead, h, w};
  return pyramid_nd_impl(out, out_dtype, scratch_f32, levels_f32, partials_f64, n_slots, level_ws, seeds_dev, stream_base, stream_levels, batch,
                         3, shape, resize_h ? 1 : -1, 2, strength, depth, with_base, stream);
}

extern "C" int skr_noise_pyramid_nd(void* out, int32_t out_dtype, float* scratch_f32, float* levels_f32, double* partials_f64, int32_t n_slots,
                                    int32_t* level_ws, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch,
                                    int32_t ndim, const int64_t* unit_shape, int32_t axis_a, int32_t axis_b, double strength, int32_t depth,
                                    int32_t with_base, void* stream) {
  return pyramid_nd_impl(out, out_dtype, scratch_f32, levels_f32, partials_f64, n_slots, level_ws, seeds_dev, stream_base, stream_levels, batch,
                         ndim, unit_shape, axis_a, axis_b, strength, depth, with_base, stream);
}
