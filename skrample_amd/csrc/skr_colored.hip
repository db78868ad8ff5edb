// Colored (power-law) noise for MI355X -- reference skrample/pytorch/noise.py:284-425.
//
//   white = N()  ->  F = rfftn(white)  ->  F *= clamp(radius, eps)^(-exponent/2)  ->  irfftn  ->  rescale so
//   that std(out) = std(white) (or `energy`), per sample.
//
// The per-sample transform is 2-D or 3-D over power-of-two sizes and is done axis by axis with a shared
// LDS radix-2 FFT (`fft_tile`): a 256-thread block owns a tile of L lines x N points (<= 4096 complex), loads
// it coalesced whatever the axis stride, transforms all lines together and stores it back.
//   pass A  last axis, real -> half spectrum; the real input is *drawn in place* (Philox), never read
//   pass B  middle axis forward                                   (3-D only)
//   pass C  outermost axis forward, x radial weights, inverse     (fused: the weights need the full transform)
//   pass D  middle axis inverse                                   (3-D only)
//   pass E  last axis, half spectrum -> real (fp32 scratch) + per-block sums
//   pass F  per-sample unbiased std of white (from A) and coloured (from E), rescale, round to out dtype
// Planes that fit the LDS take the fused kernels instead (colored_plane): A+B in one kernel, D+E+F in another -- the
// coloured std comes from the weighted spectrum by Parseval (reduced in pass C), so the inverse kernel writes the final
// dtype itself and the fp32 scratch and pass F disappear; 2-D units are a single kernel.
// Reductions are per block into fixed slots and summed in a fixed order (bit-reproducible, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "skr_device.h"
#include "../../include/skrample_hip.h"
#include "skr_philox.h"
#include "skr_pack.h"
#include "skr_dft.h"
#include "skr_fft_tile.h"

namespace skr {


struct ColoredArgs {
  float2* spec;         // [batch][d1][d2][d3h]
  float* real_out;      // [batch][d1][d2][d3] fp32 scratch (pass E)
  double* partials;     // [2][batch][n_slots][2]  (0: white from pass A, 1: coloured from pass E)
  const uint64_t* seeds;
  uint64_t stream;
  int64_t batch;
  int32_t d1, d2, d3, d3h;   // d1 = 1 for 2-D
  int32_t n_slots;           // partial slots in use per sample (white half; coloured half of the unfused path)
  int32_t n_slots_c;         // coloured-half slots written by the Parseval reduction of the fused 3-D path
  int32_t has_energy;
  double energy;
  void* out;                 // fused paths write the result dtype directly
  float exponent_half_neg;   // -exponent / 2
  float eps_clip;
  float inv_rmax;
  int32_t raw;               // MODE 1 of the plane kernels: plain inverse transform (no 1/N, no rescale factor) -- colored_planes
  const float* factors;      // colored_inverse128: [batch] per-sample rescale factors (colored_factors), or null
  uint32_t* ticket;          // colored_inverse128: the next unclaimed plane, or null = planes dealt out statically
  uint32_t first_ticket;     // its starting value (2 x the inverse kernel's grid: every block's first two planes are fixed)
  uint32_t* done_count;      // [batch] blocks of the outer-axis kernel that have stored their Parseval slot; the last one of a sample computes factors[smp]
  float* factors_out;        // where (null: the factors come from the colored_factors launch)
#ifdef SKR_COLORED_TRACE
  uint64_t* trace;           // tools/tune/tune_colored.hip only: [block][16] phase stamps of the plane kernels (s_memrealtime, 10 ns)
#endif
};

#ifdef SKR_INV_STEP_MODEL  // experiment (tools/tune/tune_inverse128_step.hip); compiled out of the library
struct StepModel { const void* narrow[7]; const void* wide[2]; float* state; };
__device__ StepModel g_step_model;
#endif

// Phase stamps for the timeline harness (tools/tune/tune_colored.hip); compiled out of the library.
#ifdef SKR_COLORED_TRACE
uint64_t* g_colored_trace = nullptr;
#define SKR_STAMP(i)                                                                                          \
  do {                                                                                                        \
    if (threadIdx.x == 0 && a.trace) {                                                                        \
      uint64_t* t_ = a.trace + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;                           \
      t_[(MODE == 1 ? 6 : 0) + (i)] = __builtin_amdgcn_s_memrealtime();                                       \
      if ((i) == 0) { t_[MODE == 1 ? 14 : 12] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t_[MODE == 1 ? 15 : 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } \
    }                                                                                                         \
  } while (0)
#else
#define SKR_STAMP(i) do {} while (0)
#endif


// 128-point lines, round 4: stages 3-6 in ONE pass.  With n = 16 n1 + n2 the 8-point pass leaves A_n2[k1] = sum_n1 x[16 n1 + n2] W_8^(n1 k1)
// at position 8 bitrev4(n2) + k1 of its line (the input sits in bit-reversed order), and
//     X[k1 + 8 k2] = sum_n2 ( W_128^(n2 k1) A_n2[k1] ) W_16^(n2 k2):
// an item -- (line, k1) -- takes its sixteen values, multiplies them by their twiddles (table of the FULL circle: n2 k1 <= 105) and
// transforms them in registers; the results go back over the sixteen inputs in natural order (or to global memory: TO_GLOBAL as in
// fft_tile).  Against the two radix-2^2 passes this replaces: one LDS round trip and one barrier instead of two, one twiddle read
// per point instead of three per four, and the index arithmetic of one item per sixteen points instead of two per four.
// Consecutive lanes take consecutive lines (pitch 129 complex: conflict-free 8-byte accesses).
template <bool INVERSE, bool SKIP8, bool TO_GLOBAL>
__device__ __forceinline__ void fft_tile_128(float2* buf, const float2* tw_full, int L, float2* gout = nullptr, int gpitch = 0) {
  constexpr int N = 128, ld = N + 1;
  const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)L - 1) / (uint32_t)L);
  if constexpr (!SKIP8) {
    __syncthreads();
    const int total = L * (N >> 3);
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      const int g = (int)__umulhi((uint32_t)t, magic), line = t - g * L;
      float2* p = buf + line * ld + 8 * g;
      float2 v[8];
      v[0] = p[0]; v[4] = p[1]; v[2] = p[2]; v[6] = p[3]; v[1] = p[4]; v[5] = p[5]; v[3] = p[6]; v[7] = p[7];
      dft8<INVERSE>(v);
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = v[k];
    }
  }
  __syncthreads();
  const int total = L * 8;
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    const int k1 = (int)__umulhi((uint32_t)t, magic), line = t - k1 * L;
    float2* p = buf + line * ld + k1;
    float2 v[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = p[8 * (int)(__builtin_bitreverse32((unsigned)n2) >> 28)];
    const float2* w = tw_full;
#pragma unroll
    for (int n2 = 1; n2 < 16; ++n2) {
      w += k1;  // W_128^(n2 k1)
      float2 c = *w;
      if (INVERSE) c.y = -c.y;
      v[n2] = cmul(v[n2], c);
    }
    dft16<INVERSE>(v);
    if constexpr (TO_GLOBAL) {
      float2* g = gout + (int64_t)k1 * gpitch + line;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) g[(int64_t)(8 * k2) * gpitch] = v[k2];
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) p[8 * k2] = v[k2];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void make_twiddles_full(float2* tw, int N) {  // exp(-2 pi i k / N) for the whole circle, k < N
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)N, &s, &c);
    tw[k] = make_float2(c, s);
  }
}

__device__ __forceinline__ float axis_freq(int k, int d) { const int m = k < d - k ? k : d - k; return (float)m / (float)d; }

// clamp(radius, eps)^(-exponent/2) on the raw transcendental units: radius >= eps > 0 is never denormal, so
// exp2(e * log2(r)) needs neither libm's powf special cases (~40 instructions) nor sqrt's correctly-rounded fix-up;
// relative error ~1e-6 for the exponents in use (|e| <= 2), inside the generator's 1e-5 parity bar.
__device__ __forceinline__ float radial_weight(float sum_sq, float inv_rmax, float eps_clip, float exponent_half_neg) {
  float radius = __builtin_amdgcn_sqrtf(sum_sq) * inv_rmax;
  radius = radius < eps_clip ? eps_clip : radius;
  return __builtin_amdgcn_exp2f(exponent_half_neg * __builtin_amdgcn_logf(radius));
}

// block-wide sum of two doubles into partial slot (fixed order)
template <bool COH = false>
__device__ __forceinline__ void block_sums(double s1, double s2, double* slot) {
  __shared__ double red[2][16];  // up to 1024 threads
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[0][w]; b += red[1][w]; }
    gstore<COH>(slot, a); gstore<COH>(slot + 1, b);
  }
}

// per-sample rescale factor from the partial sums (reference noise.py:395-403): white std / coloured std (or energy / coloured std)
__device__ __forceinline__ float rescale_factor(double w1, double w2, double c1, double c2, double n, int has_energy, double energy) {
  const double wstd = sqrt((w2 - w1 * w1 / n) / (n - 1.0));
  const double cstd = sqrt((c2 - c1 * c1 / n) / (n - 1.0));
  float factor = 1.0f;  // only rescale when the coloured std is not degenerate
  if ((float)cstd > 1e-8f) factor = has_energy ? (float)energy / (float)cstd : (float)wstd / (float)cstd;
  return factor;
}

// block-wide sums of four doubles, result broadcast to every thread (fixed order)
__device__ __forceinline__ void block_sums4_bcast(double v[4]) {
  __shared__ double red4[4][16];
  __shared__ double tot4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { for (int i = 0; i < 4; ++i) red4[i][wave] = v[i]; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double acc = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) acc += red4[threadIdx.x][w];
    tot4[threadIdx.x] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = tot4[i];
}

// ---- pass A / E: last axis (contiguous lines of d3 reals <-> d3h complex) -------------------------------------
// Two real lines share one complex transform ("two for one"): z = x_a + i x_b, Z = FFT(z), and
//   X_a[k] = (Z[k] + conj Z[N-k]) / 2,   X_b[k] = (Z[k] - conj Z[N-k]) / (2i);
// backwards, Z[k] = X_a[k] + i X_b[k] for k <= N/2 and conj X_a[N-k] + i conj X_b[N-k] above, x_a = Re z, x_b = Im z.
// L is the number of REAL lines per tile (even); the LDS tile holds L/2 complex lines.
template <bool FORWARD>
__global__ __launch_bounds__(FFT_THREADS) void colored_last_axis(const ColoredArgs a, int logN, int L) {
  extern __shared__ float2 smem[];
  const int N = a.d3, ld = N + 1;
  float2* tw = smem;
  float2* buf = smem + N / 2;
  const int64_t smp = blockIdx.y;
  const int64_t n_lines = (int64_t)a.d1 * a.d2;
  const int64_t line0 = (int64_t)blockIdx.x * L;
  const int lines = (int)((n_lines - line0) < L ? (n_lines - line0) : L);  // even: n_lines and L are even
  const int pairs = lines >> 1;
  make_twiddles(tw, N);
  double s1 = 0.0, s2 = 0.0;
  if (FORWARD) {
    // draw the white noise straight into LDS (bit-reversed), 4 normals per Philox call and line
    const uint64_t seed = a.seeds[smp];
    for (int q = threadIdx.x; q < pairs * (N / 4); q += FFT_THREADS) {
      const int pr = q >> (logN - 2), n4 = (q & (N / 4 - 1)) * 4;
      const int64_t ea = (line0 + 2 * pr) * N + n4;  // element index inside the sample (line 2pr), line 2pr+1 is N further
      float za[4], zb[4];
      normal4(seed, a.stream, (uint64_t)ea >> 2, za);
      normal4(seed, a.stream, (uint64_t)(ea + N) >> 2, zb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        buf[pr * ld + brev(n4 + j, logN)] = make_float2(za[j], zb[j]);
        s1 += (double)za[j] + (double)zb[j];
        s2 += (double)za[j] * (double)za[j] + (double)zb[j] * (double)zb[j];
      }
    }
    fft_tile<false>(buf, tw, N, logN, pairs);
    // untangle and store the two half spectra of every pair; [lines][d3h] is contiguous in memory
    float2* dst = a.spec + (smp * n_lines + line0) * a.d3h;
    {
      const int total = pairs * a.d3h;
      int pr = threadIdx.x / a.d3h, k = threadIdx.x - pr * a.d3h;
      const int dl = FFT_THREADS / a.d3h, dk = FFT_THREADS - dl * a.d3h;
      for (int q = threadIdx.x; q < total; q += FFT_THREADS) {
        const float2 zk = buf[pr * ld + k], zn = buf[pr * ld + ((N - k) & (N - 1))];
        dst[(int64_t)(2 * pr) * a.d3h + k] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        dst[(int64_t)(2 * pr + 1) * a.d3h + k] = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        pr += dl; k += dk;
        if (k >= a.d3h) { k -= a.d3h; ++pr; }
      }
    }
    block_sums(s1, s2, a.partials + ((0 * a.batch + smp) * a.n_slots + blockIdx.x) * 2);
  } else {
    const float2* src = a.spec + (smp * n_lines + line0) * a.d3h;
    for (int q = threadIdx.x; q < pairs * N; q += FFT_THREADS) {
      const int pr = q >> logN, k = q & (N - 1);
      const int m = k < a.d3h ? k : N - k;
      float2 xa = src[(int64_t)(2 * pr) * a.d3h + m], xb = src[(int64_t)(2 * pr + 1) * a.d3h + m];
      if (m == 0 || 2 * m == N) { xa.y = 0.f; xb.y = 0.f; }  // irfft ignores the imaginary part of DC / Nyquist
      if (k >= a.d3h) { xa.y = -xa.y; xb.y = -xb.y; }         // Hermitian half
      buf[pr * ld + brev(k, logN)] = make_float2(xa.x - xb.y, xa.y + xb.x);
    }
    fft_tile<true>(buf, tw, N, logN, pairs);
    const float scale = 1.0f / ((float)a.d1 * (float)a.d2 * (float)a.d3);
    float* dst = a.real_out + (smp * n_lines + line0) * N;
    for (int q = threadIdx.x; q < pairs * N; q += FFT_THREADS) {
      const int pr = q >> logN, n = q & (N - 1);
      const float2 z = buf[pr * ld + n];
      const float va = z.x * scale, vb = z.y * scale;
      dst[(int64_t)(2 * pr) * N + n] = va;
      dst[(int64_t)(2 * pr + 1) * N + n] = vb;
      s1 += (double)va + (double)vb;
      s2 += (double)va * (double)va + (double)vb * (double)vb;
    }
    block_sums(s1, s2, a.partials + ((1 * a.batch + smp) * a.n_slots + blockIdx.x) * 2);
  }
}

// Pass E writing the RESULT dtype itself (round 3; planes too large for the plane kernels, 3-D units with a short outer axis):
// the coloured std is known by Parseval from the weighted spectrum -- the outer-axis kernel reduced it into the partials, as it does
// for the plane path -- so the rescale factor is available before the first real value exists, and the fp32 scratch (a write and a
// read of the whole sample) and the finishing pass disappear.
template <typename T>
__global__ __launch_bounds__(FFT_THREADS) void colored_last_axis_out(const ColoredArgs a, int logN, int L) {
  extern __shared__ float2 smem[];
  const int N = a.d3, ld = N + 1;
  float2* tw = smem;
  float2* buf = smem + N / 2;
  const int64_t smp = blockIdx.y;
  const int64_t n_lines = (int64_t)a.d1 * a.d2;
  const int64_t line0 = (int64_t)blockIdx.x * L;
  const int lines = (int)((n_lines - line0) < L ? (n_lines - line0) : L);
  const int pairs = lines >> 1;
  double fa[4] = {0.0, 0.0, 0.0, 0.0};
  if (threadIdx.x < 64) {  // the sample's partial sums (white from pass A, Parseval from the outer-axis pass): loaded now, reduced below
    const double* pw = a.partials + (0 * a.batch + smp) * a.n_slots * 2;
    for (int sl = threadIdx.x; sl < a.n_slots; sl += 64) { fa[0] += pw[2 * sl]; fa[1] += pw[2 * sl + 1]; }
    const double* pc = a.partials + (int64_t)a.batch * a.n_slots * 2 + smp * a.n_slots_c * 2;
    for (int sl = threadIdx.x; sl < a.n_slots_c; sl += 64) { fa[2] += pc[2 * sl]; fa[3] += pc[2 * sl + 1]; }
  }
  make_twiddles(tw, N);
  const float2* src = a.spec + (smp * n_lines + line0) * a.d3h;
  // four items = eight loads in flight per lane (a rolled loop pays one HBM latency per item)
  for (int q0 = threadIdx.x; q0 < pairs * N; q0 += 4 * FFT_THREADS) {
    float2 xa[4], xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * FFT_THREADS;
      if (q < pairs * N) {
        const int pr = q >> logN, k = q & (N - 1);
        const int m = k < a.d3h ? k : N - k;
        xa[u] = src[(int64_t)(2 * pr) * a.d3h + m];
        xb[u] = src[(int64_t)(2 * pr + 1) * a.d3h + m];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * FFT_THREADS;
      if (q < pairs * N) {
        const int pr = q >> logN, k = q & (N - 1);
        const int m = k < a.d3h ? k : N - k;
        float2 va = xa[u], vb = xb[u];
        if (m == 0 || 2 * m == N) { va.y = 0.f; vb.y = 0.f; }
        if (k >= a.d3h) { va.y = -va.y; vb.y = -vb.y; }
        buf[pr * ld + brev(k, logN)] = make_float2(va.x - vb.y, va.y + vb.x);
      }
    }
  }
  __shared__ float factor_sh;
  if (threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (int o = 32; o > 0; o >>= 1) fa[i] += __shfl_down(fa[i], o);
    if (threadIdx.x == 0) {
      const double n = (double)a.d1 * (double)a.d2 * (double)a.d3;
      factor_sh = rescale_factor(fa[0], fa[1], fa[2], fa[3] / n, n, a.has_energy, a.energy);
    }
  }
  fft_tile<true>(buf, tw, N, logN, pairs);
  const float gain = factor_sh / ((float)a.d1 * (float)a.d2 * (float)a.d3);
  T* dst = reinterpret_cast<T*>(a.out) + (smp * n_lines + line0) * N;
  for (int q = threadIdx.x; q < pairs * (N >> 2); q += FFT_THREADS) {
    const int pr = q >> (logN - 2), n4 = (q & ((N >> 2) - 1)) * 4;
    const float2 z0 = buf[pr * ld + n4], z1 = buf[pr * ld + n4 + 1], z2 = buf[pr * ld + n4 + 2], z3 = buf[pr * ld + n4 + 3];
    if constexpr (sizeof(T) <= 4) {
      store4_from_f32<T>(dst + (int64_t)(2 * pr) * N + n4, z0.x * gain, z1.x * gain, z2.x * gain, z3.x * gain);
      store4_from_f32<T>(dst + (int64_t)(2 * pr + 1) * N + n4, z0.y * gain, z1.y * gain, z2.y * gain, z3.y * gain);
    } else {
      T* da = dst + (int64_t)(2 * pr) * N + n4;
      T* db = dst + (int64_t)(2 * pr + 1) * N + n4;
      da[0] = (T)(z0.x * gain); da[1] = (T)(z1.x * gain); da[2] = (T)(z2.x * gain); da[3] = (T)(z3.x * gain);
      db[0] = (T)(z0.y * gain); db[1] = (T)(z1.y * gain); db[2] = (T)(z2.y * gain); db[3] = (T)(z3.y * gain);
    }
  }
}

// ---- fused passes over the two inner axes: one (sample, i1) plane of d2 x d3 per block, entirely in LDS ---------
// MODE 0: white noise -> rows (two-for-one) -> columns -> half spectrum to HBM          (replaces passes A + B)
// MODE 1: half spectrum -> columns^-1 -> rows^-1 -> real plane + statistics            (replaces passes D + E)
// MODE 2: both, with the radial weights applied in LDS in between (2-D units: the spectrum never leaves the CU)
// The row tile ((d2/2) x (d3+1) complex, row pairs) and the column tile (d3h x (d2+1) complex) share ONE LDS
// region: the transposing steps between them stage every item in registers across a barrier.  A 128 x 128 plane
// then needs 67 KiB, so two blocks fit a CU and one block's HBM phase overlaps the other's butterflies.
constexpr int PLANE_THREADS = 512;
constexpr int PLANE_ITEMS = 18;  // staged items per thread: the host keeps (d2/2)*max(d3h, d3) <= PLANE_THREADS * PLANE_ITEMS

// MODE 1 and 2 write the result dtype T themselves.  The per-sample rescale factor needs the coloured std of the WHOLE
// sample before any of it is written; by Parseval it is known from the weighted spectrum -- sum x^2 = (1/N) sum |V|^2 over the
// full grid (half-spectrum bins off the k3 = 0 / Nyquist planes count twice), sum x = V[0] -- which MODE 2 reduces
// in place and the outer-axis kernel of the 3-D path reduces per block (partials).  No fp32 scratch, no finishing pass.
// CH / CW: log2 of the plane's height / width as compile-time constants (0 = runtime).  With them every LDS row pitch
// (W + 1, H + 1) multiply becomes a shift-add and the loop trip counts are known: the runtime-size kernel spends a quarter
// of its VALU issue on quarter-rate v_mul_lo_u32 index arithmetic.
template <int MODE, typename T, int CH, int CW, bool COH = false>
__device__ __forceinline__ void plane_body(const ColoredArgs& a, int logH_rt, int logW_rt, const int64_t smp, const int i1, float2* smem) {
  const int logH = CH ? CH : logH_rt, logW = CW ? CW : logW_rt;
  const int H = CH ? (1 << CH) : a.d2, W = CW ? (1 << CW) : a.d3, WH = CW ? (1 << CW) / 2 + 1 : a.d3h, ldw = W + 1, ldh = H + 1, pairs = H >> 1;
  // (the twiddle region holds W + H entries: the 128-point lines of the compile-time 128 x 128 instantiation index the full circle)
  constexpr bool R16_W = CW == 7, R16_H = CH == 7 && CW >= 3;  // 128-point lines: 8-point pass + ONE radix-16 pass (fft_tile_128)
  float2* tw_w = smem;
  float2* tw_h = tw_w + W;
  float2* t1 = tw_h + H;  // row-pair tile
  float2* t2 = t1;        // column tile (aliases t1)
  const uint32_t magic_wh = (uint32_t)((0x100000000ull + (uint32_t)WH - 1) / (uint32_t)WH);  // q / WH == umulhi(q, magic) for q < 2^16
  SKR_STAMP(0);
  if constexpr (R16_W) make_twiddles_full(tw_w, W); else make_twiddles(tw_w, W);
  if constexpr (R16_H) make_twiddles_full(tw_h, H); else make_twiddles(tw_h, H);
  float2* plane = a.spec + ((smp * a.d1 + i1) * (int64_t)H) * WH;
  double s1 = 0.0, s2 = 0.0;
  double fa[4] = {0.0, 0.0, 0.0, 0.0};  // MODE 1, first wave: this lane's share of the sample's partial sums (white s1 s2, coloured s1 s2)
  if (MODE == 1 && !a.raw && threadIdx.x < 64) {
    const double* pw = a.partials + (0 * a.batch + smp) * a.n_slots * 2;
    for (int sl = threadIdx.x; sl < a.n_slots; sl += 64) { fa[0] += gload<COH>(pw + 2 * sl); fa[1] += gload<COH>(pw + 2 * sl + 1); }
    const double* pc = a.partials + (int64_t)a.batch * a.n_slots * 2 + smp * a.n_slots_c * 2;
    for (int sl = threadIdx.x; sl < a.n_slots_c; sl += 64) { fa[2] += gload<COH>(pc + 2 * sl); fa[3] += gload<COH>(pc + 2 * sl + 1); }
  }

  if (MODE != 1) {
    const uint64_t seed = a.seeds[smp];
    float ws1 = 0.f, ws2 = 0.f;  // this thread's <= 144 values in fp32, one widening at the end: the per-block totals stay double
    // 128-point rows: the draw FUSED with the 8-point pass of the row transform.  With n = 16 n1 + n2 a Philox block holds four
    // consecutive n2 of one n1, so an item -- (row pair, q = n2 / 4) -- draws the sixteen blocks of its two rows that cover
    // n2 = 4q .. 4q+3 for all eight n1, owns four complete 8-point subsequences in natural order, transforms them in registers and
    // writes A_n2[k1] where fft_tile_128's radix-16 pass expects it (position 8 bitrev4(n2) + k1).  The white noise never sits in
    // LDS untransformed: one LDS round trip (16 reads + 16 writes per thread and their index arithmetic) and one barrier fewer.
    // pairs x 4 items fill half the block; the other four waves go straight to the barrier.
    constexpr bool FUSE_DRAW = R16_W && CH >= 1;
    if constexpr (FUSE_DRAW) {
      if (threadIdx.x < (unsigned)(pairs * 4)) {
        const int pr = threadIdx.x & (pairs - 1), q = threadIdx.x >> (CH - 1);  // consecutive lanes: consecutive pairs (lines one odd pitch apart)
        float2 z[4][8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
          const int64_t ea = ((int64_t)i1 * H + 2 * pr) * W + 16 * n1 + 4 * q;
          float za[4], zb[4];
          normal4(seed, a.stream, (uint64_t)ea >> 2, za);
          normal4(seed, a.stream, (uint64_t)(ea + W) >> 2, zb);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            z[j][n1] = make_float2(za[j], zb[j]);
            ws1 += za[j] + zb[j];
            ws2 = __builtin_fmaf(za[j], za[j], __builtin_fmaf(zb[j], zb[j], ws2));
          }
        }
        const int rq = ((q & 1) << 1) | (q >> 1);  // bitrev2(q): bitrev4(4 q + j) = 4 bitrev2(j) + bitrev2(q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dft8<false>(z[j]);
          float2* p = t1 + pr * ldw + 8 * (4 * (((j & 1) << 1) | (j >> 1)) + rq);
#pragma unroll
          for (int k1 = 0; k1 < 8; ++k1) p[k1] = z[j][k1];
        }
      }
    } else
    for (int q = threadIdx.x; q < pairs * (W / 4); q += PLANE_THREADS) {
      const int pr = q >> (logW - 2), n4 = (q & (W / 4 - 1)) * 4;
      const int64_t ea = ((int64_t)i1 * H + 2 * pr) * W + n4;
      float za[4], zb[4];
      normal4(seed, a.stream, (uint64_t)ea >> 2, za);
      normal4(seed, a.stream, (uint64_t)(ea + W) >> 2, zb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t1[pr * ldw + brev(n4 + j, logW)] = make_float2(za[j], zb[j]);
        ws1 += za[j] + zb[j];
        ws2 = __builtin_fmaf(za[j], za[j], __builtin_fmaf(zb[j], zb[j], ws2));
      }
    }
    s1 = (double)ws1; s2 = (double)ws2;
    SKR_STAMP(1);
    if constexpr (R16_W) fft_tile_128<false, FUSE_DRAW, false>(t1, tw_w, pairs);
    else fft_tile<false>(t1, tw_w, W, logW, pairs);
    SKR_STAMP(2);
    constexpr bool FUSE_COL = CH >= 5 && (CH & 1) && CW >= 3;  // 128-row planes: the column transform opens with an 8-point pass
    if constexpr (FUSE_COL) {
      // Untangling FUSED with stages 0-2 of the column transform.  Positions 8g..8g+7 of a column hold rows r0 + (H/8) n in
      // 3-bit-reversed order, r0 = bitrev(g); rows 2 p0 and 2 p0 + 1 come out of the same row pair, so one item -- (kw, p0) --
      // untangles the eight pairs p0 + (H/16) n, owns both 8-row subsequences, transforms them in registers and, behind the
      // barrier (the column tile overwrites the row tile), puts each into its group of 8 consecutive column positions.  One
      // LDS round trip of the whole tile and one barrier less than untangle-then-transform; W/2 x H/16 items with kw >= 1 fill
      // the block exactly at 128 x 128, the DC column's H/16 items ride on the first lanes.
      constexpr int P0 = H / 16, MAIN = (W / 2) * P0, ROUNDS = (MAIN + PLANE_THREADS - 1) / PLANE_THREADS;
      float2 va[ROUNDS + 1][8], vb[ROUNDS + 1][8];
      auto untangle8 = [&](int k, int p0, float2* xa, float2* xb) {
        const int kn = (W - k) & (W - 1);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const float2 zk = t1[(p0 + P0 * n) * ldw + k], zn = t1[(p0 + P0 * n) * ldw + kn];
          xa[n] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
          xb[n] = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        }
        dft8<false>(xa);
        dft8<false>(xb);
      };
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int q = threadIdx.x + r * PLANE_THREADS;
        if (q < MAIN) untangle8(1 + (q & (W / 2 - 1)), q >> (CW - 1), va[r], vb[r]);
      }
      if (threadIdx.x < P0) untangle8(0, threadIdx.x, va[ROUNDS], vb[ROUNDS]);
      __syncthreads();
      auto put8 = [&](int k, int p0, const float2* xa, const float2* xb) {
        float2* ca = t2 + k * ldh + 8 * (int)brev(2 * p0, CH - 3);
        float2* cb = t2 + k * ldh + 8 * (int)brev(2 * p0 + 1, CH - 3);
#pragma unroll
        for (int m = 0; m < 8; ++m) { ca[m] = xa[m]; cb[m] = xb[m]; }
      };
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int q = threadIdx.x + r * PLANE_THREADS;
        if (q < MAIN) put8(1 + (q & (W / 2 - 1)), q >> (CW - 1), va[r], vb[r]);
      }
      if (threadIdx.x < P0) put8(0, threadIdx.x, va[ROUNDS], vb[ROUNDS]);
      if constexpr (R16_H) fft_tile_128<false, true, MODE == 0>(t2, tw_h, WH, plane, WH);
      else fft_tile<false, true, MODE == 0, COH>(t2, tw_h, H, logH, WH, plane, WH);
    } else {
    {
      // untangle the row pairs into the column tile (bit-reversed along H for the column transform)
      float2 ra[PLANE_ITEMS], rb[PLANE_ITEMS];
      const int total = pairs * WH;
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * PLANE_THREADS;
        if (q < total) {
          const int pr = (int)__umulhi((uint32_t)q, magic_wh), k = q - pr * WH;
          const float2 zk = t1[pr * ldw + k], zn = t1[pr * ldw + ((W - k) & (W - 1))];
          ra[i] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
          rb[i] = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * PLANE_THREADS;
        if (q < total) {
          const int pr = (int)__umulhi((uint32_t)q, magic_wh), k = q - pr * WH;
          t2[k * ldh + brev(2 * pr, logH)] = ra[i];
          t2[k * ldh + brev(2 * pr + 1, logH)] = rb[i];
        }
      }
    }
    fft_tile<false, false, MODE == 0 && (CH >= 2), COH>(t2, tw_h, H, logH, WH, plane, WH);
    }
    SKR_STAMP(3);
    if (MODE == 0) block_sums<COH>(s1, s2, a.partials + ((0 * a.batch + smp) * a.n_slots + i1) * 2);
    if (MODE == 0 && a.done_count != nullptr && i1 == 0 && threadIdx.x == 0) {  // bookkeeping of the two kernels behind this one (visible at the kernel boundary)
      a.done_count[smp] = 0u;
      if (smp == 0 && a.ticket != nullptr) *a.ticket = a.first_ticket;
    }
    if (MODE == 0) {
      if constexpr (CH < 2) {  // (runtime-size instantiation: the transform's last pass may be the 2-point one, which writes the tile)
        for (int q = threadIdx.x; q < H * WH; q += PLANE_THREADS) {
          const int row = (int)__umulhi((uint32_t)q, magic_wh), k = q - row * WH;
          gstore<COH>(plane + q, t2[k * ldh + row]);
        }
      }
      SKR_STAMP(4);
      return;  // (compile-time sizes: the column transform's last pass stored the half spectrum itself)
    }
    // MODE 2: weights in place (+ the Parseval sums of the weighted spectrum), then bit-reverse the columns for the inverse
    double p1 = 0.0, p2 = 0.0;
    for (int q = threadIdx.x; q < H * WH; q += PLANE_THREADS) {
      const int k = q >> logH, row = q & (H - 1);
      const float f1 = 0.f, f2 = axis_freq(row, H), f3 = (float)k / (float)W;
      const float wgt = radial_weight(f1 * f1 + f2 * f2 + f3 * f3, a.inv_rmax, a.eps_clip, a.exponent_half_neg);
      float2 v = t2[k * ldh + row];
      v = make_float2(v.x * wgt, v.y * wgt);
      t2[k * ldh + row] = v;
      const float e = __builtin_fmaf(v.x, v.x, v.y * v.y);
      p2 += (double)((k == 0 || 2 * k == W) ? e : 2.f * e);
      if (q == 0) p1 = (double)v.x;
    }
    {
      double tot[4] = {s1, s2, p1, p2};
      block_sums4_bcast(tot);  // (also the barrier between the weighting and the bit reversal)
      const double n = (double)H * (double)W;
      s1 = (double)rescale_factor(tot[0], tot[1], tot[2], tot[3] / n, n, a.has_energy, a.energy);  // s1 now carries the factor
    }
    for (int q = threadIdx.x; q < H * WH; q += PLANE_THREADS) {
      const int k = q >> logH, row = q & (H - 1);
      const int r = (int)brev(row, logH);
      if (row < r) { float2 t = t2[k * ldh + row]; t2[k * ldh + row] = t2[k * ldh + r]; t2[k * ldh + r] = t; }
    }
  } else {
    if constexpr (CH >= 5 && (CH & 1) && CW >= 3) {
      // The spectrum load FUSED with stages 0-2 of the column transform: an item -- (kw, g) -- loads the eight rows
      // r0 + (H/8) n, r0 = bitrev(g), of one frequency column straight into registers (consecutive lanes = consecutive kw:
      // 8-byte loads in runs of a spectrum row), transforms them and writes positions 8g..8g+7 of the column once.  All loads
      // of the block go out before the first butterfly (W/2 x H/8 items fill two rounds exactly at 128 x 128; the Nyquist
      // column's H/8 items ride on the first lanes).
      constexpr int G = H / 8, MAIN = (W / 2) * G, ROUNDS = (MAIN + PLANE_THREADS - 1) / PLANE_THREADS;
      float2 lv[ROUNDS + 1][8];
      auto load8 = [&](int k, int g, float2* v) {
        const float2* src = plane + (int64_t)brev(g, CH - 3) * WH + k;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = gload<COH>(src + (int64_t)n * G * WH);
      };
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int q = threadIdx.x + r * PLANE_THREADS;
        if (q < MAIN) load8(q & (W / 2 - 1), q >> (CW - 1), lv[r]);
      }
      if (threadIdx.x < G) load8(W / 2, threadIdx.x, lv[ROUNDS]);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int q = threadIdx.x + r * PLANE_THREADS;
        if (q < MAIN) {
          dft8<true>(lv[r]);
          float2* col = t2 + (q & (W / 2 - 1)) * ldh + 8 * (q >> (CW - 1));
#pragma unroll
          for (int m = 0; m < 8; ++m) col[m] = lv[r][m];
        }
      }
      if (threadIdx.x < G) {
        dft8<true>(lv[ROUNDS]);
        float2* col = t2 + (W / 2) * ldh + 8 * threadIdx.x;
#pragma unroll
        for (int m = 0; m < 8; ++m) col[m] = lv[ROUNDS][m];
      }
    } else {
    // all of the plane's loads are issued before the first LDS write (a rolled loop would pay one HBM latency per trip)
    const int total = H * WH;  // <= 2 * PLANE_THREADS * PLANE_ITEMS (host check)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float2 rz[PLANE_ITEMS];
      const int base = half * PLANE_THREADS * PLANE_ITEMS;
      if (base >= total) break;
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = base + threadIdx.x + i * PLANE_THREADS;
        if (q < total) rz[i] = gload<COH>(plane + q);
      }
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = base + threadIdx.x + i * PLANE_THREADS;
        if (q < total) {
          const int row = (int)__umulhi((uint32_t)q, magic_wh), k = q - row * WH;
          t2[k * ldh + brev(row, logH)] = rz[i];
        }
      }
    }
    }
    SKR_STAMP(1);
  }

  if constexpr (R16_H) fft_tile_128<true, MODE == 1, false>(t2, tw_h, WH);
  else fft_tile<true, MODE == 1 && CH >= 5 && (CH & 1) && CW >= 3>(t2, tw_h, H, logH, WH);
  SKR_STAMP(2);
  constexpr bool FUSE_ROW = CW >= 5 && (CW & 1) && CH >= 2;  // 128-point rows: the row transform opens with an 8-point pass
  if constexpr (FUSE_ROW) {
    // Hermitian packing FUSED with stages 0-2 of the row transform (the mirror image of the fused untangling above): positions
    // 8g..8g+7 of a row-pair line hold frequencies k0 + (W/8) n in 3-bit-reversed order, k0 = bitrev(g); an item -- (pair, g) --
    // builds those eight packed values from the column tile, transforms them in registers and writes the group behind the
    // barrier.  Consecutive lanes take consecutive pairs: 16-byte reads of (row 2p, row 2p+1) side by side, writes one line apart.
    constexpr int G = W / 8, ROUNDS = ((H / 2) * G + PLANE_THREADS - 1) / PLANE_THREADS;
    float2 rv[ROUNDS][8];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int q = threadIdx.x + r * PLANE_THREADS;
      if (q < pairs * G) {
        const int pr = q & (pairs - 1), k0 = (int)brev(q >> (CH - 1), CW - 3);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const int k = k0 + G * n;
          const int m = k < WH ? k : W - k;
          float2 xa = t2[m * ldh + 2 * pr], xb = t2[m * ldh + 2 * pr + 1];
          if (m == 0 || 2 * m == W) { xa.y = 0.f; xb.y = 0.f; }
          if (k >= WH) { xa.y = -xa.y; xb.y = -xb.y; }
          rv[r][n] = make_float2(xa.x - xb.y, xa.y + xb.x);
        }
        dft8<true>(rv[r]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int q = threadIdx.x + r * PLANE_THREADS;
      if (q < pairs * G) {
        float2* dst8 = t1 + (q & (pairs - 1)) * ldw + 8 * (q >> (CH - 1));
#pragma unroll
        for (int m = 0; m < 8; ++m) dst8[m] = rv[r][m];
      }
    }
  } else
  {
    // pack row pairs (Hermitian expansion along W), bit-reversed along W
    float2 rz[PLANE_ITEMS];
    const int total = pairs * W;
#pragma unroll
    for (int i = 0; i < PLANE_ITEMS; ++i) {
      const int q = threadIdx.x + i * PLANE_THREADS;
      if (q < total) {
        // consecutive lanes take consecutive row PAIRS of one frequency: the bit-reversed write below then walks
        // lines (odd stride, conflict-free) instead of scattering 16 lanes over two banks of one line
        const int pr = q & (pairs - 1), k = q >> (logH - 1);
        const int m = k < WH ? k : W - k;
        float2 xa = t2[m * ldh + 2 * pr], xb = t2[m * ldh + 2 * pr + 1];
        if (m == 0 || 2 * m == W) { xa.y = 0.f; xb.y = 0.f; }
        if (k >= WH) { xa.y = -xa.y; xb.y = -xb.y; }
        rz[i] = make_float2(xa.x - xb.y, xa.y + xb.x);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PLANE_ITEMS; ++i) {
      const int q = threadIdx.x + i * PLANE_THREADS;
      if (q < total) t1[(q & (pairs - 1)) * ldw + brev(q >> (logH - 1), logW)] = rz[i];
    }
  }
  // The sample's rescale factor from the partial sums (white: one slot per plane, coloured: one per block of the outer-axis
  // kernel).  Their loads were issued by the first wave when the block started (fa[]: lane l holds slots l, l + 64, ... summed in
  // that order); the first wave finishes the four sums with a fixed-order shuffle tree BEFORE the row transform -- whose barriers
  // publish the one LDS word -- so nothing of it is left on the block's critical path.  (Round 2 had EVERY thread walk all
  // slots in a rolled loop of dependent loads after the last transform: 5.5 us of each block's 24 us, tools/tune/tune_colored.hip.)
  __shared__ float factor_sh;
  if (MODE == 1 && threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (int o = 32; o > 0; o >>= 1) fa[i] += __shfl_down(fa[i], o);
    if (threadIdx.x == 0) {
      const double n = (double)a.d1 * (double)H * (double)W;
      factor_sh = a.raw ? 1.0f : rescale_factor(fa[0], fa[1], fa[2], fa[3] / n, n, a.has_energy, a.energy);
    }
  }
  if constexpr (R16_W) fft_tile_128<true, FUSE_ROW, false>(t1, tw_w, pairs);
  else fft_tile<true, FUSE_ROW>(t1, tw_w, W, logW, pairs);
  SKR_STAMP(3);
  const float scale = a.raw ? 1.0f : 1.0f / ((float)a.d1 * (float)H * (float)W);
  const float factor = MODE == 2 ? (float)s1 : factor_sh;
  SKR_STAMP(4);
  T* dst = reinterpret_cast<T*>(a.out) + ((smp * a.d1 + i1) * (int64_t)H) * W;
  // 4 consecutive values of one row per item: one 8-byte (16-bit T) or 16-byte (fp32) store
  for (int q = threadIdx.x; q < pairs * (W / 4); q += PLANE_THREADS) {
    const int pr = q >> (logW - 2), n4 = (q & (W / 4 - 1)) * 4;
    const float2 z0 = t1[pr * ldw + n4], z1 = t1[pr * ldw + n4 + 1], z2 = t1[pr * ldw + n4 + 2], z3 = t1[pr * ldw + n4 + 3];
    if constexpr (sizeof(T) <= 4) {
      store4_from_f32<T>(dst + (int64_t)(2 * pr) * W + n4, z0.x * scale * factor, z1.x * scale * factor, z2.x * scale * factor, z3.x * scale * factor);
      store4_from_f32<T>(dst + (int64_t)(2 * pr + 1) * W + n4, z0.y * scale * factor, z1.y * scale * factor, z2.y * scale * factor, z3.y * scale * factor);
    } else {
      T* da = dst + (int64_t)(2 * pr) * W + n4;
      T* db = dst + (int64_t)(2 * pr + 1) * W + n4;
      da[0] = (T)(z0.x * scale * factor); da[1] = (T)(z1.x * scale * factor); da[2] = (T)(z2.x * scale * factor); da[3] = (T)(z3.x * scale * factor);
      db[0] = (T)(z0.y * scale * factor); db[1] = (T)(z1.y * scale * factor); db[2] = (T)(z2.y * scale * factor); db[3] = (T)(z3.y * scale * factor);
    }
  }
  SKR_STAMP(5);
}

template <int MODE, typename T, int CH, int CW>
__global__ __launch_bounds__(PLANE_THREADS) void colored_plane(const ColoredArgs a, int logH_rt, int logW_rt) {
  extern __shared__ float2 smem[];
  plane_body<MODE, T, CH, CW>(a, logH_rt, logW_rt, (int64_t)blockIdx.y, (int)blockIdx.x, smem);
}

// ---- inverse planes of 128 x 128 (BASELINE config 3's unit): persistent blocks, the next plane's spectrum prefetched into registers ----
// Round 5.  colored_plane<1, T, 7, 7> spent 4.7 of a block's 12.8 us waiting for its spectrum (two 67 KiB blocks per CU cannot cover it),
// a third of its LDS cycles in bank conflicts, and ran the 65th frequency column as a second trip of the radix-16 pass for 8 lanes.
// This kernel keeps the arithmetic (8-point pass in registers + one radix-16 pass per axis) and changes the structure around it:
//   * a block loops over planes (grid = 2 blocks per CU) and issues the NEXT plane's 16 loads per thread right after the current
//     plane's registers are free, so a plane's HBM latency lies under the previous plane's transforms.  The barriers are raw
//     `s_waitcnt lgkmcnt(0); s_barrier` -- __syncthreads() would drain the loads in flight (its fence waits for vmcnt(0)) -- and the
//     result stores are compiler-visible non-temporal stores, so that the wait for the prefetched registers is a counted vmcnt(N) that
//     does not wait for the stores issued after them;
//   * the DC and Nyquist frequency columns (both Hermitian along the rows, so both transform to REAL sequences) travel as ONE complex
//     column, Z = X[:,0] + i X[:,64]: 64 columns, i.e. one radix-16 trip for every wave and no 65th line.  The imaginary parts the
//     reference's irfftn drops there (rounding residue of the forward transform) land on the other column at that same size;
//   * the sixteen twiddles of the radix-16 pass depend on k1 = the wave's index only: SGPRs, computed once per block, instead of
//     fifteen LDS reads per thread and pass;
//   * Hermitian packing: a thread takes the row pairs' frequencies k0 + 16 n AND 16 - k0 + 16 n, which mirror into each other, so
//     every column-tile element is read once (was twice); lanes are laid out (8 pairs x 2 neighbouring frequencies per 16 lanes) so
//     that those reads are conflict-free;
//   * the row tile keeps element k1 + 8 k2 of a line at slot k2 + 16 (k1 >> 2) + 32 (k1 & 3): packing writes, the radix-16 pass and
//     the final reads (4 consecutive elements per lane for 8-byte stores) are all conflict-free in that layout.
// LDS: 64 x 129 complex + the Nyquist staging line = 67 KiB, two blocks per CU.  (Reference: skrample/pytorch/noise.py:380-403.)
// per-sample rescale factors for colored_inverse128: factors[smp] from the white sums of the forward kernel (one slot per plane) and the
// Parseval sums of the outer-axis kernel (one slot per block); one wave per sample, fixed summation order
__global__ __launch_bounds__(64) void colored_factors(const ColoredArgs a, float* factors, uint32_t first_ticket) {
  const int64_t smp = blockIdx.x;
  if (smp == 0 && threadIdx.x == 0 && a.ticket) *a.ticket = first_ticket;
  double fa[4] = {0.0, 0.0, 0.0, 0.0};
  const double* pw = a.partials + smp * a.n_slots * 2;
  for (int sl = threadIdx.x; sl < a.n_slots; sl += 64) { fa[0] += pw[2 * sl]; fa[1] += pw[2 * sl + 1]; }
  const double* pc = a.partials + (int64_t)a.batch * a.n_slots * 2 + smp * a.n_slots_c * 2;
  for (int sl = threadIdx.x; sl < a.n_slots_c; sl += 64) { fa[2] += pc[2 * sl]; fa[3] += pc[2 * sl + 1]; }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int o = 32; o > 0; o >>= 1) fa[i] += __shfl_down(fa[i], o);
  if (threadIdx.x == 0) {
    const double n = (double)a.d1 * (double)a.d2 * (double)a.d3;
    factors[smp] = rescale_factor(fa[0], fa[1], fa[2], fa[3] / n, n, a.has_energy, a.energy);
  }
}

// phase stamps of the persistent kernel (tools/tune/tune_inverse128.hip; compiled out of the library): [block][plane trip < 16][16]: eight of s_memrealtime,
// the shader clock (s_memtime) and the hardware ids (HW_ID, XCC_ID) at the trip's start
#ifdef SKR_COLORED_TRACE
#define SKR_STAMP_P(i) do { if (threadIdx.x == 0 && a.trace && trip < 16) { uint64_t* t_ = a.trace + ((int64_t)blockIdx.x * 16 + trip) * 16; t_[(i)] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 0) { t_[8] = __builtin_amdgcn_s_memtime(); t_[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } } while (0)
#else
#define SKR_STAMP_P(i) do {} while (0)
#endif

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(512, 4) void colored_inverse128(const ColoredArgs a, const int64_t n_planes) {
  extern __shared__ float2 smem[];
  constexpr int H = 128, W = 128, WH = 65, LD = 129;
  float2* tile = smem;             // 64 lines x 129: frequency columns (natural order), then row pairs (slot layout above)
  float2* stage = smem + 64 * LD;  // [128] Nyquist column of the plane whose registers are in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = k1 of both radix-16 passes
  float twc[16], tws[16];  // exp(+2 pi i n2 k1 / 128): wave-uniform
  {
    // the circle once per block (one sincospif per thread instead of fifteen), through the staging line, which is free until the first plane
    if (tid < H) {
      float sn, cs;
      sincospif(2.0f * (float)tid / 128.0f, &sn, &cs);
      stage[tid] = make_float2(cs, sn);
    }
    lds_barrier();
#pragma unroll
    for (int n2 = 1; n2 < 16; ++n2) {
      const float2 w = stage[n2 * wave];  // n2 k1 <= 105
      twc[n2] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.x)));
      tws[n2] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.y)));
    }
    lds_barrier();
  }
  const int r0a = (int)brev((unsigned)wave, 4), r0b = (int)brev((unsigned)wave + 8u, 4);  // first row of this thread's two 8-row groups

  float2 lv[2][8];
  float2 nyq = make_float2(0.f, 0.f);
  // Every address below is a wave-uniform base plus a 32-bit lane offset (the saddr form of the load), recomputed where it is used: a 64-bit
  // pointer per load kept across the four phases under the prefetch would push the kernel past its 128 registers.
  const uint32_t lane8 = (uint32_t)lane * 8u;
  // Planes are taken from the LAST one down: the outer-axis kernel in front of this one wrote the spectrum first sample first, so the last
  // planes are the ones still in the Infinity Cache (r05 timeline: 83.9 us against 92.5 first plane first, same static deal otherwise).
#ifdef SKR_INV_FORWARD_ORDER  // experiment (tools/tune/tune_inverse128.hip)
  auto pl = [&](int64_t x) { return x; };
#else
  auto pl = [&](int64_t x) { return n_planes - 1 - x; };
#endif
  auto prefetch = [&](int64_t q) {
#ifdef SKR_INV_NOLOAD  // experiment: no spectrum traffic (results are garbage) -- what the transforms alone cost
    if (a.n_slots >= 0) return;
#endif
    const char* plane = reinterpret_cast<const char*>(a.spec) + pl(q) * (int64_t)(H * WH * 8);
    // (first, so that its use waits for nothing younger; every thread loads -- row tid & 127 -- to keep the loop free of divergent branches,
    //  behind which the compiler's vmcnt bookkeeping falls back to "wait for everything")
    nyq = *reinterpret_cast<const float2*>(plane + ((uint32_t)(tid & (H - 1)) * (uint32_t)(WH * 8) + (uint32_t)(W / 2 * 8)));
    const char* pa = plane + r0a * (WH * 8);
    const char* pb = plane + r0b * (WH * 8);
#pragma unroll
    for (int n = 0; n < 8; ++n) lv[0][n] = *reinterpret_cast<const float2*>(pa + (lane8 + (uint32_t)(n * 16 * WH * 8)));
#pragma unroll
    for (int n = 0; n < 8; ++n) lv[1][n] = *reinterpret_cast<const float2*>(pb + (lane8 + (uint32_t)(n * 16 * WH * 8)));
  };

  // Planes are claimed two trips ahead from a device counter (a.ticket; the first two trips of every block are fixed, the counter starts behind
  // them): of the two blocks a CU holds, the older one wins the issue arbitration and runs ~2x as fast as the younger while both are there --
  // dealt out statically, the older block ran out of planes after 60 % of the kernel and left its half of the CU idle (r05 timeline).
  __shared__ int32_t next_sh;
  const int64_t stride = gridDim.x;
  int64_t p = blockIdx.x, pn = p + stride;
  if (p >= n_planes) return;
  prefetch(p);
  {
    // Eight placeholder stores (zeros, to the very addresses this thread's first plane goes to: same thread, same address, program order -- the
    // results overwrite them).  Why: inside the loop the prefetched registers are consumed with the eight result stores of the previous plane
    // behind them in the queue, and the wait should be vmcnt(8), not vmcnt(0) -- which would hold every plane until the previous plane's
    // stores have completed.  The compiler emits ONE wait for both ways into the loop, so the way in from here must look like the back edge.
    char* dst = reinterpret_cast<char*>(a.out) + pl(p) * (int64_t)(H * W * sizeof(T));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      T* da = reinterpret_cast<T*>(dst + (uint32_t)((2 * ((tid >> 5) + 16 * it) * W + 4 * (2 * (tid & 15) + ((tid >> 4) & 1))) * (int)sizeof(T)));
      if constexpr (sizeof(T) == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u32x2{0u, 0u}, reinterpret_cast<u32x2*>(da));
        __builtin_nontemporal_store(u32x2{0u, 0u}, reinterpret_cast<u32x2*>(da + W));
      } else {
        __builtin_nontemporal_store(pk_f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<pk_f32x4*>(da));
        __builtin_nontemporal_store(pk_f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<pk_f32x4*>(da + W));
      }
    }
  }
  stage[tid & (H - 1)] = nyq;
  lds_barrier();
  const float scale = a.raw ? 1.0f : 1.0f / ((float)a.d1 * (float)H * (float)W);
  // packing items: k0 = the frequency residue mod 16 (0..7; its partner is 16 - k0, or 8 for k0 = 0), pr = the row pair
  const int k0 = ((tid >> 3) & 1) | (((tid >> 7) & 3) << 1);
  const int pr = (tid & 7) | (((tid >> 4) & 7) << 3);
  const int gA = (int)brev((unsigned)k0, 4), gB = (int)brev(k0 ? 16u - (unsigned)k0 : 8u, 4);

  [[maybe_unused]] int trip = 0;
  for (;;) {
    const int64_t smp = pl(p) / a.d1;
    SKR_STAMP_P(0);

    // ---- A: stages 0-2 of the column transform on the prefetched registers; column 0 becomes X[:,0] + i X[:,64]
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const float2 ba = stage[r0a + 16 * n], bb = stage[r0b + 16 * n];
        lv[0][n] = make_float2(lv[0][n].x - ba.y, lv[0][n].y + ba.x);
        lv[1][n] = make_float2(lv[1][n].x - bb.y, lv[1][n].y + bb.x);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      dft8<true>(lv[r]);
      float2* col = tile + lane * LD + 8 * (wave + 8 * r);
#pragma unroll
      for (int m = 0; m < 8; ++m) col[m] = lv[r][m];
    }
    // the sample's rescale factor (colored_factors), loaded BEFORE the prefetch goes out: loads return in order, so anything issued behind
    // the prefetch and needed before the next plane would wait for all of it
    // the plane of the trip after next: one returning atomic by one lane, as inline assembly -- the compiler does not know it is in flight, so
    // it adds no wait of its own; ours, below, counts the (at most) 18 loads issued behind it
    uint32_t claimed = 0;
    if (a.ticket != nullptr && tid == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(claimed) : "v"(0u), "v"(1u), "s"(a.ticket) : "memory");
    const float factor = a.raw ? 1.0f : a.factors[smp];
    const bool more = pn < n_planes;
    prefetch(more ? pn : p);  // lands under the four phases below (last plane of the block: itself again, unused)
    SKR_STAMP_P(1);
    lds_barrier();
    SKR_STAMP_P(2);
    if (tid == 0) {
      if (a.ticket != nullptr) {
        asm volatile("s_waitcnt vmcnt(17)" : "+v"(claimed) : : "memory");
        next_sh = (int32_t)claimed;
      } else {
        next_sh = (int32_t)(pn + stride < n_planes ? pn + stride : n_planes);
      }
    }
    // ---- B: radix-16 pass of the columns (item = (line = lane, k1 = wave)): X[k1 + 8 k2] -> position k1 + 8 k2
    {
      float2* q = tile + lane * LD + wave;
      float2 v[16];
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) v[n2] = q[8 * (int)(__builtin_bitreverse32((unsigned)n2) >> 28)];
#pragma unroll
      for (int n2 = 1; n2 < 16; ++n2) v[n2] = make_float2(v[n2].x * twc[n2] - v[n2].y * tws[n2], v[n2].x * tws[n2] + v[n2].y * twc[n2]);
      dft16<true>(v);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) q[8 * k2] = v[k2];
    }
    SKR_STAMP_P(3);
    lds_barrier();
    // ---- C: Hermitian packing of the row pairs + stages 0-2 of the row transform
    float2 A[8], B[8];
    {
      float2 xa[8], xb[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2* sa = tile + (k0 + 16 * j) * LD + 2 * pr;
        xa[j] = sa[0]; xb[j] = sa[1];
        const float2* sb = tile + (k0 ? 16 * (j + 1) - k0 : 8 + 16 * j) * LD + 2 * pr;
        xa[4 + j] = sb[0]; xb[4 + j] = sb[1];
      }
      auto direct = [&](int i) { return make_float2(xa[i].x - xb[i].y, xa[i].y + xb[i].x); };  // X_a + i X_b
      auto mirror = [&](int i) { return make_float2(xa[i].x + xb[i].y, xb[i].x - xa[i].y); };  // conj X_a + i conj X_b
#pragma unroll
      for (int j = 0; j < 4; ++j) { A[j] = direct(j); B[j] = direct(4 + j); }
      if (k0 != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { A[4 + i] = mirror(7 - i); B[4 + i] = mirror(3 - i); }
      } else {  // residue 0: DC and Nyquist are the real and imaginary parts of the packed column; its partner residue 8 mirrors into itself
        A[0] = make_float2(xa[0].x, xb[0].x);
        A[4] = make_float2(xa[0].y, xb[0].y);
#pragma unroll
        for (int i = 1; i < 4; ++i) A[4 + i] = mirror(4 - i);
#pragma unroll
        for (int i = 0; i < 4; ++i) B[4 + i] = mirror(7 - i);
      }
    }
    dft8<true>(A);
    dft8<true>(B);
    SKR_STAMP_P(4);
    lds_barrier();  // every thread has read the column tile: the row tile may overwrite it
    {
      float2* la = tile + pr * LD + gA;
      float2* lb = tile + pr * LD + gB;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int slot = 16 * (m >> 2) + 32 * (m & 3);
        la[slot] = A[m];
        lb[slot] = B[m];
      }
    }
    lds_barrier();
    SKR_STAMP_P(5);
    // ---- D: radix-16 pass of the row pairs, in the slot layout
    {
      float2* q = tile + lane * LD + 16 * (wave >> 2) + 32 * (wave & 3);
      float2 v[16];
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) v[n2] = q[(int)(__builtin_bitreverse32((unsigned)n2) >> 28)];
#pragma unroll
      for (int n2 = 1; n2 < 16; ++n2) v[n2] = make_float2(v[n2].x * twc[n2] - v[n2].y * tws[n2], v[n2].x * tws[n2] + v[n2].y * twc[n2]);
      dft16<true>(v);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) q[k2] = v[k2];
    }
    lds_barrier();
    SKR_STAMP_P(6);
    // ---- E: the next plane's Nyquist column goes to its staging line; scale and store (row 2 pr from the real, 2 pr + 1 from the imaginary parts)
    stage[tid & (H - 1)] = nyq;
    const float f = scale * factor;
    char* dst = reinterpret_cast<char*>(a.out) + pl(p) * (int64_t)(H * W * sizeof(T));
    // lane -> the four elements 4 j .. 4 j + 3 of a row, j = 2 (lane & 15) + ((lane >> 4) & 1): sixteen consecutive lanes then read sixteen
    // consecutive slots (the compiler pairs the reads into ds_read2_b64, whose bank groups are 16 lanes wide), a half wave still covers its row
    const int j = 2 * (tid & 15) + ((tid >> 4) & 1);
    const float2* src = tile + (j >> 1) + 16 * (j & 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int line = (tid >> 5) + 16 * it;
      const float2* z = src + line * LD;
      const float2 z0 = z[0], z1 = z[32], z2 = z[64], z3 = z[96];
      T* da = reinterpret_cast<T*>(dst + (uint32_t)((2 * line * W + 4 * j) * (int)sizeof(T)));
#ifdef SKR_INV_NOSTORE  // experiment: results stay on the chip (one store per block keeps the arithmetic alive)
      if (z0.x != 12345.678f) continue;
#endif
#ifdef SKR_INV_STEP_MODEL  // experiment (tools/tune/tune_inverse128_step.hip): the traffic and arithmetic of BASELINE config 3's two-output
      // step as this plane's epilogue -- seven 16-bit and two fp32 operands read at the plane's own offsets beside the noise in registers, an fp32 state
      // and a 16-bit result written; the values are a model (fixed coefficients), the bytes per element (22 read, 6 written, noise never stored) are the step's
      if constexpr (sizeof(T) == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const int64_t e = pl(p) * (int64_t)(H * W) + (2 * line * W + 4 * j);
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          float nz[4] = {row ? z0.y * f : z0.x * f, row ? z1.y * f : z1.x * f, row ? z2.y * f : z2.x * f, row ? z3.y * f : z3.x * f};
          const u32x2 nq = u32x2{pack_pair<T>(nz[0], nz[1]), pack_pair<T>(nz[2], nz[3])};  // the noise as the tensor dtype holds it
          float s0[4], s1[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = __uint_as_float(i & 1 ? nq[i >> 1] & 0xFFFF0000u : nq[i >> 1] << 16);
            s0[i] = 0.37f * v; s1[i] = -0.11f * v;
          }
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const u32x2 w = *(reinterpret_cast<const u32x2*>(g_step_model.narrow[k]) + ((e + row * W) >> 2));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = __uint_as_float(i & 1 ? w[i >> 1] & 0xFFFF0000u : w[i >> 1] << 16);
              s0[i] = fmaf(0.05f * (float)(k + 1), v, s0[i]); s1[i] = fmaf(-0.03f * (float)(k + 2), v, s1[i]);
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const pk_f32x4 w = *(reinterpret_cast<const pk_f32x4*>(g_step_model.wide[k]) + ((e + row * W) >> 2));
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = fmaf(0.4f, w[i], s0[i]); s1[i] = fmaf(0.2f, w[i], s1[i]); }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) s1[i] = fmaf(0.9f, s0[i], s1[i]);  // (the chained second output)
          __builtin_nontemporal_store(pk_f32x4{s0[0], s0[1], s0[2], s0[3]}, reinterpret_cast<pk_f32x4*>(g_step_model.state) + ((e + row * W) >> 2));
          __builtin_nontemporal_store(u32x2{pack_pair<T>(s1[0], s1[1]), pack_pair<T>(s1[2], s1[3])}, reinterpret_cast<u32x2*>(da + row * W));
        }
        continue;
      }
#endif
      if constexpr (sizeof(T) == 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u32x2{pack_pair<T>(z0.x * f, z1.x * f), pack_pair<T>(z2.x * f, z3.x * f)}, reinterpret_cast<u32x2*>(da));
        __builtin_nontemporal_store(u32x2{pack_pair<T>(z0.y * f, z1.y * f), pack_pair<T>(z2.y * f, z3.y * f)}, reinterpret_cast<u32x2*>(da + W));
      } else {
        __builtin_nontemporal_store(pk_f32x4{z0.x * f, z1.x * f, z2.x * f, z3.x * f}, reinterpret_cast<pk_f32x4*>(da));
        __builtin_nontemporal_store(pk_f32x4{z0.y * f, z1.y * f, z2.y * f, z3.y * f}, reinterpret_cast<pk_f32x4*>(da + W));
      }
    }
    SKR_STAMP_P(7);
    ++trip;
    lds_barrier();  // the tile and the staging line are free for the next plane
    p = pn;
    pn = __builtin_amdgcn_readfirstlane(next_sh);
    if (p >= n_planes) break;  // (the prefetch of a block's last trip re-read its own plane)
  }
}

// ---- planes whose sides are 2^a * r, r odd <= 63 (96, 112, 144, 152, 160, 168, 192 ... : latents of 768 / 896 / 1152 / 1216 / 1280 /
// 1344 / 1536-pixel images) ----
// Round 3.  Cooley-Tukey with ONE odd factor per axis: n = r * m (r odd, 1 ... 63, m = 2^a >= 2).  A line is kept as r sub-lines of m
// points (+1 pad each): sub-line q holds x[r j + q] in bit-reversed order, the power-of-two transform above runs over all
// sub-lines at once (they are just r * L lines of m points), and one more pass combines them,
//     X[k2 + m j] = sum_q  w_n^(q k2) * w_r^(q j) * Y_q[k2],
// in place at the sub-line positions -- so natural index k lives at k + k / m (`mixed_nat`), and an input element n goes to
// (n % r) * (m + 1) + bitrev(n / r) (`mixed_pos`).  Everything around the transforms is the plane kernel's scheme (drawn in
// place, two real rows per complex row transform, transposes staged through registers into the same LDS region, Parseval
// rescale), with runtime sizes; the outer (channel) axis is the same register kernel.  hipFFT was the only route for these
// shapes before: 13.9 Gelem/s at 64 x (4, 96, 96) against 95 Gelem/s at 64 x (4, 128, 128).
struct MixedAxis {
  int32_t n, r, a, m, pitch;  // n = r << a, m = 1 << a, pitch = r * (m + 1)
  uint32_t magic_r;           // v / r == umulhi(v, magic_r) for v < 2^16
};
__device__ __forceinline__ int mixed_pos(const MixedAxis& x, int n) {
  const int hi = x.r == 1 ? n : (int)__umulhi((uint32_t)n, x.magic_r);
  return (n - hi * x.r) * (x.m + 1) + (int)brev((unsigned)hi, x.a);
}
__device__ __forceinline__ int mixed_nat(const MixedAxis& x, int k) { return k + (k >> x.a); }
// q / d by multiply-high with magic = ceil(2^32 / d) (exact for q < 2^16); d = 1 has no 32-bit magic (it wraps to 0): q itself
__device__ __forceinline__ int div_magic(int q, uint32_t magic) { return magic == 0u ? q : (int)__umulhi((uint32_t)q, magic); }

// the combining pass of the odd factor, over `lines` lines of the tile (in place; r = 1: nothing to do).
// r = 3, 5: one thread per output group, the r twiddled inputs in registers.
// other r (7 ... 63): one LANE per output, the r outputs of a group on adjacent lanes of ONE wave in the same iteration,
//     X[k] = sum_q  w_n^(q k mod n) * Y_q[k mod m],
// read from LDS by all r lanes (same address: a broadcast) and written back over the inputs.  A wave's LDS instructions
// execute in order and every lane's store depends on all of its loads, so no lane's store can pass another lane's load of
// the same group: in place without a block barrier, for any r at run time, and no register array indexed by r.
template <bool INVERSE>
__device__ __forceinline__ void mixed_combine(float2* tile, const float2* tw_full, const MixedAxis& x, int lines) {
  if (x.r == 1) return;
  if (x.r <= 5) {
    const int total = lines << x.a;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      const int line = t >> x.a, k2 = t & (x.m - 1);
      float2* p = tile + line * x.pitch + k2;
      float2 y[5];
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (q < x.r) y[q] = q == 0 ? p[0] : cmul(p[q * (x.m + 1)], twid<INVERSE>(tw_full, q * k2));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < x.r) {
          float2 acc = y[0];
#pragma unroll
          for (int q = 1; q < 5; ++q) {
            if (q < x.r) {
              int e = q * j;
              e -= (e >= 3 * x.r) ? 3 * x.r : ((e >= 2 * x.r) ? 2 * x.r : ((e >= x.r) ? x.r : 0));  // (q j) mod r, q j <= 16 < 4 r
              const float2 w = twid<INVERSE>(tw_full, e * x.m);
              const float2 t2 = cmul(y[q], w);
              acc = make_float2(acc.x + t2.x, acc.y + t2.y);
            }
          }
          p[j * (x.m + 1)] = acc;
        }
      }
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int gl = (int)__umulhi((uint32_t)lane, x.magic_r), j = lane - gl * x.r;  // group within the wave's batch, output within the group
    const int per_wave = 64 / x.r, groups = lines << x.a;
    for (int base = wave * per_wave; base < groups; base += waves * per_wave) {
      const int grp = base + gl;
      const bool on = gl < per_wave && grp < groups;
      const int line = grp >> x.a, k2 = grp & (x.m - 1), k = k2 + (j << x.a);
      float2* p = tile + (on ? line * x.pitch + k2 : 0);
      float2 acc = p[0];
      int e = 0;  // (q k) mod n
#pragma unroll 2
      for (int q = 1; q < x.r; ++q) {
        e += k;
        e -= e >= x.n ? x.n : 0;
        const float2 t2 = cmul(p[q * (x.m + 1)], twid<INVERSE>(tw_full, on ? e : 0));
        acc = make_float2(acc.x + t2.x, acc.y + t2.y);
      }
      __builtin_amdgcn_wave_barrier();
      if (on) p[j * (x.m + 1)] = acc;
    }
  }
  __syncthreads();
}

struct MixedGeom {
  MixedAxis h, w;
  uint32_t magic_wh;  // v / (w.n / 2 + 1)
};

// MODE as colored_plane: 0 white -> half spectrum, 1 half spectrum -> real plane (result dtype), 2 both (2-D units)
// THREADS: 512 (two blocks per CU for planes up to ~128 x 128) or 1024 (one block: 160- and 192-point sides)
constexpr MixedAxis const_mixed_axis(int d) {
  int v = d;
  while (v % 2 == 0) v /= 2;
  const int r = v;
  v = d / r;
  int lg = 0;
  while ((1 << lg) < v) ++lg;
  return MixedAxis{d, r, lg, v, r * (v + 1), (uint32_t)((0x100000000ull + (uint32_t)r - 1) / (uint32_t)r)};
}

template <int MODE, typename T, int THREADS, int SIDE = 0>
__global__ __launch_bounds__(THREADS) void colored_plane_mixed(const ColoredArgs a, const MixedGeom g) {
  extern __shared__ float2 smem[];
  // SIDE: square planes of the common sizes with their geometry as compile-time constants -- every pitch multiply, digit
  // reversal and loop bound folds (the same cure as colored_plane's CH / CW: 256 x (16, 96, 96) 0.389 -> 0.259 ms)
  constexpr MixedAxis cax = const_mixed_axis(SIDE ? SIDE : 4);
  const MixedAxis ax_h = SIDE ? cax : g.h, ax_w = SIDE ? cax : g.w;
  const uint32_t magic_wh = SIDE ? (uint32_t)((0x100000000ull + (uint32_t)(SIDE / 2 + 1) - 1) / (uint32_t)(SIDE / 2 + 1)) : g.magic_wh;
  const int H = ax_h.n, W = ax_w.n, WH = W / 2 + 1, ldw = ax_w.pitch, ldh = ax_h.pitch, pairs = H >> 1;
  float2* tw_w = smem;                 // full circle, W entries
  float2* tw_h = tw_w + W;             // full circle, H entries
  float2* tw_ws = tw_h + H;            // sub-transform tables: m / 2 entries each
  float2* tw_hs = tw_ws + (ax_w.m >> 1);
  float2* t1 = tw_hs + (ax_h.m >> 1);  // row-pair tile: pairs lines of pitch ldw
  float2* t2 = t1;                     // column tile: WH lines of pitch ldh (aliases t1)
  const int64_t smp = blockIdx.y;
  const int i1 = blockIdx.x;
  double fa[4] = {0.0, 0.0, 0.0, 0.0};
  if (MODE == 1 && !a.raw && threadIdx.x < 64) {  // the sample's partial sums: loaded now, reduced before the last transform (see colored_plane)
    const double* pw = a.partials + (0 * a.batch + smp) * a.n_slots * 2;
    for (int sl = threadIdx.x; sl < a.n_slots; sl += 64) { fa[0] += pw[2 * sl]; fa[1] += pw[2 * sl + 1]; }
    const double* pc = a.partials + (int64_t)a.batch * a.n_slots * 2 + smp * a.n_slots_c * 2;
    for (int sl = threadIdx.x; sl < a.n_slots_c; sl += 64) { fa[2] += pc[2 * sl]; fa[3] += pc[2 * sl + 1]; }
  }
  make_twiddles_full(tw_w, W);
  make_twiddles_full(tw_h, H);
  make_twiddles(tw_ws, ax_w.m);
  make_twiddles(tw_hs, ax_h.m);
  float2* plane = a.spec + ((smp * a.d1 + i1) * (int64_t)H) * WH;
  double s1 = 0.0, s2 = 0.0;

  if (MODE != 1) {
    const uint64_t seed = a.seeds[smp];
    const int quads = W >> 2;
    const uint32_t magic_q = (uint32_t)((0x100000000ull + (uint32_t)quads - 1) / (uint32_t)quads);
    for (int q = threadIdx.x; q < pairs * quads; q += THREADS) {
      const int pr = div_magic(q, magic_q), n4 = (q - pr * quads) * 4;
      const int64_t ea = ((int64_t)i1 * H + 2 * pr) * W + n4;
      float za[4], zb[4];
      normal4(seed, a.stream, (uint64_t)ea >> 2, za);
      normal4(seed, a.stream, (uint64_t)(ea + W) >> 2, zb);
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t1[pr * ldw + mixed_pos(ax_w, n4 + j)] = make_float2(za[j], zb[j]);
        p1 += za[j] + zb[j];
        p2 = __builtin_fmaf(za[j], za[j], __builtin_fmaf(zb[j], zb[j], p2));
      }
      s1 += (double)p1; s2 += (double)p2;
    }
    fft_tile<false>(t1, tw_ws, ax_w.m, ax_w.a, pairs * ax_w.r);
    mixed_combine<false>(t1, tw_w, ax_w, pairs);
    {
      // untangle the row pairs into the column tile (digit-reversed along H)
      float2 ra[PLANE_ITEMS], rb[PLANE_ITEMS];
      const int total = pairs * WH;
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * THREADS;
        if (q < total) {
          const int pr = div_magic(q, magic_wh), k = q - pr * WH;
          const int kn = k == 0 ? 0 : W - k;
          const float2 zk = t1[pr * ldw + mixed_nat(ax_w, k)], zn = t1[pr * ldw + mixed_nat(ax_w, kn)];
          ra[i] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
          rb[i] = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * THREADS;
        if (q < total) {
          const int pr = div_magic(q, magic_wh), k = q - pr * WH;
          t2[k * ldh + mixed_pos(ax_h, 2 * pr)] = ra[i];
          t2[k * ldh + mixed_pos(ax_h, 2 * pr + 1)] = rb[i];
        }
      }
    }
    fft_tile<false>(t2, tw_hs, ax_h.m, ax_h.a, WH * ax_h.r);
    mixed_combine<false>(t2, tw_h, ax_h, WH);
    if (MODE == 0) {
      block_sums(s1, s2, a.partials + ((0 * a.batch + smp) * a.n_slots + i1) * 2);
      for (int q = threadIdx.x; q < H * WH; q += THREADS) {
        const int row = div_magic(q, magic_wh), k = q - row * WH;
        plane[q] = t2[k * ldh + mixed_nat(ax_h, row)];
      }
      return;
    }
    // MODE 2: weights in place (+ Parseval sums), then the columns go back to digit-reversed order for the inverse transform
    double p1 = 0.0, p2 = 0.0;
    const uint32_t magic_h = (uint32_t)((0x100000000ull + (uint32_t)H - 1) / (uint32_t)H);
    {
      float2 rz[PLANE_ITEMS];
      const int total = H * WH;
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * THREADS;
        if (q < total) {
          const int k = div_magic(q, magic_h), row = q - k * H;
          const float f2 = axis_freq(row, H), f3 = (float)k / (float)W;
          const float wgt = radial_weight(f2 * f2 + f3 * f3, a.inv_rmax, a.eps_clip, a.exponent_half_neg);
          float2 v = t2[k * ldh + mixed_nat(ax_h, row)];
          v = make_float2(v.x * wgt, v.y * wgt);
          rz[i] = v;
          const float e = __builtin_fmaf(v.x, v.x, v.y * v.y);
          p2 += (double)((k == 0 || 2 * k == W) ? e : 2.f * e);
          if (q == 0) p1 = (double)v.x;
        }
      }
      {
        double tot[4] = {s1, s2, p1, p2};
        block_sums4_bcast(tot);  // (also the barrier between the reads above and the writes below)
        const double n = (double)H * (double)W;
        s1 = (double)rescale_factor(tot[0], tot[1], tot[2], tot[3] / n, n, a.has_energy, a.energy);  // s1 now carries the factor
      }
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = threadIdx.x + i * THREADS;
        if (q < total) {
          const int k = div_magic(q, magic_h), row = q - k * H;
          t2[k * ldh + mixed_pos(ax_h, row)] = rz[i];
        }
      }
    }
  } else {
    const int total = H * WH;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float2 rz[PLANE_ITEMS];
      const int base = half * THREADS * PLANE_ITEMS;
      if (base >= total) break;
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = base + threadIdx.x + i * THREADS;
        if (q < total) rz[i] = plane[q];
      }
#pragma unroll
      for (int i = 0; i < PLANE_ITEMS; ++i) {
        const int q = base + threadIdx.x + i * THREADS;
        if (q < total) {
          const int row = div_magic(q, magic_wh), k = q - row * WH;
          t2[k * ldh + mixed_pos(ax_h, row)] = rz[i];
        }
      }
    }
  }

  fft_tile<true>(t2, tw_hs, ax_h.m, ax_h.a, WH * ax_h.r);
  mixed_combine<true>(t2, tw_h, ax_h, WH);
  {
    // pack row pairs (Hermitian expansion along W), digit-reversed along W; consecutive lanes = consecutive pairs of one frequency
    float2 rz[PLANE_ITEMS];
    const int total = pairs * W;
    const uint32_t magic_p = (uint32_t)((0x100000000ull + (uint32_t)pairs - 1) / (uint32_t)pairs);
#pragma unroll
    for (int i = 0; i < PLANE_ITEMS; ++i) {
      const int q = threadIdx.x + i * THREADS;
      if (q < total) {
        const int k = div_magic(q, magic_p), pr = q - k * pairs;
        const int m = k < WH ? k : W - k;
        float2 xa = t2[m * ldh + mixed_nat(ax_h, 2 * pr)], xb = t2[m * ldh + mixed_nat(ax_h, 2 * pr + 1)];
        if (m == 0 || 2 * m == W) { xa.y = 0.f; xb.y = 0.f; }
        if (k >= WH) { xa.y = -xa.y; xb.y = -xb.y; }
        rz[i] = make_float2(xa.x - xb.y, xa.y + xb.x);
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PLANE_ITEMS; ++i) {
      const int q = threadIdx.x + i * THREADS;
      if (q < total) {
        const int k = div_magic(q, magic_p), pr = q - k * pairs;
        t1[pr * ldw + mixed_pos(ax_w, k)] = rz[i];
      }
    }
  }
  __shared__ float factor_sh;
  if (MODE == 1 && threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (int o = 32; o > 0; o >>= 1) fa[i] += __shfl_down(fa[i], o);
    if (threadIdx.x == 0) {
      const double n = (double)a.d1 * (double)H * (double)W;
      factor_sh = a.raw ? 1.0f : rescale_factor(fa[0], fa[1], fa[2], fa[3] / n, n, a.has_energy, a.energy);
    }
  }
  fft_tile<true>(t1, tw_ws, ax_w.m, ax_w.a, pairs * ax_w.r);
  mixed_combine<true>(t1, tw_w, ax_w, pairs);
  const float scale = a.raw ? 1.0f : 1.0f / ((float)a.d1 * (float)H * (float)W);
  const float factor = MODE == 2 ? (float)s1 : factor_sh;
  T* dst = reinterpret_cast<T*>(a.out) + ((smp * a.d1 + i1) * (int64_t)H) * W;
  const int quads = W >> 2;
  const uint32_t magic_q = (uint32_t)((0x100000000ull + (uint32_t)quads - 1) / (uint32_t)quads);
  for (int q = threadIdx.x; q < pairs * quads; q += THREADS) {
    const int pr = div_magic(q, magic_q), n4 = (q - pr * quads) * 4;
    const float2* line = t1 + pr * ldw;
    const float2 z0 = line[mixed_nat(ax_w, n4)], z1 = line[mixed_nat(ax_w, n4 + 1)], z2 = line[mixed_nat(ax_w, n4 + 2)], z3 = line[mixed_nat(ax_w, n4 + 3)];
    if constexpr (sizeof(T) <= 4) {
      store4_from_f32<T>(dst + (int64_t)(2 * pr) * W + n4, z0.x * scale * factor, z1.x * scale * factor, z2.x * scale * factor, z3.x * scale * factor);
      store4_from_f32<T>(dst + (int64_t)(2 * pr + 1) * W + n4, z0.y * scale * factor, z1.y * scale * factor, z2.y * scale * factor, z3.y * scale * factor);
    } else {
      T* da = dst + (int64_t)(2 * pr) * W + n4;
      T* db = dst + (int64_t)(2 * pr + 1) * W + n4;
      da[0] = (T)(z0.x * scale * factor); da[1] = (T)(z1.x * scale * factor); da[2] = (T)(z2.x * scale * factor); da[3] = (T)(z3.x * scale * factor);
      db[0] = (T)(z0.y * scale * factor); db[1] = (T)(z1.y * scale * factor); db[2] = (T)(z2.y * scale * factor); db[3] = (T)(z3.y * scale * factor);
    }
  }
}

// ---- pass B / C / D: a strided axis of length N; lines start at consecutive complex positions --------------------
//   element n of line q (within a sample):  (q / inner) * outer + (q % inner) + n * stride
// MODE 0 forward, 1 inverse, 2 forward + radial weights + inverse (outermost axis)
template <int MODE>
__global__ __launch_bounds__(FFT_THREADS) void colored_strided_axis(const ColoredArgs a, int N, int logN, int logL, int64_t n_lines, int64_t inner, int64_t outer, int64_t stride, int axis /*1 or 2*/) {
  const int L = 1 << logL;
  extern __shared__ float2 smem[];
  const int ld = N + 1;
  float2* tw = smem;
  float2* buf = smem + N / 2;
  const int64_t smp = blockIdx.y;
  const int64_t line0 = (int64_t)blockIdx.x * L;
  const int lines = (int)((n_lines - line0) < L ? (n_lines - line0) : L);
  float2* base = a.spec + smp * (int64_t)a.d1 * a.d2 * a.d3h;
  make_twiddles(tw, N);
  // coalesced along the line index j (adjacent lines are adjacent in memory).  Per-thread line offsets are
  // loop invariant when FFT_THREADS is a multiple of L (L is a power of two <= 64).
  const int jt = threadIdx.x & (L - 1), nt0 = threadIdx.x >> logL, dn = FFT_THREADS >> logL;
  const int64_t qt = line0 + jt;
  const int64_t off_t = (qt / inner) * outer + (qt % inner);
  const bool live = jt < lines;
  if (live) {
    // eight loads in flight per lane (a rolled loop pays one HBM latency per element: 62 -> see DESIGN, 64 x (4,256,256))
    for (int n0 = nt0; n0 < N; n0 += 8 * dn) {
      float2 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = n0 + u * dn;
        if (n < N) r[u] = base[off_t + (int64_t)n * stride];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = n0 + u * dn;
        if (n < N) buf[jt * ld + brev(n, logN)] = r[u];
      }
    }
  }
  fft_tile<MODE == 1>(buf, tw, N, logN, lines);
  if (MODE == 2) {
    // natural-order spectrum along this axis: weight, then bit-reverse in place for the inverse transform
    for (int q = threadIdx.x; q < lines * N; q += FFT_THREADS) {
      const int j = q >> logN, k = q & (N - 1);
      const int64_t ql = line0 + j;
      // frequency coordinates of this line's other axes
      int k2, k3;
      float f1, f2, f3;
      if (axis == 1) {  // 3-D: this axis is d1, line position = (k2, k3)
        k2 = (int)(ql / a.d3h); k3 = (int)(ql - (int64_t)k2 * a.d3h);
        f1 = axis_freq(k, a.d1); f2 = axis_freq(k2, a.d2);
      } else {          // 2-D: this axis is d2, line position = k3
        k3 = (int)ql; f1 = 0.f; f2 = axis_freq(k, a.d2);
      }
      f3 = (float)k3 / (float)a.d3;
      const float wgt = radial_weight(f1 * f1 + f2 * f2 + f3 * f3, a.inv_rmax, a.eps_clip, a.exponent_half_neg);
      float2 v = buf[j * ld + k];
      buf[j * ld + k] = make_float2(v.x * wgt, v.y * wgt);
    }
    __syncthreads();
    // in-place bit reversal (swap pairs once)
    for (int q = threadIdx.x; q < lines * N; q += FFT_THREADS) {
      const int j = q >> logN, k = q & (N - 1);
      const int r = (int)brev(k, logN);
      if (k < r) { float2 t = buf[j * ld + k]; buf[j * ld + k] = buf[j * ld + r]; buf[j * ld + r] = t; }
    }
    fft_tile<true>(buf, tw, N, logN, lines);
  }
  if (live) {
    for (int n = nt0; n < N; n += dn) base[off_t + (int64_t)n * stride] = buf[jt * ld + n];
  }
}

// ---- pass C for short outer axes (d1 <= 16, i.e. the channel axis of every latent): one LINE PER LANE --------
// The whole column lives in registers: 2..16-point forward DFT, radial weights, inverse DFT, with no LDS and
// no barrier.  Adjacent lanes own adjacent columns, so every load/store instruction is fully coalesced.
// one column of the outer axis (element n at base[q + n * cols]): forward, radial weights, Parseval terms, inverse -- in registers
template <int N, bool PACE = false>
__device__ __forceinline__ void outer_column(const ColoredArgs& a, float2 (&v)[N], int64_t q, double& p1, double& p2) {
  dft_n<N, false>(v);
  const int k2 = (int)(q / a.d3h), k3 = (int)(q - (int64_t)k2 * a.d3h);
  const float f2 = axis_freq(k2, a.d2), f3 = (float)k3 / (float)a.d3;
  const float rest = f2 * f2 + f3 * f3;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float f1 = axis_freq(k, N);
    const float wgt = radial_weight(f1 * f1 + rest, a.inv_rmax, a.eps_clip, a.exponent_half_neg);
    v[k] = make_float2(v[k].x * wgt, v[k].y * wgt);
    if constexpr (PACE) __builtin_amdgcn_sched_barrier(0);  // (sixteen weights in flight at once cost 50 registers)
  }
  {
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) e = __builtin_fmaf(v[k].x, v[k].x, __builtin_fmaf(v[k].y, v[k].y, e));
    p2 += (double)((k3 == 0 || 2 * k3 == a.d3) ? e : 2.f * e);
    if (q == 0) p1 = (double)v[0].x;
  }
  dft_n<N, true>(v);
}

// ---- passes B + C + D in ONE tile residency: a short channel axis over planes too large for the fused plane kernels ------------
// (4, 256, 256): the separate-axis route sends the half spectrum through HBM five times -- row transform out, column transform in / out,
// channel axis in / out, inverse columns in / out, rows in: 34 B / element, 220 us for 64 such units.  The three middle passes touch the
// same (channel, row) lines of a block of T neighbouring frequency columns: with N1 = 2 or 4 channels those N1 T lines of d2 points fit one
// tile (N1 T d2 <= 8192 points), so the block keeps them for the forward column transform, the channel-axis transform + radial weights +
// inverse (registers, outer_column: the very code of colored_outer_axis_regs) and the inverse column transform -- one read and one write of
// the spectrum, 18 B / element over the whole draw.  The arithmetic per line is that of colored_strided_axis / outer_column unchanged.
template <int N1>
__global__ __launch_bounds__(FFT_THREADS) void colored_mid_axes(const ColoredArgs a, int N, int logN, int logT) {
  extern __shared__ float2 smem[];
  const int T = 1 << logT, ld = N + 1, L = N1 * T;
  float2* tw = smem;
  float2* buf = smem + N / 2;
  const int64_t smp = blockIdx.y;
  const int k30 = (int)blockIdx.x * T;
  const int cols_here = (int)a.d3h - k30 < T ? (int)a.d3h - k30 : T;
  float2* base = a.spec + smp * (int64_t)N1 * a.d2 * a.d3h + k30;
  make_twiddles(tw, N);
  // loads: consecutive lanes take consecutive frequency columns (runs of 8 T bytes), a thread walks the (channel, row) pairs FFT_THREADS / T apart
  const int t = threadIdx.x & (T - 1), r0 = threadIdx.x >> logT, dn = FFT_THREADS >> logT;
  const bool live = t < cols_here;
  const int rows = N1 * N;
  for (int i0 = r0; i0 < rows; i0 += 8 * dn) {
    float2 r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * dn;
      r[u] = (live && i < rows) ? base[(int64_t)i * a.d3h + t] : make_float2(0.f, 0.f);  // (row i = c d2 + h: the spectrum's own order)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * dn;
      if (i < rows) buf[((i >> logN) * T + t) * ld + (int)brev(i & (N - 1), logN)] = r[u];
    }
  }
  fft_tile<false>(buf, tw, N, logN, L);
  // the channel axis of every (column, row frequency) of the tile, in registers
  double p1 = 0.0, p2 = 0.0;
  for (int q = threadIdx.x; q < T * N; q += FFT_THREADS) {
    const int tt = q & (T - 1), kh = q >> logT;
    if (tt >= cols_here) continue;
    float2 v[N1];
#pragma unroll
    for (int c = 0; c < N1; ++c) v[c] = buf[(c * T + tt) * ld + kh];
    outer_column<N1>(a, v, (int64_t)kh * a.d3h + k30 + tt, p1, p2);
#pragma unroll
    for (int c = 0; c < N1; ++c) buf[(c * T + tt) * ld + kh] = v[c];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < L * N; q += FFT_THREADS) {  // natural order -> bit-reversed, in place (each pair swapped once)
    const int j = q >> logN, k = q & (N - 1);
    const int r = (int)brev(k, logN);
    if (k < r) { const float2 x = buf[j * ld + k]; buf[j * ld + k] = buf[j * ld + r]; buf[j * ld + r] = x; }
  }
  fft_tile<true>(buf, tw, N, logN, L);
  if (live) {
    for (int i = r0; i < rows; i += dn) base[(int64_t)i * a.d3h + t] = buf[((i >> logN) * T + t) * ld + (i & (N - 1))];
  }
  if (a.n_slots_c > 0) block_sums(p1, p2, a.partials + (int64_t)a.batch * a.n_slots * 2 + (smp * a.n_slots_c + blockIdx.x) * 2);
}

template <int N>
__global__ __launch_bounds__(256) void colored_outer_axis_regs(const ColoredArgs a) {
  const int64_t cols = (int64_t)a.d2 * a.d3h;  // columns per sample; element n of column q sits at q + n*cols
  const int64_t smp = blockIdx.y;
  float2* base = a.spec + smp * (int64_t)N * cols;
  double p1 = 0.0, p2 = 0.0;  // Parseval: sum x = V[0,0,0], sum x^2 = (1/N_total) sum mult * |V|^2 (the caller divides)
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < cols; q += (int64_t)gridDim.x * 256) {
    float2 v[N];
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = base[q + (int64_t)n * cols];
    outer_column<N>(a, v, q, p1, p2);
#pragma unroll
    for (int n = 0; n < N; ++n) base[q + (int64_t)n * cols] = v[n];
  }
  if (a.n_slots_c <= 0) return;
  double* slot = a.partials + (int64_t)a.batch * a.n_slots * 2 + (smp * a.n_slots_c + blockIdx.x) * 2;
  if (a.done_count == nullptr) { block_sums(p1, p2, slot); return; }
  // The sample's rescale factor, by whichever block of the sample stores its Parseval slot last (round 5: this was a launch of its own in
  // front of the persistent inverse kernel, 5 us).  Hand-over as guides/MI355X_MICROARCH.md prescribes for small records: the slot goes out
  // with agent-scope (sc1, write-through) stores by ONE lane, that lane drains them (vmcnt(0)) and then adds to the sample's counter; the block
  // whose add returns gridDim.x - 1 reads every slot with agent-scope loads.  The summation order is fixed (lane l takes slots l, l + 64, ...;
  // shuffle tree), so the factor does not depend on which block comes last.
  block_sums<true>(p1, p2, slot);
  __shared__ uint32_t last_sh;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last_sh = __hip_atomic_fetch_add(a.done_count + smp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (last_sh == 0u || threadIdx.x >= 64) return;
  double fa[4] = {0.0, 0.0, 0.0, 0.0};
  const double* pw = a.partials + smp * a.n_slots * 2;  // (the forward kernel's: a launch ago)
  for (int sl = threadIdx.x; sl < a.n_slots; sl += 64) { fa[0] += pw[2 * sl]; fa[1] += pw[2 * sl + 1]; }
  const double* pc = a.partials + (int64_t)a.batch * a.n_slots * 2 + smp * a.n_slots_c * 2;
  for (int sl = threadIdx.x; sl < a.n_slots_c; sl += 64) { fa[2] += gload<true>(pc + 2 * sl); fa[3] += gload<true>(pc + 2 * sl + 1); }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int o = 32; o > 0; o >>= 1) fa[i] += __shfl_down(fa[i], o);
  if (threadIdx.x == 0) {
    const double n = (double)a.d1 * (double)a.d2 * (double)a.d3;
    a.factors_out[smp] = rescale_factor(fa[0], fa[1], fa[2], fa[3] / n, n, a.has_energy, a.energy);
  }
}


// ---- pass F: rescale per sample ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colored_finish(T* out, const ColoredArgs a, int64_t unit, int has_energy, double energy) {
  const int64_t smp = blockIdx.y;
  double w1 = 0, w2 = 0, c1 = 0, c2 = 0;
  for (int s = 0; s < a.n_slots; ++s) {
    const double* pw = a.partials + ((0 * a.batch + smp) * a.n_slots + s) * 2;
    const double* pc = a.partials + ((1 * a.batch + smp) * a.n_slots + s) * 2;
    w1 += pw[0]; w2 += pw[1]; c1 += pc[0]; c2 += pc[1];
  }
  const double n = (double)unit;
  const double wstd = sqrt((w2 - w1 * w1 / n) / (n - 1.0));
  const double cstd = sqrt((c2 - c1 * c1 / n) / (n - 1.0));
  float factor = 1.0f;  // reference: only rescale when the coloured std is not degenerate (noise.py:401-403)
  if ((float)cstd > 1e-8f) factor = has_energy ? (float)energy / (float)cstd : (float)wstd / (float)cstd;
  const float* src = a.real_out + smp * unit;
  T* dst = out + smp * unit;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < unit; i += (int64_t)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    store4_from_f32<T>(dst + i, v.x * factor, v.y * factor, v.z * factor, v.w * factor);
  }
}

}  // namespace skr

// points per block tile; default 4096 (32 KiB of LDS).  SKR_FFT_TILE overrides it for tuning runs.
static int fft_tile_points() {
  static int cached = 0;
  if (!cached) {
    const char* e = getenv("SKR_FFT_TILE");
    int v = e ? atoi(e) : 0;
    cached = (v >= 256 && v <= 16384 && !(v & (v - 1))) ? v : skr::FFT_MAX_TILE;
  }
  return cached;
}

static int ilog2_exact(int64_t v) {
  if (v < 2 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1ll << l) < v) ++l;
  return l;
}

#define SKR_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH; } while (0)
// kernels that need more than the default 48 KiB of dynamic LDS must opt in
#define SKR_ALLOW_LDS(kernel, bytes) do { if ((bytes) > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) return SKR_ERR_UNSUPPORTED; } while (0)

namespace skr {

// persistent inverse kernel of 128 x 128 planes (colored_inverse128); -1: not taken (the caller launches colored_plane<1, ...>)
// grid of the persistent inverse kernel (two resident blocks per CU), 0 when the kernel does not apply
static int64_t inverse128_blocks(int32_t d2, int32_t d3, int32_t out_dtype, int64_t n_planes) {
  static const bool off = getenv("SKR_COLORED_OLD_INVERSE") != nullptr;
  if (off || d2 != 128 || d3 != 128 || n_planes < 1 || (out_dtype != SKR_BF16 && out_dtype != SKR_F16 && out_dtype != SKR_F32)) return 0;
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
    cus[dev] = n;
  }
  static const int per_cu = [] { const char* e = getenv("SKR_COLORED_INV_BLOCKS"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 8 ? v : 2; }();
  const int64_t blocks = (int64_t)per_cu * cus[dev];
  return blocks > n_planes ? n_planes : blocks;
}

static int launch_inverse128(ColoredArgs a, int32_t out_dtype, int64_t n_planes, float* factors /* [batch] workspace, unused when a.raw */, hipStream_t s) {
  if (a.d2 != 128 || a.d3 != 128 || n_planes < 1 || (!a.raw && factors == nullptr) || (out_dtype != SKR_BF16 && out_dtype != SKR_F16 && out_dtype != SKR_F32)) return -1;
  const int64_t blocks = inverse128_blocks(a.d2, a.d3, out_dtype, n_planes);
  if (blocks < 1) return -1;
  const size_t lds = sizeof(float2) * (64 * 129 + 128);
  a.factors = factors;
#ifdef SKR_COLORED_TRACE
  if (a.trace) a.trace += 65536;  // behind the forward kernel's stamps (4096 blocks x 16 words)
#endif
  if (!a.raw && a.factors_out == nullptr) hipLaunchKernelGGL(colored_factors, dim3((unsigned)a.batch), dim3(64), 0, s, a, factors, a.first_ticket);  // (not fused: SKR_COLORED_FACTORS_KERNEL)
#define SKR_INV128(T) do { SKR_ALLOW_LDS((colored_inverse128<T>), lds); hipLaunchKernelGGL((colored_inverse128<T>), dim3((unsigned)blocks), dim3(512), lds, s, a, n_planes); } while (0)
  if (out_dtype == SKR_BF16) SKR_INV128(__bf16);
  else if (out_dtype == SKR_F16) SKR_INV128(_Float16);
  else SKR_INV128(float);
#undef SKR_INV128
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

static bool mixed_factor_axis(int d, MixedAxis& x) {
  int v = d;
  while (v % 2 == 0) v /= 2;
  const int r = v;  // the odd part: one direct r-point combining pass
  v = d / r;
  if (r > 63 || v < 2 || d > 4096) return false;  // (d2 even and d3 % 4 == 0 are the plane kernel's own conditions)
  int lg = 0;
  while ((1 << lg) < v) ++lg;
  x.n = d; x.r = r; x.a = lg; x.m = v; x.pitch = r * (v + 1);
  x.magic_r = (uint32_t)((0x100000000ull + (uint32_t)r - 1) / (uint32_t)r);
  return true;
}

// geometry, LDS bytes and block size of colored_plane_mixed for a d2 x d3 plane (two_d: the plane is a whole 2-D unit, whose
// half plane is staged in registers once more); false if the plane is not covered
static bool mixed_plane_geometry(int d2, int d3, bool two_d, MixedGeom& mg, size_t& lds, int& threads) {
  if (d2 % 2 != 0 || d3 % 4 != 0 || !mixed_factor_axis(d2, mg.h) || !mixed_factor_axis(d3, mg.w)) return false;
  const int64_t pairs = d2 / 2, wh = d3 / 2 + 1;
  const size_t tile = (size_t)(pairs * mg.w.pitch > wh * mg.h.pitch ? pairs * mg.w.pitch : wh * mg.h.pitch);
  lds = sizeof(float2) * ((size_t)d3 + d2 + mg.w.m / 2 + mg.h.m / 2 + tile);
  mg.magic_wh = (uint32_t)((0x100000000ull + (uint32_t)wh - 1) / (uint32_t)wh);
  // 512-thread blocks while every register-staged transpose fits 18 items per thread, else 1024 (3-D units only need the
  // row-pair transposes to fit; 2-D units also the full half plane)
  auto fits = [&](int t) {
    return pairs * d3 <= (int64_t)t * PLANE_ITEMS && pairs * wh <= (int64_t)t * PLANE_ITEMS && (int64_t)d2 * wh <= 2ll * t * PLANE_ITEMS &&
           (!two_d || (int64_t)d2 * wh <= (int64_t)t * PLANE_ITEMS);
  };
  threads = fits(512) ? 512 : (fits(1024) ? 1024 : 0);
  return lds <= 156 * 1024 && threads != 0;
}

// Independent d2 x d3 planes on the LDS plane kernels, for skr_colored_any.hip (3-D units whose leading axis is a direct DFT there:
// not a power of two <= 16).  mode 0: plane p of sample b is drawn
// (elements p * d2 * d3 ... of the sample's Philox stream) and transformed, half spectrum to spec[b][p][d2][d3/2+1], its (sum,
// sum of squares) to plane_partials[b][p][2].  mode 1: the plain inverse of every plane (no 1/N: hipFFT's C2R convention), fp32,
// to real_out[b][p][d2][d3].  SKR_ERR_UNSUPPORTED for a plane shape the kernels do not cover.
int colored_planes(int mode, float2* spec, double* plane_partials, float* real_out, const uint64_t* seeds, uint64_t stream_id,
                   int64_t batch, int64_t planes, int32_t d2, int32_t d3, hipStream_t s) {
  if (batch <= 0 || planes <= 0 || batch > 65535 || planes > 0x7fffffffll || d2 < 2 || d3 < 4 || getenv("SKR_FFT_NO_PLANES") != nullptr) return SKR_ERR_UNSUPPORTED;
  ColoredArgs a;
  a.spec = spec; a.real_out = nullptr; a.partials = plane_partials; a.seeds = seeds; a.stream = stream_id;
  a.batch = batch; a.d1 = (int32_t)planes; a.d2 = d2; a.d3 = d3; a.d3h = d3 / 2 + 1;
  a.n_slots = (int32_t)planes; a.n_slots_c = 0; a.has_energy = 0; a.energy = 0.0; a.out = real_out;
  a.exponent_half_neg = 0.f; a.eps_clip = 1.f; a.inv_rmax = 1.f; a.raw = 1; a.factors = nullptr; a.ticket = nullptr; a.first_ticket = 0; a.done_count = nullptr; a.factors_out = nullptr;
#ifdef SKR_COLORED_TRACE
  a.trace = nullptr;
#endif
  const dim3 grid((unsigned)planes, (unsigned)batch);
  const int l3 = ilog2_exact(d3), l2 = ilog2_exact(d2);
  const int64_t d3h = a.d3h;
  const size_t tile_points = (size_t)(d2 / 2) * (d3 + 1) > (size_t)d3h * (d2 + 1) ? (size_t)(d2 / 2) * (d3 + 1) : (size_t)d3h * (d2 + 1);
  const size_t lds_plane = sizeof(float2) * ((size_t)d3 + d2 + tile_points);  // twiddles of both axes (full circle) + the tile
  const bool pow2_plane = l3 >= 2 && l2 >= 1 && lds_plane <= 150 * 1024 && (int64_t)(d2 / 2) * d3 <= PLANE_THREADS * PLANE_ITEMS &&
                          (int64_t)(d2 / 2) * d3h <= PLANE_THREADS * PLANE_ITEMS && (int64_t)d2 * d3h <= 2ll * PLANE_THREADS * PLANE_ITEMS;
  if (pow2_plane) {
#define SKR_PLANES_T(MODE, CH, CW) do { SKR_ALLOW_LDS((colored_plane<MODE, float, CH, CW>), lds_plane); hipLaunchKernelGGL((colored_plane<MODE, float, CH, CW>), grid, dim3(PLANE_THREADS), lds_plane, s, a, l2, l3); } while (0)
#define SKR_PLANES(MODE)                                \
    if (l2 == 7 && l3 == 7) SKR_PLANES_T(MODE, 7, 7);   \
    else if (l2 == 6 && l3 == 6) SKR_PLANES_T(MODE, 6, 6); \
    else SKR_PLANES_T(MODE, 0, 0)
    if (mode == 0) { SKR_PLANES(0); }
    else {
      const int rc = launch_inverse128(a, SKR_F32, batch * planes, nullptr, s);
      if (rc >= 0) return rc;
      SKR_PLANES(1);
    }
#undef SKR_PLANES
#undef SKR_PLANES_T
    SKR_CHECK_LAUNCH();
    return SKR_OK;
  }
  MixedGeom mg;
  size_t lds_mixed = 0;
  int threads = 0;
  // (measurement switch.  Rounds 3-4 stopped at odd parts summing to 10: beyond that hipFFT's 3-D plan was faster than planes here + a direct
  //  outer-axis pass.  The N-D transform is the library's own since (skr_fft_own.hip), and every plane this kernel can hold beats it:
  //  5 x 60 x 104 100 against 196 us, 12 x 168 x 96 115 against 245, 16 x 13 x 60 x 104 322 against 542.)
  static const int odd_limit = [] { const char* e = getenv("SKR_FFT_ODD_LIMIT"); return e ? atoi(e) : 1 << 20; }();
  if (getenv("SKR_FFT_NO_MIXED") != nullptr || !mixed_plane_geometry(d2, d3, false, mg, lds_mixed, threads) || mg.h.r + mg.w.r > odd_limit) return SKR_ERR_UNSUPPORTED;
#define SKR_PLANES_M(MODE) do {                                                                                                                  \
    if (threads == 512) { SKR_ALLOW_LDS((colored_plane_mixed<MODE, float, 512>), lds_mixed); hipLaunchKernelGGL((colored_plane_mixed<MODE, float, 512>), grid, dim3(512), lds_mixed, s, a, mg); } \
    else { SKR_ALLOW_LDS((colored_plane_mixed<MODE, float, 1024>), lds_mixed); hipLaunchKernelGGL((colored_plane_mixed<MODE, float, 1024>), grid, dim3(1024), lds_mixed, s, a, mg); }              \
  } while (0)
  if (mode == 0) SKR_PLANES_M(0); else SKR_PLANES_M(1);
#undef SKR_PLANES_M
  SKR_CHECK_LAUNCH();
  return SKR_OK;
}

}  // namespace skr

static int colored_batch(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64, int64_t partial_slots,
                         const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t d1, int32_t d2, int32_t d3,
                         double exponent, int32_t has_energy, double energy, void* stream);

extern "C" int skr_noise_colored(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64, int64_t partial_slots,
                                 const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t d1, int32_t d2, int32_t d3,
                                 double exponent, int32_t has_energy, double energy, void* stream) {
  skr::DeviceGuard device_guard(out);
  if (batch < 0 || d1 < 1 || d2 < 2 || d3 < 4) return SKR_ERR_SHAPE;
  if (batch == 0) return SKR_OK;
  if (!out || !spec_c64 || !scratch_f32 || !partials_f64 || !seeds_dev) return SKR_ERR_NULL;
  // Samples are independent, and the half spectrum of a batch is written once and read three times (forward planes -> channel axis ->
  // inverse planes).  Groups of samples whose spectrum fits the 256 MB Infinity Cache go through the three kernels one after the other,
  // over the SAME spectrum buffer, so that those re-reads are served on the chip instead of from HBM.
  static const int64_t group_mb = [] { const char* e = getenv("SKR_COLORED_GROUP_MB"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
  const int64_t per_sample = (int64_t)d1 * d2 * (d3 / 2 + 1) * (int64_t)sizeof(float2);
  int64_t group = batch;
  if (group_mb > 0 && d1 > 1 && per_sample * batch > group_mb * (1ll << 20)) {
    group = group_mb * (1ll << 20) / per_sample;
    if (group < 1) group = 1;
    const int64_t n_groups = (batch + group - 1) / group;
    group = (batch + n_groups - 1) / n_groups;  // even groups
  }
  const size_t esz = out_dtype == SKR_F64 ? 8 : (out_dtype == SKR_F32 ? 4 : 2);
  const int64_t unit = (int64_t)d1 * d2 * d3;
  for (int64_t s0 = 0; s0 < batch; s0 += group) {
    const int64_t n = batch - s0 < group ? batch - s0 : group;
    const int rc = colored_batch(static_cast<char*>(out) + (size_t)(s0 * unit) * esz, out_dtype, spec_c64, scratch_f32, partials_f64 + s0 * 4 * partial_slots, partial_slots,
                                 seeds_dev + s0, stream_id, n, d1, d2, d3, exponent, has_energy, energy, stream);
    if (rc != SKR_OK) return rc;
  }
  return SKR_OK;
}

static int colored_batch(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64, int64_t partial_slots,
                         const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t d1, int32_t d2, int32_t d3,
                         double exponent, int32_t has_energy, double energy, void* stream) {
  using namespace skr;
  const int l3 = ilog2_exact(d3), l2 = ilog2_exact(d2), l1 = d1 == 1 ? 0 : ilog2_exact(d1);
  const bool pow2 = l3 >= 2 && l2 >= 1 && l1 >= 0 && d3 <= FFT_MAX_TILE && d2 <= FFT_MAX_TILE && d1 <= FFT_MAX_TILE;
  // planes whose sides are a power of two times an odd factor up to 63 (96, 112, 144, 160, 192 ...) under a power-of-two channel axis: colored_plane_mixed
  MixedGeom mg;
  bool mixed = false;
  size_t lds_mixed = 0;
  int mixed_threads = 0;
  if (!pow2 && l1 >= 0 && d1 <= 16 && getenv("SKR_FFT_NO_MIXED") == nullptr)
    mixed = mixed_plane_geometry(d2, d3, d1 == 1, mg, lds_mixed, mixed_threads) && d1 <= partial_slots;
  if (!pow2 && !mixed) return SKR_ERR_UNSUPPORTED;  // (the caller takes skr_noise_colored_any: hipFFT)
  if (batch > 65535) return SKR_ERR_UNSUPPORTED;
  ColoredArgs a;
  a.spec = reinterpret_cast<float2*>(spec_c64); a.real_out = scratch_f32; a.partials = partials_f64; a.seeds = seeds_dev;
  a.stream = stream_id; a.batch = batch; a.d1 = d1; a.d2 = d2; a.d3 = d3; a.d3h = d3 / 2 + 1;
  a.exponent_half_neg = (float)(-exponent / 2.0);
  a.out = out; a.has_energy = has_energy; a.energy = energy; a.n_slots_c = 0; a.raw = 0; a.factors = nullptr; a.ticket = nullptr; a.first_ticket = 0; a.done_count = nullptr; a.factors_out = nullptr;
#ifdef SKR_COLORED_TRACE
  a.trace = g_colored_trace;
#endif
  const int nd = d1 > 1 ? 3 : 2;
  const double n_eff = nd == 3 ? ((double)d1 + d2 + d3) / 3.0 : ((double)d2 + d3) / 2.0;
  a.eps_clip = (float)(0.5 / (n_eff > 4.0 ? n_eff : 4.0));
  // r_max over the rfftn grid: every axis reaches floor(d/2)/d
  auto fmaxf_axis = [](int d) { return (float)(d / 2) / (float)d; };
  const float m1 = d1 > 1 ? fmaxf_axis(d1) : 0.f, m2 = fmaxf_axis(d2), m3 = fmaxf_axis(d3);
  const float rmax = sqrtf(m1 * m1 + m2 * m2 + m3 * m3);
  a.inv_rmax = rmax > 0.f ? 1.0f / rmax : 1.0f;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  const int64_t d3h = a.d3h;
  // fused plane kernels when one d2 x d3 plane (+ its half spectrum) fits the CU's LDS
  const size_t tile_points = (size_t)(d2 / 2) * (d3 + 1) > (size_t)d3h * (d2 + 1) ? (size_t)(d2 / 2) * (d3 + 1) : (size_t)d3h * (d2 + 1);
  const size_t lds_plane = sizeof(float2) * ((size_t)d3 + d2 + tile_points);  // twiddles of both axes (full circle) + the tile
  const bool fused = lds_plane <= 150 * 1024 && (int64_t)(d2 / 2) * d3 <= PLANE_THREADS * PLANE_ITEMS && (int64_t)(d2 / 2) * d3h <= PLANE_THREADS * PLANE_ITEMS &&
                     (nd == 2 || d1 <= 16) && d3 % 4 == 0 && getenv("SKR_FFT_NO_FUSE") == nullptr;
  auto outer_axis = [&]() -> int {
    // axis 1 (length d1, stride d2*d3h): forward, radial weights, inverse in one pass
    if (d1 <= 16) {
      const int64_t cols = (int64_t)d2 * d3h;
      int64_t bx = (cols + 255) / 256; if (bx > 4096) bx = 4096;
      if (a.n_slots_c > 0) { if (bx > a.n_slots_c) bx = a.n_slots_c; a.n_slots_c = (int32_t)bx; }  // one Parseval slot per block
      dim3 grid((unsigned)bx, (unsigned)batch);
      switch (d1) {
        case 2: hipLaunchKernelGGL(colored_outer_axis_regs<2>, grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(colored_outer_axis_regs<4>, grid, dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL(colored_outer_axis_regs<8>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(colored_outer_axis_regs<16>, grid, dim3(256), 0, s, a); break;
      }
      return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
    }
    return -1;  // caller uses the LDS tile kernel
  };

  if (mixed) {
    a.n_slots = d1;
    dim3 grid((unsigned)d1, (unsigned)batch);
    const int const_side = (d2 == d3 && getenv("SKR_FFT_NO_CONST_SIDE") == nullptr) ? d2 : 0;  // 96 / 160 / 192: compile-time geometry
#define SKR_MIXED_SIDE(MODE, T, THREADS, SIDE) do { SKR_ALLOW_LDS((colored_plane_mixed<MODE, T, THREADS, SIDE>), lds_mixed); hipLaunchKernelGGL((colored_plane_mixed<MODE, T, THREADS, SIDE>), grid, dim3(THREADS), lds_mixed, s, a, mg); } while (0)
#define SKR_MIXED_T(MODE, T) do {                                                                                                              \
      if (const_side == 96 && mixed_threads == 512 && sizeof(T) <= 4) SKR_MIXED_SIDE(MODE, T, 512, 96);                                          \
      else if (const_side == 160 && mixed_threads == 1024 && sizeof(T) <= 4) SKR_MIXED_SIDE(MODE, T, 1024, 160);                                 \
      else if (const_side == 192 && mixed_threads == 1024 && sizeof(T) <= 4 && MODE != 2) SKR_MIXED_SIDE((MODE == 2 ? 1 : MODE), T, 1024, 192);  \
      else if (mixed_threads == 512) { SKR_ALLOW_LDS((colored_plane_mixed<MODE, T, 512>), lds_mixed); hipLaunchKernelGGL((colored_plane_mixed<MODE, T, 512>), grid, dim3(512), lds_mixed, s, a, mg); } \
      else { SKR_ALLOW_LDS((colored_plane_mixed<MODE, T, 1024>), lds_mixed); hipLaunchKernelGGL((colored_plane_mixed<MODE, T, 1024>), grid, dim3(1024), lds_mixed, s, a, mg); }              \
    } while (0)
#define SKR_MIXED(MODE)                                             \
    switch (out_dtype) {                                            \
      case SKR_BF16: SKR_MIXED_T(MODE, __bf16); break;              \
      case SKR_F16: SKR_MIXED_T(MODE, _Float16); break;             \
      case SKR_F32: SKR_MIXED_T(MODE, float); break;                \
      case SKR_F64: SKR_MIXED_T(MODE, double); break;               \
      default: return SKR_ERR_DTYPE;                                \
    }                                                               \
    SKR_CHECK_LAUNCH()
    if (nd == 2) {
      SKR_MIXED(2);
    } else {
      SKR_MIXED_T(0, float);
      SKR_CHECK_LAUNCH();
      a.n_slots_c = (int32_t)(2 * partial_slots - d1 < 4096 ? 2 * partial_slots - d1 : 4096);
      const int rc = outer_axis();
      if (rc != SKR_OK) return rc;
      SKR_MIXED(1);
    }
#undef SKR_MIXED
#undef SKR_MIXED_T
#undef SKR_MIXED_SIDE
    return SKR_OK;
  }
  if (fused) {
    if (d1 > partial_slots) return SKR_ERR_SHAPE;
    a.n_slots = d1;
    dim3 grid((unsigned)d1, (unsigned)batch);
#define SKR_PLANE_T(MODE, T, CH, CW) do { SKR_ALLOW_LDS((colored_plane<MODE, T, CH, CW>), lds_plane); hipLaunchKernelGGL((colored_plane<MODE, T, CH, CW>), grid, dim3(PLANE_THREADS), lds_plane, s, a, l2, l3); } while (0)
#define SKR_PLANE_SZ(MODE, T)                                                    \
    if (l2 == 7 && l3 == 7) SKR_PLANE_T(MODE, T, 7, 7);                          \
    else if (l2 == 6 && l3 == 6) SKR_PLANE_T(MODE, T, 6, 6);                     \
    else SKR_PLANE_T(MODE, T, 0, 0)
#define SKR_PLANE(MODE)                                                          \
    switch (out_dtype) {                                                         \
      case SKR_BF16: SKR_PLANE_SZ(MODE, __bf16); break;                          \
      case SKR_F16: SKR_PLANE_SZ(MODE, _Float16); break;                         \
      case SKR_F32: SKR_PLANE_SZ(MODE, float); break;                            \
      case SKR_F64: SKR_PLANE_T(MODE, double, 0, 0); break;                      \
      default: return SKR_ERR_DTYPE;                                             \
    }                                                                            \
    SKR_CHECK_LAUNCH()
    if (nd == 2) {
      SKR_PLANE(2);
    } else {
      // room left in the partials buffer, less one slot per sample: the last 2 * batch doubles (4 floats per sample) are the bookkeeping of
      // colored_inverse128 -- [batch] rescale factors, the plane ticket, [batch] arrival counters of the outer-axis kernel
      const int32_t slots_c = (int32_t)(2 * partial_slots - d1 - 1 < 4096 ? 2 * partial_slots - d1 - 1 : 4096);
      if (slots_c < 1) return SKR_ERR_SHAPE;
      float* tail = reinterpret_cast<float*>(partials_f64 + 4 * batch * partial_slots - 2 * batch);
      const int64_t inv_blocks = inverse128_blocks(d2, d3, out_dtype, batch * (int64_t)d1);
      if (inv_blocks > 0) {
        static const bool static_deal = getenv("SKR_COLORED_INV_STATIC") != nullptr, factors_kernel = getenv("SKR_COLORED_FACTORS_KERNEL") != nullptr;
        a.first_ticket = (uint32_t)(2 * inv_blocks);
        a.ticket = static_deal ? nullptr : reinterpret_cast<uint32_t*>(tail + batch);
        if (!factors_kernel) { a.done_count = reinterpret_cast<uint32_t*>(tail + 2 * batch); a.factors_out = tail; }
      }
      SKR_PLANE_SZ(0, float);
      SKR_CHECK_LAUNCH();
      a.n_slots_c = slots_c;
      const int rc = outer_axis();  // d1 <= 16: register kernel, with the Parseval partials of the weighted spectrum
      if (rc != SKR_OK) return rc;
      const int inv = launch_inverse128(a, out_dtype, batch * (int64_t)d1, tail, s);  // 128 x 128 planes: the persistent kernel
      if (inv >= 0) return inv;
      SKR_PLANE(1);
    }
#undef SKR_PLANE
#undef SKR_PLANE_SZ
#undef SKR_PLANE_T
    return SKR_OK;  // the plane kernels wrote `out` themselves
  } else {
  // pass A: last axis forward
  const int64_t lines_last = (int64_t)d1 * d2;
  const int tile = fft_tile_points();
  int La = 2 * (tile / d3); if (La > lines_last) La = (int)lines_last; if (La < 2) La = 2;  // real lines per tile (pairs share a transform)
  const int64_t blocks_a = (lines_last + La - 1) / La;
  if (blocks_a > partial_slots) return SKR_ERR_SHAPE;
  a.n_slots = (int32_t)blocks_a;
  const size_t lds_a = sizeof(float2) * ((size_t)d3 / 2 + (size_t)(La / 2) * (d3 + 1));
  if (lds_a > 150 * 1024) return SKR_ERR_UNSUPPORTED;
  SKR_ALLOW_LDS(colored_last_axis<true>, lds_a);
  SKR_ALLOW_LDS(colored_last_axis<false>, lds_a);
  hipLaunchKernelGGL(colored_last_axis<true>, dim3((unsigned)blocks_a, (unsigned)batch), dim3(FFT_THREADS), lds_a, s, a, l3, La);
  SKR_CHECK_LAUNCH();

  auto strided = [&](int mode, int N, int logN, int64_t n_lines, int64_t inner, int64_t outer, int64_t stride, int axis) -> int {
    int logL = 0;
    while ((2 << logL) * N <= tile && (2 << logL) <= 64) ++logL;  // L = 2^logL lines per tile, <= 64 (512 B runs)
    const int L = 1 << logL;
    const int64_t blocks = (n_lines + L - 1) / L;
    const size_t lds = sizeof(float2) * ((size_t)N / 2 + (size_t)L * (N + 1));
    dim3 grid((unsigned)blocks, (unsigned)batch);
    if (lds > 150 * 1024) return SKR_ERR_UNSUPPORTED;
    SKR_ALLOW_LDS(colored_strided_axis<0>, lds);
    SKR_ALLOW_LDS(colored_strided_axis<1>, lds);
    SKR_ALLOW_LDS(colored_strided_axis<2>, lds);
    if (mode == 0) hipLaunchKernelGGL(colored_strided_axis<0>, grid, dim3(FFT_THREADS), lds, s, a, N, logN, logL, n_lines, inner, outer, stride, axis);
    else if (mode == 1) hipLaunchKernelGGL(colored_strided_axis<1>, grid, dim3(FFT_THREADS), lds, s, a, N, logN, logL, n_lines, inner, outer, stride, axis);
    else hipLaunchKernelGGL(colored_strided_axis<2>, grid, dim3(FFT_THREADS), lds, s, a, N, logN, logL, n_lines, inner, outer, stride, axis);
    return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
  };
  int rc;
  // 3-D units with a short outer axis: Parseval partials from the outer-axis kernel, so that pass E can write the result itself
  const bool direct_out = nd == 3 && d1 <= 16 && d3 % 4 == 0 && 2 * partial_slots - blocks_a >= 1;
  // a short channel axis: columns + channels + inverse columns in one tile residency (colored_mid_axes).  Tile width T: N1 T d2 <= 8192 points
  // (66 KB: two blocks per CU; a 132 KB tile left a CU one block of four waves and ran 20 % SLOWER than the three separate passes), halved
  // while the launch has fewer than six blocks per CU; measured on 256 x 256 planes (us per draw, separate passes -> fused): 4 channels x 64
  // 215 -> 188, x 256 755 -> 616, 2 channels x 128 205 -> 159; 8 channels stay on the separate passes (186 vs 211 at T = 4).
  int mid_logT = -1;
  if (nd == 3 && direct_out && (d1 == 2 || d1 == 4) && getenv("SKR_COLORED_NO_MID") == nullptr) {
    int logT = 0;
    while ((int64_t)d1 * (2 << logT) * d2 <= 8192) ++logT;
    while (logT > 2 && ((d3h + (1 << logT) - 1) >> logT) * batch < 6 * 256) --logT;
    if (const char* e = getenv("SKR_COLORED_MID_LOGT")) { const int v = atoi(e); if (v >= 2 && (int64_t)d1 * (1 << v) * d2 <= 16384 && (d1 << v) <= 64) logT = v; }
    const int64_t tiles = (d3h + (1 << logT) - 1) >> logT;
    if (logT >= 2 && tiles <= 2 * partial_slots - blocks_a && (int64_t)d1 * (1 << logT) * d2 <= 16384) mid_logT = logT;
  }
  if (mid_logT >= 0) {
    const int T = 1 << mid_logT;
    const int64_t tiles = (d3h + T - 1) / T;
    a.n_slots_c = (int32_t)tiles;
    const size_t lds = sizeof(float2) * ((size_t)d2 / 2 + (size_t)d1 * T * (d2 + 1));
    dim3 grid((unsigned)tiles, (unsigned)batch);
    switch (d1) {
      case 2: SKR_ALLOW_LDS(colored_mid_axes<2>, lds); hipLaunchKernelGGL(colored_mid_axes<2>, grid, dim3(FFT_THREADS), lds, s, a, d2, l2, mid_logT); break;
      case 4: SKR_ALLOW_LDS(colored_mid_axes<4>, lds); hipLaunchKernelGGL(colored_mid_axes<4>, grid, dim3(FFT_THREADS), lds, s, a, d2, l2, mid_logT); break;
      default: return SKR_ERR_SHAPE;
    }
    SKR_CHECK_LAUNCH();
  } else if (nd == 3) {
    // axis 2 (length d2, stride d3h): lines = (i1, k3)
    if ((rc = strided(0, d2, l2, (int64_t)d1 * d3h, d3h, (int64_t)d2 * d3h, d3h, 2)) != SKR_OK) return rc;
    if (direct_out) a.n_slots_c = (int32_t)(2 * partial_slots - blocks_a < 4096 ? 2 * partial_slots - blocks_a : 4096);
    rc = outer_axis();
    if (rc == -1) rc = strided(2, d1, l1, (int64_t)d2 * d3h, (int64_t)d2 * d3h, 0, (int64_t)d2 * d3h, 1);
    if (rc != SKR_OK) return rc;
    if ((rc = strided(1, d2, l2, (int64_t)d1 * d3h, d3h, (int64_t)d2 * d3h, d3h, 2)) != SKR_OK) return rc;
  } else {
    if ((rc = strided(2, d2, l2, d3h, d3h, 0, d3h, 2)) != SKR_OK) return rc;
  }

  if (direct_out) {  // pass E writes the result dtype; no scratch, no pass F
    dim3 grid_e((unsigned)blocks_a, (unsigned)batch);
    switch (out_dtype) {
#define SKR_E(T) SKR_ALLOW_LDS(colored_last_axis_out<T>, lds_a); hipLaunchKernelGGL(colored_last_axis_out<T>, grid_e, dim3(FFT_THREADS), lds_a, s, a, l3, La)
      case SKR_BF16: SKR_E(__bf16); break;
      case SKR_F16: SKR_E(_Float16); break;
      case SKR_F32: SKR_E(float); break;
      case SKR_F64: SKR_E(double); break;
#undef SKR_E
      default: return SKR_ERR_DTYPE;
    }
    SKR_CHECK_LAUNCH();
    return SKR_OK;
  }
  // pass E: last axis inverse -> real scratch
  hipLaunchKernelGGL(colored_last_axis<false>, dim3((unsigned)blocks_a, (unsigned)batch), dim3(FFT_THREADS), lds_a, s, a, l3, La);
  SKR_CHECK_LAUNCH();
  }

  // pass F
  const int64_t unit = (int64_t)d1 * d2 * d3;
  int64_t bx = (unit / 4 + 255) / 256; if (bx > 64) bx = 64;
  dim3 grid((unsigned)bx, (unsigned)batch);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(colored_finish<__bf16>, grid, dim3(256), 0, s, (__bf16*)out, a, unit, has_energy, energy); break;
    case SKR_F16: hipLaunchKernelGGL(colored_finish<_Float16>, grid, dim3(256), 0, s, (_Float16*)out, a, unit, has_energy, energy); break;
    case SKR_F32: hipLaunchKernelGGL(colored_finish<float>, grid, dim3(256), 0, s, (float*)out, a, unit, has_energy, energy); break;
    case SKR_F64: hipLaunchKernelGGL(colored_finish<double>, grid, dim3(256), 0, s, (double*)out, a, unit, has_energy, energy); break;
    default: return SKR_ERR_DTYPE;
  }
  SKR_CHECK_LAUNCH();
  return SKR_OK;
}
