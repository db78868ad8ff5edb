// Colored noise for per-sample shapes that are NOT powers of two (e.g. 96x96 or 152x104 latents): the same
// pipeline as skr_colored.hip -- Philox white noise, rfftn, radial power-law weights, irfftn, per-sample
// rescale -- with the transforms delegated to hipFFT (a plain library FFT, loaded lazily with dlopen so the
// engine has no link-time dependency on it).  Generation, weighting, statistics and rescaling stay hand-written.
// Plans are cached per (rank, dims, batch); creating one allocates hipFFT's work area (first call per shape only).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <map>
#include <set>
#include <mutex>
#include <tuple>

#include "skr_device.h"
#include "../../include/skrample_hip.h"
#include "skr_philox.h"
#include "skr_dft.h"

namespace skr {
int g_fft_rank = 0;
int g_use_hipfft = -1;  // skr_set_tuning("hipfft"): 1 = the inner axes on hipFFT, 0 = on skr_fft_own.hip, -1 = by the environment (SKR_FFT_HIPFFT set: hipFFT)
// (skr_fft_own.hip) rfftn / irfftn of any axis lengths on the LDS tile transform
int own_prepare(int dev, int n0, int n1, int n2, hipStream_t s);
bool own_length_ok(int n, bool last);
int own_rfftn(int dev, bool inverse, float* real, float2* spec, int64_t entries, int n0, int n1, int n2, hipStream_t s, bool skip_outer);
int own_outer_weighted(int dev, float2* spec, int64_t entries, int n0, int n1, int n2, float inv_rmax, float eps_clip, float exponent_half_neg, hipStream_t s);
}

namespace {

typedef void* hipfftHandle_t;  // hipfftHandle is an opaque pointer
typedef int (*plan_many_fn)(hipfftHandle_t*, int, int*, int*, int, int, int*, int, int, int /*type*/, int);
typedef int (*exec_r2c_fn)(hipfftHandle_t, float*, void*);
typedef int (*exec_c2r_fn)(hipfftHandle_t, void*, float*);
typedef int (*set_stream_fn)(hipfftHandle_t, hipStream_t);
typedef int (*destroy_fn)(hipfftHandle_t);
constexpr int HIPFFT_R2C = 0x2a, HIPFFT_C2R = 0x2c;

struct FftApi {
  plan_many_fn plan_many = nullptr;
  exec_r2c_fn r2c = nullptr;
  exec_c2r_fn c2r = nullptr;
  set_stream_fn set_stream = nullptr;
  destroy_fn destroy = nullptr;
  bool ok = false;
};

FftApi& api() {
  static FftApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    a.plan_many = (plan_many_fn)dlsym(h, "hipfftPlanMany");
    a.r2c = (exec_r2c_fn)dlsym(h, "hipfftExecR2C");
    a.c2r = (exec_c2r_fn)dlsym(h, "hipfftExecC2R");
    a.set_stream = (set_stream_fn)dlsym(h, "hipfftSetStream");
    a.destroy = (destroy_fn)dlsym(h, "hipfftDestroy");
    a.ok = a.plan_many && a.r2c && a.c2r && a.set_stream && a.destroy;
  });
  return a;
}

// One plan pair per (device, stream, shape): hipFFT plans own a work area and carry their stream, so a plan shared between
// devices runs on the wrong one and a plan shared between streams races on hipfftSetStream.  The cache is bounded; the
// least recently used pair is destroyed when it is full.
struct Plans { hipfftHandle_t fwd, inv; uint64_t last_use; };
typedef std::tuple<int /*device*/, hipStream_t, int, int, int, int, int64_t> PlanKey;
std::mutex g_mutex;
std::map<PlanKey, Plans> g_plans;
std::set<PlanKey> g_bad_plans;  // plans whose self-check failed
uint64_t g_plan_clock = 0;
constexpr size_t MAX_PLANS = 32;
std::atomic<int64_t> g_hipfft_plans{0}, g_hipfft_execs{0}, g_own_execs{0};  // skr_stat("hipfft_plans" / "hipfft_execs" / "own_fft_execs"): plan pairs created, forward transforms run by hipFFT / by skr_fft_own.hip

constexpr int SLOTS = 256;  // partial-sum slots per sample (one per block of the stats kernels)

constexpr int SKR_ANY_MAX_OUTER = 9;  // leading axes on the direct-DFT kernels: units of up to 12 transform axes (reference noise.py:373-403 takes any)

struct AnyArgs {
  float* real;          // [batch][unit]
  float2* spec;         // [batch][d1][d2][d3h]
  double* partials;     // [2][batch][SLOTS][2]
  const uint64_t* seeds;
  uint64_t stream;
  int64_t batch, unit;
  int32_t n_outer;      // axes outside the hipFFT axes (0..5), outermost first; each is a direct DFT (any_outer_axis)
  int32_t outer[SKR_ANY_MAX_OUTER];
  int32_t d1, d2, d3, d3h;
  float exponent_half_neg, eps_clip, inv_rmax;
};

__device__ __forceinline__ void block_sums(double s1, double s2, double* slot) {
  __shared__ double red[2][4];
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) { slot[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]; slot[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]; }
}

// The statistics kernels run with 1 ... SLOTS blocks per sample (stats_blocks: small samples do not pay for 256 nearly empty blocks and
// their reductions -- 78 + 74 us of a 292 us draw at 256 x (4, 30, 90)); any_finish sums all SLOTS slots, so block x zeroes the slots
// x + gridDim.x, x + 2 gridDim.x, ... nobody writes.
__device__ __forceinline__ void clear_unused_slots(double* sample_slots) {
  if (threadIdx.x == 0)
    for (int s = blockIdx.x + gridDim.x; s < SLOTS; s += gridDim.x) { sample_slots[2 * s] = 0.0; sample_slots[2 * s + 1] = 0.0; }
}

// white noise (one lane per Philox block of a sample) + per-block sums; grid = (stats_blocks, batch)
__global__ __launch_bounds__(256) void any_white(const AnyArgs a) {
  const int64_t smp = blockIdx.y;
  const uint64_t seed = a.seeds[smp];
  const int64_t blocks = (a.unit + 3) / 4;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < blocks; b += (int64_t)gridDim.x * 256) {
    float z[4];
    skr::normal4(seed, a.stream, (uint64_t)b, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = b * 4 + j;
      if (e < a.unit) { a.real[smp * a.unit + e] = z[j]; s1 += (double)z[j]; s2 += (double)z[j] * (double)z[j]; }
    }
  }
  block_sums(s1, s2, a.partials + ((0 * a.batch + smp) * SLOTS + blockIdx.x) * 2);
  clear_unused_slots(a.partials + (0 * a.batch + smp) * SLOTS * 2);
}

// the white noise is already in a.real (colorize_noise on a caller's tensor): only its statistics are needed
__global__ __launch_bounds__(256) void any_white_stats(const AnyArgs a) {
  const int64_t smp = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.unit; e += (int64_t)gridDim.x * 256) {
    const float v = a.real[smp * a.unit + e];
    s1 += (double)v; s2 += (double)v * (double)v;
  }
  block_sums(s1, s2, a.partials + ((0 * a.batch + smp) * SLOTS + blockIdx.x) * 2);
  clear_unused_slots(a.partials + (0 * a.batch + smp) * SLOTS * 2);
}

__device__ __forceinline__ float axis_freq(int k, int d) { const int m = k < d - k ? k : d - k; return (float)m / (float)d; }

__global__ __launch_bounds__(256) void any_weights(const AnyArgs a) {
  const int64_t per = (int64_t)a.d1 * a.d2 * a.d3h, total = per * a.batch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i % per;
    const int k3 = (int)(r % a.d3h); r /= a.d3h;
    const int k2 = (int)(r % a.d2);
    const int k1 = (int)(r / a.d2);
    const float f1 = a.d1 > 1 ? axis_freq(k1, a.d1) : 0.f, f2 = a.d2 > 1 ? axis_freq(k2, a.d2) : 0.f, f3 = (float)k3 / (float)a.d3;
    // clamp(radius, eps)^(-exponent/2) on the raw transcendental units (radius >= eps > 0: never denormal), as in skr_colored.hip
    float radius = __builtin_amdgcn_sqrtf(f1 * f1 + f2 * f2 + f3 * f3) * a.inv_rmax;
    radius = radius < a.eps_clip ? a.eps_clip : radius;
    const float w = __builtin_amdgcn_exp2f(a.exponent_half_neg * __builtin_amdgcn_logf(radius));
    float2 v = a.spec[i];
    a.spec[i] = make_float2(v.x * w, v.y * w);
  }
}

// Units with 4 to 6 transform axes: hipFFT takes the three inner axes (batched over batch * prod(outer)); every outer axis
// (length <= 128, e.g. channels or frames) is a direct DFT per spectral line -- one lane per line, its values staged in LDS
// (line-private, no barriers after the twiddle table).  Lines of axis j are `inner` complex elements apart, inner = everything
// below the axis.  MODE 0 forward, 1 inverse (both in place), 2 forward + radial weights of the full N-D frequency + inverse:
// the outermost axis, which is transformed last on the way in and first on the way out.
// Round 3: a block is FOUR waves over one tile of 64 lines -- the waves share the staged tile and split the d outputs of every
// line between them (wave w: k = w, w + 4, ...), so a tile costs the same LDS as before but runs four times as many waves, and the
// twiddle index (n k mod d) is uniform per wave.  (The one-wave version left ~7 waves per CU at video-latent sizes.)
constexpr int AXIS_LINES = 64, AXIS_WAVES = 4, AXIS_THREADS = AXIS_LINES * AXIS_WAVES;
template <int MODE>
__global__ __launch_bounds__(AXIS_THREADS) void any_outer_axis(const AnyArgs a, int axis) {
  extern __shared__ float2 sh[];  // [d] twiddles, [d][64] values, [d][64] spectrum (MODE 2)
  const int lane = threadIdx.x & (AXIS_LINES - 1), wave = threadIdx.x >> 6, d = a.outer[axis];
  float2* tw = sh;
  float2* v = tw + d;
  float2* x = v + d * AXIS_LINES;
  for (int j = threadIdx.x; j < d; j += AXIS_THREADS) {
    float sn, cs;
    sincospif(-2.0f * (float)j / (float)d, &sn, &cs);
    tw[j] = make_float2(cs, sn);
  }
  const int64_t plane = (int64_t)a.d1 * a.d2 * a.d3h;
  int64_t inner = plane, sample = plane;
  for (int j = 0; j < a.n_outer; ++j) { sample *= a.outer[j]; if (j > axis) inner *= a.outer[j]; }
  const int64_t lines = sample / d;
  float2* base = a.spec + (int64_t)blockIdx.y * sample;
  for (int64_t q0 = (int64_t)blockIdx.x * AXIS_LINES; q0 < lines; q0 += (int64_t)gridDim.x * AXIS_LINES) {
    const int64_t q = q0 + lane;
    const bool on = q < lines;
    const int64_t hi = on ? q / inner : 0, lo = on ? q - hi * inner : 0;
    float2* line = base + hi * d * inner + lo;
    __syncthreads();  // (the tile of the trip before has been consumed; first trip: the twiddles are in place)
    for (int n = wave; n < d; n += AXIS_WAVES) v[n * AXIS_LINES + lane] = on ? line[(int64_t)n * inner] : make_float2(0.f, 0.f);
    __syncthreads();
    if (MODE == 1) {
      for (int n = wave; n < d; n += AXIS_WAVES) {
        float2 acc = make_float2(0.f, 0.f);
        int idx = 0;  // (n * k) mod d
        for (int k = 0; k < d; ++k) {
          const float2 t = tw[idx], u = v[k * AXIS_LINES + lane];
          acc.x += u.x * t.x + u.y * t.y;  // conj(t)
          acc.y += u.y * t.x - u.x * t.y;
          idx += n; if (idx >= d) idx -= d;
        }
        if (on) line[(int64_t)n * inner] = acc;
      }
      continue;
    }
    float rest = 0.f;
    if (MODE == 2) {  // squared frequency of every other axis at this line (axis == 0: hi == 0, lo enumerates them all)
      int64_t r = lo;
      const int k3 = (int)(r % a.d3h); r /= a.d3h;
      const int k2 = (int)(r % a.d2); r /= a.d2;
      const int k1 = (int)(r % a.d1); r /= a.d1;
      const float f1 = a.d1 > 1 ? axis_freq(k1, a.d1) : 0.f, f2 = a.d2 > 1 ? axis_freq(k2, a.d2) : 0.f, f3 = (float)k3 / (float)a.d3;
      rest = f1 * f1 + f2 * f2 + f3 * f3;
      for (int j = a.n_outer - 1; j >= 1; --j) {
        const int kj = (int)(r % a.outer[j]); r /= a.outer[j];
        const float fj = axis_freq(kj, a.outer[j]);
        rest += fj * fj;
      }
    }
    for (int k = wave; k < d; k += AXIS_WAVES) {
      float2 acc = make_float2(0.f, 0.f);
      int idx = 0;
      for (int n = 0; n < d; ++n) {
        const float2 t = tw[idx], u = v[n * AXIS_LINES + lane];
        acc.x += u.x * t.x - u.y * t.y;
        acc.y += u.x * t.y + u.y * t.x;
        idx += k; if (idx >= d) idx -= d;
      }
      if (MODE == 0) { if (on) line[(int64_t)k * inner] = acc; continue; }
      const float f0 = axis_freq(k, d);
      float radius = __builtin_amdgcn_sqrtf(f0 * f0 + rest) * a.inv_rmax;
      radius = radius < a.eps_clip ? a.eps_clip : radius;
      const float w = __builtin_amdgcn_exp2f(a.exponent_half_neg * __builtin_amdgcn_logf(radius));
      x[k * AXIS_LINES + lane] = make_float2(acc.x * w, acc.y * w);
    }
    if (MODE == 2) {
      __syncthreads();
      for (int n = wave; n < d; n += AXIS_WAVES) {
        float2 acc = make_float2(0.f, 0.f);
        int idx = 0;
        for (int k = 0; k < d; ++k) {
          const float2 t = tw[idx], u = x[k * AXIS_LINES + lane];
          acc.x += u.x * t.x + u.y * t.y;  // conj(t)
          acc.y += u.y * t.x - u.x * t.y;
          idx += n; if (idx >= d) idx -= d;
        }
        if (on) line[(int64_t)n * inner] = acc;
      }
    }
  }
}

// ---- TWO outer axes in one pass (round 4): channels x frames of a video latent, [C][T][plane] ---------------------------------------
// any_outer_axis moves the whole spectrum through HBM once per call, and a unit with two outer axes needs three calls (frames forward;
// channels forward + weights + inverse; frames inverse).  Here a block stages a tile of L consecutive plane positions with ALL C x T
// values of each (two LDS buffers of C T L complex), runs the frame-axis direct DFT from one buffer into the other, the channel axis
// (a power of two <= 16) as a register transform with the radial weights of the full 4-D frequency in between, the frame axis back,
// and stores: one read and one write of the spectrum instead of three.  Values are unnormalised like any_outer_axis (1/N in any_finish).
constexpr int TWO_THREADS = 256;
template <int C>
__global__ __launch_bounds__(TWO_THREADS) void any_outer_two(const AnyArgs a, int L, int logL) {
  extern __shared__ float2 sh[];  // [T] twiddles, A [C T L], B [C T L]
  const int T = a.outer[1];
  float2* tw = sh;
  float2* A = tw + T;
  float2* B = A + (size_t)C * T * L;
  for (int j = threadIdx.x; j < T; j += TWO_THREADS) {
    float sn, cs;
    sincospif(-2.0f * (float)j / (float)T, &sn, &cs);
    tw[j] = make_float2(cs, sn);
  }
  const int64_t plane = (int64_t)a.d1 * a.d2 * a.d3h, sample = plane * C * T;
  float2* base = a.spec + (int64_t)blockIdx.y * sample;
  const int CT = C * T, items = CT << logL, lmask = L - 1;
  for (int64_t p0 = (int64_t)blockIdx.x * L; p0 < plane; p0 += (int64_t)gridDim.x * L) {
    __syncthreads();  // (the tile of the trip before has been stored; first trip: the twiddles are in place)
    for (int i = threadIdx.x; i < items; i += TWO_THREADS) {
      const int l = i & lmask, ct = i >> logL;
      A[i] = p0 + l < plane ? base[(int64_t)ct * plane + p0 + l] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    // frames forward: A -> B, item (c, k, l); the twiddle index (n k) mod T is uniform over the L lanes of an item group
    for (int i = threadIdx.x; i < items; i += TWO_THREADS) {
      const int l = i & lmask, ck = i >> logL, c = ck / T, k = ck - c * T;
      const float2* src = A + ((size_t)c * T << logL) + l;
      float2 acc = make_float2(0.f, 0.f);
      int idx = 0;
      for (int n = 0; n < T; ++n) {
        const float2 t = tw[idx], u = src[(size_t)n << logL];
        acc.x += u.x * t.x - u.y * t.y;
        acc.y += u.x * t.y + u.y * t.x;
        idx += k; if (idx >= T) idx -= T;
      }
      B[i] = acc;
    }
    __syncthreads();
    // channels: item (t, l) holds its C values in registers: forward, weights of the full frequency, inverse; B in place
    for (int i = threadIdx.x; i < (T << logL); i += TWO_THREADS) {
      const int l = i & lmask, t = i >> logL;
      float2 v[C];
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = B[(((size_t)c * T + t) << logL) + l];
      skr::dft_n<C, false>(v);
      const int64_t p = p0 + l < plane ? p0 + l : 0;
      const int k3 = (int)(p % a.d3h), k2 = (int)((p / a.d3h) % a.d2);
      const float f2 = a.d2 > 1 ? axis_freq(k2, a.d2) : 0.f, f3 = (float)k3 / (float)a.d3, ft = axis_freq(t, T);
      const float rest = f2 * f2 + f3 * f3 + ft * ft;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float fc = axis_freq(c, C);
        float radius = __builtin_amdgcn_sqrtf(fc * fc + rest) * a.inv_rmax;
        radius = radius < a.eps_clip ? a.eps_clip : radius;
        const float w = __builtin_amdgcn_exp2f(a.exponent_half_neg * __builtin_amdgcn_logf(radius));
        v[c] = make_float2(v[c].x * w, v[c].y * w);
      }
      skr::dft_n<C, true>(v);
#pragma unroll
      for (int c = 0; c < C; ++c) B[(((size_t)c * T + t) << logL) + l] = v[c];
    }
    __syncthreads();
    // frames inverse: B -> global
    for (int i = threadIdx.x; i < items; i += TWO_THREADS) {
      const int l = i & lmask, cn = i >> logL, c = cn / T, n = cn - c * T;
      const float2* src = B + ((size_t)c * T << logL) + l;
      float2 acc = make_float2(0.f, 0.f);
      int idx = 0;
      for (int k = 0; k < T; ++k) {
        const float2 t = tw[idx], u = src[(size_t)k << logL];
        acc.x += u.x * t.x + u.y * t.y;  // conj(t)
        acc.y += u.y * t.x - u.x * t.y;
        idx += n; if (idx >= T) idx -= T;
      }
      if (p0 + l < plane) base[(int64_t)cn * plane + p0 + l] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void any_stats(const AnyArgs a) {
  const int64_t smp = blockIdx.y;
  const float scale = 1.0f / (float)a.unit;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.unit; e += (int64_t)gridDim.x * 256) {
    const float v = a.real[smp * a.unit + e] * scale;
    s1 += (double)v; s2 += (double)v * (double)v;
  }
  block_sums(s1, s2, a.partials + ((1 * a.batch + smp) * SLOTS + blockIdx.x) * 2);
  clear_unused_slots(a.partials + (1 * a.batch + smp) * SLOTS * 2);
}

template <typename T>
__global__ __launch_bounds__(256) void any_finish(T* out, const AnyArgs a, int has_energy, double energy) {
  const int64_t smp = blockIdx.y;
  // The sample's four sums over the SLOTS partials: by the first wave (lane l takes slots l, l + 64, ... in that order, then a fixed-order
  // shuffle tree), one LDS word for everybody else.  (Every thread used to walk all 256 slots in a loop of dependent loads: 40 us of a
  // 57 us kernel at 64 x (16, 66, 130).)
  __shared__ float factor_sh;
  if (threadIdx.x < 64) {
    double t[4] = {0, 0, 0, 0};
    for (int s = threadIdx.x; s < SLOTS; s += 64) {
      const double* pw = a.partials + ((0 * a.batch + smp) * SLOTS + s) * 2;
      const double* pc = a.partials + ((1 * a.batch + smp) * SLOTS + s) * 2;
      t[0] += pw[0]; t[1] += pw[1]; t[2] += pc[0]; t[3] += pc[1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (int o = 32; o > 0; o >>= 1) t[i] += __shfl_down(t[i], o);
    if (threadIdx.x == 0) {
      const double n = (double)a.unit;
      const double wstd = sqrt((t[1] - t[0] * t[0] / n) / (n - 1.0)), cstd = sqrt((t[3] - t[2] * t[2] / n) / (n - 1.0));
      float f = 1.0f / (float)a.unit;
      if ((float)cstd > 1e-8f) f *= has_energy ? (float)energy / (float)cstd : (float)wstd / (float)cstd;
      factor_sh = f;
    }
  }
  __syncthreads();
  const float factor = factor_sh;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.unit; e += (int64_t)gridDim.x * 256) out[smp * a.unit + e] = (T)(a.real[smp * a.unit + e] * factor);
}

// ---- plan self-check -------------------------------------------------------------------------------------------------------
// Found by the soak runs of round 2 (tools/fft_probe.py reproduces it with torch.fft alone): in a process that has created many
// plans of other shapes, rocFFT 7.2 can hand back a real multi-dimensional plan (small power-of-two lengths, last axis 8 or 32)
// that computes a WRONG transform -- 20-60 % off, persistently for that plan.  So every new plan pair is run once on unit
// impulses (one per batch entry, each at its own position), whose spectrum is known in closed form, and then back; a plan that
// fails is destroyed and the call returns SKR_ERR_LIBRARY instead of colouring noise with a broken transform.
struct SelfTest {
  float* real;
  float2* spec;
  unsigned long long* bad;
  int64_t entries;  // batch * prod(outer)
  int32_t n0, n1, n2, n2h;
};

__device__ __forceinline__ int64_t selftest_position(int64_t b, int64_t n) { return (int64_t)(((uint64_t)(b + 1) * 40503ull + 7ull) % (uint64_t)n); }

__global__ __launch_bounds__(256) void selftest_fill(const SelfTest t) {
  const int64_t n = (int64_t)t.n0 * t.n1 * t.n2, total = n * t.entries;
  if (blockIdx.x == 0 && threadIdx.x == 0) *t.bad = 0ull;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / n, e = i - b * n;
    t.real[i] = e == selftest_position(b, n) ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(256) void selftest_check_forward(const SelfTest t) {
  const int64_t n = (int64_t)t.n0 * t.n1 * t.n2, nh = (int64_t)t.n0 * t.n1 * t.n2h, total = nh * t.entries;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / nh;
    int64_t r = i - b * nh;
    const int k2 = (int)(r % t.n2h); r /= t.n2h;
    const int k1 = (int)(r % t.n1);
    const int k0 = (int)(r / t.n1);
    int64_t p = selftest_position(b, n);
    const int p2 = (int)(p % t.n2); p /= t.n2;
    const int p1 = (int)(p % t.n1);
    const int p0 = (int)(p / t.n1);
    // exp(-2 pi i (k0 p0/n0 + k1 p1/n1 + k2 p2/n2)), the phase reduced exactly in integers before it meets floating point
    const double turns = (double)(((int64_t)k0 * p0) % t.n0) / t.n0 + (double)(((int64_t)k1 * p1) % t.n1) / t.n1 + (double)(((int64_t)k2 * p2) % t.n2) / t.n2;
    double sn, cs;
    sincospi(-2.0 * turns, &sn, &cs);
    const float2 v = t.spec[i];
    if (!(fabs((double)v.x - cs) < 1e-3 && fabs((double)v.y - sn) < 1e-3)) ++bad;
  }
  if (bad) atomicAdd(t.bad, bad);
}

__global__ __launch_bounds__(256) void selftest_check_inverse(const SelfTest t) {
  const int64_t n = (int64_t)t.n0 * t.n1 * t.n2, total = n * t.entries;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / n, e = i - b * n;
    const float want = e == selftest_position(b, n) ? (float)n : 0.0f;  // unnormalised round trip
    if (!(fabsf(t.real[i] - want) < 1e-3f * (float)n)) ++bad;
  }
  if (bad) atomicAdd(t.bad, bad);
}

}  // namespace

// (skr_colored.hip) rfft2 / irfft2 of independent planes on the LDS plane kernels
namespace skr {
int colored_planes(int mode, float2* spec, double* plane_partials, float* real_out, const uint64_t* seeds, uint64_t stream_id,
                   int64_t batch, int64_t planes, int32_t d2, int32_t d3, hipStream_t s);
}

// the plane kernels leave one (sum, sum of squares) per drawn plane: folded into the SLOTS white-noise slots of the sample that
// any_finish reads (slot t = planes t, t + SLOTS, ... in that order)
__global__ __launch_bounds__(SLOTS) void any_fold_plane_sums(const AnyArgs a, const double* plane_sums, int64_t planes) {
  const int64_t smp = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t p = threadIdx.x; p < planes; p += SLOTS) { s1 += plane_sums[(smp * planes + p) * 2]; s2 += plane_sums[(smp * planes + p) * 2 + 1]; }
  double* slot = a.partials + ((0 * a.batch + smp) * SLOTS + threadIdx.x) * 2;
  slot[0] = s1; slot[1] = s2;
}

// blocks per sample of the statistics kernels: at least four items per thread, at most one block per slot
static unsigned stats_blocks(int64_t items) {
  const int64_t b = (items + 1023) / 1024;
  return (unsigned)(b < 1 ? 1 : (b > SLOTS ? SLOTS : b));
}

// one attempt with the last `fft_rank` axes given to hipFFT and every axis outside them to the direct-DFT kernels;
// planes = true (fft_rank 2, noise drawn here): the last two axes on the LDS plane kernels of skr_colored.hip instead of hipFFT --
// the white noise is drawn inside the forward plane kernel, so the fp32 buffer first carries the per-plane sums
static int colored_any_attempt(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64,
                               const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t rank, const int32_t* dims,
                               double exponent, int32_t has_energy, double energy, void* stream, bool white_given, int fft_rank, bool planes = false) {
  if (planes && (white_given || fft_rank != 2 || rank < 3)) return SKR_ERR_UNSUPPORTED;
  // (rank 4 on the plane kernels: the two outer axes go through any_outer_two -- channels a power of two <= 16, frames <= 32)
  if (planes && rank == 4 && !((dims[0] == 2 || dims[0] == 4 || dims[0] == 8 || dims[0] == 16) && dims[1] <= 32)) return SKR_ERR_UNSUPPORTED;
  if (planes && rank > 4) return SKR_ERR_UNSUPPORTED;
  // The transforms of the inner axes: the engine's own (skr_fft_own.hip: any length up to 2048, powers of two up to 4096) unless
  // SKR_FFT_HIPFFT is set or an axis is longer than that; hipFFT only then.
  static const bool env_hipfft = getenv("SKR_FFT_HIPFFT") != nullptr;
  const bool prefer_hipfft = skr::g_use_hipfft < 0 ? env_hipfft : skr::g_use_hipfft != 0;
  bool own = !planes && !prefer_hipfft;
  for (int i = (rank > fft_rank ? rank - fft_rank : 0); own && i < rank; ++i) own = skr::own_length_ok(dims[i], i == rank - 1);  // (skr_fft_own.hip)
  // hipFFT only when it is asked for (skr_set_tuning "hipfft" 1 / SKR_FFT_HIPFFT): a shape the own transforms do not take -- an axis that is
  // neither a tile length nor, as the last axis, an even product of two -- is refused, not handed to the vendor library behind the caller's back
  if (!planes && !own && (!prefer_hipfft || !api().ok)) return SKR_ERR_UNSUPPORTED;
  const int full_rank = rank;
  const int32_t* full_dims = dims;
  AnyArgs a;
  a.n_outer = 0;
  for (int j = 0; j < SKR_ANY_MAX_OUTER; ++j) a.outer[j] = 1;
  int64_t d0 = 1;  // product of the outer axes
  if (rank > fft_rank) {  // outer axes by direct DFT (any_outer_axis), the inner ones by hipFFT batched over batch * prod(outer)
    a.n_outer = rank - fft_rank;
    if (a.n_outer > SKR_ANY_MAX_OUTER) return SKR_ERR_UNSUPPORTED;
    for (int j = 0; j < a.n_outer; ++j) {
      if (dims[j] > 128) return SKR_ERR_UNSUPPORTED;  // (d + 2*d*64) float2 of LDS: 132 KiB at 128
      a.outer[j] = dims[j];
      d0 *= dims[j];
    }
    if (batch * d0 > 0x7fffffffll) return SKR_ERR_UNSUPPORTED;
    dims += a.n_outer; rank = fft_rank;
  }
  int n[3] = {1, 1, 1};
  for (int i = 0; i < rank; ++i) n[3 - rank + i] = dims[i];
  // own transforms of a unit without outer axes and with at least two transformed axes: the outermost one runs forward, weights, inverse fused
  static const bool no_fuse_outer = getenv("SKR_FFT_NO_FUSE_OUTER") != nullptr;
  const bool own_fused = own && a.n_outer == 0 && n[1] > 1 && !no_fuse_outer;  // (with outer axes the any_outer_axis passes carry the weights: every own axis must then run)
  a.real = scratch_f32; a.spec = reinterpret_cast<float2*>(spec_c64); a.partials = partials_f64; a.seeds = seeds_dev; a.stream = stream_id;
  a.batch = batch; a.d1 = n[0]; a.d2 = n[1]; a.d3 = n[2]; a.d3h = n[2] / 2 + 1; a.unit = d0 * n[0] * n[1] * n[2];
  a.exponent_half_neg = (float)(-exponent / 2.0);
  double n_eff = 0;
  for (int i = 0; i < full_rank; ++i) n_eff += full_dims[i];
  n_eff /= full_rank;
  a.eps_clip = (float)(0.5 / (n_eff > 4.0 ? n_eff : 4.0));
  float r2 = 0.f;
  for (int i = 0; i < full_rank; ++i) { const float m = (float)(full_dims[i] / 2) / (float)full_dims[i]; r2 += m * m; }
  a.inv_rmax = r2 > 0.f ? 1.0f / sqrtf(r2) : 1.0f;

  skr::DeviceGuard guard(out);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Plans plans{};
  if (!planes && !own) {
    FftApi& f = api();
    std::lock_guard<std::mutex> lock(g_mutex);
    const int64_t fft_batch = batch * d0;
    const PlanKey key = std::make_tuple(guard.dev, s, rank, n[0], n[1], n[2], fft_batch);
    if (g_bad_plans.count(key)) return SKR_ERR_LIBRARY;  // failed its self-check before (the defect is persistent): do not re-plan per call
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      static const size_t max_plans = [] { const char* e = getenv("SKR_FFT_MAX_PLANS"); return e ? (size_t)atoll(e) : MAX_PLANS; }();
      if (g_plans.size() >= max_plans) {  // evict the least recently used pair
        auto victim = g_plans.begin();
        for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt) if (jt->second.last_use < victim->second.last_use) victim = jt;
        skr::DeviceGuard owner(nullptr);
        const int vdev = std::get<0>(victim->first);
        if (vdev >= 0 && vdev != owner.prev) (void)hipSetDevice(vdev);
        (void)hipStreamSynchronize(std::get<1>(victim->first));  // its work area may still be in use
        (void)hipGetLastError();
        f.destroy(victim->second.fwd);
        f.destroy(victim->second.inv);
        if (vdev >= 0 && vdev != owner.prev) (void)hipSetDevice(guard.dev);
        g_plans.erase(victim);
      }
      int nn[3];
      for (int i = 0; i < rank; ++i) nn[i] = dims[i];
      Plans p;
      if (f.plan_many(&p.fwd, rank, nn, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, (int)fft_batch) != 0) return SKR_ERR_UNSUPPORTED;
      if (f.plan_many(&p.inv, rank, nn, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, (int)fft_batch) != 0) { f.destroy(p.fwd); return SKR_ERR_UNSUPPORTED; }
      if (f.set_stream(p.fwd, s) != 0 || f.set_stream(p.inv, s) != 0) { f.destroy(p.fwd); f.destroy(p.inv); return SKR_ERR_LAUNCH; }
      {  // self-check on unit impulses (see SelfTest above); uses this call's own workspaces before they carry anything
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
          (void)hipGetLastError();
          f.destroy(p.fwd); f.destroy(p.inv);
          return SKR_ERR_CAPTURE;  // plans are created (and checked) outside stream capture: run the shape once eagerly first
        }
        SelfTest t;
        float* test_real = a.real;
        if (white_given) {  // a.real holds the caller's white noise: the impulses get a temporary buffer of their own
          if (hipMalloc(reinterpret_cast<void**>(&test_real), sizeof(float) * (size_t)fft_batch * n[0] * n[1] * n[2]) != hipSuccess) {
            (void)hipGetLastError();
            f.destroy(p.fwd); f.destroy(p.inv);
            return SKR_ERR_LAUNCH;
          }
        }
        t.real = test_real; t.spec = a.spec; t.bad = reinterpret_cast<unsigned long long*>(a.partials); t.entries = fft_batch;
        t.n0 = n[0]; t.n1 = n[1]; t.n2 = n[2]; t.n2h = n[2] / 2 + 1;
        int64_t tb = ((int64_t)fft_batch * n[0] * n[1] * n[2] + 255) / 256; if (tb > 256 * 16) tb = 256 * 16;
        unsigned long long bad = 1;
        hipLaunchKernelGGL(selftest_fill, dim3((unsigned)tb), dim3(256), 0, s, t);
        const bool ran = f.r2c(p.fwd, t.real, t.spec) == 0;
        if (ran) hipLaunchKernelGGL(selftest_check_forward, dim3((unsigned)tb), dim3(256), 0, s, t);
        const bool back = ran && f.c2r(p.inv, t.spec, t.real) == 0;
        if (back) hipLaunchKernelGGL(selftest_check_inverse, dim3((unsigned)tb), dim3(256), 0, s, t);
        if (!back || hipMemcpyAsync(&bad, t.bad, sizeof(bad), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) bad = 1;
        if (white_given) (void)hipFree(test_real);
        if (bad != 0) {
          (void)hipGetLastError();
          f.destroy(p.fwd); f.destroy(p.inv);
          if (g_bad_plans.size() < 4096) g_bad_plans.insert(key);
          return SKR_ERR_LIBRARY;
        }
      }
      ++g_hipfft_plans;
      it = g_plans.emplace(key, p).first;
    }
    it->second.last_use = ++g_plan_clock;
    plans = it->second;
  }

  if (own) {  // every table the transforms will need, BEFORE the first launch of this attempt (a first-use Bluestein length under stream
    // capture must not leave half a draw recorded in the caller's graph)
    const int rc = skr::own_prepare(guard.dev, n[0], n[1], n[2], s);
    if (rc != SKR_OK) return rc;
  }
  if (planes) {
    if ((reinterpret_cast<uintptr_t>(a.real) & 7) != 0) return SKR_ERR_UNSUPPORTED;
    double* plane_sums = reinterpret_cast<double*>(a.real);  // batch * d0 * 2 doubles <= batch * unit floats (a plane has >= 8 elements)
    const int rc = skr::colored_planes(0, a.spec, plane_sums, nullptr, seeds_dev, stream_id, batch, d0, a.d2, a.d3, s);
    if (rc != SKR_OK) return rc;
    hipLaunchKernelGGL(any_fold_plane_sums, dim3((unsigned)batch), dim3(SLOTS), 0, s, a, plane_sums, d0);
  } else {
    if (white_given) hipLaunchKernelGGL(any_white_stats, dim3(stats_blocks(a.unit), (unsigned)batch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(any_white, dim3(stats_blocks((a.unit + 3) / 4), (unsigned)batch), dim3(256), 0, s, a);
    if (own) {
      ++g_own_execs;
      const int rc = skr::own_rfftn(guard.dev, false, a.real, a.spec, batch * d0, n[0], n[1], n[2], s, own_fused);
      if (rc != SKR_OK) return rc;
    } else {
      ++g_hipfft_execs;
      if (api().r2c(plans.fwd, a.real, a.spec) != 0) return SKR_ERR_LAUNCH;
    }
  }
  if (a.n_outer > 0) {
    const int64_t sample = d0 * a.d1 * a.d2 * a.d3h;
    auto pass = [&](int mode, int axis) -> int {
      const int d = a.outer[axis];
      const int64_t lines = sample / d;
      int64_t ab = (lines + AXIS_LINES - 1) / AXIS_LINES; if (ab > 4096) ab = 4096;
      const size_t lds = sizeof(float2) * ((size_t)d + (mode == 2 ? 2 : 1) * (size_t)d * AXIS_LINES);
      const void* fn = mode == 0 ? reinterpret_cast<const void*>(any_outer_axis<0>) : (mode == 1 ? reinterpret_cast<const void*>(any_outer_axis<1>) : reinterpret_cast<const void*>(any_outer_axis<2>));
      if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SKR_ERR_UNSUPPORTED;
      const dim3 grid((unsigned)ab, (unsigned)batch);
      if (mode == 0) hipLaunchKernelGGL(any_outer_axis<0>, grid, dim3(AXIS_THREADS), lds, s, a, axis);
      else if (mode == 1) hipLaunchKernelGGL(any_outer_axis<1>, grid, dim3(AXIS_THREADS), lds, s, a, axis);
      else hipLaunchKernelGGL(any_outer_axis<2>, grid, dim3(AXIS_THREADS), lds, s, a, axis);
      return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
    };
    int rc;
    if (planes && a.n_outer == 2) {  // one pass for both outer axes
      const int C = a.outer[0], T = a.outer[1];
      int logL = 6;
      while (logL > 3 && ((size_t)C * T << logL) > 4096) --logL;  // two buffers of C T L complex: <= 64 KiB
      if (((size_t)C * T << logL) > 6144) return SKR_ERR_UNSUPPORTED;
      const int L = 1 << logL;
      const int64_t plane = (int64_t)a.d1 * a.d2 * a.d3h;
      int64_t tiles = (plane + L - 1) / L; if (tiles > 65535) tiles = 65535;
      const size_t lds = sizeof(float2) * ((size_t)T + 2 * ((size_t)C * T << logL));
      const dim3 grid((unsigned)tiles, (unsigned)batch);
#define SKR_TWO(CC) do { if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(any_outer_two<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SKR_ERR_UNSUPPORTED; \
                         hipLaunchKernelGGL(any_outer_two<CC>, grid, dim3(TWO_THREADS), lds, s, a, L, logL); } while (0)
      switch (C) { case 2: SKR_TWO(2); break; case 4: SKR_TWO(4); break; case 8: SKR_TWO(8); break; default: SKR_TWO(16); break; }
#undef SKR_TWO
      if (hipGetLastError() != hipSuccess) return SKR_ERR_LAUNCH;
    } else {
    for (int j = a.n_outer - 1; j >= 1; --j) if ((rc = pass(0, j)) != SKR_OK) return rc;
    if ((rc = pass(2, 0)) != SKR_OK) return rc;  // outermost: forward, weights, inverse
    for (int j = 1; j < a.n_outer; ++j) if ((rc = pass(1, j)) != SKR_OK) return rc;
    }
  } else if (own_fused) {  // the outermost transformed axis forward, the weights, and the same axis back: one pass (skr_fft_own.hip)
    const int rc = skr::own_outer_weighted(guard.dev, a.spec, batch, n[0], n[1], n[2], a.inv_rmax, a.eps_clip, a.exponent_half_neg, s);
    if (rc != SKR_OK) return rc;
  } else {
    int64_t wb = ((int64_t)a.d1 * a.d2 * a.d3h * batch + 255) / 256; if (wb > 256 * 32) wb = 256 * 32;
    hipLaunchKernelGGL(any_weights, dim3((unsigned)wb), dim3(256), 0, s, a);
  }
  if (planes) {
    const int rc = skr::colored_planes(1, a.spec, nullptr, a.real, seeds_dev, stream_id, batch, d0, a.d2, a.d3, s);
    if (rc != SKR_OK) return rc;
  } else if (own) {
    const int rc = skr::own_rfftn(guard.dev, true, a.real, a.spec, batch * d0, n[0], n[1], n[2], s, own_fused);
    if (rc != SKR_OK) return rc;
  } else if (api().c2r(plans.inv, a.spec, a.real) != 0) return SKR_ERR_LAUNCH;
  hipLaunchKernelGGL(any_stats, dim3(stats_blocks(a.unit), (unsigned)batch), dim3(256), 0, s, a);
  int64_t fb = (a.unit + 1023) / 1024;  // (64 blocks per sample starved the chip at small batches: 33 us for one 4.8 M-element sample)
  { const int64_t cap = batch >= 32 ? 64 : 2048 / batch; if (fb > cap) fb = cap; if (fb < 1) fb = 1; }
  dim3 grid((unsigned)fb, (unsigned)batch);
  switch (out_dtype) {
    case SKR_BF16: hipLaunchKernelGGL(any_finish<__bf16>, grid, dim3(256), 0, s, (__bf16*)out, a, has_energy, energy); break;
    case SKR_F16: hipLaunchKernelGGL(any_finish<_Float16>, grid, dim3(256), 0, s, (_Float16*)out, a, has_energy, energy); break;
    case SKR_F32: hipLaunchKernelGGL(any_finish<float>, grid, dim3(256), 0, s, (float*)out, a, has_energy, energy); break;
    case SKR_F64: hipLaunchKernelGGL(any_finish<double>, grid, dim3(256), 0, s, (double*)out, a, has_energy, energy); break;
    default: return SKR_ERR_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

static int colored_any_impl(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64,
                            const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t rank, const int32_t* dims,
                            double exponent, int32_t has_energy, double energy, void* stream, bool white_given) {
  if (batch < 0 || rank < 1 || rank > 3 + SKR_ANY_MAX_OUTER || !dims) return SKR_ERR_SHAPE;
  for (int i = 0; i < rank; ++i) if (dims[i] < 2) return SKR_ERR_SHAPE;
  if (batch == 0) return SKR_OK;
  if (!out || !spec_c64 || !scratch_f32 || !partials_f64 || (!white_given && !seeds_dev)) return SKR_ERR_NULL;
  if (batch > 65535) return SKR_ERR_UNSUPPORTED;
  // hipFFT gets the last three axes; if that plan fails its self-check (SelfTest above), only the last axis -- 1-D real plans have
  // not shown the defect -- and the direct-DFT kernels take every other axis (each <= 128 long)
  // 3-D units the dedicated kernels of skr_noise_colored do not take (a leading axis that is not a power of two <= 16: 3, 5, 12
  // channels ...) whose planes fit the LDS plane kernels: planes there, the leading axis (<= 128 long) a direct DFT here -- no
  // hipFFT, 20-25 % faster (tools/bench_colored_planes.py).  Measured and NOT routed this way: units with two or more outer axes
  // (video latents, channels x frames x height x width: the frame axis costs two more direct-DFT passes where hipFFT's 3-D plan
  // has it inside -- 0.135 vs 0.128 ms at 2 x (16, 21, 64, 64)) and planes with large odd factors (90 x 160: 0.44 vs 0.25 ms, the
  // 45-term combining pass), see colored_planes' limit on the odd parts.
  if (!white_given && (rank == 3 || rank == 4) && skr::g_fft_rank < 1) {
    const int rcp = colored_any_attempt(out, out_dtype, spec_c64, scratch_f32, partials_f64, seeds_dev, stream_id, batch, rank, dims, exponent, has_energy, energy, stream, false, 2, true);
    if (rcp != SKR_ERR_UNSUPPORTED) return rcp;
  }
  int first = rank < 3 ? rank : 3;
  if (skr::g_fft_rank >= 1 && skr::g_fft_rank < first) first = skr::g_fft_rank;  // test switch (skr_set_tuning "fft_rank"): 1 exercises the fallback
  int rc = colored_any_attempt(out, out_dtype, spec_c64, scratch_f32, partials_f64, seeds_dev, stream_id, batch, rank, dims, exponent, has_energy, energy, stream, white_given, first);
  if (rc == SKR_ERR_LIBRARY && first > 1) {
    const int rc1 = colored_any_attempt(out, out_dtype, spec_c64, scratch_f32, partials_f64, seeds_dev, stream_id, batch, rank, dims, exponent, has_energy, energy, stream, white_given, 1);
    if (rc1 != SKR_ERR_UNSUPPORTED) rc = rc1;  // (an axis too long for the direct kernels: the library error stands)
  }
  return rc;
}

extern "C" int skr_noise_colored_any(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64,
                                     const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t rank, const int32_t* dims,
                                     double exponent, int32_t has_energy, double energy, void* stream) {
  return colored_any_impl(out, out_dtype, spec_c64, scratch_f32, partials_f64, seeds_dev, stream_id, batch, rank, dims, exponent, has_energy, energy, stream, false);
}

extern "C" int skr_colorize(void* out, int32_t out_dtype, void* spec_c64, float* white_f32, double* partials_f64, int64_t batch, int32_t rank,
                            const int32_t* dims, double exponent, int32_t has_energy, double energy, void* stream) {
  return colored_any_impl(out, out_dtype, spec_c64, white_f32, partials_f64, nullptr, 0, batch, rank, dims, exponent, has_energy, energy, stream, true);
}

extern "C" int64_t skr_stat(const char* key) {
  if (!key) return -1;
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!strcmp(key, "hipfft_plans")) return g_hipfft_plans;
  if (!strcmp(key, "hipfft_execs")) return g_hipfft_execs;
  if (!strcmp(key, "own_fft_execs")) return g_own_execs;
  return -1;
}
