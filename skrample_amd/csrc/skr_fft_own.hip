// Real N-D transforms of ANY axis lengths on the engine's own LDS tile transform (round 4): what skr_colored_any.hip asked hipFFT for.
//
// Reference: skrample/pytorch/noise.py:395-425 calls torch.fft.rfftn / irfftn over the unit's axes, whatever their lengths (152 x 104
// latents, odd heights, 13 frames ...).  The plane kernels of skr_colored.hip take powers of two and 2^a x (odd <= 63); everything else
// went to a vendor library.  Here an axis of length n is transformed
//   * directly by fft_tile when n is a power of two (<= 4096);
//   * directly when n = 2^a r, a >= 1, r a product of at most three factors out of 3, 5, 7, 11, 13 (n <= 4096: 6, 10, 12, 14, 20, 22, 26,
//     30, 60, 66, 90, 120, 130, 250, 720, 1280 ...): the r interleaved sub-sequences of 2^a points each go through fft_tile and are joined
//     by one radix-3 / 5 / 7 / 11 / 13 pass per factor (own_radix), innermost decimation first;
//   * otherwise
//   * by Bluestein's chirp-z identity  n k = (n^2 + k^2 - (k - n)^2) / 2 :
//       X[k] = w[k] * sum_j (x[j] w[j]) conj(w[k - j]),   w[k] = exp(-i pi k^2 / n),
//     a cyclic convolution of any length m >= 2n - 1 done with two transforms of m points: a = x w zero-padded -> FFT_m, times
//     K = FFT_m(conj(w) wrapped) (a table per length, computed once in float64), inverse FFT_m, times w.  Any n <= 2048.
//     m = r p with p a power of two and r = 1, 3 or 5, whichever is cheapest (130 points: 320 instead of 512; 1280: 2560 instead of
//     4096): the tile holds a line as r segments of p points, one radix-r pass (own_radix) splits FFT_m into r transforms of p points
//     on the way in and joins them on the way out.
//     The first transform is decimation in frequency (natural order in, bit-reversed out: fft_tile_dif below, which also applies K,
//     stored bit-reversed, in its last pass) and the second fft_tile's decimation in time (bit-reversed in, natural out): no
//     permutation pass and no scattered LDS access anywhere in the convolution.
// Angles are reduced exactly (k^2 mod 2n in integers) and evaluated in float64 when the tables are built; the transforms run in fp32
// like the rest of the generator.  The inverse transform is conj(DFT(conj(x))), so one set of tables and one code path serve both signs.
// The last axis is real: two real lines ride one complex transform (z = a + i b) and are untangled with the Hermitian symmetry, the
// same trick as the plane kernels' row pairs; its inverse builds z's full spectrum from the two half spectra (the imaginary parts of
// the DC and Nyquist bins are ignored, as a C2R transform does).  Results are unnormalised both ways, like the library transforms they
// replace (any_finish divides by N).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>

#include "skr_device.h"
#include "../../include/skrample_hip.h"
#include "skr_fft_tile.h"

namespace skr {

struct OwnAxis {
  int32_t n, m, p, logp, r;  // m = r p: size of the tile's transform (n itself when no chirp is needed); p a power of two, r odd
  int32_t nlev, rad[3];      // r = rad[0] * ... * rad[nlev - 1], each 3 or 5 (outermost decimation first); Bluestein: at most one level
  const float2* chirp;       // w[k], k < n (nullptr: power of two, no chirp)
  const float2* kernel;      // K = FFT_m(b), b[j] = conj(w[j]) for |j| < n (indices mod m), 0 elsewhere, in the order the forward
                             // transform leaves a line: position s p + q holds bin r bitrev_p(q) + s
};

namespace {

constexpr int OWN_MAX_M = 4096;

// ---- tables --------------------------------------------------------------------------------------------------------------------------
// One thread per table entry; the 2n - 1 non-zero terms of the kernel's transform summed directly in float64 with exactly reduced
// angles (a few ms once per length and process).
__global__ __launch_bounds__(256) void own_tables(float2* chirp, float2* kernel, int n, int m, int p, int logp, int r) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  auto w_conj = [&](int q) {  // conj(w[q]) = exp(+i pi q^2 / n)
    const int64_t t = ((int64_t)q * q) % (2 * (int64_t)n);
    double sn, cs;
    sincospi((double)t / (double)n, &sn, &cs);
    return make_double2(cs, sn);
  };
  if (e < n) {
    const double2 v = w_conj(e);
    chirp[e] = make_float2((float)v.x, (float)-v.y);
  }
  if (e >= m) return;
  const int s = e / p, q = e - s * p;
  const int64_t bin = (int64_t)r * (__brev((unsigned)q) >> (32 - logp)) + s;  // (p = 1 cannot happen: m >= 3)
  double re = 1.0, im = 0.0;  // j = 0: conj(w[0]) = 1
  for (int j = 1; j < n; ++j) {  // b[j] = b[m - j] = conj(w[j]): 2 b cos(2 pi j bin / m)
    const double2 bj = w_conj(j);
    double sn, cs;
    sincospi(2.0 * (double)((j * bin) % m) / (double)m, &sn, &cs);
    re += 2.0 * bj.x * cs;
    im += 2.0 * bj.y * cs;
  }
  kernel[e] = make_float2((float)re, (float)im);
}

// ---- the transform of a tile ---------------------------------------------------------------------------------------------------------
// Decimation in frequency over L lines of N points (pitch N + 1): natural order in, bit-reversed order out, forward sign.  Two radix-2
// stages per LDS round trip like fft_tile: an item takes the points p, p + q, p + 2q, p + 3q of a block of 4q through the stages of span
// 2q and q; one twiddle read (W_4q^j; W_2q^j is its square, W_4q^(j+q) = -i W_4q^j).  A last radix-2 stage when log2 N is odd.
// `table` multiplies the results in the last pass: line l takes table + (l mod table_lines) N, entries in the output's (bit-reversed) order.
__device__ __forceinline__ void fft_tile_dif(float2* buf, const float2* tw, int N, int logN, int L, const float2* table, int table_lines) {
  const int ld = N + 1;
  const bool by_line = L >= 2;  // (small spans: consecutive lanes on consecutive LINES, an odd pitch apart -- see fft_tile)
  const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)L - 1) / (uint32_t)L);
  int span = logN;  // log2 of the current block size
  for (; span >= 2; span -= 2) {
    const int q = 1 << (span - 2), step = N >> span;
    const bool last = span == 2;
    __syncthreads();
    const int total = L * (N >> 2);
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      int line, kk;
      if (by_line && q < 32) { kk = (int)__umulhi((uint32_t)t, magic); line = t - kk * L; }
      else { line = t >> (logN - 2); kk = t & ((N >> 2) - 1); }
      const int j = kk & (q - 1), at = ((kk >> (span - 2)) << span) + j;
      float2* p = buf + line * ld + at;
      const float2 x0 = p[0], x1 = p[q], x2 = p[2 * q], x3 = p[3 * q];
      const float2 w1 = tw[j * step];
      const float2 w2 = make_float2(__builtin_fmaf(w1.x, w1.x, -(w1.y * w1.y)), 2.f * w1.x * w1.y);
      const float2 y0 = cadd(x0, x2), y2 = cmul(csub(x0, x2), w1), y1 = cadd(x1, x3), y3 = mul_i<false>(cmul(csub(x1, x3), w1));
      float2 z0 = cadd(y0, y1), z1 = cmul(csub(y0, y1), w2), z2 = cadd(y2, y3), z3 = cmul(csub(y2, y3), w2);
      if (last && table) {
        const float2* tb = table + (line % table_lines) * N + at;
        z0 = cmul(z0, tb[0]); z1 = cmul(z1, tb[1]); z2 = cmul(z2, tb[2]); z3 = cmul(z3, tb[3]);
      }
      p[0] = z0; p[q] = z1; p[2 * q] = z2; p[3 * q] = z3;
    }
  }
  if (span == 1) {
    __syncthreads();
    const int total = L * (N >> 1);
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      int line, kk;
      if (by_line) { kk = (int)__umulhi((uint32_t)t, magic); line = t - kk * L; }
      else { line = t >> (logN - 1); kk = t & ((N >> 1) - 1); }
      float2* p = buf + line * ld + 2 * kk;
      float2 z0 = cadd(p[0], p[1]), z1 = csub(p[0], p[1]);
      if (table) {
        const float2* tb = table + (line % table_lines) * N + 2 * kk;
        z0 = cmul(z0, tb[0]); z1 = cmul(z1, tb[1]);
      }
      p[0] = z0; p[1] = z1;
    }
  }
}

// ---- radix-r pass between FFT_m and r transforms of p points (m = r p) --------------------------------------------------------------
// A logical line is r consecutive tile lines (segments) of p points: x[t p + j] at (segment t, offset j).  Decimation in frequency,
//     y_s[j] = W_m^(j s) sum_t x[t p + j] W_r^(t s),      X[r k + s] = FFT_p(y_s)[k],
// in place over the r segments (an item owns offset j of all of them); JOIN = the transposed pass of the way back,
//     y[t p + j] = sum_s W_r^(-t s) W_m^(-j s) z_s[j],    z_s = FFT_p^-1 of the bins r k + s.
template <bool INV> __device__ __forceinline__ void dft3(float2 v[3]) {
  constexpr float h = 0.86602540378443864676f;
  const float2 t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const float2 c = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
  const float2 e = INV ? make_float2(-h * d.y, h * d.x) : make_float2(h * d.y, -h * d.x);  // -+ i h d
  v[0] = cadd(v[0], t); v[1] = cadd(c, e); v[2] = csub(c, e);
}
template <bool INV> __device__ __forceinline__ void dft5(float2 v[5]) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
  const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
  const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
  float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y), b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
  b1 = INV ? make_float2(-b1.y, b1.x) : make_float2(b1.y, -b1.x);  // -+ i b
  b2 = INV ? make_float2(-b2.y, b2.x) : make_float2(b2.y, -b2.x);
  v[0] = cadd(v[0], cadd(t1, t2));
  v[1] = cadd(a1, b1); v[4] = csub(a1, b1); v[2] = cadd(a2, b2); v[3] = csub(a2, b2);
}
// 7, 11 and 13 points (round 5: lengths such as 66 = 2 3 11 and 130 = 2 5 13 went through Bluestein, 1.5x slower than the vendor library):
// the direct sum over the (R - 1) / 2 conjugate pairs,  X[k], X[R - k] = x0 + sum_j (x_j + x_(R-j)) cos(2 pi j k / R) -+ i sum_j (x_j - x_(R-j)) sin(2 pi j k / R)
// -- (R - 1)^2 / 2 real-by-complex multiply-adds, every index a compile-time constant.
template <int R> struct PrimeTable;
template <> struct PrimeTable<7> {
  static constexpr float c[3] = {0.62348980185873359f, -0.22252093395631434f, -0.90096886790241903f};
  static constexpr float s[3] = {0.7818314824680298f, 0.97492791218182362f, 0.43388373911755823f};
};
template <> struct PrimeTable<11> {
  static constexpr float c[5] = {0.84125353283118121f, 0.41541501300188644f, -0.142314838273285f, -0.65486073394528499f, -0.95949297361449737f};
  static constexpr float s[5] = {0.54064081745559756f, 0.90963199535451833f, 0.9898214418809328f, 0.75574957435425827f, 0.28173255684142967f};
};
template <> struct PrimeTable<13> {
  static constexpr float c[6] = {0.88545602565320991f, 0.56806474673115592f, 0.12053668025532301f, -0.35460488704253545f, -0.74851074817110119f, -0.97094181742605201f};
  static constexpr float s[6] = {0.46472317204376851f, 0.82298386589365635f, 0.99270887409805397f, 0.93501624268541483f, 0.66312265824079519f, 0.23931566428755768f};
};
template <int R, bool INV> __device__ __forceinline__ void dft_prime(float2 v[R]) {
  constexpr int H = (R - 1) / 2;
  float2 t[H], d[H];
#pragma unroll
  for (int j = 0; j < H; ++j) { t[j] = cadd(v[j + 1], v[R - 1 - j]); d[j] = csub(v[j + 1], v[R - 1 - j]); }
  float2 sum = v[0];
#pragma unroll
  for (int j = 0; j < H; ++j) sum = cadd(sum, t[j]);
  const float2 x0 = v[0];
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    float2 a = x0, b = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 1; j <= H; ++j) {
      const int m = (j * k) % R;  // cos(2 pi m / R) = c[min(m, R - m) - 1], sin: s[..] with the sign of the half m falls in
      const float cs = PrimeTable<R>::c[(m <= H ? m : R - m) - 1], sn = (m <= H ? 1.f : -1.f) * PrimeTable<R>::s[(m <= H ? m : R - m) - 1];
      a = make_float2(__builtin_fmaf(cs, t[j - 1].x, a.x), __builtin_fmaf(cs, t[j - 1].y, a.y));
      b = make_float2(__builtin_fmaf(sn, d[j - 1].x, b.x), __builtin_fmaf(sn, d[j - 1].y, b.y));
    }
    const float2 ib = INV ? make_float2(-b.y, b.x) : make_float2(b.y, -b.x);  // -+ i b
    v[k] = cadd(a, ib);
    v[R - k] = csub(a, ib);
  }
  v[0] = sum;
}
template <int R, bool INV> __device__ __forceinline__ void dft_odd(float2 v[R]) {
  if constexpr (R == 3) dft3<INV>(v);
  else if constexpr (R == 5) dft5<INV>(v);
  else dft_prime<R, INV>(v);
}
// FWD_JOIN: the joining pass with the forward sign -- a step of a DIRECT transform whose R decimated sub-sequences x[R i + s] were
// transformed in the segments:  X[t q + j] = sum_s W_R^(t s) W_(R q)^(j s) Z_s[j],  q = the segments' length.
// Levels: a logical line is `lines` tile lines; at this level it splits into `groups` independent joins of R segments, each segment
// `seg` tile lines (seg p points, natural order: point j at tile line j / p, offset j % p).  twl = exp(-2 pi i j / (R seg p)), j < seg p.
template <int R, bool JOIN, bool FWD_JOIN = false>
__device__ __forceinline__ void own_radix(float2* tile, const float2* twl, int p, int logp, int L, int logL, int lines, int seg, int groups) {
  const int pitch = p + 1, q = seg << logp, per_line = groups * q;
  __syncthreads();
  for (int t = threadIdx.x; t < L * per_line; t += blockDim.x) {
    const int l = t & (L - 1), rest = t >> logL;  // consecutive lanes: consecutive logical lines, lines (p + 1) points apart -- an odd pitch
    const int g = groups == 1 ? 0 : rest / q, j = rest - g * q;
    float2* at = tile + (l * lines + g * R * seg + (j >> logp)) * pitch + (j & (p - 1));
    const int hop = seg * pitch;
    float2 v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = at[i * hop];
    float2 w = twl[j];
    if (JOIN && !FWD_JOIN) w.y = -w.y;
    if (JOIN) {
      float2 ws = w;
#pragma unroll
      for (int i = 1; i < R; ++i) { v[i] = cmul(v[i], ws); ws = cmul(ws, w); }
    }
    dft_odd<R, JOIN && !FWD_JOIN>(v);
    if (!JOIN) {
      float2 ws = w;
#pragma unroll
      for (int i = 1; i < R; ++i) { v[i] = cmul(v[i], ws); ws = cmul(ws, w); }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) at[i * hop] = v[i];
  }
}
// BIG: the instantiation that also holds the 7 / 11 / 13-point passes (their 13 complex values in flight cost every kernel 20 registers and two
// waves per SIMD: lengths without those factors run the lean one)
template <bool JOIN, bool FWD_JOIN, bool BIG>
__device__ __forceinline__ void own_radix_any(int R, float2* tile, const float2* twl, int p, int logp, int L, int logL, int lines, int seg, int groups) {
  if constexpr (!BIG) {
    if (R == 3) own_radix<3, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups);
    else own_radix<5, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups);
    return;
  }
  switch (R) {
    case 3: own_radix<3, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups); break;
    case 5: own_radix<5, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups); break;
    case 7: own_radix<7, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups); break;
    case 11: own_radix<11, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups); break;
    default: own_radix<13, JOIN, FWD_JOIN>(tile, twl, p, logp, L, logL, lines, seg, groups); break;
  }
}

// The tile: L logical lines, each r segments of p points, every segment p + 1 apart.  own_put fills it: x[k] at the bit-reversed
// position (power of two), or x[k] w[k] in natural order (segment k / p, offset k % p) with own_clear's zeros from n on (Bluestein).
// After own_transform position k < n holds the forward DFT -- for Bluestein short of the factor w[k] / m, which own_get applies.
__device__ __forceinline__ int own_at(const OwnAxis& ax, int k) { return (k >> ax.logp) * (ax.p + 1) + (k & (ax.p - 1)); }
__device__ __forceinline__ float2* own_line(float2* tile, const OwnAxis& ax, int l) { return tile + l * ax.r * (ax.p + 1); }
template <bool BIG>
__device__ __forceinline__ void own_transform(float2* tile, const float2* tw, const float2* twm, const OwnAxis& ax, int L, int logL) {
  if (!ax.chirp) {  // direct: the r decimated sub-sequences through fft_tile, then the joins from the innermost level outwards
    fft_tile<false>(tile, tw, ax.p, ax.logp, L * ax.r, nullptr, 0, true);
    int seg = 1, groups = ax.r;
    const float2* twl = twm;
    for (int lev = ax.nlev - 1; lev >= 0; --lev) {
      groups /= ax.rad[lev];
      own_radix_any<true, true, BIG>(ax.rad[lev], tile, twl, ax.p, ax.logp, L, logL, ax.r, seg, groups);
      twl += seg << ax.logp;
      seg *= ax.rad[lev];
    }
    if (ax.nlev) __syncthreads();
    return;
  }
  if (ax.nlev) own_radix_any<false, false, BIG>(ax.rad[0], tile, twm, ax.p, ax.logp, L, logL, ax.r, 1, 1);
  fft_tile_dif(tile, tw, ax.p, ax.logp, L * ax.r, ax.kernel, ax.r);
  fft_tile<true>(tile, tw, ax.p, ax.logp, L * ax.r, nullptr, 0, true);  // (every pass opens with a barrier, and the transform ends with one)
  if (ax.nlev) { own_radix_any<true, false, BIG>(ax.rad[0], tile, twm, ax.p, ax.logp, L, logL, ax.r, 1, 1); __syncthreads(); }
}
template <bool INVERSE>
__device__ __forceinline__ void own_put(float2* line, const OwnAxis& ax, int k, float2 v) {
  if (INVERSE) v.y = -v.y;
  if (ax.chirp) line[own_at(ax, k)] = cmul(v, ax.chirp[k]);
  else {  // direct: x[k] to the tile line of its residues (outermost decimation first), bit-reversed position of what is left of k
    int rest = k, tl = 0;
    for (int lev = 0; lev < ax.nlev; ++lev) {
      const int rd = ax.rad[lev];  // rest / rd by the rounded-up reciprocal (exact far beyond 4096)
      const uint32_t magic = rd == 3 ? 0x55555556u : (rd == 5 ? 0x33333334u : (rd == 7 ? 0x24924925u : (rd == 11 ? 0x1745d175u : 0x13b13b14u)));
      const int q = (int)__umulhi((uint32_t)rest, magic);
      tl = tl * ax.rad[lev] + (rest - q * ax.rad[lev]);
      rest = q;
    }
    line[tl * (ax.p + 1) + (int)brev((unsigned)rest, ax.logp)] = v;
  }
}
template <bool INVERSE>
__device__ __forceinline__ float2 own_get(const float2* line, const OwnAxis& ax, int k, float inv_m) {
  float2 y = line[own_at(ax, k)];  // (r = 1: k itself)
  if (ax.chirp) { y = cmul(y, ax.chirp[k]); y.x *= inv_m; y.y *= inv_m; }
  if (INVERSE) y.y = -y.y;
  return y;
}
// Bluestein: positions n .. m - 1 of every line are zero (a power of two fills every position).  No barrier: the lines' other
// positions are written by own_put, and the transform's first pass opens with one.
__device__ __forceinline__ void own_clear(float2* tile, const OwnAxis& ax, int L) {
  if (!ax.chirp) return;
  const int pad = ax.m - ax.n;
  for (int t = threadIdx.x; t < L * pad; t += blockDim.x) {
    const int line = t / pad, k = ax.n + (t - line * pad);
    own_line(tile, ax, line)[own_at(ax, k)] = make_float2(0.f, 0.f);
  }
}
// LDS of a block: twiddles of the p-point transforms (p / 2), the joins' twiddles level by level from the innermost
// (level with segments of q points joined R at a time: exp(-2 pi i j / (R q)), j < q), the tile
__device__ __host__ __forceinline__ int own_join_entries(const OwnAxis& ax) {
  int total = 0, seg = 1;
  for (int lev = ax.nlev - 1; lev >= 0; --lev) { total += seg * ax.p; seg *= ax.rad[lev]; }
  return total;
}
__device__ __forceinline__ void own_setup(float2* smem, const OwnAxis& ax, float2*& tw, float2*& twm, float2*& tile) {
  tw = smem;
  twm = smem + ax.p / 2;
  tile = twm + own_join_entries(ax);
  make_twiddles(tw, ax.p);
  float2* twl = twm;
  int seg = 1;
  for (int lev = ax.nlev - 1; lev >= 0; --lev) {
    const int q = seg * ax.p;
    const float scale = -2.0f / (float)(q * ax.rad[lev]);
    for (int j = threadIdx.x; j < q; j += blockDim.x) {
      float sn, cs;
      sincospif(scale * (float)j, &sn, &cs);
      twl[j] = make_float2(cs, sn);
    }
    twl += q;
    seg *= ax.rad[lev];
  }
}

// ---- last axis: real lines <-> half spectra, two lines per transform --------------------------------------------------------------
// The block's pairs of lines are walked in chunks of 64 values (one wave, one chunk: runs of 4-byte / 8-byte accesses along the line), the
// a wave per pair -- unless the tile holds fewer pairs than the block has waves (one or two long lines: 2002 points), where the chunks of all
// pairs are dealt round-robin to the waves instead: a wave per pair left three of four idle and one wave walking 32 dependent steps
// ((2, 2002) x 256: 90 -> 66 us per draw, profiles/r05_bench_fft_own.txt).
template <typename F> __device__ __forceinline__ void own_walk(int pairs_here, int n, F&& f) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
  if (pairs_here >= waves) {  // the usual tile: a wave per pair, the pair's addresses hoisted out of its walk
    for (int pl = wave; pl < pairs_here; pl += waves)
      for (int k = lane; k < n; k += 64) f(pl, k);
    return;
  }
  const int chunks = (n + 63) >> 6;
  for (int it = wave; it < pairs_here * chunks; it += waves) {
    const int pl = it / chunks, k = ((it - pl * chunks) << 6) + lane;  // (wave-uniform division)
    if (k < n) f(pl, k);
  }
}
template <bool BIG>
__global__ __launch_bounds__(FFT_THREADS) void own_last_forward(const float* real, float2* spec, int64_t lines, OwnAxis ax, int L, int logL) {
  extern __shared__ float2 smem[];
  float2 *tw, *twm, *tile;
  own_setup(smem, ax, tw, twm, tile);
  const int n = ax.n, nh = n / 2 + 1;
  const float inv_m = 1.0f / (float)ax.m;
  const int64_t pairs = (lines + 1) / 2, p0 = (int64_t)blockIdx.x * L;
  const int here = (int)(pairs - p0 < L ? pairs - p0 : L);
  own_clear(tile, ax, L);
  own_walk(here, n, [&](int pl, int k) {
    const int64_t la = 2 * (p0 + pl), lb = la + 1;
    own_put<false>(own_line(tile, ax, pl), ax, k, make_float2(real[la * n + k], lb < lines ? real[lb * n + k] : 0.f));
  });
  own_transform<BIG>(tile, tw, twm, ax, L, logL);
  own_walk(here, nh, [&](int pl, int k) {
    const int64_t la = 2 * (p0 + pl), lb = la + 1;
    const float2 zk = own_get<false>(own_line(tile, ax, pl), ax, k, inv_m), zn = own_get<false>(own_line(tile, ax, pl), ax, k ? n - k : 0, inv_m);
    spec[la * nh + k] = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    if (lb < lines) spec[lb * nh + k] = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
  });
}

template <bool BIG>
__global__ __launch_bounds__(FFT_THREADS) void own_last_inverse(const float2* spec, float* real, int64_t lines, OwnAxis ax, int L, int logL) {
  extern __shared__ float2 smem[];
  float2 *tw, *twm, *tile;
  own_setup(smem, ax, tw, twm, tile);
  const int n = ax.n, nh = n / 2 + 1;
  const float inv_m = 1.0f / (float)ax.m;
  const int64_t pairs = (lines + 1) / 2, p0 = (int64_t)blockIdx.x * L;
  const int here = (int)(pairs - p0 < L ? pairs - p0 : L);
  own_clear(tile, ax, L);
  own_walk(here, n, [&](int pl, int k) {
    const int64_t la = 2 * (p0 + pl), lb = la + 1;
    const int f = k < nh ? k : n - k;
    float2 xa = spec[la * nh + f], xb = lb < lines ? spec[lb * nh + f] : make_float2(0.f, 0.f);
    if (f == 0 || 2 * f == n) { xa.y = 0.f; xb.y = 0.f; }
    if (k >= nh) { xa.y = -xa.y; xb.y = -xb.y; }
    own_put<true>(own_line(tile, ax, pl), ax, k, make_float2(xa.x - xb.y, xa.y + xb.x));
  });
  own_transform<BIG>(tile, tw, twm, ax, L, logL);
  own_walk(here, n, [&](int pl, int k) {
    const int64_t la = 2 * (p0 + pl), lb = la + 1;
    const float2 z = own_get<true>(own_line(tile, ax, pl), ax, k, inv_m);
    real[la * n + k] = z.x;
    if (lb < lines) real[lb * n + k] = z.y;
  });
}

// ---- any other axis: complex, in place; line l = (o, i), element k at (o n + k) inner + i ------------------------------------------
// L is a power of two <= FFT_THREADS: a thread keeps ONE line (consecutive lanes = consecutive i: runs of 8-byte accesses along the
// fastest axis, LDS lines an odd pitch apart) and walks its elements FFT_THREADS / L apart; one 64-bit division per thread.
template <bool INVERSE, bool BIG>
__global__ __launch_bounds__(FFT_THREADS) void own_strided(float2* spec, int64_t lines, int64_t inner, OwnAxis ax, int L, int logL) {
  extern __shared__ float2 smem[];
  float2 *tw, *twm, *tile;
  own_setup(smem, ax, tw, twm, tile);
  const int n = ax.n;
  const float inv_m = 1.0f / (float)ax.m;
  const int64_t l0 = (int64_t)blockIdx.x * L;
  const int here = (int)(lines - l0 < L ? lines - l0 : L);
  const int pl = threadIdx.x & (L - 1), k0 = threadIdx.x >> logL, kstep = blockDim.x >> logL;
  own_clear(tile, ax, L);
  float2* base = nullptr;
  if (pl < here) {
    const int64_t l = l0 + pl, o = l / inner, i = l - o * inner;
    base = spec + o * n * inner + i;
    for (int k = k0; k < n; k += kstep) own_put<INVERSE>(own_line(tile, ax, pl), ax, k, base[(int64_t)k * inner]);
  }
  own_transform<BIG>(tile, tw, twm, ax, L, logL);
  if (pl < here)
    for (int k = k0; k < n; k += kstep) base[(int64_t)k * inner] = own_get<INVERSE>(own_line(tile, ax, pl), ax, k, inv_m);
}

// ---- the OUTERMOST of the transformed axes: forward, radial weights, inverse in one tile residency ------------------------------------
// (what colored_outer_axis is to the plane kernels: the weights need the full transform, and the way back starts with this very axis, so
// the spectrum makes one round trip through HBM instead of three -- strided forward, a weights kernel, strided inverse.)
// A thread keeps its line's values (at most 16: L m <= 4096 points on 256 threads) in registers between the two transforms.
struct OwnWeights {
  int32_t n0, n1, n2;  // the unit's (up to) three transform axes; this pass runs along n0, or along n1 when n0 == 1
  float inv_rmax, eps_clip, exponent_half_neg;
};
__device__ __forceinline__ float own_axis_freq(int k, int d) { const int m = k < d - k ? k : d - k; return (float)m / (float)d; }

template <bool BIG>
__global__ __launch_bounds__(FFT_THREADS) void own_strided_weighted(float2* spec, int64_t lines, int64_t inner, OwnAxis ax, int L, int logL, OwnWeights wp) {
  extern __shared__ float2 smem[];
  float2 *tw, *twm, *tile;
  own_setup(smem, ax, tw, twm, tile);
  const int n = ax.n;
  const float inv_m = 1.0f / (float)ax.m;
  const int64_t l0 = (int64_t)blockIdx.x * L;
  const int here = (int)(lines - l0 < L ? lines - l0 : L);
  const int pl = threadIdx.x & (L - 1), k0 = threadIdx.x >> logL, kstep = blockDim.x >> logL;
  own_clear(tile, ax, L);
  float2* base = nullptr;
  float rest_sq = 0.f;  // the other axes' share of the squared radius: the same for the whole line
  if (pl < here) {
    const int64_t l = l0 + pl, o = l / inner, i = l - o * inner;
    base = spec + o * n * inner + i;
    const int n2h = wp.n2 / 2 + 1;
    if (wp.n0 > 1) {
      const int k2 = (int)(i / n2h), k3 = (int)(i - (int64_t)k2 * n2h);
      const float f2 = own_axis_freq(k2, wp.n1), f3 = (float)k3 / (float)wp.n2;
      rest_sq = f2 * f2 + f3 * f3;
    } else {
      const float f3 = (float)i / (float)wp.n2;
      rest_sq = f3 * f3;
    }
    for (int k = k0; k < n; k += kstep) own_put<false>(own_line(tile, ax, pl), ax, k, base[(int64_t)k * inner]);
  }
  own_transform<BIG>(tile, tw, twm, ax, L, logL);
  float2 hold[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j * kstep;
    if (pl < here && k < n) {
      const float f = own_axis_freq(k, n);
      float radius = __builtin_amdgcn_sqrtf(f * f + rest_sq) * wp.inv_rmax;
      radius = radius < wp.eps_clip ? wp.eps_clip : radius;
      const float w = __builtin_amdgcn_exp2f(wp.exponent_half_neg * __builtin_amdgcn_logf(radius));  // (any_weights' arithmetic)
      const float2 y = own_get<false>(own_line(tile, ax, pl), ax, k, inv_m);
      hold[j] = make_float2(y.x * w, y.y * w);
    }
  }
  __syncthreads();  // everybody has read the forward result: the tile is free for the way back
  own_clear(tile, ax, L);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j * kstep;
    if (pl < here && k < n) own_put<true>(own_line(tile, ax, pl), ax, k, hold[j]);
  }
  own_transform<BIG>(tile, tw, twm, ax, L, logL);
  if (pl < here)
    for (int k = k0; k < n; k += kstep) base[(int64_t)k * inner] = own_get<true>(own_line(tile, ax, pl), ax, k, inv_m);
}

// ---- a LAST axis too long for one tile (round 5): the half-length complex transform in four steps ------------------------------------
// A real line of n = 2 h points, read as h complex values z[j] = x[2 j] + i x[2 j + 1] (no copy: that IS its memory image), is transformed
// by Z = FFT_h(z) and untangled:  E = (Z[k] + conj Z[h - k]) / 2,  O = (Z[k] - conj Z[h - k]) / (2 i),  X[k] = E + W_n^k O,  k = 0 .. h.
// FFT_h with h = A B runs on the strided kernels above, in place: B-point transforms over b of z[a + A b] (stride A), a twiddle
// W_h^(a f_b), A-point transforms over a (contiguous) -- after which Z[f_b + B f_a] sits at position f_a + A f_b.  Only the two small
// kernels below know about that order: own_long_post gathers Z[k], Z[h - k] from their positions and writes X in natural order to the
// spectrum buffer; own_long_pre is its transpose for the way back (unnormalised, like every transform here: it hands FFT_h^-1 the
// values 2 Z, so that the result is n x).  Angles are reduced in integers and evaluated in float64 (h can reach millions of points).
__global__ __launch_bounds__(256) void own_long_twiddle(float2* z, int64_t total, int h, int A, int inverse) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int pos = (int)(e % h), a = pos % A, fb = pos / A;
    if (a == 0 || fb == 0) continue;
    double sn, cs;
    sincospi(2.0 * (double)(((int64_t)a * fb) % h) / (double)h, &sn, &cs);
    const float2 w = make_float2((float)cs, inverse ? (float)sn : (float)-sn);
    z[e] = cmul(z[e], w);
  }
}
__device__ __forceinline__ int64_t own_long_pos(int k, int A, int B) { return (int64_t)(k / B) + (int64_t)A * (k % B); }  // where Z[k] sits
__global__ __launch_bounds__(256) void own_long_post(const float2* z, float2* spec, int64_t lines, int h, int A, int B) {
  const int64_t total = lines * (h + 1);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t line = e / (h + 1);
    const int k = (int)(e - line * (h + 1));
    const float2* zl = z + line * h;
    const float2 zk = zl[own_long_pos(k == h ? 0 : k, A, B)], zn = zl[own_long_pos(k == 0 || k == h ? 0 : h - k, A, B)];
    const float2 ev = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)), od = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    double sn, cs;
    sincospi((double)k / (double)h, &sn, &cs);  // W_n^k = exp(-i pi k / h)
    spec[e] = cadd(ev, cmul(od, make_float2((float)cs, (float)-sn)));
  }
}
__global__ __launch_bounds__(256) void own_long_pre(const float2* spec, float2* z, int64_t lines, int h, int A, int B) {
  const int64_t total = lines * h;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t line = e / h;
    const int k = (int)(e - line * h);
    const float2* xl = spec + line * (h + 1);
    float2 xk = xl[k], xn = xl[h - k];
    if (k == 0) { xk.y = 0.f; xn.y = 0.f; }  // (the imaginary parts of the DC and Nyquist bins are ignored, as a C2R transform does)
    const float2 sm = make_float2(xk.x + xn.x, xk.y - xn.y), df = make_float2(xk.x - xn.x, xk.y + xn.y);
    double sn, cs;
    sincospi((double)k / (double)h, &sn, &cs);
    const float2 t = cmul(df, make_float2((float)cs, (float)sn));  // W_n^-k (X[k] - conj X[h - k])
    z[line * h + own_long_pos(k, A, B)] = make_float2(sm.x - t.y, sm.y + t.x);
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
std::mutex g_own_mutex;
std::map<std::tuple<int, int>, OwnAxis> g_own_axes;  // (device, n) -> tables; a few KB each, kept for the life of the process

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// m = r p >= 2 n - 1 with the least work: p log2 p butterflies per segment plus the radix-r pass
void own_size(int n, int& m, int& p, int& r) {
  double best = 1e30;
  for (int rr : {1, 3, 5})
    for (int pp = 2; pp * rr <= OWN_MAX_M; pp *= 2) {
      if (pp * rr < 2 * n - 1) continue;
      const double cost = (double)pp * rr * (ilog2(pp) + (rr > 1 ? 2.5 : 0.0));
      if (cost < best) { best = cost; m = pp * rr; p = pp; r = rr; }
      break;
    }
}

int own_axis(int dev, int n, hipStream_t s, OwnAxis& ax) {
  if (n < 2) return SKR_ERR_SHAPE;
  if ((n & (n - 1)) == 0) {
    if (n > OWN_MAX_M) return SKR_ERR_UNSUPPORTED;
    ax = OwnAxis{n, n, n, ilog2(n), 1, 0, {1, 1, 1}, nullptr, nullptr};
    return SKR_OK;
  }
  {  // n = 2^a x (at most three factors out of 3, 5, 7, 11, 13), a >= 1: interleaved power-of-two sub-sequences, no chirp, no padding
    int rest = n, nlev = 0, rad[3] = {1, 1, 1}, r = 1;
    for (int f : {13, 11, 7, 5, 3})
      while (rest % f == 0 && nlev < 3) { rad[nlev++] = f; rest /= f; r *= f; }
    if (nlev > 0 && rest >= 2 && (rest & (rest - 1)) == 0 && n <= OWN_MAX_M) {
      ax = OwnAxis{n, n, rest, ilog2(rest), r, nlev, {rad[0], rad[1], rad[2]}, nullptr, nullptr};
      return SKR_OK;
    }
  }
  if (2 * n - 1 > OWN_MAX_M) return SKR_ERR_UNSUPPORTED;
  int m = 0, p = 0, r = 0;
  own_size(n, m, p, r);
  std::lock_guard<std::mutex> lock(g_own_mutex);
  const auto key = std::make_tuple(dev, n);
  auto it = g_own_axes.find(key);
  if (it == g_own_axes.end()) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return SKR_ERR_CAPTURE;  // tables are allocated outside stream capture: run the shape once eagerly first
    }
    float2* buf = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), sizeof(float2) * (size_t)(n + m)) != hipSuccess) { (void)hipGetLastError(); return SKR_ERR_LAUNCH; }
    hipLaunchKernelGGL(own_tables, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, buf, buf + n, n, m, p, ilog2(p), r);
    // the tables are shared by every stream of the device from here on: finished before anybody can look them up (once per length)
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(buf); return SKR_ERR_LAUNCH; }
    it = g_own_axes.emplace(key, OwnAxis{n, m, p, ilog2(p), r, r > 1 ? 1 : 0, {r, 1, 1}, buf, buf + n}).first;
  }
  ax = it->second;
  return SKR_OK;
}

// logical lines per block: a power of two, about FFT_MAX_TILE points -- fewer when the launch would otherwise leave CUs without a block
// (256 transforms of 2002 points in tiles of two filled half the chip with one wave per SIMD: 28 us for 4 MB): halve the tile while
// there are fewer than two blocks per CU and a tile still gives every thread two points
int tile_lines(const OwnAxis& ax, int64_t logical_lines) {
  int L = 1;
  while (2 * L * ax.m <= FFT_MAX_TILE && 2 * L <= FFT_THREADS) L *= 2;
  while (L > 1 && (logical_lines + L - 1) / L < 2 * 256 && (L / 2) * ax.m >= 2 * FFT_THREADS) L /= 2;
  return L;
}
size_t tile_bytes(const OwnAxis& ax, int L) { return sizeof(float2) * ((size_t)ax.p / 2 + (size_t)own_join_entries(ax) + (size_t)L * ax.r * (ax.p + 1)); }

bool own_big(const OwnAxis& ax) { return ax.rad[0] > 5 || ax.rad[1] > 5 || ax.rad[2] > 5; }
template <typename K, typename... A>
int own_launch(K kernel, int64_t blocks, const OwnAxis& ax, int L, hipStream_t s, A... args) {
  if (blocks < 1) return SKR_OK;
  if (blocks > 0x7fffffffll) return SKR_ERR_UNSUPPORTED;
  const size_t lds = tile_bytes(ax, L);
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SKR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(FFT_THREADS), lds, s, args...);
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

// each kernel in its lean or its BIG instantiation (the one with the 7 / 11 / 13-point passes), by the axis's factors
template <typename... A> int launch_last_forward(int64_t blocks, const OwnAxis& ax, int L, hipStream_t s, A... args) {
  return own_big(ax) ? own_launch(own_last_forward<true>, blocks, ax, L, s, args...) : own_launch(own_last_forward<false>, blocks, ax, L, s, args...);
}
template <typename... A> int launch_last_inverse(int64_t blocks, const OwnAxis& ax, int L, hipStream_t s, A... args) {
  return own_big(ax) ? own_launch(own_last_inverse<true>, blocks, ax, L, s, args...) : own_launch(own_last_inverse<false>, blocks, ax, L, s, args...);
}
template <bool INVERSE, typename... A> int launch_strided(int64_t blocks, const OwnAxis& ax, int L, hipStream_t s, A... args) {
  return own_big(ax) ? own_launch(own_strided<INVERSE, true>, blocks, ax, L, s, args...) : own_launch(own_strided<INVERSE, false>, blocks, ax, L, s, args...);
}
template <typename... A> int launch_strided_weighted(int64_t blocks, const OwnAxis& ax, int L, hipStream_t s, A... args) {
  return own_big(ax) ? own_launch(own_strided_weighted<true>, blocks, ax, L, s, args...) : own_launch(own_strided_weighted<false>, blocks, ax, L, s, args...);
}

// a length the tile transforms take: the direct forms up to 4096, anything up to 2048 through Bluestein
bool own_short(int n) {
  if (n < 2) return false;
  if (n <= (OWN_MAX_M + 1) / 2) return true;
  if (n > OWN_MAX_M) return false;
  int rest = n, odd = 0;
  for (int f : {13, 11, 7, 5, 3})
    while (rest % f == 0 && odd < 3) { rest /= f; ++odd; }
  return (rest & (rest - 1)) == 0 && (odd == 0 || rest >= 2);
}
// n = 2 A B for a long last axis: both factors short lengths, as close to each other as they come, direct forms preferred over Bluestein
bool own_long_split(int n, int& A, int& B) {
  if (n < 4 || (n & 1)) return false;
  const int h = n / 2;
  long best = -1;
  for (int a = 2; (long)a * a <= h; ++a) {
    if (h % a) continue;
    const int b = h / a;
    if (!own_short(a) || !own_short(b)) continue;
    OwnAxis xa{}, xb{};
    auto direct = [](int m) { int r = m; for (int f : {13, 11, 7, 5, 3}) for (int c = 0; c < 3 && r % f == 0; ++c) r /= f; return (r & (r - 1)) == 0; };
    const long score = (direct(a) ? 1000000 : 0) + (direct(b) ? 1000000 : 0) + a;  // (a <= sqrt(h): the larger, the squarer)
    if (score > best) { best = score; A = a; B = b; }
  }
  return best >= 0;
}

int own_long_last(int dev, bool inverse, float* real, float2* spec, int64_t lines, int n, hipStream_t s) {
  int A = 0, B = 0;
  if (!own_long_split(n, A, B)) return SKR_ERR_UNSUPPORTED;
  const int h = n / 2;
  OwnAxis xa{}, xb{};
  int rc;
  if ((rc = own_axis(dev, A, s, xa)) != SKR_OK || (rc = own_axis(dev, B, s, xb)) != SKR_OK) return rc;
  float2* z = reinterpret_cast<float2*>(real);  // [lines][h]: the real lines, two values per complex
  const int64_t lines_b = lines * A, lines_a = lines * B, total = lines * h;
  const int La = tile_lines(xa, lines_a), Lb = tile_lines(xb, lines_b);
  if (total > 0x7fffffffffffll) return SKR_ERR_UNSUPPORTED;
  int64_t blocks = (total + 255) / 256; if (blocks > 256 * 64) blocks = 256 * 64;
  if (!inverse) {
    if ((rc = launch_strided<false>((lines_b + Lb - 1) / Lb, xb, Lb, s, z, lines_b, (int64_t)A, xb, Lb, ilog2(Lb))) != SKR_OK) return rc;  // over b, stride A
    hipLaunchKernelGGL(own_long_twiddle, dim3((unsigned)blocks), dim3(256), 0, s, z, total, h, A, 0);
    if ((rc = launch_strided<false>((lines_a + La - 1) / La, xa, La, s, z, lines_a, (int64_t)1, xa, La, ilog2(La))) != SKR_OK) return rc;  // over a, contiguous
    hipLaunchKernelGGL(own_long_post, dim3((unsigned)blocks), dim3(256), 0, s, (const float2*)z, spec, lines, h, A, B);
  } else {
    hipLaunchKernelGGL(own_long_pre, dim3((unsigned)blocks), dim3(256), 0, s, (const float2*)spec, z, lines, h, A, B);
    if ((rc = launch_strided<true>((lines_a + La - 1) / La, xa, La, s, z, lines_a, (int64_t)1, xa, La, ilog2(La))) != SKR_OK) return rc;
    hipLaunchKernelGGL(own_long_twiddle, dim3((unsigned)blocks), dim3(256), 0, s, z, total, h, A, 1);
    if ((rc = launch_strided<true>((lines_b + Lb - 1) / Lb, xb, Lb, s, z, lines_b, (int64_t)A, xb, Lb, ilog2(Lb))) != SKR_OK) return rc;
  }
  return hipGetLastError() == hipSuccess ? SKR_OK : SKR_ERR_LAUNCH;
}

}  // namespace

// can the own transforms take this axis?  `last`: the innermost (real) axis, which may be long (own_long_split)
bool own_length_ok(int n, bool last) {
  int A, B;
  return own_short(n) || (last && own_long_split(n, A, B));
}

// the per-length tables of every axis, built now (own_rfftn / own_outer_weighted then only look them up); SKR_ERR_CAPTURE when a
// table is missing and `s` is capturing
int own_prepare(int dev, int n0, int n1, int n2, hipStream_t s) {
  OwnAxis ax{};
  int A = 0, B = 0;
  if (!own_short(n2) && own_long_split(n2, A, B)) {  // a long last axis: its two factors
    int rc;
    if ((rc = own_axis(dev, A, s, ax)) != SKR_OK || (rc = own_axis(dev, B, s, ax)) != SKR_OK) return rc;
    n2 = 1;
  }
  for (int n : {n2, n1, n0}) {
    if (n <= 1) continue;
    const int rc = own_axis(dev, n, s, ax);
    if (rc != SKR_OK) return rc;
  }
  return SKR_OK;
}

// rfftn / irfftn over the last three axes n0 x n1 x n2 (leading ones may be 1) of `entries` independent units:
// real [entries][n0][n1][n2] fp32  <->  spec [entries][n0][n1][n2/2 + 1] complex64.  SKR_ERR_UNSUPPORTED: an axis beyond the tile
// (see own_short) other than an even last axis that splits into two tile lengths (own_long_last); SKR_ERR_CAPTURE: tables needed during stream capture.
// skip_outer: leave out the outermost transformed axis (n0, or n1 when n0 == 1) -- own_outer_weighted does it both ways in one pass.
int own_rfftn(int dev, bool inverse, float* real, float2* spec, int64_t entries, int n0, int n1, int n2, hipStream_t s, bool skip_outer) {
  OwnAxis a0{}, a1{}, a2{};
  int rc;
  const bool long2 = !own_short(n2);
  if (!long2 && (rc = own_axis(dev, n2, s, a2)) != SKR_OK) return rc;
  if (n1 > 1 && (rc = own_axis(dev, n1, s, a1)) != SKR_OK) return rc;
  if (n0 > 1 && (rc = own_axis(dev, n0, s, a0)) != SKR_OK) return rc;
  const int64_t n2h = n2 / 2 + 1, lines2 = entries * n0 * n1, lines1 = entries * n0 * n2h, lines0 = entries * n1 * n2h;
  const int L2 = long2 ? 1 : tile_lines(a2, (lines2 + 1) / 2), L1 = n1 > 1 ? tile_lines(a1, lines1) : 1, L0 = n0 > 1 ? tile_lines(a0, lines0) : 1;
  const bool do0 = n0 > 1 && !skip_outer, do1 = n1 > 1 && !(skip_outer && n0 == 1);
  if (!inverse) {
    if (long2) { if ((rc = own_long_last(dev, false, real, spec, lines2, n2, s)) != SKR_OK) return rc; }
    else if ((rc = launch_last_forward(((lines2 + 1) / 2 + L2 - 1) / L2, a2, L2, s, (const float*)real, spec, lines2, a2, L2, ilog2(L2))) != SKR_OK) return rc;
    if (do1 && (rc = launch_strided<false>((lines1 + L1 - 1) / L1, a1, L1, s, spec, lines1, n2h, a1, L1, ilog2(L1))) != SKR_OK) return rc;
    if (do0 && (rc = launch_strided<false>((lines0 + L0 - 1) / L0, a0, L0, s, spec, lines0, (int64_t)n1 * n2h, a0, L0, ilog2(L0))) != SKR_OK) return rc;
  } else {
    if (do0 && (rc = launch_strided<true>((lines0 + L0 - 1) / L0, a0, L0, s, spec, lines0, (int64_t)n1 * n2h, a0, L0, ilog2(L0))) != SKR_OK) return rc;
    if (do1 && (rc = launch_strided<true>((lines1 + L1 - 1) / L1, a1, L1, s, spec, lines1, n2h, a1, L1, ilog2(L1))) != SKR_OK) return rc;
    if (long2) { if ((rc = own_long_last(dev, true, real, spec, lines2, n2, s)) != SKR_OK) return rc; }
    else if ((rc = launch_last_inverse(((lines2 + 1) / 2 + L2 - 1) / L2, a2, L2, s, (const float2*)spec, real, lines2, a2, L2, ilog2(L2))) != SKR_OK) return rc;
  }
  return SKR_OK;
}

// forward transform, radial weights (of the full frequency of the n0 x n1 x n2 unit) and inverse transform along the outermost axis, in place;
// needs n1 > 1 (a unit with the last axis alone keeps the separate weights kernel)
int own_outer_weighted(int dev, float2* spec, int64_t entries, int n0, int n1, int n2, float inv_rmax, float eps_clip, float exponent_half_neg, hipStream_t s) {
  if (n1 <= 1) return SKR_ERR_UNSUPPORTED;
  OwnAxis ax{};
  const int n = n0 > 1 ? n0 : n1;
  int rc;
  if ((rc = own_axis(dev, n, s, ax)) != SKR_OK) return rc;
  const int64_t n2h = n2 / 2 + 1, inner = n0 > 1 ? (int64_t)n1 * n2h : n2h, lines = n0 > 1 ? entries * n1 * n2h : entries * n0 * n2h;
  const int L = tile_lines(ax, lines);
  const OwnWeights wp{n0, n1, n2, inv_rmax, eps_clip, exponent_half_neg};
  return launch_strided_weighted((lines + L - 1) / L, ax, L, s, spec, lines, inner, ax, L, ilog2(L), wp);
}

}  // namespace skr
