// The shared LDS transform of the Colored generators: a block owns a tile of L lines x N points (N a power of two) and transforms all
// lines together (skr_colored.hip: the axis-by-axis and plane kernels; skr_fft_own.hip: transforms of any length through Bluestein).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skr_dft.h"

namespace skr {

constexpr int FFT_THREADS = 256;
constexpr int FFT_MAX_TILE = 4096;  // complex points per block tile (32 KiB) + twiddles

__device__ __forceinline__ unsigned brev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// Spectrum / partial-sum accesses of colored_sample, whose blocks exchange data INSIDE a launch: relaxed agent-scope atomics,
// i.e. plain 8-byte loads and stores that go to the coherence point (`sc1`) instead of this XCD's write-back L2 -- what lets
// sample_barrier get by without the agent-scope fences (a full L2 write-back + invalidate each: 4.3 ms instead of 0.35 ms per draw
// when every wave fenced).  COH = false: the ordinary accesses of the three-launch path.
template <bool COH>
__device__ __forceinline__ void gstore(float2* p, float2 v) {
  if constexpr (COH) __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ float2 gload(const float2* p) {
  if constexpr (COH) return __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void gstore(double* p, double v) {
  if constexpr (COH) __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ double gload(const double* p) {
  if constexpr (COH) return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}

// In-place DIT over L lines of N points held in `buf` (line-major, stride N+1), input already in bit-reversed
// order.  `tw` holds exp(-2 pi i k / N), k < N/2; INVERSE conjugates it.  Two radix-2 stages are fused per LDS
// round trip (each work item carries 4 points through stages s and s+1 in registers), with one plain radix-2
// stage left over when log2 N is odd: half the LDS traffic and half the barriers of a stage-by-stage FFT.
template <bool INVERSE>
__device__ __forceinline__ float2 twid(const float2* tw, int idx) {
  float2 w = tw[idx];
  if (INVERSE) w.y = -w.y;
  return w;
}

// SKIP8: the caller has already applied stages 0-2 (it produced the tile through 8-point transforms in registers: the plane
// kernels fuse the transposing steps between row and column tiles with the first pass of the transform that follows)
// TO_GLOBAL: the LAST pass sends its (natural-order) results to global memory instead of back to the tile -- element n of
// line l goes to gout[n * gpitch + l] (the forward plane kernel's half spectrum: lines are frequency columns, so with
// consecutive lanes on consecutive lines every store instruction covers a run of a spectrum row); the tile is dead afterwards.
template <bool INVERSE, bool SKIP8 = false, bool TO_GLOBAL = false, bool COH = false>
__device__ __forceinline__ void fft_tile(float2* buf, const float2* tw, int N, int logN, int L, float2* gout = nullptr, int gpitch = 0, bool few_lines_by_line = false) {
  const int half_n = N >> 1, ld = N + 1;
  // Work-item -> (line, k) mapping.  Early stages (butterfly span h < 32) touch points 4h apart, which lands
  // consecutive k on the same LDS banks; there consecutive lanes take consecutive LINES instead (line stride
  // N+1 complex is odd, so 32 lanes cover all 64 banks and share one twiddle).  t / L by multiply-high
  // (exact for t, L < 2^16).
  // (few_lines_by_line, skr_fft_own.hip: also for 2 ... 16 lines -- L lines x 64 / L neighbouring items per wave still spread an 8-byte
  //  access over 32 bank pairs, where 64 neighbouring items of ONE line sit 4h points apart on 8 of them)
  const bool by_line = L >= 32 || (few_lines_by_line && L >= 2);
  const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)L - 1) / (uint32_t)L);
  int s = 0;
  if constexpr (SKIP8) {
    s = 3;
  } else if ((logN & 1) && logN >= 3) {
    // odd log2 N (128-point lines: every BASELINE shape): stages 0-2 in ONE LDS round trip -- each item takes 8 consecutive
    // points (they hold an 8-point subsequence in 3-bit-reversed order), transforms them in registers and puts them back.
    // Round 2 spent two round trips on these stages (a bare radix-2 pass, 23 instructions per butterfly, then a radix-2^2 pass):
    // 432 instructions per thread and transform against ~150 here, and one barrier + one full LDS write of the tile fewer.
    // Consecutive lanes take consecutive LINES (pitch N+1 complex = 2 banks mod 64: 8-byte accesses of 32 lanes tile all banks).
    __syncthreads();
    const int groups = N >> 3, total = L * groups;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      int line, g;
      if (by_line) { g = (int)__umulhi((uint32_t)t, magic); line = t - g * L; }
      else { line = t >> (logN - 3); g = t & (groups - 1); }
      float2* p = buf + line * ld + 8 * g;
      float2 v[8];
      v[0] = p[0]; v[4] = p[1]; v[2] = p[2]; v[6] = p[3]; v[1] = p[4]; v[5] = p[5]; v[3] = p[6]; v[7] = p[7];
      dft8<INVERSE>(v);
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = v[k];
    }
    s = 3;
  } else if (logN & 1) {  // 2-point lines: the single radix-2 stage (half = 1, twiddle = 1)
    __syncthreads();
    const int total = L * half_n;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      int line, k;
      if (by_line) { k = (int)__umulhi((uint32_t)t, magic); line = t - k * L; }
      else { line = t >> (logN - 1); k = t & (half_n - 1); }
      float2* p = buf + line * ld + 2 * k;
      const float2 a = p[0], b = p[1];
      p[0] = make_float2(a.x + b.x, a.y + b.y);
      p[1] = make_float2(a.x - b.x, a.y - b.y);
    }
    s = 1;
  }
  const int quarter = N >> 2;
  for (; s < logN; s += 2) {
    const int h = 1 << s;
    const int step2 = half_n >> (s + 1);
    __syncthreads();
    const int total = L * quarter;
    const bool last_out = TO_GLOBAL && s + 2 == logN;
    const bool line_major = by_line && (h < 32 || last_out);
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      int line, k;
      if (line_major) { k = (int)__umulhi((uint32_t)t, magic); line = t - k * L; }
      else { line = t >> (logN - 2); k = t & (quarter - 1); }
      const int pos = k & (h - 1);
      float2* p = buf + line * ld + ((k >> s) << (s + 2)) + pos;
      float2 e0 = p[0], e1 = p[h], e2 = p[2 * h], e3 = p[3 * h];
      // ONE twiddle load per item (round 4; three before): with w2 = exp(-+2 pi i pos / 4h) the stage-s twiddle is w1 = w2^2
      // (its index pos * step1 is twice pos * step2) and the second stage-(s+1) twiddle is w3 = -+i w2 (index + N/4): a
      // three-instruction square and a swap replace two LDS reads and their address arithmetic
      const float2 w2 = twid<INVERSE>(tw, pos * step2);
      const float2 w1 = make_float2(__builtin_fmaf(w2.x, w2.x, -(w2.y * w2.y)), 2.f * w2.x * w2.y);
      // stage s: (e0,e1) and (e2,e3), same twiddle
      const float2 b1 = cmul(e1, w1), b3 = cmul(e3, w1);
      const float2 f0 = make_float2(e0.x + b1.x, e0.y + b1.y), f1 = make_float2(e0.x - b1.x, e0.y - b1.y);
      const float2 f2 = make_float2(e2.x + b3.x, e2.y + b3.y), f3 = make_float2(e2.x - b3.x, e2.y - b3.y);
      // stage s+1: (f0,f2) at position pos, (f1,f3) at position pos + h
      const float2 c2 = cmul(f2, w2), c3 = mul_i<INVERSE>(cmul(f3, w2));
      if (TO_GLOBAL && last_out) {
        float2* g = gout + (int64_t)pos * gpitch + line;  // (the last pass has one block of 4h = N points per line: position = pos + i h)
        gstore<COH>(g, make_float2(f0.x + c2.x, f0.y + c2.y));
        gstore<COH>(g + (int64_t)2 * h * gpitch, make_float2(f0.x - c2.x, f0.y - c2.y));
        gstore<COH>(g + (int64_t)h * gpitch, make_float2(f1.x + c3.x, f1.y + c3.y));
        gstore<COH>(g + (int64_t)3 * h * gpitch, make_float2(f1.x - c3.x, f1.y - c3.y));
      } else {
        p[0] = make_float2(f0.x + c2.x, f0.y + c2.y);
        p[2 * h] = make_float2(f0.x - c2.x, f0.y - c2.y);
        p[h] = make_float2(f1.x + c3.x, f1.y + c3.y);
        p[3 * h] = make_float2(f1.x - c3.x, f1.y - c3.y);
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void make_twiddles(float2* tw, int N) {
  for (int k = threadIdx.x; k < N / 2; k += blockDim.x) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)N, &s, &c);
    tw[k] = make_float2(c, s);
  }
}

}  // namespace skr
