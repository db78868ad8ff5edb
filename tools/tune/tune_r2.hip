// Round-2 experiment harness for the headline launch (DPM-2 SDE steady state: 4 bf16 reads + Philox + 1 bf16 write,
// B=256x4x128x128).  Every variant is its own template instantiation so `rocprofv3 --kernel-trace --stats` separates them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tune_r2 tune_r2.hip -ldl
//   ./tune_r2 [place=slab|torch] [gap_us=0] [reps=3] [only=substring]
// Variants: block size, vectors per lane, chunk->workgroup map (XCD aware), wave priority while the loads are issued,
// barrier before the stores, persistent grid with register prefetch.  All variants must produce the same bytes
// (checked against variant 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <dlfcn.h>
#include "../../include/skrample_hip.h"
#include "../../skrample_amd/csrc/skr_philox.h"

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

struct Args { const u32x4_t* in[4]; u32x4_t* out; const uint64_t* seeds; float c[4]; float zeta; uint64_t stream; int64_t nvec; int vps_shift; };

__device__ __forceinline__ u32x4_t ldg(const u32x4_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stg(u32x4_t* p, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// chunk -> workgroup maps.  Blocks b and b+8 share an XCD (round-robin dispatch).
//   0 linear                     : chunk = block
//   1 XCD owns a contiguous 1/8  : chunk = (b%8)*(G/8) + b/8
//   2 XCD owns runs of 8 chunks  : chunk = (b/64)*64 + (b%8)*8 + (b/8)%8
template <int MAP> __device__ __forceinline__ int64_t chunk_of(int64_t b, int64_t G) {
  if constexpr (MAP == 1) return (b & 7) * (G >> 3) + (b >> 3);
  else if constexpr (MAP == 2) return ((b >> 6) << 6) + ((b & 7) << 3) + ((b >> 3) & 7);
  else if constexpr (MAP >= 100) {  // XCD owns runs of R = 2^(MAP-100) chunks inside groups of 8R
    constexpr int LR = MAP - 100;
    return ((b >> (3 + LR)) << (3 + LR)) + ((b & 7) << LR) + ((b >> 3) & ((1 << LR) - 1));
  }
  else return b;
}

__device__ __forceinline__ void combine_store(const Args& a, const u32x4_t r[4], const float z[8], int64_t v) {
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(r[j][i] << 16), s[2 * i]);
      s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(r[j][i] & 0xFFFF0000u), s[2 * i + 1]);
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[i], s[i]);
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
  stg(a.out + v, q);
}

// one trip: each workgroup owns BLK*UV consecutive vectors
template <int BLK, int UV, int MAP, int PRIO, int SYNC, bool NOISE>
__global__ __launch_bounds__(BLK) void kv(const Args a) {
  if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
  const int64_t c = chunk_of<MAP>(blockIdx.x, gridDim.x);
  const int64_t base = c * (BLK * UV);
  const int64_t v0 = base + threadIdx.x;
  u32x4_t r[UV][4];
#pragma unroll
  for (int u = 0; u < UV; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) r[u][j] = ldg(a.in[j] + v0 + u * BLK);
  if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  if constexpr (SYNC == 2) __builtin_amdgcn_sched_barrier(0);
  float z[UV][8];
  if constexpr (NOISE) {
    const int64_t smp = base >> a.vps_shift;  // BLK*UV divides a sample
    const uint64_t seed = a.seeds[smp];
    const int64_t vlo = smp << a.vps_shift;
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const uint64_t blk = (uint64_t)(v0 + u * BLK - vlo) * 2;
      skr::normal4(seed, a.stream, blk, z[u]);
      skr::normal4(seed, a.stream, blk + 1, z[u] + 4);
    }
  } else {
#pragma unroll
    for (int u = 0; u < UV; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) z[u][i] = 0.f;
  }
  if constexpr (SYNC) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < UV; ++u) combine_store(a, r[u], z[u], v0 + u * BLK);
}

// persistent: gridDim.x workgroups walk the chunks with stride gridDim.x, the next chunk's loads are issued before
// the current chunk's Philox + FMAs
template <int BLK, int MAP, bool NOISE>
__global__ __launch_bounds__(BLK) void kp(const Args a, int64_t nchunks) {
  int64_t c = blockIdx.x;
  if (c >= nchunks) return;
  u32x4_t cur[4], nxt[4];
  {
    const int64_t v = chunk_of<MAP>(c, nchunks) * BLK + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = ldg(a.in[j] + v);
  }
  for (;;) {
    const int64_t cn = c + gridDim.x;
    const bool more = cn < nchunks;
    if (more) {
      const int64_t vn = chunk_of<MAP>(cn, nchunks) * BLK + threadIdx.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) nxt[j] = ldg(a.in[j] + vn);
    }
    const int64_t base = chunk_of<MAP>(c, nchunks) * BLK;
    const int64_t v = base + threadIdx.x;
    float z[8];
    if constexpr (NOISE) {
      const int64_t smp = base >> a.vps_shift;
      const uint64_t seed = a.seeds[smp];
      const uint64_t blk = (uint64_t)(v - (smp << a.vps_shift)) * 2;
      skr::normal4(seed, a.stream, blk, z);
      skr::normal4(seed, a.stream, blk + 1, z + 4);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = 0.f;
    }
    combine_store(a, cur, z, v);
    if (!more) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    c = cn;
  }
}


// persistent, second attempt: the next chunk's seed (scalar) and loads are issued before the current chunk's Philox;
// chunk ids are wave-uniform by construction (readfirstlane keeps them in SGPRs)
__device__ __forceinline__ void stg_nc(u32x4_t* p, u32x4_t v) {  // no "memory" clobber: nothing here reads `out`
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v));
}
template <bool NOISE>
__device__ __forceinline__ void finish_chunk(const Args& a, const u32x4_t cur[4], uint64_t seed, int64_t base, u32x4_t* __restrict__ out) {
  const int64_t v = base + threadIdx.x;
  float z[8];
  if constexpr (NOISE) {
    const uint64_t blk = (uint64_t)(v - ((base >> a.vps_shift) << a.vps_shift)) * 2;
    skr::normal4(seed, a.stream, blk, z);
    skr::normal4(seed, a.stream, blk + 1, z + 4);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = 0.f;
  }
  float s8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s8[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s8[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(cur[j][i] << 16), s8[2 * i]);
      s8[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(cur[j][i] & 0xFFFF0000u), s8[2 * i + 1]);
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) s8[i] = __builtin_fmaf(a.zeta, z[i], s8[i]);
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s8[2 * i], s8[2 * i + 1]);
  stg_nc(out + v, q);
}

// two register buffers, loop unrolled by two so no buffer is ever copied (a copy would wait for the prefetch)
template <int BLK, int MAP, bool NOISE>
__global__ __launch_bounds__(BLK) void kp2(const Args a, int nchunks, const uint64_t* __restrict__ seeds, u32x4_t* __restrict__ out) {
  int c = blockIdx.x;
  if (c >= nchunks) return;
  const int G = gridDim.x;
  u32x4_t A[4], B[4];
  int64_t baseA = chunk_of<MAP>(c, nchunks) * BLK, baseB = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) A[j] = ldg(a.in[j] + baseA + threadIdx.x);
  uint64_t seedA = seeds[baseA >> a.vps_shift], seedB = 0;
  for (;;) {
    bool more = c + G < nchunks;
    if (more) {
      baseB = chunk_of<MAP>(c + G, nchunks) * BLK;
#pragma unroll
      for (int j = 0; j < 4; ++j) B[j] = ldg(a.in[j] + baseB + threadIdx.x);
      seedB = seeds[baseB >> a.vps_shift];
    }
    __builtin_amdgcn_sched_barrier(0);
    finish_chunk<NOISE>(a, A, seedA, baseA, out);
    if (!more) break;
    c += G;
    more = c + G < nchunks;
    if (more) {
      baseA = chunk_of<MAP>(c + G, nchunks) * BLK;
#pragma unroll
      for (int j = 0; j < 4; ++j) A[j] = ldg(a.in[j] + baseA + threadIdx.x);
      seedA = seeds[baseA >> a.vps_shift];
    }
    __builtin_amdgcn_sched_barrier(0);
    finish_chunk<NOISE>(a, B, seedB, baseB, out);
    if (!more) break;
    c += G;
  }
}

// persistent, explicit waits: loads are inline asm (invisible to the compiler's vmcnt bookkeeping), each phase waits
// for exactly its own four loads (vmcnt(4) while a prefetch is behind them, vmcnt(0) on the last chunk)
__device__ __forceinline__ void ldg4_asm(u32x4_t r[4], const Args& a, int64_t v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[j]) : "v"(a.in[j] + v));
}
__device__ __forceinline__ void pin8(float z[8]) {  // the normals exist before the wait (volatile asms keep their order)
  asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]));
}
template <int N> __device__ __forceinline__ void wait_asm(u32x4_t r[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
}
template <int BLK, int MAP, bool NOISE>
__global__ __launch_bounds__(BLK) void kp3(const Args a, int nchunks, const uint64_t* __restrict__ seeds, u32x4_t* __restrict__ out) {
  int c = blockIdx.x;
  if (c >= nchunks) return;
  const int G = gridDim.x;
  u32x4_t A[4], B[4];
  int64_t baseA = chunk_of<MAP>(c, nchunks) * BLK, baseB = 0;
  ldg4_asm(A, a, baseA + threadIdx.x);
  uint64_t seedA = seeds[baseA >> a.vps_shift], seedB = 0;
  for (;;) {
    bool more = c + G < nchunks;
    if (more) {
      baseB = chunk_of<MAP>(c + G, nchunks) * BLK;
      ldg4_asm(B, a, baseB + threadIdx.x);
      seedB = seeds[baseB >> a.vps_shift];
      __builtin_amdgcn_sched_barrier(0);
      // Philox first (inside finish_chunk the z values are computed before the data is touched)
    }
    {
      const int64_t v = baseA + threadIdx.x;
      float z[8];
      const uint64_t blk = (uint64_t)(v - ((baseA >> a.vps_shift) << a.vps_shift)) * 2;
      skr::normal4(seedA, a.stream, blk, z);
      skr::normal4(seedA, a.stream, blk + 1, z + 4);
      pin8(z);
      if (more) wait_asm<4>(A); else wait_asm<0>(A);
      combine_store(a, A, z, v);
    }
    if (!more) break;
    c += G;
    more = c + G < nchunks;
    if (more) {
      baseA = chunk_of<MAP>(c + G, nchunks) * BLK;
      ldg4_asm(A, a, baseA + threadIdx.x);
      seedA = seeds[baseA >> a.vps_shift];
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const int64_t v = baseB + threadIdx.x;
      float z[8];
      const uint64_t blk = (uint64_t)(v - ((baseB >> a.vps_shift) << a.vps_shift)) * 2;
      skr::normal4(seedB, a.stream, blk, z);
      skr::normal4(seedB, a.stream, blk + 1, z + 4);
      pin8(z);
      if (more) wait_asm<4>(B); else wait_asm<0>(B);
      combine_store(a, B, z, v);
    }
    if (!more) break;
    c += G;
  }
}

struct Opts { int gap_us = 0; int reps = 3; std::string only; int iters = 300; };
static Opts g;
static std::vector<uint16_t> g_ref;
static int64_t g_n;

static void spin_us(int us) {
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() < us * 1000ll) {}
}

template <typename F>
static void bench(const char* name, std::vector<Args>& sets, F launch_one) {
  if (!g.only.empty() && !strstr(name, g.only.c_str())) return;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // correctness: run on set 0, compare with the reference bytes (first variant run defines them)
  launch_one(sets[0]);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> got(g_n);
  CK(hipMemcpy(got.data(), sets[0].out, g_n * 2, hipMemcpyDeviceToHost));
  const char* verdict = "ref";
  if (g_ref.empty()) g_ref = got;
  else verdict = memcmp(g_ref.data(), got.data(), g_n * 2) == 0 ? "same" : "DIFFERENT";
  double best = 1e9, sum = 0;
  for (int rep = 0; rep < g.reps; ++rep) {
    for (int i = 0; i < 12; ++i) launch_one(sets[i % sets.size()]);
    CK(hipDeviceSynchronize());
    if (g.gap_us == 0) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < g.iters; ++i) launch_one(sets[i % sets.size()]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / g.iters; sum += us; if (us < best) best = us;
    } else {  // queue-empty cadence: each launch individually bracketed by events
      double tot = 0;
      for (int i = 0; i < g.iters; ++i) {
        CK(hipEventRecord(e0));
        launch_one(sets[i % sets.size()]);
        CK(hipEventRecord(e1));
        spin_us(g.gap_us);
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms * 1e3;
      }
      const double us = tot / g.iters; sum += us; if (us < best) best = us;
    }
  }
  const double bytes = (double)g_n * 10;
  printf("%-44s avg %7.2f us  best %7.2f us  %6.3f TB/s  frac8=%.3f  [%s]\n", name, sum / g.reps, best, bytes / best / 1e6, bytes / best / 1e6 / 8.0, verdict);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}


// issue-order experiments (blk256, one trip, R128 map): STAG
//   0 compiler's order   1 all four loads, barrier, Philox   2 two loads | key set-up + first Philox block | two loads
//   3 one load between each Philox half (4 segments)   4 Philox entirely first, then the loads
template <int STAG>
__global__ __launch_bounds__(256) void ks(const Args a) {
  const int64_t c = chunk_of<107>(blockIdx.x, gridDim.x);
  const int64_t base = c * 256;
  const int64_t v0 = base + threadIdx.x;
  const int64_t smp = base >> a.vps_shift;
  u32x4_t r[4];
  float z[8];
  auto seg = [&]() { __builtin_amdgcn_sched_barrier(0); };
  if constexpr (STAG == 0 || STAG == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ldg(a.in[j] + v0);
    if constexpr (STAG == 1) seg();
    const uint64_t seed = a.seeds[smp];
    const uint64_t blk = (uint64_t)(v0 - (smp << a.vps_shift)) * 2;
    skr::normal4(seed, a.stream, blk, z);
    skr::normal4(seed, a.stream, blk + 1, z + 4);
  } else if constexpr (STAG == 2) {
    r[0] = ldg(a.in[0] + v0); r[1] = ldg(a.in[1] + v0);
    seg();
    const uint64_t seed = a.seeds[smp];
    const uint64_t blk = (uint64_t)(v0 - (smp << a.vps_shift)) * 2;
    skr::normal4(seed, a.stream, blk, z);
    seg();
    r[2] = ldg(a.in[2] + v0); r[3] = ldg(a.in[3] + v0);
    seg();
    skr::normal4(seed, a.stream, blk + 1, z + 4);
  } else if constexpr (STAG == 3) {
    r[0] = ldg(a.in[0] + v0);
    seg();
    const uint64_t seed = a.seeds[smp];
    const uint64_t blk = (uint64_t)(v0 - (smp << a.vps_shift)) * 2;
    r[1] = ldg(a.in[1] + v0);
    seg();
    skr::normal4(seed, a.stream, blk, z);
    seg();
    r[2] = ldg(a.in[2] + v0);
    seg();
    skr::normal4(seed, a.stream, blk + 1, z + 4);
    seg();
    r[3] = ldg(a.in[3] + v0);
  } else {
    const uint64_t seed = a.seeds[smp];
    const uint64_t blk = (uint64_t)(v0 - (smp << a.vps_shift)) * 2;
    skr::normal4(seed, a.stream, blk, z);
    skr::normal4(seed, a.stream, blk + 1, z + 4);
    seg();
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ldg(a.in[j] + v0);
  }
  combine_store(a, r, z, v0);
}
template <int STAG>
static void run_ks(const char* name, std::vector<Args>& sets) {
  const unsigned grid = (unsigned)(sets[0].nvec / 256);
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((ks<STAG>), dim3(grid), dim3(256), 0, 0, a); });
}

// stream-major issue: a workgroup owns UV*256 consecutive vectors; it reads operand 0 for all of them, then operand 1, ...
// with the Philox blocks of one vector between the operand phases, and accumulates an operand as soon as the next one has
// been requested.  MAPLR = log2(run length) of the XCD map in units of workgroups.
template <int UV, int MAPLR, int ACCLAG>
__global__ __launch_bounds__(256) void ku(const Args a) {
  const int64_t c = chunk_of<100 + MAPLR>(blockIdx.x, gridDim.x);
  const int64_t base = c * (256 * UV);
  const int64_t v0 = base + threadIdx.x;
  const int64_t smp = base >> a.vps_shift;
  auto seg = [&]() { __builtin_amdgcn_sched_barrier(0); };
  u32x4_t r[4][UV];
  float z[UV][8], s[UV][8];
#pragma unroll
  for (int u = 0; u < UV; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) s[u][i] = 0.f;
  auto acc = [&](int j) {
#pragma unroll
    for (int u = 0; u < UV; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[u][2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(r[j][u][i] << 16), s[u][2 * i]);
        s[u][2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(r[j][u][i] & 0xFFFF0000u), s[u][2 * i + 1]);
      }
  };
#pragma unroll
  for (int u = 0; u < UV; ++u) r[0][u] = ldg(a.in[0] + v0 + u * 256);
  seg();
  const uint64_t seed = a.seeds[smp];
  const int64_t vlo = smp << a.vps_shift;
  // UV Philox jobs (two blocks each) spread over the 3 gaps between the 4 operand phases (+ the tail)
  int job = 0;
  auto philox_jobs = [&](int upto) {
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      if (u >= job && u < upto) {
        const uint64_t blk = (uint64_t)(v0 + u * 256 - vlo) * 2;
        skr::normal4(seed, a.stream, blk, z[u]);
        skr::normal4(seed, a.stream, blk + 1, z[u] + 4);
      }
    }
    job = upto;
  };
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    philox_jobs((UV * j + 3) / 4);
    seg();
#pragma unroll
    for (int u = 0; u < UV; ++u) r[j][u] = ldg(a.in[j] + v0 + u * 256);
    seg();
    if (ACCLAG && j >= 2) { acc(j - 2); seg(); }
  }
  philox_jobs(UV);
  seg();
#pragma unroll
  for (int j = ACCLAG ? 2 : 0; j < 4; ++j) acc(j);
#pragma unroll
  for (int u = 0; u < UV; ++u) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[u][i] = __builtin_fmaf(a.zeta, z[u][i], s[u][i]);
    u32x4_t q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[u][2 * i], s[u][2 * i + 1]);
    stg(a.out + v0 + u * 256, q);
  }
}
template <int UV, int MAPLR, int ACCLAG>
static void run_ku(const char* name, std::vector<Args>& sets) {
  const unsigned grid = (unsigned)(sets[0].nvec / (256 * UV));
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((ku<UV, MAPLR, ACCLAG>), dim3(grid), dim3(256), 0, 0, a); });
}

// no-noise pacing: SLEEP x 64 clocks (s_sleep) between the four loads; finer noise stagger variants
template <int SLEEP>
__global__ __launch_bounds__(256) void kn(const Args a) {
  const int64_t c = chunk_of<107>(blockIdx.x, gridDim.x);
  const int64_t v0 = c * 256 + threadIdx.x;
  u32x4_t r[4];
  float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = ldg(a.in[j] + v0);
    if (SLEEP > 0 && j < 3) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(SLEEP); __builtin_amdgcn_sched_barrier(0); }
  }
  combine_store(a, r, z, v0);
}
template <int SLEEP>
static void run_kn(const char* name, std::vector<Args>& sets, int lds_bytes = 0) {
  const unsigned grid = (unsigned)(sets[0].nvec / 256);
  if (lds_bytes > 65536) CK(hipFuncSetAttribute((const void*)kn<SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kn<SLEEP>), dim3(grid), dim3(256), lds_bytes, 0, a); });
}
// noise kernel, stag3 order, extra s_sleep after each load
template <int SLEEP>
__global__ __launch_bounds__(256) void kss(const Args a) {
  const int64_t c = chunk_of<107>(blockIdx.x, gridDim.x);
  const int64_t base = c * 256;
  const int64_t v0 = base + threadIdx.x;
  const int64_t smp = base >> a.vps_shift;
  u32x4_t r[4];
  float z[8];
  auto seg = [&]() { __builtin_amdgcn_sched_barrier(0); if (SLEEP) { __builtin_amdgcn_s_sleep(SLEEP); __builtin_amdgcn_sched_barrier(0); } };
  r[0] = ldg(a.in[0] + v0);
  seg();
  const uint64_t seed = a.seeds[smp];
  const uint64_t blk = (uint64_t)(v0 - (smp << a.vps_shift)) * 2;
  r[1] = ldg(a.in[1] + v0);
  seg();
  skr::normal4(seed, a.stream, blk, z);
  __builtin_amdgcn_sched_barrier(0);
  r[2] = ldg(a.in[2] + v0);
  seg();
  skr::normal4(seed, a.stream, blk + 1, z + 4);
  __builtin_amdgcn_sched_barrier(0);
  r[3] = ldg(a.in[3] + v0);
  combine_store(a, r, z, v0);
}
template <int SLEEP>
static void run_kss(const char* name, std::vector<Args>& sets, int lds_bytes = 0) {
  const unsigned grid = (unsigned)(sets[0].nvec / 256);
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kss<SLEEP>), dim3(grid), dim3(256), lds_bytes, 0, a); });
}

// other operand counts, no noise: NIN inputs (re-using the 4 input buffers of a set and of the next set), NOUT outputs
template <int NIN, int NOUT, int SLEEP, int SLEEP_ST>
__global__ __launch_bounds__(256) void kq(const Args a, const Args b) {
  const int64_t c = chunk_of<107>(blockIdx.x, gridDim.x);
  const int64_t v0 = c * 256 + threadIdx.x;
  u32x4_t r[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    r[j] = ldg((j < 4 ? a.in[j] : b.in[j - 4]) + v0);
    if (SLEEP > 0 && j < NIN - 1) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(SLEEP); __builtin_amdgcn_sched_barrier(0); }
  }
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
  for (int j = 0; j < NIN; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] = __builtin_fmaf(a.c[j & 3], __uint_as_float(r[j][i] << 16), s[2 * i]);
      s[2 * i + 1] = __builtin_fmaf(a.c[j & 3], __uint_as_float(r[j][i] & 0xFFFF0000u), s[2 * i + 1]);
    }
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
  stg(a.out + v0, q);
  if constexpr (NOUT == 2) {
    if (SLEEP_ST > 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(SLEEP_ST); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i] * 0.5f, s[2 * i + 1] * 0.5f);
    stg(b.out + v0, q);
  }
}
template <int NIN, int NOUT, int SLEEP, int SLEEP_ST>
static void run_kq(const char* name, std::vector<Args>& sets) {
  const unsigned grid = (unsigned)(sets[0].nvec / 256);
  if (!g.only.empty() && !strstr(name, g.only.c_str())) return;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e9, sum = 0;
  const int NS = (int)sets.size();
  auto go = [&](int i) { hipLaunchKernelGGL((kq<NIN, NOUT, SLEEP, SLEEP_ST>), dim3(grid), dim3(256), 0, 0, sets[i % NS], sets[(i + 1) % NS]); };
  for (int rep = 0; rep < g.reps; ++rep) {
    for (int i = 0; i < 12; ++i) go(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < g.iters; ++i) go(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / g.iters; sum += us; if (us < best) best = us;
  }
  const double bytes = (double)g_n * 2 * (NIN + NOUT);
  printf("%-44s avg %7.2f us  best %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, sum / g.reps, best, bytes / best / 1e6, bytes / best / 1e6 / 8.0);
  fflush(stdout);
}

// few operands: UV vectors per lane (consecutive 4 KiB pieces of one workgroup-owned span), BLK threads
template <int NIN, int UV, int BLK, int MAPLR>
__global__ __launch_bounds__(BLK) void kw(const Args a) {
  const int64_t c = chunk_of<100 + MAPLR>(blockIdx.x, gridDim.x);
  const int64_t v0 = c * (BLK * UV) + threadIdx.x;
  u32x4_t r[UV][NIN];
#pragma unroll
  for (int u = 0; u < UV; ++u)
#pragma unroll
    for (int j = 0; j < NIN; ++j) r[u][j] = ldg(a.in[j] + v0 + u * BLK);
#pragma unroll
  for (int u = 0; u < UV; ++u) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NIN; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(r[u][j][i] << 16), s[2 * i]);
        s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(r[u][j][i] & 0xFFFF0000u), s[2 * i + 1]);
      }
    u32x4_t q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
    stg(a.out + v0 + u * BLK, q);
  }
}
template <int NIN, int UV, int BLK, int MAPLR>
static void run_kw(const char* name, std::vector<Args>& sets) {
  const unsigned grid = (unsigned)(sets[0].nvec / (BLK * UV));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e9, sum = 0;
  const int NS = (int)sets.size();
  auto go = [&](int i) { hipLaunchKernelGGL((kw<NIN, UV, BLK, MAPLR>), dim3(grid), dim3(BLK), 0, 0, sets[i % NS]); };
  for (int rep = 0; rep < g.reps; ++rep) {
    for (int i = 0; i < 12; ++i) go(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < g.iters; ++i) go(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / g.iters; sum += us; if (us < best) best = us;
  }
  const double bytes = (double)g_n * 2 * (NIN + 1);
  printf("%-44s avg %7.2f us  best %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, sum / g.reps, best, bytes / best / 1e6, bytes / best / 1e6 / 8.0);
  fflush(stdout);
}

template <int BLK, int UV, int MAP, int PRIO, int SYNC, bool NOISE>
static void run_kv(const char* name, std::vector<Args>& sets) {
  const unsigned grid = (unsigned)(sets[0].nvec / (BLK * UV));
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kv<BLK, UV, MAP, PRIO, SYNC, NOISE>), dim3(grid), dim3(BLK), 0, 0, a); });
}
template <int BLK, int MAP, bool NOISE>
static void run_kp2(const char* name, std::vector<Args>& sets, int blocks_per_cu) {
  const int nchunks = (int)(sets[0].nvec / BLK);
  const unsigned grid = 256u * blocks_per_cu;
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kp2<BLK, MAP, NOISE>), dim3(grid), dim3(BLK), 0, 0, a, nchunks, a.seeds, a.out); });
}
template <int BLK, int MAP, bool NOISE>
static void run_kp3(const char* name, std::vector<Args>& sets, int blocks_per_cu) {
  const int nchunks = (int)(sets[0].nvec / BLK);
  const unsigned grid = 256u * blocks_per_cu;
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kp3<BLK, MAP, NOISE>), dim3(grid), dim3(BLK), 0, 0, a, nchunks, a.seeds, a.out); });
}
template <int BLK, int MAP, bool NOISE>
static void run_kp(const char* name, std::vector<Args>& sets, int blocks_per_cu) {
  const int64_t nchunks = sets[0].nvec / BLK;
  const unsigned grid = 256u * blocks_per_cu;
  bench(name, sets, [&](const Args& a) { hipLaunchKernelGGL((kp<BLK, MAP, NOISE>), dim3(grid), dim3(BLK), 0, 0, a, nchunks); });
}

int main(int argc, char** argv) {
  std::string place = "slab";
  std::vector<int64_t> off_kib = {-1, -1, -1};
  bool sweep = false;  // x, x_prev, y offsets for place=custom
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s.rfind("place=", 0) == 0) place = s.substr(6);
    else if (s.rfind("gap_us=", 0) == 0) g.gap_us = atoi(s.c_str() + 7);
    else if (s.rfind("reps=", 0) == 0) g.reps = atoi(s.c_str() + 5);
    else if (s.rfind("only=", 0) == 0) g.only = s.substr(5);
    else if (s.rfind("iters=", 0) == 0) g.iters = atoi(s.c_str() + 6);
    else if (s == "sweep=1") sweep = true;
    else if (s.rfind("off=", 0) == 0) sscanf(s.c_str() + 4, "%ld,%ld,%ld", &off_kib[0], &off_kib[1], &off_kib[2]);
  }
  const int B = 256;
  const int64_t sample = 4 * 128 * 128, n = (int64_t)B * sample;
  g_n = n;
  const int NS = 6;
  std::vector<Args> sets(NS);
  uint64_t* seeds; CK(hipMalloc(&seeds, B * 8));
  std::vector<uint64_t> hs(B); for (int i = 0; i < B; ++i) hs[i] = 42 + i;
  CK(hipMemcpy(seeds, hs.data(), B * 8, hipMemcpyHostToDevice));
  std::vector<uint16_t> host(n);
  srand(7);
  for (int64_t i = 0; i < n; ++i) host[i] = 0x3f80 + (rand() & 0x7f);
  // roles: 0 x, 1 out, 2 x_prev, 3 out_prev (inputs in kernel order), 4 y
  //   slab : pitch 36 MiB, offsets 4 KiB * (2j+1)                     (round-1 harness layout)
  //   torch: every buffer at a 2 MiB multiple (pitch 34 MiB); x / x_prev / y shifted by 4 KiB + k*8 KiB, k cycling as
  //          lazy.empty_output does (y = slot k, x = k-1, x_prev = k-2); out / out_prev unshifted
  //   custom: torch bases, x / x_prev / y shifted by off= KiB
  const int64_t pitch = place == "slab" ? (36ll << 20) : (34ll << 20);
  char* slab0; CK(hipMalloc((void**)&slab0, pitch * 5 * NS + (66ll << 20)));
  char* slab = (char*)(((uintptr_t)slab0 + (2ll << 20) - 1) & ~(uintptr_t)((2ll << 20) - 1));
  auto fill = [&](void* p) { CK(hipMemcpy(p, host.data(), n * 2, hipMemcpyHostToDevice)); };
  auto place_sets = [&](const std::string& mode, const int64_t* offk, bool init) {
    for (int s = 0; s < NS; ++s) {
      void* p[5];
      for (int j = 0; j < 5; ++j) {
        int64_t off;
        if (mode == "slab") off = 4096 * (2 * j + 1);
        else if (mode == "torch") { const int k = (3 * s + (j == 4 ? 2 : j == 0 ? 1 : 0)) % 8; off = (j == 1 || j == 3) ? 0 : 4096 + k * 8192; }
        else off = (j == 1 || j == 3) ? 0 : offk[j == 0 ? 0 : j == 2 ? 1 : 2] * 1024;
        p[j] = slab + pitch * (s * 5 + j) + off;
      }
      for (int j = 0; j < 4; ++j) { if (init) fill(p[j]); sets[s].in[j] = (const u32x4_t*)p[j]; }
      sets[s].out = (u32x4_t*)p[4];
      sets[s].seeds = seeds; sets[s].c[0] = 1.01f; sets[s].c[1] = -0.53f; sets[s].c[2] = 0.12f; sets[s].c[3] = 0.43f;
      sets[s].zeta = 0.3f; sets[s].stream = 1; sets[s].nvec = n / 8; sets[s].vps_shift = 13;  // 8192 vectors per sample
    }
  };
  place_sets(place, off_kib.data(), true);
  printf("== place=%s gap_us=%d reps=%d iters=%d\n", place.c_str(), g.gap_us, g.reps, g.iters);

  void* h = dlopen("skrample_amd/csrc/libskrample_hip.so", RTLD_NOW);
  typedef int (*launch_fn)(const skr_step_plan*, const void* const*, void*, void*, const uint64_t*, int64_t, void*);
  launch_fn launch = h ? (launch_fn)dlsym(h, "skr_step_launch") : nullptr;
  typedef int (*tune_fn)(const char*, int32_t);
  tune_fn tune = h ? (tune_fn)dlsym(h, "skr_set_tuning") : nullptr;
  skr_step_plan p = {};
  p.n_terms = 4; p.n_group_a = 4; p.dtype_a = SKR_BF16; p.dtype_b = SKR_BF16; p.out0_dtype = SKR_BF16; p.out1_dtype = SKR_NONE;
  p.coef0[0] = 1.01; p.coef0[1] = -0.53; p.coef0[2] = 0.12; p.coef0[3] = 0.43; p.sample_numel = sample;
  p.noise_mode = 1; p.zeta0 = 0.3; p.stream0 = 1;
  auto lib = [&](const char* name) {
    if (!launch) { printf("(library not found)\n"); return; }
    bench(name, sets, [&](const Args& a) { const void* ins[4] = {a.in[0], a.in[1], a.in[2], a.in[3]}; launch(&p, ins, a.out, nullptr, seeds, n, nullptr); });
  };

  if (sweep) {
    // engine outputs cycle through `slots` start offsets: y = slot k, x = slot k-1, x_prev = slot k-2
    // (a step's result is the next step's sample and the history sample of the one after)
    CK(hipMemset(slab, 0x3f, pitch * 5 * NS + (60ll << 20)));
    g.reps = 2;
    const int64_t units[] = {8, 16, 32, 64, 128, 256, 512};
    for (int64_t unit : units) {
      for (int slots : {8, 3}) {
        double worst = 0, mean = 0;
        for (int k = 0; k < slots; ++k) {
          const int64_t offk[3] = {4 + ((k + slots - 1) % slots) * unit, 4 + ((k + slots - 2) % slots) * unit, 4 + k * unit};
          place_sets("custom", offk, false);
          char name[96];
          snprintf(name, sizeof name, "SWEEP unit %3ldK slots %d k=%d blk1024", (long)unit, slots, k);
          g_ref.clear();
          run_kv<1024, 1, 0, 0, 0, true>(name, sets);
        }
      }
    }
    return 0;
  }
  //            BLK  UV MAP PRIO SYNC NOISE
  run_kw<1, 1, 256, 7>("kw 1in uv1 blk256 R128", sets);
  run_kw<1, 2, 256, 6>("kw 1in uv2 blk256 R64", sets);
  run_kw<1, 4, 256, 5>("kw 1in uv4 blk256 R32", sets);
  run_kw<1, 8, 256, 4>("kw 1in uv8 blk256 R16", sets);
  run_kw<1, 1, 512, 6>("kw 1in uv1 blk512 R64", sets);
  run_kw<1, 1, 1024, 5>("kw 1in uv1 blk1024 R32", sets);
  run_kw<1, 2, 512, 5>("kw 1in uv2 blk512 R32", sets);
  run_kw<2, 1, 256, 7>("kw 2in uv1 blk256 R128", sets);
  run_kw<2, 2, 256, 6>("kw 2in uv2 blk256 R64", sets);
  run_kw<2, 4, 256, 5>("kw 2in uv4 blk256 R32", sets);
  run_kw<2, 1, 512, 6>("kw 2in uv1 blk512 R64", sets);
  run_kw<2, 2, 512, 5>("kw 2in uv2 blk512 R32", sets);
  run_kw<3, 1, 256, 7>("kw 3in uv1 blk256 R128", sets);
  run_kw<3, 2, 256, 6>("kw 3in uv2 blk256 R64", sets);
  run_kw<4, 1, 256, 7>("kw 4in uv1 blk256 R128", sets);
  run_kw<4, 2, 256, 6>("kw 4in uv2 blk256 R64", sets);
  return 0;
}
