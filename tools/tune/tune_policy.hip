// Cache-policy experiment for the headline kernel shape (4 bf16 reads + Philox + 1 bf16 write, per-sample grid).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tune_policy tune_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../skrample_amd/csrc/skr_philox.h"

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

struct Args { const u32x4_t* in[4]; u32x4_t* out; const uint64_t* seeds; float c[4]; float zeta; uint64_t stream; int64_t nvec, vps; };

// LD: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1, 6 sc0
template <int LD> __device__ __forceinline__ u32x4_t ldg(const u32x4_t* p) {
  u32x4_t v;
  if constexpr (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (LD == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int ST> __device__ __forceinline__ void stg(u32x4_t* p, u32x4_t v) {
  if constexpr (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if constexpr (ST == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}

// UV vectors per lane (BLK apart), one trip, per-sample grid
template <int LD, int ST, int UV, int BLK>
__global__ __launch_bounds__(BLK) void kuv(const Args a) {
  const int64_t smp = blockIdx.y;
  const int64_t vlo = smp * a.vps, vhi = vlo + a.vps;
  const uint64_t seed = a.seeds[smp];
  const int64_t v0 = vlo + (int64_t)blockIdx.x * BLK * UV + threadIdx.x;
  u32x4_t r[UV][4];
#pragma unroll
  for (int u = 0; u < UV; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) r[u][j] = ldg<LD>(a.in[j] + v0 + u * BLK);
  float z[UV][8];
#pragma unroll
  for (int u = 0; u < UV; ++u) {
    const uint64_t blk = (uint64_t)(v0 + u * BLK - vlo) * 2;
    skr::normal4(seed, a.stream, blk, z[u]);
    skr::normal4(seed, a.stream, blk + 1, z[u] + 4);
  }
#pragma unroll
  for (int u = 0; u < UV; ++u) {
    if (u == 0) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[u][0]), "+v"(r[u][1]), "+v"(r[u][2]), "+v"(r[u][3]) : "n"((UV - 1) * 4));
    else if (u == 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[u][0]), "+v"(r[u][1]), "+v"(r[u][2]), "+v"(r[u][3]) : "n"((UV - 2) * 4 > 0 ? (UV - 2) * 4 : 0));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[u][0]), "+v"(r[u][1]), "+v"(r[u][2]), "+v"(r[u][3]));
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(r[u][j][i] << 16), s[2 * i]);
        s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(r[u][j][i] & 0xFFFF0000u), s[2 * i + 1]);
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[u][i], s[i]);
    u32x4_t q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
    stg<ST>(a.out + v0 + u * BLK, q);
  }
}

template <int LD, int ST, int UV, int BLK>
void runuv(const char* name, std::vector<Args>& sets, int B, int iters = 300) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid((unsigned)(sets[0].vps / (BLK * UV)), B);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kuv<LD, ST, UV, BLK>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((kuv<LD, ST, UV, BLK>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 16 * 5;
  printf("%-34s grid=%ux%d %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, grid.x, B, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

template <int LD, int ST, bool NOISE>
__global__ __launch_bounds__(256) void k(const Args a) {
  const int64_t smp = blockIdx.y;
  const int64_t vlo = smp * a.vps, vhi = vlo + a.vps;
  const uint64_t seed = a.seeds[smp];
  const int64_t v = vlo + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= vhi) return;
  u32x4_t r0 = ldg<LD>(a.in[0] + v), r1 = ldg<LD>(a.in[1] + v), r2 = ldg<LD>(a.in[2] + v), r3 = ldg<LD>(a.in[3] + v);
  float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (NOISE) {
    const uint64_t blk = (uint64_t)(v - vlo) * 2;
    skr::normal4(seed, a.stream, blk, z);
    skr::normal4(seed, a.stream, blk + 1, z + 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
  u32x4_t raw[4] = {r0, r1, r2, r3};
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(raw[j][i] << 16), s[2 * i]);
      s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(raw[j][i] & 0xFFFF0000u), s[2 * i + 1]);
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[i], s[i]);
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
  stg<ST>(a.out + v, q);
}

template <int LD, int ST, bool NOISE>
void run(const char* name, std::vector<Args>& sets, int B, int iters = 300) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(32, B);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<LD, ST, NOISE>), grid, dim3(256), 0, 0, sets[i % sets.size()]);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<LD, ST, NOISE>), grid, dim3(256), 0, 0, sets[i % sets.size()]);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 16 * 5;
  printf("%-34s %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

int main() {
  const int B = 256;
  const int64_t sample = 4 * 128 * 128, n = (int64_t)B * sample;
  const int NS = 6;
  std::vector<Args> sets(NS);
  uint64_t* seeds; CK(hipMalloc(&seeds, B * 8));
  std::vector<uint64_t> hs(B); for (int i = 0; i < B; ++i) hs[i] = 42 + i;
  CK(hipMemcpy(seeds, hs.data(), B * 8, hipMemcpyHostToDevice));
  std::vector<uint16_t> host(n);
  for (int64_t i = 0; i < n; ++i) host[i] = 0x3f80 + (rand() & 0x7f);
  const int64_t pitch = 36ll << 20;
  char* slab; CK(hipMalloc((void**)&slab, pitch * 5 * NS + (64 << 20)));
  for (int s = 0; s < NS; ++s) {
    for (int j = 0; j < 4; ++j) { void* p = slab + pitch * (s * 5 + j) + 4096 * (2 * j + 1); CK(hipMemcpy(p, host.data(), n * 2, hipMemcpyHostToDevice)); sets[s].in[j] = (const u32x4_t*)p; }
    sets[s].out = (u32x4_t*)(slab + pitch * (s * 5 + 4) + 4096 * 9);
    sets[s].seeds = seeds; sets[s].c[0] = 1.01f; sets[s].c[1] = -0.53f; sets[s].c[2] = 0.12f; sets[s].c[3] = 0.43f;
    sets[s].zeta = 0.3f; sets[s].stream = 1; sets[s].nvec = n / 8; sets[s].vps = sample / 8;
  }
  for (int rep = 0; rep < 2; ++rep) {
    printf("-- with noise, st sc0 sc1 (rep %d)\n", rep);
    run<1, 3, true>("base uv1 blk256", sets, B);
    runuv<1, 3, 1, 256>("uv1 blk256", sets, B);
    runuv<1, 3, 2, 256>("uv2 blk256", sets, B);
    runuv<1, 3, 4, 256>("uv4 blk256", sets, B);
    runuv<1, 3, 1, 512>("uv1 blk512", sets, B);
    runuv<1, 3, 2, 512>("uv2 blk512", sets, B);
    runuv<1, 3, 1, 1024>("uv1 blk1024", sets, B);
    runuv<1, 3, 1, 128>("uv1 blk128", sets, B);
    runuv<1, 3, 2, 128>("uv2 blk128", sets, B);
    runuv<1, 3, 1, 64>("uv1 blk64", sets, B);
  }
  return 0;
}
