// Stand-alone tuning harness for the fused step kernel structure (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tune_step tune_step.hip && ./tune_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../skrample_amd/csrc/skr_philox.h"
#include "../../include/skrample_hip.h"
#include <dlfcn.h>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2_t f = {a, b};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}

struct Args {
  const u32x4_t* in[4];
  u32x4_t* out;
  const uint64_t* seeds;
  float c[4];
  float zeta;
  uint64_t stream;
  int64_t nvec;
  int64_t vps;  // vectors per sample
};

template <bool NT>
__device__ __forceinline__ u32x4_t ld(const u32x4_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4_t* p, u32x4_t v) {
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");  // "nt" slot now = write-through
  else *p = v;
}

__device__ __forceinline__ void noise8(const Args& a, int64_t vec, float z[8]) {
  const int64_t smp = vec / a.vps;
  const int64_t r = (vec - smp * a.vps) * 8;
  const uint64_t seed = a.seeds[smp];
  skr::normal4(seed, a.stream, (uint64_t)r >> 2, z);
  skr::normal4(seed, a.stream, ((uint64_t)r >> 2) + 1, z + 4);
}

template <int K, bool NT_LD, bool NT_ST, int UV, bool NOISE, bool NOISE_FIRST, bool MEM>
__global__ __launch_bounds__(256) void k(const Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256 * UV;
  for (int64_t v0 = ((int64_t)blockIdx.x * 256) * UV + threadIdx.x; v0 < a.nvec; v0 += stride) {
    u32x4_t raw[UV][K];
    if constexpr (MEM) {
#pragma unroll
      for (int u = 0; u < UV; ++u)
#pragma unroll
        for (int j = 0; j < K; ++j) raw[u][j] = ld<NT_LD>(a.in[j] + v0 + u * 256);
    }
    float z[UV][8];
    if constexpr (NOISE && NOISE_FIRST) {
#pragma unroll
      for (int u = 0; u < UV; ++u) noise8(a, v0 + u * 256, z[u]);
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      float s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = 0.f;
      if constexpr (MEM) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(raw[u][j][i] << 16), s[2 * i]);
            s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(raw[u][j][i] & 0xFFFF0000u), s[2 * i + 1]);
          }
        }
      }
      if constexpr (NOISE) {
        if constexpr (!NOISE_FIRST) noise8(a, v0 + u * 256, z[u]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[u][i], s[i]);
      }
      u32x4_t q;
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
      if (MEM || q[0] == 0x12345678u) st<NT_ST>(a.out + v0 + u * 256, q);
    }
  }
}

// per-sample grid: blockIdx.y = sample, scalar seed, no division
template <int K, int UV, int BLK, bool NOISE_FIRST>
__global__ __launch_bounds__(BLK) void k2(const Args a) {
  const int64_t smp = blockIdx.y;
  const int64_t vlo = smp * a.vps, vhi = vlo + a.vps;
  const uint64_t seed = a.seeds[smp];
  const int64_t stride = (int64_t)gridDim.x * BLK * UV;
  for (int64_t v0 = vlo + ((int64_t)blockIdx.x * BLK) * UV + threadIdx.x; v0 < vhi; v0 += stride) {
    u32x4_t raw[UV][K];
#pragma unroll
    for (int u = 0; u < UV; ++u)
#pragma unroll
      for (int j = 0; j < K; ++j) if (v0 + u * BLK < vhi) raw[u][j] = __builtin_nontemporal_load(a.in[j] + v0 + u * BLK);
    float z[UV][8];
    if constexpr (NOISE_FIRST) {
#pragma unroll
      for (int u = 0; u < UV; ++u) { const uint64_t blk = (uint64_t)(v0 + u * BLK - vlo) * 2; skr::normal4(seed, a.stream, blk, z[u]); skr::normal4(seed, a.stream, blk + 1, z[u] + 4); }
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      if (v0 + u * BLK >= vhi) continue;
      float s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(raw[u][j][i] << 16), s[2 * i]);
          s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(raw[u][j][i] & 0xFFFF0000u), s[2 * i + 1]);
        }
      }
      if constexpr (!NOISE_FIRST) { const uint64_t blk = (uint64_t)(v0 + u * BLK - vlo) * 2; skr::normal4(seed, a.stream, blk, z[u]); skr::normal4(seed, a.stream, blk + 1, z[u] + 4); }
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[u][i], s[i]);
      u32x4_t q;
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(a.out + v0 + u * BLK), "v"(q) : "memory");
    }
  }
}

// software-pipelined per-sample grid: the loads of trip i+1 are in flight while trip i does Philox + FMAs
template <int K, int BLK>
__global__ __launch_bounds__(BLK) void k3(const Args a) {
  const int64_t smp = blockIdx.y;
  const int64_t vlo = smp * a.vps, vhi = vlo + a.vps;
  const uint64_t seed = a.seeds[smp];
  const int64_t stride = (int64_t)gridDim.x * BLK;
  int64_t v = vlo + (int64_t)blockIdx.x * BLK + threadIdx.x;
  u32x4_t nxt[K];
  if (v < vhi) {
#pragma unroll
    for (int j = 0; j < K; ++j) nxt[j] = __builtin_nontemporal_load(a.in[j] + v);
  }
  while (v < vhi) {
    u32x4_t raw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) raw[j] = nxt[j];
    const int64_t vn = v + stride;
    if (vn < vhi) {
#pragma unroll
      for (int j = 0; j < K; ++j) nxt[j] = __builtin_nontemporal_load(a.in[j] + vn);
    }
    float z[8];
    const uint64_t blk = (uint64_t)(v - vlo) * 2;
    skr::normal4(seed, a.stream, blk, z);
    skr::normal4(seed, a.stream, blk + 1, z + 4);
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(raw[j][i] << 16), s[2 * i]);
        s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(raw[j][i] & 0xFFFF0000u), s[2 * i + 1]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(a.zeta, z[i], s[i]);
    u32x4_t q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pack_bf16(s[2 * i], s[2 * i + 1]);
    __builtin_nontemporal_store(q, a.out + v);
    v = vn;
  }
}

template <int K, int BLK>
void run3(const char* name, std::vector<Args>& sets, int B, int bx, int iters = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(bx, B);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k3<K, BLK>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k3<K, BLK>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 16 * (K + 1);
  printf("%-46s grid=%dx%d  %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, bx, B, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

template <int K, int UV, int BLK, bool NF>
void run2(const char* name, std::vector<Args>& sets, int B, int bx, int iters = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(bx, B);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k2<K, UV, BLK, NF>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k2<K, UV, BLK, NF>), grid, dim3(BLK), 0, 0, sets[i % sets.size()]);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 16 * (K + 1);
  printf("%-46s grid=%dx%d  %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, bx, B, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

template <int K, bool NT_LD, bool NT_ST, int UV, bool NOISE, bool NOISE_FIRST, bool MEM>
void run(const char* name, std::vector<Args>& sets, int blocks_cap, int iters = 200) {
  int64_t nvec = sets[0].nvec;
  int64_t blocks = (nvec + 256 * UV - 1) / (256 * UV);
  if (blocks_cap > 0 && blocks > blocks_cap) blocks = blocks_cap;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<K, NT_LD, NT_ST, UV, NOISE, NOISE_FIRST, MEM>), dim3(blocks), dim3(256), 0, 0, sets[i % sets.size()]);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<K, NT_LD, NT_ST, UV, NOISE, NOISE_FIRST, MEM>), dim3(blocks), dim3(256), 0, 0, sets[i % sets.size()]);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / iters;
  double bytes = (double)nvec * 16 * (K + 1);
  printf("%-46s blocks=%6lld  %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, (long long)blocks, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

template <int K, int UV>
__global__ __launch_bounds__(256) void k_read(const Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256 * UV;
  u32x4_t acc = {0, 0, 0, 0};
  for (int64_t v0 = ((int64_t)blockIdx.x * 256) * UV + threadIdx.x; v0 < a.nvec; v0 += stride) {
#pragma unroll
    for (int u = 0; u < UV; ++u)
#pragma unroll
      for (int j = 0; j < K; ++j) acc ^= __builtin_nontemporal_load(a.in[j] + v0 + u * 256);
  }
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) a.out[0] = acc;
}
template <int UV>
__global__ __launch_bounds__(256) void k_write(const Args a) {
  const int64_t stride = (int64_t)gridDim.x * 256 * UV;
  for (int64_t v0 = ((int64_t)blockIdx.x * 256) * UV + threadIdx.x; v0 < a.nvec; v0 += stride) {
#pragma unroll
    for (int u = 0; u < UV; ++u) { u32x4_t q = {(uint32_t)v0, 1, 2, 3}; __builtin_nontemporal_store(q, a.out + v0 + u * 256); }
  }
}
template <typename F>
void timeit(const char* name, F launch, double bytes, int iters = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(i);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  printf("%-46s %7.2f us  %6.3f TB/s\n", name, us, bytes / us / 1e6);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  const int64_t stagger = argc > 2 ? atoll(argv[2]) : 0;
  const int64_t pitch_mib = argc > 3 ? atoll(argv[3]) : 36;
  const int mask = argc > 4 ? atoi(argv[4]) : 31;
  const int64_t sample = 4 * 128 * 128;
  const int64_t n = (int64_t)B * sample;
  const int NS = 4;
  std::vector<Args> sets(NS);
  uint64_t* seeds; CK(hipMalloc(&seeds, B * 8));
  std::vector<uint64_t> hs(B); for (int i = 0; i < B; ++i) hs[i] = 42 + i;
  CK(hipMemcpy(seeds, hs.data(), B * 8, hipMemcpyHostToDevice));
  std::vector<uint16_t> host(n);
  for (int64_t i = 0; i < n; ++i) host[i] = 0x3f80 + (rand() & 0x7f);
  char* slab; const int64_t pitch = pitch_mib << 20; CK(hipMalloc((void**)&slab, pitch * 5 * NS + (64 << 20)));
  printf("stagger=%lld bytes pitch=%lld MiB mask=%d\n", (long long)stagger, (long long)pitch_mib, mask);
  for (int s = 0; s < NS; ++s) {
    for (int j = 0; j < 4; ++j) { void* p = slab + pitch * (s * 5 + j) + (((mask >> j) & 1) ? stagger * (j + 1) : 0); CK(hipMemcpy(p, host.data(), n * 2, hipMemcpyHostToDevice)); sets[s].in[j] = (const u32x4_t*)p; }
    void* o = slab + pitch * (s * 5 + 4) + (((mask >> 4) & 1) ? stagger * 5 : 0); sets[s].out = (u32x4_t*)o;
    sets[s].seeds = seeds; sets[s].c[0] = 1.01f; sets[s].c[1] = -0.53f; sets[s].c[2] = 0.12f; sets[s].c[3] = 0.43f;
    sets[s].zeta = 0.3f; sets[s].stream = 1; sets[s].nvec = n / 8; sets[s].vps = sample / 8;
  }
  printf("B=%d n=%lld\n", B, (long long)n);
  auto lib_block = [&]()   {
    // the shipped library kernel on the very same buffers
    void* h = dlopen("skrample_amd/csrc/libskrample_hip.so", RTLD_NOW);
    if (h) {
      typedef int (*launch_fn)(const skr_step_plan*, const void* const*, void*, void*, const uint64_t*, int64_t, void*);
      launch_fn launch = (launch_fn)dlsym(h, "skr_step_launch");
      skr_step_plan p = {};
      p.n_terms = 4; p.n_group_a = 4; p.dtype_a = SKR_BF16; p.dtype_b = SKR_BF16; p.out0_dtype = SKR_BF16; p.out1_dtype = SKR_NONE;
      p.coef0[0] = 1.01; p.coef0[1] = -0.53; p.coef0[2] = 0.12; p.coef0[3] = 0.43;
      p.sample_numel = sample;
      for (int noise = 0; noise < 2; ++noise) {
        p.noise_mode = noise; p.zeta0 = noise ? 0.3 : 0.0; p.stream0 = 1;
        auto go = [&](int i) { const void* ins[4] = {sets[i % NS].in[0], sets[i % NS].in[1], sets[i % NS].in[2], sets[i % NS].in[3]}; launch(&p, ins, sets[i % NS].out, nullptr, seeds, n, nullptr); };
        timeit(noise ? "LIB skr_step_launch K=4 philox" : "LIB skr_step_launch K=4 no noise", go, (double)n * 10);
      }
    } else printf("(library not found: %s)\n", dlerror());
  };
  lib_block();
  run2<4, 1, 256, true>("k2 uv1 blk256 nf  bx32 (early)", sets, B, 32);
  lib_block();
  {
    double rb = (double)n * 2 * 4, wb = (double)n * 2;
    timeit("read-only 4 streams uv1 grid2048", [&](int i) { hipLaunchKernelGGL((k_read<4, 1>), dim3(2048), dim3(256), 0, 0, sets[i % NS]); }, rb);
    timeit("read-only 4 streams uv4 grid2048", [&](int i) { hipLaunchKernelGGL((k_read<4, 4>), dim3(2048), dim3(256), 0, 0, sets[i % NS]); }, rb);
    timeit("read-only 1 stream  uv4 grid2048", [&](int i) { hipLaunchKernelGGL((k_read<1, 4>), dim3(2048), dim3(256), 0, 0, sets[i % NS]); }, rb / 4);
    timeit("write-only uv1 grid2048", [&](int i) { hipLaunchKernelGGL((k_write<1>), dim3(2048), dim3(256), 0, 0, sets[i % NS]); }, wb);
    timeit("write-only uv4 grid2048", [&](int i) { hipLaunchKernelGGL((k_write<4>), dim3(2048), dim3(256), 0, 0, sets[i % NS]); }, wb);
    timeit("hipMemcpyAsync D2D (1R+1W)", [&](int i) { CK(hipMemcpyAsync((void*)sets[i % NS].out, (const void*)sets[i % NS].in[0], n * 2, hipMemcpyDeviceToDevice, 0)); }, wb * 2);
  }
  //                 K  NTLD  NTST  UV NOISE NFIRST MEM
  run<4, true, true, 1, false, false, true>("nt/nt uv1 cap2048", sets, 2048);
  run<4, true, true, 1, false, false, true>("nt/nt uv1 cap1024", sets, 1024);
  run<4, true, true, 1, false, false, true>("nt/nt uv1 cap4096", sets, 4096);
  run<4, true, true, 1, false, false, true>("nt/nt uv1 nocap", sets, 0);
  run<4, false, true, 1, false, false, true>("plain/nt uv1 cap2048", sets, 2048);
  run<4, false, false, 1, false, false, true>("plain/plain uv1 cap2048", sets, 2048);
  run<4, false, false, 1, false, false, true>("plain/plain uv1 nocap", sets, 0);
  run<4, true, false, 1, false, false, true>("nt/plain uv1 cap2048", sets, 2048);
  run<4, true, true, 2, false, false, true>("nt/nt uv2 cap2048", sets, 2048);
  run<4, true, true, 2, false, false, true>("nt/nt uv2 cap1024", sets, 1024);
  run<4, true, true, 2, false, false, true>("nt/nt uv2 nocap", sets, 0);
  run<4, false, false, 2, false, false, true>("plain/plain uv2 nocap", sets, 0);
  run<4, true, true, 4, false, false, true>("nt/nt uv4 cap1024", sets, 1024);
  run<4, true, true, 4, false, false, true>("nt/nt uv4 nocap", sets, 0);
  printf("-- with noise\n");
  run<4, true, true, 1, true, false, true>("noise last  nt/nt uv1 cap2048", sets, 2048);
  run<4, true, true, 1, true, true, true>("noise first nt/nt uv1 cap2048", sets, 2048);
  run<4, true, true, 1, true, true, true>("noise first nt/nt uv1 nocap", sets, 0);
  run<4, true, true, 1, true, true, true>("noise first nt/nt uv1 cap4096", sets, 4096);
  run<4, true, true, 2, true, true, true>("noise first nt/nt uv2 cap2048", sets, 2048);
  run<4, true, true, 2, true, true, true>("noise first nt/nt uv2 nocap", sets, 0);
  run<4, false, false, 1, true, true, true>("noise first plain uv1 nocap", sets, 0);
  printf("-- pipelined per-sample grid with noise\n");
  run3<4, 256>("k3 pipelined blk256 bx32 (1 trip)", sets, B, 32);
  run3<4, 256>("k3 pipelined blk256 bx16 (2 trips)", sets, B, 16);
  run3<4, 256>("k3 pipelined blk256 bx8  (4 trips)", sets, B, 8);
  run3<4, 256>("k3 pipelined blk256 bx4  (8 trips)", sets, B, 4);
  run3<4, 512>("k3 pipelined blk512 bx4  (4 trips)", sets, B, 4);
  run3<4, 128>("k3 pipelined blk128 bx16 (4 trips)", sets, B, 16);
  printf("-- per-sample grid with noise\n");
  run2<4, 1, 256, true>("k2 uv1 blk256 nf  bx32", sets, B, 32);
  run2<4, 1, 256, false>("k2 uv1 blk256 nl  bx32", sets, B, 32);
  run2<4, 1, 256, true>("k2 uv1 blk256 nf  bx16", sets, B, 16);
  run2<4, 1, 256, true>("k2 uv1 blk256 nf  bx8", sets, B, 8);
  run2<4, 2, 256, true>("k2 uv2 blk256 nf  bx16", sets, B, 16);
  run2<4, 2, 256, false>("k2 uv2 blk256 nl  bx16", sets, B, 16);
  run2<4, 2, 256, true>("k2 uv2 blk256 nf  bx8", sets, B, 8);
  run2<4, 4, 256, true>("k2 uv4 blk256 nf  bx8", sets, B, 8);
  run2<4, 4, 256, false>("k2 uv4 blk256 nl  bx8", sets, B, 8);
  run2<4, 1, 512, true>("k2 uv1 blk512 nf  bx16", sets, B, 16);
  run2<4, 2, 512, true>("k2 uv2 blk512 nf  bx8", sets, B, 8);
  run2<4, 1, 1024, true>("k2 uv1 blk1024 nf bx8", sets, B, 8);
  run2<4, 1, 128, true>("k2 uv1 blk128 nf  bx64", sets, B, 64);
  run2<4, 1, 64, true>("k2 uv1 blk64 nf  bx128", sets, B, 128);
  lib_block();
  printf("-- VALU only (no memory)\n");
  run<4, true, true, 1, true, true, false>("philox+boxmuller only uv1 cap2048", sets, 2048);
  run<4, true, true, 1, true, true, false>("philox+boxmuller only uv1 nocap", sets, 0);
  return 0;
}
