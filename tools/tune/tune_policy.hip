// Cache-policy experiment for the headline kernel shape (4 bf16 reads + Philox + 1 bf16 write, per-sample grid).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tune_policy tune_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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

// fp32 output (32 B per lane): SPLIT = each lane writes its own 32 contiguous bytes with two instructions (each
// instruction covers half of every line); FULL = instruction 0 writes the wave's first KiB, instruction 1 the second
// (what a lane exchange before the store would produce).  Timing only: values land in permuted places under FULL.
template <int ST, bool FULL>
__global__ __launch_bounds__(256) void kf32(const Args a, u32x4_t* out32) {
  const int64_t v = (int64_t)blockIdx.y * a.vps + (int64_t)blockIdx.x * 256 + threadIdx.x;
  u32x4_t r0 = ldg<1>(a.in[0] + v), r1 = ldg<1>(a.in[1] + v), r2 = ldg<1>(a.in[2] + v), r3 = ldg<1>(a.in[3] + v);
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
  u32x4_t q0 = {__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])};
  u32x4_t q1 = {__float_as_uint(s[4]), __float_as_uint(s[5]), __float_as_uint(s[6]), __float_as_uint(s[7])};
  if constexpr (FULL) {
    const int64_t wave_base = (v & ~(int64_t)63) * 2;  // in 16-byte units
    const int lane = threadIdx.x & 63;
    stg<ST>(out32 + wave_base + lane, q0);
    stg<ST>(out32 + wave_base + 64 + lane, q1);
  } else {
    stg<ST>(out32 + 2 * v, q0);
    stg<ST>(out32 + 2 * v + 1, q1);
  }
}

template <int ST, bool FULL>
void runf32(const char* name, std::vector<Args>& sets, std::vector<u32x4_t*>& outs, int B, int iters = 300) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(32, B);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kf32<ST, FULL>), grid, dim3(256), 0, 0, sets[i % sets.size()], outs[i % outs.size()]);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((kf32<ST, FULL>), grid, dim3(256), 0, 0, sets[i % sets.size()], outs[i % outs.size()]);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 16 * 6;
  printf("%-34s %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
}

// UniPC-like mix: 2 bf16 inputs + 3 fp32 inputs -> fp32 out0 + bf16 out1 (26 B/element).
// L44 = false: lane owns 8 consecutive elements (bf16: one 16-B access; fp32: two 16-B accesses to its own 32 B).
// L44 = true : lane owns elements {4l..4l+3} and {256+4l..} of its wave's 512-element tile (bf16: two 8-B accesses,
//              fp32: two 16-B accesses) -- every wave instruction covers whole lines.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <bool L44, int ST32>
__global__ __launch_bounds__(256) void kmix(const Args a, const u32x4_t* f0, const u32x4_t* f1, const u32x4_t* f2, u32x4_t* out32) {
  const int64_t v = (int64_t)blockIdx.y * a.vps + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t tile = v >> 6;  // 512 elements
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  u32x4_t h[2];   // bf16 operands, 8 elements each
  u32x4_t w[3][2];  // fp32 operands
  const u32x4_t* fp[3] = {f0, f1, f2};
  if constexpr (L44) {
    const u32x2_t* b0 = (const u32x2_t*)a.in[0]; const u32x2_t* b1 = (const u32x2_t*)a.in[1];
    const int64_t g0 = tile * 128 + lane, g1 = g0 + 64;  // in 4-element groups
    u32x2_t x0 = __builtin_nontemporal_load(b0 + g0), x1 = __builtin_nontemporal_load(b0 + g1);
    u32x2_t y0 = __builtin_nontemporal_load(b1 + g0), y1 = __builtin_nontemporal_load(b1 + g1);
#pragma unroll
    for (int j = 0; j < 3; ++j) { w[j][0] = __builtin_nontemporal_load(fp[j] + g0); w[j][1] = __builtin_nontemporal_load(fp[j] + g1); }
    h[0] = u32x4_t{x0[0], x0[1], x1[0], x1[1]}; h[1] = u32x4_t{y0[0], y0[1], y1[0], y1[1]};
  } else {
    h[0] = __builtin_nontemporal_load(a.in[0] + v); h[1] = __builtin_nontemporal_load(a.in[1] + v);
#pragma unroll
    for (int j = 0; j < 3; ++j) { w[j][0] = __builtin_nontemporal_load(fp[j] + 2 * v); w[j][1] = __builtin_nontemporal_load(fp[j] + 2 * v + 1); }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[2 * i] = __builtin_fmaf(a.c[j], __uint_as_float(h[j][i] << 16), s[2 * i]);
      s[2 * i + 1] = __builtin_fmaf(a.c[j], __uint_as_float(h[j][i] & 0xFFFF0000u), s[2 * i + 1]);
    }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] = __builtin_fmaf(a.c[j], __uint_as_float(w[j][0][i]), s[i]); s[4 + i] = __builtin_fmaf(a.c[j], __uint_as_float(w[j][1][i]), s[4 + i]); }
  u32x4_t q0 = {__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])};
  u32x4_t q1 = {__float_as_uint(s[4]), __float_as_uint(s[5]), __float_as_uint(s[6]), __float_as_uint(s[7])};
  u32x4_t qb;
#pragma unroll
  for (int i = 0; i < 4; ++i) qb[i] = pack_bf16(s[2 * i] * 0.5f, s[2 * i + 1] * 0.5f);
  if constexpr (L44) {
    const int64_t g0 = tile * 128 + lane, g1 = g0 + 64;
    stg<ST32>(out32 + g0, q0);
    stg<ST32>(out32 + g1, q1);
    u32x2_t* ob = (u32x2_t*)a.out;
    u32x2_t lo = {qb[0], qb[1]}, hi = {qb[2], qb[3]};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(ob + g0), "v"(lo) : "memory");
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(ob + g1), "v"(hi) : "memory");
  } else {
    stg<ST32>(out32 + 2 * v, q0);
    stg<ST32>(out32 + 2 * v + 1, q1);
    stg<3>(a.out + v, qb);
  }
}

template <bool L44, int ST32>
void runmix(const char* name, std::vector<Args>& sets, std::vector<u32x4_t*>& f32s, int B, int iters = 300) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(32, B);
  auto go = [&](int i) { const int k = i % 2; hipLaunchKernelGGL((kmix<L44, ST32>), grid, dim3(256), 0, 0, sets[i % sets.size()], f32s[4 * k], f32s[4 * k + 1], f32s[4 * k + 2], f32s[4 * k + 3]); };
  for (int i = 0; i < 10; ++i) go(i);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) go(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
  double bytes = (double)sets[0].nvec * 8 * 22;  // 2*2 + 3*4 + 4 + 2 = 22 B/element
  printf("%-40s %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
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
  void* h = dlopen("skrample_amd/csrc/libskrample_hip.so", RTLD_NOW);
  typedef int (*launch_fn)(const skr_step_plan*, const void* const*, void*, void*, const uint64_t*, int64_t, void*);
  launch_fn launch = h ? (launch_fn)dlsym(h, "skr_step_launch") : nullptr;
  skr_step_plan p = {};
  p.n_terms = 4; p.n_group_a = 4; p.dtype_a = SKR_BF16; p.dtype_b = SKR_BF16; p.out0_dtype = SKR_BF16; p.out1_dtype = SKR_NONE;
  p.coef0[0] = 1.01; p.coef0[1] = -0.53; p.coef0[2] = 0.12; p.coef0[3] = 0.43; p.sample_numel = sample;
  p.noise_mode = 1; p.zeta0 = 0.3; p.stream0 = 1;
  auto lib = [&](const char* name) {
    if (!launch) { printf("(library not found)\n"); return; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&](int i) { const Args& a = sets[i % sets.size()]; const void* ins[4] = {a.in[0], a.in[1], a.in[2], a.in[3]}; launch(&p, ins, a.out, nullptr, seeds, n, nullptr); };
    for (int i = 0; i < 10; ++i) go(i);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 300; ++i) go(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / 300;
    printf("%-34s               %7.2f us  %6.3f TB/s  frac8=%.3f\n", name, us, (double)n * 10 / us / 1e6, (double)n * 10 / us / 1e6 / 8.0);
  };
  for (int rep = 0; rep < 3; ++rep) {
    printf("-- with noise, st sc0 sc1 (rep %d)\n", rep);
    runuv<1, 3, 1, 256>("uv1 blk256", sets, B);
    lib("LIB skr_step_launch philox");
    runuv<1, 3, 1, 1024>("uv1 blk1024", sets, B);
    run<1, 3, true>("k (early exit) uv1 blk256", sets, B);
    lib("LIB skr_step_launch philox");
  }
  return 0;
}
