// Placement experiment for the headline launch: how the start offsets of the engine-allocated streams (x, x_prev, y)
// relative to 2 MiB-aligned model outputs (out, out_prev) change the step time.  Calls the shipped library.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tune_place tune_place.hip -ldl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include "../../include/skrample_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int B = 256, NS = 6;
  const int64_t sample = 4 * 128 * 128, n = (int64_t)B * sample;
  const int64_t slot = 40ll << 20;  // room per buffer (32 MiB payload + offsets up to 8 MiB)
  const int64_t pitch_mib = argc > 1 ? atoll(argv[1]) : 32;  // distance between buffer bases
  char* slab; CK(hipMalloc((void**)&slab, (pitch_mib << 20) * 5 * NS + slot + (64 << 20)));
  CK(hipMemset(slab, 0x3f, (pitch_mib << 20) * 5 * NS + slot));
  uint64_t* seeds; CK(hipMalloc(&seeds, B * 8));
  std::vector<uint64_t> hs(B); for (int i = 0; i < B; ++i) hs[i] = 42 + i;
  CK(hipMemcpy(seeds, hs.data(), B * 8, hipMemcpyHostToDevice));
  void* h = dlopen("skrample_amd/csrc/libskrample_hip.so", RTLD_NOW);
  if (!h) { printf("library not found: %s\n", dlerror()); return 1; }
  typedef int (*launch_fn)(const skr_step_plan*, const void* const*, void*, void*, const uint64_t*, int64_t, void*);
  launch_fn launch = (launch_fn)dlsym(h, "skr_step_launch");
  skr_step_plan p = {};
  p.n_terms = 4; p.n_group_a = 4; p.dtype_a = SKR_BF16; p.dtype_b = SKR_BF16; p.out0_dtype = SKR_BF16; p.out1_dtype = SKR_NONE;
  p.coef0[0] = 1.01; p.coef0[1] = -0.53; p.coef0[2] = 0.12; p.coef0[3] = 0.43; p.sample_numel = sample;
  p.noise_mode = 1; p.zeta0 = 0.3; p.stream0 = 1;
  // roles: 0 x, 1 out, 2 x_prev, 3 out_prev, 4 y ; offsets in KiB for x, x_prev, y (out / out_prev stay at 0)
  struct Cfg { const char* name; int64_t ox, oxp, oy; };
  std::vector<Cfg> cfgs = {
    {"T1 100,300,700", 100, 300, 700},
    {"T1 300,700,1100", 300, 700, 1100},
    {"T1 700,1100,1500", 700, 1100, 1500},
    {"T1 1100,1500,1900", 1100, 1500, 1900},
    {"T1 1500,1900,2300", 1500, 1900, 2300},
    {"T1 1900,2300,2700", 1900, 2300, 2700},
    {"T1 2300,2700,100", 2300, 2700, 100},
    {"T1 2700,100,300", 2700, 100, 300},
    {"T2 100,300,700", 100, 300, 700},
    {"T2 300,700,200", 300, 700, 200},
    {"T2 700,200,600", 700, 200, 600},
    {"T2 200,600,1400", 200, 600, 1400},
    {"T2 600,1400,500", 600, 1400, 500},
    {"T2 1400,500,1100", 1400, 500, 1100},
    {"T2 500,1100,100", 500, 1100, 100},
    {"T2 1100,100,300", 1100, 100, 300},
    {"cur 4,12,20", 4, 12, 20},
    {"cur 12,20,28", 12, 20, 28},
    {"cur 20,28,36", 20, 28, 36},
    {"cur 28,36,44", 28, 36, 44},
    {"cur 36,44,52", 36, 44, 52},
    {"cur 44,52,60", 44, 52, 60},
    {"cur 52,60,4", 52, 60, 4},
    {"cur 60,4,12", 60, 4, 12},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    for (auto& c : cfgs) {
      auto ptr = [&](int set, int role) -> void* {
        int64_t off = role == 0 ? c.ox : role == 2 ? c.oxp : role == 4 ? c.oy : 0;
        return slab + (pitch_mib << 20) * (set * 5 + role) + off * 1024;
      };
      auto go = [&](int i) { const int s = i % NS; const void* ins[4] = {ptr(s, 0), ptr(s, 1), ptr(s, 2), ptr(s, 3)}; launch(&p, ins, ptr(s, 4), nullptr, seeds, n, nullptr); };
      for (int i = 0; i < 12; ++i) go(i);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int i = 0; i < 300; ++i) go(i);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / 300;
      printf("pitch %lld MiB  offsets KiB %-28s %7.2f us  %6.3f TB/s\n", (long long)pitch_mib, c.name, us, (double)n * 10 / us / 1e6);
    }
  }
  return 0;
}
