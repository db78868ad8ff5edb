// Timeline harness for the Colored plane kernels (round 3): the product source is compiled WITH phase stamps
// (SKR_COLORED_TRACE) and one 256x(16,128,128) draw is run; per block the stamps (s_memrealtime, 10 ns) of
//   forward  (MODE 0): start | drawn | rows done | columns done | stores issued
//   inverse  (MODE 1): start | spectrum landed in LDS | columns done | rows done | factor known | stores issued
// and the hardware ids (XCC, SE, CU) come back, so the timeline of every CU can be rebuilt on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSKR_COLORED_TRACE -o tune_colored tune_colored.hip
//   ./tune_colored [batch=256] > timeline.txt
// -DSKR_STUB_DRAW (round 4): the Philox / Box-Muller draw replaced by a four-instruction hash, to MEASURE the generator's share of the
// forward kernel (the product source is untouched: its normal4 calls are renamed by the macro below, after the real header is in)
#ifdef SKR_STUB_DRAW
#include "../../skrample_amd/csrc/skr_philox.h"
namespace skr {
__device__ __forceinline__ void stub_normal4(uint64_t seed, uint64_t stream, uint64_t blk, float z[4]) {
  const uint32_t h = ((uint32_t)blk * 2654435761u) ^ (uint32_t)seed ^ (uint32_t)stream;
  z[0] = __uint_as_float(0x3f800000u | (h & 0x7fffffu)) - 1.5f;
  z[1] = __uint_as_float(0x3f800000u | ((h >> 3) & 0x7fffffu)) - 1.5f;
  z[2] = __uint_as_float(0x3f800000u | ((h >> 6) & 0x7fffffu)) - 1.5f;
  z[3] = __uint_as_float(0x3f800000u | ((h >> 9) & 0x7fffffu)) - 1.5f;
}
}  // namespace skr
#define normal4 stub_normal4
#endif
#include "../../skrample_amd/csrc/skr_colored.hip"
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int64_t batch = argc > 1 ? atoi(argv[1]) : 256;
  const int d1 = 16, d2 = 128, d3 = 128, d3h = d3 / 2 + 1;
  const int64_t unit = (int64_t)d1 * d2 * d3;
  void *out, *spec; float* scratch; double* partials; uint64_t* seeds; uint64_t* trace;
  const int64_t slots = 4096;
  CK(hipMalloc(&out, batch * unit * 2));
  CK(hipMalloc(&spec, batch * d1 * d2 * d3h * 8));
  CK(hipMalloc(&scratch, 16));
  CK(hipMalloc(&partials, 4 * batch * slots * 8));
  CK(hipMalloc(&seeds, batch * 8));
  const int64_t blocks = batch * d1;
  CK(hipMalloc(&trace, blocks * 16 * 8));
  CK(hipMemset(trace, 0, blocks * 16 * 8));
  std::vector<uint64_t> hs(batch);
  for (int64_t i = 0; i < batch; ++i) hs[i] = 1000 + i;
  CK(hipMemcpy(seeds, hs.data(), batch * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    skr::g_colored_trace = rep == 3 ? trace : nullptr;
    CK(hipEventRecord(e0));
    const int st = skr_noise_colored(out, SKR_BF16, spec, scratch, partials, slots, seeds, 256 * rep, batch, d1, d2, d3, 1.0, 0, 0.0, nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    if (st) { printf("skr_noise_colored -> %d\n", st); return 1; }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("# rep %d: %.1f us for the whole draw\n", rep, ms * 1e3);
  }
  std::vector<uint64_t> h(blocks * 16);
  CK(hipMemcpy(h.data(), trace, blocks * 16 * 8, hipMemcpyDeviceToHost));
  for (int mode = 0; mode < 2; ++mode) {
    const int base = mode ? 6 : 0, n = mode ? 6 : 5;
    uint64_t t0 = ~0ull, t1 = 0;
    for (int64_t b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 16 + base]); t1 = std::max(t1, h[b * 16 + base + n - 1]); }
    printf("# %s kernel: first start -> last store issue %.1f us\n", mode ? "inverse" : "forward", (t1 - t0) * 0.01);
    std::vector<double> sum(n, 0.0);
    std::map<uint64_t, std::vector<std::pair<uint64_t, int64_t>>> per_cu;
    for (int64_t b = 0; b < blocks; ++b) {
      for (int i = 1; i < n; ++i) sum[i] += (double)(h[b * 16 + base + i] - h[b * 16 + base + i - 1]) * 0.01;
      const uint64_t hw = h[b * 16 + (mode ? 14 : 12)], xcc = h[b * 16 + (mode ? 15 : 13)] & 0xf;
      const uint64_t cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
      per_cu[cu].push_back({h[b * 16 + base], b});
    }
    printf("# mean phase durations (us):");
    for (int i = 1; i < n; ++i) printf(" %.2f", sum[i] / blocks);
    printf("   (%zu distinct CUs seen)\n", per_cu.size());
    // dump the timeline of three CUs: every block that ran there, relative to the kernel start
    int shown = 0;
    for (auto& kv : per_cu) {
      if (shown++ >= 3) break;
      std::sort(kv.second.begin(), kv.second.end());
      printf("# CU %06llx: %zu blocks\n", (unsigned long long)kv.first, kv.second.size());
      for (auto& sb : kv.second) {
        printf("  block %5lld:", (long long)sb.second);
        for (int i = 0; i < n; ++i) printf(" %8.2f", (double)(h[sb.second * 16 + base + i] - t0) * 0.01);
        printf("\n");
      }
    }
    // overlap statistic: per CU, fraction of the kernel's span during which NO resident block is in a compute phase
    double idle_sum = 0; int cus = 0;
    for (auto& kv : per_cu) {
      std::vector<std::pair<uint64_t, int>> ev;  // compute intervals: forward [start, columns done], inverse [landed, rows done]
      for (auto& sb : kv.second) {
        const int64_t b = sb.second;
        const uint64_t c0 = mode ? h[b * 16 + base + 1] : h[b * 16 + base], c1 = h[b * 16 + base + 3];
        ev.push_back({c0, +1}); ev.push_back({c1, -1});
      }
      std::sort(ev.begin(), ev.end());
      uint64_t last = t0; int depth = 0; uint64_t idle = 0;
      for (auto& e : ev) { if (depth == 0) idle += e.first - last; depth += e.second; last = e.first; }
      idle += t1 - last;
      idle_sum += (double)idle / (double)(t1 - t0); ++cus;
    }
    printf("# mean fraction of the kernel span in which a CU has NO block in a compute phase: %.3f\n", idle_sum / cus);
  }
  if (getenv("SKR_FFT_NO_SAMPLE") == nullptr) {
    // one-launch kernel (colored_sample): the twelve stamps of a block are one timeline --
    // start | drawn | rows | columns | stored || arrival 1 + outer axis done || arrival 2 = inverse start | landed | columns | rows | factor | stored
    uint64_t t0 = ~0ull;
    for (int64_t b = 0; b < blocks; ++b) t0 = std::min(t0, h[b * 16]);
    std::vector<double> sum(12, 0.0);
    for (int64_t b = 0; b < blocks; ++b)
      for (int i = 1; i < 12; ++i) sum[i] += (double)(h[b * 16 + i] - h[b * 16 + i - 1]) * 0.01;
    printf("# one launch, mean interval durations (us):");
    for (int i = 1; i < 12; ++i) printf(" %.2f", sum[i] / blocks);
    printf("\n");
    std::vector<std::pair<uint64_t, int64_t>> order;
    for (int64_t b = 0; b < blocks; ++b) order.push_back({h[b * 16], b});
    std::sort(order.begin(), order.end());
    for (int64_t k = 0; k < (int64_t)order.size(); k += order.size() / 48) {
      const int64_t b = order[k].second;
      printf("  start rank %5lld block %5lld:", (long long)k, (long long)b);
      for (int i = 0; i < 12; ++i) printf(" %8.2f", (double)(h[b * 16 + i] - t0) * 0.01);
      printf("\n");
    }
  }
  return 0;
}
