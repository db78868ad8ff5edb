// Timeline harness for the persistent inverse kernel of 128 x 128 planes (round 5): the product source compiled WITH phase stamps
// (-DSKR_COLORED_TRACE), one 256 x (16,128,128) draw; per block and plane trip eight stamps (s_memrealtime, 10 ns):
//   0 trip start | 1 column stage written + prefetch issued | 2 past the barrier | 3 column radix-16 pass done | 4 packing + 8-point row stage
//   computed | 5 row tile written, past the barrier | 6 row radix-16 pass done, past the barrier | 7 stores issued
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSKR_COLORED_TRACE -o tune_inverse128 tune_inverse128.hip && ./tune_inverse128
#include "../../skrample_amd/csrc/skr_colored.hip"
#include <cstdio>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int64_t batch = argc > 1 ? atoi(argv[1]) : 256;
  const int d1 = 16, d2 = 128, d3 = 128, d3h = d3 / 2 + 1;
  const int64_t unit = (int64_t)d1 * d2 * d3;
  void *out, *spec; float* scratch; double* partials; uint64_t* seeds; uint64_t* trace;
  const int64_t slots = 64;
  CK(hipMalloc(&out, batch * unit * 2));
  CK(hipMalloc(&spec, batch * d1 * d2 * d3h * 8));
  CK(hipMalloc(&scratch, 16));
  CK(hipMalloc(&partials, 4 * batch * slots * 8));
  CK(hipMalloc(&seeds, batch * 8));
  const int64_t blocks = 4096;  // upper bound on the persistent grid
  CK(hipMalloc(&trace, (65536 + blocks * 256) * 8));
  CK(hipMemset(trace, 0, (65536 + blocks * 256) * 8));
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
  std::vector<uint64_t> h(blocks * 256);
  CK(hipMemcpy(h.data(), trace + 65536, blocks * 256 * 8, hipMemcpyDeviceToHost));
  // stamps live at trace[65536 + (block * 8 + trip) * 8 + i] (the forward kernel of the same draw stamps the first 65536 words in its own layout)
  double sum[16][8] = {}; int64_t cnt[16] = {};
  uint64_t first = ~0ull, last = 0;
  int64_t used = 0;
  for (int64_t b = 0; b < blocks; ++b) {
    bool ok = true;
    for (int t = 0; t < 16 && ok; ++t) {
      const uint64_t* s = &h[(b * 16 + t) * 16];
      if (s[0] == 0) { ok = t > 0; break; }
      for (int i = 1; i < 8; ++i) if (s[i] < s[i - 1]) ok = false;
    }
    if (!ok || h[b * 256] == 0) continue;
    ++used;
    for (int t = 0; t < 16; ++t) {
      const uint64_t* s = &h[(b * 16 + t) * 16];
      if (s[0] == 0) break;
      if (s[0] < first) first = s[0];
      if (s[7] > last) last = s[7];
      for (int i = 1; i < 8; ++i) sum[t][i] += (double)(s[i] - s[i - 1]) * 0.01;
      if (t + 1 < 16 && h[(b * 16 + t + 1) * 16] != 0) sum[t][0] += (double)(h[(b * 16 + t + 1) * 16] - s[7]) * 0.01;  // closing barrier -> next trip
      ++cnt[t];
    }
  }
  printf("# %lld blocks with stamps; first trip start -> last stores issued %.1f us\n", (long long)used, (double)(last - first) * 0.01);
  printf("# mean us per phase, by plane trip:  A+prefetch | barrier | B cols r16 | C pack+dft8 | barrier+write+barrier | D rows r16+barrier | E stores | closing barrier\n");
  for (int t = 0; t < 16; ++t) {
    if (!cnt[t]) continue;
    printf("trip %d (%lld blocks):", t, (long long)cnt[t]);
    double tot = 0;
    for (int i = 1; i < 8; ++i) { printf(" %6.2f", sum[t][i] / cnt[t]); tot += sum[t][i] / cnt[t]; }
    printf(" %6.2f | total %6.2f\n", sum[t][0] / cnt[t], tot + sum[t][0] / cnt[t]);
  }
  {  // planes finished and blocks inside a trip, per 5 us of the kernel
    const int bins = 24;
    std::vector<int> done(bins, 0), active(bins, 0);
    for (int64_t b = 0; b < blocks; ++b)
      for (int t = 0; t < 16 && h[(b * 16 + t) * 16]; ++t) {
        const double t0 = (double)(h[(b * 16 + t) * 16] - first) * 0.01, t1 = (double)(h[(b * 16 + t) * 16 + 7] - first) * 0.01;
        const int b1 = (int)(t1 / 5.0);
        if (b1 >= 0 && b1 < bins) done[b1] += 1;
        for (int k = (int)(t0 / 5.0); k <= b1 && k < bins; ++k) if (k >= 0) active[k] += 1;
      }
    printf("# per 5 us: planes finished (blocks inside a trip):");
    for (int k = 0; k < bins; ++k) if (done[k] || active[k]) printf(" %d(%d)", done[k], active[k]);
    printf("\n");
  }
  // shader clock per trip (s_memtime ticks per 10 ns of s_memrealtime between consecutive trip starts), and how many blocks share a CU
  {
    double clk[16] = {}; int64_t n[16] = {};
    std::map<uint64_t, int> per_cu;
    for (int64_t b = 0; b < blocks; ++b) {
      if (h[b * 256] == 0) continue;
      const uint64_t id = h[b * 256 + 9];
      per_cu[((id >> 32) & 0xf) << 16 | (id & 0xff00) | ((id >> 13) & 0x7) << 4] += 1;  // xcc | se | sh.. cu bits (HW_ID: cu_id 11:8, sh 12, se 15:13)
      for (int t = 0; t + 1 < 16 && h[(b * 16 + t + 1) * 16]; ++t) {
        const double dt = (double)(h[(b * 16 + t + 1) * 16] - h[(b * 16 + t) * 16]) * 10e-9, dc = (double)(h[(b * 16 + t + 1) * 16 + 8] - h[(b * 16 + t) * 16 + 8]);
        clk[t] += dc / dt * 1e-9; ++n[t];
      }
    }
    printf("# shader clock (GHz) over trip t:");
    for (int t = 0; t < 7; ++t) printf(" %.3f", n[t] ? clk[t] / n[t] : 0.0);
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second] += 1;
    printf("\n# blocks per (xcc, se, cu) id:");
    for (auto& kv : hist) printf("  %d ids with %d blocks", kv.second, kv.first);
    printf("\n");
  }
  // one block's raw timeline
  for (int64_t b = 0; b < blocks; b += 97) {
    if (h[b * 256] == 0) continue;
    printf("block %lld:", (long long)b);
    for (int t = 0; t < 16 && h[(b * 16 + t) * 16]; ++t) { printf(" |"); for (int i = 0; i < 8; ++i) printf(" %.2f", (double)(h[(b * 16 + t) * 16 + i] - first) * 0.01); }
    printf("\n");
    if (b > 300) break;
  }
  return 0;
}
