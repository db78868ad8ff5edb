// What would BASELINE config 3's step cost as the epilogue of the inverse plane transform?  The product source compiled with
// -DSKR_INV_STEP_MODEL: colored_inverse128 reads the step's other operands (seven 16-bit, two fp32 tensors: 22 B/element) at each finished plane's offsets,
// combines them with the noise in registers and writes an fp32 state and a 16-bit result (6 B/element) -- the bytes of the two-output UniPC-3 step, the noise
// itself never stored.  Timed: the whole 256 x (16,128,128) draw, to be set against the same draw without the model + the step kernel's own launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSKR_INV_STEP_MODEL -o tune_inverse128_step tune_inverse128_step.hip && ./tune_inverse128_step
//   (without -DSKR_INV_STEP_MODEL: the plain draw, for the difference)
#include "../../skrample_amd/csrc/skr_colored.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int64_t batch = argc > 1 ? atoi(argv[1]) : 256;
  const int d1 = 16, d2 = 128, d3 = 128, d3h = d3 / 2 + 1;
  const int64_t unit = (int64_t)d1 * d2 * d3, numel = batch * unit;
  void *out, *spec; float* scratch; double* partials; uint64_t* seeds;
  const int64_t slots = 64;
  CK(hipMalloc(&out, numel * 2));
  CK(hipMalloc(&spec, batch * d1 * d2 * d3h * 8));
  CK(hipMalloc(&scratch, 16));
  CK(hipMalloc(&partials, 4 * batch * slots * 8));
  CK(hipMalloc(&seeds, batch * 8));
  std::vector<uint64_t> hs(batch);
  for (int64_t i = 0; i < batch; ++i) hs[i] = 1000 + i;
  CK(hipMemcpy(seeds, hs.data(), batch * 8, hipMemcpyHostToDevice));
#ifdef SKR_INV_STEP_MODEL
  skr::StepModel m;
  for (int k = 0; k < 7; ++k) { void* p; CK(hipMalloc(&p, numel * 2)); CK(hipMemset(p, 0x3c, numel * 2)); m.narrow[k] = p; }
  for (int k = 0; k < 2; ++k) { void* p; CK(hipMalloc(&p, numel * 4)); CK(hipMemset(p, 0, numel * 4)); m.wide[k] = p; }
  { void* p; CK(hipMalloc(&p, numel * 4)); m.state = (float*)p; }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(skr::g_step_model), &m, sizeof(m)));
  printf("# model epilogue: 22 B/element read, 6 B/element written beside the transform's own 8.1 B/element of spectrum (%.0f MB in all)\n", (double)numel * 28 / 1e6 + (double)batch * d1 * d2 * d3h * 8 / 1e6);
#else
  printf("# plain draw (no model)\n");
#endif
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipEventRecord(e0));
    const int st = skr_noise_colored(out, SKR_BF16, spec, scratch, partials, slots, seeds, 256 * rep, batch, d1, d2, d3, 1.0, 0, 0.0, nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    if (st) { printf("skr_noise_colored -> %d\n", st); return 1; }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rep %d: %.1f us for the whole draw\n", rep, ms * 1e3);
  }
  return 0;
}
