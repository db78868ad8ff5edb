/* A plain C client of the drop-in boundary: no Python, no torch -- only include/skrample_hip.h, the HIP runtime for device memory,
 * and libskrample_hip.so.  It runs one fused step (out = c0*a + c1*b + c2*c over fp32 tensors, the shape of an Euler / DPM update),
 * one two-output step, and one draw of in-kernel Philox noise, and checks them against host arithmetic.
 * Built and run by tests/test_boundary.py::test_plain_c_client (GPU box):
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include tests/abi_client.c -L skrample_amd/csrc -lskrample_hip -L/opt/rocm/lib -lamdhip64 -lm */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skrample_hip.h"

#define N 8192 /* four whole 2048-element chunks: the one-trip kernels */
#define CHECK(call)                                                            \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d\n", #call, rc_); return 1; }     \
  } while (0)

int main(void) {
  if (skr_abi_version() != SKR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  float *a = malloc(sizeof(float) * N), *b = malloc(sizeof(float) * N), *c = malloc(sizeof(float) * N), *y = malloc(sizeof(float) * N), *y1 = malloc(sizeof(float) * N);
  for (int i = 0; i < N; ++i) { a[i] = sinf(0.37f * i); b[i] = cosf(0.11f * i) * 2.f; c[i] = (float)(i % 17) - 8.f; }
  float *da, *db, *dc, *dy, *dy1;
  uint64_t* dseeds;
  const uint64_t seeds[2] = {42u, 43u};
  CHECK(hipMalloc((void**)&da, sizeof(float) * N)); CHECK(hipMalloc((void**)&db, sizeof(float) * N)); CHECK(hipMalloc((void**)&dc, sizeof(float) * N));
  CHECK(hipMalloc((void**)&dy, sizeof(float) * N)); CHECK(hipMalloc((void**)&dy1, sizeof(float) * N)); CHECK(hipMalloc((void**)&dseeds, sizeof(seeds)));
  CHECK(hipMemcpy(da, a, sizeof(float) * N, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b, sizeof(float) * N, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, c, sizeof(float) * N, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dseeds, seeds, sizeof(seeds), hipMemcpyHostToDevice));

  skr_step_plan plan;
  memset(&plan, 0, sizeof(plan));
  plan.n_terms = 3; plan.n_group_a = 3; plan.dtype_a = SKR_F32; plan.dtype_b = SKR_F32; plan.out0_dtype = SKR_F32; plan.out1_dtype = SKR_NONE;
  plan.coef0[0] = 0.75; plan.coef0[1] = -0.5; plan.coef0[2] = 0.125; plan.sample_numel = N / 2;
  const void* inputs[3] = {da, db, dc};
  CHECK(skr_step_launch(&plan, inputs, dy, NULL, NULL, N, NULL));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(y, dy, sizeof(float) * N, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < N; ++i) {
    const double want = 0.75 * a[i] - 0.5 * b[i] + 0.125 * c[i];
    const double err = fabs(y[i] - want) / (fabs(want) + 1.0);
    if (err > worst) worst = err;
  }
  if (worst > 1e-6) { fprintf(stderr, "one-output step: relative error %g\n", worst); return 1; }

  /* two outputs (the UniPC / SPC shape): out0 = sum c0*in, out1 = chain*out0 + sum c1*in */
  plan.out1_dtype = SKR_F32; plan.chain = 0.25; plan.coef1[0] = 1.0; plan.coef1[1] = 0.0; plan.coef1[2] = -2.0;
  CHECK(skr_step_launch(&plan, inputs, dy, dy1, NULL, N, NULL));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(y, dy, sizeof(float) * N, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(y1, dy1, sizeof(float) * N, hipMemcpyDeviceToHost));
  for (int i = 0; i < N; ++i) {
    const double o0 = 0.75 * a[i] - 0.5 * b[i] + 0.125 * c[i], o1 = 0.25 * o0 + a[i] - 2.0 * c[i];
    if (fabs(y[i] - o0) > 1e-5 * (fabs(o0) + 1.0) || fabs(y1[i] - o1) > 1e-5 * (fabs(o1) + 1.0)) { fprintf(stderr, "two-output step: element %d\n", i); return 1; }
  }

  /* in-kernel Philox noise: out = zeta * N(seed of the sample, stream 5); per-sample moments, and the draw depends on the seed only */
  memset(&plan, 0, sizeof(plan));
  plan.n_terms = 0; plan.n_group_a = 0; plan.dtype_a = SKR_F32; plan.dtype_b = SKR_F32; plan.out0_dtype = SKR_F32; plan.out1_dtype = SKR_NONE;
  plan.noise_mode = 1; plan.zeta0 = 1.0; plan.stream0 = 5; plan.sample_numel = N / 2;
  CHECK(skr_step_launch(&plan, NULL, dy, NULL, dseeds, N, NULL));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(y, dy, sizeof(float) * N, hipMemcpyDeviceToHost));
  for (int s = 0; s < 2; ++s) {
    double m = 0, v = 0;
    for (int i = 0; i < N / 2; ++i) m += y[s * (N / 2) + i];
    m /= N / 2;
    for (int i = 0; i < N / 2; ++i) v += (y[s * (N / 2) + i] - m) * (y[s * (N / 2) + i] - m);
    v /= N / 2 - 1;
    if (fabs(m) > 0.06 || fabs(v - 1.0) > 0.08) { fprintf(stderr, "noise moments of sample %d: mean %g var %g\n", s, m, v); return 1; }
  }
  CHECK(skr_step_launch(&plan, NULL, dy1, NULL, dseeds, N / 2, NULL)); /* the first sample alone: same values as inside the batch */
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(y1, dy1, sizeof(float) * (N / 2), hipMemcpyDeviceToHost));
  if (memcmp(y, y1, sizeof(float) * (N / 2)) != 0) { fprintf(stderr, "noise depends on the batch it is drawn in\n"); return 1; }

  /* a step program: the plan handed over once, launched by handle with today's pointers and stream ids -- same bits as
   * skr_step_launch of the plan with those stream ids */
  {
    skr_program* prog = NULL;
    CHECK(skr_program_create(&plan, N, &prog));
    if (!prog) return 1;
    plan.stream0 = 9;
    CHECK(skr_step_launch(&plan, NULL, dy, NULL, dseeds, N, NULL));
    CHECK(skr_program_launch(prog, NULL, dy1, NULL, dseeds, 9, 0, NULL));
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(y, dy, sizeof(float) * N, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(y1, dy1, sizeof(float) * N, hipMemcpyDeviceToHost));
    if (memcmp(y, y1, sizeof(float) * N) != 0) { fprintf(stderr, "a program launch differs from the plan launch\n"); return 1; }
    if (skr_program_launch(NULL, NULL, dy, NULL, dseeds, 0, 0, NULL) != SKR_ERR_NULL) return 1;
    skr_program_destroy(prog);
    plan.n_terms = -1;
    if (skr_program_create(&plan, N, &prog) != SKR_ERR_TERMS || prog != NULL) return 1;
    plan.n_terms = 0;
  }

  /* argument checking happens before anything is launched */
  if (skr_step_launch(NULL, NULL, NULL, NULL, NULL, N, NULL) != SKR_ERR_NULL) return 1;
  printf("abi client ok: %s\n", skr_build_info());
  return 0;
}
