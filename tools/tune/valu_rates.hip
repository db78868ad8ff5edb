// Issue cost of the vector instructions the Colored / Pyramid generators are made of, measured on the chip (round 4).
// One wave (and, second column, two waves on the same SIMD) issues an unrolled stream of INDEPENDENT instructions of one
// kind; cycles come from s_memtime around the stream.  The numbers price the generators' "VALU roofline"
// (profiles/r04_colored_valu_roofline.json): minimum wave-cycles = sum over the instructions an algorithm cannot do without
// of count x issue cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 32, ITERS = 64;

// body: 8 independent destination registers, each instruction reads two sources that no instruction of the stream writes
#define STREAM(NAME, ASM, CLOB)                                                                                         \
  __global__ void NAME(uint64_t* out, float seedf, uint32_t seedu) {                                                    \
    float f0 = seedf, f1 = seedf + 1.f;                                                                                 \
    uint32_t u0 = seedu, u1 = seedu * 3u + 1u;                                                                          \
    uint64_t w0 = seedu, acc = 0;                                                                                       \
    double d0 = seedf, d1 = seedf * 2.0;                                                                                \
    (void)f0; (void)f1; (void)u0; (void)u1; (void)w0; (void)d0; (void)d1;                                               \
    __shared__ float lds[4096];                                                                                         \
    lds[threadIdx.x] = seedf;                                                                                           \
    __syncthreads();                                                                                                    \
    uint32_t laddr = (threadIdx.x & 63) * 8;                                                                            \
    (void)laddr;                                                                                                        \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                                                                \
      _Pragma("unroll") for (int k = 0; k < UNROLL / 8; ++k) { ASM }                                                    \
    }                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");                                                         \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                   \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0 + acc;                  \
  }

#define R8(I) I(v10) I(v11) I(v12) I(v13) I(v14) I(v15) I(v16) I(v17)
#define CLOB8 "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"
#define CLOB16 CLOB8, "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25"

#define OP2(op, a, b) asm volatile(op " v10, %0, %1\n" op " v11, %0, %1\n" op " v12, %0, %1\n" op " v13, %0, %1\n" op " v14, %0, %1\n" op " v15, %0, %1\n" op " v16, %0, %1\n" op " v17, %0, %1" :: "v"(a), "v"(b) : CLOB8);
#define OP1(op, a) asm volatile(op " v10, %0\n" op " v11, %0\n" op " v12, %0\n" op " v13, %0\n" op " v14, %0\n" op " v15, %0\n" op " v16, %0\n" op " v17, %0" :: "v"(a) : CLOB8);
#define OP3(op, a, b, c) asm volatile(op " v10, %0, %1, %2\n" op " v11, %0, %1, %2\n" op " v12, %0, %1, %2\n" op " v13, %0, %1, %2\n" op " v14, %0, %1, %2\n" op " v15, %0, %1, %2\n" op " v16, %0, %1, %2\n" op " v17, %0, %1, %2" :: "v"(a), "v"(b), "v"(c) : CLOB8);
// 64-bit destinations: register pairs
#define OP3W(op, a, b, c) asm volatile(op " v[10:11], %0, %1, %2\n" op " v[12:13], %0, %1, %2\n" op " v[14:15], %0, %1, %2\n" op " v[16:17], %0, %1, %2\n" op " v[18:19], %0, %1, %2\n" op " v[20:21], %0, %1, %2\n" op " v[22:23], %0, %1, %2\n" op " v[24:25], %0, %1, %2" :: "v"(a), "v"(b), "v"(c) : CLOB16);
#define OP2W(op, a, b) asm volatile(op " v[10:11], %0, %1\n" op " v[12:13], %0, %1\n" op " v[14:15], %0, %1\n" op " v[16:17], %0, %1\n" op " v[18:19], %0, %1\n" op " v[20:21], %0, %1\n" op " v[22:23], %0, %1\n" op " v[24:25], %0, %1" :: "v"(a), "v"(b) : CLOB16);
#define MAD64(a, b, c) asm volatile("v_mad_u64_u32 v[10:11], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[12:13], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[14:15], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[16:17], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[18:19], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[20:21], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[22:23], s[20:21], %0, %1, %2\nv_mad_u64_u32 v[24:25], s[20:21], %0, %1, %2" :: "v"(a), "v"(b), "v"(c) : CLOB16, "s20", "s21");
#define DSR(op, w) asm volatile(op " " w ", %0\n" op " " w ", %0 offset:512\n" op " " w ", %0 offset:1024\n" op " " w ", %0 offset:1536\n" op " " w ", %0 offset:2048\n" op " " w ", %0 offset:2560\n" op " " w ", %0 offset:3072\n" op " " w ", %0 offset:3584" :: "v"(laddr) : CLOB16);
#define DSW(op, src) asm volatile(op " %0, %1\n" op " %0, %1 offset:512\n" op " %0, %1 offset:1024\n" op " %0, %1 offset:1536\n" op " %0, %1 offset:2048\n" op " %0, %1 offset:2560\n" op " %0, %1 offset:3072\n" op " %0, %1 offset:3584" :: "v"(laddr), "v"(src) : "memory");

STREAM(k_fma, OP3("v_fma_f32", f0, f1, f0), )
STREAM(k_add, OP2("v_add_f32", f0, f1), )
STREAM(k_mul, OP2("v_mul_f32", f0, f1), )
STREAM(k_pk_fma, OP3W("v_pk_fma_f32", d0, d1, d0), )
STREAM(k_pk_add, OP2W("v_pk_add_f32", d0, d1), )
STREAM(k_pk_mul, OP2W("v_pk_mul_f32", d0, d1), )
STREAM(k_mov, OP1("v_mov_b32", f0), )
STREAM(k_xor, OP2("v_xor_b32", u0, u1), )
STREAM(k_addu, OP2("v_add_u32", u0, u1), )
STREAM(k_lshl_add, OP3("v_lshl_add_u32", u0, u1, u0), )
STREAM(k_mul_lo, OP2("v_mul_lo_u32", u0, u1), )
STREAM(k_mul_hi, OP2("v_mul_hi_u32", u0, u1), )
STREAM(k_mad64, MAD64(u0, u1, w0), )
STREAM(k_log, OP1("v_log_f32", f0), )
STREAM(k_exp, OP1("v_exp_f32", f0), )
STREAM(k_sqrt, OP1("v_sqrt_f32", f0), )
STREAM(k_sin, OP1("v_sin_f32", f0), )
STREAM(k_cos, OP1("v_cos_f32", f0), )
STREAM(k_rcp, OP1("v_rcp_f32", f0), )
STREAM(k_cvt_f32_u32, OP1("v_cvt_f32_u32", u0), )
STREAM(k_cvt_pk_bf16, OP2("v_cvt_pk_bf16_f32", f0, f1), )
STREAM(k_add_f64, OP2W("v_add_f64", d0, d1), )
STREAM(k_cndmask, OP2("v_cndmask_b32", f0, f1), )
STREAM(k_bfrev, OP1("v_bfrev_b32", u0), )
STREAM(k_ds_read_b64, DSR("ds_read_b64", "v[10:11]"), )
STREAM(k_ds_read_b128, DSR("ds_read_b128", "v[10:13]"), )
STREAM(k_ds_write_b64, DSW("ds_write_b64", d0), )

struct Case { const char* name; void (*fn)(uint64_t*, float, uint32_t); };

int main() {
  uint64_t* out;
  CK(hipMalloc(&out, 4096 * 8));
  const Case cases[] = {
      {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
      {"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshl_add},
      {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad64},
      {"v_log_f32", k_log}, {"v_exp_f32", k_exp}, {"v_sqrt_f32", k_sqrt}, {"v_sin_f32", k_sin}, {"v_cos_f32", k_cos}, {"v_rcp_f32", k_rcp},
      {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_pk_bf16_f32", k_cvt_pk_bf16}, {"v_add_f64", k_add_f64}, {"v_cndmask_b32", k_cndmask}, {"v_bfrev_b32", k_bfrev},
      {"ds_read_b64", k_ds_read_b64}, {"ds_read_b128", k_ds_read_b128}, {"ds_write_b64", k_ds_write_b64},
  };
  printf("# s_memtime ticks per wave-instruction (ticks of the 100 MHz reference clock are NOT shader cycles: the ratio to v_fma_f32 = 4 shader cycles is what counts)\n");
  printf("# %-20s %12s %12s %12s   %s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "cost relative to v_fma_f32 (1 wave | 2 waves | 4 waves)");
  double base[3] = {0, 0, 0};
  for (const Case& c : cases) {
    double per[3];
    for (int w = 0; w < 3; ++w) {
      const int waves_per_simd = 1 << w;
      const int threads = 64 * 4 * waves_per_simd;  // one block on one CU: 4 SIMDs x waves_per_simd
      std::vector<uint64_t> h(threads / 64);
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(c.fn, dim3(1), dim3(threads), 0, 0, out, 1.5f, 12345u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (uint64_t v : h) worst = worst > (double)v ? worst : (double)v;
        best = best < worst ? best : worst;
      }
      per[w] = best / ((double)UNROLL * ITERS) / waves_per_simd;  // ticks per wave-instruction of the SIMD's combined stream
    }
    if (base[0] == 0) { base[0] = per[0]; base[1] = per[1]; base[2] = per[2]; }
    printf("  %-20s %12.4f %12.4f %12.4f   %.2f | %.2f | %.2f\n", c.name, per[0], per[1], per[2], per[0] / base[0], per[1] / base[1], per[2] / base[2]);
  }
  return 0;
}
