// Philox4x32-10 + Box-Muller for gfx950.  Specification: oracle/skr_oracle/noise.py
// (philox4x32 / box_muller / philox_normal), which restates Salmon et al. SC'11 / Random123.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skr {

struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)c.x * M0;
    const uint64_t p1 = (uint64_t)c.z * M1;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// u = x * 2^-32 + 2^-33  in (0, 1]: the product is exact, the sum rounds once (fused or not), and the largest
// input gives exactly 1.0f (2^32 * 2^-32 + 2^-33 rounds to 1), so no clamp is needed; the smallest gives 2^-33.
__device__ __forceinline__ float u01(uint32_t x) {
  return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// (r cos 2*pi*u1, r sin 2*pi*u1), r = sqrt(-2 ln u0).
// Raw hardware transcendentals: v_log_f32 is log2 (so -2 ln u0 = log2(u0) * (-2 ln 2), one multiply), v_sqrt_f32,
// v_sin_f32 / v_cos_f32 take their argument in revolutions (the 2*pi never materialises).  Their operands are never
// denormal here (u0 >= 2^-33, and -2 ln u0 is 0 or >= 1.1e-7), so the library's denormal pre-scaling and the
// Newton fix-up of the correctly rounded sqrt -- together ~100 of the ~270 instructions of two blocks -- are not
// needed; each is accurate to 1 ulp, well inside the 2e-6 the oracle comparison allows for the hardware sin/cos.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u0 = u01(a), u1 = u01(b);
  const float r = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u0) * -1.3862943611198906f);
  z0 = r * __builtin_amdgcn_cosf(u1);
  z1 = r * __builtin_amdgcn_sinf(u1);
}

// four normals of Philox block `blk` of (seed, stream)
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t stream, uint64_t blk, float z[4]) {
  u32x4 c{(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  c = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  box_muller(c.x, c.y, z[0], z[1]);
  box_muller(c.z, c.w, z[2], z[3]);
}

}  // namespace skr
