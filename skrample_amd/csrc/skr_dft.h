// Small complex transforms in registers (natural order in and out): the short outer axes of the Colored generators and the
// first LDS pass of their tile transform (skr_colored.hip, skr_colored_any.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace skr {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <bool INV> __device__ __forceinline__ float2 mul_i(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <bool INV> __device__ __forceinline__ void dft4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = mul_i<INV>(csub(v1, v3));
  v0 = cadd(a, c); v2 = csub(a, c); v1 = cadd(b, d); v3 = csub(b, d);
}
template <bool INV> __device__ __forceinline__ float2 rot8(float2 a) {  // a * exp(-+ i pi/4)
  constexpr float r = 0.70710678118654752440f;
  return INV ? make_float2(r * (a.x - a.y), r * (a.x + a.y)) : make_float2(r * (a.x + a.y), r * (a.y - a.x));
}
template <bool INV> __device__ __forceinline__ void dft8(float2 v[8]) {
  float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4<INV>(e0, e1, e2, e3);
  dft4<INV>(o0, o1, o2, o3);
  o1 = rot8<INV>(o1); o2 = mul_i<INV>(o2); o3 = mul_i<INV>(rot8<INV>(o3));
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0); v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2); v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <bool INV> __device__ __forceinline__ void dft16(float2 v[16]) {
  float2 e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  dft8<INV>(e);
  dft8<INV>(o);
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  const float sg = INV ? 1.f : -1.f;
  const float2 w[8] = {{1.f, 0.f}, {c1, sg * s1}, {h, sg * h}, {s1, sg * c1}, {0.f, sg}, {-s1, sg * c1}, {-h, sg * h}, {-c1, sg * s1}};
#pragma unroll
  for (int k = 0; k < 8; ++k) { const float2 t = cmul(o[k], w[k]); v[k] = cadd(e[k], t); v[k + 8] = csub(e[k], t); }
}
template <int N, bool INV> __device__ __forceinline__ void dft_n(float2 v[N]) {
  if constexpr (N == 16) dft16<INV>(v);
  else if constexpr (N == 8) dft8<INV>(v);
  else if constexpr (N == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
  else { const float2 t = v[0]; v[0] = cadd(t, v[1]); v[1] = csub(t, v[1]); }
}

}  // namespace skr
