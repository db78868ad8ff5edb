// One-trip compile-time kernels of the fused solver step for MI355X (gfx950): whole 2048-element chunks, 1-D grid,
// XCD-aware chunk map, paced load issue.  Bit-identical to the grid-stride / general kernels of skr_step.hip.
#include "skr_step_common.h"

namespace skr {

// ---- one-trip kernels: loads first, XCD-aware chunk map ------------------------------------------------
// Launches made of whole 2048-element chunks (every BASELINE config) take these.  One workgroup = one chunk of
// BLOCK lane-vectors, one trip, exactly numel/2048 workgroups on a 1-D grid:
//  * the operand pointers are the first thing read from the kernarg and the K loads are issued right behind them;
//    everything else the wave needs (seed, Philox key schedule, coefficients) is fetched while they are in flight.
//    (The grid-stride kernels above read geometry -> seed pointer -> seed -> key schedule before their first load:
//    three dependent scalar round trips per wave, 0.7 us on the 26 us headline launch.)
//  * chunk map: workgroups b and b+8 run on the same XCD (round-robin dispatch), so with the identity map every XCD
//    walks the tensor with a stride of 8 chunks.  The map hands each XCD runs of 2^lr consecutive chunks inside every
//    group of 8 runs: -0.2..0.3 us on the headline launch (tools/tune/tune_r2.hip).
// Arithmetic, lane ownership and Philox block numbering are exactly those of step_kernel_k / step_kernel_rk, so the
// results are bit-identical (tests/test_step_gpu.py::test_one_trip_kernels_agree_bitwise).
// Kernarg of the one-trip kernel: what the first instructions need (pointers, chunk map) leads, and launches of <= 4
// operands carry a 2-line block instead of 3 (every CU's scalar cache misses each line once per launch).
template <int KMAX>
struct OneTripArgs {
  const void* in[KMAX];
  void* out0;
  const uint64_t* seeds;
  int32_t xmap_lr;      // log2(run length) of the XCD chunk map
  int32_t bps_shift;    // log2(blocks per sample)
  uint64_t stream0;
  float c0[KMAX];
  float zeta0;
  RowRef tab;
};

// chunk -> (sample, chunk within the sample).  bps_shift >= 0: a power-of-two number of chunks per sample (shift);
// bps_shift < 0: -bps_shift chunks per sample (e.g. 18 for 4x96x96 latents), one uniform integer division per wave.
__device__ __forceinline__ void sample_of(uint32_t c, int32_t bps_shift, uint32_t& smp, uint32_t& within) {
  if (bps_shift >= 0) { smp = c >> bps_shift; within = c - (smp << bps_shift); }
  else { const uint32_t bps = (uint32_t)(-bps_shift); smp = c / bps; within = c - smp * bps; }
}

__device__ __forceinline__ uint32_t chunk_of(uint32_t b, int lr) {  // lr = log2(run length); 0 = identity
  const uint32_t g = 3 + lr;
  return ((b >> g) << g) + ((b & 7u) << lr) + ((b >> 3) & ((1u << lr) - 1u));
}

// kernarg sizes: 4 / 8 / 12 / 16 / 20 operand slots (Adams-Bashforth 5-9 and UniP >= 5 give 10-18 operands: round 3)
constexpr int one_trip_kmax(int k) { return k <= 4 ? 4 : (k <= 8 ? 8 : (k <= 12 ? 12 : (k <= 16 ? 16 : 20))); }
constexpr int ONE_TRIP_MAX_K = 20;

template <typename T, int K, bool NOISE, bool TILE, bool PACE, bool TAB>
__global__ __launch_bounds__(BLOCK) void step_kernel_k1(const OneTripArgs<one_trip_kmax(K)> a) {
  const uint32_t c = chunk_of(blockIdx.x, a.xmap_lr);
  const int64_t v = (int64_t)c * BLOCK + threadIdx.x;
  Raw<T> raw[K];
  float z[VEC];
  // The step's scalars come from the kernarg or, in the TAB instantiation (indexed launches), from the device-resident row
  // (two dependent scalar loads).  They are fetched AFTER the first global loads have been issued, and the choice is a
  // template parameter: as a run-time branch in front it put three scalar round trips before the first load of every wave
  // (+0.3 us on the headline launch), and the compiler sank the loads behind the branch wherever it stood.
  float cf[K], zeta0;
  uint64_t stream0;
#define SKR_SCALARS()                                                        \
  __builtin_amdgcn_sched_barrier(0);                                         \
  zeta0 = a.zeta0; stream0 = a.stream0;                                      \
  _Pragma("unroll") for (int j = 0; j < K; ++j) cf[j] = a.c0[j];             \
  if constexpr (TAB) {                                                       \
    const skr_step_row* row = row_of(a.tab);                                 \
    _Pragma("unroll") for (int j = 0; j < K; ++j) cf[j] = (float)row->coef0[j]; \
    zeta0 = (float)row->zeta0;                                               \
    stream0 = row->stream0;                                                  \
  }
  // Paced issue.  One burst of K loads per wave is not the fastest order on this memory system: on the headline
  // launch (tools/tune/tune_r2.hip, 256x4x128x128 bf16, K = 4) all loads first runs 26.3 us, loads after the Philox
  // set-up 27.0 us, and the loads spread over the wave's Philox work -- one before the seed fetch, one after it, one
  // after each Philox block -- 25.9 us; without noise, ~1000 idle clocks (s_sleep 16) between the loads of a
  // 4-operand launch give 25.6 instead of 26.1 us (no gain measured for 2-output or 7/8-operand launches, which stay
  // unpaced).  The order is pinned by data dependencies: each later load takes its lane-vector index from an empty
  // asm statement that sits behind the work it has to follow (volatile asm statements keep their order), and
  // sched_barrier stops the machine scheduler from regrouping the segments.
  if constexpr (NOISE && PACE) {
    int64_t vj = v;
#define SKR_ISSUE(SLOT)                                                                      \
    _Pragma("unroll") for (int j = 0; j < K; ++j)                                            \
      if ((j * 4) / K == SLOT) raw[j] = load_raw<T, TILE>(a.in[j], vj);                      \
    __builtin_amdgcn_sched_barrier(0)
    SKR_ISSUE(0);
    SKR_ISSUE(1);  // (both before the wave parks on its scalar fetches: holding the second one back behind them cost 0.3 us)
    SKR_SCALARS();
    uint32_t smp, within;
    sample_of(c, a.bps_shift, smp, within);
    const uint64_t seed = a.seeds[smp];
    uint32_t vs = within * BLOCK + threadIdx.x;  // lane-vector within the sample
    normal4(seed, stream0, (uint64_t)group0<TILE>((int64_t)vs), z);
    asm volatile("" : "+v"(vj), "+v"(vs) : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]));  // ... behind the first block
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(2);
    normal4(seed, stream0, (uint64_t)group1<TILE>((int64_t)vs), z + 4);
    asm volatile("" : "+v"(vj) : "v"(z[4]), "v"(z[5]), "v"(z[6]), "v"(z[7]));           // ... behind the second
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(3);
#undef SKR_ISSUE
    // the operands are first touched here: left alone, the compiler starts unpacking the early ones between the
    // segments and parks the wave on their arrival (microseconds under load) before the later loads are issued
#pragma unroll
    for (int j = 0; j < K; ++j) pin_raw(raw[j]);
  } else if constexpr (NOISE) {  // unpaced: every load first, then the Philox work
#pragma unroll
    for (int j = 0; j < K; ++j) raw[j] = load_raw<T, TILE>(a.in[j], v);
    SKR_SCALARS();
    uint32_t smp, within;
    sample_of(c, a.bps_shift, smp, within);
    const uint64_t seed = a.seeds[smp];
    const uint32_t vs = within * BLOCK + threadIdx.x;
    normal4(seed, stream0, (uint64_t)group0<TILE>((int64_t)vs), z);
    normal4(seed, stream0, (uint64_t)group1<TILE>((int64_t)vs), z + 4);
  } else if constexpr (PACE && (K == 4 || K == 5)) {  // tools/bench_plan.py: K=4 -1.1 %, K=5 -1.7 %, K=3 +6 % (left unpaced), K>=6 no change
    int64_t vj = v;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      raw[j] = load_raw<T, TILE>(a.in[j], vj);
      if (j < K - 1) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_sleep 16" : "+v"(vj));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    SKR_SCALARS();
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) raw[j] = load_raw<T, TILE>(a.in[j], v);
    SKR_SCALARS();
  }
#undef SKR_SCALARS
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    float w[VEC];
    widen<T, float>(raw[j], w);
    const float cj = cf[j];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = fma_(cj, w[i], s[i]);
  }
  if constexpr (NOISE) { if (zeta0 != 0.f) fma_noise8<float>(zeta0, z, s); }  // (a zero row skips the draw, as a launch without noise does)
  store8<T, float, TILE>(a.out0, v, s);
}

// Kernarg of the one-trip Runge-Kutta stage kernel: as for step_kernel_k1, everything the first instructions need -- the
// operand pointers, the chunk map, both output pointers -- leads, and stages of <= 4 operands have all of it in the first
// 64-byte line (round 2's RkArgs kept xmap_lr in the third line, behind the coefficients: the first load of every wave
// waited for two scalar lines instead of one; tools/tune/tune_r3.hip "lib": K=2 24.05 -> 23.1 us, K=5 38.97 -> 37.8 us).
template <int KMAX>
struct RkOneTripArgs {
  const void* in[KMAX];
  void* out0;
  void* out1;
  const uint64_t* seeds;
  int32_t xmap_lr;
  int32_t bps_shift;
  float c1[KMAX];
  float chain;
  float ck[4];
  int32_t conv_to, conv_from;
  float zeta1;
  uint64_t stream1;
  RowRef tab;
};

// NOISE: the step's last stage of a stochastic tableau step adds zeta1 * N(stream1) to out1 (the derivative out0 is never noisy)
// BLK: threads per workgroup (256, or 128 = 1024-element chunks: tools/tune/tune_r3.hip measured the 5-operand stage 1-4 % faster so)
template <typename T, int K, bool TILE, bool NOISE, bool TAB, int BLK>
__global__ __launch_bounds__(BLK) void step_kernel_rk1(const RkOneTripArgs<(K <= 4 ? 4 : 8)> a) {
  const uint32_t c = chunk_of(blockIdx.x, a.xmap_lr);
  const int64_t v = (int64_t)c * BLK + threadIdx.x;
  Raw<T> raw[K];
#pragma unroll
  for (int j = 0; j < K; ++j) raw[j] = load_raw<T, TILE>(a.in[j], v);
  __builtin_amdgcn_sched_barrier(0);  // every load is out before the first scalar of the arithmetic is fetched
  float k[4] = {a.ck[0], a.ck[1], a.ck[2], a.ck[3]}, cf[K], chain = a.chain, zeta1 = a.zeta1;
  uint64_t stream1 = a.stream1;
#pragma unroll
  for (int j = 0; j < K; ++j) cf[j] = a.c1[j];
  if constexpr (TAB) {
    const skr_step_row* row = row_of(a.tab);
#pragma unroll
    for (int j = 0; j < K; ++j) cf[j] = (float)row->coef1[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = (float)row->convert_k[i];
    chain = (float)row->chain;
    zeta1 = (float)row->zeta1;
    stream1 = row->stream1;
  }
  float z1[VEC];
  bool n1 = false;
  if constexpr (NOISE) {
    n1 = zeta1 != 0.f;
    if (n1) {
      uint32_t smp, within;
      sample_of(c, a.bps_shift, smp, within);
      const uint64_t seed = a.seeds[smp];
      const uint32_t vs = within * BLK + threadIdx.x;
      normal4(seed, stream1, (uint64_t)group0<TILE>((int64_t)vs), z1);
      normal4(seed, stream1, (uint64_t)group1<TILE>((int64_t)vs), z1 + 4);
    }
  }
  float sv[VEC], ov[VEC], d[VEC], s1[VEC];
  widen<T, float>(raw[0], sv);
  widen<T, float>(raw[1], ov);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    d[i] = convert_rounded<T, float>(sv[i], ov[i], a.conv_to, a.conv_from, k);
    s1[i] = fma_(cf[1], ov[i], fma_(cf[0], sv[i], 0.f));
  }
#pragma unroll
  for (int j = 2; j < K; ++j) {
    float w[VEC];
    widen<T, float>(raw[j], w);
    const float cj = cf[j];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = fma_(cj, w[i], s1[i]);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = fma_(chain, d[i], s1[i]);
  if constexpr (NOISE) { if (n1) fma_noise8<float>(zeta1, z1, s1); }
  store8<T, float, TILE>(a.out1, v, s1);
  store8<T, float, TILE>(a.out0, v, d);
}

// one-trip launches: whole chunks, and with in-kernel noise samples made of whole chunks (any number of them)
static bool one_trip_ok(int64_t numel, int64_t sample_numel, bool noise, int* bps_shift, bool forced = false) {
  constexpr int64_t CHUNK = (int64_t)BLOCK * VEC;
  if ((!g_tune.one_trip && !forced) || numel % CHUNK != 0 || numel / CHUNK > 0x7fffffffll) return false;
  *bps_shift = 0;
  if (!noise) return true;
  if (sample_numel % CHUNK != 0) return false;
  const int64_t bps = sample_numel / CHUNK;
  if (bps > 0x3fffffffll) return false;
  if ((bps & (bps - 1)) == 0) { while ((1ll << *bps_shift) < bps) ++*bps_shift; }
  else *bps_shift = -(int)bps;  // any chunk count per sample: the kernel divides
  return true;
}

// run length of the XCD chunk map: the largest power of two <= the tuned one whose group of 8 runs divides the grid
static int xmap_lr_for(int64_t chunks) {
  int lr = g_tune.xmap;
  if (lr < 0) lr = 0;
  if (lr > 20) lr = 20;
  while (lr > 0 && chunks % (8ll << lr) != 0) --lr;
  return lr;
}

// ---- host side ---------------------------------------------------------------------------------------------------
template <typename T, bool NOISE, int KMAX>
static int launch_k1(const StepArgs<float>& args, int bps_shift, hipStream_t stream) {
  constexpr bool TILE = sizeof(T) == 4;  // whole chunks are whole tiles
  const int64_t chunks = args.numel / ((int64_t)BLOCK * VEC);
  OneTripArgs<KMAX> fa;
  for (int k = 0; k < KMAX; ++k) { fa.in[k] = k < args.n_terms ? args.in[k] : nullptr; fa.c0[k] = k < args.n_terms ? args.c0[k] : 0.f; }
  fa.out0 = args.out0; fa.seeds = args.seeds; fa.zeta0 = args.zeta0; fa.stream0 = args.stream0;
  fa.bps_shift = bps_shift; fa.xmap_lr = xmap_lr_for(chunks);
  fa.tab = RowRef{args.rows, args.index, args.row_offset};
#define SKR_K(N) case N: if (args.rows != nullptr) hipLaunchKernelGGL((step_kernel_k1<T, N, NOISE, TILE, true, true>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, fa); \
                        else if (g_tune.pace) hipLaunchKernelGGL((step_kernel_k1<T, N, NOISE, TILE, true, false>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, fa); \
                        else hipLaunchKernelGGL((step_kernel_k1<T, N, NOISE, TILE, false, false>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, fa); break
  // more than 8 operands: one unpaced instantiation each (plus the table form while the operands fit a device-resident row)
#define SKR_KB(N) case N: if (args.rows != nullptr) { if constexpr (N <= SKR_ROW_TERMS) hipLaunchKernelGGL((step_kernel_k1<T, N, NOISE, TILE, false, true>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, fa); else return SKR_ERR_UNSUPPORTED; } \
                         else hipLaunchKernelGGL((step_kernel_k1<T, N, NOISE, TILE, false, false>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, fa); break
  if constexpr (KMAX == 4) { switch (args.n_terms) { SKR_K(1); SKR_K(2); SKR_K(3); SKR_K(4); } }
  else if constexpr (KMAX == 8) { switch (args.n_terms) { SKR_K(5); SKR_K(6); SKR_K(7); SKR_K(8); } }
  else if constexpr (KMAX == 12) { switch (args.n_terms) { SKR_KB(9); SKR_KB(10); SKR_KB(11); SKR_KB(12); } }
  else if constexpr (KMAX == 16) { switch (args.n_terms) { SKR_KB(13); SKR_KB(14); SKR_KB(15); SKR_KB(16); } }
  else { switch (args.n_terms) { SKR_KB(17); SKR_KB(18); SKR_KB(19); SKR_KB(20); } }
#undef SKR_K
#undef SKR_KB
  return finish_launch();
}


template <typename T>
int launch_one_trip_k(const StepArgs<float>& args, bool noise, hipStream_t stream, bool& taken) {
  taken = false;
  int bps_shift = 0;
  if (!one_trip_ok(args.numel, args.sample_numel, noise, &bps_shift, args.rows != nullptr) || (sizeof(T) == 4 && !g_tune.tile && args.rows == nullptr)) return SKR_OK;
  if (args.n_terms > ONE_TRIP_MAX_K) return SKR_OK;
  taken = true;
#define SKR_GO(NOISE)                                                              \
  switch (one_trip_kmax(args.n_terms)) {                                           \
    case 4: return launch_k1<T, NOISE, 4>(args, bps_shift, stream);                \
    case 8: return launch_k1<T, NOISE, 8>(args, bps_shift, stream);                \
    case 12: return launch_k1<T, NOISE, 12>(args, bps_shift, stream);              \
    case 16: return launch_k1<T, NOISE, 16>(args, bps_shift, stream);              \
    default: return launch_k1<T, NOISE, 20>(args, bps_shift, stream);              \
  }
  if (noise) { SKR_GO(true) }
  SKR_GO(false)
#undef SKR_GO
}
template int launch_one_trip_k<bf16_t>(const StepArgs<float>&, bool, hipStream_t, bool&);
template int launch_one_trip_k<f16_t>(const StepArgs<float>&, bool, hipStream_t, bool&);
template int launch_one_trip_k<float>(const StepArgs<float>&, bool, hipStream_t, bool&);

template <typename T, bool NOISE, int KMAX, int BLK>
static int launch_rk1(const StepArgs<float>& args, unsigned chunks, int bps_shift, hipStream_t stream) {
  constexpr bool TILE = sizeof(T) == 4;
  if constexpr (BLK == 128) {  // half-size chunks: twice as many of them, per sample too
    chunks *= 2;
    bps_shift = bps_shift >= 0 ? bps_shift + 1 : 2 * bps_shift;
  }
  RkOneTripArgs<KMAX> ra;
  ra.seeds = args.seeds; ra.zeta1 = args.zeta1; ra.stream1 = args.stream1; ra.bps_shift = bps_shift;
  for (int k = 0; k < KMAX; ++k) { ra.in[k] = k < args.n_terms ? args.in[k] : nullptr; ra.c1[k] = k < args.n_terms ? args.c1[k] : 0.f; }
  ra.out0 = args.out0; ra.out1 = args.out1; ra.chain = args.chain;
  for (int i = 0; i < 4; ++i) ra.ck[i] = (float)args.ck[i];
  ra.conv_to = args.conv_to; ra.conv_from = args.conv_from; ra.xmap_lr = xmap_lr_for(chunks); ra.tab = RowRef{args.rows, args.index, args.row_offset};
#define SKR_K(N) case N: if (args.rows != nullptr) hipLaunchKernelGGL((step_kernel_rk1<T, N, TILE, NOISE, true, BLK>), dim3(chunks), dim3(BLK), 0, stream, ra); \
                        else hipLaunchKernelGGL((step_kernel_rk1<T, N, TILE, NOISE, false, BLK>), dim3(chunks), dim3(BLK), 0, stream, ra); break
  if constexpr (KMAX == 4) { switch (args.n_terms) { SKR_K(2); SKR_K(3); SKR_K(4); } }
  else { switch (args.n_terms) { SKR_K(5); SKR_K(6); SKR_K(7); SKR_K(8); } }
#undef SKR_K
  return finish_launch();
}


template <typename T>
int launch_one_trip_rk(const StepArgs<float>& args, bool noise, hipStream_t stream, bool& taken) {
  taken = false;
  int bps_shift = 0;
  if (!one_trip_ok(args.numel, args.sample_numel, noise, &bps_shift, args.rows != nullptr) || (sizeof(T) == 4 && !g_tune.tile && args.rows == nullptr)) return SKR_OK;
  if (noise && args.rows == nullptr && args.zeta0 != 0.f) return SKR_OK;  // (a noisy derivative does not occur; left to the general kernel)
  taken = true;
  const unsigned chunks = (unsigned)(args.numel / ((int64_t)BLOCK * VEC));
  // 128-thread workgroups for the 2-6 operand stages (tools/bench_plan.py rk, same box: K=4 33.5 vs 34.0 us, K=5 38.7 vs 39.8, K=6 43.7 vs
  // 45.2, K >= 7 unchanged; round 4, A/B/A/B on one box, profiles/r04_bench_plan_ab.txt: K=2 23.6-23.8 vs 24.1-24.2 us, K=3 29.16-29.26 vs
  // 29.17-29.30 -- round 3 had read K=2 / 3 as 1-2 % slower from runs on different boxes); rk_blk = 128 / 256 forces one size
  const bool small_blocks = g_tune.rk_blk == 128 || (g_tune.rk_blk == 0 && args.n_terms <= 6);
  if (small_blocks && chunks < 0x40000000u && bps_shift > -0x20000000) {
    if (args.n_terms <= 4) return noise ? launch_rk1<T, true, 4, 128>(args, chunks, bps_shift, stream) : launch_rk1<T, false, 4, 128>(args, chunks, bps_shift, stream);
    return noise ? launch_rk1<T, true, 8, 128>(args, chunks, bps_shift, stream) : launch_rk1<T, false, 8, 128>(args, chunks, bps_shift, stream);
  }
  if (args.n_terms <= 4) return noise ? launch_rk1<T, true, 4, 256>(args, chunks, bps_shift, stream) : launch_rk1<T, false, 4, 256>(args, chunks, bps_shift, stream);
  return noise ? launch_rk1<T, true, 8, 256>(args, chunks, bps_shift, stream) : launch_rk1<T, false, 8, 256>(args, chunks, bps_shift, stream);
}
template int launch_one_trip_rk<bf16_t>(const StepArgs<float>&, bool, hipStream_t, bool&);
template int launch_one_trip_rk<f16_t>(const StepArgs<float>&, bool, hipStream_t, bool&);
template int launch_one_trip_rk<float>(const StepArgs<float>&, bool, hipStream_t, bool&);

// ---- two outputs (UniPC / SPC steps): out0 fp32 state, out1 = chain*out0 + ... in the operands' 16-bit dtype ---------
//   out0 = sum_k c0[k]*in_k + zeta0*N(stream0)          NA 16-bit operands, then NB (0 or 1) fp32 operand
//   out1 = chain*out0 + sum_k c1[k]*in_k + zeta1*N(stream1)
// Same FMA order, zero-zeta guards, tile layout and Philox numbering as step_kernel<..., HAS1, TILE = true>, so the bits
// are the same; what changes is that every pointer and coefficient is in SGPRs before the first load, the term list is
// a template constant and the launch is one trip over an XCD-mapped 1-D grid.
template <int NMAX>
struct TwoOutArgs {
  const void* in[NMAX];
  void* out0;
  void* out1;
  const uint64_t* seeds;
  int32_t xmap_lr, bps_shift;
  uint64_t stream0, stream1;
  float chain, zeta0, zeta1;
  float c0[NMAX];
  float c1[NMAX];
  RowRef tab;
};
// operand count from which the no-noise two-output launches store non-temporally.  Round-4 A/B/A/B through the library on one box
// (tools/bench_plan.py ab, profiles/r04_bench_plan_ab.txt): 8 + 1 operands 294.0-295.1 us vs 299.4-299.9 us with write-through stores,
// 10 + 1 (BASELINE config 3 with Colored noise tensors) 338.2-338.5 vs 346.0-346.1 us
constexpr int NT_FROM_OPERANDS = 9;
constexpr int two_out_nmax(int n) { return n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 12 ? 12 : (n <= 16 ? 16 : (n <= 20 ? 20 : 24)))); }

template <typename TA, int NA, int NB, bool NOISE, bool PACE, bool TAB, bool NT = false>
__global__ __launch_bounds__(BLOCK) void step_kernel_k2(const TwoOutArgs<two_out_nmax(NA + NB)> a) {
  const uint32_t c = chunk_of(blockIdx.x, a.xmap_lr);
  const int64_t v = (int64_t)c * BLOCK + threadIdx.x;
  Raw<TA> ra[NA];
  Raw<float> rb[NB > 0 ? NB : 1];
  float z0[VEC], z1[VEC];
  bool n0 = false, n1 = false;
  float cf0[NA + NB], cf1[NA + NB], chain, zeta0, zeta1;  // kernarg or device-resident row, fetched behind the first loads
  uint64_t stream0, stream1;
#define SKR_SCALARS()                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                       \
  chain = a.chain; zeta0 = a.zeta0; zeta1 = a.zeta1; stream0 = a.stream0; stream1 = a.stream1; \
  _Pragma("unroll") for (int j = 0; j < NA + NB; ++j) { cf0[j] = a.c0[j]; cf1[j] = a.c1[j]; } \
  if constexpr (TAB) {                                                                     \
    const skr_step_row* row = row_of(a.tab);                                               \
    _Pragma("unroll") for (int j = 0; j < NA + NB; ++j) { cf0[j] = (float)row->coef0[j]; cf1[j] = (float)row->coef1[j]; } \
    chain = (float)row->chain; zeta0 = (float)row->zeta0; zeta1 = (float)row->zeta1;       \
    stream0 = row->stream0; stream1 = row->stream1;                                        \
  }
  if constexpr (NOISE && PACE) {
    // loads paced over the Philox work (see step_kernel_k1): six slots around the seed fetch and the four blocks
    int64_t vj = v;
#define SKR_ISSUE(SLOT)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NA; ++j)                                                        \
      if ((j * 6) / (NA + NB) == SLOT) ra[j] = load_raw<TA, true>(a.in[j], vj);                           \
    _Pragma("unroll") for (int j = 0; j < NB; ++j)                                                        \
      if (((NA + j) * 6) / (NA + NB) == SLOT) rb[j] = load_raw<float, true>(a.in[NA + j], vj);            \
    __builtin_amdgcn_sched_barrier(0)
    SKR_ISSUE(0);
    SKR_ISSUE(1);
    SKR_SCALARS();
    uint32_t smp, within;
    sample_of(c, a.bps_shift, smp, within);
    const uint64_t seed = a.seeds[smp];
    uint32_t vs = within * BLOCK + threadIdx.x;
    n0 = zeta0 != 0.f;
    n1 = zeta1 != 0.f;
    if (n0) normal4(seed, stream0, (uint64_t)group0<true>((int64_t)vs), z0);
    asm volatile("" : "+v"(vj), "+v"(vs));
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(2);
    if (n0) normal4(seed, stream0, (uint64_t)group1<true>((int64_t)vs), z0 + 4);
    asm volatile("" : "+v"(vj), "+v"(vs));
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(3);
    if (n1) normal4(seed, stream1, (uint64_t)group0<true>((int64_t)vs), z1);
    asm volatile("" : "+v"(vj), "+v"(vs));
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(4);
    if (n1) normal4(seed, stream1, (uint64_t)group1<true>((int64_t)vs), z1 + 4);
    asm volatile("" : "+v"(vj));
    __builtin_amdgcn_sched_barrier(0);
    SKR_ISSUE(5);
#undef SKR_ISSUE
#pragma unroll
    for (int j = 0; j < NA; ++j) pin_raw(ra[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) pin_raw(rb[j]);
  } else {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = load_raw<TA, true>(a.in[j], v);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = load_raw<float, true>(a.in[NA + j], v);
    SKR_SCALARS();
    if constexpr (NOISE) {
      uint32_t smp, within;
      sample_of(c, a.bps_shift, smp, within);
      const uint64_t seed = a.seeds[smp];
      const uint32_t vs = within * BLOCK + threadIdx.x;
      n0 = zeta0 != 0.f;
      n1 = zeta1 != 0.f;
      if (n0) { normal4(seed, stream0, (uint64_t)group0<true>((int64_t)vs), z0); normal4(seed, stream0, (uint64_t)group1<true>((int64_t)vs), z0 + 4); }
      if (n1) { normal4(seed, stream1, (uint64_t)group0<true>((int64_t)vs), z1); normal4(seed, stream1, (uint64_t)group1<true>((int64_t)vs), z1 + 4); }
    }
  }
#undef SKR_SCALARS
  float s0[VEC], s1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    float w[VEC];
    widen<TA, float>(ra[j], w);
    const float w0 = cf0[j], w1 = cf1[j];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s0[i] = fma_(w0, w[i], s0[i]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = fma_(w1, w[i], s1[i]);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float w[VEC];
    widen<float, float>(rb[j], w);
    const float w0 = cf0[NA + j], w1 = cf1[NA + j];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s0[i] = fma_(w0, w[i], s0[i]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = fma_(w1, w[i], s1[i]);
  }
  if constexpr (NOISE) { if (n0) fma_noise8<float>(zeta0, z0, s0); }
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = fma_(chain, s0[i], s1[i]);
  if constexpr (NOISE) { if (n1) fma_noise8<float>(zeta1, z1, s1); }
  store8<TA, float, true, NT>(a.out1, v, s1);
  store8<float, float, true, NT>(a.out0, v, s0);
}

template <typename TA, int NA, int NB, bool NOISE, bool PACE, bool TAB, bool NT = false>
static int launch_k2(const StepArgs<float>& args, int bps_shift, hipStream_t stream) {
  constexpr int NMAX = two_out_nmax(NA + NB);
  const int64_t chunks = args.numel / ((int64_t)BLOCK * VEC);
  TwoOutArgs<NMAX> ta;
  for (int k = 0; k < NMAX; ++k) {
    const bool live = k < NA + NB;
    ta.in[k] = live ? args.in[k] : nullptr; ta.c0[k] = live ? args.c0[k] : 0.f; ta.c1[k] = live ? args.c1[k] : 0.f;
  }
  ta.out0 = args.out0; ta.out1 = args.out1; ta.seeds = args.seeds;
  ta.xmap_lr = xmap_lr_for(chunks); ta.bps_shift = bps_shift;
  ta.stream0 = args.stream0; ta.stream1 = args.stream1;
  ta.chain = args.chain; ta.zeta0 = args.zeta0; ta.zeta1 = args.zeta1;
  ta.tab = RowRef{args.rows, args.index, args.row_offset};
  hipLaunchKernelGGL((step_kernel_k2<TA, NA, NB, NOISE, PACE, TAB, NT>), dim3((unsigned)chunks), dim3(BLOCK), 0, stream, ta);
  return finish_launch();
}

// the operand counts the samplers emit (tools/trace_plans.py): UniPC / SPC of order n give 2n+2 (+2 with a noise
// tensor) 16-bit operands and the previous corrected state in fp32; their first steps have no fp32 operand yet
// store policy of the no-noise two-output launches (see store8): "two_nt" 1 / 0 forces it, -1 = where it measured faster
static bool nt_stores(int operands) { return g_tune.two_nt >= 0 ? g_tune.two_nt != 0 : NT_FROM_OPERANDS > 0 && operands >= NT_FROM_OPERANDS; }

template <typename TA>
int launch_one_trip_two(const StepArgs<float>& args, bool noise, bool group_b_f32, hipStream_t stream, bool& taken) {
  taken = false;
  const int na = args.n_a, nb = args.n_terms - args.n_a;
  if (nb > 1 || (nb == 1 && !group_b_f32) || ((!g_tune.tile || !g_tune.two_out) && args.rows == nullptr)) return SKR_OK;
  // measured (tools/bench_plan.py, 256x16x128x128): with Philox and <= 7 operands the general kernel is 1-2 % faster
  // (193 vs 197 us at 4+1, 239 vs 241 us at 6+1); from 8+1 on, and without noise, this kernel wins (344 vs 365 us at 10+1)
  if (noise && na + nb <= 7 && g_tune.two_out != 2 && args.rows == nullptr) return SKR_OK;
  int bps_shift = 0;
  if (!one_trip_ok(args.numel, args.sample_numel, noise, &bps_shift, args.rows != nullptr)) return SKR_OK;
#define SKR_GO(A, B)                                                                             \
  if (na == A && nb == B) {                                                                      \
    taken = true;                                                                                \
    if (args.rows != nullptr) return noise ? launch_k2<TA, A, B, true, true, true>(args, bps_shift, stream) : launch_k2<TA, A, B, false, false, true>(args, bps_shift, stream); \
    if (!noise) {                                                                                \
      if constexpr (A + B >= 9) { if (nt_stores(A + B)) return launch_k2<TA, A, B, false, false, false, true>(args, bps_shift, stream); } \
      return launch_k2<TA, A, B, false, false, false>(args, bps_shift, stream);                  \
    }                                                                                            \
    return g_tune.pace ? launch_k2<TA, A, B, true, true, false>(args, bps_shift, stream) : launch_k2<TA, A, B, true, false, false>(args, bps_shift, stream); \
  }
  SKR_GO(2, 0) SKR_GO(3, 0) SKR_GO(4, 0) SKR_GO(4, 1) SKR_GO(6, 1) SKR_GO(7, 1) SKR_GO(8, 1) SKR_GO(10, 1)
#undef SKR_GO
  // UniPC / SPC of order 5-9 (2n+2 operands, +2 with noise tensors) and the previous state: unpaced, and the table form
  // while the operands fit a device-resident row (tools/trace_plans.py)
#define SKR_GO(A, B)                                                                             \
  if (na == A && nb == B) {                                                                      \
    if (args.rows != nullptr) {                                                                  \
      if constexpr (A + B <= SKR_ROW_TERMS) { taken = true; return noise ? launch_k2<TA, A, B, true, false, true>(args, bps_shift, stream) : launch_k2<TA, A, B, false, false, true>(args, bps_shift, stream); } \
      else return SKR_OK;                                                                        \
    }                                                                                            \
    taken = true;                                                                                \
    return noise ? launch_k2<TA, A, B, true, false, false>(args, bps_shift, stream) : launch_k2<TA, A, B, false, false, false>(args, bps_shift, stream); \
  }
  SKR_GO(12, 1) SKR_GO(14, 1) SKR_GO(16, 1) SKR_GO(18, 1) SKR_GO(20, 1) SKR_GO(22, 1)
#undef SKR_GO
  return SKR_OK;
}
template int launch_one_trip_two<bf16_t>(const StepArgs<float>&, bool, bool, hipStream_t, bool&);
template int launch_one_trip_two<f16_t>(const StepArgs<float>&, bool, bool, hipStream_t, bool&);

}  // namespace skr
