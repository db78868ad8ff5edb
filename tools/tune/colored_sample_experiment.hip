// RECORD of a round-3 experiment, not built: the one-launch Colored kernel (`colored_sample`) that lived behind -DSKR_COLORED_SAMPLE
// inside skrample_amd/csrc/skr_colored.hip until round 4.  It measured SLOWER than the shipped three launches (622 vs 339 us at
// 256 x (16, 128, 128): profiles/r03_colored_one_launch_experiment.txt) and was removed from the product source;
// the text below is the three pieces as they stood (kernel, host-side eligibility test, launch), for whoever wants to pick the idea up.
// To revive: paste them back into skr_colored.hip (kernel after colored_outer_axis_regs; the two host pieces inside skr_noise_colored's
// fused branch), give ColoredArgs the `uint32_t* ctl; int32_t* failed;` fields and a `static int32_t* g_sample_failed` again, and build a
// variant library with -DSKR_COLORED_SAMPLE (tools/ab_colored_sample.py loads it through SKR_HIP_LIB).
#if 0
#ifdef SKR_COLORED_SAMPLE
// ---- one launch per draw for 3-D units: the planes of a sample meet twice inside the kernel ------------------------------
// Round 3 EXPERIMENT, not compiled into the library (-DSKR_COLORED_SAMPLE: tools/tune/tune_colored.hip, tools/ab_colored_sample.py).
// Measured SLOWER than the three launches: 622 vs 339 us at 256 x (16, 128, 128), 367 vs 332 us at 1024 x (4, 128, 128)
// (profiles/r03_colored_one_launch_experiment.txt).  A block spends 27 us between its last spectrum store and the end of
// its outer-axis share and 12 us more at the second arrival -- its sample's planes start up to 30 us apart once the first
// dispatch round is over -- and holds its 67 KB of LDS all the while, so a CU has no block in a transform 63-82 % of the
// time; and the exchange has to bypass the XCD's write-back L2 (`sc1` accesses: +3 us per forward plane), because the
// agent-scope fences that would make ordinary accesses visible across XCDs write back and invalidate the whole L2 each time
// (4.3 ms per draw with them).  A task queue that never waits (A / B / C tasks of different samples interleaved by ticket) would
// still pay the `sc1` stores: 19 + 15 + ~5 us per plane against 15.5 + 15.5 + the hidden share of 83 us now -- a few per
// cent at best.  Kept for the record of what the exchange costs on this memory system.
// The three launches above (plane kernel -> outer axis -> plane kernel) move the half spectrum through HBM four
// times, and the outer-axis launch in the middle is nothing but that traffic (83 of 341 us at 256 x (16, 128, 128), at
// 6.5 TB/s).  Here ONE block owns plane i1 of a sample from the draw to the result: it transforms its plane and stores the
// half spectrum, waits for the sample's other planes, runs the outer axis over its 1 / d1 share of the columns (forward,
// weights, Parseval sums, inverse: in registers, as above), waits again and takes its plane back through the inverse
// transforms.  The exchange is 66 KB per block and step, most of it still in L2 / the Infinity Cache, while the CU's other
// block computes: no phase of the draw is bound by HBM any more.
// Waiting for sibling blocks is safe only if they are guaranteed to be running or to start without this block's help.  Blocks
// therefore take their (sample, plane) from a TICKET drawn when they start: tickets below the oldest unfinished one all belong to
// finished blocks, every ticket holder is resident, and a sample's d1 tickets are consecutive -- so whenever the chip has no room
// for a new block, the resident ones (>= d1 of them) include every sibling of the oldest sample, which can finish.  No
// assumption about the order in which the hardware starts blocks.  A wait is bounded all the same (2 s of the real-time
// counter): on expiry the block raises `failed` (host-visible, checked by the next call) and leaves.
// Every global access the planes of a sample exchange goes to the coherence point (gstore / gload<true>), so the arrival needs no
// cache maintenance: each wave waits for its own stores to be acknowledged, the block meets, one lane signs in and polls.
__device__ __forceinline__ bool sample_barrier(uint32_t* counter, uint32_t expected, int32_t* failed) {
  __shared__ int arrived_sh;
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt / lgkmcnt / expcnt 0: this wave's stores of the phase before have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { ok = 0; break; }  // 100 MHz: 2 s
    }
    if (!ok) __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    arrived_sh = ok;
  }
  __syncthreads();
  return arrived_sh != 0;
}

template <typename T, int CH, int CW, int N>
__global__ __launch_bounds__(PLANE_THREADS, 4) void colored_sample(const ColoredArgs a, int logH_rt, int logW_rt) {
  extern __shared__ float2 smem[];
  __shared__ uint32_t ticket_sh;
  if (threadIdx.x == 0) ticket_sh = __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t ticket = ticket_sh;
  const int64_t smp = ticket / (uint32_t)N;
  const int i1 = (int)(ticket % (uint32_t)N);
  if (smp >= a.batch) return;  // (cannot happen: the grid is batch * N blocks)
  plane_body<0, float, CH, CW, true>(a, logH_rt, logW_rt, smp, i1, smem);
  if (!sample_barrier(a.ctl + 1 + 2 * smp, (uint32_t)N, a.failed)) return;
  {
    // this block's share of the sample's columns: [c0, c1), two columns per thread in flight
    const int64_t cols = (int64_t)a.d2 * a.d3h;
    const int64_t per = (cols + N - 1) / N, c0 = per * i1, c1 = c0 + per < cols ? c0 + per : cols;
    float2* base = a.spec + smp * (int64_t)N * cols;
    double p1 = 0.0, p2 = 0.0;
    // (32-bit column indices on uniform per-plane pointers: one offset register serves the N loads and stores of a column)
    const uint32_t cols32 = (uint32_t)cols, e1 = (uint32_t)c1;
    if constexpr (N >= 16) {  // one column per thread and trip (two would not fit the 128 registers of four waves per SIMD)
      for (uint32_t q = (uint32_t)c0 + threadIdx.x; q < e1; q += PLANE_THREADS) {
        float2 v[N];
#pragma unroll
        for (int n = 0; n < N; ++n) v[n] = gload<true>(base + (size_t)n * cols32 + q);
        outer_column<N, true>(a, v, q, p1, p2);
#pragma unroll
        for (int n = 0; n < N; ++n) gstore<true>(base + (size_t)n * cols32 + q, v[n]);
      }
    } else {
      for (uint32_t q = (uint32_t)c0 + threadIdx.x; q < e1; q += 2 * PLANE_THREADS) {
        const uint32_t q2 = q + PLANE_THREADS;
        const bool two = q2 < e1;
        float2 v[N], w[N];
#pragma unroll
        for (int n = 0; n < N; ++n) v[n] = gload<true>(base + (size_t)n * cols32 + q);
        if (two) {
#pragma unroll
          for (int n = 0; n < N; ++n) w[n] = gload<true>(base + (size_t)n * cols32 + q2);
        }
        outer_column<N>(a, v, q, p1, p2);
#pragma unroll
        for (int n = 0; n < N; ++n) gstore<true>(base + (size_t)n * cols32 + q, v[n]);
        if (two) {
          outer_column<N>(a, w, q2, p1, p2);
#pragma unroll
          for (int n = 0; n < N; ++n) gstore<true>(base + (size_t)n * cols32 + q2, w[n]);
        }
      }
    }
    block_sums<true>(p1, p2, a.partials + (int64_t)a.batch * a.n_slots * 2 + (smp * a.n_slots_c + i1) * 2);
  }
#ifdef SKR_COLORED_TRACE
  if (threadIdx.x == 0 && a.trace) a.trace[(int64_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
  if (!sample_barrier(a.ctl + 2 + 2 * smp, (uint32_t)N, a.failed)) return;
  plane_body<1, T, CH, CW, true>(a, logH_rt, logW_rt, smp, i1, smem);
}

#endif  // SKR_COLORED_SAMPLE
// ---- host side, eligibility ----
#ifdef SKR_COLORED_SAMPLE
    // 3-D units of 128 x 128 / 64 x 64 planes (16-bit and fp32 results): the whole draw in one launch (colored_sample)
    bool sample_fused = nd == 3 && l2 == l3 && (l2 == 7 || l2 == 6) && (out_dtype == SKR_BF16 || out_dtype == SKR_F16 || out_dtype == SKR_F32) &&
                        4 * batch * (int64_t)d1 + batch + 1 <= 4 * batch * partial_slots && getenv("SKR_FFT_NO_SAMPLE") == nullptr;
    if (sample_fused) {
      if (!g_sample_failed && hipHostMalloc(reinterpret_cast<void**>(&g_sample_failed), sizeof(int32_t), hipHostMallocMapped) == hipSuccess) *g_sample_failed = 0;
      if (!g_sample_failed) sample_fused = false;
      else if (*g_sample_failed) return SKR_ERR_LAUNCH;  // an earlier colored_sample launch gave up waiting: its result was not valid
    }
#endif
// ---- host side, launch ----
#ifdef SKR_COLORED_SAMPLE
    } else if (sample_fused) {
      // one launch: tickets + two arrival counters per sample at the tail of the partials buffer, zeroed on the stream
      a.n_slots_c = d1;
      a.ctl = reinterpret_cast<uint32_t*>(partials_f64 + 4 * batch * (int64_t)d1);
      a.failed = g_sample_failed;
      if (hipMemsetAsync(a.ctl, 0, sizeof(uint32_t) * (size_t)(2 * batch + 1), s) != hipSuccess) return SKR_ERR_LAUNCH;
      dim3 grid1((unsigned)(batch * d1));
#define SKR_SAMPLE_N(T, CH, CW, N) do { SKR_ALLOW_LDS((colored_sample<T, CH, CW, N>), lds_plane); hipLaunchKernelGGL((colored_sample<T, CH, CW, N>), grid1, dim3(PLANE_THREADS), lds_plane, s, a, l2, l3); } while (0)
#define SKR_SAMPLE_SZ(T, CH, CW)                                     \
      switch (d1) {                                                  \
        case 2: SKR_SAMPLE_N(T, CH, CW, 2); break;                   \
        case 4: SKR_SAMPLE_N(T, CH, CW, 4); break;                   \
        case 8: SKR_SAMPLE_N(T, CH, CW, 8); break;                   \
        default: SKR_SAMPLE_N(T, CH, CW, 16); break;                 \
      }
#define SKR_SAMPLE(T) do { if (l2 == 7) { SKR_SAMPLE_SZ(T, 7, 7) } else { SKR_SAMPLE_SZ(T, 6, 6) } } while (0)
      switch (out_dtype) {
        case SKR_BF16: SKR_SAMPLE(__bf16); break;
        case SKR_F16: SKR_SAMPLE(_Float16); break;
        default: SKR_SAMPLE(float); break;
      }
      SKR_CHECK_LAUNCH();
#undef SKR_SAMPLE
#undef SKR_SAMPLE_SZ
#undef SKR_SAMPLE_N
#endif
#endif
