/*
 * skrample_hip.h -- C ABI of the MI355X (gfx950) sampler-step engine.
 *
 * This is the drop-in boundary for the per-step hot path of Beinsezii/skrample.  The reference is
 * pure Python and has no FFI of its own; each entry point below names the reference interface whose
 * tensor arithmetic it replaces (file:line under the reference tree).  The Python side
 * (skrample_amd/_hip.py) binds these with ctypes; INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every buffer is caller-owned device memory; launches
 * are asynchronous on the caller's hipStream_t (passed as void*); every function returns an int
 * status (0 = SKR_OK) and never throws; no hidden global state except the loaded code object.
 * Thread-safe for concurrent callers on distinct streams.
 */
#ifndef SKRAMPLE_HIP_H
#define SKRAMPLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKR_ABI_VERSION 13
#define SKR_MAX_TERMS 80 /* 2 x 35-stage tableau pairs + base + noise, see skr_step_plan */

/* Devices and streams: every entry point launches on the device that owns its output buffer (queried from the pointer when
 * the process sees more than one GPU; single-GPU processes skip the query) and restores the caller's current device before
 * returning.  `stream` is a hipStream_t of that device (NULL = its default stream).  Calls are asynchronous; distinct
 * streams may be driven from distinct threads. */

enum skr_status {
  SKR_OK = 0,
  SKR_ERR_NULL = 1,        /* required pointer is NULL */
  SKR_ERR_DTYPE = 2,       /* dtype combination has no kernel */
  SKR_ERR_TERMS = 3,       /* n_terms out of range or terms not grouped by dtype */
  SKR_ERR_ALIGN = 4,       /* a buffer is not 16-byte aligned */
  SKR_ERR_SHAPE = 5,       /* numel / sample_numel / shape arguments inconsistent */
  SKR_ERR_LAUNCH = 6,      /* hipLaunchKernel failed (see skr_last_hip_error) */
  SKR_ERR_UNSUPPORTED = 7, /* valid request outside what the kernels cover */
  SKR_ERR_LIBRARY = 8,     /* an external library failed its self-check: a new hipFFT plan transformed unit impulses wrongly (skr_noise_colored_any / skr_colorize) */
  SKR_ERR_CAPTURE = 9,     /* the stream is capturing and this shape's first use must build per-length tables or plans: run it once eagerly first */
};

enum skr_dtype { SKR_BF16 = 0, SKR_F16 = 1, SKR_F32 = 2, SKR_F64 = 3, SKR_NONE = -1 };

/*
 * One fused solver step:
 *
 *     out0[e] = sum_k coef0[k] * in_k[e]  + zeta0 * N(seed[s(e)], stream0, e)
 *     out1[e] = chain * out0[e] + sum_k coef1[k] * in_k[e] + zeta1 * N(seed[s(e)], stream1, e)
 *
 * evaluated in fp32 (fp64 if acc_f64) registers and rounded once per output, in a single pass over
 * HBM.  `chain` uses out0 before its rounding to out0_dtype.  N() is the Philox4x32-10 / Box-Muller
 * normal of oracle/skr_oracle/noise.py::philox_normal, keyed by the per-sample seed (s(e) =
 * e / sample_numel), so results do not depend on how a batch is sharded over GPUs.
 *
 * Replaces, per sampler (coefficients are computed on the host in fp64 by skrample_amd.sampling):
 *   - DiffusionModel.to_x / from_x / ModelConvert.output_to   skrample/sampling/models.py:92-224
 *   - DiffusionModel.forward (sample*G + output*D + noise*z)   skrample/sampling/models.py:53-67
 *   - Euler / DPM / Adams / UniP._sample_packed tensor math    skrample/sampling/structured.py:167-436
 *   - UniPC.sample_packed (corrector out0 + predictor out1)    skrample/sampling/structured.py:469-497
 *   - RKWrapperCore.step_tableau_inside_out (one stage)        skrample/diffusers.py:746-796
 *   - the dtype casts around them                              skrample/diffusers.py:575-599
 *
 * Inputs must be grouped by dtype: terms [0, n_group_a) have dtype_a, the rest dtype_b.
 */
typedef struct skr_step_plan {
  int32_t n_terms;     /* 0..SKR_MAX_TERMS */
  int32_t n_group_a;   /* terms [0,n_group_a) are dtype_a, [n_group_a,n_terms) dtype_b */
  int32_t dtype_a;     /* skr_dtype */
  int32_t dtype_b;     /* skr_dtype; ignored when n_group_a == n_terms */
  int32_t out0_dtype;  /* skr_dtype, SKR_NONE if out0 is not stored */
  int32_t out1_dtype;  /* skr_dtype, SKR_NONE if there is no second output */
  int32_t acc_f64;     /* 1 = accumulate in double (compute_scale=float64), else float */
  int32_t noise_mode;  /* 0 = none, 1 = in-kernel Philox normals */
  double coef0[SKR_MAX_TERMS];
  double coef1[SKR_MAX_TERMS];
  double chain;
  double zeta0, zeta1;
  uint64_t stream0, stream1; /* Philox stream id of the draw feeding out0 / out1 */
  int64_t sample_numel;      /* elements per batch item (prod(shape[1:])) */
  /* Optional rounded pair conversion (Runge-Kutta wrapper, skrample/diffusers.py:819-834): when
   * convert_to or convert_from is non-zero, out0 is NOT the coef0 combination but
   *     out0 = from_x(s, to_x(s, o)),  s = inputs[0], o = inputs[1]  (both of dtype_a)
   * evaluated one rounded op at a time in dtype_a's arithmetic, exactly as the reference's tensor ops:
   *   to_x   1: (s - k0*o)/k1   2: k1*s - k0*o   3: o*k0        (0: o)
   *   from_x 1: (s - k2*x)/k3   2: (k2*s - x)/k3 3: x/k2        (0: x)
   * out1 = chain*out0 + sum_k coef1[k]*in_k (+ noise) as usual.  Requires out1. */
  int32_t convert_to, convert_from;
  double convert_k[4];
} skr_step_plan;

int skr_step_launch(const skr_step_plan* plan, const void* const* inputs, void* out0, void* out1,
                    const uint64_t* seeds_dev /* [batch] device, may be NULL if noise_mode==0 */,
                    int64_t numel, void* stream);

/*
 * Device-resident step scalars -- SURVEY.md 8(f) rank 1; replaces the per-step host work of
 * SkrampleWrapperScheduler.step (skrample/diffusers.py:565-567: timestep -> index lookup with an .item() sync) and of the
 * schedule evaluation inside every sampler step (skrample/scheduling.py:51-62, skrample/sampling/interface.py:34-59).
 *
 * A launch recorded into a HIP graph freezes its kernel arguments, so a graph built on skr_step_launch serves exactly one
 * schedule.  skr_step_launch_indexed takes the step's scalars from device memory instead: the kernel reads
 *     row = rows_dev[(index_dev ? index_dev[0] : 0) + row_offset]
 * when it runs.  `plan` fixes the structure only (operand count and dtypes, outputs, noise_mode, whether a rounded conversion
 * exists); its coef0 / coef1 / chain / zeta / stream / convert_k values are ignored, and convert_to / convert_from must be
 * the row's kinds.  One captured loop therefore serves any schedule of its length (rewrite the rows: new sigmas, shift,
 * begin index, stochasticity) and several resident schedules (move index_dev), with no re-capture and no host sync.
 * Rows with zeta = 0 skip the draw exactly as a launch without noise does.  Covered: launches the one-trip kernels take
 * (whole 2048-element chunks, <= 16 operands -- the size of a row --, fp32 accumulation); anything else returns SKR_ERR_UNSUPPORTED.
 */
#define SKR_ROW_TERMS 16
typedef struct skr_step_row {
  double coef0[SKR_ROW_TERMS];
  double coef1[SKR_ROW_TERMS];
  double chain;
  double zeta0, zeta1;
  uint64_t stream0, stream1;
  double convert_k[4];
} skr_step_row;

int skr_step_launch_indexed(const skr_step_plan* plan, const void* const* inputs, void* out0, void* out1,
                            const uint64_t* seeds_dev, int64_t numel, const skr_step_row* rows_dev,
                            const int32_t* index_dev /* device int32, may be NULL */, int32_t row_offset, void* stream);

/*
 * Step programs -- a plan the library keeps, launched by handle.  Replaces the per-step host work of a REPLAYED step
 * (skrample/diffusers.py:565-599 redoes the whole step algebra every call; skrample_amd lowers each distinct step once,
 * sampling/program.py): the plan -- coefficients, dtypes, conversion kinds, sample size -- is handed over and validated once,
 * a launch passes only what changes from call to call: operand / output / seed pointers and the two Philox stream ids.
 * skr_program_launch(prog, ...) == skr_step_launch(plan with stream0 / stream1 replaced, ..., numel, stream), bit for bit.
 * A program is immutable after creation and may be launched from several threads; destroy it when no launch is in flight
 * on the host side (device work already enqueued is unaffected).
 */
typedef struct skr_program skr_program;
int skr_program_create(const skr_step_plan* plan, int64_t numel, skr_program** out);
int skr_program_launch(const skr_program* prog, const void* const* inputs, void* out0, void* out1,
                       const uint64_t* seeds_dev, uint64_t stream0, uint64_t stream1, void* stream);
void skr_program_destroy(skr_program* prog);

/*
 * A solver step replayed one rounded tensor operation at a time -- the reference's arithmetic when its samplers are called on tensors
 * directly (StructuredSampler.sample, skrample/sampling/structured.py:70-86 with :167-497 and models.py:53-224; no scheduler wrapper,
 * or a wrapper with compute_scale=None): every `*`, `+`, `-`, `/` is a torch op of its own in the TENSOR dtype -- operands widened to the
 * op-math type (fp32 for bf16 / fp16 / fp32 tensors, fp64 for fp64; a Python scalar is converted to it first), one operation, the result
 * rounded to the tensor dtype.  skr_step_launch evaluates the collapsed form in fp32 and rounds once (closer to the exact value, not
 * the reference's bits); skr_tape_launch runs the recorded sequence itself, values on chip (a register file in LDS), one pass over HBM:
 *
 *     LOAD   r[dst] = inputs[a][e]                        STORE  outputs[b][e] = r[a]
 *     MUL_S  r[dst] = rnd(r[a] * k)    DIV_S  rnd(r[a] / k)    ADD_S  rnd(r[a] + k)    RSUB_S  rnd(k - r[a])    RDIV_S  rnd(k / r[a])
 *     ADD    r[dst] = rnd(r[a] + r[b]) SUB    rnd(r[a] - r[b]) MUL    rnd(r[a] * r[b]) DIV     rnd(r[a] / r[b]) NEG     -r[a]
 *     ADD_MS r[dst] = rnd(r[a] + rnd(r[b] * k))   SUB_MS  rnd(r[a] - rnd(r[b] * k))   RSUB_MS  rnd(rnd(r[b] * k) - r[a])   MULZ_S  rnd(rnd(r[a] * k) + 0)
 *       (two ops of the lines above in one visit, for a product nothing else reads: `total + p * q`, `s - sigma * o`, `0 + p * q` -- same roundings)
 *
 * (IEEE operations, no contraction; rnd = round-to-nearest-even to `dtype`.  The scalar k: converted to the op-math type for MUL_S / DIV_S /
 * RDIV_S, but rounded to `dtype` FIRST for ADD_S / RSUB_S -- torch's add / sub / rsub of a Python number to a 16-bit CPU tensor do that, its
 * mul / div do not.)  Every tensor has `dtype` and `numel` elements; register
 * numbers are < SKR_TAPE_REGS (the host allocates them); SKR_ERR_TERMS for a malformed tape.
 */
#define SKR_TAPE_MAX_OPS 96
#define SKR_TAPE_REGS 16
#define SKR_TAPE_MAX_INPUTS 24
#define SKR_TAPE_MAX_OUTPUTS 4
enum skr_tape_code {
  SKR_TAPE_LOAD = 0, SKR_TAPE_STORE = 1, SKR_TAPE_MUL_S = 2, SKR_TAPE_DIV_S = 3, SKR_TAPE_ADD_S = 4, SKR_TAPE_RSUB_S = 5, SKR_TAPE_RDIV_S = 6,
  SKR_TAPE_ADD = 7, SKR_TAPE_SUB = 8, SKR_TAPE_MUL = 9, SKR_TAPE_DIV = 10, SKR_TAPE_NEG = 11,
  SKR_TAPE_ADD_MS = 12, SKR_TAPE_SUB_MS = 13, SKR_TAPE_RSUB_MS = 14, SKR_TAPE_MULZ_S = 15
};
typedef struct skr_tape_op {
  int32_t code; /* skr_tape_code */
  int32_t dst, a, b;
  double k;
} skr_tape_op;
typedef struct skr_tape {
  int32_t n_ops, n_inputs, n_outputs;
  int32_t dtype; /* skr_dtype of every tensor */
  skr_tape_op ops[SKR_TAPE_MAX_OPS];
} skr_tape;
int skr_tape_launch(const skr_tape* tape, const void* const* inputs, void* const* outputs, int64_t numel, void* stream);

/*
 * Noise generators -- replace skrample/pytorch/noise.py behind BatchTensorNoise.generate
 * (noise.py:438-446) / SkrampleWrapperCore.get_step_noise (diffusers.py:312-346).
 * One launch (or a short fixed chain of launches) produces the whole [batch, *unit] tensor; every
 * reduction is per sample.  `seeds_dev` holds one 64-bit seed per batch item; `stream` numbers the
 * draw (the wrapper passes the step index) so repeated calls give fresh, reproducible noise.
 */

/* Random.generate (noise.py:58-74): out = N() */
int skr_noise_random(void* out, int32_t out_dtype, const uint64_t* seeds_dev, uint64_t stream_id,
                     int64_t batch, int64_t sample_numel, void* stream);

/* Brownian.generate (noise.py:210-242; the tree itself is torchsde.BrownianInterval, an un-vendored dependency):
 *   out = scale * (S_to - S_from),   S_x = sum_{k < n_streams} weights_x[k] * N(stream_ids[k])
 * The Brownian path W(t) on [0,1] is a fixed linear function of per-node normals (terminal value + one
 * Brownian-bridge midpoint normal per dyadic interval), so W(t) is a weighted sum of at most depth+1 node normals; the
 * host walks the tree in fp64 and passes the node stream ids (ascending) with each endpoint's weights (host pointers,
 * copied into the launch; a node off one endpoint's path has weight 0 there).  n_streams <= 64.
 * cache_f32 (optional, [batch*sample_numel] fp32, updated in place) receives S_to; with from_cache != 0 it is read as
 * S_from first (weights_from may then be null) -- the sequential-steps case, where each query starts where the last
 * one ended, costs one path instead of two.  Cache hits and misses produce identical bits. */
#define SKR_MAX_WEIGHTED_STREAMS 64
int skr_noise_brownian(void* out, int32_t out_dtype, const uint64_t* seeds_dev, const uint64_t* stream_ids,
                       const double* weights_to, const double* weights_from, int32_t n_streams, double scale,
                       float* cache_f32, int32_t from_cache, int64_t batch, int64_t sample_numel, void* stream);

/* Offset.generate (noise.py:84-113): out = N(stream_base) + strength^2 * N(stream_offset)[broadcast].
 * `unit_shape[ndim]` (ndim <= 4) is the per-sample shape; bit k of keep_mask set = dimension k of the unit
 * shape keeps its size in the offset tensor (reference OffsetProps.dims), otherwise it is broadcast. */
int skr_noise_offset(void* out, int32_t out_dtype, const uint64_t* seeds_dev, uint64_t stream_base,
                     uint64_t stream_offset, int64_t batch, const int64_t* unit_shape, int32_t ndim,
                     uint32_t keep_mask, double strength, void* stream);

/* Pyramid.generate (noise.py:146-207) over the last two dims (resize_h = 1) or the last dim (resize_h = 0,
 * h must be 1) of a [batch][lead][h][w] tensor:
 *   out = (N(base) + sum_{l >= skip} strength^l * upsample_bilinear(N(level l)))  /  per-sample unbiased std
 * The base normal is stream_base+0; the pyramid component uses stream_levels (= stream_base for a fresh pyramid
 * per draw, the first draw's id for PyramidProps.static): geometry uniforms = stream_levels+255, level l normals
 * (stream_levels+1+l, shape [lead][h_l][w_l]) are generated into LDS and sampled there; level 0 is full resolution;
 * skip = max(0, n_levels-1-depth).  Workspaces: scratch_f32 [batch*lead*h*w], partials_f64 [batch*lead*2],
 * level_ws int32 [batch*17] (receives the level table, readable for tests).  with_base = 0 returns the un-normalised
 * pyramid component alone (reference Pyramid.pyramid(), noise.py:146-200).  Limits: w % 4 == 0 and
 * about h*w <= 380*380 (the level stage must fit 152 KiB of LDS); returns SKR_ERR_UNSUPPORTED beyond them --
 * skr_noise_pyramid_any below covers every shape. */
int skr_noise_pyramid(void* out, int32_t out_dtype, float* scratch_f32, double* partials_f64, int32_t* level_ws,
                      const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch,
                      int64_t lead, int64_t h, int64_t w, int32_t resize_h, double strength, int32_t depth,
                      int32_t with_base, void* stream);

/* Same generator for any plane size and width (no `w % 4`, no LDS limit): the level normals are generated into
 * levels_f32 ([batch * lead*h*w] fp32) and sampled from global memory.  partials_f64 = [batch * n_slots * 2],
 * n_slots (1..65535) = workgroups per sample of the main pass.  Same values as skr_noise_pyramid where both apply
 * (up to the summation order of the per-sample statistics). */
int skr_noise_pyramid_any(void* out, int32_t out_dtype, float* scratch_f32, float* levels_f32, double* partials_f64,
                          int32_t n_slots, int32_t* level_ws, const uint64_t* seeds_dev, uint64_t stream_base,
                          uint64_t stream_levels, int64_t batch, int64_t lead, int64_t h, int64_t w, int32_t resize_h,
                          double strength, int32_t depth, int32_t with_base, void* stream);

/* Pyramid over ANY one or two axes of the unit (reference noise.py:146-193: the resized axes are permuted to the end,
 * interpolated slice by slice and permuted back; level l normals are drawn in the unit's own axis order, reduced sizes on the
 * resized axes).  unit_shape[ndim] (ndim <= 4; merge adjacent untouched axes), axis_a < axis_b the resized axes (axis_a = -1:
 * only axis_b).  Same streams, level geometry, workspaces and normalisation as skr_noise_pyramid_any, which is the special
 * case unit_shape = (lead, h, w), axes (1, 2). */
int skr_noise_pyramid_nd(void* out, int32_t out_dtype, float* scratch_f32, float* levels_f32, double* partials_f64, int32_t n_slots,
                         int32_t* level_ws, const uint64_t* seeds_dev, uint64_t stream_base, uint64_t stream_levels, int64_t batch,
                         int32_t ndim, const int64_t* unit_shape, int32_t axis_a, int32_t axis_b, double strength, int32_t depth,
                         int32_t with_base, void* stream);

/* Colored.generate / colorize_noise (noise.py:337-425): white Philox noise shaped in the Fourier domain by
 * clamp(radial_frequency, eps)^(-exponent/2) and rescaled per sample to the white noise's std (or `energy`).
 * The per-sample transform is over (d1, d2, d3) (d1 = 1 for a 2-D unit).  Covered by the hand-written LDS transforms: every axis a
 * power of two <= 4096; or d1 a power of two <= 16 (or 1) over planes whose sides are 2^a * r, r odd <= 63, with d2 even and
 * d3 a multiple of 4 (every such side up to 126 / 252, then 96 * 2^k, 160 * 2^k ...: 90, 96, 104, 112, 144, 152, 160, 168, 192 ...) as long as one plane fits a CU's LDS (up to 192 x 192).  Other shapes:
 * SKR_ERR_UNSUPPORTED -- use skr_noise_colored_any.
 * Workspaces (caller-provided): spec_c64 = batch*d1*d2*(d3/2+1) complex64, scratch_f32 = batch*d1*d2*d3,
 * partials_f64 = 4*batch*partial_slots doubles with partial_slots >= ceil(d1*d2 / max(1, 4096/d3)). */
int skr_noise_colored(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64,
                      int64_t partial_slots, const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch,
                      int32_t d1, int32_t d2, int32_t d3, double exponent, int32_t has_energy, double energy,
                      void* stream);

/* Same generator for per-sample shapes that are not powers of two (rank 1-12 after dropping size-1 dims, any
 * sizes >= 2): identical pipeline; the real N-D transform of the inner (up to three) axes runs on the library's own
 * any-length kernels (skr_fft_own.hip): lengths 2^a r with r a product of at most three factors out of 3, 5, 7, 11, 13 directly
 * (up to 4096), every other length <= 2048 through Bluestein's chirp-z on the same LDS tile transform (its per-length tables are
 * allocated on first use, outside stream capture: SKR_ERR_CAPTURE), and a LAST axis of any even length n = 2 A B with A, B such
 * lengths (8192, 65536, 5000, 10010 ...) as a half-length complex transform in four steps over the same kernels.
 * hipFFT (dlopen'ed on first use) runs only when asked for (skr_set_tuning "hipfft" 1 / SKR_FFT_HIPFFT): a shape outside the above
 * is SKR_ERR_UNSUPPORTED otherwise.  More than three axes (e.g. channels x frames x height x width): every outer axis
 * (<= 128 long, at most nine of them) is a direct DFT kernel, the outermost one fused with the radial weights.
 * Workspaces: spec_c64 = batch*prod(dims[:-1])*(dims[-1]/2+1) complex64, scratch_f32 = batch*prod(dims), partials_f64 = 4*batch*256 doubles. */
int skr_noise_colored_any(void* out, int32_t out_dtype, void* spec_c64, float* scratch_f32, double* partials_f64,
                          const uint64_t* seeds_dev, uint64_t stream_id, int64_t batch, int32_t rank,
                          const int32_t* dims, double exponent, int32_t has_energy, double energy, void* stream);

/* mean(|a - b|^power), power 1 or 2 (FunctionalAdaptive.mae / .mse, skrample/sampling/functional.py:197-214);
 * a may be NULL (= zeros).  Deterministic two-stage reduction in double; the result lands in out_dev[0]
 * (the adaptive sampler reads it back: step-size control is a host decision).  partials_dev: 1024 doubles. */
int skr_error_mean(const void* a_or_null, const void* b, int32_t dtype, int64_t numel, int32_t power,
                   double* out_dev, double* partials_dev, void* stream);

/* raw generator outputs, for parity tests of the RNG itself */
int skr_philox_u32(uint32_t* out /* [n_blocks*4] device */, uint64_t seed, uint64_t stream_id,
                   uint64_t first_block, int64_t n_blocks, void* stream);

/* Colored.colorize_noise (noise.py:337-403) on a caller's white noise: white_f32 ([batch * prod(dims)] fp32, used as
 * the transform workspace and overwritten) is shaped exactly as skr_noise_colored_any shapes its own draws; same
 * workspaces, any dims (hipFFT).  exponent = 0 is the caller's fast path (plain rescale), not handled here. */
int skr_colorize(void* out, int32_t out_dtype, void* spec_c64, float* white_f32, double* partials_f64, int64_t batch,
                 int32_t rank, const int32_t* dims, double exponent, int32_t has_energy, double energy, void* stream);

/* SPC's signed-power blend (structured.py:568-572 with common.py:187-190), the one non-linear tensor op of the
 * samplers: out = spowf(p * spowf(a, P) + c * spowf(b, P), 1/P), spowf(x, f) = |x|^f * sign(x).  out is fp32 or
 * fp64 (the wrappers' compute_scale; fp64 uses double-precision pow); a and b may be any of the four dtypes.  P != 0. */
int skr_power_blend(void* out, int32_t out_dtype, const void* a, int32_t a_dtype, const void* b, int32_t b_dtype, double p,
                    double c, double power, int64_t numel, void* stream);

/* Diagnostics counters of this process (tests assert that a shape did NOT go to the vendor FFT): "hipfft_plans" = hipFFT plan pairs
 * created so far, "hipfft_execs" = forward hipFFT transforms run so far, "own_fft_execs" = forward N-D transforms run by the library's
 * own any-length kernels so far; -1 for an unknown key. */
int64_t skr_stat(const char* key);

int skr_abi_version(void);
const char* skr_strerror(int status);
int skr_last_hip_error(void); /* hipError_t of the most recent failed launch on this thread */
const char* skr_build_info(void);

/* Kernel-selection switches for tests and tuning tools (process-wide, not synchronised with launches in flight on
 * other threads; results are bit-identical under every setting -- only speed changes):
 *   "one_trip" 1|0  one-trip loads-first kernels for launches made of whole 2048-element chunks (default 1)
 *   "xmap"     n    XCD-aware chunk map of the one-trip kernels: every XCD takes runs of 2^n consecutive chunks
 *                   (0 = identity map)
 *   "tile"     1|0  whole-line tile layout when a 32-bit tensor takes part (default 1)
 *   "two_out"  2|1|0  compile-time one-trip kernel for two-output (UniPC / SPC) launches: 1 where it measured faster
 *                   (default), 2 wherever it is instantiated, 0 never
 *   "pace"     1|0  paced load issue in the one-trip kernels (default 1)
 *   "two_nt"   -1|0|1  stores of two-output launches without in-kernel noise and >= 9 operands: non-temporal (1), write-through (0),
 *                   by operand count where it measured faster (-1, the default)
 *   "rk_uv"    0|1|2|4  vectors per lane of the grid-stride Runge-Kutta stage kernel (0 = default)
 *   "rk_blk"   0|128|256  threads per workgroup of the one-trip Runge-Kutta stage kernel (0 = by operand count, the default:
 *                   128 for 4-6 operands, 256 otherwise)
 *   "tape_words" 0|1|2  skr_tape_launch: 16-byte words a lane carries per trip (0 = by tensor size, the default: two from 2 Mi elements up)
 *   "fft_rank" 0|1|2  skr_noise_colored_any / skr_colorize: trailing axes handed to the N-D transform (0 = up to three, the
 *                   default); the other axes run on the direct-DFT kernels (results agree to rounding, not bit for bit)
 *   "hipfft"   -1|0|1  the N-D transform of those axes: 0 the library's own kernels (skr_noise_colored_any's comment lists the lengths;
 *                   anything else is SKR_ERR_UNSUPPORTED), 1 hipFFT (dlopen'ed; any length), -1 = by the environment, the default: own
 *                   kernels unless SKR_FFT_HIPFFT is set
 *   "reset"    (value ignored) back to the defaults
 * Initial values can also be set by the environment: SKR_ONE_TRIP=0, SKR_XMAP=n, SKR_NO_TILE, SKR_NO_TWO_OUT, SKR_RK_UV=n, SKR_RK_BLK=n, SKR_TAPE_WORDS=n. */
int skr_set_tuning(const char* key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* SKRAMPLE_HIP_H */
