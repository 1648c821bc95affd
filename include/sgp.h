/*
 * sgp.h -- C ABI of the MI355X-native sparse-GP inference core (libsgp_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of vr308/Generalised-Gaussian-Processes: the
 * collapsed (Titsias / VFE) sparse-GP bound that the reference re-evaluates on every SGPR Adam
 * step and every HMC/NUTS leapfrog.  The reference has no FFI of its own for this path; the seam
 * it replaces is the set of library calls below (paths relative to the reference repo root):
 *
 *   ScaleKernel(RBFKernel(ard_num_dims=d))(X, Z)          models/sgpr.py:36   models/bayesian_sgpr_hmc.py:41
 *   sig_f**2 * pm.gp.cov.ExpQuad(input_dim, ls=ls)         models/bayesian_sgpr_hmc.py:65
 *   InducingPointKernel(base, inducing_points=Z, lik)     models/sgpr.py:37   models/bayesian_sgpr_hmc.py:42
 *   ExactMarginalLogLikelihood(lik, model)(output, y)     models/sgpr.py:114,125  models/bayesian_sgpr_hmc.py:92,111,130
 *   pm.gp.MarginalSparse(approx="VFE").marginal_likelihood models/bayesian_sgpr_hmc.py:66,71
 *   likelihood(model(test_x))  (posterior predictive)     models/sgpr.py:150-160  models/bayesian_sgpr_hmc.py:186-231
 *
 * Conventions
 *   - every pointer named X, y, Z, Phi, ... is a DEVICE pointer to fp64 data (row-major, leading
 *     dimension given in elements); `inv_ls` is a HOST pointer to d doubles (1/lengthscale_j);
 *     scalars sf2 (= outputscale = sig_f^2), s2 (= noise = sig_n^2), jitter are passed by value.
 *   - the caller owns every buffer.  The library never allocates or frees user-visible memory;
 *     scratch comes from `ws` whose size the matching *_workspace_bytes() query returns.
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = default stream).  Nothing is
 *     synchronised, no host copies are made: results (including `info`) stay on the device.
 *   - return value: 0 = launched; <0 = rejected before any launch (SGP_ERR_*).  Numerical failure
 *     (non-positive Cholesky pivot) is reported LAPACK-style through the device int `*info`:
 *     0 = ok, k>0 = leading minor of order k is not positive definite (1..M refer to Kuu,
 *     M+1..2M to B = I + L^-1 Phi L^-T / s2).  No exceptions cross the boundary.  SGP_INFO_TIMEOUT (< 0)
 *     means the single-launch Cholesky gave up waiting for a tile (seconds of spinning: never expected;
 *     reported instead of hanging the device) -- treat the outputs as invalid.  The single-launch kernels (the dataflow
 *     Cholesky, sgp_small_eval, sgp_small_nuts) synchronise workgroups through device memory: they launch at most one
 *     workgroup per CU of the device and assume that all of them become resident (they do once kernels that hold CUs
 *     retire; forward progress never depends on a workgroup that is not yet resident being scheduled FIRST).
 *   - kernel_id selects k(x,z): SGP_KERNEL_RBF     sf2 * exp(-r2/2)
 *                               SGP_KERNEL_MATERN32 sf2 * (1+sqrt3 r) exp(-sqrt3 r)
 *                               SGP_KERNEL_MATERN52 sf2 * (1+sqrt5 r+5r2/3) exp(-sqrt5 r)
 *     with r2 = sum_j ((x_j - z_j) * inv_ls_j)^2.
 */
#ifndef SGP_H
#define SGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is the whole exported surface. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef void* sgp_stream_t; /* hipStream_t */

#define SGP_ABI_VERSION 3       /* 2: contexts (sgp_ctx_*); the process-wide setters are shims over the default context
                                 * 3: every entry point that reads an option has a sgp_ctx_* twin (the three evaluation orders, the
                                 *    factored pass 2, the whitened bound); sgp_streaming_error_report; phi_diag of the extended order */
#define SGP_MAX_DIM 32          /* largest input dimension d the streaming kernels accept */
#define SGP_MAX_INDUCING 4096   /* largest M */

#define SGP_INFO_TIMEOUT (-7777) /* value of *info, not a return code */

#define SGP_KERNEL_RBF 0
#define SGP_KERNEL_MATERN32 1
#define SGP_KERNEL_MATERN52 2
/* Sum of products of isotropic factors (the reference's CO2 covariance,
 * experiments/co2_bayesian_sgpr_hmc.py:74-83,107-149: n_per^2 Periodic*ExpQuad + n_med^2 RatQuad + n_trend^2 ExpQuad
 * + n_noise^2 Matern32, PyMC3 pm.gp.cov definitions).  With this kernel_id every `inv_ls` argument points to a HOST
 * parameter block of SGP_COMP_LEN doubles instead of d reciprocal lengthscales, sf2 is ignored, d <= 8, and
 * the gradient entry points write dF/d(block) into g_ls (SGP_COMP_LEN doubles, zero in the slots that carry no
 * parameter) and 0 into g_sf2.  Block layout:
 *   [0] nterms (1..SGP_COMP_MAX_TERMS) ; term t at base = 1 + 8 t:  [base] amp2 (> 0)  [base+1] nfac (1..2)
 *   factor f at fb = base + 2 + 3 f:  [fb] SGP_FAC_*  [fb+1] lengthscale (> 0)  [fb+2] aux (RATQUAD: alpha, PERIODIC: period)
 * The composite path materialises K_fu in row chunks (it is meant for workloads of the CO2 size), shares the
 * O(M^3) tail with the other kernels and is not available to the SVGP entry points.                          */
#define SGP_KERNEL_COMPOSITE 3
#define SGP_COMP_MAX_TERMS 4
#define SGP_COMP_MAX_FACTORS 2
#define SGP_COMP_LEN 33
#define SGP_FAC_EXPQUAD 0   /* exp(-r2 / (2 l^2)) */
#define SGP_FAC_MATERN32 1  /* (1 + sqrt3 r / l) exp(-sqrt3 r / l) */
#define SGP_FAC_MATERN52 2  /* (1 + sqrt5 r / l + 5 r2 / (3 l^2)) exp(-sqrt5 r / l) */
#define SGP_FAC_RATQUAD 3   /* (1 + r2 / (2 alpha l^2))^-alpha */
#define SGP_FAC_PERIODIC 4  /* exp(-sum_j sin^2(pi (x_j - z_j) / T) / (2 l^2)) */

#define SGP_OK 0
#define SGP_ERR_ARG (-1)        /* null pointer / non-positive size / bad kernel_id */
#define SGP_ERR_DIM (-2)        /* d > SGP_MAX_DIM or M > SGP_MAX_INDUCING */
#define SGP_ERR_WORKSPACE (-3)  /* ws == NULL or ws_bytes too small */
#define SGP_ERR_LAUNCH (-4)     /* hipGetLastError() != hipSuccess after a launch */

/* slots of the `out` vector written by sgp_bound_from_stats */
#define SGP_OUT_F 0             /* F = logmarg - trace_term  (the collapsed bound, NOT divided by N) */
#define SGP_OUT_LOGMARG 1       /* log N(y | 0, Qff + s2 I) */
#define SGP_OUT_TRACE 2         /* tr(Kff - Qff) / (2 s2) */
#define SGP_OUT_LOGDETB 3       /* log det B */
#define SGP_OUT_QUAD 4          /* y^T (Qff + s2 I)^-1 y */
#define SGP_OUT_TRW 5           /* tr(Kuu^-1 Phi) */
#define SGP_OUT_S2BAR 6         /* dF/d s2      (with_adjoints only) */
#define SGP_OUT_KAPPABAR 7      /* dF/d kappa   (with_adjoints only) */
#define SGP_OUT_LEN 8

int sgp_abi_version(void);

/* ---- contexts (ABI version 2) -------------------------------------------------------------------------------------------------
 * A context carries every switch and every piece of library-side state a call consults: which matrix cores contract pass 1, the
 * conditioning limit, the K'_fu budget, the CU budget, the optional timing events, the one-shot pass-1 gate (kernel attributes are
 * kept per device, not per context).  Environment knobs (SGP_CONTRACTION, SGP_ASM_OVERLAP, SGP_SYRK_*, SGP_I8_PRIO) are read ONCE, when a context is
 * created.  Two contexts in one process are independent (two bounds with different contraction modes on two streams, two devices);
 * one context must not be used from two host threads at the same time.  The caller still owns every buffer and passes the stream:
 * a context owns no user-visible memory.  ctx == NULL everywhere means the DEFAULT context, which is also what every entry point
 * without a context argument runs in and what the deprecated sgp_set_* setters below modify.
 * The reference has no counterpart (GPyTorch / PyMC3 keep such switches in Python-side settings objects, e.g.
 * gpytorch.settings.cholesky_jitter used at models/sgpr.py:14).                                                                  */
typedef struct sgp_ctx sgp_ctx;
#define SGP_OPT_CONTRACTION 0      /* 0 fp64 matrix cores, 1 (default) integer cores where they win, 2 integer cores always */
#define SGP_OPT_ASM_OVERLAP 1      /* 0 (default) one block, 1 head + tail overlapped, 2 head + tail serial (A/B knob) */
#define SGP_OPT_KFU_BUDGET_BYTES 2 /* bytes of K'_fu the library materialises at a time (0 restores the 16 GiB default) */
#define SGP_OPT_COND_LIMIT 3       /* conditioning gate of the explicit-inverse path (default 1e13; 0 disables; < 0 restores) */
#define SGP_OPT_CU_BUDGET 4        /* CUs the context's launches may occupy (CU-masked streams); 0 = the whole device */
#define SGP_OPT_TIMING 5           /* != 0: record HIP events around the dominant kernels (sgp_ctx_timing_last_ms) */
#define SGP_OPT_SHARED_DEVICE 6    /* != 0: the GPU is shared with other processes / ranks (joblib workers as in the reference's
                                      experiments/regression.py:219-231, several ranks on one device): the single-launch Cholesky hands its work
                                      items out by TICKET to workgroups that are running instead of dealing them statically, so that its
                                      progress no longer needs every workgroup of the launch resident at once -- two such launches of two
                                      processes can otherwise starve each other into SGP_INFO_TIMEOUT.  Same bits; 5-10 % slower factorizations
                                      (profiles/r06_potrf_ticket_ab.txt).  Default 0, or the environment's SGP_SHARED_DEVICE when the context is
                                      created.  The Python layer switches it on by itself after a first time-out (CollapsedBound) */
sgp_ctx* sgp_ctx_create(int device); /* device = the HIP device index the context will be used on (the caller selects it); the
                                        sgp_ctx_* compute entry points return SGP_ERR_ARG when another device is current */
void sgp_ctx_destroy(sgp_ctx* ctx);
int sgp_ctx_device(const sgp_ctx* ctx);
int sgp_ctx_set_option(sgp_ctx* ctx, int option, double value);   /* SGP_OK or SGP_ERR_ARG */
double sgp_ctx_get_option(const sgp_ctx* ctx, int option);        /* -1 for an unknown option */
/* Binds `ctx` to the CALLING HOST THREAD (NULL unbinds): from now on the entry points that take no context argument -- sgp_chol_lower,
 * sgp_trsm_lower, sgp_predict, sgp_mixture_predict, the sgp_svgp_* and sgp_small_* families -- read their options (CU budget,
 * conditioning limit, shared-device mode, timing) from it instead of the default context when this thread calls them.  A sgp_ctx_*
 * entry point still runs in the context it is handed.  (Round 6, VERDICT r5 next-7: no deprecated per-thread setter is needed to steer
 * sgp_chol_lower from an engine with a context of its own.)                                                                       */
void sgp_ctx_bind_thread(sgp_ctx* ctx);
void sgp_ctx_set_pass1_gate(sgp_ctx* ctx, void* hip_event);       /* see sgp_set_pass1_gate */
int sgp_ctx_contraction_last(const sgp_ctx* ctx);                 /* what the context's last pass 1 ran: 0 fp64, 1 integer cores */
/* the rule sgp_suffstats_fwd will apply to an N-row shard with M inducing inputs in this context (1 = integer cores) */
int sgp_ctx_contraction_would_use_i8(const sgp_ctx* ctx, int64_t N, int M);
int sgp_ctx_timing_last_ms(sgp_ctx* ctx, int slot, float* ms);
int64_t sgp_ctx_timing_last_rows(const sgp_ctx* ctx, int slot);
/* The entry points whose behaviour depends on an option, with the context as first argument (same arguments, same return values
 * as their namesakes below).  Since ABI version 3 the list is complete: the SVGP and single-launch entry points and the M x M helpers
 * read no option except the CU budget / conditioning limit of the DEFAULT context (sgp_chol_lower, sgp_svgp_*: use the deprecated
 * setters, or the context twins of the bound, to change those). */
size_t sgp_ctx_suffstats_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_kfu);
int sgp_ctx_suffstats_fwd(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                          const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id, double* Phi, double* b,
                          double* yy, double* kappa, double* Kfu_out, void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_ctx_suffstats_bwd(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                          const double* inv_ls, double sf2, const double* Phibar, const double* bbar, double kappabar,
                          const double* Kfu_in, int64_t N, int M, int d, int kernel_id, double* g_ls, double* g_sf2, double* g_Z,
                          void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_ctx_kuu_factor(sgp_ctx* ctx, const double* Kuu, int M, double* Linv_out, int* info, void* ws, size_t ws_bytes,
                       sgp_stream_t stream);
int sgp_ctx_kuu_factor_ex(sgp_ctx* ctx, const double* Kuu, int M, double* Linv_out, int* info, double* trace_out, void* ws,
                          size_t ws_bytes, sgp_stream_t stream);
int sgp_ctx_bound_from_stats(sgp_ctx* ctx, const double* Kuu, const double* Phi, const double* b, const double* yy,
                             const double* kappa, double s2, int64_t N, int M, int with_adjoints, double* out, double* Phibar,
                             double* bbar, double* Kuubar, double* factors, const double* kuu_linv, int* info, void* ws,
                             size_t ws_bytes, sgp_stream_t stream);
/* ABI version 3: the evaluation orders the streaming-order guard falls back to, the factored pass 2 and the whitened bound (they read
 * the context's K'_fu budget, timing slots, CU budget and conditioning limit).  sgp_ctx_suffstats_fwd_extended has one output more
 * than its namesake: phi_diag (DEVICE, M doubles, or NULL) = diag(K_uf K_fu) of this shard, for sgp_streaming_error_report.         */
size_t sgp_ctx_suffstats_whitened_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d);
int sgp_ctx_suffstats_fwd_whitened(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                   const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id, const double* kuu_linv,
                                   double* W, double* u, double* yy, double* kappa, void* ws, size_t ws_bytes, sgp_stream_t stream);
size_t sgp_ctx_suffstats_whitened_rows_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_t);
int sgp_ctx_suffstats_fwd_whitened_rows(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                        const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                        const double* kuu_linv, double* W, double* u, double* yy, double* kappa, double* T_out,
                                        void* ws, size_t ws_bytes, sgp_stream_t stream);
size_t sgp_ctx_suffstats_extended_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d);
int sgp_ctx_suffstats_fwd_extended(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                   const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id, const double* kuu_linv,
                                   int level, double* W, double* u, double* yy, double* kappa, double* Kfu_out, double* phi_diag,
                                   void* ws, size_t ws_bytes, sgp_stream_t stream);
size_t sgp_ctx_suffstats_bwd_factored_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_t);
int sgp_ctx_suffstats_bwd_factored(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                   const double* inv_ls, double sf2, const double* kuu_linv, const double* Cw, double s2,
                                   const double* bbar, double kappabar, int64_t N, int M, int d, int kernel_id, const double* T_in,
                                   double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_ctx_bound_from_whitened_stats(sgp_ctx* ctx, const double* W, const double* u, const double* yy, const double* kappa, double s2,
                                      int64_t N, int M, int with_adjoints, double* out, double* Phibar, double* bbar, double* Kuubar,
                                      double* factors, const double* kuu_linv, int* info, double* Cw, void* ws, size_t ws_bytes,
                                      sgp_stream_t stream);
int sgp_ctx_mixture_predict(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, int64_t N, const double* Xs, int64_t ldxs,
                            int64_t T, const double* Z, int64_t ldz, int S, const double* inv_ls, const double* sf2, const double* s2,
                            double jitter, int M, int d, int kernel_id, int pred_noise, double gate_jitter, double* mean, double* var,
                            double* cov, int* info, int* gate_info, void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- deprecated process-wide switches (ABI version 1): each one sets the matching option of the DEFAULT context ---- */
/* The single-launch dataflow factorization keeps every workgroup resident (one per CU) and sizes its grid from the device's CU
 * count.  A caller that enqueues on a CU-masked stream (hipExtStreamCreateWithCUMask) must say how many CUs the mask leaves:
 * n CUs for the calling HOST THREAD's following launches, 0 = the whole device (default).  Deprecated: SGP_OPT_CU_BUDGET of a
 * context (this per-thread value, while non-zero, overrides the context's).                                                  */
void sgp_set_cu_budget(int n);
const char* sgp_status_string(int status);

/* Optional measurement aid: with timing enabled the library records HIP events on the launch stream
 * around its dominant kernels (slot 0 = kernel assembly, 1 = SYRK contraction of pass 1, 2 = Kbar
 * contraction of pass 2; with several super-chunks the last one).  sgp_timing_last_ms synchronises on
 * the closing event of the slot and returns the elapsed device time of the most recent launch.        */
void sgp_timing_enable(int on);
int sgp_timing_last_ms(int slot, float* ms);
/* data rows of the contraction launch the slot-1 events bracketed (the tail block when pass 1 runs head + tail, below); -1 for
 * the other slots */
int64_t sgp_timing_last_rows(int slot);
/* A/B knob (measured a loss, off by default; csrc/sgp_suffstats_fwd.hip): pass 1 as head block (one round of resident
 * workgroups) + tail block with the tail's kernel assembly on a library-owned side stream beside the head's contraction.
 * mode 0 (default): one block; 1: head + tail, overlapped; 2: head + tail enqueued serially on the caller's stream (the same
 * numbers as mode 1 bit for bit; another split geometry than mode 0, so those differ at rounding level); -1: back to the
 * environment variable SGP_ASM_OVERLAP / the default.                                                                     */
void sgp_set_asm_overlap(int mode);

/* ---- streaming pass 1: sufficient statistics over the local row shard -------------------------
 * Phi = Kuf Kuf^T (M x M, ld M, full symmetric), b = Kuf y (M), yy = y^T y, kappa = sum_n k(x_n,x_n).
 * Replaces the N x M kernel matrix + N M^2 contraction inside InducingPointKernel /
 * ExactMarginalLogLikelihood (models/sgpr.py:37,125) and MarginalSparse (models/bayesian_sgpr_hmc.py:71).
 * Two kernels: "kernel assembly" (K'_fu written once, HBM-write bound) and "contraction" (SYRK on the
 * fp64 matrix cores reading it back).  On several GPUs every rank calls this on its own rows and the
 * caller all-reduces [Phi | b | yy | kappa].  N == 0 is allowed (all outputs zero).             */
size_t sgp_suffstats_workspace_bytes(int64_t N, int M, int d);
/* the same query for a caller that passes its own Kfu_out (caller_owns_kfu != 0): without the library's K'_fu super-chunk
 * (up to 16 GiB) the workspace is a few MB + the split slabs */
size_t sgp_suffstats_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_kfu);
/* doubles in the materialised K'_fu block of an N-row shard: roundup(N,256) x roundup(M,128), row-major.
 * K'_fu[n][m] = k(x_n, z_m) / sf2 (zero in the padding) -- the reference's K_fu (N x M) without the
 * output scale.  Optional: a caller that passes such a buffer as Kfu_out keeps the assembled block and
 * can hand it to sgp_suffstats_bwd (same X, Z, inv_ls, kernel_id) so pass 2 does not re-assemble it.
 * With Kfu_out == NULL the library assembles into `ws`, at most 16 GiB of rows at a time.            */
size_t sgp_kfu_len(int64_t N, int M);
/* Which matrix cores run the pass-1 contraction Phi = K_uf K_fu of sgp_suffstats_fwd (csrc/sgp_suffstats_i8.hip):
 *   0  fp64 (v_mfma_f64_16x16x4_f64) always;
 *   1  (default, SGP_CONTRACTION) the integer cores where they win: rows x padded-M^2 >= 2^32 (65536 rows at M <= 256, 4096 at
 *      M = 1024).  K'_fu in [0, 1] is split into seven balanced 8-bit digit planes (|K' - q 2^-54| <= 2^-55), the 28 digit-pair
 *      products p + r >= 6 are exact int32 sums, folded to fp64 once per 16384 rows: the result is as accurate as the fp64
 *      contraction (2.4-2.8e-16 of max |Phi| against long-double arithmetic).  With Kfu_out != NULL (value + gradient: pass 2
 *      reads the fp64 block) kernel assembly writes the fp64 block AND the digit planes (the planes a super-chunk at a time in
 *      the workspace -- query the workspace size in the mode the call will run in);
 *   2  the integer cores for every call (tests).
 * Returns the previous mode; an out-of-range mode restores the default (1).  sgp_contraction_last(): what the last
 * sgp_suffstats_fwd call of the default context ran (0 fp64, 1 int8).
 * Value-only and value + gradient evaluations agree to rounding, not bit for bit, across contraction modes AND inside mode 1: a
 * value-only call and a value + gradient call at the same theta both contract on the integer cores, but their split geometry
 * differs (the planes of a kept K'_fu live in their own super-chunk), so F differs at the 1e-14 level between `value()` and
 * `value_and_grad()`.  NUTS takes its energy from the value + gradient call alone; a caller comparing the two must allow 1e-13.
 * Replaces nothing in the reference -- torch.matmul in gpytorch's InducingPointKernel (models/sgpr.py:37) is the fp64 GEMM. */
int sgp_set_contraction(int mode);
int sgp_contraction_last(void);
/* One-shot gate for the NEXT sgp_suffstats_fwd call: if that call contracts on the integer cores, the contraction launch waits for
 * `hip_event` (a hipEvent_t the caller has ALREADY recorded on another stream) between kernel assembly and the contraction.  Why: that
 * kernel takes 129 KB of LDS and two 188-register waves per SIMD, so next to nothing co-schedules with it -- a side-stream chain that must be done by the end of
 * pass 1 (the factorization of K_uu, core.py) finishes beside the assembly instead of being starved.  NULL clears; a call that takes the
 * fp64 contraction ignores and clears it; so does any call that returns early (bad argument, composite kernel): the gate belongs to
 * exactly one call and the event may be destroyed as soon as that call has returned. */
void sgp_set_pass1_gate(void* hip_event);

/* bytes of K'_fu the library materialises at a time when it owns the buffer (default 16 GiB; 0 restores
 * the default).  Process-wide; changes what the *_workspace_bytes queries return.                     */
void sgp_set_kfu_budget_bytes(size_t bytes);
int sgp_suffstats_fwd(const double* X, int64_t ldx, const double* y,
                      const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                      int64_t N, int M, int d, int kernel_id,
                      double* Phi, double* b, double* yy, double* kappa, double* Kfu_out,
                      void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- exchange format for several GPUs ("optionally pack only the lower triangle") ----
 * stats = [Phi (M*M, ld M) | b (M) | yy | kappa] as sgp_suffstats_fwd writes them into one contiguous buffer;
 * tri   = [Phi[i][j] for j <= i, row by row (M (M+1)/2) | b | yy | kappa]  (sgp_stats_packed_len(M) doubles).
 * The caller packs, all-reduces `tri` (4.2 MB instead of 8.4 MB at M = 1024) and unpacks (which also mirrors).  */
size_t sgp_stats_packed_len(int M);
int sgp_stats_pack_lower(const double* stats, int M, double* tri, sgp_stream_t stream);
int sgp_stats_unpack_lower(const double* tri, int M, double* stats, sgp_stream_t stream);

/* ---- inducing block: Kuu = k(Z,Z) + jitter I  (M x M, ld M) -----------------------------------
 * ScaleKernel(RBFKernel)(Z,Z) (models/sgpr.py:36-37); jitter = 1e-6 reproduces PyMC3's stabilize(). */
int sgp_kuu(const double* Z, int64_t ldz, const double* inv_ls, double sf2, double jitter,
            int M, int d, int kernel_id, double* Kuu, sgp_stream_t stream);

/* ---- M x M back end (usable on their own; all in place, lower triangle, row-major) ------------ */
size_t sgp_chol_workspace_bytes(int M);
int sgp_chol_lower(double* A, int64_t lda, int M, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* B <- L^-1 B (trans=0) or L^-T B (trans=1); B is M x k, ld ldb */
size_t sgp_trsm_workspace_bytes(int M, int k);
int sgp_trsm_lower(const double* L, int64_t ldl, double* B, int64_t ldb, int trans, int M, int k,
                   void* ws, size_t ws_bytes, sgp_stream_t stream);
/* out[0] = sum_i log L[i][i] */
int sgp_logdiag_sum(const double* L, int64_t ldl, int M, double* out, sgp_stream_t stream);

/* ---- the whole O(M^3) tail in one call ---------------------------------------------------------
 * L = chol(Kuu); W = L^-1 Phi L^-T; B = I + W/s2; LB = chol(B); q = LB^-1 L^-1 b
 * F = -[ N/2 log 2pi + N/2 log s2 + sum log diag LB + (yy/s2 - q.q/s2^2)/2 + (kappa - tr W)/(2 s2) ]
 * with_adjoints != 0 additionally writes Phibar, bbar, Kuubar (M x M / M / M x M, ld M) and the
 * S2BAR / KAPPABAR slots of out.  `factors` (optional, may be NULL) receives what sgp_predict needs:
 * [ L^-1 (M*M) | LB^-1 (M*M) | q (M) ] (opaque; the predictive applies the two inverses one after the other -- their
 * product has entries of size sqrt(cond K_uu) that cancel against k_u*).  yy, kappa are device scalars.               */
size_t sgp_bound_workspace_bytes(int M, int with_adjoints);
size_t sgp_bound_factors_len(int M); /* number of doubles in `factors` */
/* Optional split of the tail: chol(Kuu) and its inverse depend on (Z, theta) only, not on the streamed
 * statistics.  sgp_kuu_factor writes the padded L^-1 (sgp_kuu_factor_len(M) doubles, opaque) and its own `info`
 * (1..M); issued on a second stream it runs underneath pass 1.  Passing the result as `kuu_linv` makes
 * sgp_bound_from_stats skip that part (Kuu may then be NULL).  In that case `info` is NOT cleared on entry:
 * pass the word sgp_kuu_factor wrote (the first failure stays; chol(B) reports M+1..2M only into a word
 * that is still 0), so one status word -- one host read -- covers the whole evaluation.                 */
/* Conditioning gate: the explicit-inverse products downstream of sgp_kuu_factor turn to noise once cond(K_uu + J I) passes
 * ~1e13 (where LAPACK's substitution still evaluates the bound).  sgp_kuu_factor therefore reports a matrix whose estimate
 * max_j ||L e_j||^2 / min_i L_ii^2 (lambda_max >= every column norm of L, lambda_min <= every pivot: the estimate can only UNDERSHOOT
 * cond, so a well-posed problem is never refused; round 3's trace(K_uu) overshot by up to M) exceeds `limit` as numerically not
 * positive definite at its smallest pivot (info = argmin + 1).  Default limit 1e13; 0 disables the gate; negative restores the
 * default.  Deprecated shim over SGP_OPT_COND_LIMIT of the default context.                                                       */
void sgp_set_cond_limit(double limit);
/* Guard of the streaming evaluation order (sgp_suffstats_fwd + sgp_bound_from_stats).  Phi = K_uf K_fu is stored in fp64, i.e. with
 * an error of ~2^-53 max_i Phi_ii per entry however it was summed, and W = L^-1 Phi L^-T amplifies it by 1 / lambda_k(K_uu): for long
 * lengthscales x small noise the bound comes out wrong by far more than the 1e-8 per datum the reference's op order (PyMC3's
 * MarginalSparse, models/bayesian_sgpr_hmc.py:66-71: A = L^-1 K_uf first) keeps -- measured over lengthscales 0.2 .. 20 x noise
 * 0.01 .. 3, profiles/r04_theta_sweep_streaming.jsonl.  sgp_kuu_inverse_trace: trace_out[0] = tr(K_uu^-1) = ||L^-1||_F^2 from the
 * output of sgp_kuu_factor (trace_out: sgp_kuu_inverse_trace_len() doubles, the rest is scratch); sgp_streaming_error_estimate:
 * est[0] = 2^-53 max_i Phi_ii tr(K_uu^-1) / (s2 N), a first-order estimate of |dF| / N in the streaming order from the (all-reduced)
 * statistics.  A caller compares it with its tolerance (core.py: 1e-9) and re-evaluates through sgp_suffstats_fwd_whitened when it is
 * exceeded.  Both sums run in a fixed order: ranks that hold the same inputs get the same bits and take the same decision.       */
size_t sgp_kuu_inverse_trace_len(void);
int sgp_kuu_inverse_trace(const double* kuu_linv, int M, double* trace_out, sgp_stream_t stream);
int sgp_streaming_error_estimate(const double* stats, const double* trace_inv, double s2, int64_t N, int M, double* est,
                                 sgp_stream_t stream);
/* est[0] = 2^-53 sf2^2 tr(K_uu^-1) / s2: the same estimate with max_i Phi_ii at its upper bound N sf2^2 (a stationary profile is
 * <= 1) -- for a caller that is evaluating in the WHITENED order and wants to know when the streaming order is worth trying again
 * (core.py multiplies it by the ratio estimate / bound it saw at the theta where the guard tripped). */
int sgp_streaming_error_bound(const double* trace_inv, double sf2, double s2, double* est, sgp_stream_t stream);
/* ABI version 3: both numbers in one launch, from whatever an evaluation order holds of Phi's diagonal (est: TWO doubles):
 *   est[0] = 2^-53 max_i diag[i * stride] tr(K_uu^-1) / (s2 N)  -- diag = the all-reduced Phi with stride M + 1 (streaming order), the
 *            all-reduced phi_diag of sgp_ctx_suffstats_fwd_extended with stride 1 (extended order), or NULL (whitened order: est[0] = est[1]);
 *   est[1] = 2^-53 sf2^2 tr(K_uu^-1) / s2                       -- the estimate's upper bound (max_i Phi_ii <= N sf2^2).
 * The estimate is exact in the two orders that form Phi: the tier an evaluation NEEDS is then a function of its own theta, and what a
 * caller remembers from earlier evaluations (core.py: estimate / bound of the last evaluation that knew both) only decides where the
 * next one starts, never what is accepted.
 * A PROPERTY OF THE BOUND callers must know: the three evaluation orders (sgp_suffstats_fwd + sgp_bound_from_stats; sgp_suffstats_fwd_extended;
 * sgp_suffstats_fwd_whitened[_rows] + sgp_bound_from_whitened_stats) compute the same scalar to ~1e-9 per datum, NOT to the same bits, and a
 * guarded caller may accept an evaluation in a higher order than its own estimate requires when its previous evaluations suggested so:
 * F(theta) from such a caller is a function of theta up to that difference, and of the call history within it ("The guard").  */
int sgp_streaming_error_report(const double* diag, int64_t stride, const double* trace_inv, double sf2, double s2, int64_t N, int M,
                               double* est, sgp_stream_t stream);
size_t sgp_kuu_factor_len(int M);
size_t sgp_kuu_factor_workspace_bytes(int M);
int sgp_kuu_factor(const double* Kuu, int M, double* Linv_out, int* info,
                   void* ws, size_t ws_bytes, sgp_stream_t stream);
/* The same call with tr(K_uu^-1) delivered by its last launch: trace_out (sgp_kuu_inverse_trace_len() doubles, NULL = not wanted) receives
 * exactly what sgp_kuu_inverse_trace(Linv_out, M, trace_out) would write (same bits), two launches and their gaps earlier -- the K_uu chain
 * is on the critical path of small shards.  Five launches in all: L^-1 is formed inside the factorization's launch.   */
int sgp_kuu_factor_ex(const double* Kuu, int M, double* Linv_out, int* info, double* trace_out,
                      void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_bound_from_stats(const double* Kuu, const double* Phi, const double* b,
                         const double* yy, const double* kappa, double s2, int64_t N, int M,
                         int with_adjoints, double* out,
                         double* Phibar, double* bbar, double* Kuubar, double* factors,
                         const double* kuu_linv /* from sgp_kuu_factor, or NULL */,
                         int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- whitened statistics: the PyMC3 op order for small / ill-conditioned problems ----------------------------
 * pm.gp.MarginalSparse (models/bayesian_sgpr_hmc.py:66,71; experiments/co2_bayesian_sgpr_hmc.py:150-158) forms
 * A = L^-1 K_uf (M x N) and B = I + A A^T / s2.  B is then positive definite by construction and F keeps ~1e-10
 * absolute accuracy at cond(Kuu) ~ 1e8, where W = L^-1 (K_uf K_fu) L^-T formed from the streamed Phi is off by up
 * to 1 and chol(B) can fail (profiles/r02_logp_noise.json).  The price is a second N M^2 product and the
 * materialised A (row chunks of 32768), so this path is for N M up to a few million; the callers of this
 * library choose it below 2^20 row x inducing pairs.  Same kernel_id / inv_ls conventions as sgp_suffstats_fwd.
 *   W = A A^T (M x M, ld M), u = A y (M), yy, kappa as in sgp_suffstats_fwd; kuu_linv from sgp_kuu_factor.
 * Ranks all-reduce [W | u | yy | kappa] exactly like [Phi | b | yy | kappa] (L is replicated).
 * sgp_bound_from_whitened_stats is sgp_bound_from_stats without the L^-1 . L^-T sandwich; Phibar, bbar, Kuubar are
 * the adjoints with respect to the UNwhitened Phi, b, Kuu, so sgp_suffstats_bwd / sgp_kuu_bwd follow unchanged.  */
size_t sgp_suffstats_whitened_workspace_bytes(int64_t N, int M, int d);
int sgp_suffstats_fwd_whitened(const double* X, int64_t ldx, const double* y,
                               const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                               int64_t N, int M, int d, int kernel_id, const double* kuu_linv,
                               double* W, double* u, double* yy, double* kappa,
                               void* ws, size_t ws_bytes, sgp_stream_t stream);
/* The same statistics for a LARGE shard, in sgp_suffstats_fwd's streaming layout (round 4: the streaming-order guard of a caller sends
 * whole 10^6-row evaluations here): kernel assembly, ONE product T = K'_fu L^-T (clipped to L^-T's triangle), the tuned
 * fp64 contraction T^T T and the fixed-order reductions, instead of 31 x 4 launches over 32768-column chunks.  Stationary kernels only
 * (SGP_ERR_ARG for SGP_KERNEL_COMPOSITE).  T_out (DEVICE, sgp_kfu_len(N, M) doubles, or NULL): T is left there, unit amplitude, for
 * sgp_suffstats_bwd_factored_ex; the workspace is sized with caller_owns_t = 1 then.  Same outputs and all-reduce as above; the
 * results agree with sgp_suffstats_fwd_whitened to fp64 rounding of a different summation order, not bit for bit.                   */
size_t sgp_suffstats_whitened_rows_workspace_bytes(int64_t N, int M, int d, int caller_owns_t);
int sgp_suffstats_fwd_whitened_rows(const double* X, int64_t ldx, const double* y,
                                    const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                                    int64_t N, int M, int d, int kernel_id, const double* kuu_linv,
                                    double* W, double* u, double* yy, double* kappa, double* T_out,
                                    void* ws, size_t ws_bytes, sgp_stream_t stream);
/* The EXTENDED streaming order (round 4): the same statistics [W | u | yy | kappa] from the streaming design -- Phi = K_uf K_fu on the
 * integer matrix cores with 34 instead of 28 digit pairs and a double-double fold / reduction (2^-61 of the largest entry instead of
 * fp64's 2^-53), W = L^-1 Phi L^-T by two double-double M^3 products, u = L^-1 (K_uf y) in fp64.  What the explicit-inverse sandwich
 * amplifies is 2^8 (level 1) or 2^16 (level 2) times smaller than in sgp_suffstats_fwd + sgp_bound_from_stats: a caller whose
 * sgp_streaming_error_estimate exceeds its tolerance by less than that factor can stay in the streaming design for its VALUE (one
 * N M^2 contraction, 14.0 / 16.3 instead of 11.7 ms at C5) instead of paying the whitened order's two extra N M^2 products.  Stationary kernels only; the integer contraction is used whatever the context's
 * contraction mode says.  Kfu_out (DEVICE, sgp_kfu_len(N, M) doubles, or NULL): the fp64 K'_fu for sgp_suffstats_bwd with the explicit
 * Phibar that sgp_bound_from_whitened_stats returns.  THAT is this order's limit: Phibar K_uf cancels, and against the factored pass 2
 * of the whitened order the gradients are good to 1e-6 only while the estimate is within ~3 x a 1e-9 tolerance (measured at C5:
 * 7e-7 at an estimate of 4e-9, 3e-5 at 8e-8, 3e-2 at 2e-7) -- values hold 1.6e-10 per datum up to estimates of 3e-5 (level 2).
 * Same all-reduce as the other two orders.
 * level: 1 = 34 digit pairs (p + r >= 5, Phi to 2^-61, 14.0 ms of contraction at C5), 2 = 39 pairs (p + r >= 4, 2^-69, 16.3 ms).       */
size_t sgp_suffstats_extended_workspace_bytes(int64_t N, int M, int d);
int sgp_suffstats_fwd_extended(const double* X, int64_t ldx, const double* y,
                               const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                               int64_t N, int M, int d, int kernel_id, const double* kuu_linv, int level,
                               double* W, double* u, double* yy, double* kappa, double* Kfu_out,
                               void* ws, size_t ws_bytes, sgp_stream_t stream);
/* Round 6: the explicit Phibar of that order formed in DOUBLE-DOUBLE.  Cw (DEVICE, M x M, ld M) = C = I - B^-1 - g g^T / s2^2 as
 * sgp_bound_from_whitened_stats_ex returns it, kuu_linv the padded L^-1 of sgp_kuu_factor: Phibar = L^-T (C / 2 s2) L^-1 by two
 * double-double products (1.3 ms at M = 1024), its leading word to Phibar_hi (M x M, ld M) -- a drop-in for the Phibar the bound
 * returns, whose fp64 FORMATION error (eps |L^-T| |C| |L^-1| >> eps |Phibar|) is what limits the extended order's gradients
 * (tests/studies/explicit_phibar_pass2.py: 5-15 x closer gradients from the leading word alone) -- and, if Phibar_lo != NULL, the
 * trailing word (35-700 x with a low-precision product K' Phibar_lo added in pass 2).                                         */
size_t sgp_phibar_dd_workspace_bytes(int M);
int sgp_phibar_dd(const double* Cw, const double* kuu_linv, int M, double s2, double* Phibar_hi, double* Phibar_lo, void* ws,
                  size_t ws_bytes, sgp_stream_t stream);
/* ... and the pass-2 contribution of the trailing word: g_ls (d doubles) and g_sf2 receive, IN PLACE, what sgp_suffstats_bwd would have added
 * had its Phibar carried Phibar_lo as well -- dC = K' Phibar_lo on the fp16 matrix cores (three digits are all a 2^-53-relative term
 * needs), contracted with dK there as well (hi + lo fp16 pairs of the centred inputs; an inducing point more than 128 lengthscales from the
 * mean inducing point is beyond that format: delta then receives NaN and nothing is added).  Call it behind sgp_suffstats_bwd (same stream, same inputs, Phibar = the leading word, the same
 * caller-owned fp64 K'_fu as Kfu_in) and before the gradients are all-reduced.  RBF kernel (SGP_ERR_ARG otherwise), any d <= SGP_MAX_DIM (groups of eight dimensions);
 * dF/dZ is not corrected.  delta (DEVICE, d + 1 doubles, or NULL) receives the correction itself, [d lengthscales | sf2]: its size against
 * the gradient is the a-posteriori check a caller applies before trusting the explicit pass 2 at a theta (measured at C5 over 34 theta
 * against the factored pass 2 of the whitened order: what is left after the correction is <= 5 % of it while the streaming-order
 * estimate is <= 1e-6 -- profiles/r06_extended_order_gradients_*; core.py accepts corrections up to 1e-4 of the gradient).            */
size_t sgp_suffstats_bwd_lo_workspace_bytes(int64_t N, int M, int d);
int sgp_suffstats_bwd_lo(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                         const double* Phibar_lo, const double* Kfu_in, int64_t N, int M, int d, int kernel_id, double* g_ls,
                         double* g_sf2, double* delta, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* ... with the fp16 image of K'_fu that sgp_suffstats_fwd_extended_f16 / sgp_ctx_suffstats_fwd_extended_f16 leave in Kfu_f16_out (DEVICE,
 * sgp_kfu_len(N, M) 16-bit words): the conversion pass (2.0 of 4.6 ms at C5) is not run, Kfu_in may be NULL and the workspace is the _ex
 * size with have_f16 = 1 (2 bytes per element of K'_fu smaller).  Same results, bit for bit.                                              */
size_t sgp_suffstats_bwd_lo_workspace_bytes_ex(int64_t N, int M, int d, int have_f16);
int sgp_suffstats_bwd_lo_f16(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                             const double* Phibar_lo, const double* Kfu_in, const uint16_t* Kfu_f16_in, int64_t N, int M, int d,
                             int kernel_id, double* g_ls, double* g_sf2, double* delta, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* sgp_suffstats_fwd_extended_ex with one output more: Kfu_f16_out (DEVICE, sgp_kfu_len(N, M) 16-bit words, or NULL; needs Kfu_out) */
int sgp_suffstats_fwd_extended_f16(const double* X, int64_t ldx, const double* y,
                                   const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                                   int64_t N, int M, int d, int kernel_id, const double* kuu_linv, int level,
                                   double* W, double* u, double* yy, double* kappa, double* Kfu_out, uint16_t* Kfu_f16_out,
                                   double* phi_diag, void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_ctx_suffstats_fwd_extended_f16(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y,
                                       const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                                       int64_t N, int M, int d, int kernel_id, const double* kuu_linv, int level,
                                       double* W, double* u, double* yy, double* kappa, double* Kfu_out, uint16_t* Kfu_f16_out,
                                       double* phi_diag, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* ... with phi_diag (ABI version 3; see sgp_ctx_suffstats_fwd_extended) */
int sgp_suffstats_fwd_extended_ex(const double* X, int64_t ldx, const double* y,
                                  const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                                  int64_t N, int M, int d, int kernel_id, const double* kuu_linv, int level,
                                  double* W, double* u, double* yy, double* kappa, double* Kfu_out, double* phi_diag,
                                  void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_bound_from_whitened_stats(const double* W, const double* u, const double* yy, const double* kappa,
                                  double s2, int64_t N, int M, int with_adjoints, double* out,
                                  double* Phibar, double* bbar, double* Kuubar, double* factors,
                                  const double* kuu_linv, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);

/* The same call with one more output, Cw (M x M, DEVICE, may be NULL): C = I - B^-1 - g g^T / s2^2, the whitened core of the
 * adjoint 2 s2 Phibar = L^-T C L^-1 -- what sgp_suffstats_bwd_factored takes instead of Phibar.  Needs with_adjoints = 1.   */
int sgp_bound_from_whitened_stats_ex(const double* W, const double* u, const double* yy, const double* kappa,
                                     double s2, int64_t N, int M, int with_adjoints, double* out,
                                     double* Phibar, double* bbar, double* Kuubar, double* factors,
                                     const double* kuu_linv, int* info, double* Cw,
                                     void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- single-launch evaluation for small problems (M <= 128; stationary kernels d <= 24, composite d <= 8) -------------------------
 * The size class of the reference's own HMC runs (models/bayesian_sgpr_hmc.py:58-80,144-157: N ~ 250-1300, M = 100).
 * ONE cooperative kernel launch evaluates the bound and its gradient in the PyMC3 op order (A = L^-1 K_uf by blocked
 * substitution on the matrix cores; B = I + A A^T / s2), reading the hyper-parameters from DEVICE memory:
 *   mode SGP_SMALL_NATURAL : theta = [ls_1..d | sf2 | s2]                    out = [F | dF/dls_1..d | dF/dsf2 | dF/ds2 | logmarg | trace]
 *   mode SGP_SMALL_HMC     : theta = [log ls_1..d | log sig_f | log sig_n]   out = [logp | dlogp/dtheta (d + 2) | logmarg | trace]
 *                            logp = F + log Gamma(ls; 2, 1) + log HalfCauchy(sig_f; 1) + log HalfCauchy(sig_n; 1) + log-Jacobians,
 *                            the NUTS target of models/bayesian_sgpr_hmc.py:60-71; theta outside exp()'s range gives logp = -inf
 * out (d + 5 doubles), g_Z (M*d, ld d, optional: NULL skips dF/dZ) and info are DEVICE pointers; nothing is synchronised.
 * want_grad = 0 writes out[0] (and the two parts) only.  The first sgp_small_sync_bytes() bytes of `ws` must be zero
 * before the FIRST launch on a workspace (the kernel leaves them zero; after info = SGP_INFO_TIMEOUT zero them again).
 * All workgroups (1 + ceil(M/64) + row workgroups: one per 64-row slab up to min(208, CUs - 3), evened out beyond) must be co-resident.  The library checks the grid against
 * (occupancy of the kernel) x (CUs of the device, or the calling thread's sgp_set_cu_budget() for a CU-masked stream):
 * sgp_small_supported() answers 0 and the launch returns SGP_ERR_LAUNCH when it cannot fit (use the multi-launch entry
 * points then).  A plain launch, not hipLaunchCooperativeKernel: beside a kernel that already FILLS the device the
 * workgroups become resident as that kernel's retire -- late, never deadlocked (the chain workgroup is block 0 and is
 * dispatched first) -- and a bounded spin reports SGP_INFO_TIMEOUT instead of hanging.
 * One workspace = one stream: the sync words live in `ws`, so two launches on the same workspace must not overlap
 * (give every concurrent stream its own workspace).                                                                       */
#define SGP_SMALL_NATURAL 0
#define SGP_SMALL_HMC 1
int sgp_small_supported(int64_t N, int M, int d, int kernel_id);
/* measurement aid: device buffer of (3 + 208) * 16 uint64 filled with s_memrealtime ticks (100 MHz) at the phase
 * boundaries of every workgroup by the following launches; NULL (default) switches it off */
void sgp_small_debug_stamps(void* dev_buffer);
size_t sgp_small_workspace_bytes(int64_t N, int M, int d);
size_t sgp_small_sync_bytes(void);
int sgp_small_eval(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                   const double* theta, int64_t N, int M, int d, int kernel_id, double jitter, int mode,
                   int want_grad, double* out, double* g_Z, int* info,
                   void* ws, size_t ws_bytes, sgp_stream_t stream);

/* The same launch for SGP_KERNEL_COMPOSITE (the reference's CO2 covariance, experiments/co2_bayesian_sgpr_hmc.py:107-149;
 * d <= 8, no dF/dZ).  `structure` (HOST, SGP_COMP_LEN doubles) is a valid parameter block: it fixes the term / factor types
 * and the value of every parameter the caller does not pass in theta.
 *   mode SGP_SMALL_NATURAL : theta (device) = [parameter block (SGP_COMP_LEN) | s2]   out = [F | dF/d block (SGP_COMP_LEN, zero
 *                            at the structural slots) | dF/ds2 | logmarg | trace]; the free_* arguments are ignored
 *   mode SGP_SMALL_HMC     : theta (device) = [log of the n_free free parameters | log sigma]   out = [logp | dlogp/dtheta
 *                            (n_free + 1) | logmarg | trace].  Free parameter k fills block slot free_slot[k] (HOST arrays;
 *                            free_role[k] = 0: an amplitude slot, the parameter is the amplitude's standard deviation and the
 *                            slot gets its square; 1: a lengthscale slot; 2: an aux slot) and carries the prior log-parameter ~
 *                            Normal(0, free_prior_sd[k]); sigma ~ HalfNormal(1), log-transformed: the PyMC3 model of
 *                            co2_bayesian_sgpr_hmc.py:99-158.  n_free <= 20.                                              */
int sgp_small_eval_composite(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                             const double* theta, const double* structure, int n_free, const int* free_slot,
                             const int* free_role, const double* free_prior_sd, int64_t N, int M, int d,
                             double jitter, int mode, int want_grad, double* out, int* info,
                             void* ws, size_t ws_bytes, sgp_stream_t stream);

/* S evaluations in one launch (the same X, y, Z; S hyper-parameter sets): the theta-averaged loss of the reference's
 * alternating schedule (models/bayesian_sgpr_hmc.py:121-134).  thetas (S x (d + 2)), outs (S x (d + 5)), g_Z (S x M x d or
 * NULL), infos (S) are device arrays with sgp_small_eval's per-sample layout; theta_scratch: d + 2 doubles.          */
int sgp_small_eval_batch(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                         const double* thetas, int S, int64_t N, int M, int d, int kernel_id, double jitter, int mode,
                         int want_grad, double* theta_scratch, double* outs, double* g_Z, int* infos,
                         void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- device-resident NUTS over the same target ------------------------------------------------
 * pm.sample(n_draws, tune=n_tune, chains=1) with pm.NUTS() defaults (models/bayesian_sgpr_hmc.py:73-78) in ONE persistent
 * launch: multinomial NUTS, dual-averaging step size, jitter+adapt_diag mass matrix; the sampler state, theta and the
 * momentum never leave the GPU, every leapfrog is one cooperative evaluation (SGP_SMALL_HMC) inside the running kernel.
 *   q0 (device, d + 2): start in the unconstrained space; theta_scratch (device, d + 2) and out (device, d + 5) are work buffers
 *   samples (device, n_draws x (d + 2)): post-tuning draws, unconstrained;  stats (device, sgp_small_nuts_stat_cols() arrays
 *   of n_draws... see below);  counters (device, 2 x int64): evaluations (leapfrogs + 1), draws finished;  info as above.
 *   stats layout: n_draws rows x 8 columns [step_size, tree_size, depth, mean_tree_accept, diverging, energy, logp,
 *   cumulative evaluations] followed by n_draws doubles: seconds spent on each draw (device clock).
 * Random numbers: splitmix64 seeded with `seed` (host twin: hmc.SplitMix) -- a run is reproducible bit for bit.
 * Same workspace, co-residency and zeroed-sync-words rules as sgp_small_eval; max_treedepth <= 11.                       */
size_t sgp_small_nuts_stat_cols(void);
int sgp_small_nuts(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* q0,
                   int64_t N, int M, int d, int kernel_id, double jitter, int n_tune, int n_draws, int max_treedepth,
                   double step_scale, double target_accept, uint64_t seed, double* theta_scratch, double* samples,
                   double* stats, long long* counters, double* out, int* info,
                   void* ws, size_t ws_bytes, sgp_stream_t stream);

/* device-resident NUTS over the composite target of sgp_small_eval_composite (mode SGP_SMALL_HMC): q0, theta_scratch and the
 * rows of `samples` have n_free + 1 entries (<= 18), `out` n_free + 4.                                                    */
int sgp_small_nuts_composite(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                             const double* q0, const double* structure, int n_free, const int* free_slot,
                             const int* free_role, const double* free_prior_sd, int64_t N, int M, int d,
                             double jitter, int n_tune, int n_draws, int max_treedepth, double step_scale,
                             double target_accept, uint64_t seed, double* theta_scratch, double* samples,
                             double* stats, long long* counters, double* out, int* info,
                             void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- streaming pass 2: gradients through Kuf -------------------------------------------------------
 * Kbar_uf = 2 Phibar Kuf + bbar y^T is formed tile by tile and contracted with dKuf/d(.) on the fly.
 * Writes (overwrites) g_ls[d] = dF/d lengthscale_j, g_sf2[1] = dF/d sf2 (including the kappa term
 * kappabar * N), g_Z[M*d] (ld d; skipped when g_Z == NULL).  Local shard only; caller all-reduces.  */
size_t sgp_suffstats_bwd_workspace_bytes(int64_t N, int M, int d);
size_t sgp_suffstats_bwd_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_kfu); /* with Kfu_in from pass 1 */
int sgp_suffstats_bwd(const double* X, int64_t ldx, const double* y,
                      const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                      const double* Phibar, const double* bbar, double kappabar,
                      const double* Kfu_in /* from sgp_suffstats_fwd's Kfu_out, or NULL */,
                      int64_t N, int M, int d, int kernel_id,
                      double* g_ls, double* g_sf2, double* g_Z,
                      void* ws, size_t ws_bytes, sgp_stream_t stream);
/* Pass 2 from the FACTORED adjoint, for the whitened evaluation order on ill-conditioned K_uu (inducing inputs closer than the
 * lengthscale: the reference's CO2 model with M = 480 random training times, experiments/co2_bayesian_sgpr_hmc.py:384).
 * Phibar = L^-T C L^-1 / (2 s2) has entries of size cond(K_uu) that cancel in Phibar K_uf: formed explicitly it leaves 1e-2 ..
 * 1e-1 relative error on the gradients at cond 3e9 .. 5e10; applied to K_fu one factor after the other, 1e-5
 * (tests/studies/logp_noise.py).  kuu_linv: Mp x Mp from sgp_kuu_factor, Cw: M x M from sgp_bound_from_whitened_stats_ex, bbar
 * and kappabar as for sgp_suffstats_bwd; same outputs.  K_fu is materialised whole (meant for N M <= ~2^22).               */
size_t sgp_suffstats_bwd_factored_workspace_bytes(int64_t N, int M, int d);
int sgp_suffstats_bwd_factored(const double* X, int64_t ldx, const double* y,
                               const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                               const double* kuu_linv, const double* Cw, double s2,
                               const double* bbar, double kappabar,
                               int64_t N, int M, int d, int kernel_id,
                               double* g_ls, double* g_sf2, double* g_Z,
                               void* ws, size_t ws_bytes, sgp_stream_t stream);

/* The same with T = K'_fu L^-T handed over from pass 1 (T_in: sgp_suffstats_fwd_whitened_rows' T_out, or NULL = as above): the
 * assembly and the first N M^2 product are not repeated, and the workspace (caller_owns_t = 1) holds neither K'_fu nor T.
 * Since round 4 both entry points multiply the two trailing factors first, Q = (Cw / s2)(L^-1 / 2) (an M^3 product), and leave ONE
 * N M^2 product behind T; SGP_BWD_FULLY_FACTORED=1 in the environment keeps the three-product chain.                              */
size_t sgp_suffstats_bwd_factored_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_t);
int sgp_suffstats_bwd_factored_ex(const double* X, int64_t ldx, const double* y,
                                  const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                                  const double* kuu_linv, const double* Cw, double s2,
                                  const double* bbar, double kappabar,
                                  int64_t N, int M, int d, int kernel_id, const double* T_in,
                                  double* g_ls, double* g_sf2, double* g_Z,
                                  void* ws, size_t ws_bytes, sgp_stream_t stream);

/* gradient through Kuu: ADDS sum(Kuubar o dKuu/d(.)) into g_ls, g_sf2, g_Z (g_Z may be NULL).
 * Kuubar is used as a symmetric matrix.  Replicated on every rank (call it after the all-reduce of
 * the streamed gradients, or on one rank before it).                                             */
size_t sgp_kuu_bwd_workspace_bytes(int M, int d);
int sgp_kuu_bwd(const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                const double* Kuubar, int M, int d, int kernel_id,
                double* g_ls, double* g_sf2, double* g_Z,
                void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- posterior predictive at T test rows (models/sgpr.py:150-160, 256-286) ----------------------
 * mean = K*u Sigma^-1 b / s2 ; var = k** - |L^-1 k_u*|^2 + |LB^-1 L^-1 k_u*|^2 (+ s2 if pred_noise)
 * cov (T x T, optional, may be NULL) is the full predictive covariance with the base kernel K**.   */
size_t sgp_predict_workspace_bytes(int64_t T, int M, int d, int want_cov);
int sgp_predict(const double* Xs, int64_t ldxs, int64_t T,
                const double* Z, int64_t ldz, const double* inv_ls, double sf2, double s2,
                const double* factors, int M, int d, int kernel_id, int pred_noise,
                double* mean, double* var, double* cov,
                void* ws, size_t ws_bytes, sgp_stream_t stream);

/* ---- uncollapsed (SVGP) minibatch bound: SURVEY section 8 f-3 ------------------------------------------
 * VariationalELBO(likelihood, model, num_data=N)(model(x_batch), y_batch) over a whitened VariationalStrategy with
 * a Cholesky variational distribution (models/svgp.py:37,46,88-127), and loss.backward() through it.
 *   m[M], LS[M*M] (lower triangle used, ld M): q(u) = N(m, LS LS^T) in the whitened parametrisation
 *   out[0] = mean_b E_q log p(y_b|f_b) - KL / N_total ; out[1] = sum_b E_q log p ; out[2] = KL
 *   likelihood_id: SGP_LIK_GAUSSIAN (noise s2) or SGP_LIK_BERNOULLI_PROBIT (y in {-1,+1}, 20-point Gauss-Hermite)
 *   with_grads: d out[0] / d{m, LS (lower), Z, lengthscale_j, sf2, s2}; info as for the collapsed bound (1..M).  */
#define SGP_LIK_GAUSSIAN 0
#define SGP_LIK_BERNOULLI_PROBIT 1
size_t sgp_svgp_workspace_bytes(int64_t B, int M, int d);
int sgp_svgp_elbo(const double* Xb, int64_t ldx, const double* yb, int64_t B,
                  const double* Z, int64_t ldz, const double* inv_ls, double sf2, double s2, double jitter,
                  const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id,
                  int with_grads, double* out,
                  double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2, double* g_s2,
                  int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* The same bound at S hyper-parameter samples in ONE chain of launches (S <= 8): the 5 reparametrised theta samples per
 * minibatch of BayesianStochasticVariationalGP (models/bayesian_svgp.py:156-167) share the minibatch, Z, m, L_S and the KL
 * term; every launch carries the sample index, so the chain is ~36 launches whatever S is.
 *   inv_ls (S x d), sf2 (S), s2 (S): HOST arrays, sample-major (s2 ignored for SGP_LIK_BERNOULLI_PROBIT)
 *   out (S x 4: sgp_svgp_elbo's three values per sample, then the sample's status word as a double, so that one copy brings
 *   bounds and statuses to the host), info (S ints), and with_grads: g_m (S x M), g_LS (S x M x M), g_Z (S x M x d),
 *   g_ls (S x d), g_sf2 (S), g_s2 (S) -- DEVICE arrays, the gradient of out[s][0] in slice s.
 * sgp_svgp_batch_combine: the reverse pass of a loss sum_s w_s out[s][0] in one launch (weights: HOST, S doubles) -- the shared
 *   parameters' gradients summed over the samples (gm_out [M], gLS_out [M x M], gZ_out [M x d]) and the per-sample
 *   hyper-parameter gradients scaled and packed, gtheta_out [S x (d + 2)] = w_s [d/dsf2 | d/dls_1..d | d/ds2].            */
size_t sgp_svgp_batch_workspace_bytes(int64_t B, int M, int d, int S);
int sgp_svgp_elbo_batch(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                        int S, const double* inv_ls, const double* sf2, const double* s2, double jitter,
                        const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id,
                        int with_grads, double* out,
                        double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2, double* g_s2,
                        int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* The same call in two halves, for callers that want the bounds back while the device still runs the reverse chain:
 * _forward = sgp_svgp_elbo_batch(with_grads = 0) plus the likelihood's d/ds2 (g_s2: S doubles, may be NULL); _reverse runs the
 * reverse pass from the state _forward left in `ws` (same arguments, same workspace, nothing else in between).            */
int sgp_svgp_elbo_batch_forward(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                                int S, const double* inv_ls, const double* sf2, const double* s2, double jitter,
                                const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id,
                                double* out, double* g_s2, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_svgp_elbo_batch_reverse(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                                int S, const double* inv_ls, const double* sf2, const double* s2, double jitter,
                                const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id,
                                double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2,
                                void* ws, size_t ws_bytes, sgp_stream_t stream);
/* latent predictive of q(f*) at T rows for S hyper-parameter samples in one chain (mean, var: S x T on the device; the mixture
 * predictive of BayesianStochasticVariationalGP, models/bayesian_svgp.py:183-207); workspace: sgp_svgp_batch_workspace_bytes(T, ..) */
int sgp_svgp_predict_batch(const double* Xs, int64_t ldxs, int64_t T, const double* Z, int64_t ldz, int S, const double* inv_ls,
                           const double* sf2, double jitter, const double* m, const double* LS, int M, int d, int kernel_id,
                           double* mean, double* var, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
int sgp_svgp_batch_combine(int S, const double* weights, int M, int d, const double* g_m, const double* g_LS, const double* g_Z,
                           const double* g_ls, const double* g_sf2, const double* g_s2, double* gm_out, double* gLS_out,
                           double* gZ_out, double* gtheta_out, sgp_stream_t stream);
/* ---- mixture posterior predictive of the collapsed bound: S hyper-parameter samples per chain of launches (S <= 8) -----------
 * mixture_posterior_predictive(model, test_x, trace_hyper) of models/bayesian_sgpr_hmc.py:198-231 evaluates one full
 * predictive per theta sample in a Python loop; here the S samples ride in the launch grid.  PyMC3 op
 * order per sample (A = L^-1 K_uf, B = I + A A^T / s2); stationary kernels; X, y: ALL training rows (single rank).
 *   inv_ls (S x d), sf2 (S), s2 (S): HOST arrays;   mean, var (S x T; var may be NULL), cov (S x T x T or NULL; T <= 8192): DEVICE
 *   info (S ints): 0, 1..M (K_uu, incl. the conditioning gate of sgp_set_cond_limit), M+1..2M (B)
 *   gate_info (S ints or NULL, needs cov): status of cholesky(cov + gate_jitter I) per sample -- the reference's PSD gate
 *   (cov + 1e-4 I, :225-229) for all S samples in one dataflow launch; 0 = positive definite.                              */
size_t sgp_mixture_predict_workspace_bytes(int64_t N, int64_t T, int M, int d, int S, int want_cov, int want_gate);
int sgp_mixture_predict(const double* X, int64_t ldx, const double* y, int64_t N, const double* Xs, int64_t ldxs, int64_t T,
                        const double* Z, int64_t ldz, int S, const double* inv_ls, const double* sf2, const double* s2,
                        double jitter, int M, int d, int kernel_id, int pred_noise, double gate_jitter,
                        double* mean, double* var, double* cov, int* info, int* gate_info,
                        void* ws, size_t ws_bytes, sgp_stream_t stream);
/* latent predictive mean / variance of q(f*) at T rows (models/svgp.py:132-141 continues through the likelihood) */
int sgp_svgp_predict(const double* Xs, int64_t ldxs, int64_t T, const double* Z, int64_t ldz, const double* inv_ls,
                     double sf2, double jitter, const double* m, const double* LS, int M, int d, int kernel_id,
                     double* mean, double* var, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream);
/* host utility: n-point Gauss-Hermite rule for the standard normal (sum w_i f(x_i) ~ E f(N(0,1))) */
int sgp_gauss_hermite(int n, double* x, double* w);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SGP_H */
