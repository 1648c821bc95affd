// Internal interface of the M x M dense back end (all matrices row-major fp64, every dimension a
// multiple of 64 -- callers pad to sgp::PADM = 128 -- leading dimensions in elements).
#pragma once
#include "sgp_common.hpp"

namespace sgp {

// C = alpha * op(A) * op(B) + beta * C, optionally batched (blockIdx.z) with element strides.
// klo/khi masks restrict the k-range of an output tile to where triangular operands are non-zero:
//   bit 0 -> bound by the tile's row range, bit 1 -> bound by the tile's column range
//   klo = max(enabled starts), khi = min(enabled ends)
// lower_only skips tiles strictly above the diagonal (they are left untouched).
struct GemmDesc {
  const double* A = nullptr;
  const double* B = nullptr;
  double* C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int64_t sA = 0, sB = 0, sC = 0;
  int m = 0, n = 0, k = 0, batch = 1;
  // a second, outer batch level (blockIdx.z = outer * batch + inner) with its own strides: S independent problems
  // (the SVGP bound at S hyper-parameter samples) around a product that is already batched (tri_inverse levels, k-slices)
  int batch2 = 1;
  int64_t s2A = 0, s2B = 0, s2C = 0;
  double alpha = 1.0, beta = 0.0;
  bool ta = false, tb = false;
  int klo_mask = 0, khi_mask = 0;
  bool lower_only = false;
  bool mirror = false;      // with lower_only: every tile below the diagonal is also stored transposed (a symmetric product computed once)
};
void gemm(const GemmDesc& g, hipStream_t st);
// Condition estimate of a factored matrix (sgp_tail.hip): cond_stats fills `scratch` (cond_scratch_doubles(M) doubles per matrix) from
// the factor L and its explicit inverse, cond_gate reports est = lambda_max_estimate / lambda_min_estimate > limit into info[s].
size_t cond_scratch_doubles(int M);
void cond_stats(const double* L, const double* Linv, int64_t ld, int64_t stride, int M, int S, double* scratch, hipStream_t st);
void cond_gate(const double* scratch, int M, int S, double limit, int* info, hipStream_t st);
// one workgroup of 256 threads over one matrix's scratch:  lam <= lambda_max(L L^T)  from max_j ||L e_j||^2 and ||L^T 1||^2 / M,
// inv_min <= 1 / lambda_min  from max_i ||e_i^T L^-1||^2 (`at` = that row).  Fixed order throughout; the results are valid in thread 0.
__device__ __forceinline__ void cond_estimate_block(const double* __restrict__ sc, int M, double* red /* 12 */, int* redi /* 4 */,
                                                    double& lam, double& inv_min, int& at) {
  const int nb = (M + 63) / 64, Mc = nb * 64;
  double hi = 0.0, rq = 0.0, best = -1.0;
  int a = 0;
  for (int j = threadIdx.x; j < M; j += 256) {
    double n2 = 0.0, c1 = 0.0;
    for (int br = j / 64; br < nb; ++br) {
      n2 += sc[(int64_t)br * Mc + j];
      c1 += sc[(int64_t)(nb + br) * Mc + j];
    }
    hi = fmax(hi, n2);
    rq = fma(c1, c1, rq);
  }
  const double* rowN = sc + (int64_t)2 * nb * Mc;
  for (int i = threadIdx.x; i < M; i += 256)
    if (rowN[i] > best) { best = rowN[i]; a = i; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi = fmax(hi, __shfl_xor(hi, o, 64));
    rq += __shfl_xor(rq, o, 64);
    const double ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(a, o, 64);
    if (ob > best || (ob == best && oa < a)) { best = ob; a = oa; }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = hi; red[4 + w] = rq; red[8 + w] = best; redi[w] = a; }
  __syncthreads();
  hi = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  rq = (red[4] + red[5]) + (red[6] + red[7]);
  best = red[8]; a = redi[0];
  for (int t = 1; t < 4; ++t)
    if (red[8 + t] > best || (red[8 + t] == best && redi[t] < a)) { best = red[8 + t]; a = redi[t]; }
  lam = fmax(hi, rq / (double)M);
  inv_min = best;
  at = a;
}
double cond_gate_limit();   // sgp_set_cond_limit's current value (sgp_tail.hip): the explicit-inverse paths refuse above it
int available_cus();       // CUs a launch of this host thread can occupy (device count, or the budget below)
void set_cu_budget(int n);  // CUs the calling host thread's launches may occupy (CU-masked streams); 0 = all
// the same product with the contraction cut into S slices (S * m * n doubles of scratch; falls back to gemm() when k is
// not a multiple of 16 S): for few output tiles and a long k
// (batch must be 1; batch2 problems are cut alike: scratch = batch2 * S * m * n doubles)
void gemm_splitk(const GemmDesc& g, int S, double* scratch, hipStream_t st);

// In-place lower Cholesky of the Mp x Mp matrix A (Mp multiple of 64); strictly-upper part of the
// result is zeroed.  Linv (Mp x Mp, same ld) receives the inverses of the 64 x 64 diagonal blocks of
// L (and is zero elsewhere) -- level 0 of tri_inverse().  info_base offsets the reported pivot index
// (info is only written when still 0, so two factorizations can share one flag).  scratch: potrf_scratch_ints(Mp)
// ints (tile-ready flags of the single-launch dataflow factorization; cleared here, reusable right after on the
// same stream).  info = -7777 reports a dataflow time-out (never expected; instead of a hang).
size_t potrf_scratch_ints(int Mp);
size_t potrf_flag_ints(int Mp);  // the leading part of the scratch that must be zero when the launch starts
// rhs / sol (optional, Mp doubles each): sol = L^-1 rhs, computed inside the same launch.  Linv may be null when
// the caller needs neither tri_inverse() nor the block inverses.
// caller_managed: the caller has already zeroed `scratch` on this stream and reads the abort flag
// (potrf_abort_flag) itself after the launch -- saves two tiny launches on the latency-critical tail.
// prepped: bit 0 -- the flags part of `scratch` is already zero on this stream, bit 1 -- Linv is already zero (a caller that fills
// its operand with a kernel of its own clears both there: two launches fewer on the K_uu chain, which IS C3's critical path)
// Returns true when Linv has received the WHOLE inverse (the chain-workgroup kernel forms its blocks inside the launch: sgp_potrf_chain.hpp),
// false when only the diagonal blocks are there and the caller still has to run tri_inverse().
bool potrf_lower(double* A, double* Linv, int64_t ld, int Mp, int* info, int info_base, int* scratch, hipStream_t st,
                 const double* rhs = nullptr, double* sol = nullptr, bool caller_managed = false, int prepped = 0);
// the word of `scratch` the dataflow launch raises when it gave up waiting (then info must become SGP_INFO_TIMEOUT)
const int* potrf_abort_flag(const int* scratch, int Mp);
// S independent factorizations in ONE dataflow launch (grid = workgroups x S): matrix s at A + s * stride (Linv likewise),
// its status word info[s], its scratch at scratch + s * potrf_scratch_ints(Mp).  4 launches whatever S is.
void potrf_lower_batch(double* A, double* Linv, int64_t ld, int Mp, int S, int64_t stride, int* info, int* scratch, hipStream_t st);

// Completes Linv (diagonal 64-blocks already inverted by potrf_lower) to the full inverse of L.
// tmp: Mp x Mp scratch with the same ld.
// nbatch > 1: the same for nbatch matrices `stride` doubles apart (L, Linv and tmp alike), in the same number of launches.
void tri_inverse(const double* L, double* Linv, double* tmp, int64_t ld, int Mp, hipStream_t st, int nbatch = 1, int64_t stride = 0);

// y = op(A) x for a lower-triangular (or general) Mp x Mp matrix; one wave per row.
void gemv(const double* A, int64_t ld, int Mp, bool trans, const double* x, double* y, hipStream_t st);

// dst (rows x cols, ld ldd) <- src (rs x cs, ld lds) zero padded; diag_pad: value put on the padded
// part of the diagonal (1.0 keeps padded Cholesky factors the identity).
void pad_copy(const double* src, int64_t lds, int rs, int cs, double* dst, int64_t ldd, int rows, int cols,
              double diag_pad, hipStream_t st);
// dst (rs x cs, ld ldd) <- top-left corner of src (ld lds)
void crop_copy(const double* src, int64_t lds, double* dst, int64_t ldd, int rs, int cs, hipStream_t st);
void fill_zero(double* p, size_t n, hipStream_t st);
// n ints <- 0 by a kernel launch (graph-replay safe; see sgp_dense.hip)
void zero_ints(int* p, int n, hipStream_t st);
// out = in^T for an n x n matrix (ld n, n a multiple of 32)
void transpose_square(const double* in, int n, double* out, hipStream_t st);
// upper triangle <- transpose of lower triangle
void mirror_lower(double* A, int64_t ld, int Mp, hipStream_t st);

}  // namespace sgp
