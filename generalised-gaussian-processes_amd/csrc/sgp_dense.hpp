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
};
void gemm(const GemmDesc& g, hipStream_t st);
// part[s][2 b], part[s][2 b + 1], b < ceil(M/64): per 64-column block of the S factors L (ld x ld each, `stride` doubles apart) the
// largest squared column norm and the sum of squared column sums; cond_lambda_max() turns them into a LOWER bound of lambda_max(L L^T)
void cond_colnorms(const double* L, int64_t ld, int64_t stride, int M, int S, double* part, hipStream_t st);
// part[s][2 b] = largest squared row norm of L^-1 in row block b, part[s][2 b + 1] = that row: 1 / lambda_min >= the largest of them
void cond_rownorms(const double* Linv, int64_t ld, int64_t stride, int M, int S, double* part, hipStream_t st);
__device__ __forceinline__ double cond_inv_lambda_min(const double* __restrict__ part, int npart, int* at) {
  double best = -1.0;
  int a = 0;
  for (int i = 0; i < npart; ++i)
    if (part[2 * i] > best) { best = part[2 * i]; a = (int)part[2 * i + 1]; }
  if (at) *at = a;
  return best;
}
__device__ __forceinline__ double cond_lambda_max(const double* __restrict__ part, int npart, int M) {
  double hi = 0.0, rq = 0.0;
  for (int i = 0; i < npart; ++i) {  // fixed order
    hi = fmax(hi, part[2 * i]);      // ||L e_j||^2 <= lambda_max
    rq += part[2 * i + 1];           // ||L^T 1||^2 = 1^T K 1
  }
  return fmax(hi, rq / (double)M);
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
void potrf_lower(double* A, double* Linv, int64_t ld, int Mp, int* info, int info_base, int* scratch, hipStream_t st,
                 const double* rhs = nullptr, double* sol = nullptr, bool caller_managed = false);
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
// upper triangle <- transpose of lower triangle
void mirror_lower(double* A, int64_t ld, int Mp, hipStream_t st);

}  // namespace sgp
