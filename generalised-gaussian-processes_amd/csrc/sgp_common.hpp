// Shared device helpers for the gfx950 (CDNA4) sparse-GP kernels.
// fp64 throughout; wavefront = 64 lanes; MFMA = v_mfma_f64_16x16x4_f64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sgp.h"

namespace sgp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr int PADM = 128;  // every M x M matrix handled inside the library is padded to a multiple of this

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int padded_m(int M) { return round_up(M, PADM); }

// D(16x16) += A(16x4) * B(4x16).  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// it receives D[(l>>4) + 4r][l&15] in element r of the accumulator (r = 0..3).
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Hyper-parameters travel as kernel arguments (no device round trip per leapfrog).
struct KernArgs {
  double inv_ls[SGP_MAX_DIM];
  double sf2;
  int d;
};

// exp(x) for the kernel profiles (x <= 0 there; correct for any finite x, saturating to 0 below -745): range reduction
// x = k ln2 + r with the two-part ln2 of fdlibm, Taylor polynomial of degree 13 on |r| <= ln2 / 2 (remainder 4e-18
// relative), v_ldexp_f64.  19 fp64 VALU instructions against ~28 for the library routine; <= 2 ulp against libm over
// [-745, 0] (tests/test_gpu_parity.py::test_kernel_exp_accuracy).  Same-box A/B (tools/ab_build.sh, three alternations):
// kernel assembly 1.62 vs 1.66 ms -- it sits at the HBM write ceiling (5.05 TB/s), not at the VALU -- so this routine is
// used where values are generated (kprofile); in the epilogue of pass 2 its 13-deep dependent FMA chain is SLOWER than the
// library's shorter chains (34.8 vs 33.5 ms for the launch), so kprofile_grad keeps exp().
__device__ __forceinline__ double sgp_exp(double x) {
#if defined(SGP_AB_LIBRARY_EXP)  // A/B knob of tools/ab_build.sh only: the device library's routine
  return exp(x);
#endif
  x = (x < -800.0) ? -800.0 : x;  // not fmax(): a NaN distance (NaN in X / Z / a lengthscale) must stay NaN, not become k = 0
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1 / 13!
  p = fma(p, r, 2.08767569878681e-09);          // 1 / 12!
  p = fma(p, r, 2.505210838544172e-08);         // 1 / 11!
  p = fma(p, r, 2.755731922398589e-07);         // 1 / 10!
  p = fma(p, r, 2.7557319223985893e-06);        // 1 / 9!
  p = fma(p, r, 2.48015873015873e-05);          // 1 / 8!
  p = fma(p, r, 1.984126984126984e-04);         // 1 / 7!
  p = fma(p, r, 1.388888888888889e-03);         // 1 / 6!
  p = fma(p, r, 8.333333333333333e-03);         // 1 / 5!
  p = fma(p, r, 4.1666666666666664e-02);        // 1 / 4!
  p = fma(p, r, 1.6666666666666666e-01);        // 1 / 3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// The same function for the two kernel-assembly kernels, where it is ~40 % of the arithmetic (round 4): x = (32 k + j) ln2 / 32 + r,
// exp(x) = 2^k T[j] (1 + q(r)) with T[j] = 2^(j/32) from a 32-entry table in LDS (one ds_read_b64, no bank conflict: 32 entries = the
// 64 banks) and q = r (1 + r/2 + ... + r^5/720) on |r| <= ln2 / 64 (remainder 3.6e-18): 17 instead of 21 VALU instructions, and
// fma(T, q, T) rounds ONCE after the table value -- <= 1 ulp against libm (tests/test_gpu_parity.py::test_assembly_exp_accuracy).
// n = rint(32 x / ln2) is read off the low mantissa bits of a magic-constant sum (|n| < 2^16); C_HI has 17 trailing zero bits, so
// n C_HI is exact.  NaN stays NaN (the table index is masked), x < -800 saturates to 0.
constexpr int EXP_TAB_N = 32;
__device__ const double SGP_EXP2_TAB[EXP_TAB_N] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0, 0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0,
    0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0, 0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0, 0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0,
    0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0, 0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0, 0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0,
    0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
// call from every thread of the block before the barrier that precedes the first sgp_exp_tab
__device__ __forceinline__ void sgp_exp_tab_load(double* tab) {
  if (threadIdx.x < EXP_TAB_N) tab[threadIdx.x] = SGP_EXP2_TAB[threadIdx.x];
}
__device__ __forceinline__ double sgp_exp_tab(double x, const double* tab) {
#pragma clang fp contract(off)
  x = (x < -800.0) ? -800.0 : x;
  const double t = fma(x, 0x1.71547652b82fep+5, 0x1.8p52);  // 32 / ln2
  const int n = (int)__double_as_longlong(t);               // the low 32 bits: n as two's complement
  const double nf = t - 0x1.8p52;
  double r = fma(nf, -0x1.62e42fefa0000p-6, x);             // ln2 / 32, high part
  r = fma(nf, -0x1.cf79abc9e3b3ap-45, r);                   // ... low part
  const double T = tab[n & (EXP_TAB_N - 1)];
  double q = 1.388888888888889e-03;                         // 1 / 6!
  q = fma(q, r, 8.333333333333333e-03);
  q = fma(q, r, 4.1666666666666664e-02);
  q = fma(q, r, 1.6666666666666666e-01);
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  q *= r;
  return ldexp(fma(T, q, T), n >> 5);
}
// The two assembly kernels must write the same bits (test_int8_kept_block_equals_the_fp64_assembly).  The compiler's own contraction of
// the Matern prefactors came out differently in them (1 + a as fma(sqrt(r2), c, 1) in one, as an add of the rounded a in the other):
// no implicit contraction here, the fused operations are written out.
template <int KID>
__device__ __forceinline__ double kprofile_tab(double r2, const double* tab) {
#pragma clang fp contract(off)
  if constexpr (KID == SGP_KERNEL_RBF) {
    return sgp_exp_tab(-0.5 * r2, tab);
  } else if constexpr (KID == SGP_KERNEL_MATERN32) {
    const double a = 1.7320508075688772 * sqrt(r2);
    const double p = 1.0 + a;
    return p * sgp_exp_tab(-a, tab);
  } else {
    const double a = 2.23606797749979 * sqrt(r2);
    const double a2 = a * a, p1 = 1.0 + a;
    return fma(a2, 1.0 / 3.0, p1) * sgp_exp_tab(-a, tab);
  }
}

// k'(r2): the stationary profile WITHOUT the sf2 factor; r2 = scaled squared distance.
// Also returns h = dk'/d(r2) when asked (used by the backward pass).
template <int KID>
__device__ __forceinline__ double kprofile(double r2) {
  if constexpr (KID == SGP_KERNEL_RBF) {
    return sgp_exp(-0.5 * r2);
  } else if constexpr (KID == SGP_KERNEL_MATERN32) {
    const double a = 1.7320508075688772 * sqrt(r2);
    return (1.0 + a) * sgp_exp(-a);
  } else {
    const double a = 2.23606797749979 * sqrt(r2);
    return (1.0 + a + a * a * (1.0 / 3.0)) * sgp_exp(-a);
  }
}

template <int KID>
__device__ __forceinline__ void kprofile_grad(double r2, double& k, double& h) {
  if constexpr (KID == SGP_KERNEL_RBF) {
    k = exp(-0.5 * r2);
    h = -0.5 * k;
  } else if constexpr (KID == SGP_KERNEL_MATERN32) {
    const double a = 1.7320508075688772 * sqrt(r2);
    const double e = exp(-a);
    k = (1.0 + a) * e;
    h = -1.5 * e;
  } else {
    const double a = 2.23606797749979 * sqrt(r2);
    const double e = exp(-a);
    k = (1.0 + a + a * a * (1.0 / 3.0)) * e;
    h = -(5.0 / 6.0) * (1.0 + a) * e;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in every thread.  `red` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum256(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Workspace carving: 256-byte aligned bump allocator over the caller's scratch buffer.
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* r = reinterpret_cast<T*>(base ? base + off : nullptr);
    off += n * sizeof(T);
    return r;
  }
  size_t used() const { return (off + 255) & ~size_t(255); }
};

inline int check_launch() { return hipGetLastError() == hipSuccess ? SGP_OK : SGP_ERR_LAUNCH; }

}  // namespace sgp
