// Pass 2, the LOW WORD of a double-double Phibar (round 6; tests/studies/explicit_phibar_pass2.py).
//
// The extended evaluation order's pass 2 contracts the materialised K'_fu with the EXPLICIT Phibar = L^-T (C / 2 s2) L^-1, whose
// cond(K_uu)-sized entries cancel in Kbar = 2 K Phibar.  Against an 80-bit yardstick the error of that gradient is, in this order: the
// fp64 FORMATION of Phibar (cured by sgp_phibar_dd: two double-double products), then the ROUNDING of the formed matrix to one fp64 word
// -- a fixed perturbation eps |Phibar| that every data row sees alike, so its effect grows like N where the rounding of the N M^2
// accumulation grows like sqrt(N) -- and only then the fp64 accumulation.  The rounding is cured by keeping the second word:
//
//     Kbar = 2 K' (Phibar_hi + Phibar_lo),     |Phibar_lo| <= 2^-53 |Phibar_hi|
//
// and the product with the low word needs three significant digits only.  This file is that product and its contraction with dK:
//
//     dC[n, m]  = sum_m' K'[n, m'] Phibar_lo[m', m]                         fp16 operands, fp32 accumulation: v_mfma_f32_32x32x16_f16
//                 (K' lies in [0, 1]; every row of the symmetric Phibar_lo is scaled by a power of two into fp16's range and the scale
//                 taken out again per output column.  bf16 -- the first version -- left 1-2 % of the correction behind, which showed
//                 at the far end of the range: 1.1e-5 where the leading word alone is off by 5.5e-4; fp16's 11 bits leave 0.1 %)
//     S_0       = sum_nm dC k'           S_j = sum_nm dC k' (z~_mj - x~_nj)^2          (fp64, k' read back from the fp64 K'_fu)
//     g_sf2    += 2 sf2 S_0              g_ls[j] += 2 inv_ls_j sf2^2 S_j               (RBF: dk'/dr2 = -k'/2; the signs as kbar_contract_kernel)
//
// i.e. exactly what kbar_contract_kernel's epilogue would add had its C = K' Phibar carried the low word.  2 N M^2 flop on the fp16 / bf16
// matrix cores (2.5 PFLOP/s dense) against the same count on the fp64 ones for the leading word: a few per cent of a leapfrog.
// Layout: workgroup <-> (128 data rows, 128 inducing columns), four waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32, 64 fp32 accumulators);
// 32-deep k-chunks of the fp16 image of K' (made once per call) through LDS (registers hold the next chunk while this one is multiplied), the fp16 image of
// the symmetric Phibar_lo read ROW-wise as B^T (so both fragments are 16 contiguous bytes of LDS).  The eight column blocks of a row block
// share an XCD (ids 8 apart), whose L2 serves seven of the eight reads of every K' row block.
// Reference: this is the reverse pass of pm.gp.MarginalSparse's logp (models/bayesian_sgpr_hmc.py:66-78) at a precision Theano's fp64
// graph has by construction (it never forms Phibar); no counterpart in the reference's code.
#include "sgp_common.hpp"
#include "sgp_stream.hpp"
#include "sgp_dense.hpp"
#include <cstdint>
#include <cstdlib>

namespace sgp {

typedef __attribute__((ext_vector_type(8))) _Float16 lo_h8;
typedef __attribute__((ext_vector_type(16))) float lo_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int lo_u4;   // (a native vector: arrays of HIP's uint4 struct were kept in scratch)

constexpr int LO_T = 128;     // tile edge (rows and columns)
constexpr int LO_BK = 128;    // k-chunk: eight MFMA k-steps of 16 between two barriers (first version: 32 -- the next chunk's loads were issued 256
                              // cycles ahead of their use, a tenth of the memory latency: 13.7 ms, 6 % of the matrix peak)
constexpr int LO_LD = LO_BK + 8;  // fp16 elements per LDS row: 272 bytes -- 16-byte aligned fragments, consecutive rows 4 banks apart

__device__ __forceinline__ uint32_t lo_h16(float f) {   // round to nearest even (v_cvt_f16_f32)
  const _Float16 hv = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, hv);
}
__device__ __forceinline__ uint32_t lo_pack2(double a, double b) { return lo_h16((float)a) | (lo_h16((float)b) << 16); }

// fp16 image of the symmetric M x M low word, zero-padded to Mp x Mp: row r scaled by 2^sh(r) so that its largest entry lies in
// [2^13, 2^14) (fp16: normal down to 2^-14, 65504 at the top), unscale[r] = 2^-sh(r).  One workgroup per row.
__global__ __launch_bounds__(256) void lo_prep_kernel(const double* __restrict__ Plo, int M, int Mp, uint16_t* __restrict__ out,
                                                      double* __restrict__ unscale) {
  __shared__ double red[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  double mx = 0.0;
  if (r < M)
    for (int c = tid; c < M; c += 256) mx = fmax(mx, fabs(Plo[(size_t)r * M + c]));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  int ex = 0;
  if (mx > 0.0 && mx < 1e300) (void)frexp(mx, &ex);   // mx = f 2^ex, f in [0.5, 1)
  const int sh = (mx > 0.0 && mx < 1e300) ? 14 - ex : 0;
  if (tid == 0) unscale[r] = ldexp(1.0, -sh);
  for (int c = tid; c < Mp; c += 256)
    out[(size_t)r * Mp + c] = (r < M && c < M) ? (uint16_t)lo_h16((float)ldexp(Plo[(size_t)r * M + c], sh)) : (uint16_t)0;
}

// fp16 image of K'_fu, once per call (first version: converted on the fly by each of the Mp / 128 column-block workgroups that read a row
// block -- 65 GB of fp64 through the L2s for an 8 GB matrix: 13.7 ms at C5; with the image the product reads 2 GB eight times)
__global__ __launch_bounds__(256) void lo_kfu_f16_kernel(const double* __restrict__ Kfu, int64_t n8, uint4* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
    const double2* src = reinterpret_cast<const double2*>(Kfu) + 4 * e;
    const double2 a = src[0], b = src[1], c = src[2], d = src[3];
    uint4 o;
    o.x = lo_pack2(a.x, a.y); o.y = lo_pack2(b.x, b.y); o.z = lo_pack2(c.x, c.y); o.w = lo_pack2(d.x, d.y);
    out[e] = o;
  }
}

template <int DP>
__global__ __launch_bounds__(256, 2) void kphi_lo_kernel(const double* __restrict__ Kfu, const uint16_t* __restrict__ Kh, const uint16_t* __restrict__ Pl,
                                                      const double* __restrict__ unscale, const double* __restrict__ Xs,
                                                      const double* __restrict__ Zs, int Mp, int64_t nrb, int ncb,
                                                      double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) uint16_t ABs[2][LO_T][LO_LD];   // (one array: the epilogue reuses it as four 32 x 33 fp32 images)
  uint16_t (*As)[LO_LD] = ABs[0];
  uint16_t (*Bs)[LO_LD] = ABs[1];
  __shared__ double red[4][DP + 1];
  // (epilogue images inside the main-loop buffers: four wave-private 32 x 33 fp32 tiles, then the scaled rows / columns of this tile)
  // and the fp16 image of this tile's own K' block (the contraction needs k' to three digits only; read back from the fp64 matrix it was
  // sixteen dependent HBM round trips per wave -- 14.4 of the kernel's first 17 ms)
  static_assert(4 * 32 * 33 * sizeof(float) + 2 * LO_T * DP * sizeof(double) + LO_T * LO_LD * sizeof(uint16_t) <= 2 * LO_T * LO_LD * sizeof(uint16_t),
                "epilogue images fit");
  double (*Xl)[DP] = reinterpret_cast<double (*)[DP]>(reinterpret_cast<char*>(&ABs[0][0][0]) + 4 * 32 * 33 * sizeof(float));
  double (*Zl)[DP] = Xl + LO_T;
  _Float16 (*Kt)[LO_LD] = reinterpret_cast<_Float16 (*)[LO_LD]>(Zl + LO_T);
  // id -> (xcd, column block, row block): the ncb column blocks of a row block share id % 8, i.e. one XCD under round-robin dispatch
  const int xcd = blockIdx.x & 7;
  const int64_t jj = blockIdx.x >> 3;
  const int cb = (int)(jj % ncb);
  const int64_t rb = (jj / ncb) * 8 + xcd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* mypart = part + (size_t)blockIdx.x * (DP + 1);
  if (rb >= nrb) {  // (the grid is rounded up to whole groups of eight row blocks)
    if (tid <= DP) mypart[tid] = 0.0;
    return;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  const int64_t n0 = rb * LO_T;
  const int m0 = cb * LO_T;

  // staging roles: thread <-> (row, half of the chunk): LO_BK / 2 consecutive k, 16 bytes at a time
  constexpr int NV = LO_BK / 16;   // uint4 per thread and operand
  const int srow = tid >> 1, skh = (tid & 1) * (LO_BK / 2);
  const uint16_t* asrc = Kh + (size_t)(n0 + srow) * Mp + skh;
  const uint16_t* bsrc = Pl + (size_t)(m0 + srow) * Mp + skh;
  // (plain macros, no lambdas, no conditions around them: with either the compiler kept this 256-byte register ring in SCRATCH -- 16 GB of
  // private-memory traffic per call at C5, WRITE_SIZE 9.0 GB where the kernel writes 4.5 MB: profiles/r06_lo_pmc_counters.csv)
  lo_u4 areg[NV], breg[NV];
#define LO_FETCH(K0)                                                                        \
  _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                          \
    areg[q] = *reinterpret_cast<const lo_u4*>(asrc + (K0) + 8 * q);                         \
    breg[q] = *reinterpret_cast<const lo_u4*>(bsrc + (K0) + 8 * q);                         \
  }
#define LO_STASH()                                                                          \
  _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                          \
    *reinterpret_cast<lo_u4*>(&As[srow][skh + 8 * q]) = areg[q];                            \
    *reinterpret_cast<lo_u4*>(&Bs[srow][skh + 8 * q]) = breg[q];                            \
  }

  lo_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  LO_FETCH(0)
  for (int k0 = 0; k0 < Mp; k0 += LO_BK) {
    LO_STASH()
    __syncthreads();
    {  // the next chunk, in flight under this chunk's MFMAs (the last trip re-reads the last chunk: no condition around the ring)
      const int kn = k0 + LO_BK < Mp ? k0 + LO_BK : k0;
      LO_FETCH(kn)
    }
#pragma unroll
    for (int ks = 0; ks < LO_BK / 16; ++ks) {
      lo_h8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const lo_h8*>(&As[wr * 64 + i * 32 + r31][ks * 16 + 8 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const lo_h8*>(&Bs[wc * 64 + j * 32 + r31][ks * 16 + 8 * h]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue.  Accumulator element (reg e of a 32 x 32 tile): row = (e & 3) + 8 (e >> 2) + 4 h, column = r31 (C/D layout of the 32 x 32 forms).
  // The four tiles of a wave go through a wave-private 32 x 33 LDS image one after the other, so that the contraction is a ROLLED loop
  // over rows (straight from the registers it is 64 unrolled elements whose loads the compiler hoists: 512 VGPRs, 150 spilled).
#undef LO_FETCH
#undef LO_STASH
  float (*Cl)[33] = reinterpret_cast<float (*)[33]>(reinterpret_cast<float*>(&ABs[0][0][0]) + wave * 32 * 33);
  // the scaled inputs of this tile's rows and columns (the main loop's last barrier is behind every wave)
  {
    lo_u4 kt[LO_T / 16];   // this thread's share of the 128 x 128 fp16 block: row srow, 64 columns (all loads in flight together)
    const uint16_t* ksrc = Kh + (size_t)(n0 + srow) * Mp + m0 + (tid & 1) * (LO_T / 2);
#pragma unroll
    for (int q = 0; q < LO_T / 16; ++q) kt[q] = *reinterpret_cast<const lo_u4*>(ksrc + 8 * q);
#pragma unroll
    for (int q = 0; q < LO_T / 16; ++q) *reinterpret_cast<lo_u4*>(&Kt[srow][(tid & 1) * (LO_T / 2) + 8 * q]) = kt[q];
#pragma unroll
    for (int e0 = 0; e0 < LO_T * DP; e0 += 256) {   // (LO_T DP is a multiple of 256 for DP = 2, 4, 8)
      const int e = e0 + tid;
      Xl[e / DP][e % DP] = Xs[(size_t)n0 * DP + e];
      Zl[e / DP][e % DP] = Zs[(size_t)m0 * DP + e];
    }
  }
  __syncthreads();
  double S[DP + 1];
#pragma unroll
  for (int q = 0; q <= DP; ++q) S[q] = 0.0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wc * 64 + j * 32 + r31;
    const double us = unscale[m0 + col];   // this column's row of the (symmetric) low word was scaled by 1 / us
    double z[DP];
#pragma unroll
    for (int q = 0; q < DP; ++q) z[q] = Zl[col][q];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) Cl[(e & 3) + 8 * (e >> 2) + 4 * h][r31] = acc[i][j][e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int rbase = wr * 64 + i * 32 + 16 * h;
#pragma unroll 4
      for (int rr = 0; rr < 16; ++rr) {
        const double kp = (double)Kt[rbase + rr][col];   // (zero in the padding: padded rows / columns add nothing)
        const double w = (double)Cl[16 * h + rr][r31] * (kp * us);
        S[DP] += w;
#pragma unroll
        for (int q = 0; q < DP; ++q) {
          const double df = z[q] - Xl[rbase + rr][q];
          S[q] = fma(w * df, df, S[q]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int q = 0; q <= DP; ++q) {
    const double v = wave_sum(S[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid <= DP) mypart[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// g_ls[j] += 2 inv_ls_j sf2^2 sum_parts S_j ; g_sf2 += 2 sf2 sum_parts S_DP -- one block, a fixed thread <-> partial mapping and a fixed tree
// The same product on 256 x 256 tiles (eight waves of 64 x 128: 2 x 4 MFMA tiles, 128 fp32 accumulators): the operands are re-read
// 2 N M K (1/256 + 1/256) = 16 GB instead of 32 -- the 128 x 128 kernel above is bound by that traffic once its register ring really lives in registers.  One workgroup per CU (139 KB of LDS).  The epilogue runs in two phases of two tile columns per wave,
// because the fp16 image of the tile's own K' block (128 KB) does not fit beside the other images: phase p holds columns 64 p ... 64 p + 63 of
// either 128-column half.  Mp must be a multiple of 256 (the caller falls back to the 128 x 128 kernel otherwise).
constexpr int LO2_T = 256;
template <int DP>
__global__ __launch_bounds__(512) void kphi_lo256_kernel(const uint16_t* __restrict__ Kh, const uint16_t* __restrict__ Pl,
                                                         const double* __restrict__ unscale, const double* __restrict__ Xs,
                                                         const double* __restrict__ Zs, int Mp, int64_t nrb, int ncb,
                                                         double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) uint16_t ABs[2][LO2_T][LO_LD];
  __shared__ double red[8][DP + 1];
  uint16_t (*As)[LO_LD] = ABs[0];
  uint16_t (*Bs)[LO_LD] = ABs[1];
  // epilogue images inside the main-loop buffers: eight wave-private 32 x 33 fp32 tiles | Xl | Zl | Kt (256 rows x (2 x 64) columns of fp16)
  constexpr size_t CL_BYTES = 8 * 32 * 33 * sizeof(float), XZ_BYTES = (size_t)LO2_T * DP * sizeof(double);
  static_assert(CL_BYTES + 2 * XZ_BYTES + (size_t)LO2_T * LO_LD * sizeof(uint16_t) <= 2 * (size_t)LO2_T * LO_LD * sizeof(uint16_t), "epilogue images fit");
  char* lbase = reinterpret_cast<char*>(&ABs[0][0][0]);
  double (*Xl)[DP] = reinterpret_cast<double (*)[DP]>(lbase + CL_BYTES);
  double (*Zl)[DP] = reinterpret_cast<double (*)[DP]>(lbase + CL_BYTES + XZ_BYTES);
  _Float16 (*Kt)[LO_LD] = reinterpret_cast<_Float16 (*)[LO_LD]>(lbase + CL_BYTES + 2 * XZ_BYTES);

  const int xcd = blockIdx.x & 7;
  const int64_t jj = blockIdx.x >> 3;
  const int cb = (int)(jj % ncb);
  const int64_t rb = (jj / ncb) * 8 + xcd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* mypart = part + (size_t)blockIdx.x * (DP + 1);
  if (rb >= nrb) {
    if (tid <= DP) mypart[tid] = 0.0;
    return;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  const int64_t n0 = rb * LO2_T;
  const int m0 = cb * LO2_T;

  constexpr int NV = LO_BK / 16;
  const int srow = tid >> 1, skh = (tid & 1) * (LO_BK / 2);
  const uint16_t* asrc = Kh + (size_t)(n0 + srow) * Mp + skh;
  const uint16_t* bsrc = Pl + (size_t)(m0 + srow) * Mp + skh;
  lo_u4 areg[NV], breg[NV];
#define LO_FETCH(K0)                                                                        \
  _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                          \
    areg[q] = *reinterpret_cast<const lo_u4*>(asrc + (K0) + 8 * q);                         \
    breg[q] = *reinterpret_cast<const lo_u4*>(bsrc + (K0) + 8 * q);                         \
  }
#define LO_STASH()                                                                          \
  _Pragma("unroll") for (int q = 0; q < NV; ++q) {                                          \
    *reinterpret_cast<lo_u4*>(&As[srow][skh + 8 * q]) = areg[q];                            \
    *reinterpret_cast<lo_u4*>(&Bs[srow][skh + 8 * q]) = breg[q];                            \
  }

  lo_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  LO_FETCH(0)
  for (int k0 = 0; k0 < Mp; k0 += LO_BK) {
    LO_STASH()
    __syncthreads();
    {
      const int kn = k0 + LO_BK < Mp ? k0 + LO_BK : k0;
      LO_FETCH(kn)
    }
#pragma unroll
    for (int ks = 0; ks < LO_BK / 16; ++ks) {
      lo_h8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const lo_h8*>(&As[wr * 64 + i * 32 + r31][ks * 16 + 8 * h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const lo_h8*>(&Bs[wc * 128 + j * 32 + r31][ks * 16 + 8 * h]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#undef LO_FETCH
#undef LO_STASH
  float (*Cl)[33] = reinterpret_cast<float (*)[33]>(reinterpret_cast<float*>(lbase) + wave * 32 * 33);
  double S[DP + 1];
#pragma unroll
  for (int q = 0; q <= DP; ++q) S[q] = 0.0;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    {  // this phase's fp16 block of K': thread <-> (row, 128-column half): columns m0 + 128 half + 64 ph ... + 63
      lo_u4 kt[8];
      const uint16_t* ksrc = Kh + (size_t)(n0 + srow) * Mp + m0 + (tid & 1) * 128 + 64 * ph;
#pragma unroll
      for (int q = 0; q < 8; ++q) kt[q] = *reinterpret_cast<const lo_u4*>(ksrc + 8 * q);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<lo_u4*>(&Kt[srow][(tid & 1) * 64 + 8 * q]) = kt[q];
      if (ph == 0) {
#pragma unroll
        for (int e0 = 0; e0 < LO2_T * DP; e0 += 512) {
          const int e = e0 + tid;
          Xl[e / DP][e % DP] = Xs[(size_t)n0 * DP + e];
          Zl[e / DP][e % DP] = Zs[(size_t)m0 * DP + e];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) {
      const int j = 2 * ph + jl;
      const int col = wc * 128 + j * 32 + r31;          // column inside the 256-wide tile
      const int kcol = wc * 64 + jl * 32 + r31;          // ... and inside this phase's K' image
      const double us = unscale[m0 + col];
      double z[DP];
#pragma unroll
      for (int q = 0; q < DP; ++q) z[q] = Zl[col][q];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Cl[(e & 3) + 8 * (e >> 2) + 4 * h][r31] = acc[i][j][e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rbase = wr * 64 + i * 32 + 16 * h;
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
          const double kp = (double)Kt[rbase + rr][kcol];
          const double w = (double)Cl[16 * h + rr][r31] * (kp * us);
          S[DP] += w;
#pragma unroll
          for (int q = 0; q < DP; ++q) {
            const double df = z[q] - Xl[rbase + rr][q];
            S[q] = fma(w * df, df, S[q]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();   // (the next phase overwrites the K' image)
  }
#pragma unroll
  for (int q = 0; q <= DP; ++q) {
    const double v = wave_sum(S[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid <= DP) mypart[tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
}

// delta (optional, d + 1 doubles): the correction itself [d lengthscales | sf2] -- what the caller holds against the gradient to decide whether
// the explicit pass 2 can be trusted at this theta (core.py: extended_lo_max_correction)
// (one workgroup per slot: 256 threads stride over the partials -- 62 500 of them at C5 --, wave sums, four wave totals added in order)
__global__ __launch_bounds__(256) void lo_reduce_kernel(const double* __restrict__ part, int nparts, int DP, KernArgs ka,
                                                        double* __restrict__ g_ls, double* __restrict__ g_sf2, double* __restrict__ delta) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int q = blockIdx.x;
    const int slot = q == ka.d ? DP : q;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) s += part[(size_t)p * (DP + 1) + slot];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    if (threadIdx.x == 0) {
      const double c = q == ka.d ? 2.0 * ka.sf2 * s : 2.0 * ka.inv_ls[q] * ka.sf2 * ka.sf2 * s;
      if (q == ka.d) *g_sf2 += c;
      else g_ls[q] += c;
      if (delta) delta[q] = c;
    }
  }
}

struct LoWs {
  double *Xs, *ys, *Zs, *yypart, *part, *unscale;
  uint16_t *Pl, *Kh;
  size_t bytes;
  int grid;
};
static LoWs carve_lo(void* ws, const StreamPlan& p) {
  Carver c(ws);
  LoWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.yypart = c.take<double>(256);
  const int64_t nrb = p.Npad / LO_T;
  const int ncb = p.Mp / LO_T;
  w.grid = (int)(((nrb + 7) / 8) * 8 * ncb);
  w.part = c.take<double>((size_t)(w.grid > 0 ? w.grid : 1) * (p.DP + 1));
  w.Pl = c.take<uint16_t>((size_t)p.Mp * p.Mp);
  w.unscale = c.take<double>((size_t)p.Mp);
  w.Kh = c.take<uint16_t>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.Mp);
  w.bytes = c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

extern "C" size_t sgp_suffstats_bwd_lo_workspace_bytes(int64_t N, int M, int d) {
  if (N < 0 || M <= 0 || d <= 0 || d > 8 || M > SGP_MAX_INDUCING) return 0;
  return carve_lo(nullptr, make_stream_plan(N, M, d)).bytes;
}

// Adds the low word's contribution to g_ls (d doubles) and g_sf2 IN PLACE, behind sgp_suffstats_bwd on the same stream with the same
// inputs and Phibar = the leading word.  Kfu_in: the fp64 K'_fu of this shard (sgp_kfu_len doubles) as pass 1 left it.  RBF, d <= 8
// (SGP_ERR_ARG / SGP_ERR_DIM otherwise: the caller then keeps the leading word's gradient); g_Z is not corrected.  delta (optional,
// d + 1 doubles): receives the correction itself.
extern "C" int sgp_suffstats_bwd_lo(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls,
                                    double sf2, const double* Phibar_lo, const double* Kfu_in, int64_t N, int M, int d, int kernel_id,
                                    double* g_ls, double* g_sf2, double* delta, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!Z || !inv_ls || !Phibar_lo || !Kfu_in || !g_ls || !g_sf2 || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id != SGP_KERNEL_RBF) return SGP_ERR_ARG;
  if (d > 8 || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (N == 0) {
    if (delta) fill_zero(delta, (size_t)d + 1, (hipStream_t)stream);
    return check_launch();
  }
  StreamPlan p = make_stream_plan(N, M, d);
  LoWs w = carve_lo(ws, p);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st);
  lo_prep_kernel<<<p.Mp, 256, 0, st>>>(Phibar_lo, M, p.Mp, w.Pl, w.unscale);
  const int64_t nrb = p.Npad / LO_T;
  const int ncb = p.Mp / LO_T;
  lo_kfu_f16_kernel<<<4096, 256, 0, st>>>(Kfu_in, (int64_t)p.Npad * p.Mp / 8, reinterpret_cast<uint4*>(w.Kh));
  static const int tile128 = getenv("SGP_LO_TILE128") ? atoi(getenv("SGP_LO_TILE128")) : 0;   // A/B: the 128 x 128 kernel also where 256 divides Mp
  int nparts = w.grid;
  if (p.Mp % LO2_T == 0 && !tile128) {
    const int64_t nrb2 = p.Npad / LO2_T;
    const int ncb2 = p.Mp / LO2_T;
    nparts = (int)(((nrb2 + 7) / 8) * 8 * ncb2);
    switch (p.DP) {
      case 2: kphi_lo256_kernel<2><<<nparts, 512, 0, st>>>(w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb2, ncb2, w.part); break;
      case 4: kphi_lo256_kernel<4><<<nparts, 512, 0, st>>>(w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb2, ncb2, w.part); break;
      default: kphi_lo256_kernel<8><<<nparts, 512, 0, st>>>(w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb2, ncb2, w.part); break;
    }
  } else {
    switch (p.DP) {
      case 2: kphi_lo_kernel<2><<<w.grid, 256, 0, st>>>(Kfu_in, w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
      case 4: kphi_lo_kernel<4><<<w.grid, 256, 0, st>>>(Kfu_in, w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
      default: kphi_lo_kernel<8><<<w.grid, 256, 0, st>>>(Kfu_in, w.Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
    }
  }
  lo_reduce_kernel<<<d + 1, 256, 0, st>>>(w.part, nparts, p.DP, ka, g_ls, g_sf2, delta);
  return check_launch();
}
