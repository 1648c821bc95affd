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
#include <atomic>
#include <cstdlib>
#include <type_traits>

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
// [2^(target_exp - 1), 2^target_exp) (fp16: normal down to 2^-14, 65504 at the top; 14 for the kernels that keep dC in fp32, 3 for the one
// that rounds dC <= 8 Mp to fp16), unscale[r] = 2^-sh(r).  One workgroup per row.
__global__ __launch_bounds__(256) void lo_prep_kernel(const double* __restrict__ Plo, int M, int Mp, int target_exp, uint16_t* __restrict__ out,
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
  const int sh = (mx > 0.0 && mx < 1e300) ? target_exp - ex : 0;   // largest entry in [2^(target_exp - 1), 2^target_exp)
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Third version (round 6.  The second -- this tile shape with the first version's register ring and fp64 contraction, 5.0 ms at C5 -- is in
// the history (8505ef4); its split A/B, profiles/r06_lo_split_ab.txt: main loop 3.0 ms, contraction 2.0, one workgroup per CU, so neither
// hides behind the other).
//
// Main loop: the same 256 x 256 tile and eight waves of 64 x 128, but the operands travel global -> LDS by LDS-DMA in full 128-byte lines
// (64-deep stages, 64 KB each, two of them): no register ring, no ds_write pass (128 KB per chunk at ~79 B / clk had the matrix pipe idle a
// quarter of the time), stage t + 1 in flight under the 32 MFMAs per wave of stage t, one raw barrier per stage.  An LDS-DMA instruction
// writes 1 KB contiguously (lane l -> base + 16 l), so a piece is 8 rows x 128 B and the image is unpadded; the bank spread comes from an
// XOR on the SOURCE side: 16-byte slot c of row R holds k-segment c ^ ((R >> 1) & 7), and the fragment reads apply the same XOR -- a
// ds_read_b128 lane group (16 rows, distinct mod 16) then covers all 64 banks.
//
// Contraction: with w = dC k' (this tile's product times the kernel value, per element)
//     S_j = sum_m us_m [ z_mj^2 A0_m  -  2 z_mj A1_mj  +  A2_mj ],   A0_m = sum_n w_nm,  A1_mj = sum_n w_nm x_nj,  A2_mj = sum_n w_nm x_nj^2
// and A1, A2 are one more matrix product, W^T [x | x^2], whose A operand the accumulators ALREADY hold in the right lanes: lane (column m,
// half h) of a 32 x 32 accumulator tile owns rows {4 h + (e & 3) + 8 (e >> 2)}, and the k order of an MFMA is free as long as both operands
// use the same one -- so registers 8 s .. 8 s + 7, rounded to fp16 and multiplied by k', ARE the A fragment of k-step s, and the B fragment
// [x_hi | x_lo | (x^2)_hi | (x^2)_lo] x 8 dimensions = 32 columns is prepared once per call in that row order (lo_bx_kernel).  fp16 pairs
// (hi + lo: 22 bits) because the expansion cancels where |z - x| << |z|; inputs are centred on the mean inducing point first.  A0 by
// v_dot2c_f32_f16 from the same rounded w (the three terms must see the same w).  Per 32 x 32 tile and lane: 16 two-byte loads of k'
// (fp16 image, L2), 8 v_cvt_pk_f16_f32, 8 v_pk_mul_f16, 8 v_dot2c, 2 MFMAs -- against 16 x 27 fp64 operations before.
// Range: rows of the low word are scaled into [4, 8) (lo_prep_kernel, target_exp 3), so |dC| <= 8 Mp <= 2^15 fits fp16; |x~ - c| is
// clamped to 255 (beyond it k' is zero against every inducing point with |z~ - c| <= 128; an inducing point further out raises `flag`
// and the call reports NaN corrections -- the caller then keeps the whitened order, core.py).
constexpr int L3_T = 256;
constexpr int L3_BK = 64;
constexpr int L3_OPB = L3_T * L3_BK * 2;   // bytes per operand and stage
constexpr int L3_STAGE = 2 * L3_OPB;       // 65 536
constexpr int L3_TTS = 28;                 // floats per column of the factor table: [-2 z us (8) | us (8) | z^2 us (8) | 0 (4)]; 112-byte rows
                                           // keep 16-byte reads of 16 consecutive rows on distinct banks (28 r mod 64 = 4 (7 r mod 16))
constexpr int L3_TT_BYTES = L3_T * L3_TTS * 4;   // 28 672: behind the two stages (the table is loaded once, at the start)
constexpr int L3_LDS_BYTES = 2 * L3_STAGE + L3_TT_BYTES;   // 159 744 of 163 840
typedef _Float16 lo_h2 __attribute__((ext_vector_type(2)));
typedef float lo_f2 __attribute__((ext_vector_type(2)));
typedef float lo_f4 __attribute__((ext_vector_type(4)));

// c_q = mean over the inducing points of z~_q (fixed order; q < 32); clears the range flag
__global__ __launch_bounds__(256) void lo3_centre_kernel(const double* __restrict__ Zs, int M, int DP, double* __restrict__ centre, int* __restrict__ flag) {
  __shared__ double red[8][32];
  const int q = threadIdx.x & 31, part = threadIdx.x >> 5;
  double s = 0.0;
  if (q < DP)
    for (int m = part; m < M; m += 8) s += Zs[(size_t)m * DP + q];
  red[part][q] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    double t = 0.0;
    for (int p2 = 0; p2 < 8; ++p2) t += red[p2][threadIdx.x];
    centre[threadIdx.x] = t / (double)M;
  }
  if (threadIdx.x == 0) *flag = 0;
}

// factor tables, one per group of eight dimensions (TT + g Mp 28 floats: the kernel holds one group's table at a time); one thread per padded column
// (behind lo_prep_kernel: needs unscale)
__global__ __launch_bounds__(256) void lo3_tt_kernel(const double* __restrict__ Zs, const double* __restrict__ unscale, const double* __restrict__ centre,
                                                     int M, int Mp, int DP, int NG, float* __restrict__ TT, int* __restrict__ flag) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= Mp) return;
  const double us = unscale[m];
  bool far = false;
  for (int g = 0; g < NG; ++g) {
    float* row = TT + ((size_t)g * Mp + m) * L3_TTS;
    for (int q = 0; q < 8; ++q) {
      const int qq = 8 * g + q;
      double zc = (m < M && qq < DP) ? Zs[(size_t)m * DP + qq] - centre[qq] : 0.0;
      if (!(fabs(zc) <= 128.0)) { far = true; zc = 0.0; }   // (also a NaN coordinate)
      row[q] = (float)(-2.0 * zc * us);
      row[8 + q] = (float)us;
      row[16 + q] = (float)(zc * zc * us);
    }
    for (int q = 24; q < L3_TTS; ++q) row[q] = 0.0f;
  }
  if (far) atomicOr(flag, 1);
}

// B operand of the contraction product.  16-byte vector v = (((rt 2 + s) 2 + h) 32 + c): data rows n = 32 rt + 16 s + 4 h + (t & 3) + 8 (t >> 2),
// t = 0 .. 7 (the rows lane half h of an accumulator tile holds in registers 8 s .. 8 s + 7); column c = 8 kind + q.
__global__ __launch_bounds__(256) void lo_bx_kernel(const double* __restrict__ Xs, const double* __restrict__ centre, int64_t ngroups, int DP, int NG,
                                                    lo_u4* __restrict__ out) {
  // thread <-> (row tile rt, k-step s, lane half h, dimension q < 8 NG): its eight rows once, the four kinds of column from them
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = 8 * NG;
  if (u >= ngroups * nq) return;
  const int q = (int)(u % nq);
  const int64_t grp = u / nq;   // (rt 2 + s) 2 + h
  const int hh = (int)(grp & 1), s = (int)((grp >> 1) & 1);
  const int64_t rt = grp >> 2;
  const double cq = q < DP ? centre[q] : 0.0;
  uint32_t w[4][4];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int64_t n = rt * 32 + 16 * s + 4 * hh + (t & 3) + 8 * (t >> 2);
    double xc = q < DP ? Xs[(size_t)n * DP + q] - cq : 0.0;
    xc = xc > 255.0 ? 255.0 : (xc < -255.0 ? -255.0 : xc);
    if (!(xc == xc)) xc = 0.0;   // (a NaN coordinate: its row of K' is NaN already and the leading word's gradient with it)
    const double x2 = xc * xc;
    const _Float16 xh = (_Float16)(float)xc, qh = (_Float16)(float)x2;
    const _Float16 xl = (_Float16)(float)(xc - (double)(float)xh), ql = (_Float16)(float)(x2 - (double)(float)qh);
    const _Float16 v4[4] = {xh, xl, qh, ql};
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) {
      const uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, v4[kind]);
      if (t & 1) w[kind][t >> 1] |= bits << 16;
      else w[kind][t >> 1] = bits;
    }
  }
  // vector ((((rt NG + g) 2 + s) 2 + h) 32 + 8 kind + (q % 8)), g = q / 8
  lo_u4* dst = out + ((((rt * NG + (q >> 3)) * 2 + s) * 2 + hh) * 32 + (q & 7));
#pragma unroll
  for (int kind = 0; kind < 4; ++kind) {
    lo_u4 o;
    o[0] = w[kind][0]; o[1] = w[kind][1]; o[2] = w[kind][2]; o[3] = w[kind][3];
    dst[8 * kind] = o;
  }
}

__device__ __forceinline__ void lo3_glds(const char* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// VAR (A/B instruments in the same binary, SGP_LO_VARIANT): 0 = the product; 1 = without the contraction; 2 = two stages only.
// Measured and dropped (profiles/r06_lo_v3_variants_*.txt, whole call at C5):
// touching the A operand's lines one / two stages ahead (HBM -> L2) 2.82 / 2.83 against 2.71 ms; all eight DMAs up front behind the barrier
// 2.69 against 2.71; nobody waiting for the DMAs at all (wrong results, timing only) 2.54 against 2.62 and every workgroup reading row
// block 0 2.59 against 2.62; two fragment sets with the loop rotated by one k-step (a stage's last MFMAs issued behind the next stage's
// barrier and first reads) 2.65 against 2.64.  What the loop does wait for, by removal (timing-only builds, profiles/r06_lo_v3_loop_components.txt,
// whole call): without the loop's DMAs 2.08 against 2.55 ms, without its fragment reads 2.50, with neither 1.94 -- the ISSUE of the LDS-DMAs is
// what the waves stand in (64 KB per CU and stage through the L2 -> LDS path: 33 GB/s per CU, 8.5 TB/s over the chip; the matrix pipe is 0.46
// busy at 2.03 GHz, SQ counters of the bench run), not their latency and not the LDS.
template <int VAR>
__global__ __launch_bounds__(512) void kphi_lo3_kernel(const uint16_t* __restrict__ Kh, const uint16_t* __restrict__ Pl,
                                                       const lo_u4* __restrict__ Bx, const float* __restrict__ TT, int Mp, int Mp2, int NG, int64_t nrb,
                                                       int ncb, double* __restrict__ part) {
  // (dynamic: with a static array hipcc knows that the LDS-DMA and the fragment reads touch one object and drains the DMA -- vmcnt(0) --
  // before the first read behind it, i.e. before the MFMAs it was to hide under)
  extern __shared__ __attribute__((aligned(1024))) uint8_t l3[];
  __shared__ double red[8][33];
  const int xcd = blockIdx.x & 7;
  const int64_t jj = blockIdx.x >> 3;
  const int cb = (int)(jj % ncb);
  const int64_t rb = (jj / ncb) * 8 + xcd;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nslot = 8 * NG + 1;   // [8 NG dimensions | S_0]
  double* mypart = part + (size_t)blockIdx.x * nslot;
  if (rb >= nrb) {
    if (tid < nslot) mypart[tid] = 0.0;
    return;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  const int64_t n0 = rb * L3_T;
  const int m0 = cb * L3_T;

  // LDS-DMA roles: wave w moves pieces 4 w .. 4 w + 3 (rows 32 w .. 32 w + 31) of either operand; lane l of piece p: row R = 32 w + 8 p + l / 8,
  // slot l % 8 <- k-segment (l % 8) ^ ((R >> 1) & 7)
  const char* abase = reinterpret_cast<const char*>(Kh + (size_t)n0 * Mp);
  // (Mp2 = Mp rounded up to 256: the image of the low word and the factor table are padded with zeros to it; the image of K' is not -- its
  // rows are Mp long, and for the k beyond them (two stages, where Mp is an odd multiple of 128) the A operand re-reads the two stages
  // before, against zeros of the B operand; likewise the K' block of output columns beyond Mp, whose dC is zero)
  const char* bbase = reinterpret_cast<const char*>(Pl + (size_t)m0 * Mp2);
  unsigned goff[4], goffb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int R = 32 * wave + 8 * p + (lane >> 3);
    const int kseg = (lane & 7) ^ ((R >> 1) & 7);
    goff[p] = (unsigned)(R * Mp + kseg * 8) * 2u;
    goffb[p] = (unsigned)(R * Mp2 + kseg * 8) * 2u;
  }
  const int ta_last = Mp / L3_BK;   // stages of the A operand that exist
  auto issue1 = [&](int t, int buf, int q) {   // DMA q of this wave's eight of stage t: A pieces 0 .. 3, B pieces 4 .. 7
    const char* gb = q < 4 ? abase + (size_t)(t < ta_last ? t : t - 2) * (L3_BK * 2) + goff[q & 3] : bbase + (size_t)t * (L3_BK * 2) + goffb[q & 3];
    lo3_glds(gb, l3 + buf * L3_STAGE + wave * 4096 + (q < 4 ? 0 : L3_OPB) + (q & 3) * 1024);
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) issue1(t, buf, q);
  };
  auto issue_tt = [&](int gq) {   // this column block's 28 KB of dimension group gq's factor table, a linear copy into its own region
    const char* src = reinterpret_cast<const char*>(TT + ((size_t)gq * Mp2 + m0) * L3_TTS) + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int P = wave + 8 * k;
      if (P < L3_TT_BYTES / 1024) lo3_glds(src + P * 1024, l3 + 2 * L3_STAGE + P * 1024);
    }
  };
  // the fp16 block of K' this wave's accumulators are multiplied with, in two halves of 64 columns: 64 rows x 128 bytes = 8 pieces of
  // 8 rows, wave-private (its own 8 KB of a stage buffer: ordered by the wave's own vmcnt, no barrier)
  const char* kbase = reinterpret_cast<const char*>(Kh + (size_t)(n0 + wr * 64) * Mp + (m0 + wc * 128 < Mp ? m0 + wc * 128 : m0 + wc * 128 - 128));
  auto issue_kp = [&](int half, int buf) {
#pragma unroll
    for (int p2 = 0; p2 < 8; ++p2)
      lo3_glds(kbase + (size_t)(8 * p2 + (lane >> 3)) * Mp * 2 + half * 128 + (lane & 7) * 16, l3 + buf * L3_STAGE + wave * 8192 + p2 * 1024);
  };
  // fragment reads: row R of an operand image lives at R * 128 + ((kseg ^ ((R >> 1) & 7)) * 16, kseg = 2 ks + h
  const int tsw = ((r31 >> 1) & 7) ^ h;
  const int arow0 = (wr * 64 + r31) * 128;
  const int brow0 = L3_OPB + (wc * 128 + r31) * 128;

  lo_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int T = VAR == 2 ? 2 : Mp2 / L3_BK;
  auto step = [&](int t, auto buf_tag) {
    constexpr int B = decltype(buf_tag)::value;
    // my pieces of stage t have landed; my reads of the buffer about to be restaged have returned
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
    __builtin_amdgcn_s_barrier();                            // ... everybody's have, and stage t - 1 has been read by all
    asm volatile("" ::: "memory");
    const bool more = t + 1 < T;
    if (!more) issue_kp(0, B ^ 1);   // the last stage: the other buffer is free -- the first half of the K' block lands under these MFMAs
    const uint8_t* sb = l3 + B * L3_STAGE;
#pragma unroll
    for (int ks = 0; ks < L3_BK / 16; ++ks) {
      const int ko = ((2 * ks) ^ tsw) << 4;
      lo_h8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const lo_h8*>(sb + arow0 + i * 4096 + ko);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const lo_h8*>(sb + brow0 + j * 4096 + ko);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      // two DMAs of the next stage in the shadow of this k-step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        issue1(t + 1, B ^ 1, 2 * ks);
        issue1(t + 1, B ^ 1, 2 * ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the contraction's B fragments: loaded here, ahead of every LDS-DMA (behind one, hipcc waits vmcnt(0) at their first use and would
  // drain the second half of the K' block before the first half's arithmetic)
  lo_h8 bx[2][2];
  {
    const lo_h8* bxg = reinterpret_cast<const lo_h8*>(Bx);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) bx[i][s] = bxg[(((((size_t)rb * 8 + 2 * wr + i) * NG + 0) * 2 + s) * 2 + h) * 32 + r31];
  }
  issue_tt(0);
  issue(0, 0);
  for (int t = 0; t < T; t += 2) {   // (T is a multiple of 4: Mp2 of 256)
    step(t, std::integral_constant<int, 0>());
    step(t + 1, std::integral_constant<int, 1>());
  }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
  __builtin_amdgcn_s_barrier();   // (every wave is through with the last stage's buffer -- buffer 1, T is even -- and the factor table is everybody's)
  asm volatile("" ::: "memory");
  const float* ttl = reinterpret_cast<const float*>(l3 + 2 * L3_STAGE);
  if (VAR == 1) {   // (keep the accumulators alive)
    float sacc = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
    const double v = wave_sum((double)sacc);
    if (lane == 0) red[wave][0] = v;
    __syncthreads();
    if (tid < nslot) mypart[tid] = tid == 0 ? red[0][0] + red[1][0] + red[2][0] + red[3][0] + red[4][0] + red[5][0] + red[6][0] + red[7][0] : 0.0;
    return;
  }
  // ---- contraction
  if (VAR != 1) issue_kp(1, 1);   // the second half, under the first half's arithmetic
  // lane (column r31 of a 32-column block, half h) reads rows 4 h + (e & 3) + 8 (e >> 2) (+ 32 i) of its column: two bytes each
  const uint8_t* kpl = l3 + wave * 8192 + (4 * h) * 128 + r31 * 2;
  uint32_t raw[16];
#define L3_LOADKP(I, J)                                                                                        \
  _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                               \
    raw[e] = *reinterpret_cast<const uint16_t*>(kpl + ((J) >> 1) * L3_STAGE + ((I) * 32 + (e & 3) + 8 * (e >> 2)) * 128 + ((J) & 1) * 64);
  const int oc = (r31 >> 4) ? 8 + (r31 & 7) : (r31 & 7);
  const lo_h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  // one pass over the accumulators per group of eight dimensions (d <= 8: one): the group's B fragments and factor table, w recomputed
  for (int gq = 0; gq < NG; ++gq) {
    if (gq > 0) {
      __syncthreads();   // every wave is through with the previous group's factor table
      {
        const lo_h8* bxg = reinterpret_cast<const lo_h8*>(Bx);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s) bx[i][s] = bxg[(((((size_t)rb * 8 + 2 * wr + i) * NG + gq) * 2 + s) * 2 + h) * 32 + r31];
      }
      issue_tt(gq);
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    L3_LOADKP(0, 0)
    float P = 0.0f, S0 = 0.0f, Sz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) Sz[q] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo_f32x16 out2;
#pragma unroll
      for (int e = 0; e < 16; ++e) out2[e] = 0.0f;
      float a0 = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint32_t pk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pk[u] = raw[2 * u] | (raw[2 * u + 1] << 16);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) { L3_LOADKP(1, j) }
        else if (j < 3) {
          if (j == 1) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // the second half of the K' block (this wave's own DMAs)
          L3_LOADKP(0, j + 1)
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const lo_f2 v2 = {acc[i][j][2 * u], acc[i][j][2 * u + 1]};
          const lo_h2 w2 = __builtin_convertvector(v2, lo_h2) * __builtin_bit_cast(lo_h2, pk[u]);
          a0 = __builtin_amdgcn_fdot2(w2, ones, a0, false);
          wv[u] = __builtin_bit_cast(uint32_t, w2);
        }
        lo_u4 f0, f1;
        f0[0] = wv[0]; f0[1] = wv[1]; f0[2] = wv[2]; f0[3] = wv[3];
        f1[0] = wv[4]; f1[1] = wv[5]; f1[2] = wv[6]; f1[3] = wv[7];
        out2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lo_h8, f0), bx[i][0], out2, 0, 0, 0);
        out2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lo_h8, f1), bx[i][1], out2, 0, 0, 0);
      }
      const int mb = wc * 128 + j * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) P = fmaf(out2[e], ttl[(mb + (e & 3) + 8 * (e >> 2) + 4 * h) * L3_TTS + oc], P);
      const float* tm = ttl + (mb + r31) * L3_TTS;
      const lo_f4 z0 = *reinterpret_cast<const lo_f4*>(tm + 16), z1 = *reinterpret_cast<const lo_f4*>(tm + 20);
      S0 = fmaf(a0, tm[8], S0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Sz[q] = fmaf(a0, z0[q], Sz[q]);
        Sz[4 + q] = fmaf(a0, z1[q], Sz[4 + q]);
      }
    }
    // ten butterflies side by side (one after the other, as nine wave_sum() calls on doubles, they were 54 dependent ds_bpermute round trips:
    // most of this contraction's time); P only over the lanes that share lane % 8
    {
      float v[9];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = Sz[q];
      v[8] = S0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] += __shfl_xor(v[q], o, 64);
        if (o >= 8) P += __shfl_xor(P, o, 64);
      }
      // lane q < 8: S_q = the A0 part (every lane holds it) + the product part (lanes = q mod 8 hold it); lane 8: S_0 (the same in every group)
      float mine = v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) mine = lane == q ? v[q] + P : mine;
      if (lane < 8) red[wave][8 * gq + lane] = (double)mine;
      if (lane == 8 && gq == 0) red[wave][8 * NG] = (double)mine;
    }
  }
#undef L3_LOADKP
  __syncthreads();
  if (tid < nslot) mypart[tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
}

// delta (optional, d + 1 doubles): the correction itself [d lengthscales | sf2] -- what the caller holds against the gradient to decide whether
// the explicit pass 2 can be trusted at this theta (core.py: extended_lo_max_correction)
// (one workgroup per slot: 1024 threads stride over the partials, wave sums, sixteen wave totals added in order)
__global__ __launch_bounds__(1024) void lo_reduce_kernel(const double* __restrict__ part, int nparts, int DP, KernArgs ka, const int* __restrict__ flag,
                                                         double* __restrict__ g_ls, double* __restrict__ g_sf2, double* __restrict__ delta) {
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int q = blockIdx.x;
    const int slot = q == ka.d ? DP : q;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 1024) s += part[(size_t)p * (DP + 1) + slot];   // (15 dependent loads per thread at C5; 61 with 256 threads: 32 us)
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.0;
    for (int w2 = 0; w2 < 16; ++w2) s += red[w2];
    if (threadIdx.x == 0) {
      const double c = q == ka.d ? 2.0 * ka.sf2 * s : 2.0 * ka.inv_ls[q] * ka.sf2 * ka.sf2 * s;
      if (flag && *flag) {   // an inducing point beyond the fp16 range of the contraction's inputs: no correction, and the caller is told
        if (delta) delta[q] = __builtin_nan("");
      } else {
        if (q == ka.d) *g_sf2 += c;
        else g_ls[q] += c;
        if (delta) delta[q] = c;
      }
    }
  }
}

struct LoWs {
  double *Xs, *ys, *Zs, *yypart, *part, *unscale, *centre;
  uint16_t *Pl, *Kh;
  lo_u4* Bx;
  float* TT;
  int* flag;
  size_t bytes;
  int grid;
};
static LoWs carve_lo(void* ws, const StreamPlan& p, bool have_f16 = false) {
  Carver c(ws);
  LoWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.yypart = c.take<double>(256);
  const int64_t nrb = p.Npad / LO_T;
  const int ncb = p.Mp / LO_T;
  w.grid = (int)(((nrb + 7) / 8) * 8 * ncb);
  {
    const size_t g256 = (size_t)((p.Npad / 256 + 7) / 8) * 8 * (size_t)((p.Mp + 255) / 256);   // the 256 x 256 kernel's grid, 8 NG + 1 slots each
    const size_t a = (size_t)(w.grid > 0 ? w.grid : 1) * (p.DP + 1), b = g256 * (size_t)(8 * ((p.DP + 7) / 8) + 1);
    w.part = c.take<double>(a > b ? a : b);
  }
  const size_t mp2 = (size_t)(p.Mp + 255) / 256 * 256;   // (the 256 x 256 kernel's padding)
  w.Pl = c.take<uint16_t>(mp2 * mp2);
  w.unscale = c.take<double>(mp2);
  w.centre = c.take<double>(32);
  w.flag = c.take<int>(4);
  const size_t ng = (size_t)(p.DP + 7) / 8;   // groups of eight dimensions
  w.TT = c.take<float>(mp2 * L3_TTS * ng);
  w.Bx = c.take<lo_u4>((size_t)(p.Npad > 0 ? p.Npad : 1) * 4 * ng);   // Npad / 32 row tiles x NG groups x 128 vectors of 16 bytes
  w.Kh = have_f16 ? nullptr : c.take<uint16_t>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.Mp);   // (last: a caller that brings the image saves it)
  w.bytes = c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

extern "C" size_t sgp_suffstats_bwd_lo_workspace_bytes_ex(int64_t N, int M, int d, int have_f16) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  return carve_lo(nullptr, make_stream_plan(N, M, d), have_f16 != 0).bytes;
}
extern "C" size_t sgp_suffstats_bwd_lo_workspace_bytes(int64_t N, int M, int d) { return sgp_suffstats_bwd_lo_workspace_bytes_ex(N, M, d, 0); }

// Adds the low word's contribution to g_ls (d doubles) and g_sf2 IN PLACE, behind sgp_suffstats_bwd on the same stream with the same
// inputs and Phibar = the leading word.  Kfu_in: the fp64 K'_fu of this shard (sgp_kfu_len doubles) as pass 1 left it.  RBF (d <= 8 for the first version's kernel)
// (SGP_ERR_ARG / SGP_ERR_DIM otherwise: the caller then keeps the leading word's gradient); g_Z is not corrected.  delta (optional,
// d + 1 doubles): receives the correction itself.
extern "C" int sgp_suffstats_bwd_lo(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls,
                                    double sf2, const double* Phibar_lo, const double* Kfu_in, int64_t N, int M, int d, int kernel_id,
                                    double* g_ls, double* g_sf2, double* delta, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  return sgp_suffstats_bwd_lo_f16(X, ldx, y, Z, ldz, inv_ls, sf2, Phibar_lo, Kfu_in, nullptr, N, M, d, kernel_id, g_ls, g_sf2, delta, ws, ws_bytes,
                                  stream);
}
// ... with the fp16 image of K'_fu from sgp_suffstats_fwd_extended_f16 (Kfu_f16_in; Kfu_in may then be NULL and the workspace is the _ex
// size with have_f16 = 1): the 10 GB conversion pass at C5 (2.0 of 4.6 ms) is not run.
extern "C" int sgp_suffstats_bwd_lo_f16(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls,
                                        double sf2, const double* Phibar_lo, const double* Kfu_in, const uint16_t* Kfu_f16_in, int64_t N,
                                        int M, int d, int kernel_id, double* g_ls, double* g_sf2, double* delta, void* ws, size_t ws_bytes,
                                        sgp_stream_t stream) {
  if (!Z || !inv_ls || !Phibar_lo || (!Kfu_in && !Kfu_f16_in) || !g_ls || !g_sf2 || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id != SGP_KERNEL_RBF) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (N == 0) {
    if (delta) fill_zero(delta, (size_t)d + 1, (hipStream_t)stream);
    return check_launch();
  }
  StreamPlan p = make_stream_plan(N, M, d);
  LoWs w = carve_lo(ws, p, Kfu_f16_in != nullptr);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  const uint16_t* Kh = Kfu_f16_in ? Kfu_f16_in : w.Kh;
  hipStream_t st = (hipStream_t)stream;
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st);
  static const int lo_kernel = getenv("SGP_LO_KERNEL") ? atoi(getenv("SGP_LO_KERNEL")) : 3;   // A/B: 1 = the first version (128 x 128 tiles, register ring, fp64 contraction)
  if (lo_kernel != 3 && d > 8) return SGP_ERR_DIM;   // (the first version's kernel: d <= 8)
  const bool v3 = lo_kernel == 3;
  const int NG = (p.DP + 7) / 8;
  const int Mp2 = (p.Mp + L3_T - 1) / L3_T * L3_T, plm = v3 ? Mp2 : p.Mp;   // the padded edge of the low word's image
  if (v3) lo3_centre_kernel<<<1, 256, 0, st>>>(w.Zs, M, p.DP, w.centre, w.flag);
  lo_prep_kernel<<<plm, 256, 0, st>>>(Phibar_lo, M, plm, v3 ? 3 : 14, w.Pl, w.unscale);
  const int64_t nrb = p.Npad / LO_T;
  const int ncb = p.Mp / LO_T;
  if (!Kfu_f16_in) lo_kfu_f16_kernel<<<4096, 256, 0, st>>>(Kfu_in, (int64_t)p.Npad * p.Mp / 8, reinterpret_cast<uint4*>(w.Kh));
  int nparts = w.grid, part_dp = p.DP;
  if (v3) {
    lo3_tt_kernel<<<(Mp2 + 255) / 256, 256, 0, st>>>(w.Zs, w.unscale, w.centre, M, Mp2, p.DP, NG, w.TT, w.flag);
    const int64_t ngroups = p.Npad / 8;   // (row tile, k-step, lane half): 32 vectors per group of dimensions each
    lo_bx_kernel<<<(unsigned)((ngroups * 8 * NG + 255) / 256), 256, 0, st>>>(w.Xs, w.centre, ngroups, p.DP, NG, w.Bx);
    const int64_t nrb2 = p.Npad / L3_T;
    const int ncb2 = Mp2 / L3_T;
    nparts = (int)(((nrb2 + 7) / 8) * 8 * ncb2);
    part_dp = 8 * NG;
    static const int lo_var = getenv("SGP_LO_VARIANT") ? atoi(getenv("SGP_LO_VARIANT")) : 0;
    typedef void (*lo3_fn)(const uint16_t*, const uint16_t*, const lo_u4*, const float*, int, int, int, int64_t, int, double*);
    static const lo3_fn fns[3] = {kphi_lo3_kernel<0>, kphi_lo3_kernel<1>, kphi_lo3_kernel<2>};
    const lo3_fn fn = fns[lo_var < 0 || lo_var > 2 ? 0 : lo_var];
    constexpr int lds_bytes = L3_LDS_BYTES;
    static std::atomic<bool> attr_done[64];   // per device, as i8_contract
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return SGP_ERR_LAUNCH;
    if (dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
      for (int b = 0; b < 3; ++b)
        if (hipFuncSetAttribute((const void*)fns[b], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return SGP_ERR_LAUNCH;
      if (dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    fn<<<nparts, 512, lds_bytes, st>>>(Kh, w.Pl, w.Bx, w.TT, p.Mp, Mp2, NG, nrb2, ncb2, w.part);
  } else {
    switch (p.DP) {
      case 2: kphi_lo_kernel<2><<<w.grid, 256, 0, st>>>(Kfu_in, Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
      case 4: kphi_lo_kernel<4><<<w.grid, 256, 0, st>>>(Kfu_in, Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
      default: kphi_lo_kernel<8><<<w.grid, 256, 0, st>>>(Kfu_in, Kh, w.Pl, w.unscale, w.Xs, w.Zs, p.Mp, nrb, ncb, w.part); break;
    }
  }
  lo_reduce_kernel<<<d + 1, 1024, 0, st>>>(w.part, nparts, part_dp, ka, v3 ? w.flag : nullptr, g_ls, g_sf2, delta);
  return check_launch();
}
