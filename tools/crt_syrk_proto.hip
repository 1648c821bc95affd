// Prototype (diagnostic, not part of the library): Phi = K'^T K' on the integer matrix cores by MODULAR splitting.
//
//   q = rint(K' 2^53) in [0, 2^53 + 1];   r_i = q mod m_i, balanced into int8, for 16 pairwise-coprime moduli m_i <= 256
//   S_i = sum_n r_i,nI r_i,nJ  (int32, exact: |r r| <= 2^14, <= 131 072 rows per split)      -- ONE plain int8 SYRK per modulus
//   X = sum_n q_nI q_nJ  is the unique 0 <= X < P = prod m_i with X = S_i (mod m_i): CRT fold per split, exact (X < 2^106 rows << P = 2^125.5)
//
// 16 GEMMs instead of the 28 digit-pair GEMMs of csrc/sgp_suffstats_i8.hip, ONE int32 accumulator group per 32 x 32 tile (so a wave
// holds a 128 x 64 tile), and the EXACT integer sum (no dropped digit pairs).  The price: 16 instead of 7 bytes per element of K',
// no cross-plane operand reuse (an operand fragment feeds one modulus only), ~4 lane-operations per residue in the conversion.
//
// Residue planes in HBM: R[rb][i][m][16 bytes] = residue i of rows 16 rb .. 16 rb + 15 of column m (the digit-plane layout with 16 planes).
// Workgroup = 256 x 256 tile of the lower triangle x one split of the rows x ONE modulus; 8 waves (two per SIMD, the two groups half a
// step apart as in the library kernel), each a 128 x 64 tile = 4 x 2 MFMA tiles x 16 accumulators = 128 accumulator registers.
// 64-row stages (32 KB: 4 row blocks x 512 columns) travel global -> LDS by LDS-DMA through a ring of three.
//
//   build: hipcc --offload-arch=gfx950 -O3 tools/crt_syrk_proto.hip -o tools/crt_syrk_proto
//   run:   tools/crt_syrk_proto [N] [M] [nsplit] [reps]     (checks a small case against exact host integers first)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NMOD = 16;
__host__ __device__ constexpr int modulus(int i) {
  constexpr int M_[NMOD] = {256, 253, 251, 249, 247, 245, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193};
  return M_[i];
}
__host__ __device__ constexpr int bal(int v, int m) { return ((v % m) + m) % m > m / 2 ? ((v % m) + m) % m - m : ((v % m) + m) % m; }
__host__ __device__ constexpr int pow2mod(int e, int m) { int r = 1; for (int k = 0; k < e; ++k) r = (r * 2) % m; return r; }

constexpr int TI = 256, TJ = 256;       // tile of Phi per workgroup
constexpr int SCOLS = TI + TJ;          // columns of a stage (diagonal tiles fill the first 256 only)
constexpr int KSTEPS = 2;               // 32-row MFMA steps per stage
constexpr int RBS = 2 * KSTEPS;         // row blocks of 16 per stage
constexpr int STAGE_BYTES = RBS * SCOLS * 16;  // 32 768
#ifndef NSTAGE
#define NSTAGE 3
#endif
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;

__host__ __device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// synthetic K'[n][m] 2^53: smooth-ish magnitude spread with full-entropy low bits
__host__ __device__ inline uint64_t kq(int64_t n, int m) {
  const uint64_t h = mix((uint64_t)n * 1315423911ULL + (uint64_t)m * 2654435761ULL + 12345);
  const int sh = (int)(mix(h) % 12);  // magnitudes from 1 down to 2^-11
  return (h >> 11) >> sh;             // < 2^53
}

// ---------------------------------------------------------------------------------------------
// 1. residue planes: thread <-> (column m, 16 rows).  q = p0 + p1 2^14 + p2 2^28 + p3 2^42 with balanced 14-bit pieces;
//    v_i = p0 + p1 c1_i + p2 c2_i + p3 c3_i (|v| < 2^22: exact in fp32), r_i = v_i - m_i rint(v_i / m_i) in [-126, 126] (checked
//    exhaustively on the host for every |v| <= 2^13 383), two moduli per packed fp32 instruction; m = 256: the low byte itself.
// ---------------------------------------------------------------------------------------------
struct ResConst { f2 c1[8], c2[8], c3[8], inv[8], negm[8]; };
__host__ __device__ constexpr float cf(int e, int i) { return (float)bal(pow2mod(e, modulus(i)), modulus(i)); }

__device__ __forceinline__ void residues_of(uint64_t q, int e, unsigned (&out)[NMOD][4]) {
  int64_t t = (int64_t)q;
  const int p0 = (((int)t & 0x3fff) ^ 0x2000) - 0x2000; t = (t - p0) >> 14;
  const int p1 = (((int)t & 0x3fff) ^ 0x2000) - 0x2000; t = (t - p1) >> 14;
  const int p2 = (((int)t & 0x3fff) ^ 0x2000) - 0x2000; t = (t - p2) >> 14;
  const float f0 = (float)p0, f1 = (float)p1, f2_ = (float)p2, f3 = (float)(int)t;
  const f2 F0 = {f0, f0}, F1 = {f1, f1}, F2 = {f2_, f2_}, F3 = {f3, f3};
  const f2 MAGIC = {12582912.0f, 12582912.0f}, B128 = {128.0f, 128.0f};
  const int k = e >> 2, bsel = e & 3;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ia = 2 * j, ib = 2 * j + 1;
    const f2 C1 = {cf(14, ia), cf(14, ib)}, C2 = {cf(28, ia), cf(28, ib)}, C3 = {cf(42, ia), cf(42, ib)};
    const f2 INV = {1.0f / (float)modulus(ia), 1.0f / (float)modulus(ib)}, NEGM = {-(float)modulus(ia), -(float)modulus(ib)};
    const f2 v = __builtin_elementwise_fma(F3, C3, __builtin_elementwise_fma(F2, C2, __builtin_elementwise_fma(F1, C1, F0)));
    const f2 u = __builtin_elementwise_fma(v, INV, MAGIC);
    const f2 tt = u - MAGIC;
    const f2 r = __builtin_elementwise_fma(tt, NEGM, v) + B128;
    if (j > 0) out[ia][k] = __builtin_amdgcn_cvt_pk_u8_f32(r[0], bsel, out[ia][k]);
    out[ib][k] = __builtin_amdgcn_cvt_pk_u8_f32(r[1], bsel, out[ib][k]);
  }
  out[0][k] |= (((unsigned)q & 0xffu) ^ 0x80u) << (8 * bsel);  // m = 256: the low byte (two's complement after the final flip)
}

__global__ __launch_bounds__(256) void residues_kernel(int64_t nrb, int Mp, uint8_t* __restrict__ R) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int64_t rb = blockIdx.y;
  if (m >= Mp) return;
  unsigned out[NMOD][4];
#pragma unroll
  for (int i = 0; i < NMOD; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i][k] = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) residues_of(kq(rb * 16 + e, m), e, out);
#pragma unroll
  for (int i = 0; i < NMOD; ++i) {
    i4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (int)(out[i][k] ^ 0x80808080u);
    __builtin_nontemporal_store(v, reinterpret_cast<i4*>(R + (((size_t)rb * NMOD + i) * Mp + m) * 16));
  }
}

// ---------------------------------------------------------------------------------------------
// 2. contraction: one modulus per workgroup
// ---------------------------------------------------------------------------------------------
// wave classes of a 128 x 64 wave tile (4 x 2 MFMA tiles, bit 2 a + b): all / none / the two shapes a diagonal workgroup cuts
constexpr int CLS_FULL = 0xFF, CLS_NONE = 0, CLS_D0 = 0xFD, CLS_D1 = 0xD0;

// MODE 0: the kernel; 1: no LDS-DMA inside the loop; 2: no MFMA work
template <int CLS, bool DIAG, int MODE>
__device__ __forceinline__ void crt_tile_loop(uint8_t* lds, const uint8_t* __restrict__ Rm, size_t rbstride, int64_t s0, int64_t s1, int I0,
                                              int J0, int wave, int lane, int mod, uint8_t* __restrict__ res) {
  constexpr int NCG = DIAG ? 4 : 8;               // 64-column groups staged
  constexpr int PIECES = RBS * NCG;               // 1 KB pieces per stage
  constexpr int PPW = PIECES / 8;                 // per wave: 4 (2 on diagonal tiles)
  const int grp = wave >> 2, wj = wave & 3, wi = grp;
  const int l32 = lane & 31, lh = lane >> 5;
  i16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

  unsigned goff[PPW];
  int soff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int e = wave + 8 * k;
    const int rbl = e / NCG, cg = e % NCG;
    const int col = cg < 4 ? I0 + cg * 64 : J0 + (cg - 4) * 64;
    goff[k] = (unsigned)(rbl * rbstride + (size_t)(col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane((rbl * SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)RBS * rbstride;  // bytes per stage
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };
  const int64_t nst = s1 - s0;
  auto wait_stage = [&](int64_t sE) {
    if (sE + 1 >= nst)
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));                 // vmcnt(0)
    else
      __builtin_amdgcn_s_waitcnt((PPW & 15) | (7 << 4) | (15 << 8));        // vmcnt(PPW)
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  i4 a[KSTEPS][4], b[KSTEPS][2];
  const int jb = DIAG ? 0 : TI;
  auto reads = [&](int64_t sidx) {
    if (CLS != CLS_NONE) {
      const uint8_t* sb = lds + (int)(sidx % NSTAGE) * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const uint8_t* rbp = sb + (2 * ks + lh) * (SCOLS * 16);
#pragma unroll
        for (int y = 0; y < 2; ++y)
          if ((CLS >> y) & 0x55) b[ks][y] = *reinterpret_cast<const i4*>(rbp + (jb + 64 * wj + 32 * y + l32) * 16);
#pragma unroll
        for (int x = 0; x < 4; ++x)
          if ((CLS >> (2 * x)) & 3) a[ks][x] = *reinterpret_cast<const i4*>(rbp + (128 * wi + 32 * x + l32) * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto half = [&](int ks, bool dma_on, int64_t sE) {
    const bool pre = dma_on && sE + 2 < nst && MODE != 1;
    const uint8_t* gnext = Rm + (size_t)(s0 + sE + 2) * gstride;
    const int snext = (int)((sE + 2) % NSTAGE) * STAGE_BYTES;
    int issued = 0, kpiece = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (((CLS >> (2 * x + y)) & 1) && MODE != 2)
          acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks][x], b[ks][y], acc[x][y], 0, 0, 0);
        ++issued;
        if (dma_on && (issued & 1) == 0 && kpiece < PPW) {   // after slots 2, 4, 6, 8
          __builtin_amdgcn_sched_barrier(0);
          if (pre) dma_piece(gnext, snext, kpiece);
          __builtin_amdgcn_sched_barrier(0);
          ++kpiece;
        }
      }
  };
  if (nst > 0) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(Rm + (size_t)s0 * gstride, 0, k);
    if (nst > 1) {
#pragma unroll
      for (int k = 0; k < PPW; ++k) dma_piece(Rm + (size_t)(s0 + 1) * gstride, STAGE_BYTES, k);
    }
    if (grp == 0) {
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        wait_stage(sidx);
        bar();  // E(s)
        reads(sidx);
        half(0, true, sidx);
        bar();  // O(s)
        half(1, false, sidx);
      }
      wait_stage(nst);
      bar();    // E(n)
    } else {
      wait_stage(0);
      bar();    // E(0)
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        bar();  // O(s)
        reads(sidx);
        half(0, true, sidx);
        __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));  // lgkmcnt(0): my reads of stage s have left the LDS queue
        wait_stage(sidx + 1);
        bar();  // E(s + 1)
        half(1, false, sidx + 1);
      }
    }
  }
  // residue of the int32 sums: x = S mod m in [0, m), one byte per element
  if (CLS != CLS_NONE) {
    const double dm = (double)mod, inv = 1.0 / dm;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
        if ((CLS >> (2 * x + y)) & 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double s = (double)acc[x][y][r];
            int v = (int)(s - dm * floor(s * inv));
            v = v < 0 ? v + mod : (v >= mod ? v - mod : v);
            const int row = 128 * wi + 32 * x + (r >> 2) * 8 + lh * 4 + (r & 3);
            res[row * TJ + 64 * wj + 32 * y + l32] = (uint8_t)v;
          }
        }
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void crt_syrk_kernel(const uint8_t* __restrict__ R, int Mp, int64_t nstages, int nsplit, int ntiles,
                                                          uint8_t* __restrict__ res) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // id -> (xcd, tile, group): the tiles of one (split, modulus) group share id % 8, i.e. one XCD's L2 under round-robin dispatch
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, g = (jj / ntiles) * 8 + xcd;
  if (g >= nsplit * NMOD) return;
  const int split = g / NMOD, im = g % NMOD;
  int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int I0 = ti * TI, J0 = tj * TJ;
  const int64_t per = (nstages + nsplit - 1) / nsplit;
  const int64_t s0 = split * per, s1 = (s0 + per < nstages) ? s0 + per : nstages;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int MTAB[NMOD] = {256, 253, 251, 249, 247, 245, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193};
  int mod = 256;
#pragma unroll
  for (int i = 0; i < NMOD; ++i) mod = im == i ? MTAB[i] : mod;
  const size_t rbstride = (size_t)NMOD * Mp * 16;
  const uint8_t* Rm = R + (size_t)im * Mp * 16;
  uint8_t* out = res + ((size_t)g * ntiles + t) * (TI * TJ);
  if (ti != tj) {
    crt_tile_loop<CLS_FULL, false, MODE>(lds, Rm, rbstride, s0, s1, I0, J0, wave, lane, mod, out);
  } else {
    const int wi = wave >> 2, wj = wave & 3;
    const int c = wi == 0 ? (wj == 0 ? 1 : wj == 1 ? 2 : 3) : (wj < 2 ? 0 : wj == 2 ? 1 : 2);  // 0 full, 1 D0, 2 D1, 3 none
    if (c == 0) crt_tile_loop<CLS_FULL, true, MODE>(lds, Rm, rbstride, s0, s1, I0, J0, wave, lane, mod, out);
    else if (c == 1) crt_tile_loop<CLS_D0, true, MODE>(lds, Rm, rbstride, s0, s1, I0, J0, wave, lane, mod, out);
    else if (c == 2) crt_tile_loop<CLS_D1, true, MODE>(lds, Rm, rbstride, s0, s1, I0, J0, wave, lane, mod, out);
    else crt_tile_loop<CLS_NONE, true, MODE>(lds, Rm, rbstride, s0, s1, I0, J0, wave, lane, mod, out);
  }
}

// ---------------------------------------------------------------------------------------------
// 3. CRT fold: X = sum_i ((x_i w_i) mod m_i) (P / m_i) mod P, exact in 128-bit integers, one rounding to fp64; splits summed in order
// ---------------------------------------------------------------------------------------------
struct CrtConst { uint64_t plo[NMOD], phi[NMOD], Plo, Phi; int w[NMOD]; };
__constant__ CrtConst g_crt;

__device__ __forceinline__ double u128_to_double(u128 x) {
  const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
  if (hi == 0) return (double)lo;
  const int lz = __builtin_clzll(hi);
  const uint64_t top = lz ? (hi << lz) | (lo >> (64 - lz)) : hi;       // the 64 leading bits
  const uint64_t rest = lz ? (lo << lz) : lo;
  return ldexp((double)(top | (rest != 0 ? 1ull : 0ull)), 64 - lz);    // sticky bit: ONE rounding
}

__global__ __launch_bounds__(256) void crt_fold_kernel(const uint8_t* __restrict__ res, int nsplit, int ntiles, int M, double* __restrict__ Phi) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(e / (TI * TJ)), rc = (int)(e % (TI * TJ));
  if (t >= ntiles) return;
  int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int gi = ti * TI + rc / TJ, gj = tj * TJ + rc % TJ;
  if (gj > gi || gi >= M) return;
  const u128 P = ((u128)g_crt.Phi << 64) | g_crt.Plo;
  double sum = 0.0;
  for (int sp = 0; sp < nsplit; ++sp) {
    u128 acc = 0;
#pragma unroll
    for (int i = 0; i < NMOD; ++i) {
      const int x = res[((size_t)(sp * NMOD + i) * ntiles + t) * (TI * TJ) + rc];
      const unsigned yv = (unsigned)(x * g_crt.w[i]) % (unsigned)modulus(i);
      acc += (((u128)g_crt.phi[i] << 64) | g_crt.plo[i]) * yv;
      if (acc >= P) acc -= P;
    }
    sum += u128_to_double(acc);
  }
  sum *= 0x1p-106;
  Phi[(size_t)gi * M + gj] = sum;
  Phi[(size_t)gj * M + gi] = sum;
}

// ---------------------------------------------------------------------------------------------
static void setup_crt() {
  CrtConst c;
  u128 P = 1;
  for (int i = 0; i < NMOD; ++i) P *= (u128)modulus(i);
  c.Plo = (uint64_t)P; c.Phi = (uint64_t)(P >> 64);
  for (int i = 0; i < NMOD; ++i) {
    const u128 Pi = P / (u128)modulus(i);
    c.plo[i] = (uint64_t)Pi; c.phi[i] = (uint64_t)(Pi >> 64);
    const int pm = (int)(Pi % (u128)modulus(i));
    int w = 1;
    while ((pm * w) % modulus(i) != 1) ++w;
    c.w[i] = w;
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_crt), &c, sizeof(c)));
  printf("P = 2^%.2f\n", log2((double)(uint64_t)(P >> 64)) + 64.0);
}

template <int MODE>
static double run(int64_t N, int M, int nsplit, bool check, int reps) {
  const int Mp = (M + 255) / 256 * 256;
  const int64_t Npad = (N + 63) / 64 * 64, nrb = Npad / 16, nstages = Npad / 64;
  const int nb = Mp / 256, ntiles = nb * (nb + 1) / 2;
  uint8_t *R, *res;
  double* Phi;
  CK(hipMalloc(&R, (size_t)nrb * NMOD * Mp * 16));
  CK(hipMalloc(&res, (size_t)nsplit * NMOD * ntiles * TI * TJ));
  CK(hipMalloc(&Phi, (size_t)Mp * Mp * 8));
  hipEvent_t e0, e1, e2, e3;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  CK(hipFuncSetAttribute((const void*)crt_syrk_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int ngroups = nsplit * NMOD;
  const int grid = ((ngroups + 7) / 8) * 8 * ntiles;
  float ta = 0, tc = 0, tf = 0;
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    residues_kernel<<<dim3(Mp / 256, (unsigned)nrb), 256>>>(nrb, Mp, R);
    CK(hipEventRecord(e1));
    crt_syrk_kernel<MODE><<<grid, 512, LDS_BYTES>>>(R, Mp, nstages, nsplit, ntiles, res);
    CK(hipEventRecord(e2));
    crt_fold_kernel<<<(unsigned)(((size_t)ntiles * TI * TJ + 255) / 256), 256>>>(res, nsplit, ntiles, Mp, Phi);
    CK(hipEventRecord(e3));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tc, e1, e2)); CK(hipEventElapsedTime(&tf, e2, e3));
    if (tc < best) best = tc;
    if (!check) printf("  mode %d rep %d: residues %.3f ms  contraction %.3f ms  fold %.3f ms\n", MODE, r, ta, tc, tf);
  }
  if (check) {
    std::vector<double> h((size_t)Mp * Mp);
    CK(hipMemcpy(h.data(), Phi, h.size() * 8, hipMemcpyDeviceToHost));
    // exact: per split in 128-bit integers, rounded once, splits added in order (what the fold does)
    const int64_t per = (nstages + nsplit - 1) / nsplit;
    double worst = 0.0, big = 0.0;
    int bad = 0;
    for (int i = 0; i < M; i += 37)
      for (int j = 0; j <= i; j += 29) {
        double sum = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) {
          u128 x = 0;
          const int64_t r0 = sp * per * 64, r1 = std::min<int64_t>(Npad, (sp + 1) * per * 64);
          for (int64_t n = r0; n < r1; ++n) x += (u128)kq(n, i) * kq(n, j);
          const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
          long double v = (long double)hi * 18446744073709551616.0L + (long double)lo;  // 64-bit mantissa: then one rounding to double
          sum += (double)v;
        }
        sum *= 0x1p-106;
        const double d = fabs(h[(size_t)i * Mp + j] - sum);
        if (d > worst) worst = d;
        if (sum > big) big = sum;
        if (d > 2e-16 * sum) ++bad;
      }
    printf("check N=%lld M=%d nsplit=%d: max |dPhi| = %.3e (max Phi %.3e), entries off by more than an ulp: %d\n", (long long)N, M, nsplit, worst, big, bad);
  }
  CK(hipFree(R)); CK(hipFree(res)); CK(hipFree(Phi));
  return best;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1048576;
  const int M = argc > 2 ? atoi(argv[2]) : 1024;
  const int nsplit = argc > 3 ? atoi(argv[3]) : 16;
  const int reps = argc > 4 ? atoi(argv[4]) : 4;
  setup_crt();
  run<0>(4096, 512, 2, true, 1);
  run<0>(8192, 256, 1, true, 1);
  run<0>(3000, 1024, 3, true, 1);
  const double t0 = run<0>(N, M, nsplit, false, reps);
  const double macs = 16.0 * (double)N * M * (M + 1) / 2;
  printf("N=%lld M=%d nsplit=%d: contraction best %.3f ms = %.0f int8 TOP/s (16 N M (M+1) ops)\n", (long long)N, M, nsplit, t0, 2 * macs / t0 * 1e-9);
  const double t1 = run<1>(N, M, nsplit, false, 2);
  const double t2 = run<2>(N, M, nsplit, false, 2);
  printf("no DMA in the loop: %.3f ms; no MFMA: %.3f ms\n", t1, t2);
  return 0;
}
