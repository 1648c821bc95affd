// Single-launch evaluation of the collapsed bound and its gradient for SMALL problems (M <= 128) -- the size class of
// every HMC run the reference publishes (models/bayesian_sgpr_hmc.py:58-80,144-157: N ~ 250-1300, M = 100; BASELINE
// configs C1 / C2).  The multi-launch path needs ~60 dependent kernels per leapfrog there (0.39 ms for microseconds of
// arithmetic); here ONE cooperative launch does everything, with the hyper-parameters read from DEVICE memory (so a
// sampler can keep its state on the GPU) and in the PyMC3 op order:
//
//     Kuu = k(Z,Z) + J I ; L = chol(Kuu) ; A = L^-1 K_uf ; B = I + A A^T / s2 ; LB = chol(B) ; c0 = LB^-1 (A y)
//     F = -[ N/2 log 2pi + N/2 log s2 + sum log diag LB + (yy/s2 - c0.c0/s2^2)/2 + (kappa - sum A o A)/(2 s2) ]
//
// B is positive definite by construction and nothing of size cond(Kuu) is ever subtracted: every solve is a blocked
// forward / backward substitution on the matrix cores (16-column panels, 16 x 16 diagonal-block inverses from the
// factorization), never an explicit L^-1.  Reverse pass, closed form in the same basis (g = B^-1 A y):
//     Abar  = [ (I - B^-1 - g g^T / s2^2) A + g y^T / s2 ] / s2          column by column:  two solves with LB
//     Kbar_uf = L^-T Abar                                                  one solve with L^T, contracted with dK_uf on the fly
//     Kbar_uu = -1/2 L^-T (B + B^-1 - 2 I + g g^T / s2^2) L^-1             the same solves applied to the columns of I
//
// Work decomposition (one workgroup = 4 waves; all workgroups co-resident, <= 66 of the 256 CUs):
//     workgroup 0            the chain: Kuu, chol(Kuu), [wait] chol(B) with c0 riding along, g, F, final reduction
//     workgroups 1 .. MP/64  Kbar_uu ("virtual slabs": 64 columns of I each go through the slab pipeline), tr B^-1
//     the others             row slabs of 64 data rows: assemble K_uf straight into the MFMA accumulator layout, solve,
//                            partial A A^T / A y; later the reverse pass of the same rows
// A slab lives in REGISTERS during the solves: lane (wave w, l15, l4) holds element (row 16 w + l15, column 16 pb + 4 s
// + l4) in component s of block pb -- transposed 16 x 16 blocks in the accumulator layout, which is also the B-operand
// layout of the next MFMA, so X_pb^T = Dinv_pb Y_pb and Y_q -= L[q][pb] X_pb^T chain without leaving the registers.
// Synchronisation: agent-scope flags / counters behind release fences, acquire fences after every wait (the buffers are
// rewritten on every evaluation, so stale L2 lines of another XCD must be dropped); spins are bounded (SGP_INFO_TIMEOUT).
// All cross-workgroup sums go through partial arrays reduced in a fixed order: results are bit-reproducible.
#include "sgp_potrf.hpp"
#include "sgp_nuts.hpp"
#include "sgp_composite.hpp"

namespace sgp {

constexpr int SM_SLAB = 64;
constexpr int SM_MAXD = 24;  // input dimensions of the stationary kernels (Elevator: d = 18); LDS of the M <= 128 class is full at 24
constexpr int SM_MAX_ROWWG = 208;  // row workgroups at most (one 64-row slab each up to N = 13 312: Elevator's 13 279 training rows;
                                   // 64 until round 3 -- N = 13 279 then walked 3-4 slabs per workgroup on a quarter of the chip)
constexpr int SM_SPIN_LIMIT = 1 << 22;
constexpr int SM_SYNC_STRIDE = 32;  // ints between sync words: one cache line each
enum { SY_L = 0, SY_PART = 1, SY_SLICE = 2, SY_LB = 3, SY_GRAD = 4, SY_ABORT = 5, SY_Q = 6, SY_REQ = 7, SY_DONE = 8, SY_G = 9, SY_KUU = 10, SY_ACK = 11, SY_GZ = 12, SY_WORDS = 13 };
constexpr int SM_GP = 40;  // doubles per gradient partial.  Stationary: g_ls[d] at 0.., g_sf2 at 16; composite: dF/d(block) at its
                          // 33 slots.  Both: tr B^-1, u.g, g.g at SM_XTRA + 0, 1, 2
constexpr int SM_XTRA = 34;
constexpr int SM_GZ_SPREAD = 32;  // gradient partials (workgroups) above which the row workgroups share the dF/dZ reduction
constexpr int SM_MAXCP = 20;  // free parameters of a composite kernel (4 terms x (amplitude + 2 x (lengthscale, aux)))

struct SmallArgs {
  const double* X; int64_t ldx; const double* y; const double* Z; int64_t ldz; const double* theta;
  int N, M, d, kid, mode, want_grad, want_gz;
  int nh;     // kernel hyper-parameter outputs: out = [value | nh | noise | logmarg | trace]  (d + 1, 33, or ncp)
  int ndim;   // entries of theta (d + 2 ; 34 for a composite block + s2 ; ncp + 1 for the composite NUTS target)
  double jitter;
  // SGP_KERNEL_COMPOSITE: the structure (term / factor types, fixed aux values) and, for the NUTS target, which block slot
  // each sampled log-parameter fills (role 0 amplitude sd -> amp2 = v^2, 1 lengthscale, 2 aux) with its Normal prior sd
  CompSpec cs;
  int ncp;
  int cp_slot[SM_MAXCP], cp_role[SM_MAXCP];
  double cp_sd[SM_MAXCP];
  int nslab, grow;
  double *Lk, *dinvK, *Bm, *Lb, *dinvB, *A, *Ppart, *upart, *spart, *u, *c0, *g, *h, *gpart, *gzpart, *Qm;
  int *sync, *info;
  double* out;
  double* gZ;
  unsigned long long* stamps;  // optional (tools/small_eval_phases.py): [workgroup][16] s_memrealtime ticks (100 MHz)
};

struct SmHyp {
  double inv_ls[SM_MAXD], ls[SM_MAXD];
  double sf2, s2;
  double kdiag;  // k(x, x): sf2, or the sum of a composite kernel's amplitudes
  int ok;
  CompSpec cs;   // composite kernels: the structure with this evaluation's parameter values
};

template <int MP>
struct SmSlabShared {
  static constexpr int NB16 = MP / 16, NBL = NB16 * (NB16 + 1) / 2, SLD = MP + 2;
  union {
    double Lblk[NBL][16][17];  // lower 16 x 16 blocks of the current triangular factor, block (q, p) at q (q + 1) / 2 + p
    double S[SM_SLAB][SLD];    // a solved slab, [row][column] (SYRK operands, column sums, the E matrix of dF/dZ)
  };
  double Dinv[NB16][16][17];   // inverses of its diagonal blocks
  double zs[MP][SM_MAXD + 1];  // scaled inducing inputs
  double xs[SM_SLAB][SM_MAXD + 1];
  double ys[SM_SLAB];
  double gv[MP], hv[MP];
  double red[4][MP];
  double acc[SM_GP];
};

// the chain workgroup's view: the scratch of the 64 x 64 tile factorization (sgp_potrf.hpp) + the scaled inducing inputs
template <int MP>
struct SmChainShared {
  DfShared df;
  double zs[MP][SM_MAXD + 1];
  double vc0[MP], vg[MP], vh[MP];  // c0, g, h while the chain's vector solves run
};

template <int MP>
union SmShared {
  SmChainShared<MP> ch;
  SmSlabShared<MP> sl;
};

__device__ __forceinline__ int sm_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every thread of the workgroup calls; false = the launch has been aborted
__device__ __forceinline__ bool sm_wait_ge(int* word, int target, int* abort_word, int* dead) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (sm_ld(word) < target) {
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      if ((spins & 127) == 0 && sm_ld(abort_word) != 0) { *dead = 1; break; }
      if (spins > SM_SPIN_LIMIT) {
        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return *dead == 0;
}
// thread 0 of the chain workgroup only: wait until the other workgroups have READ request r's theta (cumulative
// acknowledgements) before theta is overwritten for request r + 1 -- nothing is read behind this wait, so no fence
__device__ __forceinline__ bool sm_wait_ack(int* word, int target, int* abort_word) {
  int spins = 0;
  while (sm_ld(word) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SM_SPIN_LIMIT || ((spins & 127) == 0 && sm_ld(abort_word) != 0)) {
      __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}
__device__ __forceinline__ void sm_publish_set(int* word, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sm_publish_add(int* word) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double sm_kprofile(int kid, double r2) {
  if (kid == SGP_KERNEL_RBF) return kprofile<SGP_KERNEL_RBF>(r2);
  if (kid == SGP_KERNEL_MATERN32) return kprofile<SGP_KERNEL_MATERN32>(r2);
  return kprofile<SGP_KERNEL_MATERN52>(r2);
}
__device__ __forceinline__ void sm_kprofile_grad(int kid, double r2, double& k, double& h) {
  if (kid == SGP_KERNEL_RBF) kprofile_grad<SGP_KERNEL_RBF>(r2, k, h);
  else if (kid == SGP_KERNEL_MATERN32) kprofile_grad<SGP_KERNEL_MATERN32>(r2, k, h);
  else kprofile_grad<SGP_KERNEL_MATERN52>(r2, k, h);
}

// v[i] <- scale * k'(v[i]) for NE independent entries: ONE switch on the kernel id, then straight-line code, so the NE
// exp() chains interleave (a lone wave per SIMD pays the full latency of every dependent fp64 instruction otherwise:
// 0.4 us per serial exp(), measured)
template <int NE>
__device__ __forceinline__ void sm_profile_vec(int kid, double (&v)[NE], double scale) {
  if (kid == SGP_KERNEL_RBF) {
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i] = scale * kprofile<SGP_KERNEL_RBF>(v[i]);
  } else if (kid == SGP_KERNEL_MATERN32) {
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i] = scale * kprofile<SGP_KERNEL_MATERN32>(v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i] = scale * kprofile<SGP_KERNEL_MATERN52>(v[i]);
  }
}
// r2[i] -> (k'[i] in r2[i], h[i] = dk'/dr2)
template <int NE>
__device__ __forceinline__ void sm_profile_grad_vec(int kid, double (&r2)[NE], double (&h)[NE]) {
  if (kid == SGP_KERNEL_RBF) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const double t = r2[i];
      kprofile_grad<SGP_KERNEL_RBF>(t, r2[i], h[i]);
    }
  } else if (kid == SGP_KERNEL_MATERN32) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const double t = r2[i];
      kprofile_grad<SGP_KERNEL_MATERN32>(t, r2[i], h[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const double t = r2[i];
      kprofile_grad<SGP_KERNEL_MATERN52>(t, r2[i], h[i]);
    }
  }
}

// one parameter slot of the block (amplitude / lengthscale / aux) <- v
__device__ __forceinline__ void sm_set_slot(CompSpec& cs, int slot, double v) {
  const int t = (slot - 1) >> 3, rem = (slot - 1) & 7;
  if (rem == 0) cs.amp2[t] = v;
  else if ((rem - 2) % 3 == 1) cs.ls[t][(rem - 2) / 3] = v;
  else cs.aux[t][(rem - 2) / 3] = v;
}
// theta -> hyper-parameters (thread 0), the same arithmetic in every workgroup.
//   mode 0: theta = [ls_1..d | sf2 | s2] ;  mode 1: theta = [log ls_1..d | log sig_f | log sig_n]
template <bool COMP>
__device__ __forceinline__ void sm_hypers(const SmallArgs& a, SmHyp& h) {
  if (threadIdx.x == 0) {
    int ok = 1;
    if constexpr (COMP) {
      // mode 0: theta = [parameter block (SGP_COMP_LEN) | s2] ; SGP_SMALL_HMC: theta = [log of every free parameter | log sigma]
      CompSpec cs = a.cs;
      double tn;
      if (a.mode == 0) {
        for (int t = 0; t < cs.nterms; ++t) {
          const int base = 1 + 8 * t;
          cs.amp2[t] = a.theta[base];
          ok = ok && (cs.amp2[t] > 0.0);
          for (int f = 0; f < cs.nfac[t]; ++f) {
            cs.ls[t][f] = a.theta[base + 2 + 3 * f + 1];
            cs.aux[t][f] = a.theta[base + 2 + 3 * f + 2];
            ok = ok && (cs.ls[t][f] > 0.0);
            if (cs.type[t][f] == SGP_FAC_RATQUAD || cs.type[t][f] == SGP_FAC_PERIODIC) ok = ok && (cs.aux[t][f] > 0.0);
          }
        }
        tn = a.theta[SGP_COMP_LEN];
        h.s2 = tn;
      } else {
        for (int k = 0; k < a.ncp; ++k) {
          const double t = a.theta[k], v = exp(t);
          if (!(fabs(t) < 150.0)) ok = 0;
          sm_set_slot(cs, a.cp_slot[k], a.cp_role[k] == 0 ? v * v : v);
        }
        tn = a.theta[a.ncp];
        if (!(fabs(tn) < 150.0)) ok = 0;
        h.s2 = exp(2.0 * tn);
      }
      double kd = 0.0;
      for (int t = 0; t < cs.nterms; ++t) kd += cs.amp2[t];
      cs.kdiag = kd;
      h.cs = cs;
      h.kdiag = kd;
      h.sf2 = 1.0;
      for (int j = 0; j < SM_MAXD; ++j) { h.ls[j] = 1.0; h.inv_ls[j] = j < a.d ? 1.0 : 0.0; }
      if (!(h.s2 > 0.0) || !(kd > 0.0)) ok = 0;
      h.ok = ok;
    } else {
    for (int j = 0; j < a.d; ++j) {
      const double t = a.theta[j];
      const double l = a.mode ? exp(t) : t;
      if (!(fabs(t) < 300.0) || !(l > 0.0)) ok = 0;
      h.ls[j] = l;
      h.inv_ls[j] = 1.0 / l;
    }
    for (int j = a.d; j < SM_MAXD; ++j) { h.ls[j] = 1.0; h.inv_ls[j] = 0.0; }
    const double tf = a.theta[a.d], tn = a.theta[a.d + 1];
    h.sf2 = a.mode ? exp(2.0 * tf) : tf;
    h.s2 = a.mode ? exp(2.0 * tn) : tn;
    h.kdiag = h.sf2;
    if (!(fabs(tf) < 300.0) || !(fabs(tn) < 300.0) || !(h.sf2 > 0.0) || !(h.s2 > 0.0)) ok = 0;
    if (a.mode && (fabs(tf) > 150.0 || fabs(tn) > 150.0)) ok = 0;
    h.ok = ok;
    }
  }
  __syncthreads();
}

// ---- composite kernels, NE pairs (a, b_k) at a time -------------------------------------------------------------------
// One switch per factor, then straight-line code over the NE pairs, so the exp / log / sin chains of different pairs
// interleave: a lone wave per SIMD otherwise waits out every dependent fp64 instruction (pair-at-a-time comp_value() cost
// 3.6 us per entry and thread, measured: 58 us for a 64 x 64 K_uu).  bco(k, j) = coordinate j of b_k.
// F = factor value; with GRAD also dF/d ls and dF/d aux (formulas: comp_factor, sgp_composite.hpp).
template <int NE, bool GRAD, class BF>
__device__ __forceinline__ void sm_factor_vec(int ty, double ls, double aux, const double (&r2)[NE], const double* a, BF bco, int d,
                                              double (&F)[NE], double (&DL)[NE], double (&DA)[NE]) {
  const double il2 = 1.0 / (ls * ls), il = 1.0 / ls;
  if (ty == SGP_FAC_EXPQUAD) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      F[k] = exp(-0.5 * r2[k] * il2);
      if (GRAD) { DL[k] = F[k] * r2[k] * il2 * il; DA[k] = 0.0; }
    }
  } else if (ty == SGP_FAC_MATERN32) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double t = 1.7320508075688772 * sqrt(r2[k]) * il, e = exp(-t);
      F[k] = (1.0 + t) * e;
      if (GRAD) { DL[k] = t * t * e * il; DA[k] = 0.0; }
    }
  } else if (ty == SGP_FAC_MATERN52) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double t = 2.23606797749979 * sqrt(r2[k]) * il, e = exp(-t);
      F[k] = (1.0 + t + t * t * (1.0 / 3.0)) * e;
      if (GRAD) { DL[k] = t * t * (1.0 + t) * e * il * (1.0 / 3.0); DA[k] = 0.0; }
    }
  } else if (ty == SGP_FAC_RATQUAD) {
    const double c = 0.5 * il2 / aux;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double w = fma(c, r2[k], 1.0), lw = log(w);
      F[k] = exp(-aux * lw);
      if (GRAD) {
        const double iw = 1.0 / w;
        DL[k] = F[k] * iw * r2[k] * il2 * il;
        DA[k] = F[k] * ((w - 1.0) * iw - lw);
      }
    }
  } else {  // SGP_FAC_PERIODIC
    const double w = 3.141592653589793 / aux;
    double S[NE], T[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) S[k] = T[k] = 0.0;
    for (int j = 0; j < d; ++j) {
      const double aj = a[j];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const double dl = aj - bco(k, j);
        double sn, cs;
        sincos(w * dl, &sn, &cs);
        S[k] = fma(sn, sn, S[k]);
        if (GRAD) T[k] = fma(2.0 * sn * cs, dl, T[k]);  // sin(2 w delta) delta
      }
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      F[k] = exp(-0.5 * S[k] * il2);
      if (GRAD) { DL[k] = F[k] * S[k] * il2 * il; DA[k] = 0.5 * F[k] * il2 * T[k] * w / aux; }
    }
  }
}

template <int NE, class BF>
__device__ __forceinline__ void sm_comp_r2(const double* a, BF bco, int d, double (&r2)[NE]) {
#pragma unroll
  for (int k = 0; k < NE; ++k) r2[k] = 0.0;
  for (int j = 0; j < d; ++j) {
    const double aj = a[j];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double df = aj - bco(k, j);
      r2[k] = fma(df, df, r2[k]);
    }
  }
}

// v[k] = k(a, b_k)
template <int NE, class BF>
__device__ __forceinline__ void sm_comp_value_vec(const CompSpec& cs, const double* a, BF bco, int d, double (&v)[NE]) {
  double r2[NE], F[NE], term[NE];
  sm_comp_r2<NE>(a, bco, d, r2);
#pragma unroll
  for (int k = 0; k < NE; ++k) v[k] = 0.0;
#pragma unroll 1
  for (int t = 0; t < cs.nterms; ++t) {
#pragma unroll
    for (int k = 0; k < NE; ++k) term[k] = cs.amp2[t];
#pragma unroll 1
    for (int f = 0; f < cs.nfac[t]; ++f) {
      sm_factor_vec<NE, false>(cs.type[t][f], cs.ls[t][f], cs.aux[t][f], r2, a, bco, d, F, F, F);
#pragma unroll
      for (int k = 0; k < NE; ++k) term[k] *= F[k];
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) v[k] += term[k];
  }
}

// sum_k kb[k] dk(a, b_k)/d(parameter), handed per parameter slot of the block to emit(slot, this lane's sum): the loop
// over terms stays a loop (one copy of the factor code), so the sums cannot live in statically indexed registers
template <int NE, class BF, class EM>
__device__ __forceinline__ void sm_comp_accum_vec(const CompSpec& cs, const double* a, BF bco, int d, const double (&kb)[NE], EM emit) {
  double r2[NE];
  sm_comp_r2<NE>(a, bco, d, r2);
#pragma unroll 1
  for (int t = 0; t < cs.nterms; ++t) {
    const int base = 1 + 8 * t;
    double F0[NE], L0[NE], A0[NE];
    sm_factor_vec<NE, true>(cs.type[t][0], cs.ls[t][0], cs.aux[t][0], r2, a, bco, d, F0, L0, A0);
    const double amp = cs.amp2[t];
    double s_amp = 0.0, s_l0 = 0.0, s_a0 = 0.0;
    if (cs.nfac[t] == 2) {
      double F1[NE], L1[NE], A1[NE];
      sm_factor_vec<NE, true>(cs.type[t][1], cs.ls[t][1], cs.aux[t][1], r2, a, bco, d, F1, L1, A1);
      double s_l1 = 0.0, s_a1 = 0.0;
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const double k0 = kb[k] * F0[k], k1 = kb[k] * F1[k];
        s_amp = fma(k0, F1[k], s_amp);
        s_l0 = fma(k1, L0[k], s_l0);
        s_a0 = fma(k1, A0[k], s_a0);
        s_l1 = fma(k0, L1[k], s_l1);
        s_a1 = fma(k0, A1[k], s_a1);
      }
      emit(base + 6, amp * s_l1);
      emit(base + 7, amp * s_a1);
    } else {
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        s_amp = fma(kb[k], F0[k], s_amp);
        s_l0 = fma(kb[k], L0[k], s_l0);
        s_a0 = fma(kb[k], A0[k], s_a0);
      }
    }
    emit(base, s_amp);
    emit(base + 3, amp * s_l0);
    emit(base + 4, amp * s_a0);
  }
}

// ---- register-resident slab: NB16 transposed 16 x 16 blocks per wave ---------------------------------------------
template <int NB16>
struct SlabRegs {
  d4 b[NB16];
};
__device__ __forceinline__ int sm_blk(int q, int p) { return q * (q + 1) / 2 + p; }

// Y <- Y L^-T  (every slab row t solves L a = t)
template <int NB16>
__device__ __forceinline__ void sm_trsm_fwd(SlabRegs<NB16>& Y, const double (*Lb)[16][17], const double (*Di)[16][17], int l15, int l4) {
#pragma unroll
  for (int pb = 0; pb < NB16; ++pb) {
    d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) xa = mfma16(Di[pb][l15][4 * sq + l4], Y.b[pb][sq], xa);
    Y.b[pb] = xa;
#pragma unroll
    for (int q = pb + 1; q < NB16; ++q)
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) Y.b[q] = mfma16(-Lb[sm_blk(q, pb)][l15][4 * sq + l4], xa[sq], Y.b[q]);
  }
}
// Y <- Y L^-1  (every slab row t solves L^T x = t)
template <int NB16>
__device__ __forceinline__ void sm_trsm_bwd(SlabRegs<NB16>& Y, const double (*Lb)[16][17], const double (*Di)[16][17], int l15, int l4) {
#pragma unroll
  for (int pb = NB16 - 1; pb >= 0; --pb) {
    d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) xa = mfma16(Di[pb][4 * sq + l4][l15], Y.b[pb][sq], xa);
    Y.b[pb] = xa;
#pragma unroll
    for (int q = 0; q < pb; ++q)
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) Y.b[q] = mfma16(-Lb[sm_blk(pb, q)][4 * sq + l4][l15], xa[sq], Y.b[q]);
  }
}

// global lower-triangular factor (ld MP) + the inverses of its 16 x 16 diagonal blocks ([block][16][16]) -> LDS blocks.
// Every load is issued before the first LDS store: one round trip to a line another XCD has just written.
template <int MP>
__device__ __forceinline__ void sm_stage_factor(SmSlabShared<MP>& sl, const double* __restrict__ L, const double* __restrict__ dinv) {
  constexpr int NB16 = MP / 16, NBL = NB16 * (NB16 + 1) / 2, NP = (NBL + 3) / 4;
  const int t = threadIdx.x, grp = t >> 6, tb = t & 63, r = tb >> 2, c4 = (tb & 3) * 4;
  d2 v[NP][2], dv[NB16 / 4][2];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int b = 4 * i + grp;
    if (b < NBL) {
      int q = 0;
      while ((q + 1) * (q + 2) / 2 <= b) ++q;
      const int p = b - q * (q + 1) / 2;
      const double* src = L + (size_t)(16 * q + r) * MP + 16 * p + c4;
      v[i][0] = *reinterpret_cast<const d2*>(src);
      v[i][1] = *reinterpret_cast<const d2*>(src + 2);
    }
  }
#pragma unroll
  for (int i = 0; i < NB16 / 4; ++i) {
    const double* src = dinv + (size_t)(4 * i + grp) * 256 + r * 16 + c4;
    dv[i][0] = *reinterpret_cast<const d2*>(src);
    dv[i][1] = *reinterpret_cast<const d2*>(src + 2);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int b = 4 * i + grp;
    if (b < NBL) {
      sl.Lblk[b][r][c4] = v[i][0][0];
      sl.Lblk[b][r][c4 + 1] = v[i][0][1];
      sl.Lblk[b][r][c4 + 2] = v[i][1][0];
      sl.Lblk[b][r][c4 + 3] = v[i][1][1];
    }
  }
#pragma unroll
  for (int i = 0; i < NB16 / 4; ++i) {
    const int b = 4 * i + grp;
    sl.Dinv[b][r][c4] = dv[i][0][0];
    sl.Dinv[b][r][c4 + 1] = dv[i][0][1];
    sl.Dinv[b][r][c4 + 2] = dv[i][1][0];
    sl.Dinv[b][r][c4 + 3] = dv[i][1][1];
  }
}

// x <- L^-1 x (trans = false) or L^-T x (trans = true) for the factor staged in sl.Lblk / sl.Dinv; x (MP doubles) in LDS.
// Wave 0 works, every thread calls.  ~1.3 us at MP = 128: each workgroup derives c0 = LB^-1 u and g = LB^-T c0 itself
// instead of waiting for the chain workgroup to do it.
template <int MP>
__device__ __forceinline__ void sm_vec_solve_blocks(SmSlabShared<MP>& sl, double* xv, bool trans) {
  constexpr int NB16 = MP / 16;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  __syncthreads();
  if (tid < 64) {
#pragma unroll 1
    for (int pp = 0; pp < NB16; ++pp) {
      const int p = trans ? NB16 - 1 - pp : pp;
      double part = 0.0;
      if (!trans) {
        for (int t = l4; t < p; t += 4) {
          const double (*B)[17] = sl.Lblk[sm_blk(p, t)];
#pragma unroll
          for (int k = 0; k < 16; ++k) part = fma(B[l15][k], xv[16 * t + k], part);
        }
      } else {
        for (int t = p + 1 + l4; t < NB16; t += 4) {
          const double (*B)[17] = sl.Lblk[sm_blk(t, p)];
#pragma unroll
          for (int k = 0; k < 16; ++k) part = fma(B[k][l15], xv[16 * t + k], part);
        }
      }
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      const double r = xv[16 * p + l15] - part;
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) x = fma(trans ? sl.Dinv[p][k][l15] : sl.Dinv[p][l15][k], __shfl(r, k, 64), x);
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) xv[16 * p + l15] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
}

// out <- L^-1 in (trans = false) or L^-T in (trans = true) for a factor in GLOBAL memory (ld MP), 64 x 64 tiles: wave 0,
// lane <-> row, 64-step readlane chains with the tile's row / column preloaded into registers.  `in` / `out` live in LDS
// (an agent-scope fence per tile -- an L2 write-back -- cost more than the chains).  Every thread calls.
template <int MP>
__device__ __forceinline__ void sm_vec_solve_tiles(const double* __restrict__ L, const double* in, double* out, bool trans) {
  constexpr int NB64 = MP / 64;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    for (int jj = 0; jj < NB64; ++jj) {
      const int jb = trans ? NB64 - 1 - jj : jj;
      double lv[64];  // trans: lv[c] = L[c][lane] (column of the diagonal tile) ; else lv[c] = L[lane][c] (row)
      if (trans) {
#pragma unroll
        for (int c = 0; c < 64; ++c) lv[c] = L[(size_t)(jb * 64 + c) * MP + jb * 64 + lane];
      } else {
        const double* lsrc = L + (size_t)(jb * 64 + lane) * MP + (size_t)jb * 64;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const d2 t2 = *reinterpret_cast<const d2*>(lsrc + 2 * k);
          lv[2 * k] = t2[0];
          lv[2 * k + 1] = t2[1];
        }
      }
      double rr = in[jb * 64 + lane];
      for (int pp = 0; pp < jj; ++pp) {  // the tiles already solved
        const int p = trans ? NB64 - 1 - pp : pp;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int k = 0; k < 64; k += 2) {
          const double l0 = trans ? L[(size_t)(p * 64 + k) * MP + jb * 64 + lane] : L[(size_t)(jb * 64 + lane) * MP + p * 64 + k];
          const double l1 = trans ? L[(size_t)(p * 64 + k + 1) * MP + jb * 64 + lane] : L[(size_t)(jb * 64 + lane) * MP + p * 64 + k + 1];
          s0 = fma(l0, out[p * 64 + k], s0);
          s1 = fma(l1, out[p * 64 + k + 1], s1);
        }
        rr -= s0 + s1;
      }
      const double dinv = 1.0 / L[(size_t)(jb * 64 + lane) * (MP + 1)];
      double mine = 0.0;
      if (trans) {
#pragma unroll
        for (int c = 63; c >= 0; --c) {
          const double xc = readlane_f64(rr, c) * readlane_f64(dinv, c);
          if (lane == c) mine = xc;
          rr = fma(-lv[c], xc, rr);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 64; ++c) {
          const double xc = readlane_f64(rr, c) * readlane_f64(dinv, c);
          if (lane == c) mine = xc;
          rr = fma(-lv[c], xc, rr);
        }
      }
      out[jb * 64 + lane] = mine;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
}

// ---- the kernel -------------------------------------------------------------------------------------------------
template <int MP>
struct SmKernelShared {
  SmShared<MP> sh;
  SmHyp hyp;
  int dead;
  double redw[4];
  double gsum[SM_GP];
};

// One evaluation, executed by every workgroup of the launch.  `ev` = 1-based count of evaluations this launch has run
// (flags carry it, counters are cumulative: ev x contributors), so a persistent kernel calls this repeatedly without
// clearing anything; `persistent` = false additionally leaves the sync words zero for the next launch.
template <int MP, bool COMP>
__device__ __forceinline__ void sm_eval_body(const SmallArgs& a, SmKernelShared<MP>& ks, int ev, bool persistent) {
  constexpr int NB16 = MP / 16, NBL = NB16 * (NB16 + 1) / 2, NBW = (NBL + 3) / 4, NB64 = MP / 64;
  SmShared<MP>& sh = ks.sh;
  SmHyp& hyp = ks.hyp;
  int& dead = ks.dead;
  double (&redw)[4] = ks.redw;
  double (&gsum)[SM_GP] = ks.gsum;
  SmSlabShared<MP>& sl = sh.sl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wg = blockIdx.x;
  const int d = a.d, M = a.M, N = a.N;
  int* sy = a.sync;
  int* abortw = sy + SY_ABORT * SM_SYNC_STRIDE;
  auto stamp = [&](int k) {
    if (a.stamps && tid == 0) a.stamps[(size_t)wg * 16 + k] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  const double sf2 = hyp.sf2, s2 = hyp.s2;
  constexpr bool comp = COMP;  // SGP_KERNEL_COMPOSITE (sums of products of isotropic factors, values through comp_value()):
                               // its own instantiation, so the stationary kernels carry none of its code or registers
  const int role = wg == 0 ? 0 : (wg <= NB64 ? 1 : 2);
  const int ngrad = a.grow + NB64;  // workgroups that contribute gradient partials

  // scaled inducing inputs (rows >= M zero)
  auto stage_z = [&]() {
    for (int e = tid; e < MP * (SM_MAXD + 1); e += 256) {
      const int m = e / (SM_MAXD + 1), j = e - m * (SM_MAXD + 1);
      sl.zs[m][j] = (m < M && j < d) ? a.Z[(size_t)m * a.ldz + j] * hyp.inv_ls[j] : 0.0;
    }
  };
  // r2 of this lane's element (pb, s) against slab row nl:  acc over j outside
  auto block_sum = [&](double v) -> double {  // sum over the 256 threads, valid everywhere
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) redw[w] = v;
    __syncthreads();
    return (redw[0] + redw[1]) + (redw[2] + redw[3]);
  };

  // =================================================================================================================
  if (role == 0) {
    DfShared& df = sh.ch.df;
    double (*zs)[SM_MAXD + 1] = sh.ch.zs;
    const int r = tid & 63, g = w;                       // thread <-> (row r, columns 16 g ..) of a diagonal tile
    const int wi = w >> 1, wj = w & 1;
    if (tid == 0) *a.info = 0;
    for (int e = tid; e < MP * (SM_MAXD + 1); e += 256) {
      const int m = e / (SM_MAXD + 1), j = e - m * (SM_MAXD + 1);
      zs[m][j] = (m < M && j < d) ? a.Z[(size_t)m * a.ldz + j] * hyp.inv_ls[j] : 0.0;
    }
    __syncthreads();
    // 16 entries of the padded Kuu: row i, columns j0 + cstride * k (k = 0..15)
    auto kuu16 = [&](int i, int j0, int cstride, double (&v)[16]) {
      if (comp) {
        sm_comp_value_vec<16>(hyp.cs, zs[i], [&](int k, int q) { return zs[j0 + cstride * k][q]; }, d, v);
      } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = 0.0;
      for (int q = 0; q < d; ++q) {
        const double zi = zs[i][q];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const double dz = zi - zs[j0 + cstride * k][q];
          v[k] = fma(dz, dz, v[k]);
        }
      }
      sm_profile_vec<16>(a.kid, v, sf2);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = j0 + cstride * k;
        if (i >= M || j >= M) v[k] = i == j ? 1.0 : 0.0;
        else if (i == j) v[k] += a.jitter;
      }
    };
    double ldsum[2] = {0.0, 0.0};
    // Two factorizations through ONE copy of the tile code (pass 0: Kuu, its entries evaluated straight into the
    // registers the factorization works in; pass 1: B from the slices).  64 x 64 tiles, one workgroup, no flags:
    //   diagonal tile  -> diag_factor64_fast (sgp_potrf.hpp: pivot chain pipelined over the four waves)
    //   tile (1, 0)    -> X = T L00^-T on the matrix cores from the panels / block inverses the factorization left in LDS
    //   tile (1, 1)    -= X X^T from LDS, then factored
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      double* Lg = pass == 0 ? a.Lk : a.Lb;
      double* dg = pass == 0 ? a.dinvK : a.dinvB;
      if (pass == 1) {
        if (!sm_wait_ge(sy + SY_SLICE * SM_SYNC_STRIDE, ev * a.grow, abortw, &dead)) {
          if (tid == 0) *a.info = SGP_INFO_TIMEOUT;
          return;
        }
        stamp(3);
      }
      int bad_first = 0;
      double ld = 0.0;
      double xn[16], yv[16];  // tile (1, 1) rows / tile (1, 0) in the solve layout, fetched ahead of the factorization of (0, 0)
      auto fetch_lower_tiles = [&]() {
        const double* s11 = Lg + (size_t)(64 + r) * MP + 64 + 16 * g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const d2 v = *reinterpret_cast<const d2*>(s11 + 2 * k);
          xn[2 * k] = v[0];
          xn[2 * k + 1] = v[1];
        }
        const double* s10 = Lg + (size_t)(64 + 16 * g + l15) * MP + l4;
#pragma unroll
        for (int k = 0; k < 16; ++k) yv[k] = s10[16 * (k >> 2) + 4 * (k & 3)];
      };
#pragma unroll 1
      for (int jd = 0; jd < NB64; ++jd) {
        double x[16];
        if (jd > 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) x[k] = xn[k];
        } else if (pass == 0) {
          kuu16(r, 16 * g, 1, x);  // tile (0, 0) of Kuu evaluated straight into the factorization's registers
        } else {
          const double* src = a.Lb + (size_t)r * MP + 16 * g;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const d2 v = *reinterpret_cast<const d2*>(src + 2 * k);
            x[2 * k] = v[0];
            x[2 * k + 1] = v[1];
          }
          if (NB64 > 1) fetch_lower_tiles();  // B is complete: the loads land while tile (0, 0) is factored
        }
        if (jd > 0) {  // minus X X^T (X = tile (jd, jd - 1), still in df.Ts)
          d4 acc[2][2];
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
          df_mac_lds(df.Ts, acc, wi, wj, l15, l4);
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
              for (int q = 0; q < 4; ++q) df.Ts[wi * 32 + u * 16 + l4 + 4 * q][wj * 32 + v * 16 + l15] = acc[u][v][q];
          __syncthreads();
#pragma unroll
          for (int k = 0; k < 16; ++k) x[k] -= df.Ts[r][16 * g + k];
        }
        if (tid < 4) df.prog[tid] = 0;
        if (tid == 0) df.bad = 0;
        __syncthreads();
        diag_factor64_fast(x, df.Sp, df.Lt, df.prog, df.rdiag3, df.Dinv, &df.bad, r, g);
        if (df.bad != 0 && bad_first == 0) bad_first = 64 * jd + df.bad;
        {  // L(jd, jd) rows, the zero tile above it, the 16 x 16 block inverses
          double* dst = Lg + (size_t)(64 * jd + r) * MP + 64 * jd + 16 * g;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            *reinterpret_cast<d2*>(dst + 2 * k) = d2{(16 * g + 2 * k <= r) ? x[2 * k] : 0.0, (16 * g + 2 * k + 1 <= r) ? x[2 * k + 1] : 0.0};
          if (jd + 1 < NB64) {
            double* up = Lg + (size_t)(64 * jd + r) * MP + 64 * (jd + 1) + 16 * g;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<d2*>(up + 2 * k) = d2{0.0, 0.0};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;  // (blk, row, col) = (idx >> 8, (idx >> 4) & 15, idx & 15)
            dg[(size_t)jd * 1024 + idx] = df.Dinv[idx >> 8][(idx >> 4) & 15][idx & 15];
          }
          if ((r >> 4) == g) ld += log(x[r & 15]);
        }
        if (jd + 1 < NB64) {
          // X = T L(jd,jd)^-T : wave g <-> rows 16 g .. of the tile below, transposed blocks in the accumulator layout
          d4 yb[4];
          const int trow = 64 * (jd + 1) + 16 * g + l15;
          if (pass == 0) {
            // tiles (1, 0) and (1, 1) of Kuu were evaluated by the first Kbar_uu workgroup (idle until LB exists) while
            // this workgroup factored tile (0, 0)
            if (!sm_wait_ge(sy + SY_KUU * SM_SYNC_STRIDE, ev, abortw, &dead)) {
              if (tid == 0) *a.info = SGP_INFO_TIMEOUT;
              return;
            }
            fetch_lower_tiles();
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) yb[k >> 2][k & 3] = yv[k];
          d4 xb[4];
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) {
            d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) xa = mfma16(df.Dinv[pb][l15][4 * sq + l4], yb[pb][sq], xa);
            xb[pb] = xa;
#pragma unroll
            for (int q = pb + 1; q < 4; ++q)
#pragma unroll
              for (int sq = 0; sq < 4; ++sq) yb[q] = mfma16(-df.Sp[pb][16 * q + l15][4 * sq + l4], xa[sq], yb[q]);
          }
          __syncthreads();  // the panels have been read: Ts may be written
          double* dst = Lg + (size_t)trow * MP + 64 * jd + l4;
#pragma unroll
          for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
              dst[16 * pb + 4 * sq] = xb[pb][sq];
              df.Ts[16 * g + l15][16 * pb + 4 * sq + l4] = xb[pb][sq];
            }
          __syncthreads();
        }
      }
      ld = block_sum(ld);
      ldsum[pass] = ld;
      if (tid == 0 && bad_first != 0 && *a.info == 0) *a.info = (pass == 0 ? 0 : M) + bad_first;
      if (pass == 0) {
        stamp(1);
        sm_publish_set(sy + SY_L * SM_SYNC_STRIDE, ev);
        stamp(2);
      }
    }
    stamp(4);
    if (a.want_grad) sm_publish_set(sy + SY_LB * SM_SYNC_STRIDE, ev);
    else {
      __threadfence();
      __syncthreads();
    }
    stamp(5);
    // c0 = LB^-1 u (for the value), g = LB^-T c0 = B^-1 u and h = L^-T g: the vectors the reverse pass needs only in its
    // last step (the contraction), computed here while the other workgroups run their solves; one flag hands them over
    double cc = 0.0;
    {
      SmChainShared<MP>& ch = sh.ch;
      if (tid < MP) ch.vg[tid] = a.u[tid];
      __syncthreads();
      sm_vec_solve_tiles<MP>(a.Lb, ch.vg, ch.vc0, false);
      for (int i = tid; i < MP; i += 256) cc = fma(ch.vc0[i], ch.vc0[i], cc);
      if (a.want_grad) {
        sm_vec_solve_tiles<MP>(a.Lb, ch.vc0, ch.vg, true);
        sm_vec_solve_tiles<MP>(a.Lk, ch.vg, ch.vh, true);
        if (tid < MP) {
          a.g[tid] = ch.vg[tid];
          a.h[tid] = ch.vh[tid];
        }
        sm_publish_set(sy + SY_G * SM_SYNC_STRIDE, ev);
      }
    }
    cc = block_sum(cc);
    const double ld = ldsum[1];
    stamp(6);
    // scalars
    const double sumA2 = a.spart[0], yy = a.spart[1];  // reduced by the slice stage
    const double Nd = (double)N, kappa = Nd * hyp.kdiag;
    const double LOG2PI = 1.8378770664093453;
    const double quad = yy / s2 - cc / (s2 * s2);
    const double logmarg = -(0.5 * Nd * LOG2PI + 0.5 * Nd * log(s2) + ld + 0.5 * quad);
    const double trace_term = (kappa - sumA2) / (2.0 * s2);
    const double F = logmarg - trace_term;
    if (sm_ld(abortw) != 0 && tid == 0) *a.info = SGP_INFO_TIMEOUT;
    if (!a.want_grad) {
      if (tid == 0) {
        a.out[0] = F;
        a.out[a.nh + 2] = logmarg;
        a.out[a.nh + 3] = trace_term;
      }
      __syncthreads();
      if (!persistent)
        for (int e = tid; e < SY_WORDS * SM_SYNC_STRIDE; e += 256) sy[e] = 0;  // ready for the next launch
      return;
    }
    stamp(7);

    // ---- gradients --------------------------------------------------------------------------------------------------
    if (!sm_wait_ge(sy + SY_GRAD * SM_SYNC_STRIDE, ev * ngrad, abortw, &dead)) {
      if (tid == 0) *a.info = SGP_INFO_TIMEOUT;
      return;
    }
    stamp(8);
    {
      // slot = tid & 63 (< SM_GP), the four waves split the partials (g = w, w + 4, ...), eight loads in flight, combined
      // through LDS in a fixed order: with 210 partials one thread per slot was 53 us of dependent loads at the very end
      const int slot = tid & 63;
      double s = 0.0;
      if (slot < SM_GP) {
#pragma unroll 8
        for (int g = w; g < ngrad; g += 4) s += a.gpart[(size_t)g * SM_GP + slot];
      }
      __syncthreads();
      double* red4 = reinterpret_cast<double*>(&sh.ch.df);  // 256 doubles of the factorization's tile scratch: free by now
      static_assert(sizeof(DfShared) >= 256 * sizeof(double), "scratch for the gradient partial sums");
      if (slot < SM_GP) red4[w * 64 + slot] = s;
      __syncthreads();
      if (tid < SM_GP) gsum[tid] = (red4[tid] + red4[64 + tid]) + (red4[128 + tid] + red4[192 + tid]);
    }
    if (a.want_gz && a.gZ) {
      if (ngrad > SM_GZ_SPREAD) {  // the row workgroups reduce dF/dZ between them (see their last phase)
        if (!sm_wait_ge(sy + SY_GZ * SM_SYNC_STRIDE, ev * a.grow, abortw, &dead)) {
          if (tid == 0) *a.info = SGP_INFO_TIMEOUT;
          return;
        }
      } else {
        for (int e = tid; e < M * d; e += 256) {
          double s = 0.0;
#pragma unroll 8
          for (int g = 0; g < ngrad; ++g) s += a.gzpart[(size_t)g * MP * SM_MAXD + e];
          a.gZ[e] = s;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const double trBinv = gsum[SM_XTRA], ug = gsum[SM_XTRA + 1], gg = gsum[SM_XTRA + 2];
      const double g_sf2 = gsum[SM_MAXD] - Nd / (2.0 * s2);  // + kappabar dkappa/dsf2
      const double s22 = s2 * s2;
      const double g_s2 = -0.5 * (-((double)MP - trBinv) / s2 + Nd / s2 - yy / s22 + 2.0 * ug / (s22 * s2) - (ug - gg) / (s22 * s2)
                                  - kappa / s22 + sumA2 / s22);
      if (comp) {
        // dF/d(block): the contraction sums at the parameter slots, + kappabar dkappa/d(amp2_t) = -N / (2 s2) on the amplitudes
        for (int t = 0; t < hyp.cs.nterms; ++t) gsum[1 + 8 * t] -= Nd / (2.0 * s2);
        if (a.mode == 0) {
          a.out[0] = F;
          for (int p = 0; p < SGP_COMP_LEN; ++p) a.out[1 + p] = gsum[p];
          a.out[1 + SGP_COMP_LEN] = g_s2;
        } else {
          // the NUTS target of experiments/co2_bayesian_sgpr_hmc.py:99-158: log-parameters ~ Normal(0, sd) (sampled in log
          // space, no Jacobian), sigma ~ HalfNormal(1) log-transformed
          double lp = F;
          for (int k = 0; k < a.ncp; ++k) {
            const double t = a.theta[k], v = exp(t), sd = a.cp_sd[k];
            const double dF = gsum[a.cp_slot[k]];
            lp += -0.5 * (t / sd) * (t / sd) - log(sd) - 0.9189385332046727;
            a.out[1 + k] = (a.cp_role[k] == 0 ? 2.0 * v * v * dF : v * dF) - t / (sd * sd);
          }
          const double tn = a.theta[a.ncp], sig2 = exp(2.0 * tn);
          lp += -0.22579135264472741 - 0.5 * sig2 + tn;  // 0.5 log(2 / pi)
          a.out[1 + a.ncp] = 2.0 * sig2 * g_s2 - sig2 + 1.0;
          a.out[0] = lp;
        }
      } else if (a.mode == 0) {
        a.out[0] = F;
        for (int j = 0; j < d; ++j) a.out[1 + j] = gsum[j];
        a.out[1 + d] = g_sf2;
        a.out[2 + d] = g_s2;
      } else {
        // NUTS target of models/bayesian_sgpr_hmc.py:60-71: ls ~ Gamma(2, 1), sig_f, sig_n ~ HalfCauchy(1), log transforms
        double lp = F, sumth = 0.0;
        const double c = log(2.0) - log(3.141592653589793);
        for (int j = 0; j < d; ++j) {
          const double l = hyp.ls[j];
          lp += log(l) - l;
          sumth += a.theta[j];
          a.out[1 + j] = l * (gsum[j] + 1.0 / l - 1.0) + 1.0;
        }
        const double sf = exp(a.theta[d]), sn = exp(a.theta[d + 1]);
        lp += (c - log1p(sf * sf)) + (c - log1p(sn * sn));
        sumth += a.theta[d] + a.theta[d + 1];
        a.out[1 + d] = sf * (2.0 * sf * g_sf2 - 2.0 * sf / (1.0 + sf * sf)) + 1.0;
        a.out[2 + d] = sn * (2.0 * sn * g_s2 - 2.0 * sn / (1.0 + sn * sn)) + 1.0;
        a.out[0] = lp + sumth;
      }
      a.out[a.nh + 2] = logmarg;
      a.out[a.nh + 3] = trace_term;
    }
    __syncthreads();
    if (!persistent)
      for (int e = tid; e < SY_WORDS * SM_SYNC_STRIDE; e += 256) sy[e] = 0;
    stamp(9);
    return;
  }

  // =================================================================================================================
  // slab pipeline shared by the row workgroups (role 2) and the Kuu-adjoint workgroup (role 1)
  // dF/dZ partial of this workgroup (slot wg - 1 of gzpart): accumulated in its own line of global scratch -- L2 resident, 8 of its
  // M d entries per thread and slab -- rather than in LDS, which at M <= 128 has no room for M x d more doubles beside d <= 24
  double* __restrict__ gzmine = a.gzpart + (size_t)(wg > 0 ? wg - 1 : 0) * MP * SM_MAXD;
  // contraction of kbar (in Y, dF/dK_uf of this slab) with dK/d(ls, sf2, Z); data rows in sl.xs (scaled), validity masks
  auto contract = [&](const SlabRegs<NB16>& Y, int nvalid, double zscale) __attribute__((always_inline)) {
    // nvalid: slab rows that are real ; zscale: 1 for K_uf, 2 for the symmetric K_uu (both arguments are inducing inputs)
    const int nl = 16 * w + l15;
    if (comp) {
      // composite kernels: factor derivatives of 8 pairs at a time, summed over the wave per parameter slot
      if (lane < SGP_COMP_LEN) sl.red[w][lane] = 0.0;
      auto emit = [&](int slot, double v) {
        v = wave_sum(v);
        if (lane == 0) sl.red[w][slot] += v;
      };
      auto ysel = [&](int c, int h, int sq) -> double {  // Y.b[2 c + h][sq] for a run-time (wave-uniform) c: selects, no
        double v = Y.b[h][sq];                             // dynamically indexed registers
        if (c == 1) v = Y.b[2 + h][sq];
        if constexpr (NB16 > 4) {
          if (c == 2) v = Y.b[4 + h][sq];
          if (c == 3) v = Y.b[6 + h][sq];
        }
        return v;
      };
#pragma unroll 1
      for (int c = 0; c < NB16 / 2; ++c) {  // column blocks 2 c, 2 c + 1 ; a loop: one copy of the factor code
        double kb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int m = 32 * c + 4 * k + l4;
          kb[k] = (nl < nvalid && m < M) ? ysel(c, k >> 2, k & 3) : 0.0;
        }
        sm_comp_accum_vec<8>(hyp.cs, sl.xs[nl], [&](int k, int q) { return sl.zs[32 * c + 4 * k + l4][q]; }, d, kb, emit);
      }
      __syncthreads();
      if (tid >= 1 && tid < SGP_COMP_LEN && ((tid - 1) & 7) != 1 && ((tid - 1) & 7) != 2 && ((tid - 1) & 7) != 5)
        sl.acc[tid] += (sl.red[0][tid] + sl.red[1][tid]) + (sl.red[2][tid] + sl.red[3][tid]);
      __syncthreads();
      return;
    }
    double E[NB16][4];
    double ksum = 0.0;
    {
      double r2[NB16][4];
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) r2[pb][s] = 0.0;
      for (int j = 0; j < d; ++j) {
        const double xj = sl.xs[nl][j];
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double df = xj - sl.zs[16 * pb + 4 * s + l4][j];
            r2[pb][s] = fma(df, df, r2[pb][s]);
          }
      }
      double (&rv)[NB16 * 4] = reinterpret_cast<double (&)[NB16 * 4]>(r2);
      double (&hv)[NB16 * 4] = reinterpret_cast<double (&)[NB16 * 4]>(E);
      sm_profile_grad_vec<NB16 * 4>(a.kid, rv, hv);  // r2 <- k', E <- dk'/dr2
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bool live = nl < nvalid && (16 * pb + 4 * s + l4) < M;
          const double kb = live ? Y.b[pb][s] : 0.0;
          ksum = fma(kb, r2[pb][s], ksum);
          E[pb][s] = kb * sf2 * E[pb][s];
        }
    }
    ksum = wave_sum(ksum);
    if (lane == 0) sl.red[w][SM_MAXD] = ksum;
    // two dimensions per trip and four partial sums each: the 32-term sums were one dependent FMA chain per dimension followed
    // by a dependent wave reduction -- at d = 18 (Elevator) that latency, not the arithmetic, was most of the contraction's 45 us.
    // (xs / zs are zero beyond column d - 1 and have SM_MAXD + 1 columns: j + 1 = d reads zeros.)
    for (int j = 0; j < d; j += 2) {
      const double x0 = sl.xs[nl][j], x1 = sl.xs[nl][j + 1];
      double p0[4] = {0.0, 0.0, 0.0, 0.0}, p1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double* zr = &sl.zs[16 * pb + 4 * s + l4][j];
          const double d0 = x0 - zr[0], d1 = x1 - zr[1];
          p0[s] = fma(E[pb][s] * d0, d0, p0[s]);
          p1[s] = fma(E[pb][s] * d1, d1, p1[s]);
        }
      double s0 = (p0[0] + p0[1]) + (p0[2] + p0[3]), s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {  // the two wave reductions interleaved
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
      }
      if (lane == 0) {
        sl.red[w][j] = -2.0 * hyp.inv_ls[j] * s0;  // d r2 / d ls_j = -2 df_j^2 / ls_j
        if (j + 1 < d) sl.red[w][j + 1] = -2.0 * hyp.inv_ls[j + 1] * s1;
      }
    }
    __syncthreads();
    if (tid <= SM_MAXD && (tid < d || tid == SM_MAXD)) sl.acc[tid] += (sl.red[0][tid] + sl.red[1][tid]) + (sl.red[2][tid] + sl.red[3][tid]);
    if (a.want_gz) {
      // E -> LDS (over the factor blocks: the solves of this slab are done), then thread <-> (m, j)
      __syncthreads();
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) sl.S[nl][16 * pb + 4 * s + l4] = E[pb][s];
      __syncthreads();
      for (int e = tid; e < M * d; e += 256) {
        const int m = e / d, j = e - m * d;
        const double zj = sl.zs[m][j];
        double s = 0.0;
        for (int r = 0; r < SM_SLAB; ++r) s = fma(sl.S[r][m], sl.xs[r][j] - zj, s);
        gzmine[e] += -2.0 * zscale * hyp.inv_ls[j] * s;  // d r2 / d z_mj = -2 (x~ - z~) / ls_j
      }
    }
    __syncthreads();
  };

  stage_z();
  if (tid < SM_GP) sl.acc[tid] = 0.0;
  if (a.want_gz)
    for (int e = tid; e < M * d; e += 256) gzmine[e] = 0.0;  // element e is read, summed into and written by this thread only
  __syncthreads();

  if (role == 2) {
    const int rw = wg - 1 - NB64;
    // ---- forward: A = L^-1 K_uf for this workgroup's slabs, partial A A^T, A y, sum A o A, yy -----------------------
    // K_uf of a slab straight into the register layout (needs theta only: the first slab is assembled while the chain
    // workgroup is still factoring Kuu)
    // always_inline: an out-of-line copy receives the register slab through memory -- and the one the compiler made for
    // MP = 128 / composite faulted when reached from the persistent kernel's out-of-line evaluation call (cause not established)
    auto assemble = [&](int n0, SlabRegs<NB16>& Y) __attribute__((always_inline)) {
      __syncthreads();
      for (int e = tid; e < SM_SLAB * (SM_MAXD + 1); e += 256) {
        const int r = e / (SM_MAXD + 1), j = e - r * (SM_MAXD + 1);
        sl.xs[r][j] = (n0 + r < N && j < d) ? a.X[(size_t)(n0 + r) * a.ldx + j] * hyp.inv_ls[j] : 0.0;
      }
      if (tid < SM_SLAB) sl.ys[tid] = n0 + tid < N ? a.y[n0 + tid] : 0.0;
      __syncthreads();
      const int nl = 16 * w + l15;
      const bool rowlive = n0 + nl < N;
      double r2[NB16][4];
      if (comp) {
#pragma unroll
        for (int c = 0; c < NB16 / 4; ++c) {  // 16 pairs at a time: column blocks 4 c .. 4 c + 3
          double (&v16)[16] = reinterpret_cast<double (&)[16]>(r2[4 * c]);
          sm_comp_value_vec<16>(hyp.cs, sl.xs[nl], [&](int k, int q) { return sl.zs[64 * c + 4 * k + l4][q]; }, d, v16);
        }
      } else {
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) r2[pb][s] = 0.0;
      for (int j = 0; j < d; ++j) {
        const double xj = sl.xs[nl][j];
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double df = xj - sl.zs[16 * pb + 4 * s + l4][j];
            r2[pb][s] = fma(df, df, r2[pb][s]);
          }
      }
      double (&rv)[NB16 * 4] = reinterpret_cast<double (&)[NB16 * 4]>(r2);
      sm_profile_vec<NB16 * 4>(a.kid, rv, sf2);
      }
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) Y.b[pb][s] = (rowlive && (16 * pb + 4 * s + l4) < M) ? r2[pb][s] : 0.0;
    };
    SlabRegs<NB16> Y;
    assemble(rw * SM_SLAB, Y);
    stamp(1);
    if (!sm_wait_ge(sy + SY_L * SM_SYNC_STRIDE, ev, abortw, &dead)) return;
    stamp(2);
    d4 pacc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) pacc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double uacc = 0.0, a2acc = 0.0, yyacc = 0.0;  // thread m < MP: column sums over this workgroup's slabs
    for (int sb = rw; sb < a.nslab; sb += a.grow) {
      const int n0 = sb * SM_SLAB;
      if (sb != rw) assemble(n0, Y);
      __syncthreads();
      sm_stage_factor<MP>(sl, a.Lk, a.dinvK);
      __syncthreads();
      if (sb == rw) stamp(8);
      if (sb == rw) stamp(9);
      sm_trsm_fwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
      __syncthreads();  // everybody is done with the factor blocks: the slab takes their place
      if (sb == rw) stamp(10);
      {
        const int nl = 16 * w + l15;
        double* arow = a.A + (size_t)(n0 + nl) * MP;
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            sl.S[nl][16 * pb + 4 * s + l4] = Y.b[pb][s];
            arow[16 * pb + 4 * s + l4] = Y.b[pb][s];
          }
      }
      __syncthreads();
      {  // column sums A^T y, sum A o A: thread <-> (column, part of the rows); y^2 rides on the last threads
        constexpr int PARTS = 256 / MP, RPP = SM_SLAB / PARTS;
        const int m = tid % MP, part = tid / MP;
        double su = 0.0, sa = 0.0;
#pragma unroll 8
        for (int r = part * RPP; r < (part + 1) * RPP; ++r) {
          const double v = sl.S[r][m];
          su = fma(v, sl.ys[r], su);
          sa = fma(v, v, sa);
        }
        uacc += su;
        a2acc += sa;
        if (tid < SM_SLAB) yyacc = fma(sl.ys[tid], sl.ys[tid], yyacc);
      }
      if (sb == rw) stamp(11);
      // partial A^T A: lower 16 x 16 blocks dealt round-robin to the waves
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        const int b = 4 * i + w;
        if (b < NBL) {
          int bi = 0;
          while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
          const int bj = b - bi * (bi + 1) / 2;
#pragma unroll
          for (int ks = 0; ks < SM_SLAB / 4; ++ks)
            pacc[i] = mfma16(sl.S[4 * ks + l4][16 * bi + l15], sl.S[4 * ks + l4][16 * bj + l15], pacc[i]);
        }
      }
    }
    __syncthreads();
    stamp(12);
    {  // partials of this workgroup -> global
      double* P = a.Ppart + (size_t)rw * MP * MP;
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        const int b = 4 * i + w;
        if (b < NBL) {
          int bi = 0;
          while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
          const int bj = b - bi * (bi + 1) / 2;
#pragma unroll
          for (int r = 0; r < 4; ++r) P[(size_t)(16 * bi + l4 + 4 * r) * MP + 16 * bj + l15] = pacc[i][r];
        }
      }
      {  // the row parts of a column, added in a fixed order
        __syncthreads();
        double* ured = &sl.red[0][0];  // 4 x MP doubles >= 256
        ured[tid] = uacc;
        __syncthreads();
        if (tid < MP) {
          double su = 0.0;
          for (int part = 0; part < 256 / MP; ++part) su += ured[part * MP + tid];
          a.upart[(size_t)rw * MP + tid] = su;
        }
      }
      const double sa = block_sum(a2acc);
      const double sy2 = block_sum(yyacc);
      if (tid == 0) {
        a.spart[2 + 2 * rw] = sa;
        a.spart[3 + 2 * rw] = sy2;
      }
    }
    stamp(3);
    sm_publish_add(sy + SY_PART * SM_SYNC_STRIDE);
    if (!sm_wait_ge(sy + SY_PART * SM_SYNC_STRIDE, ev * a.grow, abortw, &dead)) return;
    stamp(4);
    // ---- slices: B = I + sum_g P_g / s2 (mirrored), u, scalars -- fixed order over g ---------------------------------
    {
      const double is2 = 1.0 / s2;
      // 64 matrix entries per workgroup and pass; the four waves split the partials of an entry (g = wave, wave + 4, ...: with
      // up to 208 row workgroups one thread walking them all was 13 dependent batches of loads on 36 of the workgroups while
      // the others idled) and combine through LDS in a fixed order -- bit-reproducible for a given launch geometry
      if (a.grow <= 32) {
        // few partials (N <= 2048, the reference's own HMC sizes): one thread per entry walks them all -- no barriers
        for (int e = rw * 256 + tid; e < NBL * 256; e += a.grow * 256) {
          const int b = e >> 8, i = (e >> 4) & 15, j = e & 15;
          int bi = 0;
          while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
          const int bj = b - bi * (bi + 1) / 2;
          const int gi = 16 * bi + i, gj = 16 * bj + j;
          if (gi < gj) continue;  // upper half of a diagonal block
          double s = 0.0;
          const double* pp = a.Ppart + (size_t)gi * MP + gj;
          for (int g = 0; g < a.grow; g += 16) {  // sixteen loads in flight; the additions keep the fixed order
            double t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = g + k < a.grow ? pp[(size_t)(g + k) * MP * MP] : 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += t[k];
          }
          const double v = (gi == gj ? 1.0 : 0.0) + s * is2;
          a.Bm[(size_t)gi * MP + gj] = v;
          a.Bm[(size_t)gj * MP + gi] = v;
          a.Lb[(size_t)gi * MP + gj] = v;  // factored in place by the chain (only the lower triangle is read)
          a.Lb[(size_t)gj * MP + gi] = v;
        }
        if (rw == 0) {
          if (tid < MP) {
            double s = 0.0;
            for (int g = 0; g < a.grow; ++g) s += a.upart[(size_t)g * MP + tid];
            a.u[tid] = s;
          }
          if (tid == 0) {
            double sa = 0.0, sy2 = 0.0;
            for (int g = 0; g < a.grow; ++g) {
              sa += a.spart[2 + 2 * g];
              sy2 += a.spart[3 + 2 * g];
            }
            a.spart[0] = sa;
            a.spart[1] = sy2;
          }
        }
      } else {
      // many partials.  Element space: the NBL x 256 entries of the lower blocks of B, then the MP entries of u = sum of the A y
      // partials, then the two scalars (sum A o A, y^T y) -- every one of them a sum over the row workgroups' partials
      const int el = tid & 63;
      constexpr int EB = NBL * 256, EU = EB + MP, ET = EU + 2;
      for (int e0 = rw * 64; e0 < ET; e0 += a.grow * 64) {
        const int e = e0 + el;
        int gi = 0, gj = 0;
        const double* pp = a.Ppart;
        size_t pstride = (size_t)MP * MP;
        bool live = e < ET;
        if (e < EB) {
          const int b = e >> 8, i = (e >> 4) & 15, j = e & 15;
          int bi = 0;
          while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
          const int bj = b - bi * (bi + 1) / 2;
          gi = 16 * bi + i;
          gj = 16 * bj + j;
          live = gi >= gj;  // not the upper half of a diagonal block
          pp = a.Ppart + (size_t)gi * MP + gj;
        } else if (e < EU) {
          pp = a.upart + (e - EB);
          pstride = MP;
        } else {
          pp = a.spart + 2 + (e - EU);
          pstride = 2;
        }
        double s = 0.0;
        if (live) {
          for (int g = w; g < a.grow; g += 64) {  // sixteen loads in flight; the additions keep the fixed order
            double t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = g + 4 * k < a.grow ? pp[(size_t)(g + 4 * k) * pstride] : 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += t[k];
          }
        }
        sl.red[w][el] = s;
        __syncthreads();
        if (w == 0 && live) {
          const double tot = (sl.red[0][el] + sl.red[1][el]) + (sl.red[2][el] + sl.red[3][el]);
          if (e < EB) {
            const double v = (gi == gj ? 1.0 : 0.0) + tot * is2;
            a.Bm[(size_t)gi * MP + gj] = v;
            a.Bm[(size_t)gj * MP + gi] = v;
            a.Lb[(size_t)gi * MP + gj] = v;  // factored in place by the chain (only the lower triangle is read)
            a.Lb[(size_t)gj * MP + gi] = v;
          } else if (e < EU) {
            a.u[e - EB] = tot;
          } else {
            a.spart[e - EU] = tot;
          }
        }
        __syncthreads();
      }
      }
    }
    sm_publish_add(sy + SY_SLICE * SM_SYNC_STRIDE);
    stamp(5);
    if (!a.want_grad) return;

    // ---- reverse: Abar per slab (two solves with LB), Kbar_uf = Abar L^-1 (one solve with L^T), contraction ----------
    if (!sm_wait_ge(sy + SY_LB * SM_SYNC_STRIDE, ev, abortw, &dead)) return;
    stamp(6);
    const double is2 = 1.0 / s2, is22 = is2 * is2;
    for (int sb = rw; sb < a.nslab; sb += a.grow) {
      const int n0 = sb * SM_SLAB, nl = 16 * w + l15;
      __syncthreads();
      sm_stage_factor<MP>(sl, a.Lb, a.dinvB);
      for (int e = tid; e < SM_SLAB * (SM_MAXD + 1); e += 256) {
        const int r = e / (SM_MAXD + 1), j = e - r * (SM_MAXD + 1);
        sl.xs[r][j] = (n0 + r < N && j < d) ? a.X[(size_t)(n0 + r) * a.ldx + j] * hyp.inv_ls[j] : 0.0;
      }
      if (tid < SM_SLAB) sl.ys[tid] = n0 + tid < N ? a.y[n0 + tid] : 0.0;
      __syncthreads();
      if (sb == rw) stamp(13);
      SlabRegs<NB16> Ya, Y;
      {
        const double* arow = a.A + (size_t)(n0 + nl) * MP;
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) Ya.b[pb][s] = arow[16 * pb + 4 * s + l4];
      }
      Y = Ya;
      sm_trsm_fwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
      sm_trsm_bwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);  // V = B^-1 a_n
      // Abar = [(a - B^-1 a) + g (y_n / s2 - g.a_n / s2^2)] / s2 ; Kbar = L^-T Abar = L^-T (a - B^-1 a) / s2 + h c_n with
      // h = L^-T g, c_n = (y_n - g.a_n / s2) / s2^2: the part that needs g joins after the solve
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) Y.b[pb][s] = (Ya.b[pb][s] - Y.b[pb][s]) * is2;
      __syncthreads();
      if (sb == rw) stamp(15);
      sm_stage_factor<MP>(sl, a.Lk, a.dinvK);
      __syncthreads();
      sm_trsm_bwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
      if (sb == rw) {  // g, h from the chain workgroup (normally there long before)
        if (!sm_wait_ge(sy + SY_G * SM_SYNC_STRIDE, ev, abortw, &dead)) return;
        if (tid < MP) {
          sl.gv[tid] = a.g[tid];
          sl.hv[tid] = a.h[tid];
        }
      }
      __syncthreads();
      {
        double tn = 0.0;
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) tn = fma(sl.gv[16 * pb + 4 * s + l4], Ya.b[pb][s], tn);
        tn += __shfl_xor(tn, 16, 64);
        tn += __shfl_xor(tn, 32, 64);  // g . a_n, in the four lanes that share the row
        const double cn = (sl.ys[nl] - tn * is2) * is22;
#pragma unroll
        for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
          for (int s = 0; s < 4; ++s) Y.b[pb][s] = fma(sl.hv[16 * pb + 4 * s + l4], cn, Y.b[pb][s]);  // Kbar_uf rows of this slab
      }
      if (sb == rw) stamp(10);
      contract(Y, N - n0 < SM_SLAB ? N - n0 : SM_SLAB, 1.0);
    }
    stamp(7);
    if (tid < SM_GP) a.gpart[(size_t)(rw + NB64) * SM_GP + tid] = sl.acc[tid];
    sm_publish_add(sy + SY_GRAD * SM_SYNC_STRIDE);
    if (a.want_gz && a.gZ && ngrad > SM_GZ_SPREAD) {
      // dF/dZ = sum over ALL workgroups' partials, M d entries: with ~210 partials the chain workgroup alone needed 185 us for it
      // (3 MB through one CU).  The row workgroups share the entries instead -- entry e by workgroup e mod grow, its 256 threads
      // striding the partials, one fixed-order block sum each -- once every partial is in
      if (!sm_wait_ge(sy + SY_GRAD * SM_SYNC_STRIDE, ev * ngrad, abortw, &dead)) return;
      for (int e = rw; e < M * d; e += a.grow) {
        double sv = 0.0;
        for (int g = tid; g < ngrad; g += 256) sv += a.gzpart[(size_t)g * MP * SM_MAXD + e];
        sv = block_sum(sv);
        if (tid == 0) a.gZ[e] = sv;
      }
      sm_publish_add(sy + SY_GZ * SM_SYNC_STRIDE);
    }
    return;
  }

  // =================================================================================================================
  // role 1: Kbar_uu = -1/2 L^-T S L^-1, S = B + B^-1 - 2 I + g g^T / s2^2 ; its contraction with dK_uu ; tr B^-1
  if (NB64 > 1 && wg == 1) {
    // first, while the chain workgroup factors tile (0, 0): rows 64.. of the padded Kuu (tiles (1, 0) and (1, 1)) -> a.Lk
    for (int e0 = tid; e0 < 64 * 8; e0 += 256) {  // (row, group of 16 columns)
      const int i = 64 + (e0 >> 3), j0 = 16 * (e0 & 7);
      double v[16];
      if (comp) {
        sm_comp_value_vec<16>(hyp.cs, sl.zs[i], [&](int k, int q) { return sl.zs[j0 + k][q]; }, d, v);
      } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = 0.0;
      for (int q = 0; q < d; ++q) {
        const double zi = sl.zs[i][q];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const double dz = zi - sl.zs[j0 + k][q];
          v[k] = fma(dz, dz, v[k]);
        }
      }
      sm_profile_vec<16>(a.kid, v, sf2);
      }
      double* dst = a.Lk + (size_t)i * MP + j0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int j = j0 + k;
        double val = v[k];
        if (i >= M || j >= M) val = i == j ? 1.0 : 0.0;
        else if (i == j) val += a.jitter;
        dst[k] = val;
      }
    }
    sm_publish_set(sy + SY_KUU * SM_SYNC_STRIDE, ev);
  }
  if (!a.want_grad) return;
  if (!sm_wait_ge(sy + SY_LB * SM_SYNC_STRIDE, ev, abortw, &dead)) return;
  stamp(1);
  const double is22 = 1.0 / (s2 * s2);
  double trb = 0.0;
  const int v = wg - 1;  // this workgroup's 64 columns of I
  {  // Q[:, cols] = L^-T S[:, cols]
    const int col = 64 * v + 16 * w + l15;
    __syncthreads();
    sm_stage_factor<MP>(sl, a.Lb, a.dinvB);
    __syncthreads();
    SlabRegs<NB16> Y;
#pragma unroll
    for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
      for (int s = 0; s < 4; ++s) Y.b[pb][s] = (16 * pb + 4 * s + l4) == col ? 1.0 : 0.0;
    sm_trsm_fwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
    sm_trsm_bwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);  // columns of B^-1
#pragma unroll
    for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int m = 16 * pb + 4 * s + l4;
        if (m == col) trb += Y.b[pb][s];
        Y.b[pb][s] += a.Bm[(size_t)m * MP + col] - (m == col ? 2.0 : 0.0);  // columns of B + B^-1 - 2 I
      }
    __syncthreads();
    sm_stage_factor<MP>(sl, a.Lk, a.dinvK);
    __syncthreads();
    sm_trsm_bwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
    // the rank-one part of S joins after the solve: L^-T (g g^T / s2^2) = h g^T / s2^2 (g, h from the chain workgroup, which
    // computes them while these solves run)
    if (!sm_wait_ge(sy + SY_G * SM_SYNC_STRIDE, ev, abortw, &dead)) return;
    if (tid < MP) {
      sl.gv[tid] = a.g[tid];
      sl.hv[tid] = a.h[tid];
    }
    __syncthreads();
    if (v == 0) {  // the two scalars of dF/ds2 that involve g
      const double ug = block_sum(tid < MP ? a.u[tid] * sl.gv[tid] : 0.0);
      const double gg = block_sum(tid < MP ? sl.gv[tid] * sl.gv[tid] : 0.0);
      if (tid == 0) {
        sl.acc[SM_XTRA + 1] = ug;
        sl.acc[SM_XTRA + 2] = gg;
      }
    }
    const double gc = sl.gv[col] * is22;
#pragma unroll
    for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
      for (int s = 0; s < 4; ++s) a.Qm[(size_t)(16 * pb + 4 * s + l4) * MP + col] = fma(sl.hv[16 * pb + 4 * s + l4], gc, Y.b[pb][s]);
  }
  trb = block_sum(trb);
  if (tid == 0) sl.acc[SM_XTRA] = trb;
  stamp(2);
  if (NB64 > 1) {  // the other workgroups' columns of Q
    sm_publish_add(sy + SY_Q * SM_SYNC_STRIDE);
    if (!sm_wait_ge(sy + SY_Q * SM_SYNC_STRIDE, ev * NB64, abortw, &dead)) return;
  } else {
    __threadfence();
    __syncthreads();
  }
  {  // the factor of L stays staged: R[:, j] = L^-T (row j of Q)^T
    const int col = 64 * v + 16 * w + l15;
    SlabRegs<NB16> Y;
    {
      const double* qrow = a.Qm + (size_t)col * MP;
#pragma unroll
      for (int pb = 0; pb < NB16; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) Y.b[pb][s] = qrow[16 * pb + 4 * s + l4];
    }
    sm_trsm_bwd<NB16>(Y, sl.Lblk, sl.Dinv, l15, l4);
#pragma unroll
    for (int pb = 0; pb < NB16; ++pb) Y.b[pb] = Y.b[pb] * -0.5;  // Kbar_uu[m][col]
    // "data rows" of this slab are the inducing inputs col: scaled copies into xs
    __syncthreads();
    for (int e = tid; e < SM_SLAB * (SM_MAXD + 1); e += 256) {
      const int r = e / (SM_MAXD + 1), j = e - r * (SM_MAXD + 1);
      sl.xs[r][j] = sl.zs[64 * v + r][j];
    }
    __syncthreads();
    contract(Y, M - 64 * v < SM_SLAB ? (M - 64 * v > 0 ? M - 64 * v : 0) : SM_SLAB, 2.0);
  }
  stamp(3);
  if (tid < SM_GP) a.gpart[(size_t)v * SM_GP + tid] = sl.acc[tid];
  sm_publish_add(sy + SY_GRAD * SM_SYNC_STRIDE);
}

template <int MP, bool COMP>
__global__ __launch_bounds__(256) void small_eval_kernel(SmallArgs a) {
  __shared__ SmKernelShared<MP> ks;
  if (threadIdx.x == 0) ks.dead = 0;
  sm_hypers<COMP>(a, ks.hyp);
  if (!ks.hyp.ok) {  // theta outside the representable range: density zero, never an exception (PyMC3: non-finite logp)
    if (blockIdx.x == 0 && threadIdx.x <= a.nh + 1) a.out[threadIdx.x] = threadIdx.x == 0 ? -INFINITY : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.info = 0;
    return;
  }
  sm_eval_body<MP, COMP>(a, ks, 1, false);
}

// The evaluation as an out-of-line call for the persistent kernel: inlined into its request loop the compiler kept the
// loop's state live across the whole evaluation (455 spilled VGPRs at MP = 128 against 13 in the single-evaluation kernel,
// +45 us per leapfrog).
template <int MP, bool COMP>
__device__ __attribute__((noinline)) void sm_eval_call(const SmallArgs* a, SmKernelShared<MP>* ks, int ev) {
  sm_eval_body<MP, COMP>(*a, *ks, ev, true);
}

// ---- device-resident NUTS ------------------------------------------------------------------------------------------
// One PERSISTENT launch runs the whole of pm.sample(n, tune=tune, chains=1) (models/bayesian_sgpr_hmc.py:73-78): thread 0
// of workgroup 0 advances the sampler's state machine (sgp_nuts.hpp, state in LDS), writes the next position into
// a.theta and raises the request word; every workgroup then runs one evaluation of the NUTS target (sm_eval_body, mode
// SGP_SMALL_HMC) and workgroup 0 feeds (logp, gradient) back.  No host round trip, no launch per leapfrog.
struct NutsArgs {
  const double* q0;   // start (unconstrained), ndim doubles
  double* theta_w;    // = a.theta, writable
  double* samples;    // n_draws x ndim
  double* stats;      // n_draws x SGP_NUTS_STAT_COLS
  long long* counters;  // [0] leapfrogs (evaluations), [1] draws finished
  int n_tune, n_draws, max_treedepth;
  double step_scale, target_accept;
  unsigned long long seed;
  // batch mode (S > 0): no sampler -- evaluate S given hyper-parameter sets one after the other in the same launch
  int S;
  const double* batch_theta;  // S x (d + 2)
  double* batch_out;          // S x (d + 5)
  double* batch_gz;           // S x M x d, or null
  int* batch_info;            // S
};
constexpr int SM_NUTS_COLS = NST_COLS + 1;  // + seconds per draw (device clock)

template <int MP, bool COMP>
__global__ __launch_bounds__(256) void small_nuts_kernel(SmallArgs a, NutsArgs na) {
  __shared__ SmKernelShared<MP> ks;
  __shared__ NutsState st;
  __shared__ int cmd;
  const int tid = threadIdx.x, wg = blockIdx.x, ndim = a.ndim;
  int* sy = a.sync;
  int* abortw = sy + SY_ABORT * SM_SYNC_STRIDE;
  if (tid == 0) ks.dead = 0;
  int ev = 0, req = 0;
  double lp = 0.0;
  unsigned long long t_draw = 0;
  const bool batch = na.S > 0;
  if (wg == 0 && tid == 0 && !batch) {
    nuts_init(st, ndim, na.n_tune, na.n_draws, na.max_treedepth, na.step_scale, na.target_accept, na.seed, na.q0);
    t_draw = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  SmallArgs al = a;  // batch mode: the outputs of evaluation s go to slot s
  for (;;) {
    ++req;
    if (batch) {
      const int sidx = req - 1 < na.S ? req - 1 : na.S - 1;
      al.out = na.batch_out + (size_t)sidx * (a.nh + 4);
      al.gZ = na.batch_gz ? na.batch_gz + (size_t)sidx * a.M * a.d : nullptr;
      al.info = na.batch_info + sidx;
    }
    if (wg == 0) {
      if (tid == 0) {
        // theta of request req - 1 may only be overwritten once every other workgroup has read it: on the skip path below
        // (theta out of range) and in value-only batches nothing else makes this workgroup wait for the others
        const int acks = (req - 1) * ((int)gridDim.x - 1);
        if (batch) {
          cmd = req <= na.S ? NUTS_EVAL : NUTS_DONE;
          if (!sm_wait_ack(sy + SY_ACK * SM_SYNC_STRIDE, acks, abortw)) ks.dead = 1;  // also before DONE: the word is reset below
          if (req <= na.S)
            for (int i = 0; i < ndim; ++i) na.theta_w[i] = na.batch_theta[(size_t)(req - 1) * ndim + i];
        } else {
          const double* qn = nullptr;
          const int it0 = st.it;
          const int c = nuts_step(st, lp, a.out + 1, &qn, na.samples, na.stats /* NST_COLS columns, re-packed at the end */);
          if (st.it != it0) {  // a draw has finished: its device time
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            const int row = it0 - na.n_tune;
            if (row >= 0) na.stats[(size_t)na.n_draws * NST_COLS + row] = (double)(now - t_draw) * 1e-8;
            t_draw = now;
          }
          if (!sm_wait_ack(sy + SY_ACK * SM_SYNC_STRIDE, acks, abortw)) ks.dead = 1;  // also before DONE: the word is reset below
          if (c == NUTS_EVAL)
            for (int i = 0; i < ndim; ++i) na.theta_w[i] = qn[i];
          cmd = c;
        }
      }
      __syncthreads();
      if (cmd != NUTS_EVAL) {
        if (tid == 0 && !batch) {
          na.counters[0] = st.n_leapfrog;
          na.counters[1] = st.it;
        }
        sm_publish_set(sy + SY_DONE * SM_SYNC_STRIDE, 1);
        sm_publish_set(sy + SY_REQ * SM_SYNC_STRIDE, req);
        break;
      }
      sm_publish_set(sy + SY_REQ * SM_SYNC_STRIDE, req);
      __syncthreads();
    } else {
      if (!sm_wait_ge(sy + SY_REQ * SM_SYNC_STRIDE, req, abortw, &ks.dead)) break;
      if (sm_ld(sy + SY_DONE * SM_SYNC_STRIDE) != 0) break;
    }
    sm_hypers<COMP>(al, ks.hyp);  // every workgroup from the same theta: the same decision everywhere
    if (wg != 0 && tid == 0)     // theta has been read (its values are in LDS): the chain workgroup may write the next one
      __hip_atomic_fetch_add(sy + SY_ACK * SM_SYNC_STRIDE, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (ks.dead) {
      if (wg == 0 && tid == 0) *al.info = SGP_INFO_TIMEOUT;
      break;
    }
    if (!ks.hyp.ok) {
      if (wg == 0 && tid == 0) {
        al.out[0] = -INFINITY;
        *al.info = 0;
        lp = -INFINITY;
      }
      continue;
    }
    ++ev;
    sm_eval_call<MP, COMP>(&al, &ks, ev);
    if (ks.dead || sm_ld(abortw) != 0) {
      if (wg == 0 && tid == 0) *al.info = SGP_INFO_TIMEOUT;
      break;
    }
    if (wg == 0) {
      __syncthreads();
      if (tid == 0) lp = (*al.info == 0) ? al.out[0] : -INFINITY;  // a failed factorization is a divergence, not an error
    }
  }
  if (wg == 0) {
    __syncthreads();
    for (int e = tid; e < SY_WORDS * SM_SYNC_STRIDE; e += 256)
      if (e / SM_SYNC_STRIDE != SY_REQ && e / SM_SYNC_STRIDE != SY_DONE) sy[e] = 0;
  }
}

struct SmallWs {
  SmallArgs a;
  size_t bytes;
};

// The workgroups of these launches wait for each other, so ALL of them have to be resident at once.  The grid (<= 211) is
// checked against what the device can hold -- occupancy of this very kernel x the CUs the calling thread's launches can use
// (device count, or sgp_set_cu_budget() on a CU-masked stream) -- and refused with SGP_ERR_LAUNCH otherwise, instead of
// spinning into SGP_INFO_TIMEOUT.  (hipLaunchCooperativeKernel would make the runtime do the same check, at the price of its
// cooperative-queue hand-over on every 70 us evaluation.)
template <auto K>
static bool small_grid_is_resident(int grid) {
  static int per_cu = -1;
  if (per_cu < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K, 256, 0) != hipSuccess || n < 1) n = 1;
    per_cu = n;
  }
  return (long long)per_cu * available_cus() >= grid;
}
static int small_grid(int M, int grow) { return 1 + (M <= 64 ? 1 : 2) + grow; }
// Row workgroups of a launch over nslab 64-row slabs: as many as there are slabs, CUs (one workgroup per CU, minus the chain and
// the K_uu-adjoint workgroups) and SM_MAX_ROWWG allow -- then evened out, so that no workgroup walks a slab more than the others
// (65 slabs on 64 workgroups would cost two rounds for one slab: 33 workgroups x 2 slabs instead).
static int small_row_workgroups(int nslab, int NB64) {
  int cap = available_cus() - 1 - NB64;
  if (cap > SM_MAX_ROWWG) cap = SM_MAX_ROWWG;
  if (cap < 1) cap = 1;
  if (nslab <= cap) return nslab;
  const int rounds = (nslab + cap - 1) / cap;
  return (nslab + rounds - 1) / rounds;
}
static SmallWs carve_small(void* ws, int64_t N, int M, int d) {
  (void)d;
  const int MP = M <= 64 ? 64 : 128, NB64 = MP / 64;
  const int nslab = (int)((N + SM_SLAB - 1) / SM_SLAB) > 0 ? (int)((N + SM_SLAB - 1) / SM_SLAB) : 1;
  const int grow_cap = nslab < SM_MAX_ROWWG ? nslab : SM_MAX_ROWWG;  // buffers are sized for the cap, whatever the CU budget is
  const int grow = small_row_workgroups(nslab, NB64);
  const size_t mm = (size_t)MP * MP;
  Carver c(ws);
  SmallWs w{};
  w.a.sync = c.take<int>(SY_WORDS * SM_SYNC_STRIDE);  // first: the caller zeroes the head of the workspace once
  w.a.Lk = c.take<double>(mm);
  w.a.dinvK = c.take<double>((size_t)(MP / 16) * 256);
  w.a.Bm = c.take<double>(mm);
  w.a.Lb = c.take<double>(mm);
  w.a.dinvB = c.take<double>((size_t)(MP / 16) * 256);
  w.a.A = c.take<double>((size_t)nslab * SM_SLAB * MP);
  w.a.Ppart = c.take<double>((size_t)grow_cap * mm);
  w.a.upart = c.take<double>((size_t)grow_cap * MP);
  w.a.spart = c.take<double>(2 + 2 * (size_t)grow_cap);
  w.a.u = c.take<double>(MP);
  w.a.c0 = c.take<double>(MP);
  w.a.g = c.take<double>(MP);
  w.a.h = c.take<double>(MP);
  w.a.gpart = c.take<double>((size_t)(grow_cap + NB64) * SM_GP);
  w.a.gzpart = c.take<double>((size_t)(grow_cap + NB64) * MP * SM_MAXD);
  w.a.Qm = c.take<double>(mm);
  w.a.nslab = nslab;
  w.a.grow = grow;
  w.bytes = c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

static unsigned long long* g_small_stamps = nullptr;
// measurement aid (tools/small_eval_phases.py): a device buffer of (3 + 208) x 16 uint64 that the next launches fill with
// s_memrealtime stamps at their phase boundaries; NULL switches it off
extern "C" void sgp_small_debug_stamps(void* dev_buffer) { g_small_stamps = static_cast<unsigned long long*>(dev_buffer); }

extern "C" int sgp_small_supported(int64_t N, int M, int d, int kernel_id) {
  if (!(N >= 1 && N <= (int64_t)1 << 22 && M >= 1 && M <= 128 && d >= 1)) return 0;
  // every workgroup of the launch must be resident (they wait for each other): at least one per available CU is assumed
  // here, the launch itself checks the kernel's real occupancy.  Callers fall back to the multi-launch path on 0.
  if (available_cus() < 4) return 0;  // chain + K_uu-adjoint workgroups + at least one row workgroup (the grid adapts to the budget)
  if (kernel_id == SGP_KERNEL_COMPOSITE) return d <= COMP_MAX_DIM;
  return d <= SM_MAXD && kernel_id >= SGP_KERNEL_RBF && kernel_id <= SGP_KERNEL_MATERN52;
}
extern "C" size_t sgp_small_workspace_bytes(int64_t N, int M, int d) {
  if (!sgp_small_supported(N, M, d, SGP_KERNEL_RBF)) return 0;
  return carve_small(nullptr, N, M, d).bytes;
}
extern "C" size_t sgp_small_sync_bytes(void) { return (size_t)SY_WORDS * SM_SYNC_STRIDE * sizeof(int); }

// the composite description an entry point hands over (host memory): the parameter block as the structure (term / factor
// types and the values of parameters that are not sampled) and, for the NUTS target, the free-parameter table
struct CompHost {
  const double* structure;
  int ncp;
  const int* slot;
  const int* role;
  const double* sd;
};
// fills the kernel-specific fields of `a` (nh, ndim, cs, cp_*) ; SGP_OK or an error code
static int small_kernel_fields(SmallArgs& a, int d, int kernel_id, int mode, const CompHost* ch) {
  a.ncp = 0;
  if (kernel_id != SGP_KERNEL_COMPOSITE) {
    if (ch) return SGP_ERR_ARG;
    a.nh = d + 1;
    a.ndim = d + 2;
    return SGP_OK;
  }
  if (!ch || !ch->structure) return SGP_ERR_ARG;
  if (comp_parse(ch->structure, d, &a.cs) != SGP_OK) return SGP_ERR_ARG;
  if (mode == 0) {
    a.nh = SGP_COMP_LEN;
    a.ndim = SGP_COMP_LEN + 1;
    return SGP_OK;
  }
  if (ch->ncp < 1 || ch->ncp > SM_MAXCP || !ch->slot || !ch->role || !ch->sd) return SGP_ERR_ARG;
  for (int k = 0; k < ch->ncp; ++k) {
    const int slot = ch->slot[k], role = ch->role[k];
    if (slot < 1 || slot >= SGP_COMP_LEN || role < 0 || role > 2 || !(ch->sd[k] > 0.0)) return SGP_ERR_ARG;
    const int t = (slot - 1) >> 3, rem = (slot - 1) & 7;
    if (t >= a.cs.nterms) return SGP_ERR_ARG;
    // the slot must be of the stated role and belong to a factor the structure has
    const bool is_amp = rem == 0, is_ls = rem == 3 || rem == 6, is_aux = rem == 4 || rem == 7;
    if ((role == 0) != is_amp || (role == 1) != is_ls || (role == 2) != is_aux) return SGP_ERR_ARG;
    if (!is_amp && (rem - 2) / 3 >= a.cs.nfac[t]) return SGP_ERR_ARG;
    a.cp_slot[k] = slot;
    a.cp_role[k] = role;
    a.cp_sd[k] = ch->sd[k];
  }
  a.ncp = ch->ncp;
  a.nh = ch->ncp;
  a.ndim = ch->ncp + 1;
  return SGP_OK;
}

static int small_eval_impl(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* theta,
                           int64_t N, int M, int d, int kernel_id, const CompHost* ch, double jitter, int mode, int want_grad,
                           double* out, double* g_Z, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!X || !y || !Z || !theta || !out || !info || ldx < d || ldz < d || (mode != 0 && mode != 1)) return SGP_ERR_ARG;
  if (!sgp_small_supported(N, M, d, kernel_id)) return SGP_ERR_DIM;
  if (kernel_id == SGP_KERNEL_COMPOSITE && g_Z) return SGP_ERR_ARG;  // dF/dZ of a composite kernel: the materialised path
  SmallWs w = carve_small(ws, N, M, d);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  SmallArgs& a = w.a;
  if (int rc = small_kernel_fields(a, d, kernel_id, mode, ch)) return rc;
  a.X = X; a.ldx = ldx; a.y = y; a.Z = Z; a.ldz = ldz; a.theta = theta;
  a.N = (int)N; a.M = M; a.d = d; a.kid = kernel_id; a.mode = mode; a.want_grad = want_grad ? 1 : 0;
  a.want_gz = (want_grad && g_Z) ? 1 : 0;
  a.jitter = jitter;
  a.info = info; a.out = out; a.gZ = g_Z;
  a.stamps = g_small_stamps;
  const int grid = small_grid(M, a.grow);
  hipStream_t st = (hipStream_t)stream;
  const bool comp = kernel_id == SGP_KERNEL_COMPOSITE;
#define SGP_SMALL_LAUNCH(KERNEL, ...)                                  \
  do {                                                                 \
    if (!small_grid_is_resident<KERNEL>(grid)) return SGP_ERR_LAUNCH;  \
    KERNEL<<<grid, 256, 0, st>>>(__VA_ARGS__);                         \
  } while (0)
  if (M <= 64) {
    if (comp) SGP_SMALL_LAUNCH((small_eval_kernel<64, true>), a);
    else SGP_SMALL_LAUNCH((small_eval_kernel<64, false>), a);
  } else {
    if (comp) SGP_SMALL_LAUNCH((small_eval_kernel<128, true>), a);
    else SGP_SMALL_LAUNCH((small_eval_kernel<128, false>), a);
  }
  return check_launch();
}

extern "C" int sgp_small_eval(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                              const double* theta, int64_t N, int M, int d, int kernel_id, double jitter, int mode,
                              int want_grad, double* out, double* g_Z, int* info, void* ws, size_t ws_bytes,
                              sgp_stream_t stream) {
  if (kernel_id == SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;  // sgp_small_eval_composite carries the structure
  return small_eval_impl(X, ldx, y, Z, ldz, theta, N, M, d, kernel_id, nullptr, jitter, mode, want_grad, out, g_Z, info, ws,
                         ws_bytes, stream);
}

extern "C" int sgp_small_eval_composite(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                        const double* theta, const double* structure, int n_free, const int* free_slot,
                                        const int* free_role, const double* free_prior_sd, int64_t N, int M, int d,
                                        double jitter, int mode, int want_grad, double* out, int* info, void* ws,
                                        size_t ws_bytes, sgp_stream_t stream) {
  const CompHost ch{structure, n_free, free_slot, free_role, free_prior_sd};
  return small_eval_impl(X, ldx, y, Z, ldz, theta, N, M, d, SGP_KERNEL_COMPOSITE, &ch, jitter, mode, want_grad, out, nullptr,
                         info, ws, ws_bytes, stream);
}

extern "C" size_t sgp_small_nuts_stat_cols(void) { return SM_NUTS_COLS; }

static int small_nuts_impl(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* q0,
                           int64_t N, int M, int d, int kernel_id, const CompHost* ch, double jitter, int n_tune, int n_draws,
                           int max_treedepth, double step_scale, double target_accept, uint64_t seed, double* theta_scratch,
                           double* samples, double* stats, long long* counters, double* out, int* info, void* ws,
                           size_t ws_bytes, sgp_stream_t stream) {
  if (!X || !y || !Z || !q0 || !theta_scratch || !samples || !stats || !counters || !out || !info || ldx < d || ldz < d)
    return SGP_ERR_ARG;
  if (n_tune < 0 || n_draws < 1 || max_treedepth < 1 || max_treedepth >= NUTS_MAXDEPTH || !(step_scale > 0.0)) return SGP_ERR_ARG;
  if (!sgp_small_supported(N, M, d, kernel_id)) return SGP_ERR_DIM;
  SmallWs w = carve_small(ws, N, M, d);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  SmallArgs& a = w.a;
  if (int rc = small_kernel_fields(a, d, kernel_id, SGP_SMALL_HMC, ch)) return rc;
  if (a.ndim > NUTS_MAXD) return SGP_ERR_DIM;
  {
    // the sync counters are cumulative ints (evaluations x contributing workgroups): a run whose worst case -- every tree at its
    // depth limit -- could pass 2^31 is refused up front (2 000 draws at depth 10 on 211 workgroups is 4.3e8)
    const double worst = ((double)n_tune + (double)n_draws + 2.0) * (double)(1 << max_treedepth) * (double)small_grid(M, a.grow);
    if (worst > 2.0e9) return SGP_ERR_DIM;
  }
  a.X = X; a.ldx = ldx; a.y = y; a.Z = Z; a.ldz = ldz; a.theta = theta_scratch;
  a.N = (int)N; a.M = M; a.d = d; a.kid = kernel_id; a.mode = SGP_SMALL_HMC; a.want_grad = 1; a.want_gz = 0;
  a.jitter = jitter;
  a.info = info; a.out = out; a.gZ = nullptr;
  a.stamps = nullptr;
  NutsArgs na{q0, theta_scratch, samples, stats, counters, n_tune, n_draws, max_treedepth, step_scale, target_accept, seed,
              0, nullptr, nullptr, nullptr, nullptr};
  const int grid = small_grid(M, a.grow);
  hipStream_t st = (hipStream_t)stream;
  const bool comp = kernel_id == SGP_KERNEL_COMPOSITE;
  if (!(M <= 64 ? (comp ? small_grid_is_resident<small_nuts_kernel<64, true>>(grid) : small_grid_is_resident<small_nuts_kernel<64, false>>(grid))
                : (comp ? small_grid_is_resident<small_nuts_kernel<128, true>>(grid) : small_grid_is_resident<small_nuts_kernel<128, false>>(grid))))
    return SGP_ERR_LAUNCH;
  // the request / done words of the previous run (the only sync words a run leaves non-zero)
  zero_ints(a.sync + SY_REQ * SM_SYNC_STRIDE, 2 * SM_SYNC_STRIDE, st);
  if (M <= 64) {
    if (comp) small_nuts_kernel<64, true><<<grid, 256, 0, st>>>(a, na);
    else small_nuts_kernel<64, false><<<grid, 256, 0, st>>>(a, na);
  } else {
    if (comp) small_nuts_kernel<128, true><<<grid, 256, 0, st>>>(a, na);
    else small_nuts_kernel<128, false><<<grid, 256, 0, st>>>(a, na);
  }
  return check_launch();
}

extern "C" int sgp_small_nuts(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* q0,
                              int64_t N, int M, int d, int kernel_id, double jitter, int n_tune, int n_draws, int max_treedepth,
                              double step_scale, double target_accept, uint64_t seed, double* theta_scratch, double* samples,
                              double* stats, long long* counters, double* out, int* info, void* ws, size_t ws_bytes,
                              sgp_stream_t stream) {
  if (kernel_id == SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;  // sgp_small_nuts_composite carries the structure
  return small_nuts_impl(X, ldx, y, Z, ldz, q0, N, M, d, kernel_id, nullptr, jitter, n_tune, n_draws, max_treedepth, step_scale,
                         target_accept, seed, theta_scratch, samples, stats, counters, out, info, ws, ws_bytes, stream);
}

extern "C" int sgp_small_nuts_composite(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                        const double* q0, const double* structure, int n_free, const int* free_slot,
                                        const int* free_role, const double* free_prior_sd, int64_t N, int M, int d,
                                        double jitter, int n_tune, int n_draws, int max_treedepth, double step_scale,
                                        double target_accept, uint64_t seed, double* theta_scratch, double* samples,
                                        double* stats, long long* counters, double* out, int* info, void* ws, size_t ws_bytes,
                                        sgp_stream_t stream) {
  const CompHost ch{structure, n_free, free_slot, free_role, free_prior_sd};
  return small_nuts_impl(X, ldx, y, Z, ldz, q0, N, M, d, SGP_KERNEL_COMPOSITE, &ch, jitter, n_tune, n_draws, max_treedepth,
                         step_scale, target_accept, seed, theta_scratch, samples, stats, counters, out, info, ws, ws_bytes, stream);
}

// S evaluations (S hyper-parameter sets, the same X, y, Z) in ONE launch: the theta-averaged loss of the reference's
// alternating schedule (models/bayesian_sgpr_hmc.py:121-134: for every sample of the current trace, set the hypers,
// evaluate the bound, average, back-propagate to Z only).  thetas (S x (d + 2)), outs (S x (d + 5)), g_Z (S x M x d, may be
// NULL) and infos (S) are device arrays laid out per sample as in sgp_small_eval.
extern "C" int sgp_small_eval_batch(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                    const double* thetas, int S, int64_t N, int M, int d, int kernel_id, double jitter, int mode,
                                    int want_grad, double* theta_scratch, double* outs, double* g_Z, int* infos, void* ws,
                                    size_t ws_bytes, sgp_stream_t stream) {
  if (!X || !y || !Z || !thetas || !theta_scratch || !outs || !infos || S < 1 || ldx < d || ldz < d || (mode != 0 && mode != 1))
    return SGP_ERR_ARG;
  if (kernel_id == SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (!sgp_small_supported(N, M, d, kernel_id)) return SGP_ERR_DIM;
  SmallWs w = carve_small(ws, N, M, d);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  SmallArgs& a = w.a;
  if ((double)S * (double)small_grid(M, a.grow) > 2.0e9) return SGP_ERR_DIM;  // cumulative int sync counters
  if (int rc = small_kernel_fields(a, d, kernel_id, mode, nullptr)) return rc;
  a.X = X; a.ldx = ldx; a.y = y; a.Z = Z; a.ldz = ldz; a.theta = theta_scratch;
  a.N = (int)N; a.M = M; a.d = d; a.kid = kernel_id; a.mode = mode; a.want_grad = want_grad ? 1 : 0;
  a.want_gz = (want_grad && g_Z) ? 1 : 0;
  a.jitter = jitter;
  a.info = infos; a.out = outs; a.gZ = g_Z;
  a.stamps = nullptr;
  NutsArgs na{thetas, theta_scratch, nullptr, nullptr, nullptr, 0, 0, 1, 0.25, 0.8, 0, S, thetas, outs, g_Z, infos};
  const int grid = small_grid(M, a.grow);
  hipStream_t st = (hipStream_t)stream;
  if (!(M <= 64 ? small_grid_is_resident<small_nuts_kernel<64, false>>(grid) : small_grid_is_resident<small_nuts_kernel<128, false>>(grid)))
    return SGP_ERR_LAUNCH;
  zero_ints(a.sync + SY_REQ * SM_SYNC_STRIDE, 2 * SM_SYNC_STRIDE, st);
  if (M <= 64) small_nuts_kernel<64, false><<<grid, 256, 0, st>>>(a, na);
  else small_nuts_kernel<128, false><<<grid, 256, 0, st>>>(a, na);
  return check_launch();
}
