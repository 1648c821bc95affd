// sgp_ctx_*: the context half of the C ABI (include/sgp.h).  See sgp_ctx.hpp.
#include "sgp_common.hpp"
#include "sgp_ctx.hpp"
#include "sgp_stream.hpp"
#include <cstdlib>
#include <new>

namespace sgp {

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

static void ctx_init(Ctx& c, int device) {
  c.device = device;
  c.contraction = env_int("SGP_CONTRACTION", 1);
  if (c.contraction < 0 || c.contraction > 2) c.contraction = 1;
  c.asm_overlap = env_int("SGP_ASM_OVERLAP", 0);
  c.kfu_budget = KFU_BUDGET_DEFAULT;
  c.syrk_skip_upper = env_int("SGP_SYRK_SKIP_UPPER", 1);
  c.syrk_waves = env_int("SGP_SYRK_WAVES", 4);
  c.syrk_glds = env_int("SGP_SYRK_GLDS", 0);
  c.i8_prio = env_int("SGP_I8_PRIO", 0);
  c.shared_device = env_int("SGP_SHARED_DEVICE", 0) != 0 ? 1 : 0;
}

Ctx& default_ctx() {
  static Ctx* c = [] {
    Ctx* p = new Ctx();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;  // (no device: the CPU-only build / symbol checks)
    ctx_init(*p, dev);
    return p;
  }();
  return *c;
}

static thread_local Ctx* t_cur = nullptr;    // installed by a sgp_ctx_* entry point for its own duration (CtxScope)
static thread_local Ctx* t_bound = nullptr;  // sgp_ctx_bind_thread: what the context-free entry points of this thread run in
Ctx& cur_ctx() { return t_cur ? *t_cur : (t_bound ? *t_bound : default_ctx()); }
void bind_thread_ctx(void* ctx) { t_bound = static_cast<Ctx*>(ctx); }
CtxScope::CtxScope(void* ctx) : prev(t_cur) { t_cur = ctx ? static_cast<Ctx*>(ctx) : &default_ctx(); }
CtxScope::~CtxScope() { t_cur = prev; }

}  // namespace sgp

using namespace sgp;

// A context belongs to the device it was created for (its side stream, events and timing slots live there): an entry point called
// with another device current is refused instead of launching on the wrong one (the field was stored and never looked at).
static bool ctx_on_current_device(const sgp_ctx* ctx) {
  if (!ctx) return true;  // the default context follows the caller
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return true;  // no device at all: the argument checks and the launch report that
  return dev == reinterpret_cast<const Ctx*>(ctx)->device;
}

extern "C" sgp_ctx* sgp_ctx_create(int device) {
  if (device < 0) return nullptr;
  Ctx* c = new (std::nothrow) Ctx();
  if (!c) return nullptr;
  ctx_init(*c, device);
  return reinterpret_cast<sgp_ctx*>(c);
}

extern "C" void sgp_ctx_destroy(sgp_ctx* ctx) {
  if (!ctx) return;
  Ctx* c = reinterpret_cast<Ctx*>(ctx);
  if (c == &default_ctx()) return;
  if (c->ev_ready)
    for (int s = 0; s < CTX_TIMING_SLOTS; ++s)
      for (int k = 0; k < 2; ++k) (void)hipEventDestroy(c->ev[s][k]);
  if (c->side) {
    (void)hipStreamDestroy(c->side);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
  }
  delete c;
}

extern "C" int sgp_ctx_set_option(sgp_ctx* ctx, int option, double value) {
  Ctx& c = ctx ? *reinterpret_cast<Ctx*>(ctx) : default_ctx();
  switch (option) {
    case SGP_OPT_CONTRACTION:
      if (!(value >= 0.0 && value <= 2.0)) return SGP_ERR_ARG;
      c.contraction = (int)value;
      return SGP_OK;
    case SGP_OPT_ASM_OVERLAP:
      if (!(value >= 0.0 && value <= 2.0)) return SGP_ERR_ARG;
      c.asm_overlap = (int)value;
      return SGP_OK;
    case SGP_OPT_KFU_BUDGET_BYTES:
      if (!(value >= 0.0)) return SGP_ERR_ARG;
      c.kfu_budget = value > 0.0 ? (size_t)value : KFU_BUDGET_DEFAULT;
      return SGP_OK;
    case SGP_OPT_COND_LIMIT:
      c.cond_limit = value >= 0.0 ? value : 1e13;
      return SGP_OK;
    case SGP_OPT_CU_BUDGET:
      c.cu_budget = value > 0.0 ? (int)value : 0;
      return SGP_OK;
    case SGP_OPT_TIMING:
      c.timing = value != 0.0 ? 1 : 0;
      return SGP_OK;
    case SGP_OPT_SHARED_DEVICE:
      c.shared_device = value != 0.0 ? 1 : 0;
      return SGP_OK;
    default:
      return SGP_ERR_ARG;
  }
}

extern "C" double sgp_ctx_get_option(const sgp_ctx* ctx, int option) {
  const Ctx& c = ctx ? *reinterpret_cast<const Ctx*>(ctx) : default_ctx();
  switch (option) {
    case SGP_OPT_CONTRACTION: return (double)c.contraction;
    case SGP_OPT_ASM_OVERLAP: return (double)c.asm_overlap;
    case SGP_OPT_KFU_BUDGET_BYTES: return (double)c.kfu_budget;
    case SGP_OPT_COND_LIMIT: return c.cond_limit;
    case SGP_OPT_CU_BUDGET: return (double)c.cu_budget;
    case SGP_OPT_TIMING: return (double)c.timing;
    case SGP_OPT_SHARED_DEVICE: return (double)c.shared_device;
    default: return -1.0;
  }
}

extern "C" void sgp_ctx_bind_thread(sgp_ctx* ctx) { sgp::bind_thread_ctx(ctx); }

extern "C" int sgp_ctx_device(const sgp_ctx* ctx) { return (ctx ? *reinterpret_cast<const Ctx*>(ctx) : default_ctx()).device; }

extern "C" void sgp_ctx_set_pass1_gate(sgp_ctx* ctx, void* hip_event) {
  (ctx ? *reinterpret_cast<Ctx*>(ctx) : default_ctx()).pass1_gate = (hipEvent_t)hip_event;
}

extern "C" int sgp_ctx_contraction_last(const sgp_ctx* ctx) {
  return (ctx ? *reinterpret_cast<const Ctx*>(ctx) : default_ctx()).contraction_used;
}

// ---- the option-dependent entry points with the context as first argument: install it, call the namesake ----
extern "C" size_t sgp_ctx_suffstats_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_kfu) {
  CtxScope scope(const_cast<sgp_ctx*>(ctx));
  return sgp_suffstats_workspace_bytes_ex(N, M, d, caller_owns_kfu);
}
extern "C" int sgp_ctx_suffstats_fwd(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                     const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id, double* Phi, double* b,
                                     double* yy, double* kappa, double* Kfu_out, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_fwd(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, Phi, b, yy, kappa, Kfu_out, ws, ws_bytes, stream);
}
extern "C" int sgp_ctx_suffstats_bwd(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                     const double* inv_ls, double sf2, const double* Phibar, const double* bbar, double kappabar,
                                     const double* Kfu_in, int64_t N, int M, int d, int kernel_id, double* g_ls, double* g_sf2,
                                     double* g_Z, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_bwd(X, ldx, y, Z, ldz, inv_ls, sf2, Phibar, bbar, kappabar, Kfu_in, N, M, d, kernel_id, g_ls, g_sf2, g_Z, ws,
                           ws_bytes, stream);
}
extern "C" int sgp_ctx_kuu_factor(sgp_ctx* ctx, const double* Kuu, int M, double* Linv_out, int* info, void* ws, size_t ws_bytes,
                                  sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_kuu_factor(Kuu, M, Linv_out, info, ws, ws_bytes, stream);
}
extern "C" int sgp_ctx_kuu_factor_ex(sgp_ctx* ctx, const double* Kuu, int M, double* Linv_out, int* info, double* trace_out, void* ws,
                                     size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_kuu_factor_ex(Kuu, M, Linv_out, info, trace_out, ws, ws_bytes, stream);
}
extern "C" int sgp_ctx_bound_from_stats(sgp_ctx* ctx, const double* Kuu, const double* Phi, const double* b, const double* yy,
                                        const double* kappa, double s2, int64_t N, int M, int with_adjoints, double* out,
                                        double* Phibar, double* bbar, double* Kuubar, double* factors, const double* kuu_linv,
                                        int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_bound_from_stats(Kuu, Phi, b, yy, kappa, s2, N, M, with_adjoints, out, Phibar, bbar, Kuubar, factors, kuu_linv, info, ws,
                              ws_bytes, stream);
}
extern "C" int sgp_ctx_mixture_predict(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, int64_t N, const double* Xs,
                                       int64_t ldxs, int64_t T, const double* Z, int64_t ldz, int S, const double* inv_ls,
                                       const double* sf2, const double* s2, double jitter, int M, int d, int kernel_id, int pred_noise,
                                       double gate_jitter, double* mean, double* var, double* cov, int* info, int* gate_info, void* ws,
                                       size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_mixture_predict(X, ldx, y, N, Xs, ldxs, T, Z, ldz, S, inv_ls, sf2, s2, jitter, M, d, kernel_id, pred_noise, gate_jitter,
                             mean, var, cov, info, gate_info, ws, ws_bytes, stream);
}

// ---- ABI 3: the remaining option-dependent entry points (the evaluation orders the streaming guard falls back to read the K'_fu
// budget, the timing slots and the CU budget of the context; the whitened bound's Cholesky reads the CU budget and the conditioning limit) ----
extern "C" size_t sgp_ctx_suffstats_whitened_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d) {
  CtxScope scope(const_cast<sgp_ctx*>(ctx));
  return sgp_suffstats_whitened_workspace_bytes(N, M, d);
}
extern "C" int sgp_ctx_suffstats_fwd_whitened(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                              const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                              const double* kuu_linv, double* W, double* u, double* yy, double* kappa, void* ws,
                                              size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_fwd_whitened(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, W, u, yy, kappa, ws, ws_bytes, stream);
}
extern "C" size_t sgp_ctx_suffstats_whitened_rows_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_t) {
  CtxScope scope(const_cast<sgp_ctx*>(ctx));
  return sgp_suffstats_whitened_rows_workspace_bytes(N, M, d, caller_owns_t);
}
extern "C" int sgp_ctx_suffstats_fwd_whitened_rows(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z,
                                                   int64_t ldz, const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                                   const double* kuu_linv, double* W, double* u, double* yy, double* kappa,
                                                   double* T_out, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_fwd_whitened_rows(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, W, u, yy, kappa, T_out, ws,
                                         ws_bytes, stream);
}
extern "C" size_t sgp_ctx_suffstats_extended_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d) {
  CtxScope scope(const_cast<sgp_ctx*>(ctx));
  return sgp_suffstats_extended_workspace_bytes(N, M, d);
}
extern "C" int sgp_ctx_suffstats_fwd_extended(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                              const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                              const double* kuu_linv, int level, double* W, double* u, double* yy, double* kappa,
                                              double* Kfu_out, double* phi_diag, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_fwd_extended_ex(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, level, W, u, yy, kappa, Kfu_out,
                                       phi_diag, ws, ws_bytes, stream);
}
extern "C" int sgp_ctx_suffstats_fwd_extended_f16(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                                  const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                                  const double* kuu_linv, int level, double* W, double* u, double* yy, double* kappa,
                                                  double* Kfu_out, uint16_t* Kfu_f16_out, double* phi_diag, void* ws, size_t ws_bytes,
                                                  sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_fwd_extended_f16(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, level, W, u, yy, kappa, Kfu_out,
                                        Kfu_f16_out, phi_diag, ws, ws_bytes, stream);
}
extern "C" size_t sgp_ctx_suffstats_bwd_factored_workspace_bytes(const sgp_ctx* ctx, int64_t N, int M, int d, int caller_owns_t) {
  CtxScope scope(const_cast<sgp_ctx*>(ctx));
  return sgp_suffstats_bwd_factored_workspace_bytes_ex(N, M, d, caller_owns_t);
}
extern "C" int sgp_ctx_suffstats_bwd_factored(sgp_ctx* ctx, const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                              const double* inv_ls, double sf2, const double* kuu_linv, const double* Cw, double s2,
                                              const double* bbar, double kappabar, int64_t N, int M, int d, int kernel_id,
                                              const double* T_in, double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                                              sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_suffstats_bwd_factored_ex(X, ldx, y, Z, ldz, inv_ls, sf2, kuu_linv, Cw, s2, bbar, kappabar, N, M, d, kernel_id, T_in, g_ls,
                                       g_sf2, g_Z, ws, ws_bytes, stream);
}
extern "C" int sgp_ctx_bound_from_whitened_stats(sgp_ctx* ctx, const double* W, const double* u, const double* yy, const double* kappa,
                                                 double s2, int64_t N, int M, int with_adjoints, double* out, double* Phibar,
                                                 double* bbar, double* Kuubar, double* factors, const double* kuu_linv, int* info,
                                                 double* Cw, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!ctx_on_current_device(ctx)) return SGP_ERR_ARG;
  CtxScope scope(ctx);
  return sgp_bound_from_whitened_stats_ex(W, u, yy, kappa, s2, N, M, with_adjoints, out, Phibar, bbar, Kuubar, factors, kuu_linv, info,
                                          Cw, ws, ws_bytes, stream);
}
