// Context of the C ABI (include/sgp.h: sgp_ctx_*).  Everything that used to be a process-wide switch -- which matrix cores
// contract, the conditioning limit, the K'_fu budget, the CU budget, the timing events, the one-shot pass-1 gate, the library-owned
// side stream, the per-device kernel attributes -- lives in a Ctx.  The legacy setters (sgp_set_*) act on the DEFAULT context,
// which is also what every entry point without a context argument runs in.  The sgp_ctx_* entry points install their context for
// the duration of the call on the calling host thread (CtxScope); internal code asks cur_ctx().
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace sgp {

constexpr int CTX_TIMING_SLOTS = 3;

struct Ctx {
  int device = 0;
  // ---- options (sgp_ctx_set_option / the legacy setters on the default context) ----
  int contraction = 1;          // SGP_OPT_CONTRACTION: 0 fp64, 1 integer cores where they win, 2 integer cores always
  int asm_overlap = 0;          // SGP_OPT_ASM_OVERLAP
  size_t kfu_budget = 0;        // SGP_OPT_KFU_BUDGET_BYTES (the default is filled in at creation)
  double cond_limit = 1e13;     // SGP_OPT_COND_LIMIT
  int cu_budget = 0;            // SGP_OPT_CU_BUDGET (0 = the whole device)
  int timing = 0;               // SGP_OPT_TIMING
  int shared_device = 0;        // SGP_OPT_SHARED_DEVICE: ticketed claim in the chain-workgroup Cholesky (env SGP_SHARED_DEVICE at creation)
  // ---- tuning knobs of the environment, read ONCE when the context is created ----
  int syrk_skip_upper = 1, syrk_waves = 4, syrk_glds = 0, i8_prio = 0;
  // ---- state ----
  int contraction_used = 0;         // what the last pass 1 of this context ran (0 fp64, 1 integer cores)
  hipEvent_t pass1_gate = nullptr;  // one-shot, consumed by the next sgp_suffstats_fwd of this context
  int64_t syrk_timed_rows = 0;
  hipEvent_t ev[CTX_TIMING_SLOTS][2] = {};
  int ev_ready = 0, ev_used[CTX_TIMING_SLOTS] = {0, 0, 0};
  hipStream_t side = nullptr;       // library-owned side stream of the head + tail A/B (asm_overlap = 1)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

Ctx& default_ctx();
Ctx& cur_ctx();  // the context of the entry point running on this host thread: a sgp_ctx_* call's own, else the one bound to the thread
                 // (sgp_ctx_bind_thread), else the default context

void bind_thread_ctx(void* ctx);

struct CtxScope {
  Ctx* prev;
  explicit CtxScope(void* ctx);  // nullptr = the default context
  ~CtxScope();
  CtxScope(const CtxScope&) = delete;
  CtxScope& operator=(const CtxScope&) = delete;
};

}  // namespace sgp
