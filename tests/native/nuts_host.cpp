// Host build of the device sampler (csrc/sgp_nuts.hpp) for the CPU tests: runs the state machine against a log-density
// supplied as a C callback (ctypes), so the very code the persistent GPU kernel executes is checked draw for draw
// against the Python sampler on this GPU-less container.
#include "sgp_nuts.hpp"

extern "C" {
typedef void (*logp_cb)(const double* q, double* logp, double* grad);

int nuts_host_state_bytes(void) { return (int)sizeof(sgp::NutsState); }

long nuts_host_run(int ndim, int n_tune, int n_draws, int max_treedepth, double step_scale, double target_accept,
                   unsigned long long seed, const double* q0, logp_cb cb, double* samples, double* stats, double* step_sizes_all) {
  static sgp::NutsState s;
  sgp::nuts_init(s, ndim, n_tune, n_draws, max_treedepth, step_scale, target_accept, seed, q0);
  double lp = 0.0, grad[sgp::NUTS_MAXD];
  const double* q = nullptr;
  int last_it = -1;
  for (;;) {
    const int cmd = sgp::nuts_step(s, lp, grad, &q, samples, stats);
    if (step_sizes_all && s.it != last_it && s.it < n_tune + n_draws) {
      last_it = s.it;
    }
    if (cmd == sgp::NUTS_DONE) break;
    cb(q, &lp, grad);
  }
  return s.n_leapfrog;
}
}
