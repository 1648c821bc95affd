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

#ifdef NUTS_HOST_MAIN
// Sanitizer build (tests/test_nuts_device_logic.py::test_sampler_header_is_clean_under_asan_ubsan): the state machine alone, on
// targets that exercise every branch -- smooth Gaussians in 1..NUTS_MAXD dimensions, a zero-density wall (logp = -inf =>
// divergence bookkeeping), a tree-depth limit of 1 and of 10 -- compiled with -fsanitize=address,undefined.
#include <cmath>
#include <cstdio>
#include <vector>
static int g_ndim = 1;
static int g_wall = 0;
static void target(const double* q, double* logp, double* grad) {
  double lp = 0.0;
  for (int i = 0; i < g_ndim; ++i) {
    const double sd = 0.2 + 0.3 * i;
    const double z = (q[i] - 0.5 * i) / sd;
    lp -= 0.5 * z * z;
    grad[i] = -z / sd;
  }
  if (g_wall && q[0] > 0.7) lp = -INFINITY;
  *logp = lp;
}
int main() {
  long total = 0;
  const int dims[] = {1, 2, 5, sgp::NUTS_MAXD};
  for (int ndim : dims)
    for (int wall = 0; wall < 2; ++wall)
      for (int depth : {1, 10}) {
        g_ndim = ndim;
        g_wall = wall;
        const int tune = 120, draws = 60;
        std::vector<double> q0(ndim, 0.1), samples((size_t)draws * ndim), stats((size_t)draws * 8);
        total += nuts_host_run(ndim, tune, draws, depth, 0.25, 0.8, 1234ull + ndim, q0.data(), target, samples.data(), stats.data(), nullptr);
        for (double v : samples)
          if (!std::isfinite(v)) { std::printf("non-finite draw (ndim %d wall %d depth %d)\n", ndim, wall, depth); return 1; }
      }
  std::printf("sanitized sampler ok: %ld leapfrogs\n", total);
  return 0;
}
#endif
