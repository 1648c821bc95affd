// Dependent-issue latencies of the fp64 building blocks of the pivot chain, one wave, gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_latency tools/fp64_latency.hip && ./fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 256
#define PIN(v) asm volatile("" : "+v"(v))
#define T0() __builtin_amdgcn_sched_barrier(0); PIN(x); PIN(r); PIN(q); PIN(w); t0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define T1() __builtin_amdgcn_sched_barrier(0); PIN(x); PIN(r); PIN(q); PIN(w); PINC(); t1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define PINC() for (int kk = 0; kk < 8; ++kk) PIN(c[kk])
__global__ void lat_kernel(double* out, long long* cyc, double seed) {
  __shared__ double buf[128];
  double x = seed + threadIdx.x * 1e-9, y = 1.000001, r = 3.0 + seed, q = seed * 2, w = seed * 3;
  double c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = seed + k;
  PIN(y);
  long long t0, t1;
  // 1. dependent v_fma_f64
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, 1e-9);
  T1();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // 2. dependent v_mul_f64
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) x = x * y;
  T1();
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // 3. dependent v_rsq_f64
  r = x * x + 2.0;
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) r = __builtin_amdgcn_rsq(r) + 0.0 * i;
  T1();
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // 4. readlane -> VALU use -> readlane (fp64 through two v_readlane_b32)
  q = r + x;
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int lo = __double2loint(q), hi = __double2hiint(q);
    lo = __builtin_amdgcn_readlane(lo, (i * 7 + 3) & 63);
    hi = __builtin_amdgcn_readlane(hi, (i * 7 + 3) & 63);
    q = __builtin_fma(__hiloint2double(hi, lo), y, q);
  }
  T1();
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  // 5. LDS write -> uniform read -> fma round trip
  w = q;
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    buf[threadIdx.x + (i & 1) * 64] = w;
    w = __builtin_fma(buf[((i * 5 + 1) & 63) + (i & 1) * 64], y, w);
  }
  T1();
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  // 6. independent fma throughput (8 chains)
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = w + k;
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __builtin_fma(c[k], y, 1e-9);
  T1();
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  // 7. uniform LDS read feeding independent FMAs (the rank-16 update pattern): 8 accumulators x N reads
  T0();
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __builtin_fma(buf[(i * 8 + k) & 127], y, c[k]);
  T1();
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  // 9. uniform ds_read_b128 feeding FMAs (two doubles per read), 8 chains
  {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v* b2 = reinterpret_cast<const d2v*>(buf);
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const d2v v = b2[(i * 4 + k) & 63];
        c[2 * k] = __builtin_fma(v[0], y, c[2 * k]);
        c[2 * k + 1] = __builtin_fma(v[1], y, c[2 * k + 1]);
      }
    T1();
    if (threadIdx.x == 0) cyc[8] = t1 - t0;
  }
  // 10. per-lane (conflict-free) ds_read_b128 feeding FMAs
  {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v* b2 = reinterpret_cast<const d2v*>(buf);
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const d2v v = b2[(threadIdx.x + i * 4 + k) & 63];
        c[2 * k] = __builtin_fma(v[0], y, c[2 * k]);
        c[2 * k + 1] = __builtin_fma(v[1], y, c[2 * k + 1]);
      }
    T1();
    if (threadIdx.x == 0) cyc[9] = t1 - t0;
  }
  // 8. realtime reference: cycles of s_memtime per 100 MHz tick
  long long r0 = __builtin_amdgcn_s_memrealtime();
  t0 = __builtin_readcyclecounter();
  while (__builtin_amdgcn_s_memrealtime() - r0 < 1000) {}
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  double s = x + r + q + w;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += c[k];
  out[threadIdx.x] = s;
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, 16 * sizeof(long long));
  for (int rep = 0; rep < 2; ++rep) lat_kernel<<<1, 64>>>(out, cyc, 1.0);
  hipDeviceSynchronize();
  long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[10] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_rsq_f64 (+add)", "readlane x2 -> fma -> readlane",
                          "LDS write -> uniform read -> fma", "8 independent fma chains (per fma)", "uniform LDS read + fma, 8 chains (per fma)",
                          "s_memtime ticks per 10 us", "uniform ds_read_b128 + 2 fma (per fma)", "per-lane ds_read_b128 + 2 fma (per fma)"};
  for (int k = 0; k < 10; ++k) {
    double per = (k == 5 || k == 6 || k == 8 || k == 9) ? (double)h[k] / (N * 8) : (k == 7 ? (double)h[k] : (double)h[k] / N);
    printf("%-48s %10.1f\n", names[k], per);
  }
  return 0;
}
