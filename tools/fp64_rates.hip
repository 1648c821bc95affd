// Micro-benchmark (diagnostic, not part of the library): what the fp64 pipes of gfx950 sustain.
//   * v_mfma_f64_16x16x4_f64 alone, by waves/SIMD and number of independent accumulators
//   * the same with N independent v_fma_f64 / v_fma_f32 issued between MFMAs (do they overlap?)
//   * v_fma_f64 alone
// Reports wall TFLOP/s, shader cycles per MFMA (s_memtime) and the clock the chip held
// (s_memtime / s_memrealtime, the latter ticks at 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_rates.hip -o tools/fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, rt; };

template <int NACC, int NF64, int NF32>
__global__ __launch_bounds__(512) void mix_kernel(double* out, Stamp* st, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double x[NF64 > 0 ? NF64 : 1];
  float f[NF32 > 0 ? NF32 : 1];
#pragma unroll
  for (int i = 0; i < (NF64 > 0 ? NF64 : 1); ++i) x[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
  for (int i = 0; i < (NF32 > 0 ? NF32 : 1); ++i) f[i] = threadIdx.x * 1e-3f + i;
  const double a = threadIdx.x * 1.37e-3 + 0.61, b = 0.7331 + threadIdx.x * 1e-4;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NF64; ++q) x[q] = fma(x[q], 1.0000001, 1e-9);
#pragma unroll
      for (int q = 0; q < NF32; ++q) f[q] = fmaf(f[q], 1.0000001f, 1e-9f);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < (NF64 > 0 ? NF64 : 1); ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < (NF32 > 0 ? NF32 : 1); ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

template <int NCH>
__global__ __launch_bounds__(512) void fma_kernel(double* out, Stamp* st, int iters) {
  double x[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = fma(x[i], 1.0000001, 1e-9);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

static double* out;
static Stamp* stamps;
static int ncu;

template <typename F>
static void run(const char* name, F launch, int grid, double flops_mfma, double n_mfma_per_wave, double flops_valu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(grid);
  hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (auto& s : h) { cyc += s.cyc; rt += s.rt; }
  cyc /= grid; rt /= grid;
  const double clk_ghz = cyc / (rt * 10.0);  // rt ticks are 10 ns
  printf("%-34s %7.3f ms  mfma %5.1f TF  valu %5.1f TF  cyc/MFMA/wave %6.1f  clock %.2f GHz\n", name, ms,
         flops_mfma / ms * 1e-9, flops_valu / ms * 1e-9, n_mfma_per_wave > 0 ? cyc / n_mfma_per_wave : 0.0, clk_ghz);
}

template <int NACC, int NF64, int NF32>
static void bench_mix(int wg_per_cu, int threads, int iters) {
  const int grid = ncu * wg_per_cu;
  const double waves = (double)grid * threads / 64;
  char name[128];
  snprintf(name, sizeof name, "mfma acc=%d +%df64 +%df32 w/SIMD=%d", NACC, NF64, NF32, wg_per_cu * threads / 256);
  run(name, [&] { mix_kernel<NACC, NF64, NF32><<<grid, threads>>>(out, stamps, iters); }, grid,
      waves * iters * NACC * 2048.0, (double)iters * NACC, waves * 64 * iters * NACC * NF64 * 2.0);
}
template <int NCH>
static void bench_fma(int wg_per_cu, int threads, int iters) {
  const int grid = ncu * wg_per_cu;
  char name[128];
  snprintf(name, sizeof name, "v_fma_f64 chains=%d w/SIMD=%d", NCH, wg_per_cu * threads / 256);
  run(name, [&] { fma_kernel<NCH><<<grid, threads>>>(out, stamps, iters); }, grid, 0.0, 0.0,
      (double)grid * threads * iters * NCH * 2.0);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  ncu = p.multiProcessorCount;
  printf("device %s CUs=%d clock=%d kHz\n", p.name, ncu, p.clockRate);
  hipMalloc(&out, (size_t)4096 * 512 * 8);
  hipMalloc(&stamps, 4096 * sizeof(Stamp));
  const int it = 4000;
  bench_mix<4, 0, 0>(1, 256, it);
  bench_mix<8, 0, 0>(1, 256, it);
  bench_mix<16, 0, 0>(1, 256, it);
  bench_mix<4, 0, 0>(2, 256, it);
  bench_mix<8, 0, 0>(2, 256, it);
  bench_mix<16, 0, 0>(2, 256, it);
  bench_mix<4, 0, 0>(2, 512, it);
  bench_mix<8, 0, 0>(2, 512, it);
  bench_mix<8, 2, 0>(1, 256, it);
  bench_mix<8, 4, 0>(1, 256, it);
  bench_mix<8, 8, 0>(1, 256, it);
  bench_mix<8, 16, 0>(1, 256, it);
  bench_mix<8, 4, 0>(2, 256, it);
  bench_mix<8, 8, 0>(2, 256, it);
  bench_mix<8, 16, 0>(2, 256, it);
  bench_mix<8, 0, 8>(1, 256, it);
  bench_mix<8, 0, 16>(1, 256, it);
  bench_mix<8, 0, 16>(2, 256, it);
  bench_fma<8>(1, 256, it * 8);
  bench_fma<8>(2, 256, it * 8);
  bench_fma<8>(2, 512, it * 8);
  bench_fma<16>(2, 512, it * 4);
  return 0;
}
