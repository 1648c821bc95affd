// Micro-benchmark (diagnostic, not part of the library): issue rates of the fp64 instructions the
// kernels are built from -- v_mfma_f64_16x16x4_f64, v_fma_f64, exp() -- on every CU at once.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_rates.hip -o tools/fp64_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void exp_loop(double* out, int iters) {
  double x[4];
  for (int i = 0; i < 4; ++i) x[i] = -(threadIdx.x * 1e-3 + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = -exp(x[i]) - 0.5;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}
// MFMA and VALU exp in the same wave (co-issue check)
__global__ __launch_bounds__(256) void mix_loop(double* out, int iters) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double x = -(threadIdx.x * 1e-3);
  double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      x = -exp(x) - 0.5;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <typename F>
static double time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out;
  hipMalloc(&out, 2048 * 256 * 8);
  const int iters = 20000;
  for (int wg_per_cu : {1, 2}) {
    const int grid = p.multiProcessorCount * wg_per_cu;
    double ms = time_ms([&] { mfma_loop<4><<<grid, 256>>>(out, iters); });
    double flops = (double)grid * 4 * iters * 4.0 * 2048.0;
    printf("mfma_f64_16x16x4 acc=4  %d WG/CU: %.3f ms  %.1f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wg_per_cu, ms, flops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / (iters * 4.0 * wg_per_cu));
    ms = time_ms([&] { mfma_loop<16><<<grid, 256>>>(out, iters / 4); });
    flops = (double)grid * 4 * (iters / 4) * 16.0 * 2048.0;
    printf("mfma_f64_16x16x4 acc=16 %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms * 1e-9);
    ms = time_ms([&] { fma_loop<<<grid, 256>>>(out, iters); });
    flops = (double)grid * 256 * iters * 8.0 * 2.0;
    printf("v_fma_f64               %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms * 1e-9);
    ms = time_ms([&] { exp_loop<<<grid, 256>>>(out, iters / 10); });
    printf("exp(f64)                %d WG/CU: %.3f ms  %.2f Texp/s\n", wg_per_cu, ms, (double)grid * 256 * (iters / 10) * 4.0 / ms * 1e-9);
    ms = time_ms([&] { mix_loop<<<grid, 256>>>(out, iters / 10); });
    printf("mix 1 mfma + 1 exp      %d WG/CU: %.3f ms  -> %.1f TFLOP/s mfma, %.2f Texp/s\n", wg_per_cu, ms,
           (double)grid * 4 * (iters / 10) * 4.0 * 2048.0 / ms * 1e-9, (double)grid * 256 * (iters / 10) * 4.0 / ms * 1e-9);
  }
  return 0;
}
