// Micro-benchmark (diagnostic, not part of the library): what the int8 matrix pipe of gfx950 sustains next to the fp64 one --
// the arithmetic behind DESIGN.md section 8 "next lever": an error-free (Ozaki-type) splitting of K'_fu in [0, 1] into 7-8 bit
// integer slices would run the contraction on v_mfma_i32_16x16x64_i8 instead of v_mfma_f64_16x16x4_f64.
//   build: hipcc --offload-arch=gfx950 -O3 tools/i8_rates.hip -o tools/i8_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, rt; };

template <int NACC>
__global__ __launch_bounds__(512) void i8_kernel(int* out, Stamp* st, int iters) {
  i4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i4{0, 0, 0, 0};
  const i4 a = i4{(int)threadIdx.x * 0x01010101, 0x03020100 + (int)threadIdx.x, 0x07060504, 0x0b0a0908};
  const i4 b = i4{0x11223344 ^ (int)threadIdx.x, 0x55667788, 0x01020304, 0x7f7e7d7c};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}
template <int NACC>
__global__ __launch_bounds__(512) void f64_kernel(int* out, Stamp* st, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  const double a = threadIdx.x * 1.37e-3 + 0.61, b = 0.7331 + threadIdx.x * 1e-4;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}


typedef int i16v __attribute__((ext_vector_type(16)));
// the SYRK prototype's shape: 14 independent 32x32x32 accumulators (224 registers), operands with full-entropy bytes
template <int NACC>
__global__ __launch_bounds__(256) void i8_32_kernel(int* out, Stamp* st, int iters) {
  i16v acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  i4 a[2], b[7];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; a[i][j] = (int)h; }
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; b[i][j] = (int)h; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}
template <int NACC>
__global__ __launch_bounds__(256) void i8_16r_kernel(int* out, Stamp* st, int iters) {
  i4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i4{0, 0, 0, 0};
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  i4 a[4], b[14];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; a[i][j] = (int)h; }
#pragma unroll
  for (int i = 0; i < 14; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; b[i][j] = (int)h; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

template <typename K>
static void run(const char* name, K kern, int block, int grid, int nacc, double macs_per_mfma, int iters, int* out, Stamp* st) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kern<<<grid, block>>>(out, st, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<grid, block>>>(out, st, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  Stamp h;
  (void)hipMemcpy(&h, st, sizeof(Stamp), hipMemcpyDeviceToHost);
  const double waves = (double)grid * block / 64.0;
  const double macs = waves * (double)iters * nacc * macs_per_mfma;
  printf("%-44s %8.3f ms  %9.2f TMAC/s  cycles/MFMA/wave %6.1f  clock %5.0f MHz\n", name, ms, macs / (ms * 1e-3) / 1e12,
         (double)h.cyc / ((double)iters * nacc), (double)h.cyc / ((double)h.rt * 10.0) * 1e3);  // s_memrealtime ticks at 100 MHz (10 ns)
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  int* out;
  Stamp* st;
  (void)hipMalloc(&out, sizeof(int) * 512 * ncu * 8);
  (void)hipMalloc(&st, sizeof(Stamp) * ncu * 8);
  printf("%s, %d CUs\n", p.name, ncu);
  const int iters = 20000;
  run("f64 16x16x4   1 wave/SIMD, 16 acc", f64_kernel<16>, 256, ncu, 16, 16.0 * 16 * 4, iters / 4, out, st);
  run("f64 16x16x4   2 waves/SIMD, 16 acc", f64_kernel<16>, 512, ncu, 16, 16.0 * 16 * 4, iters / 4, out, st);
  run("i8  16x16x64  1 wave/SIMD, 16 acc", i8_kernel<16>, 256, ncu, 16, 16.0 * 16 * 64, iters, out, st);
  run("i8  16x16x64  2 waves/SIMD, 16 acc", i8_kernel<16>, 512, ncu, 16, 16.0 * 16 * 64, iters, out, st);
  run("i8  16x16x64  2 waves/SIMD, 16 acc (2 WG/CU)", i8_kernel<16>, 512, 2 * ncu, 16, 16.0 * 16 * 64, iters, out, st);
  run("i8  32x32x32  1 wave/SIMD, 14 acc, random bytes", i8_32_kernel<14>, 256, ncu, 14, 32.0 * 32 * 32, iters / 2, out, st);
  run("i8  16x16x64  1 wave/SIMD, 56 acc, random bytes", i8_16r_kernel<56>, 256, ncu, 56, 16.0 * 16 * 64, iters / 4, out, st);
  return 0;
}
