// Device time of the 64 x 64-tile fp64 GEMM of sgp_dense.hip at the shapes the O(M^3) tail uses (M = 1024, 512):
// full product, k clipped by the lower-triangular operand (khi_mask), and the batched products of tri_inverse.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_bench tools/gemm_bench.hip && tools/gemm_bench
#include "../generalised-gaussian-processes_amd/csrc/sgp_dense.hip"
#include <cstdio>
#include <vector>
namespace sgp { size_t stream_kfu_budget() { return 0; } }
using namespace sgp;

static float time_gemm(GemmDesc g, hipStream_t st, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm(g, st);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) gemm(g, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  for (int M : {1024, 512, 256}) {
    const size_t mm = (size_t)M * M;
    std::vector<double> h(mm);
    for (size_t i = 0; i < mm; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    double *A, *B, *C;
    hipMalloc(&A, mm * 8); hipMalloc(&B, mm * 8); hipMalloc(&C, mm * 8);
    hipMemcpy(A, h.data(), mm * 8, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), mm * 8, hipMemcpyHostToDevice);
    GemmDesc g; g.A = A; g.B = B; g.C = C; g.lda = g.ldb = g.ldc = M; g.m = g.n = g.k = M;
    const double gf = 2.0 * M * (double)M * M / 1e9;
    float t = time_gemm(g, st, 50);
    printf("M=%4d NN full      %7.1f us  %6.1f TFLOP/s\n", M, t, gf / t * 1e3);
    g.tb = true; t = time_gemm(g, st, 50);
    printf("M=%4d NT full      %7.1f us  %6.1f TFLOP/s\n", M, t, gf / t * 1e3);
    g.tb = false; g.ta = true; t = time_gemm(g, st, 50);
    printf("M=%4d TN full      %7.1f us  %6.1f TFLOP/s\n", M, t, gf / t * 1e3);
    g.ta = false; g.khi_mask = 1; t = time_gemm(g, st, 50);
    printf("M=%4d NN khi(tri)  %7.1f us  (critical path = full k)\n", M, t);
    g.khi_mask = 0; g.klo_mask = 3; g.ta = true; t = time_gemm(g, st, 50);
    printf("M=%4d TN klo=3     %7.1f us\n", M, t);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
