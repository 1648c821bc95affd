// Prototype for DESIGN.md section 8's next lever: what the double-double triple product W = L^-1 Phi L^-T of an extended-precision
// streaming order would cost.  C (double-double) = A (double-double, M x M) * B^T (fp64, M x M), plain fp64 VALU (two_prod by fma,
// two_sum), 64 x 64 tiles of 256 threads, 4 x 4 outputs per thread, 16-deep k-chunks through LDS.  Applied twice:
//     Y^T = (Phi L^-T)^T   (output written transposed)      W = Y^T L^-T   (W is symmetric, so this IS L^-1 Phi L^-T)
// Checks both products against long double on the host at M = 256, then times M = 1024.
//     hipcc --offload-arch=gfx950 -O3 -o tools/dd_gemm_proto tools/dd_gemm_proto.hip && ./tools/dd_gemm_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int T = 64, KC = 16;

// C[i][j] (dd) = sum_k (Ahi + Alo)[i][k] * B[j][k];  tr: write C[j][i] instead
__global__ __launch_bounds__(256) void dd_gemm_nt(const double* __restrict__ Ahi, const double* __restrict__ Alo, const double* __restrict__ B,
                                                  int M, double* __restrict__ Chi, double* __restrict__ Clo, int tr) {
#pragma clang fp contract(off)  // the error-free transformations below must not be fused: s = hi + p would become fma(a, b, hi)
  __shared__ double sAh[KC][T + 1], sAl[KC][T + 1], sB[KC][T + 1];
  const int i0 = blockIdx.y * T, j0 = blockIdx.x * T, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  double hi[4][4], lo[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) hi[a][b] = lo[a][b] = 0.0;
  for (int k0 = 0; k0 < M; k0 += KC) {
    for (int e = tid; e < T * KC; e += 256) {  // row r of the tile, k-offset kk: consecutive threads along k (contiguous in memory)
      const int r = e / KC, kk = e % KC;
      sAh[kk][r] = Ahi[(size_t)(i0 + r) * M + k0 + kk];
      sAl[kk][r] = Alo[(size_t)(i0 + r) * M + k0 + kk];
      sB[kk][r] = B[(size_t)(j0 + r) * M + k0 + kk];
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < KC; ++kk) {
      double ah[4], al[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { ah[a] = sAh[kk][ti + 16 * a]; al[a] = sAl[kk][ti + 16 * a]; bv[a] = sB[kk][tj + 16 * a]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double p = ah[a] * bv[b];
          const double e = fma(ah[a], bv[b], -p) + al[a] * bv[b];  // exact low part of the product + the low operand's share
          const double s = hi[a][b] + p;                           // two_sum(hi, p)
          const double z = s - hi[a][b];
          const double t = (hi[a][b] - (s - z)) + (p - z);
          hi[a][b] = s;
          lo[a][b] += t + e;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double s = hi[a][b] + lo[a][b];  // renormalise
      const double l = lo[a][b] - (s - hi[a][b]);
      const int i = i0 + ti + 16 * a, j = j0 + tj + 16 * b;
      const size_t o = tr ? (size_t)j * M + i : (size_t)i * M + j;
      Chi[o] = s;
      Clo[o] = l;
    }
}

static double rnd() { return (double)rand() / RAND_MAX - 0.5; }

int main() {
  srand(1);
  for (int M : {256, 1024}) {
    const size_t mm = (size_t)M * M;
    std::vector<double> Ph(mm), Pl(mm), Li(mm, 0.0);
    // Phi-like: symmetric, large positive entries; L^-1-like: lower triangular with alternating, growing entries (cond-sized products)
    for (int i = 0; i < M; ++i)
      for (int j = 0; j <= i; ++j) {
        const double v = 1.0e5 * exp(-1e-4 * (i - j) * (i - j)) * (1.0 + 1e-3 * rnd());
        Ph[(size_t)i * M + j] = Ph[(size_t)j * M + i] = v;
        const double l = v * 1e-17 * rnd();
        Pl[(size_t)i * M + j] = Pl[(size_t)j * M + i] = l;
        Li[(size_t)i * M + j] = ((i - j) & 1 ? -1.0 : 1.0) * exp(0.01 * j - 0.02 * (i - j)) * (1.0 + 0.1 * rnd());
      }
    double *dPh, *dPl, *dLi, *dYh, *dYl, *dWh, *dWl;
    for (double** p : {&dPh, &dPl, &dLi, &dYh, &dYl, &dWh, &dWl}) CK(hipMalloc(p, mm * 8));
    CK(hipMemcpy(dPh, Ph.data(), mm * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dPl, Pl.data(), mm * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dLi, Li.data(), mm * 8, hipMemcpyHostToDevice));
    dim3 grid(M / T, M / T);
    auto run = [&]() {
      dd_gemm_nt<<<grid, 256>>>(dPh, dPl, dLi, M, dYh, dYl, 1);   // Y^T
      dd_gemm_nt<<<grid, 256>>>(dYh, dYl, dLi, M, dWh, dWl, 0);   // W = Y^T L^-T
    };
    run();
    CK(hipDeviceSynchronize());
    if (M == 256) {
      std::vector<double> Wh(mm), Wl(mm);
      CK(hipMemcpy(Wh.data(), dWh, mm * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(Wl.data(), dWl, mm * 8, hipMemcpyDeviceToHost));
      // long-double reference and the plain fp64 triple product
      std::vector<long double> Y(mm);
      std::vector<double> Yd(mm);
      for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
          long double s = 0;
          double sd = 0;
          for (int k = 0; k < M; ++k) {
            s += ((long double)Ph[(size_t)i * M + k] + Pl[(size_t)i * M + k]) * Li[(size_t)j * M + k];
            sd += Ph[(size_t)i * M + k] * Li[(size_t)j * M + k];
          }
          Y[(size_t)i * M + j] = s;
          Yd[(size_t)i * M + j] = sd;
        }
      long double emax_dd = 0, emax_d = 0, wmax = 0;
      for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
          long double s = 0;
          double sd = 0;
          for (int k = 0; k < M; ++k) {
            s += Li[(size_t)i * M + k] * Y[(size_t)k * M + j];
            sd += Li[(size_t)i * M + k] * Yd[(size_t)k * M + j];
          }
          const long double got = (long double)Wh[(size_t)i * M + j] + Wl[(size_t)i * M + j];
          emax_dd = fmaxl(emax_dd, fabsl(got - s));
          emax_d = fmaxl(emax_d, fabsl((long double)sd - s));
          wmax = fmaxl(wmax, fabsl(s));
        }
      printf("M=256: max |W| %.3Le; max error of the double-double product %.3Le (%.2Le of max |W|), of the plain fp64 product %.3Le (%.2Le)\n",
             wmax, emax_dd, emax_dd / wmax, emax_d, emax_d / wmax);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M=%d: two double-double products %.3f ms (%.1f GFLOP/s counting 2 M^3 each)\n", M, ms / 10, 2.0 * 2.0 * M * M * M / (ms / 10 * 1e-3) / 1e9);
    for (double* p : {dPh, dPl, dLi, dYh, dYl, dWh, dWl}) CK(hipFree(p));
  }
  return 0;
}
