// Standalone check + timing of caqr::qr (tools/probes/caqr.h): R-only QR of `nmat` matrices per workgroup, `nblocks` workgroups.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I matrixproductbp.jl_amd/csrc -I tools/probes -o tools/_caqr_probe.bin tools/probes/caqr_probe.hip
//   tools/_caqr_probe.bin [rows cols nblocks nmat]
// Correctness: R^T R against Y^T Y (long double on the host) and |R| against a host Householder QR.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "caqr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(512) void caqr_kernel(double* Yall, long ld, int rows, int cols, long mstride, int nmat) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int m = 0; m < nmat; m++)
    caqr::qr((gdbl*)(Yall + ((long)blockIdx.x * nmat + m) * mstride), ld, rows, cols, (ldbl*)lds);
}

static void host_qr_r(std::vector<double>& A, int rows, int cols, long ld) {
  for (int j = 0; j < cols && j < rows; j++) {
    double* x = &A[(long)j * ld];
    long double ss = 0;
    for (int r = j + 1; r < rows; r++) ss += (long double)x[r] * x[r];
    const double alpha = x[j];
    if (ss == 0) continue;
    const double nrm = (double)sqrtl((long double)alpha * alpha + ss);
    const double beta = -copysign(nrm, alpha);
    const double tau = (beta - alpha) / beta, scale = 1.0 / (alpha - beta);
    for (int r = j + 1; r < rows; r++) x[r] *= scale;
    x[j] = beta;
    for (int c = j + 1; c < cols; c++) {
      double* y = &A[(long)c * ld];
      long double d = y[j];
      for (int r = j + 1; r < rows; r++) d += (long double)x[r] * y[r];
      const double t = tau * (double)d;
      y[j] -= t;
      for (int r = j + 1; r < rows; r++) y[r] -= t * x[r];
    }
  }
}

int main(int argc, char** argv) {
  int rows = argc > 1 ? atoi(argv[1]) : 1600, cols = argc > 2 ? atoi(argv[2]) : 400;
  int nblocks = argc > 3 ? atoi(argv[3]) : 1, nmat = argc > 4 ? atoi(argv[4]) : 1;
  int decay = argc > 5 ? atoi(argv[5]) : 0;
  const long ld = (rows + 31) & ~31;
  const int colsp = ((cols + 15) & ~15) + 16;
  const long mstride = ld * colsp;
  std::vector<double> Y(mstride, 0.0);
  unsigned long long st = 0x9E3779B97F4A7C15ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st % 2000001ULL) / 1e6 - 1.0; };
  for (int c = 0; c < cols; c++)
    for (int r = 0; r < rows; r++) Y[(long)c * ld + r] = rnd();
  if (decay) {
    // numerically rank deficient, like the engine's Y_t: columns = (rows x rk) (rk x cols) + 1e-13 noise
    const int rk = cols / 2;
    std::vector<double> A((long)rows * rk), B((long)rk * cols);
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    for (int c = 0; c < cols; c++)
      for (int r = 0; r < rows; r++) {
        double s = 0;
        for (int k = 0; k < rk; k++) s += A[(long)k * rows + r] * B[(long)c * rk + k] * pow(0.8, k);
        Y[(long)c * ld + r] = s + 1e-13 * rnd();
      }
  }
  const long nm = (long)nblocks * nmat;
  double* dY;
  CK(hipMalloc(&dY, sizeof(double) * mstride * nm));
  for (long m = 0; m < nm; m++) CK(hipMemcpy(dY + m * mstride, Y.data(), sizeof(double) * mstride, hipMemcpyHostToDevice));
  const int ldsb = caqr::L_TOTAL * 8;
  CK(hipFuncSetAttribute((const void*)caqr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef CAQR_PROF
  unsigned long long* dprof;
  CK(hipMalloc(&dprof, 64 * 8)); CK(hipMemset(dprof, 0, 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(caqr::g_prof), &dprof, sizeof(dprof)));
#endif
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(caqr_kernel, dim3(nblocks), dim3(512), ldsb, 0, dY, ld, rows, cols, mstride, nmat);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * rows * (double)cols * cols - 2.0 / 3.0 * (double)cols * cols * cols;
  printf("caqr %d x %d  blocks %d  mats/block %d:  %.3f ms per QR per workgroup, %.2f TFLOP/s aggregate\n", rows, cols, nblocks,
         nmat, ms / nmat, flop * nm / (ms * 1e-3) / 1e12);
#ifdef CAQR_PROF
  {
    unsigned long long hp[64];
    CK(hipMemcpy(hp, dprof, 64 * 8, hipMemcpyDeviceToHost));
    const double sc = 1.0 / (4.0 * nm);       // per wave, per QR
    printf("  panel group (cycles per QR per wave): dep-wait %.0f  load %.0f  sub-panel0 %.0f  sub-panels1-3 %.0f  image-wait %.0f  image-write %.0f\n",
           hp[0] * sc, hp[1] * sc, hp[2] * sc, hp[3] * sc, hp[4] * sc, hp[5] * sc);
    printf("  update group: wait %.0f  images %.0f  tiles %.0f\n", hp[16] * sc, hp[17] * sc, hp[18] * sc);
  }
#endif
  // check the first and the last matrix
  double worst_g = 0, worst_r = 0;
  std::vector<double> Href = Y;
  host_qr_r(Href, rows, cols, ld);
  long double ynorm = 0;
  for (int c = 0; c < cols; c++) for (int r = 0; r < rows; r++) ynorm += (long double)Y[(long)c * ld + r] * Y[(long)c * ld + r];
  for (long m : {0L, nm - 1}) {
    std::vector<double> R(mstride);
    CK(hipMemcpy(R.data(), dY + m * mstride, sizeof(double) * mstride, hipMemcpyDeviceToHost));
    // Gram check
    long double err = 0;
    for (int a = 0; a < cols; a += (cols > 64 ? 7 : 1))
      for (int b = a; b < cols; b += (cols > 64 ? 5 : 1)) {
        long double g1 = 0, g2 = 0;
        for (int r = 0; r < rows; r++) g1 += (long double)Y[(long)a * ld + r] * Y[(long)b * ld + r];
        for (int r = 0; r <= a && r < rows; r++) g2 += (long double)R[(long)a * ld + r] * R[(long)b * ld + r];
        long double d = fabsl(g1 - g2);
        if (!(d == d)) d = 1e300;
        if (d > err) err = d;
      }
    const double relg = (double)(err / (ynorm / cols));
    if (relg > worst_g) worst_g = relg;
    if (!decay) {
      double er = 0, rn = 0;
      for (int i = 0; i < cols && i < rows; i++) {
        const double sgn = (R[(long)i * ld + i] * Href[(long)i * ld + i] < 0) ? -1.0 : 1.0;
        for (int c = i; c < cols; c++) {
          const double d = fabs(R[(long)c * ld + i] - sgn * Href[(long)c * ld + i]);
          if (!(d == d)) er = 1e300;
          if (d > er) er = d;
          if (fabs(Href[(long)c * ld + i]) > rn) rn = fabs(Href[(long)c * ld + i]);
        }
      }
      if (er / rn > worst_r) worst_r = er / rn;
    }
  }
  printf("  max |R^T R - Y^T Y| / mean column norm^2 = %.2e   max |R - R_host| / max|R| = %.2e   %s\n", worst_g, worst_r,
         (worst_g < 1e-12 && worst_r < 1e-10) ? "OK" : "FAIL");
  return (worst_g < 1e-12 && worst_r < 1e-10) ? 0 : 1;
}
