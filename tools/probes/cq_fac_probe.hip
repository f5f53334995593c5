// Timing of cq::k_cq_fac2 (one node = 256 x 64, and a full level) with parts of the column step removed (-DCQ_VAR=n):
// where do the ~1.4 us per Householder column go?  Build: see run_cq_fac_probe.sh
#include "wg_common.h"
namespace v2 { struct QrProb { double* Y; double* aux; int32_t ld, rows, cols, kmax; }; }
#include "cq_kernels.h"
#include <cstdio>
#include <vector>
#include <random>
int main() {
  const int rows = 16384, cols = 256, ld = rows, c16 = cols + 16;
  std::vector<double> Y((size_t)ld * c16, 0.0);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd;
  for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) Y[i + (size_t)ld * j] = nd(rng);
  double *dY, *dAux; v2::QrProb* dP;
  const size_t auxd = 4096 + (size_t)90 * cq::IMG_DOUBLES;
  hipMalloc(&dY, Y.size() * 8); hipMalloc(&dAux, auxd * 8); hipMalloc(&dP, sizeof(v2::QrProb));
  hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice);
  v2::QrProb hp{dY, dAux, ld, rows, cols, cols};
  hipMemcpy(dP, &hp, sizeof hp, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)cq::k_cq_fac2, hipFuncAttributeMaxDynamicSharedMemorySize, cq::FAC_LDS_DOUBLES * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int level = 0; level < 2; level++)
    for (int n : {1, 16, 64}) {
      if (level == 1 && n > 16) continue;
      for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(cq::k_cq_fac2, dim3(n, 1), dim3(256), cq::FAC_LDS_DOUBLES * 8, 0, dP, (int64_t)4096, 0, level, 0, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("variant %d level %d nodes %2d: %.1f us\n", CQ_VAR, level, n, ms * 1e3);
#ifdef CQ_PROF
        if (rep == 2 && n == 1) {
          unsigned long long h[4][8];
          hipMemcpyFromSymbol(h, HIP_SYMBOL(cq::cq_prof), sizeof h);
          const char* nm[6] = {"broadcast", "dot+LDS write", "barrier", "LDS sum+readlane", "dlarfg+tw", "update"};
          for (int s = 0; s < 6; s++) printf("    %-18s cycles per step, waves 0..3: %6.0f %6.0f %6.0f %6.0f\n", nm[s], h[0][s] / 64.0, h[1][s] / 64.0, h[2][s] / 64.0, h[3][s] / 64.0);
        }
        { unsigned long long z[4][8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(cq::cq_prof), z, sizeof z); }
#endif
      }
    }
  return 0;
}
