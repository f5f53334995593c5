// Where does a wave of cq::k_cq_upd spend its time?  Level-0 nodes of a 16384-row matrix factored by k_cq_fac2, then the
// update of `ntl` trailing tiles in tile groups of `tpg`; built with -DCQ_UPROF the kernel adds up, per wave, the cycles
// of: [0] image load + first tile request + barrier, [1] wait for the tile, [2] phase A, [3] phase B, [4] phase C.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DCQ_UPROF] -Imatrixproductbp.jl_amd/csrc tools/probes/cq_upd_probe.hip -o tools/_cq_upd_probe.bin
#include "wg_common.h"
namespace v2 { struct QrProb { double* Y; double* aux; int32_t ld, rows, cols, kmax; }; }
#include "cq_kernels.h"
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char** argv) {
  const int rows = 16384, ntl = argc > 1 ? atoi(argv[1]) : 64, cols = 64 + 16 * ntl, ld = rows + (argc > 2 ? atoi(argv[2]) : 0), c16 = cols + 16;
  std::vector<double> Y((size_t)ld * c16, 0.0);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd;
  for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) Y[i + (size_t)ld * j] = nd(rng);
  double *dY, *dAux; v2::QrProb* dP;
  const int n = rows / 256;
  const size_t auxd = 4096 + (size_t)(n + 2) * cq::IMG_DOUBLES;
  hipMalloc(&dY, Y.size() * 8); hipMalloc(&dAux, auxd * 8); hipMalloc(&dP, sizeof(v2::QrProb));
  hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice);
  v2::QrProb hp{dY, dAux, ld, rows, cols, 64};
  hipMemcpy(dP, &hp, sizeof hp, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)cq::k_cq_fac2, hipFuncAttributeMaxDynamicSharedMemorySize, cq::FAC_LDS_DOUBLES * 8);
  hipFuncSetAttribute((const void*)cq::k_cq_upd<256>, hipFuncAttributeMaxDynamicSharedMemorySize, cq::UPD_LDS_DOUBLES * 8);
  hipLaunchKernelGGL(cq::k_cq_fac2, dim3(n, 1), dim3(256), cq::FAC_LDS_DOUBLES * 8, 0, dP, (int64_t)4096, 0, 0, 0, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nthr : {256})                 // (the eight-wave build <512> of round 3 is in profiles/r04_cq_upd_probe.txt; removed since)
  for (int tpg : {8, 16, 32, 64}) {
    if (tpg > ntl) continue;
    const int ntg = (ntl + tpg - 1) / tpg;
    for (int rep = 0; rep < 3; rep++) {
#ifdef CQ_UPROF
      { unsigned long long z[8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(cq::cq_uprof), z, sizeof z); }
#endif
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(cq::k_cq_upd<256>, dim3(ntg, n, 1), dim3(256), cq::UPD_LDS_DOUBLES * 8, 0, dP, (int64_t)4096, 0, 0, 0, tpg, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) {
        const double fl = 2.0 * 2.0 * 256 * 64 * 16 * (double)ntl * n;     // W0 and C -= V W
        printf("ld %d ntl %d tpg %2d (%4d workgroups of %d threads): %8.1f us  %6.2f TFLOP/s\n", ld, ntl, tpg, ntg * n, nthr, ms * 1e3, fl / ms * 1e-9);
#ifdef CQ_UPROF
        unsigned long long h[8];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(cq::cq_uprof), sizeof h);
        const double tiles = (double)ntl * n, wvs = (double)h[6];
        printf("    shader clock while the workgroups run: %.0f MHz (s_memtime cycles / s_memrealtime 100 MHz ticks)\n", (double)h[5] / (double)h[7] * 100.0);
        printf("    per wave: image + first request + barrier %7.0f cycles;  per tile: wait %6.0f  phase A %6.0f  phase B %6.0f  phase C %6.0f  (MFMA issue: 16384 / 2560 / 16384)\n",
               h[0] / wvs, h[1] / tiles, h[2] / tiles, h[3] / tiles, h[4] / tiles);
#endif
      }
    }
  }
  return 0;
}
