// Where does a wave of cq::k_cq_upd spend its time?  Level-0 nodes of a 16384-row matrix factored by k_cq_fac2, then the
// update of `ntl` trailing tiles in tile groups of `tpg`.  Built with -DCQ_TRACE every wave notes the shader clock per tile at:
// tile requested, tile arrived (s_waitcnt vmcnt(0)), end of phase A, of phase B, of phase C (cq_kernels.h, CQ_TR).
// -DCQ_NT=512: the two-waves-per-SIMD build.  -DCQ_NO_GLOBAL: the tiles never leave the registers.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DCQ_TRACE] [-DCQ_NT=512] -Imatrixproductbp.jl_amd/csrc tools/probes/cq_upd_probe.hip -o tools/_cq_upd_probe.bin
// usage: _cq_upd_probe.bin [tiles = 64] [extra leading dimension = 0]
#include "wg_common.h"
namespace v2 { struct QrProb { double* Y; double* aux; int32_t ld, rows, cols, kmax; }; }
#include "cq_kernels.h"
#include <cstdio>
#ifndef CQ_NT
#define CQ_NT 256
#endif
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
  hipFuncSetAttribute((const void*)cq::k_cq_upd<CQ_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, cq::UPD_LDS_DOUBLES * 8);
  hipLaunchKernelGGL(cq::k_cq_fac2, dim3(n, 1), dim3(256), cq::FAC_LDS_DOUBLES * 8, 0, dP, (int64_t)4096, 0, 0, 0, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef CQ_TRACE
  unsigned long long* dTr; const size_t ntr = (size_t)4096 * 8 * 64 * 5;
  hipMalloc(&dTr, ntr * 8); hipMemset(dTr, 0, ntr * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(cq::cq_trace_buf), &dTr, sizeof dTr);
#endif
  for (int nthr : {CQ_NT})
  for (int tpg : {8, 16, 32, 64}) {
    if (tpg > ntl) continue;
    const int ntg = (ntl + tpg - 1) / tpg;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(cq::k_cq_upd<CQ_NT>, dim3(ntg, n, 1), dim3(CQ_NT), cq::UPD_LDS_DOUBLES * 8, 0, dP, (int64_t)4096, 0, 0, 0, tpg, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) {
        const double fl = 2.0 * 2.0 * 256 * 64 * 16 * (double)ntl * n;     // W0 and C -= V W
        printf("ld %d ntl %d tpg %2d (%4d workgroups of %d threads): %8.1f us  %6.2f TFLOP/s\n", ld, ntl, tpg, ntg * n, nthr, ms * 1e3, fl / ms * 1e-9);
#ifdef CQ_TRACE
        {
          const int nw = CQ_NT / 64, ntile = tpg / nw;
          std::vector<unsigned long long> tr((size_t)ntg * n * nw * 64 * 5);
          hipMemcpy(tr.data(), dTr, tr.size() * 8, hipMemcpyDeviceToHost);
          double sum[6] = {}; long cnt = 0;
          for (long w = 0; w < (long)ntg * n * nw; w++)
            for (int k = 0; k < ntile; k++) {
              const unsigned long long* q = &tr[(w * 64 + k) * 5];
              for (int j = 0; j < 4; j++) sum[j] += (double)(q[j + 1] - q[j]);
              if (k + 1 < ntile) sum[4] += (double)(q[5] - q[4]);
              cnt++;
            }
          printf("    per tile and wave (cycles): request -> arrived %7.0f   phase A %7.0f   phase B %7.0f   phase C %7.0f   C end -> next request %6.0f   (MFMA issue 16384 / 2560 / 16384)\n",
                 sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt);
          // one workgroup's timeline: wave 0..nw-1 of workgroup 0, tile 1
          const unsigned long long base = tr[0];
          { unsigned long long last = 0; for (long w = 0; w < (long)ntg * n * nw; w++) { const unsigned long long e = tr[(w * 64 + ntile - 1) * 5 + 4]; if (e > last) last = e; }
            printf("      wg 0 wave 0, every tile (request, arrived, A, B, C): "); for (int k = 0; k < ntile; k++) { const unsigned long long* q = &tr[(long)k * 5]; printf(" [%lld %lld %lld %lld %lld]", (long long)(q[0] - base), (long long)(q[1] - base), (long long)(q[2] - base), (long long)(q[3] - base), (long long)(q[4] - base)); }
            printf("\n      last stamp of the launch: %lld cycles after wg 0's first\n", (long long)(last - base)); }
          for (int w = 0; w < nw; w++) { const unsigned long long* q = &tr[((long)w * 64 + 1) * 5]; printf("      wg 0 wave %d tile 1: %8lld %8lld %8lld %8lld %8lld\n", w, (long long)(q[0] - base), (long long)(q[1] - base), (long long)(q[2] - base), (long long)(q[3] - base), (long long)(q[4] - base)); }
        }
#endif
      }
    }
  }
  return 0;
}
