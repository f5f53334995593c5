// Throughput of the 2-pass, V-in-LDS tile update of tools/probes/caqr.h (caqr::update_tile) with all eight waves of a
// workgroup updating tiles (two per SIMD, no panel work beside them): the trailing-update building block a grid-level
// CAQR of the batched sweep would use.  Every workgroup streams its own (rows x 64 + ntile*16)-column matrix.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I matrixproductbp.jl_amd/csrc -I tools/probes -o tools/_caqr_update_probe.bin tools/probes/caqr_update_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "caqr.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <bool TOP>
__global__ __launch_bounds__(512) void upd_kernel(double* Yall, long ld, int nchunks, int ntile, long mstride, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  ldbl* V = (ldbl*)lds + caqr::L_V;
  ldbl* OPS = (ldbl*)lds + caqr::L_OPS;
  for (int i = threadIdx.x; i < 16384 + 2560; i += 512) lds[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  __syncthreads();
  gdbl* Y = (gdbl*)(Yall + (long)blockIdx.x * mstride);
  const int wave = threadIdx.x >> 6;
  for (int r = 0; r < reps; r++)
    for (int ch = 0; ch < nchunks; ch++)
      for (int t = wave; t < ntile; t += 8)
        caqr::update_tile<16, TOP>(Y, ld, 64 + 256 * ch, 64 + 16 * t, 0, 4, V, OPS);
}

int main(int argc, char** argv) {
  const int rows = 1600 + 64, ntile = argc > 1 ? atoi(argv[1]) : 21, reps = 4;
  const int nchunks = 6;
  const long ld = 1696, cols = 64 + 16 * ntile;
  const long mstride = ld * cols;
  for (int nb : {1, 256}) {
    double* dY;
    CK(hipMalloc(&dY, sizeof(double) * mstride * nb));
    CK(hipMemset(dY, 0, sizeof(double) * mstride * nb));
    const int ldsb = caqr::L_TOTAL * 8;
    for (int top = 0; top < 2; top++) {
      auto kern = top ? upd_kernel<true> : upd_kernel<false>;
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(kern, dim3(nb), dim3(512), ldsb, 0, dY, ld, nchunks, ntile, mstride, 1);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(nb), dim3(512), ldsb, 0, dY, ld, nchunks, ntile, mstride, reps);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double tiles = (double)nb * reps * nchunks * ntile;
      const double mf = tiles * (256 + 40 + 256) * 2048.0;
      const double bytes = tiles * (256 * 16 * 8.0 * 2 + (top ? 64 * 16 * 8.0 * 2 : 0));
      printf("blocks %3d  TOP %d  %d tiles/chunk: %.3f us per tile per workgroup, %.1f TFLOP/s (%.0f %% of %d CUs' MFMA peak), %.2f TB/s\n", nb, top, ntile,
             ms * 1e3 / (reps * nchunks * ntile), mf / (ms * 1e-3) / 1e12, 100 * mf / (ms * 1e-3) / (78.6e12 * nb / 256.0), nb, bytes / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(dY));
  }
  return 0;
}
