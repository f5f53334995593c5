// f64 MFMA issue-rate probe (standalone tool, not part of the library): every wave issues NITER x 8
// independent v_mfma_f64_16x16x4_f64 on registers.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) probe(double* out, int niter) {
  d4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < niter; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    a += 1e-9;
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 8 * 256 * 4096);
  for (int wpb : {4, 8}) for (int bpc : {1, 2, 4}) {
    int blocks = 256 * bpc, niter = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * wpb), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * wpb), 0, 0, d, niter); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * wpb * niter * 8 * 2048.0;
    printf("waves/block %d blocks/CU %d: %.2f ms  %.2f TFLOP/s\n", wpb, bpc, ms, fl / ms / 1e9);
  }
  return 0;
}
