// f64 MFMA dependent-issue probe (standalone tool, not part of the library): one workgroup of `wpb` waves issues
// v_mfma_f64_16x16x4_f64 in NCH interleaved accumulation chains (chain c: acc[c] = mfma(a, b, acc[c])), so that a
// result is needed again after NCH - 1 independent MFMAs.  Prints GFLOP/s per CU (peak 307 at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void __launch_bounds__(512) probe(double* out, int niter) {
  d4 acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < niter; it++) {
#pragma unroll
    for (int r = 0; r < 8 / NCH; r++)
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    a += 1e-9;
  }
  double s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(double* d, int wpb, int blocks) {
  const int niter = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(64 * wpb), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(probe<NCH>, dim3(blocks), dim3(64 * wpb), 0, 0, d, niter); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * wpb * niter * 8 * 2048.0;
  printf("chains %d waves/block %d blocks %3d: %7.2f ms  %7.1f GFLOP/s per workgroup\n", NCH, wpb, blocks, ms, fl / ms / 1e6 / blocks);
}
int main() {
  double* d; hipMalloc(&d, 8 * 256 * 4096);
  for (int blocks : {1, 256}) for (int wpb : {4, 8}) { run<1>(d, wpb, blocks); run<2>(d, wpb, blocks); run<4>(d, wpb, blocks); run<8>(d, wpb, blocks); }
  return 0;
}
