// Do fp64 VALU instructions and fp64 MFMAs of DIFFERENT waves on one SIMD overlap on gfx950?
// 512-thread workgroup, one per CU: waves 0-3 run a v_fma_f64 stream, waves 4-7 a v_mfma_f64_16x16x4 stream.
//   mode 1: VALU waves only   mode 2: MFMA waves only   mode 3: both   mode 4: f32 VALU + f64 MFMA   mode 5: both f64, VALU stream with s_nop gaps
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/_dp_pipe_probe.bin tools/probes/dp_pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, double* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool valu = wave < 4;
  double acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (valu && (mode & 1)) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double m = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = fma(a0, m, b); a1 = fma(a1, m, b); a2 = fma(a2, m, b); a3 = fma(a3, m, b);
        a4 = fma(a4, m, b); a5 = fma(a5, m, b); a6 = fma(a6, m, b); a7 = fma(a7, m, b);
        if (mode == 5) asm volatile("s_nop 15\n s_nop 15");
      }
    }
    acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else if (valu && mode == 4) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float m = 1.0000001f, b = 1e-9f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = fmaf(a0, m, b); a1 = fmaf(a1, m, b); a2 = fmaf(a2, m, b); a3 = fmaf(a3, m, b);
        a4 = fmaf(a4, m, b); a5 = fmaf(a5, m, b); a6 = fmaf(a6, m, b); a7 = fmaf(a7, m, b);
      }
    }
    acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else if (!valu && (mode & 2 || mode == 4 || mode == 5)) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1e-3 * threadIdx.x, b = 1e-3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  for (int nb : {1, 256})
    for (int mode : {1, 2, 3, 4, 5}) {
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, mode, iters, out, cyc);
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      const double nv = 64.0 * iters, nm = 16.0 * iters;
      printf("blocks %3d mode %d: VALU wave %.1f cycles per v_fma (%s), MFMA wave %.1f cycles per v_mfma_f64_16x16x4\n", nb, mode,
             h[0] / nv, mode == 4 ? "f32" : "f64", h[4] / nm);
    }
  return 0;
}
