// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950 as a function of (i) the number of accumulators taking turns (NCH; 1 = a
// dependent chain), (ii) the number of distinct A and B source registers cycling (NA, NB), (iii) waves per SIMD.
// 256 workgroups (one per CU).  Prints shader cycles per MFMA and SIMD (64 = the pipe's rate) from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/mfma_operand_probe.hip -o tools/_mfma_operand.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH, int NA, int NB>
__global__ void __launch_bounds__(512) k(double* out, unsigned long long* cyc, int iters) {
  d4 acc[NCH];
  double a[NA], b[NB];
  for (int i = 0; i < NA; i++) a[i] = 1.0 + 1e-3 * (threadIdx.x * 7 + i);
  for (int i = 0; i < NB; i++) b[i] = 0.5 - 1e-3 * (threadIdx.x * 3 + i);
  for (int c = 0; c < NCH; c++) acc[c] = d4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r % NA], b[(r / (NA > 1 ? 1 : 1)) % NB], acc[r % NCH], 0, 0, 0);
    for (int i = 0; i < NA; i++) asm volatile("" : "+v"(a[i]));
    for (int i = 0; i < NB; i++) asm volatile("" : "+v"(b[i]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  d4 s = acc[0];
  for (int c = 1; c < NCH; c++) s += acc[c];
  if (s[0] == 1.2345e301) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// the same dependent chain (one accumulator) with KV independent VALU instructions behind every MFMA:
// KIND 0 v_add_u32, 1 v_accvgpr_write_b32 + v_accvgpr_read_b32 pairs, 2 ds_read_b64 (LDS operand fetches, waited for every 8 MFMAs)
template <int KV, int KIND>
__global__ void __launch_bounds__(512) kv(double* out, unsigned long long* cyc, int iters) {
  __shared__ double sh[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sh[i] = 1.0 + i;
  __syncthreads();
  d4 acc = d4{0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = 1.0 + 1e-3 * (threadIdx.x * 7 + i); b[i] = 0.5 - 1e-3 * (threadIdx.x * 3 + i); }
  unsigned x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  double ld[8];
  for (int i = 0; i < 8; i++) ld[i] = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r & 3], b[(r >> 1) & 3], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < KV; q++) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[q & 7]) : "v"(x[(q + 1) & 7]));
        if (KIND == 1) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1" : "+v"(x[q & 7]) : : "a0");
        if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(ld[q & 7]) : "v"((unsigned)((threadIdx.x * 8 + q * 512) & 8191)));
      }
      if (KIND == 2 && (r & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned xs = 0; for (int i = 0; i < 8; i++) xs += x[i];
  double ls = 0; for (int i = 0; i < 8; i++) ls += ld[i];
  if (acc[0] == 1.2345e301 || xs == 0x12345678u || ls == 1.2345e301) out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + xs;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KV, int KIND>
void runv(double* d, unsigned long long* dc) {
  const int iters = 2000;
  for (int thr : {256, 512}) {
    hipLaunchKernelGGL((kv<KV, KIND>), dim3(256), dim3(thr), 0, 0, d, dc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kv<KV, KIND>), dim3(256), dim3(thr), 0, 0, d, dc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 32;          // MFMAs per wave
    static const char* kn[3] = {"v_add_u32", "v_accvgpr_write + read", "ds_read_b64"};
    printf("%d x %-22s behind every MFMA, waves/SIMD %d: %6.1f cycles per MFMA of a wave  %6.2f TFLOP/s\n", KV, kn[KIND], thr / 256, (double)c / nm,
           256.0 * 4 * nm * (thr / 256) * 2048 / ms * 1e-9);
  }
}
template <int NCH, int NA, int NB>
void run(double* d, unsigned long long* dc) {
  const int iters = 2000;
  for (int thr : {256, 512}) {
    hipLaunchKernelGGL((k<NCH, NA, NB>), dim3(256), dim3(thr), 0, 0, d, dc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NCH, NA, NB>), dim3(256), dim3(thr), 0, 0, d, dc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 32 * (thr / 256);          // MFMAs per SIMD
    printf("acc %d  A regs %2d  B regs %2d  waves/SIMD %d: %6.1f cycles per MFMA and SIMD  %6.2f TFLOP/s\n", NCH, NA, NB, thr / 256, (double)c / nm,
           256.0 * 4 * nm * 2048 / ms * 1e-9);
  }
}
int main() {
  double* d; unsigned long long* dc; hipMalloc(&d, 8192); hipMalloc(&dc, 64);
  runv<0, 0>(d, dc); runv<1, 0>(d, dc); runv<2, 0>(d, dc); runv<4, 0>(d, dc); runv<8, 0>(d, dc); runv<12, 0>(d, dc);
  runv<1, 1>(d, dc); runv<2, 1>(d, dc); runv<4, 1>(d, dc);
  runv<1, 2>(d, dc); runv<2, 2>(d, dc);
  run<1, 1, 1>(d, dc); run<1, 4, 4>(d, dc); run<1, 16, 16>(d, dc); run<1, 16, 1>(d, dc); run<1, 1, 16>(d, dc);
  run<2, 1, 1>(d, dc); run<2, 16, 16>(d, dc);
  run<4, 1, 1>(d, dc); run<4, 4, 4>(d, dc); run<4, 16, 16>(d, dc);
  run<8, 1, 1>(d, dc); run<8, 16, 16>(d, dc); run<8, 16, 1>(d, dc); run<8, 1, 16>(d, dc);
  run<16, 1, 1>(d, dc); run<16, 16, 16>(d, dc);
  return 0;
}
