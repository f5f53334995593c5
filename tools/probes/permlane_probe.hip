#include <hip/hip_runtime.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, const unsigned* in) {
  unsigned x = in[threadIdx.x], y = in[64 + threadIdx.x];
  u2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  u2 s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned h[128], o[256]; for (int i = 0; i < 128; i++) h[i] = i;
  unsigned *di, *dout; hipMalloc(&di, 512); hipMalloc(&dout, 1024); hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di); hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
  for (int a = 0; a < 4; a++) { for (int i = 0; i < 64; i++) printf("%u ", o[64 * a + i]); printf("\n"); }
}
