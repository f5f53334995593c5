// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the engine's access shapes (round-2 verdict item 6):
// stream a known byte count with 8-, 16- and 32-byte-per-lane loads (and 8- / 32-byte stores) over a buffer far larger than
// the Infinity Cache, one kernel name per shape, so that the counter per kernel can be divided by the bytes.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/_fetch_probe.bin tools/probes/fetch_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- tools/_fetch_probe.bin      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <class T> __device__ double sum(T v);
template <> __device__ double sum(double v) { return v; }
template <> __device__ double sum(d2 v) { return v[0] + v[1]; }
template <> __device__ double sum(d4 v) { return v[0] + v[1] + v[2] + v[3]; }
#define LOADK(NAME, T)                                                                                      \
  __global__ void NAME(const T* p, size_t n, double* out) {                                                 \
    double s = 0;                                                                                           \
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) \
      s += sum(p[i]);                                                                                       \
    if (s == 1.2345) out[0] = s;                                                                            \
  }
LOADK(load_8B_per_lane, double)
LOADK(load_16B_per_lane, d2)
LOADK(load_32B_per_lane, d4)
__global__ void store_8B_per_lane(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
__global__ void store_32B_per_lane(d4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = d4{1, 2, 3, 4};
}
int main() {
  const size_t bytes = (size_t)4 << 30;          // 4 GiB: 16 x the Infinity Cache
  double* p; double* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  hipMemset(p, 0, bytes);
  hipLaunchKernelGGL(load_8B_per_lane, dim3(2048), dim3(256), 0, 0, p, bytes / 8, out);
  hipLaunchKernelGGL(load_16B_per_lane, dim3(2048), dim3(256), 0, 0, (const d2*)p, bytes / 16, out);
  hipLaunchKernelGGL(load_32B_per_lane, dim3(2048), dim3(256), 0, 0, (const d4*)p, bytes / 32, out);
  hipLaunchKernelGGL(store_8B_per_lane, dim3(2048), dim3(256), 0, 0, p, bytes / 8);
  hipLaunchKernelGGL(store_32B_per_lane, dim3(2048), dim3(256), 0, 0, (d4*)p, bytes / 32);
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel\n", bytes);
  return 0;
}
