// Probe: does a release/acquire flag at agent scope pass data between workgroups on different XCDs in plain
// hipMalloc memory?  WG b (b >= 1) waits for WG b-1's flag, checks its payload, then publishes its own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) chain(int* done, double* payload, int n, int* bad, long* spins_out) {
  const int b = blockIdx.x;
  if (b > 0) {
    if (threadIdx.x == 0) {
      long spins = 0;
      while (__hip_atomic_load(done + b - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(127);
        if (++spins > (1L << 18)) { atomicAdd(bad, 1000000); break; }
      }
      spins_out[b] = spins;
    }
    __syncthreads();
    __threadfence();
    for (int i = threadIdx.x; i < n; i += 512)
      if (payload[(long)(b - 1) * n + i] != (double)(b - 1) + i) atomicAdd(bad, 1);
  }
  // some work, then publish
  for (int i = threadIdx.x; i < n; i += 512) payload[(long)b * n + i] = (double)b + i;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(done + b, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
  const int nb = 64, n = 1 << 16;
  int *done, *bad; double* payload; long* spins;
  hipMalloc(&done, nb * 4); hipMalloc(&bad, 4); hipMalloc(&payload, sizeof(double) * nb * n); hipMalloc(&spins, 8 * nb);
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(done, 0, nb * 4); hipMemset(bad, 0, 4); hipMemset(payload, 0xff, sizeof(double) * nb * n); hipMemset(spins, 0, 8 * nb);
    hipLaunchKernelGGL(chain, dim3(nb), dim3(512), 0, 0, done, payload, n, bad, spins);
    hipError_t e = hipDeviceSynchronize();
    int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    std::vector<long> hs(nb); hipMemcpy(hs.data(), spins, 8 * nb, hipMemcpyDeviceToHost);
    long mx = 0; for (long v : hs) mx = v > mx ? v : mx;
    printf("rep %d: %s, bad = %d, max spins %ld\n", rep, hipGetErrorString(e), hb, mx);
  }
  return 0;
}
