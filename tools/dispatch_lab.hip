// dispatch_lab.hip -- development harness: how single-wave workgroups spread over the SIMDs of the
// chip.  Each workgroup runs a fixed f64 chain (issue-bound when two waves share a SIMD); the time
// of a launch of N workgroups shows at which N waves start to share SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int LDSB, int CHAINS> __global__ void __launch_bounds__(64) k(double* out, int iters, double a) {
  __shared__ double pad[LDSB / 8];
  double d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) d[i] = a + i + threadIdx.x;
  pad[threadIdx.x] = a;
  for (int it = 0; it < iters / 32; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (u & 1) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(a));
        else asm volatile("v_min_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(a));
      }
    }
  }
  double s = pad[(threadIdx.x + 1) & 63];
  for (int i = 0; i < CHAINS; ++i) s += d[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  double* out; CK(hipMalloc(&out, 64 * 8192 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name, int chains) {
    for (int blocks : {256, 1024, 2048, 4096}) {
      const int iters = 8192 / chains;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-28s blocks %5d  %8.1f us  (%.2f ns per op per wave)\n", name, blocks, ms * 1e3, ms * 1e6 / 8192);
    }
  };
  run(k<1024, 1>, "1 chain, 1 KB LDS", 1);
  run(k<1024, 4>, "4 chains, 1 KB LDS", 4);
  run(k<1024, 2>, "2 chains", 2);
  run(k<1024, 8>, "8 chains", 8);
  return 0;
}
