// k2_floors.hip -- the memory-access floor of the footprint-collision kernel, measured with no
// arithmetic: n poses as three f64 streams from HBM and one flag byte out (the algorithmic 25 B per
// pose), plus, per lane, V gathers of 16 B from one random 128-B line of a 2 MiB table (the
// L2-resident obstacle lattice of the 2048^2 benchmark map).  Prints one JSON object:
//   stream_only_ms             what any kernel over these arrays pays
//   stream_plus_one_line_ms    what any kernel that looks the obstacles up per pose in L2 pays
//   stream_plus_four_gathers_ms  the access pattern of k_collision itself (four 16-B block rows of one line)
// usage: k2_floors [n_poses] [repeats]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int V16>
__global__ void __launch_bounds__(256) k_floor(uint64_t n, const uint4* __restrict__ t, uint32_t nlines,
                                               const double* __restrict__ x, const double* __restrict__ y,
                                               const double* __restrict__ z, uint8_t* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double xv = x[i], yv = y[i], zv = z[i];
    const uint64_t h = mix(i * 0x9E3779B97F4A7C15ull + (uint64_t)(xv + yv + zv));
    const uint32_t line = (uint32_t)(h % nlines);
    unsigned acc = (unsigned)h;  // keeps the stream loads alive without a gather
#pragma unroll
    for (int b = 0; b < V16; ++b) {
      const uint4 q = t[(size_t)line * 8 + b];
      acc |= q.x ^ q.y ^ q.z ^ q.w;
    }
    out[i] = (uint8_t)acc;
  }
}

template <int V16>
static double run(int reps, unsigned blocks, uint64_t n, const uint4* t, uint32_t nlines, const double* x, const double* y,
                  const double* z, uint8_t* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_floor<V16>, dim3(blocks), dim3(256), 0, 0, n, t, nlines, x, y, z, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float m = 0;
    CK(hipEventElapsedTime(&m, a, b));
    if (r >= 3) ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 24);
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const uint32_t nlines = (2u << 20) / 128;
  double *x, *y, *z;
  uint8_t* out;
  uint4* t;
  CK(hipMalloc(&x, n * 8));
  CK(hipMalloc(&y, n * 8));
  CK(hipMalloc(&z, n * 8));
  CK(hipMalloc(&out, n));
  CK(hipMalloc(&t, (size_t)nlines * 128));
  CK(hipMemset(x, 0x3c, n * 8));
  CK(hipMemset(y, 0x3d, n * 8));
  CK(hipMemset(z, 0x3e, n * 8));
  CK(hipMemset(t, 0x55, (size_t)nlines * 128));
  double best[3] = {1e9, 1e9, 1e9};
  for (unsigned blocks : {1024u, 2048u, 8192u}) {  // the floor is the best launch shape
    best[0] = std::min(best[0], run<0>(reps, blocks, n, t, nlines, x, y, z, out));
    best[1] = std::min(best[1], run<1>(reps, blocks, n, t, nlines, x, y, z, out));
    best[2] = std::min(best[2], run<4>(reps, blocks, n, t, nlines, x, y, z, out));
  }
  printf("{\"poses\": %llu, \"stream_only_ms\": %.4f, \"stream_plus_one_line_ms\": %.4f, \"stream_plus_four_gathers_ms\": %.4f}\n",
         (unsigned long long)n, best[0], best[1], best[2]);
  return 0;
}
