// gather_lab.hip -- development harness: cost of per-lane 16-B gathers from a 544 KiB table
// (L2-resident) by placement of the addresses within 128-B lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// V: 1 = one block; 2 = blocks 0,1 of a line; 3 = blocks 0,4; 4 = two different lines;
//    5 = blocks 0..3; 6 = blocks 0..7; 7 = 8-byte loads x1 ; 8 = 4-byte x1; 9 = 4 different lines
// stream of three f64 arrays (HBM) + V16 gathers of 16 B from one random line per lane
template <int V16>
__global__ void __launch_bounds__(256) ks(uint64_t n, const uint4* __restrict__ t, uint32_t nlines, const double* __restrict__ x,
                                          const double* __restrict__ y, const double* __restrict__ z, uint8_t* out) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double xv = x[i], yv = y[i], zv = z[i];
    uint64_t h = mix(i * 0x9E3779B97F4A7C15ull + (uint64_t)(xv + yv + zv));
    uint32_t l0 = (uint32_t)(h % nlines);
    unsigned acc = (unsigned)h;  // keeps the stream loads alive when V16 == 0
#pragma unroll
    for (int b = 0; b < V16; ++b) { uint4 q = t[(size_t)l0 * 8 + b]; acc |= q.x ^ q.y ^ q.z ^ q.w; }
    out[i] = (uint8_t)acc;
  }
}
// the same with the stream loads one round ahead (PF >= 1) and the gather one round ahead (PF == 2)
template <int V16, int PF>
__global__ void __launch_bounds__(256) kp(uint64_t n, const uint4* __restrict__ t, uint32_t nlines, const double* __restrict__ x,
                                          const double* __restrict__ y, const double* __restrict__ z, uint8_t* out) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double nx = x[i], ny = y[i], nz = z[i];
  uint4 g[V16];
  auto addr = [&](uint64_t k, double a, double b, double c) { return (uint32_t)(mix(k * 0x9E3779B97F4A7C15ull + (uint64_t)(a + b + c)) % nlines); };
  if (PF == 2) {
    uint32_t l0 = addr(i, nx, ny, nz);
#pragma unroll
    for (int b = 0; b < V16; ++b) g[b] = t[(size_t)l0 * 8 + b];
    uint64_t j = i + stride;
    if (j < n) { nx = x[j]; ny = y[j]; nz = z[j]; }
  }
  for (; i < n; i += stride) {
    unsigned acc = 0;
    if (PF == 2) {
      uint4 c[V16];
#pragma unroll
      for (int b = 0; b < V16; ++b) c[b] = g[b];
      uint64_t j = i + stride, j2 = j + stride;
      if (j < n) {
        uint32_t l0 = addr(j, nx, ny, nz);
#pragma unroll
        for (int b = 0; b < V16; ++b) g[b] = t[(size_t)l0 * 8 + b];
        if (j2 < n) { nx = x[j2]; ny = y[j2]; nz = z[j2]; }
      }
#pragma unroll
      for (int b = 0; b < V16; ++b) acc |= c[b].x ^ c[b].y ^ c[b].z ^ c[b].w;
    } else {
      double xv = nx, yv = ny, zv = nz;
      uint64_t j = i + stride;
      if (j < n) { nx = x[j]; ny = y[j]; nz = z[j]; }
      uint32_t l0 = addr(i, xv, yv, zv);
#pragma unroll
      for (int b = 0; b < V16; ++b) { uint4 q = t[(size_t)l0 * 8 + b]; acc |= q.x ^ q.y ^ q.z ^ q.w; }
    }
    out[i] = (uint8_t)acc;
  }
}
template <int V>
__global__ void __launch_bounds__(256) k(uint64_t n, const uint4* __restrict__ t, uint32_t nlines, const double* __restrict__ x, uint8_t* out) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double xv = x[i];
    uint64_t h = mix(i * 0x9E3779B97F4A7C15ull + (uint64_t)(xv));
    uint32_t l0 = (uint32_t)(h % nlines), l1 = (uint32_t)((h >> 32) % nlines);
    uint32_t l2 = (uint32_t)((h >> 16) % nlines), l3 = (uint32_t)((h >> 8) % nlines);
    unsigned acc = 0;
    auto ld = [&](uint32_t line, int blk) { uint4 b = t[(size_t)line * 8 + blk]; acc |= b.x ^ b.y ^ b.z ^ b.w; };
    if (V == 1) ld(l0, 0);
    if (V == 2) { ld(l0, 0); ld(l0, 1); }
    if (V == 3) { ld(l0, 0); ld(l0, 4); }
    if (V == 4) { ld(l0, 0); ld(l1, 0); }
    if (V == 5) { ld(l0, 0); ld(l0, 1); ld(l0, 2); ld(l0, 3); }
    if (V == 6) { for (int b = 0; b < 8; ++b) ld(l0, b); }
    if (V == 7) { uint2 b = ((const uint2*)t)[(size_t)l0 * 16]; acc |= b.x ^ b.y; }
    if (V == 8) { acc |= ((const uint32_t*)t)[(size_t)l0 * 32]; }
    if (V == 9) { ld(l0, 0); ld(l1, 0); ld(l2, 0); ld(l3, 0); }
    auto ldn = [&](uint32_t line, int blk) { typedef unsigned v4u __attribute__((ext_vector_type(4))); v4u b = __builtin_nontemporal_load((const v4u*)&t[(size_t)line * 8 + blk]); acc |= b.x ^ b.y ^ b.z ^ b.w; };
    auto lds1 = [&](uint32_t line, int blk) {
      const unsigned long long* q = (const unsigned long long*)&t[(size_t)line * 8 + blk];
      unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc |= (unsigned)(a ^ b) ^ (unsigned)((a ^ b) >> 32); };
    if (V == 10) ldn(l0, 0);
    if (V == 11) { ldn(l0, 0); ldn(l1, 0); ldn(l2, 0); ldn(l3, 0); }
    if (V == 12) { ldn(l0, 0); ldn(l0, 1); ldn(l0, 2); ldn(l0, 3); }
    if (V == 13) lds1(l0, 0);
    if (V == 14) { lds1(l0, 0); lds1(l1, 0); lds1(l2, 0); lds1(l3, 0); }
    if (V == 15) { lds1(l0, 0); lds1(l0, 1); lds1(l0, 2); lds1(l0, 3); }
    out[i] = (uint8_t)acc;
  }
}
int main() {
  uint64_t n = 1 << 24; uint32_t nlines = 4356;  // 66x66 tiles of 128 B = 544.5 KiB
  uint4* t; double* x; uint8_t* o;
  CK(hipMalloc(&t, (size_t)nlines * 128)); CK(hipMemset(t, 1, (size_t)nlines * 128));
  CK(hipMalloc(&x, n * 8)); CK(hipMemset(x, 0, n * 8)); CK(hipMalloc(&o, n));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, 0, n, t, nlines, x, o);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, 0, n, t, nlines, x, o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-40s %8.3f ms\n", name, ms);
  };
  {
    double *y, *z; CK(hipMalloc(&y, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMalloc(&z, n * 8)); CK(hipMemset(z, 0, n * 8));
    uint4* t2; uint32_t nl2 = 16384; CK(hipMalloc(&t2, (size_t)nl2 * 128)); CK(hipMemset(t2, 1, (size_t)nl2 * 128));
    auto runs = [&](auto kern, const char* name, int blocks) {
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, n, t2, nl2, x, y, z, o);
      CK(hipEventRecord(e0));
      for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, n, t2, nl2, x, y, z, o);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("%-52s %8.3f ms\n", name, ms);
    };
    for (int blocks : {1024, 2048, 8192}) {
      printf("-- 3 x f64 stream (400 MB) + gathers from a 2 MiB table, %d blocks\n", blocks);
      runs(ks<0>, "stream only", blocks);
      runs(ks<1>, "stream + 1 x 16B random line", blocks);
      runs(ks<2>, "stream + 2 x 16B same line", blocks);
      runs(ks<4>, "stream + 4 x 16B same line", blocks);
      runs(kp<1, 1>, "stream ahead + 1 x 16B", blocks);
      runs(kp<4, 1>, "stream ahead + 4 x 16B same line", blocks);
      runs(kp<1, 2>, "stream and gather ahead + 1 x 16B", blocks);
      runs(kp<4, 2>, "stream and gather ahead + 4 x 16B same line", blocks);
    }
  }
  run(k<0>, "no gather (8B read + 1B write)");
  run(k<8>, "1 x 4B");
  run(k<7>, "1 x 8B");
  run(k<1>, "1 x 16B");
  run(k<2>, "2 x 16B same 64B sector");
  run(k<3>, "2 x 16B same line, other half");
  run(k<4>, "2 x 16B different lines");
  run(k<5>, "4 x 16B same 64B half-line... (0..3)");
  run(k<6>, "8 x 16B whole line");
  run(k<9>, "4 x 16B different lines");
  run(k<10>, "nt 1 x 16B");
  run(k<11>, "nt 4 x 16B different lines");
  run(k<12>, "nt 4 x 16B same line");
  run(k<13>, "sc1 (2x8B) 1 block");
  run(k<14>, "sc1 4 blocks different lines");
  run(k<15>, "sc1 4 blocks same line");
  return 0;
}
