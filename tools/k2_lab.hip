// k2_lab.hip -- development harness (not part of the library): times ablated variants of the
// footprint-collision kernel to locate its bottleneck.  Build: hipcc -O3 --offload-arch=gfx950
// -ffp-contract=off -I path_planner_amd/csrc tools/k2_lab.hip -o tools/k2_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pp_device.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void k_fill(uint64_t n, double* x, double* y, double* yaw, double size) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t G = 0x9E3779B97F4A7C15ull;
  double u0 = (double)(mix(0xC0111DE1ull + (3 * i + 1) * G) >> 11) * 0x1p-53;
  double u1 = (double)(mix(0xC0111DE1ull + (3 * i + 2) * G) >> 11) * 0x1p-53;
  double u2 = (double)(mix(0xC0111DE1ull + (3 * i + 3) * G) >> 11) * 0x1p-53;
  x[i] = u0 * (size - 1); y[i] = u1 * (size - 1); yaw[i] = u2 * 6.283185307179586 - 3.141592653589793;
}

template <int V>
__global__ void __launch_bounds__(256)
k_var(uint64_t n, const double* __restrict__ x, const double* __restrict__ y,
      const double* __restrict__ yaw, uint8_t* __restrict__ flags, PpFootprint f, PpLattice lat) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double px = x[i], py = y[i], pyaw = yaw[i];
    unsigned r = 0;
    if (V == 0) {  // stream only
      r = (px + py + pyaw) > 1000.0;
    } else if (V == 1) {  // + sincos
      double s, c; sincos(pyaw, &s, &c);
      r = (px * c + py * s) > 1000.0;
    } else if (V == 2) {  // + gathers of the window blocks, OR-reduced
      double s, c; sincos(pyaw, &s, &c);
      double cx, cy; pp_footprint_centre(px, py, c, s, f, cx, cy);
      double ac = fabs(c), as = fabs(s);
      double hx = ac * f.bound_len + as * f.bound_wid + 1.0e-6;
      double hy = as * f.bound_len + ac * f.bound_wid + 1.0e-6;
      double gx = cx - (double)lat.x0, gy = cy - (double)lat.y0;
      int ix0 = (int)fmax(ceil(gx - hx), 0.0), ix1 = (int)fmin(floor(gx + hx), (double)(lat.nx - 1));
      int iy0 = (int)fmax(ceil(gy - hy), 0.0), iy1 = (int)fmin(floor(gy + hy), (double)(lat.ny - 1));
      unsigned acc = 0;
      for (int by = iy0 >> 3; by <= (iy1 >> 3); ++by)
        for (int bx = ix0 >> 4; bx <= (ix1 >> 4); ++bx) {
          size_t addr = ((size_t)(by >> 2) * (size_t)lat.tiles_x + (size_t)(bx >> 1)) * 8 + (size_t)((by & 3) * 2 + (bx & 1));
          uint4 b = lat.tiles[addr];
          acc |= b.x | b.y | b.z | b.w;
        }
      r = acc & 1;
    } else if (V == 3) {  // full current kernel
      double s, c; sincos(pyaw, &s, &c);
      r = pp_pose_vs_lattice(px, py, c, s, f, lat);
    }
    flags[i] = (uint8_t)r;
  }
}

int main(int argc, char** argv) {
  uint64_t n = 1 << 24; int size = 2048;
  double *x, *y, *yaw; uint8_t* fl;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&yaw, n * 8)); CK(hipMalloc(&fl, n));
  hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, n, x, y, yaw, (double)size);
  // grid: border + bernoulli 0.05 via host LCG (layout as pp_ctx.hip build_lattice)
  int nx = size + 64, tx = nx / 32;
  std::vector<uint32_t> words((size_t)tx * tx * 32, 0);
  uint64_t st = 12345;
  for (int gy = 32; gy < 32 + size; ++gy) for (int gx = 32; gx < 32 + size; ++gx) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    bool occ = ((st >> 33) % 100) < 5 || gx == 32 || gy == 32 || gx == 31 + size || gy == 31 + size;
    if (!occ) continue;
    size_t tile = (size_t)(gy >> 5) * tx + (gx >> 5);
    int bx = (gx >> 4) & 1, by = (gy >> 3) & 3, row = gy & 7, col = gx & 15;
    words[tile * 32 + (by * 2 + bx) * 4 + (row >> 1)] |= 1u << ((row & 1) * 16 + col);
  }
  uint4* tiles; CK(hipMalloc(&tiles, words.size() * 4));
  CK(hipMemcpy(tiles, words.data(), words.size() * 4, hipMemcpyHostToDevice));
  PpLattice lat{tiles, tx, tx, nx, nx, -32, -32};
  PpFootprint f{1.75, 4.15, 2.9, 5.062855, 25.6325, 1e-11};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, 0, n, x, y, yaw, fl, f, lat);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, 0, n, x, y, yaw, fl, f, lat);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-28s %8.3f ms  %7.2f Gposes/s  %6.1f GB/s\n", name, ms, n / ms / 1e6, n * 25.0 / ms / 1e6);
  };
  run(k_var<0>, "V0 stream only");
  run(k_var<1>, "V1 + sincos");
  run(k_var<2>, "V2 + block gathers");
  run(k_var<3>, "V3 full (current)");
  return 0;
}
