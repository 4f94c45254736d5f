// pp_collision.hip -- batched footprint collision (is_collision, hybrid_astar.rs:417-448).
//
// One pose per lane, structure-of-arrays fp64 input (3 coalesced 512-B loads per wave), one
// byte of flags out.  The obstacle set is either the tiled bit-packed lattice (PpLattice) or
// the bucketed point list (PpBuckets); see pp_device.h for the predicate.
#include "pp_device.h"

// ROWS: window rows of the lattice filter this instance is compiled for (8 or 16; unused without a lattice)
template <bool LATTICE, int ROWS>
__global__ void __launch_bounds__(128)
k_collision(uint64_t n, const double* __restrict__ x, const double* __restrict__ y,
            const double* __restrict__ yaw, uint8_t* __restrict__ flags, PpFootprint f,
            PpLattice lat, PpBuckets bk) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the next pose is fetched while the current one is tested
  double nx = x[i], ny = y[i], nyaw = yaw[i];
  for (; i < n; i += stride) {
    const double px = nx, py = ny, pyaw = nyaw;
    const uint64_t j = i + stride;
    if (j < n) {
      nx = x[j];
      ny = y[j];
      nyaw = yaw[j];
    }
    unsigned r;
    if (LATTICE) {
      r = pp_pose_yaw_vs_lattice<ROWS>(px, py, pyaw, f, lat);
    } else {
      double s, c;
      sincos(pyaw, &s, &c);
      r = pp_pose_vs_buckets(px, py, c, s, f, bk);
    }
    flags[i] = (uint8_t)r;
  }
}

static int32_t launch_collision(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t n,
                                const double* d_x, const double* d_y, const double* d_yaw,
                                uint8_t* d_flags) {
  if (n == 0) return PP_OK;
  PpFootprint f = pp_make_footprint(conf);
  // launch geometry from a sweep on 16 M poses: 128-thread blocks, 256 waves per CU in the grid
  // (193 us; 256-thread blocks 197 us; a quarter of the waves 205 us; four times as many 204 us)
  const unsigned block = 128;
  const unsigned grid = (unsigned)std::max<uint64_t>(
      1, std::min<uint64_t>((n + block - 1) / block, (uint64_t)ctx->num_cus * 128));
  if (map->lattice && f.scan_small)
    hipLaunchKernelGGL((k_collision<true, 8>), dim3(grid), dim3(block), 0, ctx->stream, n, d_x, d_y,
                       d_yaw, d_flags, f, map->lat, map->bk);
  else if (map->lattice)
    hipLaunchKernelGGL((k_collision<true, 16>), dim3(grid), dim3(block), 0, ctx->stream, n, d_x, d_y,
                       d_yaw, d_flags, f, map->lat, map->bk);
  else
    hipLaunchKernelGGL((k_collision<false, 16>), dim3(grid), dim3(block), 0, ctx->stream, n, d_x, d_y,
                       d_yaw, d_flags, f, map->lat, map->bk);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

static int32_t check_args(pp_ctx* ctx, const pp_map* map, const pp_config* conf) {
  if (!ctx || !map || !conf) return PP_ERR_INVALID;
  if (map->device != ctx->device) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  return PP_OK;
}

extern "C" int32_t pp_collision_batch_dev(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                          uint64_t n, const double* d_x, const double* d_y,
                                          const double* d_yaw, uint8_t* d_flags) {
  int32_t st = check_args(ctx, map, conf);
  if (st != PP_OK) return st;
  if (n && (!d_x || !d_y || !d_yaw || !d_flags)) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_collision(ctx, map, conf, n, d_x, d_y, d_yaw, d_flags);
}

extern "C" int32_t pp_collision_batch(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                      uint64_t n, const double* x, const double* y,
                                      const double* yaw, uint8_t* flags) {
  int32_t st = check_args(ctx, map, conf);
  if (st != PP_OK) return st;
  if (n == 0) return PP_OK;
  if (!x || !y || !yaw || !flags) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  size_t nb = n * sizeof(double);
  size_t nb_al = (nb + 255) & ~(size_t)255;
  st = pp_scratch_reserve(ctx, 3 * nb_al + n);
  if (st != PP_OK) return st;
  char* base = (char*)ctx->scratch;
  double* d_x = (double*)base;
  double* d_y = (double*)(base + nb_al);
  double* d_yaw = (double*)(base + 2 * nb_al);
  uint8_t* d_flags = (uint8_t*)(base + 3 * nb_al);
  PP_HIP(ctx, hipMemcpyAsync(d_x, x, nb, hipMemcpyHostToDevice, ctx->stream));
  PP_HIP(ctx, hipMemcpyAsync(d_y, y, nb, hipMemcpyHostToDevice, ctx->stream));
  PP_HIP(ctx, hipMemcpyAsync(d_yaw, yaw, nb, hipMemcpyHostToDevice, ctx->stream));
  st = launch_collision(ctx, map, conf, n, d_x, d_y, d_yaw, d_flags);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipMemcpyAsync(flags, d_flags, n, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

extern "C" int32_t pp_is_collision(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                   uint64_t n, const double* x, const double* y, const double* yaw,
                                   int32_t* hit, uint64_t* ambiguous) {
  if (!hit) return PP_ERR_INVALID;
  std::vector<uint8_t> flags(n);
  int32_t st = pp_collision_batch(ctx, map, conf, n, x, y, yaw, flags.data());
  if (st != PP_OK) return st;
  // the reference walks the list in order and panics at the first non-finite query point it
  // reaches before a hit (hybrid_astar.rs:422-445)
  *hit = 0;
  uint64_t amb = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (flags[i] & PP_F_NONFINITE) return PP_ERR_NONFINITE;
    if (flags[i] & PP_F_AMBIG) ++amb;
    if (flags[i] & PP_F_HIT) {
      *hit = 1;
      break;
    }
  }
  if (ambiguous) *ambiguous = amb;
  return PP_OK;
}
