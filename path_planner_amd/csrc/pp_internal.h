// pp_internal.h -- host-side structures shared by the translation units of libpp_hip.so.
#ifndef PP_INTERNAL_H_
#define PP_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pp.h"

// ---------------------------------------------------------------- device-visible PODs

// Footprint rectangle of hybrid_astar.rs:418-421 plus the constants the kernels derive from it.
struct PpFootprint {
  double cp2center;   // (center2front - center2back) / 2
  double bound_len;   // half length + extend_bound
  double bound_wid;   // half width + extend_bound
  double bound_radius;  // hypot(bound_len, bound_wid)
  double r2_max;      // largest t with sqrt(t) <= bound_radius (sqrt correctly rounded):
                      // `sqrt(s) <= bound_radius`  <=>  `s <= r2_max`
  double guard;       // guard band for "a different libm could decide otherwise"
  // scanline raster of the footprint (pp_device.h): half sizes of the inflated rectangle, and
  // whether the footprint fits the 13-column scan window (bound_radius + inflation <= 6 cells)
  // (scan_lo, scan_wo) and of the deflated one (scan_li, scan_wi), rounded outward / inward
  float scan_lo, scan_wo, scan_li, scan_wi;
  int32_t scan_ok;
  int32_t scan_small;  // bound_radius + inflation <= 3 cells: an eight-row window holds the footprint
};

// Occupancy of an integer-lattice obstacle set, bit-packed for gather locality.
//   tile  = 28 columns x 32 rows = 128 B = one cache line: word r = row r, bit kLatticeShift + c
//   = column c (the low four bits of a row word are zero: the scan-line filter ors its row
//   number into them)
//   tiles start every 16 cells in both directions, so they overlap and every cell is stored in
//   up to four of them: tile (tx, ty) covers columns [16 tx, 16 tx + 28) and rows [16 ty, 16 ty + 32).
//   Any window of up to 13 columns x 17 rows lies inside the tile that starts at or before its
//   low corner, so a footprint test touches exactly one line however the window falls.
//   tile (tx, ty) = tiles[(tx * nty + ty) * 8 .. + 8]  (uint4 k of a tile = rows 4k .. 4k + 3)
// Cell (cx, cy) in lattice coordinates maps to grid coordinates (cx - x0, cy - y0); the grid is
// padded with a ring of 32 empty cells so windows near the rim need no per-load bounds test.
constexpr int kLatticeShift = 4;
struct PpLattice {
  const uint4* tiles;
  int32_t ntx, nty;
  int32_t nx, ny;      // cells including padding (multiples of 32)
  int64_t x0, y0;      // lattice coordinate of grid cell (0, 0)
};

// General (non-lattice) obstacle points in unit-square buckets, CSR by row-major bucket index;
// points of one bucket row are contiguous, so a window row is one [start, end) range.
struct PpBuckets {
  const uint32_t* start;  // nbx * nby + 1 entries
  const double* px;       // points sorted by bucket
  const double* py;
  int64_t bx0, by0;       // bucket (0,0) covers [bx0, bx0+1) x [by0, by0+1)
  int32_t nbx, nby;
};

// ---------------------------------------------------------------- host objects

struct pp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int last_hip_error = 0;
  int num_cus = 256;
  uint64_t max_expansions = 100000;  // per query, 0 = unlimited
  uint64_t heuristic_pass_blocks = (uint64_t)1 << 20;  // 32x32-cell blocks one relaxation of several grids may hold (8 KiB each)
  uint32_t host_threads = 15;        // worker threads pp_plan_batch may start beside the calling thread
  double heuristic_limit = -1.0;     // cost limit of the planner's first heuristic relaxation, < 0 = automatic
  // reusable device scratch (grown on demand, never shrunk)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // work lists of the heuristic relaxation (grown on demand, never shrunk)
  void* wave_buf = nullptr;
  size_t wave_bytes = 0;
  double wave_band = 128.0;  // cost band of tiles relaxed in one round
  void* wave_resume = nullptr;  // state of a relaxation stopped at a cost limit (owned by pp_heuristic.hip)
  void (*wave_resume_free)(void*) = nullptr;
  uint64_t wave_stats[4] = {0, 0, 0, 0};  // last relaxation: rounds, tile visits, sweep pairs, tiles
  uint32_t* wave_counts = nullptr;  // pinned: the round counters as last read back
  hipEvent_t wave_event = nullptr;  // that read-back
  void* ftab_buf = nullptr;  // the planner's per-query tables of known cells (pp_expand.h)
  size_t ftab_bytes = 0;
  size_t ftab_budget = (size_t)16 << 30;  // device memory those tables may take (pp_ctx_set_table_budget)
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // the planner's per-round device staging (grown on demand, never shrunk)
  void* plan_buf = nullptr;
  size_t plan_bytes = 0;
  // the planner's heuristic grids (one per distinct goal cell of the last batch)
  void* hmap_buf = nullptr;
  size_t hmap_bytes = 0;
  // the planner's per-query host state, kept between calls (owned by pp_planner.hip)
  void* plan_state = nullptr;
  void (*plan_state_free)(void*) = nullptr;
  uint64_t plan_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // last pp_plan_batch (pp_plan_batch_counters)
  // a large pp_plan_batch runs as several parts, each on a helper context of this device (owned)
  std::vector<pp_ctx*> helpers;
  bool is_helper = false;
  void* lent_pool = nullptr;  // while a part of a larger batch: the worker pool the parts share (pp_planner.hip)
  uint32_t batch_parts = 0;  // 0 = chosen from the batch size and host_threads (pp_ctx_set_batch_parts)
  // slab sessions created on this context and not yet destroyed (a list owned by pp_heuristic.hip);
  // pp_ctx_destroy releases their device memory and leaves the handles inert
  void* slabs = nullptr;
  void (*slabs_orphan)(pp_ctx*) = nullptr;
};

struct pp_map {
  pp_ctx* ctx = nullptr;  // creator (not dereferenced after creation)
  int device = 0;
  std::vector<double> ox, oy;  // the reference's lists, kept for calc_parameters / obsmap
  bool finite = true;
  bool lattice = false;
  double minx = 0, miny = 0, maxx = 0, maxy = 0;
  // device
  uint4* d_tiles = nullptr;
  PpLattice lat{};
  uint32_t* d_bstart = nullptr;
  double* d_bpx = nullptr;
  double* d_bpy = nullptr;
  PpBuckets bk{};
  double* d_ox = nullptr;  // obstacle lists on the device (for the obstacle-raster kernel)
  double* d_oy = nullptr;
};

#define PP_HIP(ctx, expr)                               \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) {                             \
      if (ctx) (ctx)->last_hip_error = (int)e_;         \
      return PP_ERR_HIP;                                \
    }                                                   \
  } while (0)

int32_t pp_scratch_reserve(pp_ctx* ctx, size_t bytes);
int32_t pp_wavefront_multi_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, const uint8_t* d_obs, uint32_t n,
                               const uint64_t* goal_x, const uint64_t* goal_y, double* d_cost,
                               uint64_t cost_stride, double limit = __builtin_inf(), bool* bounded = nullptr,
                               struct PpBlockedGrids* keep_blocked = nullptr);
int32_t pp_wavefront_multi_finish(pp_ctx* ctx);
// Where pp_wavefront_multi_dev left the grids when asked to keep them in the relaxation's blocked
// layout instead of writing the dense arrays (valid until the context's next relaxation).
struct PpBlockedGrids {
  const double* blk = nullptr;
  const uint32_t* stamp = nullptr;  // per block: 0 = never written (its cells are +inf, its memory is not initialised)
  uint32_t nty = 0;
  uint64_t plane_cols = 0;
};
// True when n grids of that size are relaxed in one pass (a precondition of keeping them blocked).
bool pp_wavefront_multi_single_pass(const pp_ctx* ctx, uint64_t xw, uint64_t yw, uint32_t n);
int32_t pp_pinned_reserve(pp_ctx* ctx, size_t bytes);
PpFootprint pp_make_footprint(const pp_config* conf);
int32_t pp_check_config(const pp_config* conf);

// heuristic building blocks (pp_heuristic.hip), device-resident forms used by the planner
int32_t pp_heuristic_checks(pp_ctx* ctx, const pp_map* map, double reso, uint64_t dims[6]);
int32_t pp_obsmap_dev(pp_ctx* ctx, const pp_map* map, double reso, double rr, const uint64_t dims[6],
                      uint8_t* d_obs);

// Rust `as usize` / `as isize` on f64 (saturating, NaN -> 0); host and device.
__host__ __device__ inline uint64_t pp_sat_usize(double v) {
  if (!(v == v)) return 0;
  if (v <= 0.0) return 0;
  if (v >= 18446744073709551615.0) return UINT64_MAX;
  return (uint64_t)v;
}
__host__ __device__ inline int64_t pp_sat_isize(double v) {
  if (!(v == v)) return 0;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  if (v >= 9223372036854775807.0) return INT64_MAX;
  return (int64_t)v;
}

// Launch geometry for one-item-per-lane streaming kernels: enough workgroups to fill 256 CUs
// several times over, grid-stride for the rest.
inline unsigned pp_stream_grid(const pp_ctx* ctx, uint64_t n, unsigned block) {
  uint64_t need = (n + block - 1) / block;
  // blocks per CU of a grid-stride launch: measured on the collision kernel 16 -> 206 us,
  // 32 and 64 -> 198 us, no cap -> 212 us (the rollout and Reeds-Shepp kernels gain 1-2 % too)
  uint64_t cap = (uint64_t)ctx->num_cus * 32;
  if (need > cap) need = cap;
  if (need == 0) need = 1;
  return (unsigned)need;
}

#endif
