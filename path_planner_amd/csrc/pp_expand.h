// pp_expand.h -- launch parameters of the motion-primitive kernel, shared with the planner.
#ifndef PP_EXPAND_H_
#define PP_EXPAND_H_

#include "pp_internal.h"

struct PpExpandParams {
  double step;        // conf.xy_reso * 2.0 (hybrid_astar.rs:260)
  double move_step, wheel_base;
  double xyreso, yawreso;
  uint64_t minx, miny, maxx, maxy;
  double backward_cost, gear_cost, steer_angle_cost, steer_change_cost;
  uint32_t nlist;       // ceil(step / move_step) (:262)
  uint32_t check_step;  // conf.collision_check_step
};

// Device output arrays, one entry per (parent, primitive).
struct PpExpandDev {
  uint8_t* valid;
  uint64_t* xind;
  uint64_t* yind;
  int64_t* yawind;
  double* cost;
  double* end_x;
  double* end_y;
  double* end_yaw;
  // optional, instead of the seven arrays above and fcost: one 64-byte record per (parent,
  // primitive) at rec + 8 t, written for valid primitives only -- xind, yind, yawind, cost, end x,
  // end y, end yaw, f as eight 8-byte words.  A reader that goes through the valid children of a
  // parent then touches one cache line per child instead of eight.
  uint64_t* rec = nullptr;
  // optional: calc_hybrid_cost (hybrid_astar.rs:491-493) of every valid node,
  // cost + h_cost * hmap[xind - minx][yind - miny], with the heuristic grid of the parent's query
  double* fcost = nullptr;
  const double* const* hmaps = nullptr;  // table of heuristic grids (x-major, hxw x hyw)
  const int32_t* hmap_id = nullptr;      // per parent: index into hmaps
  uint64_t hxw = 0, hyw = 0;
  double h_cost = 0.0;
  // optional: the grids are final only up to h_limit (a relaxation stopped there).  A valid node
  // whose grid value is above it, on a cell the raster hobs (same indexing) leaves free, gets
  // bit 4 of its valid byte: its f cost must not be used.
  const uint8_t* hobs = nullptr;
  double h_limit = 0.0;
  // optional, instead of hmaps: the grids as the relaxation leaves them, in its blocked working
  // layout (pp_blocked_value); grid p occupies stored columns p * hblk_plane_cols onwards
  const double* hblk = nullptr;
  const uint32_t* hblk_stamp = nullptr;  // per 32x32 block: 0 = never written, all +inf
  uint32_t hblk_nty = 0;
  uint64_t hblk_plane_cols = 0;
  // optional: entries of primitives that are not valid keep whatever the arrays held (only the
  // valid byte is written) -- for outputs that cross PCIe
  int32_t sparse = 0;
  // optional: per-query tables of the cells a search has already closed or holds open (key =
  // calc_index, hybrid_astar.rs:516-520; value = bit pattern of the open node's cost, 0 for a closed
  // cell; costs are non-negative, so the patterns order like the costs).  A valid node whose key is
  // in its query's table with a value not above its own cost is one the search drops
  // (hybrid_astar.rs:187, :199): it loses bit 0 of its valid byte and gets bit 5 instead.  The
  // tables may lag behind the search -- a cell is never re-opened and an open cost only falls, so an
  // old entry still rules out nothing the search would keep.
  const unsigned long long* ftab = nullptr;  // table q: ftab_slots (key, value) pairs from ftab + q * 2 * ftab_slots
  const int32_t* ftab_id = nullptr;          // per parent: its query's table, -1 = none
  uint64_t ftab_slots = 0;                   // a power of two
  int64_t f_minyaw = 0;                      // calc_index: Param.minyaw, xw, yw (minx, miny are in the launch parameters)
  uint64_t f_xw = 0, f_yw = 0;
};

// The tables above: open addressing, linear probing, at most kFtabProbes slots looked at (an entry
// that finds no room within them is simply not recorded).
constexpr int kFtabProbes = 16;
constexpr unsigned long long kFtabEmpty = ~0ull;
__host__ __device__ inline uint64_t pp_ftab_hash(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  return k;
}
struct PpFtabUpd {   // "cell `key` of table `tab` is now open at cost bits `value`" (0: closed)
  unsigned long long key, value;
  int32_t tab, pad_;
};
void pp_ftab_update_launch(pp_ctx* ctx, uint32_t n, const PpFtabUpd* upd, unsigned long long* ftab, uint64_t slots);

struct PpExpandPlan {
  uint32_t n_prim = 0;
  std::vector<double> motion;  // steer[n_prim], direc[n_prim], tan(steer)[n_prim]
  pp_param param{};
  PpExpandParams prm{};
  PpFootprint fp{};
};

int32_t pp_expand_prepare(pp_ctx* ctx, const pp_map* map, const pp_config* conf, PpExpandPlan* plan);
void pp_expand_launch(pp_ctx* ctx, const pp_map* map, const PpExpandPlan& plan, uint64_t n,
                      const double* d_px, const double* d_py, const double* d_pyaw,
                      const int64_t* d_pdir, const double* d_psteer, const double* d_pcost,
                      const double* d_motion, PpExpandDev out, uint8_t* d_parent_flag = nullptr);
// d_parent_flag: n bytes of device scratch, or null.  With it the parents' own poses are tested in
// a launch of their own (worth it from a few thousand parents on); without it every primitive
// tests its first pose itself.
constexpr uint64_t kParentPassMin = 4096;
int32_t pp_rollout_points_batch(pp_ctx* ctx, const pp_config* conf, uint32_t n, const double* in5,
                                double* x, double* y, double* yaw, uint32_t* nlist_out);

#endif
