// pp_expand_device.h -- one motion primitive of one parent (calc_next_node + is_index_ok,
// hybrid_astar.rs:252-343) as a device function, shared by the rollout kernel and the planner's
// single-launch round.
#ifndef PP_EXPAND_DEVICE_H
#define PP_EXPAND_DEVICE_H
#include "pp_device.h"
#include "pp_expand.h"

__device__ __forceinline__ void pp_turn(double& c, double& s, double cd, double sd) {
  const double nc = __builtin_fma(c, cd, -(s * sd));
  const double ns = __builtin_fma(s, cd, c * sd);
  c = nc;
  s = ns;
}

// Cell (x, y) (stored column, row) of the heuristic relaxation's blocked array: 32x32-cell blocks
// of 1024 doubles, block (x >> 5, y >> 5) at index (x >> 5) * nty + (y >> 5); inside a block the
// 4x4 patch of cells (x & 3, y & 3) belongs to lane ((x & 31) >> 2) * 8 + ((y & 31) >> 2), cells
// 2m, 2m+1 of a lane are adjacent and the 64 lanes of a pair consecutive (blk_off in
// pp_heuristic.hip, which owns the layout).  stamp[t] == 0 marks a block the relaxation never
// wrote: its memory is uninitialised and every cell of it is +inf.
__device__ __forceinline__ double pp_blocked_value(const double* __restrict__ blk,
                                                   const uint32_t* __restrict__ stamp, uint32_t nty, uint64_t x,
                                                   uint64_t y) {
  const uint64_t t = (x >> 5) * nty + (y >> 5);
  if (stamp[t] == 0u) return __longlong_as_double(0x7FF0000000000000ll);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  const uint32_t k = (lx & 3u) * 4u + (ly & 3u), lane = (lx >> 2) * 8u + (ly >> 2);
  return blk[t * 1024u + ((k >> 1) * 64u + lane) * 2u + (k & 1u)];
}

// One record of news for the tables of known cells: claim the key's slot (or find it) and lower
// its value.  Records that name the same cell, and lookups that run at the same time, are fine:
// the atomics make the outcome the lowest value, and a lookup that catches a slot between the two
// steps sees a value above every cost, i.e. rules nothing out.
__device__ __forceinline__ void pp_ftab_update_one(const PpFtabUpd& u, unsigned long long* __restrict__ ftab,
                                                   uint64_t slots) {
  unsigned long long* tab = ftab + (uint64_t)u.tab * 2 * slots;
  uint64_t i = pp_ftab_hash(u.key) & (slots - 1);
  for (int probe = 0; probe < kFtabProbes; ++probe) {
    const unsigned long long old = atomicCAS(&tab[2 * i], kFtabEmpty, u.key);
    if (old == kFtabEmpty || old == u.key) {
      atomicMin(&tab[2 * i + 1], u.value);  // a fresh slot holds all ones
      return;
    }
    i = (i + 1) & (slots - 1);
  }
}

// Item t = parent t / n_prim, primitive t % n_prim.  pflag, when given, holds the result of the
// parent's own pose (the first pose of every primitive).
__device__ __forceinline__ void pp_expand_item(uint64_t t, uint32_t n_prim, const double* __restrict__ px,
                                               const double* __restrict__ py, const double* __restrict__ pyaw,
                                               const int64_t* __restrict__ pdir, const double* __restrict__ psteer,
                                               const double* __restrict__ pcost, const double* __restrict__ motion,
                                               const PpExpandParams& prm, const PpFootprint& f, const PpLattice& lat,
                                               const PpBuckets& bk, int use_lattice,
                                               const uint8_t* __restrict__ pflag, const PpExpandDev& out) {
  uint64_t p = t / n_prim;
  uint32_t i = (uint32_t)(t - p * n_prim);
  double u = motion[i], d = motion[n_prim + i], tanu = motion[2 * n_prim + i];
  const double cd = motion[3 * n_prim + i], sd = motion[4 * n_prim + i];
  double x = px[p], y = py[p], yaw = pyaw[p];
  double dm = d * prm.move_step;            // d * conf.move_step
  double dyaw = dm / prm.wheel_base * tanu;  // d * move_step / wheel_base * u.tan()
  unsigned fl = 0;
  bool hit = false;
  uint32_t next_check = 0;
  double s, c;
  sincos(yaw, &s, &c);
  for (uint32_t k = 0; k <= prm.nlist; ++k) {
    bool check = (k == next_check);
    if (check) {
      next_check += prm.check_step;
      unsigned r;
      if (k == 0 && pflag)  // uniform over the launch
        r = pflag[p];
      else
        r = use_lattice ? pp_pose_vs_lattice(x, y, c, s, f, lat) : pp_pose_vs_buckets(x, y, c, s, f, bk);
      hit |= (r & PP_F_HIT) && !(r & PP_F_AMBIG);
      fl |= r;
    }
    if (k < prm.nlist) {
      double ax = dm * c, ay = dm * s;
      x = x + ax;
      y = y + ay;
      yaw = yaw + dyaw;
      pp_turn(c, s, cd, sd);
    }
  }
  double vx = x / prm.xyreso, vy = y / prm.xyreso, vyaw = yaw / prm.yawreso;
  uint64_t xind = pp_sat_usize(round(vx));
  uint64_t yind = pp_sat_usize(round(vy));
  int64_t yawind = pp_sat_isize(round(vyaw));
  bool in_bounds = !(xind <= prm.minx || xind >= prm.maxx || yind <= prm.miny || yind >= prm.maxy);
  // rounding decisions within 1e-9 of a .5 tie could fall either way under another libm
  bool tie = (fabs(fabs(vx - trunc(vx)) - 0.5) < 1e-9) || (fabs(fabs(vy - trunc(vy)) - 0.5) < 1e-9) ||
             (fabs(fabs(vyaw - trunc(vyaw)) - 0.5) < 1e-9);
  bool collide = (fl & PP_F_HIT) != 0;
  bool amb = tie || (!hit && (fl & PP_F_AMBIG));
  bool valid = in_bounds && !collide;

  double step = prm.step;
  double cost = 0.0;
  int64_t direction;
  if (d > 0.0) {
    direction = 1;
    cost += fabs(step);
  } else {
    direction = -1;
    cost += fabs(step) * prm.backward_cost;
  }
  if (direction != pdir[p]) cost += prm.gear_cost;
  cost += prm.steer_angle_cost * fabs(u);
  cost += prm.steer_change_cost * fabs(psteer[p] - u);
  cost += pcost[p];

  // a node the search is known to drop: its cell is closed, or open at no higher cost
  bool known = false;
  if (valid && out.ftab) {
    const int32_t tq = out.ftab_id[p];
    if (tq >= 0) {
      const unsigned long long key = (unsigned long long)(yawind - out.f_minyaw) * out.f_xw * out.f_yw +
                                     (yind - prm.miny) * out.f_xw + (xind - prm.minx);
      const unsigned long long cb = (unsigned long long)__double_as_longlong(cost);
      const unsigned long long* tab = out.ftab + (uint64_t)tq * 2 * out.ftab_slots;
      uint64_t i = pp_ftab_hash(key) & (out.ftab_slots - 1);
      for (int probe = 0; probe < kFtabProbes; ++probe) {
        const ulonglong2 e = *(const ulonglong2*)(tab + 2 * i);
        if (e.x == key) {
          known = e.y <= cb;
          break;
        }
        if (e.x == kFtabEmpty) break;
        i = (i + 1) & (out.ftab_slots - 1);
      }
    }
  }
  if (known) valid = false;
  // bit3: the index test passed, i.e. the primitive's poses were handed to is_collision
  unsigned v = (valid ? 1u : 0u) | (amb ? 2u : 0u) | ((fl & PP_F_NONFINITE) ? 4u : 0u) |
               (in_bounds ? 8u : 0u) | (known ? 32u : 0u);
  double fc = 0.0;
  if (out.fcost || out.rec) {
    if (valid) {
      // is_index_ok guarantees minx < xind < maxx, so the lookup is inside the grid
      uint64_t hx = xind - prm.minx, hy = yind - prm.miny;
      if (hx < out.hxw && hy < out.hyw) {
        const double h = out.hblk ? pp_blocked_value(out.hblk, out.hblk_stamp, out.hblk_nty,
                                                     (uint64_t)out.hmap_id[p] * out.hblk_plane_cols + hx, hy)
                                  : out.hmaps[out.hmap_id[p]][hx * out.hyw + hy];
        fc = cost + out.h_cost * h;
        if (out.hobs && h > out.h_limit && !out.hobs[hx * out.hyw + hy]) v |= 16u;
      } else {
        fc = __longlong_as_double(0x7FF8000000000000ll);
      }
    }
    if (out.fcost && !(out.sparse && !valid)) out.fcost[t] = fc;
  }
  out.valid[t] = (uint8_t)v;
  if (out.rec) {  // one 64-byte line per valid primitive instead of eight scattered words
    if (valid) {
      ulonglong2* r = (ulonglong2*)(out.rec + 8 * t);
      r[0] = make_ulonglong2(xind, yind);
      r[1] = make_ulonglong2((unsigned long long)yawind, (unsigned long long)__double_as_longlong(cost));
      r[2] = make_ulonglong2((unsigned long long)__double_as_longlong(x), (unsigned long long)__double_as_longlong(y));
      r[3] = make_ulonglong2((unsigned long long)__double_as_longlong(yaw), (unsigned long long)__double_as_longlong(fc));
    }
    return;
  }
  if (out.sparse && !valid) return;  // the caller reads the other arrays only where bit 0 is set
  out.xind[t] = valid ? xind : 0;
  out.yind[t] = valid ? yind : 0;
  out.yawind[t] = valid ? yawind : 0;
  out.cost[t] = valid ? cost : 0.0;
  out.end_x[t] = valid ? x : 0.0;
  out.end_y[t] = valid ? y : 0.0;
  out.end_yaw[t] = valid ? yaw : 0.0;
}

#endif
