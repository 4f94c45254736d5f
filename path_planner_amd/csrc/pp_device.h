// pp_device.h -- device functions shared by the kernels: the footprint-vs-obstacles predicate.
#ifndef PP_DEVICE_H_
#define PP_DEVICE_H_

#include "pp_internal.h"

// Result bits of one pose test.
#define PP_F_HIT 1u        // is_collision would return true for this pose
#define PP_F_AMBIG 2u      // the outcome rests on a comparison inside the guard band
#define PP_F_NONFINITE 4u  // kdtree.within would Err (non-finite query point) and the reference panic

// One obstacle point against one pose, exactly the reference's arithmetic
// (hybrid_astar.rs:437-441 with util.rs:1-5 for th = -yaw, cos(-yaw) = c, sin(-yaw) = -s):
//   xo = ox - cx, yo = oy - cy
//   dx = xo*cos(th) - yo*sin(th) = xo*c + yo*s          (each product rounded, then one add)
//   dy = xo*sin(th) + yo*cos(th) = yo*c - xo*s
//   hit iff |dx| < bound_len && |dy| < bound_wid, among points with
//   sqrt(xo^2 + yo^2) <= bound_radius (the KD-tree radius query, :426-431).
// Returns 0 = miss, 1 = clear hit, 2 = hit inside the guard band, 3 = miss inside the guard band.
__device__ __forceinline__ int pp_point_vs_pose(double ox, double oy, double cx, double cy, double c,
                                                double s, const PpFootprint& f) {
  double xo = ox - cx;
  double yo = oy - cy;
  double p0 = xo * c, p1 = yo * s;
  double dx = p0 + p1;
  double q0 = yo * c, q1 = xo * s;
  double dy = q0 - q1;
  double ax = fabs(dx), ay = fabs(dy);
  bool inx = ax < f.bound_len, iny = ay < f.bound_wid;
  // cheap reject with no guard-band involvement
  if (!(ax < f.bound_len + f.guard) || !(ay < f.bound_wid + f.guard)) return 0;
  double s2 = xo * xo + yo * yo;
  bool inr = s2 <= f.r2_max;
  bool hit = inx && iny && inr;
  bool near_edge = (fabs(ax - f.bound_len) < f.guard) || (fabs(ay - f.bound_wid) < f.guard) ||
                   (fabs(s2 - f.r2_max) < f.guard * 8.0);
  if (!near_edge) return hit ? 1 : 0;
  return hit ? 2 : 3;
}

// Centre of the footprint rectangle (hybrid_astar.rs:423-424) from the rear-axle pose.
__device__ __forceinline__ void pp_footprint_centre(double x, double y, double c, double s,
                                                    const PpFootprint& f, double& cx, double& cy) {
  double a = f.cp2center * c;
  double b = f.cp2center * s;
  cx = x + a;
  cy = y + b;
}

// Bit-packed lattice: test every occupied cell inside the rotated rectangle's bounding box.
// Cells outside that box (enlarged by 1e-6) can never pass the rectangle test, so skipping
// them leaves the reference's any-hit result unchanged.
__device__ __forceinline__ unsigned pp_pose_vs_lattice(double x, double y, double c, double s,
                                                       const PpFootprint& f, const PpLattice& g) {
  double cx, cy;
  pp_footprint_centre(x, y, c, s, f, cx, cy);
  if (!(fabs(cx) < 1.0e300) || !(fabs(cy) < 1.0e300)) return PP_F_NONFINITE;
  double ac = fabs(c), as = fabs(s);
  double hx = ac * f.bound_len + as * f.bound_wid + 1.0e-6;
  double hy = as * f.bound_len + ac * f.bound_wid + 1.0e-6;
  // window in grid coordinates, clamped to the (padded) grid
  double gx = cx - (double)g.x0, gy = cy - (double)g.y0;
  double fx0 = ceil(gx - hx), fx1 = floor(gx + hx);
  double fy0 = ceil(gy - hy), fy1 = floor(gy + hy);
  fx0 = fmax(fx0, 0.0);
  fy0 = fmax(fy0, 0.0);
  fx1 = fmin(fx1, (double)(g.nx - 1));
  fy1 = fmin(fy1, (double)(g.ny - 1));
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return 0;
  int ix0 = (int)fx0, ix1 = (int)fx1, iy0 = (int)fy0, iy1 = (int)fy1;
  bool amb = false, ahit = false;
  for (int by = iy0 >> 3; by <= (iy1 >> 3); ++by) {
    int ybase = by << 3;
    for (int bx = ix0 >> 4; bx <= (ix1 >> 4); ++bx) {
      int xbase = bx << 4;
      size_t addr = ((size_t)(by >> 2) * (size_t)g.tiles_x + (size_t)(bx >> 1)) * 8 +
                    (size_t)((by & 3) * 2 + (bx & 1));
      uint4 blk = g.tiles[addr];
      // columns of this block inside the window
      int c0 = max(ix0 - xbase, 0), c1 = min(ix1 - xbase, 15);
      unsigned colmask = ((2u << c1) - 1u) & ~((1u << c0) - 1u);  // bits c0..c1 of 16
      colmask |= colmask << 16;
      unsigned w[4] = {blk.x, blk.y, blk.z, blk.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int r0 = ybase + 2 * k;
        unsigned rowsel = ((r0 >= iy0 && r0 <= iy1) ? 0x0000FFFFu : 0u) |
                          ((r0 + 1 >= iy0 && r0 + 1 <= iy1) ? 0xFFFF0000u : 0u);
        unsigned m = w[k] & colmask & rowsel;
        while (m) {
          int b = __builtin_ctz(m);
          m &= m - 1;
          double ox = (double)((int64_t)(xbase + (b & 15)) + g.x0);
          double oy = (double)((int64_t)(r0 + (b >> 4)) + g.y0);
          int r = pp_point_vs_pose(ox, oy, cx, cy, c, s, f);
          if (r == 1) return PP_F_HIT;  // a clear hit settles the pose whatever else is near an edge
          ahit |= (r == 2);
          amb |= (r == 3);
        }
      }
    }
  }
  return ahit ? (PP_F_HIT | PP_F_AMBIG) : (amb ? PP_F_AMBIG : 0u);
}

// General obstacle points in unit buckets: every point of the buckets the bounding box touches.
__device__ __forceinline__ unsigned pp_pose_vs_buckets(double x, double y, double c, double s,
                                                       const PpFootprint& f, const PpBuckets& g) {
  double cx, cy;
  pp_footprint_centre(x, y, c, s, f, cx, cy);
  if (!(fabs(cx) < 1.0e300) || !(fabs(cy) < 1.0e300)) return PP_F_NONFINITE;
  double ac = fabs(c), as = fabs(s);
  double hx = ac * f.bound_len + as * f.bound_wid + 1.0e-6;
  double hy = as * f.bound_len + ac * f.bound_wid + 1.0e-6;
  double fx0 = fmax(floor(cx - hx) - (double)g.bx0, 0.0);
  double fy0 = fmax(floor(cy - hy) - (double)g.by0, 0.0);
  double fx1 = fmin(floor(cx + hx) - (double)g.bx0, (double)(g.nbx - 1));
  double fy1 = fmin(floor(cy + hy) - (double)g.by0, (double)(g.nby - 1));
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return 0;
  int ix0 = (int)fx0, ix1 = (int)fx1, iy0 = (int)fy0, iy1 = (int)fy1;
  bool amb = false, ahit = false;
  for (int by = iy0; by <= iy1; ++by) {
    size_t row = (size_t)by * (size_t)g.nbx;
    uint32_t a = g.start[row + ix0], b = g.start[row + ix1 + 1];
    for (uint32_t k = a; k < b; ++k) {
      int r = pp_point_vs_pose(g.px[k], g.py[k], cx, cy, c, s, f);
      if (r == 1) return PP_F_HIT;
      ahit |= (r == 2);
      amb |= (r == 3);
    }
  }
  return ahit ? (PP_F_HIT | PP_F_AMBIG) : (amb ? PP_F_AMBIG : 0u);
}

#endif
