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

// ---------------------------------------------------------------------------------------------
// Bit-packed lattice.  A block is 32 columns x 4 rows (one uint4, word r = row r, bit = column);
// a 128-B tile stacks 8 blocks (32 x 32 cells); see PpLattice.

__device__ __forceinline__ uint4 pp_lattice_block(const PpLattice& g, int col, int brow) {
  // col: any column of the block; brow: row >> 2
  size_t addr = ((size_t)(brow >> 3) * (size_t)g.tiles_x + (size_t)(col >> 5)) * 8 + (size_t)(brow & 7);
  return g.tiles[addr];
}

// Exhaustive form: the exact predicate on every occupied cell of the rotated rectangle's
// bounding box [ix0..ix1] x [iy0..iy1] (grid coordinates).  Cells outside that box (it is
// enlarged by 1e-6) can never pass the rectangle test, so skipping them leaves the reference's
// any-hit result unchanged.
__device__ __noinline__ unsigned pp_lattice_scan_exact(double cx, double cy, double c, double s,
                                                       int ix0, int ix1, int iy0, int iy1,
                                                       const PpFootprint& f, const PpLattice& g) {
  bool amb = false, ahit = false;
  for (int y = iy0; y <= iy1; ++y) {
    for (int xb = ix0 & ~31; xb <= ix1; xb += 32) {
      uint4 blk = pp_lattice_block(g, xb, y >> 2);
      int r = y & 3;
      unsigned w = r == 0 ? blk.x : r == 1 ? blk.y : r == 2 ? blk.z : blk.w;
      int c0 = max(ix0 - xb, 0), c1 = min(ix1 - xb, 31);
      unsigned m = w & (0xFFFFFFFFu >> (31 - c1)) & (0xFFFFFFFFu << c0);
      while (m) {
        int b = __builtin_ctz(m);
        m &= m - 1;
        double ox = (double)((int64_t)(xb + b) + g.x0);
        double oy = (double)((int64_t)y + g.y0);
        int t = pp_point_vs_pose(ox, oy, cx, cy, c, s, f);
        if (t == 1) return PP_F_HIT;  // a clear hit settles the pose whatever else is near an edge
        ahit |= (t == 2);
        amb |= (t == 3);
      }
    }
  }
  return ahit ? (PP_F_HIT | PP_F_AMBIG) : (amb ? PP_F_AMBIG : 0u);
}

// Scanline form.  The rectangle, inflated by PP_SCAN_MARGIN on every side, is rasterised row by
// row in fp32: for window row y the inflated rectangle spans [lo(y), hi(y)], with
//   lo(y) = xL + max((y - yL) * m1, (y - yL) * m2),  hi(y) = xR + min((y - yR) * m1, (y - yR) * m2)
// where (xL, yL) / (xR, yR) are its leftmost / rightmost vertices and m1 = c/s, m2 = -s/c are
// the dx/dy slopes of its edges.  fp32 evaluation errors (< 5e-5 cells for window coordinates
// below 64) are far inside the inflation, so the row masks cover every cell the exact predicate
// could accept: no occupied cell under the masks => miss.  An occupied cell under a mask is then
// put to the exact fp64 predicate; a clear hit settles the pose, anything else (the cell sits in
// the thin band between the true and the inflated rectangle) falls back to the exhaustive form.
#define PP_SCAN_MARGIN 2.5e-4

__device__ __forceinline__ unsigned pp_pose_vs_lattice(double x, double y, double c, double s,
                                                       const PpFootprint& f, const PpLattice& g) {
  double cx, cy;
  pp_footprint_centre(x, y, c, s, f, cx, cy);
  if (!(fabs(cx) < 1.0e300) || !(fabs(cy) < 1.0e300)) return PP_F_NONFINITE;
  double ac = fabs(c), as = fabs(s);
  double ex = ac * f.bound_len + as * f.bound_wid;
  double ey = as * f.bound_len + ac * f.bound_wid;
  double hx = ex + 1.0e-6, hy = ey + 1.0e-6;
  // window in grid coordinates, clamped to the (padded) grid
  double gx = cx - (double)g.x0, gy = cy - (double)g.y0;
  double fx0 = fmax(ceil(gx - hx), 0.0), fx1 = fmin(floor(gx + hx), (double)(g.nx - 1));
  double fy0 = fmax(ceil(gy - hy), 0.0), fy1 = fmin(floor(gy + hy), (double)(g.ny - 1));
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return 0;
  int ix0 = (int)fx0, ix1 = (int)fx1, iy0 = (int)fy0, iy1 = (int)fy1;
  if (ix1 - ix0 > 20 || iy1 - iy0 > 12)  // footprint larger than the scanline window handles
    return pp_lattice_scan_exact(cx, cy, c, s, ix0, ix1, iy0, iy1, f, g);

  // inflated rectangle in window coordinates: origin (ix0, yb)
  const int yb = iy0 & ~3;
  const double Lo = f.bound_len + PP_SCAN_MARGIN, Wo = f.bound_wid + PP_SCAN_MARGIN;
  const double sc = (c < 0.0) ? -1.0 : 1.0, ss = (s < 0.0) ? -1.0 : 1.0;
  const double exo = ac * Lo + as * Wo;                // half-extent in x of the inflated rectangle
  const double vyo = -sc * Lo * s + ss * Wo * c;       // y of its leftmost vertex, relative to centre
  const double wx = gx - (double)ix0, wy = gy - (double)yb;
  const float xL = (float)(wx - exo), yL = (float)(wy + vyo);
  const float xR1 = (float)(wx + exo + 1.0), yR = (float)(wy - vyo);
  // edge slopes dx/dy; a zero denominator becomes +-1e-30 (same sign convention as sc/ss)
  const float cf = (float)c, sf = (float)s;
  const float cden = (fabsf(cf) < 1e-30f) ? (float)(sc * 1e-30) : cf;
  const float sden = (fabsf(sf) < 1e-30f) ? (float)(ss * 1e-30) : sf;
  const float m1 = cf / sden, m2 = -sf / cden;

  const int sh = ix0 & 31;
  const bool two = (ix1 >> 5) != (ix0 >> 5);
  const int last_brow = (g.ny - 1) >> 2;
  unsigned cand = 0;
  int cand_row = 0;
#pragma unroll 1
  for (int br = 0; br < 4; ++br) {
    const int brow = (yb >> 2) + br;
    uint4 lb = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    if (cand == 0 && brow <= last_brow && (brow << 2) <= iy1) {
      lb = pp_lattice_block(g, ix0, brow);
      if (two) rb = pp_lattice_block(g, ix0 + 32, brow);
    }
    const unsigned lw[4] = {lb.x, lb.y, lb.z, lb.w};
    const unsigned rw[4] = {rb.x, rb.y, rb.z, rb.w};
    unsigned found[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float yr = (float)(br * 4 + r);
      const float t = yr - yL, t2 = yr - yR;
      float lo = fmaxf(__builtin_fmaf(t, m1, xL), __builtin_fmaf(t, m2, xL));
      float hi = fminf(__builtin_fmaf(t2, m1, xR1), __builtin_fmaf(t2, m2, xR1));
      lo = __builtin_amdgcn_fmed3f(lo, 0.0f, 24.0f);
      hi = __builtin_amdgcn_fmed3f(hi, 0.0f, 24.0f);
      const int ilo = (int)ceilf(lo), ihi1 = (int)floorf(hi);  // cells ilo .. ihi1 - 1
      const int w = max(ihi1 - ilo, 0);
      const unsigned mask = ((1u << w) - 1u) << ilo;
      const unsigned bits = __builtin_amdgcn_alignbit(rw[r], lw[r], sh);
      found[r] = bits & mask;
    }
    const unsigned any = found[0] | found[1] | found[2] | found[3];
    if (any != 0 && cand == 0) {
      const int r = found[0] ? 0 : found[1] ? 1 : found[2] ? 2 : 3;
      cand = found[0] ? found[0] : found[1] ? found[1] : found[2] ? found[2] : found[3];
      cand_row = yb + br * 4 + r;
    }
  }
  if (cand == 0) return 0;
  {
    const int b = __builtin_ctz(cand);
    const double ox = (double)((int64_t)(ix0 + b) + g.x0);
    const double oy = (double)((int64_t)cand_row + g.y0);
    if (pp_point_vs_pose(ox, oy, cx, cy, c, s, f) == 1) return PP_F_HIT;
  }
  return pp_lattice_scan_exact(cx, cy, c, s, ix0, ix1, iy0, iy1, f, g);
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
