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
// Bit-packed lattice (layout: PpLattice in pp_internal.h).

// Columns xb .. xb + 15 (xb a multiple of 16) of grid row y, in the low 16 bits.
__device__ __forceinline__ unsigned pp_lattice_row16(const PpLattice& g, int xb, int y) {
  const uint32_t* w = (const uint32_t*)(g.tiles + ((size_t)(xb >> 4) * (size_t)g.nty + (size_t)(y >> 4)) * 8);
  return (w[y & 15] >> kLatticeShift) & 0xFFFFu;
}

// Exhaustive form: the exact predicate on every occupied cell of the rotated rectangle's
// bounding box (enlarged by 1e-6, clamped to the grid).  Cells outside that box can never pass
// the rectangle test, so skipping them leaves the reference's any-hit result unchanged.
__device__ __forceinline__ unsigned pp_lattice_scan_exact(double cx, double cy, double gx, double gy,
                                                          double c, double s, const PpFootprint& f,
                                                          const PpLattice& g) {
  double ac = fabs(c), as = fabs(s);
  double hx = ac * f.bound_len + as * f.bound_wid + 1.0e-6;
  double hy = as * f.bound_len + ac * f.bound_wid + 1.0e-6;
  double fx0 = fmax(ceil(gx - hx), 0.0), fx1 = fmin(floor(gx + hx), (double)(g.nx - 1));
  double fy0 = fmax(ceil(gy - hy), 0.0), fy1 = fmin(floor(gy + hy), (double)(g.ny - 1));
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return 0;
  int ix0 = (int)fx0, ix1 = (int)fx1, iy0 = (int)fy0, iy1 = (int)fy1;
  bool amb = false, ahit = false;
  for (int y = iy0; y <= iy1; ++y) {
    for (int xb = ix0 & ~15; xb <= ix1; xb += 16) {
      unsigned w = pp_lattice_row16(g, xb, y);
      int c0 = max(ix0 - xb, 0), c1 = min(ix1 - xb, 15);
      unsigned m = w & (0xFFFFu >> (15 - c1)) & (0xFFFFu << c0);
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

// ---------------------------------------------------------------------------------------------
// Scanline form of the lattice test, for footprints whose bounding radius is below 6 cells
// (f.scan_ok).  Everything here is a FILTER around the exact predicate: it may only answer when
// the exact fp64 predicate (under any libm accurate to a few ulp) provably agrees.
//
// The window is the 13 columns floor(gx)-6 .. +6 by the 16 rows of the four block rows (uint4 of
// the tile) covering floor(gy)-6 .. +6, in window coordinates (origin ix0, yb).  Two rectangles around the true
// footprint rectangle are used, both from fp32 sin/cos (error < 2e-7) and fp32 arithmetic
// (error < 1e-4 cells at window coordinates below 64):
//   * inflated by 2.5e-4 on every side (f.scan_lo, f.scan_wo): rasterised row by row, window
//     row y spanning [lo(y), hi(y)],
//       lo(y) = xL + max((y - yL) m1, (y - yL) m2),  hi(y) = xR + min((y - yR) m1, (y - yR) m2),
//     (xL, yL) / (xR, yR) the leftmost / rightmost vertices, m1 = c/s and m2 = -s/c the dx/dy
//     slopes of the edges.  The row masks cover every cell the exact predicate could accept:
//     no occupied cell under them => miss.
//   * deflated by 2.5e-4 (f.scan_li, f.scan_wi): one occupied cell found under the masks is
//     tested against it; inside => the exact predicate accepts that cell => hit.
// Anything else (the cell lies in the 5e-4 band around the true edge) is undecided and the
// caller runs the exact form.  The two central block rows are fetched first; the outer two only
// by lanes still without a candidate.

struct PpScan {
  float xL, yL, xR1, yR, m1, m2;
  float c0, c1;  // column range of the window inside the 32-bit tile rows: [c0, c1]
};

// v_cvt_u32_f32 saturates (negative -> 0, huge -> 0xFFFFFFFF); the C++ cast is undefined there.
__device__ __forceinline__ unsigned pp_cvt_u32_sat(float v) {
  unsigned r;
  asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
// a - b, 0 when b > a
__device__ __forceinline__ unsigned pp_sub_sat(unsigned a, unsigned b) {
  return __builtin_elementwise_sub_sat(a, b);  // v_sub_u32 ... clamp
}
// ((1 << width) - 1) << offset on the low 5 bits of each operand
__device__ __forceinline__ unsigned pp_bfm(unsigned width, unsigned offset) {
  unsigned r;
  asm("v_bfm_b32 %0, %1, %2" : "=v"(r) : "v"(width), "v"(offset));
  return r;
}

// Four window rows (one block row).  `key` keeps one candidate: (occupied bits under the row
// mask) << 4 | window row, maximised over rows, so key >= 16 iff some row had a candidate.
template <int J0>
__device__ __forceinline__ unsigned pp_scan_rows4(const PpScan& q, uint4 lb, unsigned key) {
  const unsigned lw[4] = {lb.x, lb.y, lb.z, lb.w};
  unsigned k[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float yr = (float)(J0 + r);
    const float t = yr - q.yL, t2 = yr - q.yR;
    // lo clamped below and hi above to the window's columns by the three-operand max / min
    const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(t, q.m1, q.xL), __builtin_fmaf(t, q.m2, q.xL)), q.c0);
    const float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaf(t2, q.m1, q.xR1), __builtin_fmaf(t2, q.m2, q.xR1)), q.c1);
    // cells ceil(lo) .. floor(hi) - 1 (xR1 carries the +1); the conversion truncates, which is
    // floor for hi >= 0, and a negative hi becomes 0.  Empty row: width 0 -> mask 0 whatever the
    // offset's low bits are; width <= 24
    const unsigned cl = pp_cvt_u32_sat(__builtin_ceilf(lo));
    const unsigned mask = pp_bfm(pp_sub_sat(pp_cvt_u32_sat(hi), cl), cl);
    // x coordinates carry the window's bit position inside the tile row (4 + at most 15, so the 13
    // window columns and the masks stay below bit 32): the row word is used as it is, and with
    // its low four bits free the row number goes in with the same instruction (v_and_or_b32)
    k[r] = (lw[r] & mask) | (unsigned)(J0 + r);
  }
  return max(max(key, k[0]), max(max(k[1], k[2]), k[3]));
}

// sin and cos of yaw to fp32 accuracy (absolute error < 2e-7) for |yaw| < 1e8: one fp64
// reduction step to [-pi/4, pi/4], then fp32 polynomials.  Returns false outside that range.
__device__ __forceinline__ bool pp_sincos_f32(double yaw, float& sf, float& cf) {
  if (!(fabs(yaw) < 1.0e8)) return false;
  const double kd = rint(yaw * 0.63661977236758134308);       // 2/pi
  const double r = __builtin_fma(-kd, 1.57079632679489661923, yaw);
  const int q = (int)kd;
  const float x = (float)r, z = x * x;
  float ps = __builtin_fmaf(z, 2.7557319e-6f, -1.9841270e-4f);
  ps = __builtin_fmaf(z, ps, 8.3333333e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666667e-1f);
  const float sn = __builtin_fmaf(x * z, ps, x);
  float pc = __builtin_fmaf(z, 2.4801587e-5f, -1.3888889e-3f);
  pc = __builtin_fmaf(z, pc, 4.1666668e-2f);
  pc = __builtin_fmaf(z, pc, -0.5f);
  const float cs = __builtin_fmaf(z, pc, 1.0f);
  const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
  sf = __uint_as_float(__float_as_uint(a) ^ (((unsigned)q & 2u) << 30));
  cf = __uint_as_float(__float_as_uint(b) ^ (((unsigned)(q + 1) & 2u) << 30));
  return true;
}

// Result of the filter: 0 = miss, 1 = hit, 2 = undecided.
// ROWS = 16: the window described above.  ROWS = 8 (f.scan_small: bounding radius below 3 cells):
// the footprint reaches rows floor(gy)-3 .. +3 only; the two block rows from the one that holds
// floor(gy)-3 cover them unless that row is the third or fourth of its block row, and then a third
// block row is fetched by the lanes still without a candidate.  (Two 16-byte loads that start at
// the row itself would always do, but a 16-byte load that is only 4-byte aligned is handled one
// dword at a time: measured slower than three aligned ones.)
template <int ROWS>
__device__ __forceinline__ int pp_lattice_filter(double x, double y, float cf, float sf,
                                                 const PpFootprint& f, const PpLattice& g) {
  // footprint centre from the fp32 direction: off by < 1e-6, inside the filter's margins
  const double gx = (x + f.cp2center * (double)cf) - (double)g.x0;
  const double gy = (y + f.cp2center * (double)sf) - (double)g.y0;
  // the padding ring is 32 cells wide: from farther out than 26 the footprint reaches no cell
  // (NaN and infinities fail the comparison and are left to the exact form)
  const bool inside = gx >= 26.0 && gx <= (double)(g.nx - 26) && gy >= 26.0 && gy <= (double)(g.ny - 26);
  if (!inside) return (fabs(gx) < 1.0e300 && fabs(gy) < 1.0e300) ? 0 : 2;
  const double bxf = floor(gx), byf = floor(gy);
  const int ibx = (int)bxf, iby = (int)byf;
  const float fx = (float)(gx - bxf), fy = (float)(gy - byf);
  constexpr int kUp = ROWS == 8 ? 3 : 6;               // rows below the centre's row in the window
  const int ix0 = ibx - 6, ytop = iby - kUp;
  const int o = ytop & 3;                              // window rows start at yb = ytop - o
  const float wx = 6.0f + fx, wy = (float)(kUp + o) + fy;  // centre in window coordinates

  const float acf = __builtin_fabsf(cf), asf = __builtin_fabsf(sf);
  const float exo = __builtin_fmaf(acf, f.scan_lo, asf * f.scan_wo);  // half-extent in x
  // leftmost vertex relative to the centre: (-exo, vyo), vyo = -sgn(c) Lo s + sgn(s) Wo c, the
  // signs taken from the sign BITS so that they agree with the slope denominators below
  const float s_x = __uint_as_float(__float_as_uint(sf) ^ (__float_as_uint(cf) & 0x80000000u));
  const float c_x = __uint_as_float(__float_as_uint(cf) ^ (__float_as_uint(sf) & 0x80000000u));
  const float vyo = __builtin_fmaf(f.scan_wo, c_x, -(f.scan_lo * s_x));
  // from here on x is measured in bit positions of the tile row: window column + (ix0 & 15) +
  // kLatticeShift
  const float shf = (float)((ix0 & 15) + kLatticeShift);
  const float wxs = wx + shf;
  PpScan q;
  q.xL = wxs - exo;
  q.yL = wy + vyo;
  q.xR1 = wxs + exo + 1.0f;
  q.yR = wy - vyo;
  q.c0 = shf;
  q.c1 = shf + 24.0f;
  // edge slopes dx/dy; a vanishing denominator becomes +-1e-30 with its own sign bit
  const float sden = (asf < 1e-30f) ? __builtin_copysignf(1e-30f, sf) : sf;
  const float cden = (acf < 1e-30f) ? __builtin_copysignf(1e-30f, cf) : cf;
  q.m1 = cf * __builtin_amdgcn_rcpf(sden);
  q.m2 = -sf * __builtin_amdgcn_rcpf(cden);

  // the tile that starts at or before the window's low corner holds all of it: rows ytop & 15 ..
  // + 12 < 32, columns (ix0 & 15) .. + 12 < 32; the four block rows from yb are uint4 (yb & 15) >> 2 .. + 3
  unsigned key = 0;
#ifdef PP_LAB_ONE_LINE  // timing ablation: every lane reads the same line (wrong flags, no gather cost)
  const uint4* base = g.tiles + (size_t)((ytop & 15) >> 2);
#else
  const uint4* base = g.tiles + (((size_t)(ix0 >> 4) * (size_t)g.nty + (size_t)(ytop >> 4)) * 8 +
                                 (size_t)((ytop & 15) >> 2));
#endif
  if (ROWS == 8) {
    const uint4 l0 = base[0], l1 = base[1];
    key = pp_scan_rows4<0>(q, l0, key);
    key = pp_scan_rows4<4>(q, l1, key);
    if (key < 16) {
      if (o < 2) return 0;  // rows o .. o + 6 end inside the second block row
      const uint4 l2 = base[2];
      key = pp_scan_rows4<8>(q, l2, key);
      if (key < 16) return 0;
    }
  } else {
    const uint4 l1 = base[1], l2 = base[2];
    key = pp_scan_rows4<4>(q, l1, key);
    key = pp_scan_rows4<8>(q, l2, key);
    if (key < 16) {
      const uint4 l0 = base[0], l3 = base[3];
      key = pp_scan_rows4<0>(q, l0, key);
      key = pp_scan_rows4<12>(q, l3, key);
      if (key < 16) return 0;
    }
  }
  // one candidate cell against the deflated rectangle
  const float dx = (float)__builtin_ctz(key >> 4) + (float)kLatticeShift - wxs, dy = (float)(key & 15u) - wy;
  const float u = __builtin_fmaf(dx, cf, dy * sf), v = __builtin_fmaf(dy, cf, -(dx * sf));
  return (__builtin_fabsf(u) < f.scan_li && __builtin_fabsf(v) < f.scan_wi) ? 1 : 2;
}

// Exact form from the pose and its fp64 direction.
__device__ __forceinline__ unsigned pp_lattice_exact(double x, double y, double c, double s,
                                                     const PpFootprint& f, const PpLattice& g) {
  double cx, cy;
  pp_footprint_centre(x, y, c, s, f, cx, cy);
  if (!(fabs(cx) < 1.0e300) || !(fabs(cy) < 1.0e300)) return PP_F_NONFINITE;
  return pp_lattice_scan_exact(cx, cy, cx - (double)g.x0, cy - (double)g.y0, c, s, f, g);
}

// Pose with a known fp64 direction (the rollout and trajectory kernels have it already).
__device__ __forceinline__ unsigned pp_pose_vs_lattice(double x, double y, double c, double s,
                                                       const PpFootprint& f, const PpLattice& g) {
  if (f.scan_ok) {
    const int r = f.scan_small ? pp_lattice_filter<8>(x, y, (float)c, (float)s, f, g)
                               : pp_lattice_filter<16>(x, y, (float)c, (float)s, f, g);
    if (r < 2) return r ? PP_F_HIT : 0u;
  }
  return pp_lattice_exact(x, y, c, s, f, g);
}

// Pose given by its yaw: fp32 direction for the filter, fp64 sin/cos only when undecided.  ROWS = 0
// picks the window by the footprint at run time; 8 / 16 are for kernels compiled for one of them.
template <int ROWS = 0>
__device__ __forceinline__ unsigned pp_pose_yaw_vs_lattice(double x, double y, double yaw,
                                                           const PpFootprint& f, const PpLattice& g) {
  float sf, cf;
  if (f.scan_ok && pp_sincos_f32(yaw, sf, cf)) {
    const int r = (ROWS == 8 || (ROWS == 0 && f.scan_small)) ? pp_lattice_filter<8>(x, y, cf, sf, f, g)
                                                             : pp_lattice_filter<16>(x, y, cf, sf, f, g);
    if (r < 2) return r ? PP_F_HIT : 0u;
  }
  double s, c;
  sincos(yaw, &s, &c);
  return pp_lattice_exact(x, y, c, s, f, g);
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
