// pp_rs_device.h -- Reeds-Shepp word generation on the device (reeds_shepp.rs:130-580, :707-757).
//
// One lane evaluates every candidate word of one (start, goal) pair.  The reference's 46 call
// sites are 12 groups (base formula, optional "backwards" point, length layout) times up to four
// mirror images of (x, y, phi):
//   k=0 ( x,  y,  phi)  lengths +, word as written
//   k=1 (-x,  y, -phi)  lengths -, word as written          (time flip)
//   k=2 ( x, -y, -phi)  lengths +, L<->R swapped            (reflection)
//   k=3 (-x, -y,  phi)  lengths -, L<->R swapped
// sin(-phi) = -sin(phi) and cos(-phi) = cos(phi) hold exactly, so sin/cos of phi are computed
// once per pair and handed to the formulas with the sign applied.
//
// set_path's duplicate filter (:136-149) only compares words with identical letters, and equal
// letters occur in at most four candidates, always generated in a fixed relative order; the
// candidates are therefore evaluated word by word with a four-entry register list per word, and
// the reference's generation index (needed for tie-breaks and path numbering) is carried along.
#ifndef PP_RS_DEVICE_H_
#define PP_RS_DEVICE_H_

#include "pp_internal.h"

#define PP_RS_GUARD 1.0e-9
#define PP_RS_PI 3.14159265358979323846

struct PpTuv {
  bool ok;
  double t, u, v;
};

// The inverse trigonometric functions as real calls: each inlined copy brings two dozen 64-bit
// polynomial coefficients with it, which the compiler hoists out of every enclosing loop and then
// has to keep in registers across the loop's whole body.
#ifndef PP_RS_MATH_INLINE
#define PP_RS_MATH_ATTR __noinline__
#else
#define PP_RS_MATH_ATTR __forceinline__
#endif
__device__ PP_RS_MATH_ATTR double pp_atan2(double y, double x) { return atan2(y, x); }
__device__ PP_RS_MATH_ATTR double pp_asin(double x) { return asin(x); }
__device__ PP_RS_MATH_ATTR double pp_acos(double x) { return acos(x); }
__device__ PP_RS_MATH_ATTR double pp_tan(double x) { return tan(x); }
__device__ PP_RS_MATH_ATTR double pp_cos(double x) { return cos(x); }
__device__ PP_RS_MATH_ATTR double2 pp_sincos(double x) {
  double s, c;
  sincos(x, &s, &c);
  return make_double2(s, c);
}

// Accumulates "a comparison was decided inside the guard band".
struct PpAmb {
  bool amb;
  __device__ __forceinline__ bool ge(double a, double b) { amb |= fabs(a - b) < PP_RS_GUARD; return a >= b; }
  __device__ __forceinline__ bool le(double a, double b) { amb |= fabs(a - b) < PP_RS_GUARD; return a <= b; }
  __device__ __forceinline__ bool gt(double a, double b) { amb |= fabs(a - b) < PP_RS_GUARD; return a > b; }
  __device__ __forceinline__ bool lt(double a, double b) { amb |= fabs(a - b) < PP_RS_GUARD; return a < b; }
};

// reeds_shepp.rs:747-757.  `theta % (2 pi)` is exact (the remainder of a division by the double
// 2 pi is always representable), and so is one fused multiply-add with the truncated quotient; a
// quotient that comes out one off near a multiple of 2 pi leaves a remainder just below zero
// instead of just below 2 pi, which the wrap below maps to the same value (x - 2 pi is exact there).
__device__ __forceinline__ double pp_mod2pi(double theta, PpAmb& A) {
  const double c = 2.0 * PP_RS_PI;
  double phi;
  if (fabs(theta) < 1.0e6) {
    const double q = trunc(theta * (1.0 / (2.0 * PP_RS_PI)));
    phi = __builtin_fma(-q, c, theta);
  } else {
    phi = fmod(theta, c);
  }
  if (A.lt(phi, -PP_RS_PI)) phi += c;
  if (A.gt(phi, PP_RS_PI)) phi -= c;
  return phi;
}

// Polar form (reeds_shepp.rs:740-745) of one of the two points the formulas measure from:
//   set 0: (x - sin phi, y - 1 + cos phi)   lsl, lrl, lrsl
//   set 1: (x + sin phi, y - 1 - cos phi)   lsr, lrlrn, lrlrp, lrsr, lrslr
// Eight formulas share two points per mirror image, so a caller that evaluates all of them can
// compute the sixteen polar forms of a pair (four images, forward and backwards point) once.
struct PpPolar {
  double rho, th;
};
__device__ __forceinline__ PpPolar pp_polar(double px, double py) {
  return PpPolar{sqrt(px * px + py * py), pp_atan2(py, px)};
}
__device__ __forceinline__ PpPolar pp_polar_set(double x, double y, double sp, double cp, int set) {
  return set ? pp_polar(x + sp, y - 1.0 - cp) : pp_polar(x - sp, y - 1.0 + cp);
}
// reeds_shepp.rs:727-738
__device__ __forceinline__ double pp_pi_2_pi(double theta) {
  if (!(fabs(theta) < 1.0e9)) return theta;  // the reference would loop for ages; leave as is
  while (theta > PP_RS_PI) theta -= 2.0 * PP_RS_PI;
  while (theta < -PP_RS_PI) theta += 2.0 * PP_RS_PI;
  return theta;
}

// What two mirror images of a formula have in common once they share rho (PpPolarPair): filled in
// by the first image that gets that far, reused by the second (same lane, same rho, so the same
// range test let it in).
struct PpPairMemo {
  bool set = false;
  double a = 0.0, b = 0.0;
};

// Base formulas; sp / cp are sin(phi) / cos(phi) of the phi passed in.
// :213-229
__device__ __forceinline__ PpTuv pp_f_sls(double x, double y, double phi, PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double p = pp_mod2pi(phi, A);
  bool pos = A.gt(y, 0.), neg = A.lt(y, 0.);
  bool rng = A.lt(0., p) & A.lt(p, PP_RS_PI * 0.99);
  if ((pos | neg) & rng) {
    double tp = pp_tan(p), th = pp_tan(p / 2.);
    double xd = -y / tp + x;
    double t = xd - th;
    double rt = sqrt((x - xd) * (x - xd) + y * y);
    double v = pos ? (rt - th) : (-rt - th);
    r = PpTuv{true, t, p, v};
  }
  return r;
}
// :161-170
__device__ __forceinline__ PpTuv pp_f_lsl(double phi, const PpPolar& p0, PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double u = p0.rho, t = p0.th;
  if (A.ge(t, 0.0)) {
    double v = pp_mod2pi(phi - t, A);
    if (A.ge(v, 0.0)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :172-186
__device__ __forceinline__ PpTuv pp_f_lsr(double phi, const PpPolar& p1, PpAmb& A, PpPairMemo* M = nullptr) {
  PpTuv r{false, 0., 0., 0.};
  double u1 = p1.rho, t1 = p1.th;
  u1 = u1 * u1;
  if (A.ge(u1, 4.)) {
    double u, theta;
    if (M && M->set) {
      u = M->a;
      theta = M->b;
    } else {
      u = sqrt(u1 - 4.);
      theta = pp_atan2(2.0, u);
      if (M) *M = PpPairMemo{true, u, theta};
    }
    double t = pp_mod2pi(t1 + theta, A);
    double v = pp_mod2pi(t - phi, A);
    if (A.ge(t, 0.) & A.ge(v, 0.0)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :188-200
__device__ __forceinline__ PpTuv pp_f_lrl(double phi, const PpPolar& p0, PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double u1 = p0.rho, t1 = p0.th;
  if (A.le(u1, 4.)) {
    double u = -2. * pp_asin(0.25 * u1);
    double t = pp_mod2pi(t1 + 0.5 * u + PP_RS_PI, A);
    double v = pp_mod2pi(phi - t + u, A);
    if (A.ge(t, 0.) & A.le(u, 0.0)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :318-332
__device__ __forceinline__ void pp_tau_omega(double u, double v, double xi, double eta, double phi,
                                             double& tau, double& omega, PpAmb& A) {
  double delta = pp_mod2pi(u - v, A);
  const double2 scu = pp_sincos(u), scd = pp_sincos(delta);
  const double su = scu.x, cu = scu.y, sd = scd.x, cd = scd.y;
  double a = su - sd;
  double b = cu - cd - 1.0;
  double t1 = pp_atan2(eta * a - xi * b, xi * a + eta * b);
  double t2 = 2.0 * (cd - pp_cos(v) - cu) + 3.0;
  tau = A.lt(t2, 0.) ? pp_mod2pi(t1 + PP_RS_PI, A) : pp_mod2pi(t1, A);
  omega = pp_mod2pi(tau - u + v - phi, A);
}
// :334-347
__device__ __forceinline__ PpTuv pp_f_lrlrn(double x, double y, double phi, double sp, double cp, const PpPolar& p1,
                                            PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double xi = x + sp, eta = y - 1.0 - cp;
  double rho = 0.25 * (2.0 + p1.rho);
  if (A.le(rho, 1.0)) {
    double u = pp_acos(rho), t, v;
    pp_tau_omega(u, -u, xi, eta, phi, t, v, A);
    if (A.ge(t, 0.) & A.le(v, 0.)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :349-364
__device__ __forceinline__ PpTuv pp_f_lrlrp(double x, double y, double phi, double sp, double cp, PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double xi = x + sp, eta = y - 1.0 - cp;
  double rho = (20. - xi * xi - eta * eta) / 16.0;
  if (A.le(0.0, rho) & A.le(rho, 1.0)) {
    double u = -pp_acos(rho);
    if (A.ge(u, -0.5 * PP_RS_PI)) {
      double t, v;
      pp_tau_omega(u, u, xi, eta, phi, t, v, A);
      if (A.ge(t, 0.) & A.ge(v, 0.)) r = PpTuv{true, t, u, v};
    }
  }
  return r;
}
// :408-422
__device__ __forceinline__ PpTuv pp_f_lrsr(double phi, const PpPolar& p1, PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  // r(-eta, xi): the point (xi, eta) turned by a quarter, so its angle plus pi / 2, back in (-pi, pi]
  double rho = p1.rho, theta = p1.th + 0.5 * PP_RS_PI;
  if (theta > PP_RS_PI) theta -= 2.0 * PP_RS_PI;
  if (A.ge(rho, 2.0)) {
    double t = theta;
    double u = 2.0 - rho;
    double v = pp_mod2pi(t + 0.5 * PP_RS_PI - phi, A);
    if (A.ge(t, 0.0) & A.le(u, 0.0) & A.le(v, 0.0)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :423-438
__device__ __forceinline__ PpTuv pp_f_lrsl(double phi, const PpPolar& p0, PpAmb& A, PpPairMemo* M = nullptr) {
  PpTuv r{false, 0., 0., 0.};
  double rho = p0.rho, theta = p0.th;
  if (A.ge(rho, 2.0)) {
    double rr, alpha;
    if (M && M->set) {
      rr = M->a;
      alpha = M->b;
    } else {
      rr = sqrt(rho * rho - 4.0);
      alpha = pp_atan2(rr, -2.0);
      if (M) *M = PpPairMemo{true, rr, alpha};
    }
    double u = 2.0 - rr;
    double t = pp_mod2pi(theta + alpha, A);
    double v = pp_mod2pi(phi - 0.5 * PP_RS_PI - t, A);
    if (A.ge(t, 0.0) & A.le(u, 0.0) & A.le(v, 0.0)) r = PpTuv{true, t, u, v};
  }
  return r;
}
// :526-542
__device__ __forceinline__ PpTuv pp_f_lrslr(double x, double y, double phi, double sp, double cp, const PpPolar& p1,
                                            PpAmb& A) {
  PpTuv r{false, 0., 0., 0.};
  double xi = x + sp, eta = y - 1.0 - cp;
  double rho = p1.rho;
  if (A.ge(rho, 2.0)) {
    double u = 4.0 - sqrt(rho * rho - 4.0);
    if (A.le(u, 0.)) {
      double t = pp_mod2pi(pp_atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta), A);
      double v = pp_mod2pi(t - phi, A);
      if (A.ge(t, 0.0) & A.ge(v, 0.0)) r = PpTuv{true, t, u, v};
    }
  }
  return r;
}

enum { PP_F_SLS_, PP_F_LSL_, PP_F_LSR_, PP_F_LRL_, PP_F_LRLRN_, PP_F_LRLRP_, PP_F_LRSL_, PP_F_LRSR_, PP_F_LRSLR_ };
enum { PP_TUV, PP_VUT, PP_T_U_NU_V, PP_T_U_U_V, PP_T_H_U_V, PP_V_U_H_T, PP_T_H_U_H_V };

// Word letters packed 4 bits each (1 = 'L', 2 = 'S', 3 = 'R'), first letter in the low nibble.
__host__ __device__ constexpr uint32_t pp_word(const char* w) {
  uint32_t v = 0;
  for (int i = 0; w[i]; ++i) v |= (uint32_t)(w[i] == 'L' ? 1 : (w[i] == 'S' ? 2 : 3)) << (4 * i);
  return v;
}
__host__ __device__ constexpr uint32_t pp_word_reflect(uint32_t w) {
  uint32_t v = 0;
  for (int i = 0; i < 5; ++i) {
    uint32_t c = (w >> (4 * i)) & 15u;
    v |= (c == 1 ? 3u : (c == 3 ? 1u : c)) << (4 * i);
  }
  return v;
}
__host__ __device__ inline uint8_t pp_word_letter(uint32_t w, int i) {
  uint32_t c = (w >> (4 * i)) & 15u;
  return c == 1 ? 'L' : (c == 2 ? 'S' : (c == 3 ? 'R' : 0));
}

// Receiver of accepted words, in the order candidates are evaluated (NOT generation order):
//   void operator()(int gen_index, uint32_t word, int nseg, const double len[5], double total)
// `len` and `total` are in normalised (curvature) units.

// The accepted entries of one word (at most four), for set_path's duplicate test.
template <int N>
struct PpWordSet {
  double len[4][N];
  int cnt = 0;
};

struct PpRsState {
  PpAmb A{false};
  bool degenerate = false;  // assert!(total_length >= 0.01) would fire (:157)
  uint64_t accepted = 0;    // bit i: candidate with generation index i became a path
};

// set_path, :130-159, for one candidate of word set `ws`.
template <int N, class Sink>
__device__ __forceinline__ void pp_set_path(PpRsState& S, PpWordSet<N>& ws, const double (&len)[N],
                                            int gen, uint32_t word, Sink& sink) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < ws.cnt) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) s += ws.len[e][i] - len[i];
      if (S.A.le(s, 0.01)) return;  // an equal word exists
    }
  }
  double total = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) total += fabs(len[i]);
  if (S.A.ge(total, 1000.0)) return;  // MAX_LENGTH (:6)
  if (!S.A.ge(total, 0.01)) {
    S.degenerate = true;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e == ws.cnt) {
#pragma unroll
      for (int i = 0; i < N; ++i) ws.len[e][i] = len[i];
    }
  ws.cnt++;
  S.accepted |= 1ull << gen;
  double l5[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < N; ++i) l5[i] = len[i];
  sink(gen, word, N, l5, total);
}

struct PpRsInput {
  double x, y, phi, sp, cp, xb, yb;
};

// The point a formula measures from (pp_polar_set): -1 none, 0 or 1.
template <int F>
__host__ __device__ constexpr int pp_formula_set() {
  return F == PP_F_SLS_ || F == PP_F_LRLRP_ ? -1 : ((F == PP_F_LSL_ || F == PP_F_LRL_ || F == PP_F_LRSL_) ? 0 : 1);
}

template <int F>
__device__ __forceinline__ PpTuv pp_eval(double x, double y, double phi, double sp, double cp, const PpPolar& pol,
                                         PpAmb& A, PpPairMemo* M = nullptr) {
  if (F == PP_F_SLS_) return pp_f_sls(x, y, phi, A);
  if (F == PP_F_LSL_) return pp_f_lsl(phi, pol, A);
  if (F == PP_F_LSR_) return pp_f_lsr(phi, pol, A, M);
  if (F == PP_F_LRL_) return pp_f_lrl(phi, pol, A);
  if (F == PP_F_LRLRN_) return pp_f_lrlrn(x, y, phi, sp, cp, pol, A);
  if (F == PP_F_LRLRP_) return pp_f_lrlrp(x, y, phi, sp, cp, A);
  if (F == PP_F_LRSL_) return pp_f_lrsl(phi, pol, A, M);
  if (F == PP_F_LRSR_) return pp_f_lrsr(phi, pol, A);
  return pp_f_lrslr(x, y, phi, sp, cp, pol, A);
}

// One candidate: formula F on mirror image K of the (optionally backwards) point, laid out by
// LAY into N lengths, offered to word set `ws`.
template <int F, int K, bool BACK, int LAY, int N, class Sink>
__device__ __forceinline__ void pp_candidate(PpRsState& S, const PpRsInput& in, PpWordSet<N>& ws,
                                             int gen, uint32_t word_k0, Sink& sink) {
  const double bx = BACK ? in.xb : in.x, by = BACK ? in.yb : in.y;
  const double ax = (K & 1) ? -bx : bx;
  const double ay = (K & 2) ? -by : by;
  const bool negphi = (K == 1 || K == 2);
  const double ap = negphi ? -in.phi : in.phi;
  const double sp = negphi ? -in.sp : in.sp;
  PpPolar pol{0.0, 0.0};
  constexpr int SET = pp_formula_set<F>();
  if constexpr (SET >= 0) pol = pp_polar_set(ax, ay, sp, in.cp, SET);
  PpTuv r = pp_eval<F>(ax, ay, ap, sp, in.cp, pol, S.A);
  if (!r.ok) return;
  const double H = 0.5 * PP_RS_PI;
  double len[N];
  if constexpr (LAY == PP_TUV) { len[0] = r.t; len[1] = r.u; len[2] = r.v; }
  if constexpr (LAY == PP_VUT) { len[0] = r.v; len[1] = r.u; len[2] = r.t; }
  if constexpr (LAY == PP_T_U_NU_V) { len[0] = r.t; len[1] = r.u; len[2] = -r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_U_U_V) { len[0] = r.t; len[1] = r.u; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_H_U_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_V_U_H_T) { len[0] = r.v; len[1] = r.u; len[2] = -H; len[3] = r.t; }
  if constexpr (LAY == PP_T_H_U_H_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = -H; len[4] = r.v; }
  if (K & 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) len[i] = -len[i];
  }
  const uint32_t word = (K & 2) ? pp_word_reflect(word_k0) : word_k0;
  pp_set_path<N>(S, ws, len, gen, word, sink);
}

// The candidates of one group that share a word: mirror images K0 and K0+1.
template <int F, int K0, bool BACK, int LAY, int N, class Sink>
__device__ __forceinline__ void pp_pair(PpRsState& S, const PpRsInput& in, PpWordSet<N>& ws, int gen0,
                                        uint32_t word_k0, Sink& sink) {
  pp_candidate<F, K0, BACK, LAY, N>(S, in, ws, gen0 + K0, word_k0, sink);
  pp_candidate<F, K0 + 1, BACK, LAY, N>(S, in, ws, gen0 + K0 + 1, word_k0, sink);
}

// The start-frame, curvature-normalised goal (generate_path, :707-720).
__device__ __forceinline__ PpRsInput pp_rs_input(double sx, double sy, double syaw, double gx, double gy,
                                                 double gyaw, double maxc) {
  const double dx = gx - sx, dy = gy - sy, dth = gyaw - syaw;
  double s0, c0;
  sincos(syaw, &s0, &c0);
  PpRsInput in;
  in.x = (c0 * dx + s0 * dy) * maxc;
  in.y = (-s0 * dx + c0 * dy) * maxc;
  in.phi = dth;
  sincos(dth, &in.sp, &in.cp);
  in.xb = in.x * in.cp + in.y * in.sp;  // :294-295, :482-483
  in.yb = in.x * in.sp - in.y * in.cp;
  return in;
}

// The 46 candidates fall into seven families whose duplicate tests (set_path compares only paths
// with identical letters) never look outside the family, so the families can be evaluated in any
// order, or by different waves, and still accept exactly the reference's paths.  Inside a family
// the same holds for the words: twelve groups of at most four candidates, each group a set of
// words no other group shares.
constexpr int kRsFamilies = 7;
constexpr int kRsGroups = 12;

template <int G, class Sink>
__device__ __forceinline__ void pp_rs_group(const PpRsInput& in, PpRsState& S, Sink& sink) {
  if constexpr (G == 0) {  // scs :202-211 -- generation indices 0, 1
    PpWordSet<3> a, b;
    pp_candidate<PP_F_SLS_, 0, false, PP_TUV, 3>(S, in, a, 0, pp_word("SLS"), sink);
    pp_candidate<PP_F_SLS_, 2, false, PP_TUV, 3>(S, in, b, 1, pp_word("SLS"), sink);
  }
  if constexpr (G == 1) {  // csc :231-270 -- 2..5
    PpWordSet<3> a, b;
    pp_pair<PP_F_LSL_, 0, false, PP_TUV, 3>(S, in, a, 2, pp_word("LSL"), sink);
    pp_pair<PP_F_LSL_, 2, false, PP_TUV, 3>(S, in, b, 2, pp_word("LSL"), sink);
  }
  if constexpr (G == 2) {  // csc -- 6..9
    PpWordSet<3> c, d;
    pp_pair<PP_F_LSR_, 0, false, PP_TUV, 3>(S, in, c, 6, pp_word("LSR"), sink);
    pp_pair<PP_F_LSR_, 2, false, PP_TUV, 3>(S, in, d, 6, pp_word("LSR"), sink);
  }
  if constexpr (G == 3) {  // ccc :272-316 -- LRL collects four candidates (10, 11, 14, 15)
    PpWordSet<3> a;
    pp_pair<PP_F_LRL_, 0, false, PP_TUV, 3>(S, in, a, 10, pp_word("LRL"), sink);
    pp_pair<PP_F_LRL_, 0, true, PP_VUT, 3>(S, in, a, 14, pp_word("LRL"), sink);
  }
  if constexpr (G == 4) {  // ccc -- RLR (12, 13, 16, 17)
    PpWordSet<3> b;
    pp_pair<PP_F_LRL_, 2, false, PP_TUV, 3>(S, in, b, 10, pp_word("LRL"), sink);
    pp_pair<PP_F_LRL_, 2, true, PP_VUT, 3>(S, in, b, 14, pp_word("LRL"), sink);
  }
  if constexpr (G == 5) {  // cccc :366-406 -- LRLR (18, 19, 22, 23)
    PpWordSet<4> a;
    pp_pair<PP_F_LRLRN_, 0, false, PP_T_U_NU_V, 4>(S, in, a, 18, pp_word("LRLR"), sink);
    pp_pair<PP_F_LRLRP_, 0, false, PP_T_U_U_V, 4>(S, in, a, 22, pp_word("LRLR"), sink);
  }
  if constexpr (G == 6) {  // cccc -- RLRL (20, 21, 24, 25)
    PpWordSet<4> b;
    pp_pair<PP_F_LRLRN_, 2, false, PP_T_U_NU_V, 4>(S, in, b, 18, pp_word("LRLR"), sink);
    pp_pair<PP_F_LRLRP_, 2, false, PP_T_U_U_V, 4>(S, in, b, 22, pp_word("LRLR"), sink);
  }
  if constexpr (G == 7) {  // ccsc :440-524 -- 26..29
    PpWordSet<4> a, b;
    pp_pair<PP_F_LRSL_, 0, false, PP_T_H_U_V, 4>(S, in, a, 26, pp_word("LRSL"), sink);
    pp_pair<PP_F_LRSL_, 2, false, PP_T_H_U_V, 4>(S, in, b, 26, pp_word("LRSL"), sink);
  }
  if constexpr (G == 8) {  // ccsc -- 30..33
    PpWordSet<4> c, d;
    pp_pair<PP_F_LRSR_, 0, false, PP_T_H_U_V, 4>(S, in, c, 30, pp_word("LRSR"), sink);
    pp_pair<PP_F_LRSR_, 2, false, PP_T_H_U_V, 4>(S, in, d, 30, pp_word("LRSR"), sink);
  }
  if constexpr (G == 9) {  // ccsc, backwards -- 34..37
    PpWordSet<4> a, b;
    pp_pair<PP_F_LRSL_, 0, true, PP_V_U_H_T, 4>(S, in, a, 34, pp_word("LSRL"), sink);
    pp_pair<PP_F_LRSL_, 2, true, PP_V_U_H_T, 4>(S, in, b, 34, pp_word("LSRL"), sink);
  }
  if constexpr (G == 10) {  // ccsc, backwards -- 38..41
    PpWordSet<4> c, d;
    pp_pair<PP_F_LRSR_, 0, true, PP_V_U_H_T, 4>(S, in, c, 38, pp_word("RSRL"), sink);
    pp_pair<PP_F_LRSR_, 2, true, PP_V_U_H_T, 4>(S, in, d, 38, pp_word("RSRL"), sink);
  }
  if constexpr (G == 11) {  // ccscc :544-580 -- 42..45
    PpWordSet<5> a, b;
    pp_pair<PP_F_LRSLR_, 0, false, PP_T_H_U_H_V, 5>(S, in, a, 42, pp_word("LRSLR"), sink);
    pp_pair<PP_F_LRSLR_, 2, false, PP_T_H_U_H_V, 5>(S, in, b, 42, pp_word("LRSLR"), sink);
  }
}

template <int FAM, class Sink>
__device__ __forceinline__ void pp_rs_family(const PpRsInput& in, PpRsState& S, Sink& sink) {
  if constexpr (FAM == 0) pp_rs_group<0>(in, S, sink);
  if constexpr (FAM == 1) {
    pp_rs_group<1>(in, S, sink);
    pp_rs_group<2>(in, S, sink);
  }
  if constexpr (FAM == 2) {
    pp_rs_group<3>(in, S, sink);
    pp_rs_group<4>(in, S, sink);
  }
  if constexpr (FAM == 3) {
    pp_rs_group<5>(in, S, sink);
    pp_rs_group<6>(in, S, sink);
  }
  if constexpr (FAM == 4) {
    pp_rs_group<7>(in, S, sink);
    pp_rs_group<8>(in, S, sink);
  }
  if constexpr (FAM == 5) {
    pp_rs_group<9>(in, S, sink);
    pp_rs_group<10>(in, S, sink);
  }
  if constexpr (FAM == 6) pp_rs_group<11>(in, S, sink);
}

// ---------------------------------------------------------------------------------------------
// The same 46 candidates for ONE pair spread over the lanes of a workgroup: the four mirror images
// of a formula differ only in the signs of its inputs and outputs, so lanes 0..3 of a wave evaluate
// them in one pass over the formula's code (twelve waves, one per formula and direction), and
// set_path's order-dependent duplicate test then runs per word over the stored results.

struct PpRsRaw {       // one evaluated candidate, indexed by generation index
  double len[5];
  uint32_t word;
  int32_t nseg;        // 0: the formula rejected the candidate
};

// Formula F on mirror image K (a run-time value here) of the forward or backward point.
template <int F, bool BACK, int LAY, int N>
__device__ __forceinline__ void pp_eval_image(const PpRsInput& in, int K, uint32_t word_k0, PpAmb& A,
                                              PpRsRaw& out) {
  const double bx = BACK ? in.xb : in.x, by = BACK ? in.yb : in.y;
  const double ax = (K & 1) ? -bx : bx;
  const double ay = (K & 2) ? -by : by;
  const bool negphi = (K == 1 || K == 2);
  const double ap = negphi ? -in.phi : in.phi;
  const double sp = negphi ? -in.sp : in.sp;
  PpPolar pol{0.0, 0.0};
  constexpr int SET = pp_formula_set<F>();
  if constexpr (SET >= 0) pol = pp_polar_set(ax, ay, sp, in.cp, SET);
  const PpTuv r = pp_eval<F>(ax, ay, ap, sp, in.cp, pol, A);
  out.nseg = 0;
  if (!r.ok) return;
  const double H = 0.5 * PP_RS_PI;
  double len[5] = {0, 0, 0, 0, 0};
  if constexpr (LAY == PP_TUV) { len[0] = r.t; len[1] = r.u; len[2] = r.v; }
  if constexpr (LAY == PP_VUT) { len[0] = r.v; len[1] = r.u; len[2] = r.t; }
  if constexpr (LAY == PP_T_U_NU_V) { len[0] = r.t; len[1] = r.u; len[2] = -r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_U_U_V) { len[0] = r.t; len[1] = r.u; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_H_U_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_V_U_H_T) { len[0] = r.v; len[1] = r.u; len[2] = -H; len[3] = r.t; }
  if constexpr (LAY == PP_T_H_U_H_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = -H; len[4] = r.v; }
#pragma unroll
  for (int i = 0; i < N; ++i) out.len[i] = (K & 1) ? -len[i] : len[i];
#pragma unroll
  for (int i = N; i < 5; ++i) out.len[i] = 0.0;
  out.word = (K & 2) ? pp_word_reflect(word_k0) : word_k0;
  out.nseg = N;
}

// Wave `g` (0..11) of the workgroup evaluates its formula; lane = mirror image.  raw[] has 46 slots.
__device__ __forceinline__ void pp_rs_eval_wave(const PpRsInput& in, int g, int lane, PpAmb& A, PpRsRaw* raw) {
  if (g == 0) {  // scs :202-211: images 0 and 2 only
    if (lane < 2) pp_eval_image<PP_F_SLS_, false, PP_TUV, 3>(in, 2 * lane, pp_word("SLS"), A, raw[lane]);
    return;
  }
  if (lane >= 4) return;
  switch (g) {
    case 1: pp_eval_image<PP_F_LSL_, false, PP_TUV, 3>(in, lane, pp_word("LSL"), A, raw[2 + lane]); break;
    case 2: pp_eval_image<PP_F_LSR_, false, PP_TUV, 3>(in, lane, pp_word("LSR"), A, raw[6 + lane]); break;
    case 3: pp_eval_image<PP_F_LRL_, false, PP_TUV, 3>(in, lane, pp_word("LRL"), A, raw[10 + lane]); break;
    case 4: pp_eval_image<PP_F_LRL_, true, PP_VUT, 3>(in, lane, pp_word("LRL"), A, raw[14 + lane]); break;
    case 5: pp_eval_image<PP_F_LRLRN_, false, PP_T_U_NU_V, 4>(in, lane, pp_word("LRLR"), A, raw[18 + lane]); break;
    case 6: pp_eval_image<PP_F_LRLRP_, false, PP_T_U_U_V, 4>(in, lane, pp_word("LRLR"), A, raw[22 + lane]); break;
    case 7: pp_eval_image<PP_F_LRSL_, false, PP_T_H_U_V, 4>(in, lane, pp_word("LRSL"), A, raw[26 + lane]); break;
    case 8: pp_eval_image<PP_F_LRSR_, false, PP_T_H_U_V, 4>(in, lane, pp_word("LRSR"), A, raw[30 + lane]); break;
    case 9: pp_eval_image<PP_F_LRSL_, true, PP_V_U_H_T, 4>(in, lane, pp_word("LSRL"), A, raw[34 + lane]); break;
    case 10: pp_eval_image<PP_F_LRSR_, true, PP_V_U_H_T, 4>(in, lane, pp_word("RSRL"), A, raw[38 + lane]); break;
    default: pp_eval_image<PP_F_LRSLR_, false, PP_T_H_U_H_V, 5>(in, lane, pp_word("LRSLR"), A, raw[42 + lane]); break;
  }
}

// The twenty words and the generation indices offered to each, in the reference's order (-1 ends).
constexpr int kRsWordSets = 20;
__device__ constexpr int8_t kRsWordSetGens[kRsWordSets][4] = {
    {0, -1, -1, -1},    {1, -1, -1, -1},                                            // scs
    {2, 3, -1, -1},     {4, 5, -1, -1},     {6, 7, -1, -1},     {8, 9, -1, -1},     // csc
    {10, 11, 14, 15},   {12, 13, 16, 17},                                           // ccc
    {18, 19, 22, 23},   {20, 21, 24, 25},                                           // cccc
    {26, 27, -1, -1},   {28, 29, -1, -1},   {30, 31, -1, -1},   {32, 33, -1, -1},   // ccsc
    {34, 35, -1, -1},   {36, 37, -1, -1},   {38, 39, -1, -1},   {40, 41, -1, -1},   // ccsc, backwards
    {42, 43, -1, -1},   {44, 45, -1, -1}};                                          // ccscc

// set_path (:130-159) for word set `ws` over the stored candidates; the accepted ones go to
// slots[gen].  Returns the accepted bits; *flags gets bit 2 (degenerate) and bit 1 (guard band).
__device__ __forceinline__ unsigned long long pp_rs_commit_set(int ws, const PpRsRaw* raw, PpRsCand* slots,
                                                               uint32_t* flags) {
  PpAmb A{false};
  unsigned long long acc = 0;
  bool degenerate = false;
  int kept[4], nk = 0;
  for (int c = 0; c < 4; ++c) {
    const int gen = kRsWordSetGens[ws][c];
    if (gen < 0) break;
    const PpRsRaw& r = raw[gen];
    const int N = r.nseg;
    if (N == 0) continue;
    bool dup = false;
    for (int e = 0; e < nk && !dup; ++e) {
      double s = 0.0;
      for (int i = 0; i < N; ++i) s += raw[kept[e]].len[i] - r.len[i];
      if (A.le(s, 0.01)) dup = true;  // an equal word exists
    }
    if (dup) continue;
    double total = 0.0;
    for (int i = 0; i < N; ++i) total += fabs(r.len[i]);
    if (A.ge(total, 1000.0)) continue;  // MAX_LENGTH (:6)
    if (!A.ge(total, 0.01)) {
      degenerate = true;
      continue;
    }
    kept[nk++] = gen;
    acc |= 1ull << gen;
    PpRsCand cnd;
    for (int i = 0; i < 5; ++i) cnd.len[i] = r.len[i];
    cnd.total = total;
    cnd.word = r.word;
    cnd.nseg = N;
    slots[gen] = cnd;
  }
  *flags = (degenerate ? 4u : 0u) | (A.amb ? 2u : 0u);
  return acc;
}

// ---------------------------------------------------------------------------------------------
// The same 46 candidates with the reflection as a run-time loop: two copies of each formula's code
// instead of four (the fully unrolled form is a quarter of a megabyte of instructions).  The
// reflected half (images 2 and 3) is the unreflected half run on (x, -y, -phi) with L and R
// swapped, and image K + 1 is image K with x and phi negated and the lengths' signs flipped; a
// word set lives across the two images of a pair (and across the two formulas that share a word),
// exactly as in pp_rs_group.
// The polar forms (pp_polar_set) of images 2h and 2h + 1 of one point: the second image's point is
// the first one's mirrored in x (same y, x negated -- both coordinates are formed here exactly as
// pp_polar_set forms them), so the two share rho, and atan2(y, -x) is pi - atan2(y, x) rounded
// once, which is how the library's atan2 itself gets there: with a0 = atan2(|y|, |x|),
// atan2(y, x) = copysign(x negative ? pi - a0 : a0, y) bit for bit (checked on 2^33 random points
// of four magnitude regimes), zeros of either sign included.  One atan2 and one square root for two.
struct PpPolarPair {
  double rho, th[2];
};
template <int SET>
__device__ __forceinline__ PpPolarPair pp_polar_pair(double bx, double by, double sphi, double cp, int h) {
  // images 2h / 2h + 1: x as is / negated, y negated for h = 1, sin(phi) negated for images 1 and 2
  const double ay = h ? -by : by;
  const double sp_e = h ? -sphi : sphi, sp_o = h ? sphi : -sphi;
  const double px_e = SET ? bx + sp_e : bx - sp_e;
  const double px_o = SET ? -bx + sp_o : -bx - sp_o;
  const double py = SET ? ay - 1.0 - cp : ay - 1.0 + cp;
  const double a0 = pp_atan2(fabs(py), fabs(px_e)), a1 = PP_RS_PI - a0;
  PpPolarPair r;
  r.rho = sqrt(px_e * px_e + py * py);
  r.th[0] = copysign(signbit(px_e) ? a1 : a0, py);
  r.th[1] = copysign(signbit(px_o) ? a1 : a0, py);
  return r;
}

// P (optional): the polar forms of this h, entry (BACK ? 2 : 0) + set.
template <int F, bool BACK, int LAY, int N, class Sink>
__device__ __forceinline__ void pp_candidate_rt(PpRsState& S, const PpRsInput& in, PpWordSet<N>& ws, int K,
                                                int gen, uint32_t word_k0, Sink& sink,
                                                const PpPolarPair* P = nullptr, PpPairMemo* M = nullptr) {
  const double bx = BACK ? in.xb : in.x, by = BACK ? in.yb : in.y;
  const double ax = (K & 1) ? -bx : bx;
  const double ay = (K & 2) ? -by : by;
  const bool negphi = (K == 1 || K == 2);
  const double ap = negphi ? -in.phi : in.phi;
  const double sp = negphi ? -in.sp : in.sp;
  PpPolar pol{0.0, 0.0};
  constexpr int SET = pp_formula_set<F>();
  if constexpr (SET >= 0) {
    if (P) {
      pol.rho = P[(BACK ? 2 : 0) + SET].rho;
      pol.th = P[(BACK ? 2 : 0) + SET].th[K & 1];
    } else {
      pol = pp_polar_set(ax, ay, sp, in.cp, SET);
    }
  }
  PpTuv r = pp_eval<F>(ax, ay, ap, sp, in.cp, pol, S.A, P ? M : nullptr);  // (the memo presumes the shared rho)
  if (!r.ok) return;
  const double H = 0.5 * PP_RS_PI;
  double len[N];
  if constexpr (LAY == PP_TUV) { len[0] = r.t; len[1] = r.u; len[2] = r.v; }
  if constexpr (LAY == PP_VUT) { len[0] = r.v; len[1] = r.u; len[2] = r.t; }
  if constexpr (LAY == PP_T_U_NU_V) { len[0] = r.t; len[1] = r.u; len[2] = -r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_U_U_V) { len[0] = r.t; len[1] = r.u; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_T_H_U_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = r.v; }
  if constexpr (LAY == PP_V_U_H_T) { len[0] = r.v; len[1] = r.u; len[2] = -H; len[3] = r.t; }
  if constexpr (LAY == PP_T_H_U_H_V) { len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = -H; len[4] = r.v; }
  if (K & 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) len[i] = -len[i];
  }
  const uint32_t word = (K & 2) ? pp_word_reflect(word_k0) : word_k0;
  pp_set_path<N>(S, ws, len, gen, word, sink);
}

// Images 2h and 2h + 1 of one formula, offered to one word set.
template <int F, bool BACK, int LAY, int N, class Sink>
__device__ __forceinline__ void pp_pair_rt(PpRsState& S, const PpRsInput& in, PpWordSet<N>& ws, int h, int gen0,
                                           uint32_t word_k0, Sink& sink, const PpPolarPair* P = nullptr) {
  // unrolled: as a loop the two images keep every word set of the enclosing block in registers at
  // once (a few hundred spilled registers, 0.63 -> 0.82 ms per million pairs)
  PpPairMemo memo;
#pragma unroll
  for (int f = 0; f < 2; ++f)
    pp_candidate_rt<F, BACK, LAY, N>(S, in, ws, 2 * h + f, gen0 + 2 * h + f, word_k0, sink, P, &memo);
}

// ---------------------------------------------------------------------------------------------
// The same 46 candidates for a wave of 64 pairs, with the three formulas that only apply to nearby
// goals evaluated apart.  lrl (:188-200), lrlrn (:334-347) and lrlrp (:349-364) start with a range
// test -- the goal within four turning radii or so -- that few pairs of a typical batch pass, yet
// a wave runs a formula's body (an asin, or an acos, two sincos, an atan2 and a cos) whenever one of
// its 64 lanes does: sixteen bodies for a lane or two each.  Here a lane only takes the range test
// and, if it passes, leaves a note (lane, candidate) in the wave's queue; the queue is then worked
// off by all lanes side by side, one body per 64 notes, and every lane picks up its results and
// offers them to set_path in generation order.  The formulas see the same inputs through the same
// code, so the words are the same to the last bit.
struct PpRsDeferred {  // the wave's areas in LDS
  double* pin;         // [7][64]: x, y, phi, sin phi, cos phi, xb, yb of the lanes' pairs
  double* out;         // [8][3][64]: t, u, v of candidate j of a batch, by lane
  uint8_t* flag;       // [8][64]: bit0 the formula accepted, bit1 a comparison fell into the guard band
  uint16_t* queue;     // [512]: lane | j << 6
};
#ifndef PP_RS_LRL_ONE
constexpr int kRsDeferredSlots = 4;
#else
constexpr int kRsDeferredSlots = 8;
#endif
constexpr size_t kRsDeferredBytes = (7 * 64 + kRsDeferredSlots * 3 * 64) * 8 + kRsDeferredSlots * 64 + kRsDeferredSlots * 64 * 2;

__device__ __forceinline__ void pp_wave_sync() {
  // the lanes of a wave run in lock step; this keeps the compiler from moving LDS accesses across
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Works off `count` notes: formula F on image j & 3 of the note's lane, for lrl on the backwards
// point when j >= 4.
template <int F>
__device__ __forceinline__ void pp_deferred_eval(const PpRsDeferred& D, uint32_t count, uint32_t lane) {
  for (uint32_t it = lane; it < count; it += 64u) {
    const uint32_t e = D.queue[it], owner = e & 63u, j = e >> 6;
    const bool back = F == PP_F_LRL_ && j >= 4u;
    const int K = (int)(j & 3u);
    const double bx = D.pin[(back ? 5 : 0) * 64 + owner], by = D.pin[(back ? 6 : 1) * 64 + owner];
    const double phi = D.pin[2 * 64 + owner], sphi = D.pin[3 * 64 + owner], cp = D.pin[4 * 64 + owner];
    const double ax = (K & 1) ? -bx : bx;
    const double ay = (K & 2) ? -by : by;
    const bool negphi = (K == 1 || K == 2);
    const double ap = negphi ? -phi : phi;
    const double sp = negphi ? -sphi : sphi;
    PpPolar pol{0.0, 0.0};
    constexpr int SET = pp_formula_set<F>();
    if constexpr (SET >= 0) pol = pp_polar_set(ax, ay, sp, cp, SET);
    PpAmb A{false};
    const PpTuv r = pp_eval<F>(ax, ay, ap, sp, cp, pol, A);
    const uint32_t sl = j & (uint32_t)(kRsDeferredSlots - 1);
    D.out[(sl * 3 + 0) * 64 + owner] = r.t;
    D.out[(sl * 3 + 1) * 64 + owner] = r.u;
    D.out[(sl * 3 + 2) * 64 + owner] = r.v;
    D.flag[sl * 64 + owner] = (uint8_t)((r.ok ? 1u : 0u) | (A.amb ? 2u : 0u));
  }
}

// A lane's range test of formula F on image K of its (backwards) point: exactly the test the
// formula starts with, on the same expression.
template <int F>
__device__ __forceinline__ bool pp_deferred_wanted(PpRsState& S, const PpRsInput& in, bool back, int K) {
  const double bx = back ? in.xb : in.x, by = back ? in.yb : in.y;
  const double ax = (K & 1) ? -bx : bx;
  const double ay = (K & 2) ? -by : by;
  const double sp = (K == 1 || K == 2) ? -in.sp : in.sp;
  if constexpr (F == PP_F_LRL_) {
    const double px = ax - sp, py = ay - 1.0 + in.cp;  // pp_polar_set, set 0
    return S.A.le(sqrt(px * px + py * py), 4.);
  } else if constexpr (F == PP_F_LRLRN_) {
    const double px = ax + sp, py = ay - 1.0 - in.cp;  // set 1
    return S.A.le(0.25 * (2.0 + sqrt(px * px + py * py)), 1.0);
  } else {
    const double xi = ax + sp, eta = ay - 1.0 - in.cp;
    const double rho = (20. - xi * xi - eta * eta) / 16.0;
    return S.A.le(0.0, rho) & S.A.le(rho, 1.0);
  }
}

// One batch: the NC candidates of formula F (the four images; lrl: forward, then backwards),
// tested, queued, evaluated by the wave and offered in generation order to the word set of their
// letters (images 0, 1: wa, images 2, 3, the reflected word: wb).  Every lane of the wave must come
// here together.
template <int F, int J0, int NC, int N, class Sink>
__device__ __forceinline__ void pp_deferred_batch(PpRsState& S, const PpRsInput& in, const PpRsDeferred& D,
                                                  PpWordSet<N>& wa, PpWordSet<N>& wb, uint32_t lane,
                                                  uint32_t word_k0, Sink& sink) {
  uint32_t count = 0, mine = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = J0; j < J0 + NC; ++j) {
    const bool need = pp_deferred_wanted<F>(S, in, F == PP_F_LRL_ && j >= 4, j & 3);
    const unsigned long long m = __ballot(need);
    if (need) {
      D.queue[count + (uint32_t)__popcll(m & below)] = (uint16_t)(lane | ((uint32_t)j << 6));
      mine |= 1u << j;
    }
    count += (uint32_t)__popcll(m);
  }
  if (count == 0) return;  // uniform over the wave
  pp_wave_sync();
  pp_deferred_eval<F>(D, count, lane);
  pp_wave_sync();
#pragma unroll
  for (int j = J0; j < J0 + NC; ++j) {
    if (!((mine >> j) & 1u)) continue;
    constexpr int kSlotMask = kRsDeferredSlots - 1;
    const uint32_t f = D.flag[(j & kSlotMask) * 64 + lane];
    if (f & 2u) S.A.amb = true;
    if (!(f & 1u)) continue;
    const double t = D.out[((j & kSlotMask) * 3 + 0) * 64 + lane], u = D.out[((j & kSlotMask) * 3 + 1) * 64 + lane],
                 v = D.out[((j & kSlotMask) * 3 + 2) * 64 + lane];
    const int K = j & 3;
    double len[N];
    int gen;
    if constexpr (F == PP_F_LRL_) {
      if (j < 4) { len[0] = t; len[1] = u; len[2] = v; gen = 10 + K; }   // PP_TUV
      else { len[0] = v; len[1] = u; len[2] = t; gen = 14 + K; }         // PP_VUT, backwards
    } else if constexpr (F == PP_F_LRLRN_) {
      len[0] = t; len[1] = u; len[2] = -u; len[3] = v; gen = 18 + K;      // PP_T_U_NU_V
    } else {
      len[0] = t; len[1] = u; len[2] = u; len[3] = v; gen = 22 + K;       // PP_T_U_U_V
    }
    if (K & 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) len[i] = -len[i];
    }
    if (K & 2)
      pp_set_path<N>(S, wb, len, gen, pp_word_reflect(word_k0), sink);
    else
      pp_set_path<N>(S, wa, len, gen, word_k0, sink);
  }
  pp_wave_sync();  // the areas are written again by the next batch
}

// The wave's areas carved out of kRsDeferredBytes of LDS (16-byte aligned).
__device__ __forceinline__ PpRsDeferred pp_deferred_areas(unsigned char* mem) {
  PpRsDeferred D;
  D.pin = (double*)mem;
  D.out = D.pin + 7 * 64;
  D.flag = (uint8_t*)(D.out + kRsDeferredSlots * 3 * 64);
  D.queue = (uint16_t*)(D.flag + kRsDeferredSlots * 64);
  return D;
}
// The lanes' pairs, for the lanes that will work on one another's candidates.
__device__ __forceinline__ void pp_deferred_fill(const PpRsDeferred& D, const PpRsInput& in, uint32_t lane) {
  D.pin[0 * 64 + lane] = in.x;
  D.pin[1 * 64 + lane] = in.y;
  D.pin[2 * 64 + lane] = in.phi;
  D.pin[3 * 64 + lane] = in.sp;
  D.pin[4 * 64 + lane] = in.cp;
  D.pin[5 * 64 + lane] = in.xb;
  D.pin[6 * 64 + lane] = in.yb;
}
// ccc :272-316 -- LRL / RLR collect four candidates each (10, 11, 14, 15 / 12, 13, 16, 17); after
// pp_deferred_fill, every lane of the wave together.
template <class Sink>
__device__ __forceinline__ void pp_deferred_ccc(PpRsState& S, const PpRsInput& in, const PpRsDeferred& D,
                                                uint32_t lane, Sink& sink) {
  PpWordSet<3> a, b;
#ifndef PP_RS_LRL_ONE
  pp_deferred_batch<PP_F_LRL_, 0, 4, 3>(S, in, D, a, b, lane, pp_word("LRL"), sink);
  pp_deferred_batch<PP_F_LRL_, 4, 4, 3>(S, in, D, a, b, lane, pp_word("LRL"), sink);
#else
  pp_deferred_batch<PP_F_LRL_, 0, 8, 3>(S, in, D, a, b, lane, pp_word("LRL"), sink);
#endif
}
// cccc :366-406 -- LRLR / RLRL (18, 19, 22, 23 / 20, 21, 24, 25)
template <class Sink>
__device__ __forceinline__ void pp_deferred_cccc(PpRsState& S, const PpRsInput& in, const PpRsDeferred& D,
                                                 uint32_t lane, Sink& sink) {
  PpWordSet<4> a, b;
  pp_deferred_batch<PP_F_LRLRN_, 0, 4, 4>(S, in, D, a, b, lane, pp_word("LRLR"), sink);
  pp_deferred_batch<PP_F_LRLRP_, 0, 4, 4>(S, in, D, a, b, lane, pp_word("LRLR"), sink);
}

// generate_path for the 64 pairs of a wave (every lane of the wave takes part, whether its pair
// counts or not); candidates reach the sink word by word, not in generation order.
template <class Sink>
__device__ __forceinline__ void pp_rs_generate_wave(const PpRsInput& in, PpRsState& S, const PpRsDeferred& D,
                                                    uint32_t lane, Sink& sink) {
  pp_deferred_fill(D, in, lane);
#pragma clang loop unroll(disable)
  for (int h = 0; h < 2; ++h) {
    // the four points the formulas of this h measure from, images 2h and 2h + 1 together
    const PpPolarPair P[4] = {pp_polar_pair<0>(in.x, in.y, in.sp, in.cp, h), pp_polar_pair<1>(in.x, in.y, in.sp, in.cp, h),
                              pp_polar_pair<0>(in.xb, in.yb, in.sp, in.cp, h),
                              pp_polar_pair<1>(in.xb, in.yb, in.sp, in.cp, h)};
    {  // scs :202-211 -- generation indices 0 (image 0) and 1 (image 2)
      PpWordSet<3> a;
      pp_candidate_rt<PP_F_SLS_, false, PP_TUV, 3>(S, in, a, 2 * h, h, pp_word("SLS"), sink);
    }
    {  // csc :231-270 -- 2..5, 6..9
      PpWordSet<3> a, c;
      pp_pair_rt<PP_F_LSL_, false, PP_TUV, 3>(S, in, a, h, 2, pp_word("LSL"), sink, P);
      pp_pair_rt<PP_F_LSR_, false, PP_TUV, 3>(S, in, c, h, 6, pp_word("LSR"), sink, P);
    }
    {  // ccsc :440-524 -- 26..29, 30..33, and on the backwards point 34..37, 38..41
      PpWordSet<4> a, c;
      pp_pair_rt<PP_F_LRSL_, false, PP_T_H_U_V, 4>(S, in, a, h, 26, pp_word("LRSL"), sink, P);
      pp_pair_rt<PP_F_LRSR_, false, PP_T_H_U_V, 4>(S, in, c, h, 30, pp_word("LRSR"), sink, P);
    }
    {
      PpWordSet<4> a, c;
      pp_pair_rt<PP_F_LRSL_, true, PP_V_U_H_T, 4>(S, in, a, h, 34, pp_word("LSRL"), sink, P);
      pp_pair_rt<PP_F_LRSR_, true, PP_V_U_H_T, 4>(S, in, c, h, 38, pp_word("RSRL"), sink, P);
    }
    {  // ccscc :544-580 -- 42..45
      PpWordSet<5> a;
      pp_pair_rt<PP_F_LRSLR_, false, PP_T_H_U_H_V, 5>(S, in, a, h, 42, pp_word("LRSLR"), sink, P);
    }
  }
  pp_deferred_ccc(S, in, D, lane, sink);
  pp_deferred_cccc(S, in, D, lane, sink);
}

// generate_path, :707-725: the families in the reference's order.
template <class Sink>
__device__ __forceinline__ void pp_rs_generate(const PpRsInput& in, PpRsState& S, Sink& sink) {
  pp_rs_family<0>(in, S, sink);
  pp_rs_family<1>(in, S, sink);
  pp_rs_family<2>(in, S, sink);
  pp_rs_family<3>(in, S, sink);
  pp_rs_family<4>(in, S, sink);
  pp_rs_family<5>(in, S, sink);
  pp_rs_family<6>(in, S, sink);
}

#endif
