// pp_oracle.cpp -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY;
// see pp_oracle.h).  Scalar C++17, glibc libm, build with -ffp-contract=off.
//
// Rust -> C++ conventions used throughout:
//   f64::round            -> std::round  (half away from zero)
//   `as usize` / `as isize` on f64 -> saturating casts (NaN -> 0)
//   usize arithmetic in release builds -> uint64_t wrapping arithmetic
//   f64 % f64             -> std::fmod (truncated)
//   x.powi(2)             -> x * x
//   BinaryHeap<(Reverse<NotNan<f64>>, usize)> -> pop min cost, ties -> LARGEST index
#include "pp_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>
#include <unordered_map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr double kPi = 3.14159265358979323846;  // std::f64::consts::PI
constexpr double kInf = std::numeric_limits<double>::infinity();

inline uint64_t sat_usize(double v) {
  if (!(v == v)) return 0;  // NaN
  if (v <= 0.0) return 0;
  if (v >= 18446744073709551615.0) return UINT64_MAX;
  return (uint64_t)v;
}
inline int64_t sat_isize(double v) {
  if (!(v == v)) return 0;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  if (v >= 9223372036854775807.0) return INT64_MAX;
  return (int64_t)v;
}

// ---------------------------------------------------------------- reeds_shepp.rs:727-757
double pi_2_pi(double theta) {
  while (theta > kPi) theta -= 2.0 * kPi;
  while (theta < -kPi) theta += 2.0 * kPi;
  return theta;
}
inline void polar(double x, double y, double& r, double& theta) {  // fn r, :740-745
  r = std::hypot(x, y);
  theta = std::atan2(y, x);
}
inline double mod2pi(double theta) {  // fn m, :747-757
  double phi = std::fmod(theta, 2.0 * kPi);
  if (phi < -kPi) phi += 2.0 * kPi;
  if (phi > kPi) phi -= 2.0 * kPi;
  return phi;
}

struct Tuv {
  bool ok;
  double t, u, v;
};
const Tuv kNo{false, 0., 0., 0.};

// reeds_shepp.rs:213-229
Tuv f_sls(double x, double y, double phi) {
  double p = mod2pi(phi);
  if ((y > 0.) & (0. < p) & (p < kPi * 0.99)) {
    double xd = -y / std::tan(p) + x;
    double t = xd - std::tan(p / 2.);
    double v = std::sqrt((x - xd) * (x - xd) + y * y) - std::tan(p / 2.);
    return {true, t, p, v};
  } else if ((y < 0.) & (0. < p) & (p < kPi * 0.99)) {
    double xd = -y / std::tan(p) + x;
    double t = xd - std::tan(p / 2.);
    double v = -std::sqrt((x - xd) * (x - xd) + y * y) - std::tan(p / 2.);
    return {true, t, p, v};
  }
  return kNo;
}
// :161-170
Tuv f_lsl(double x, double y, double phi) {
  double u, t;
  polar(x - std::sin(phi), y - 1.0 + std::cos(phi), u, t);
  if (t >= 0.0) {
    double v = mod2pi(phi - t);
    if (v >= 0.0) return {true, t, u, v};
  }
  return kNo;
}
// :172-186
Tuv f_lsr(double x, double y, double phi) {
  double u1, t1;
  polar(x + std::sin(phi), y - 1.0 - std::cos(phi), u1, t1);
  u1 = u1 * u1;
  if (u1 >= 4.) {
    double u = std::sqrt(u1 - 4.);
    double theta = std::atan2(2.0, u);
    double t = mod2pi(t1 + theta);
    double v = mod2pi(t - phi);
    if ((t >= 0.) & (v >= 0.0)) return {true, t, u, v};
  }
  return kNo;
}
// :188-200
Tuv f_lrl(double x, double y, double phi) {
  double u1, t1;
  polar(x - std::sin(phi), y - 1.0 + std::cos(phi), u1, t1);
  if (u1 <= 4.) {
    double u = -2. * std::asin(0.25 * u1);
    double t = mod2pi(t1 + 0.5 * u + kPi);
    double v = mod2pi(phi - t + u);
    if ((t >= 0.) & (u <= 0.0)) return {true, t, u, v};
  }
  return kNo;
}
// :318-332
void tau_omega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
  double delta = mod2pi(u - v);
  double a = std::sin(u) - std::sin(delta);
  double b = std::cos(u) - std::cos(delta) - 1.0;
  double t1 = std::atan2(eta * a - xi * b, xi * a + eta * b);
  double t2 = 2.0 * (std::cos(delta) - std::cos(v) - std::cos(u)) + 3.0;
  tau = (t2 < 0.) ? mod2pi(t1 + kPi) : mod2pi(t1);
  omega = mod2pi(tau - u + v - phi);
}
// :334-347
Tuv f_lrlrn(double x, double y, double phi) {
  double xi = x + std::sin(phi);
  double eta = y - 1.0 - std::cos(phi);
  double rho = 0.25 * (2.0 + std::sqrt(xi * xi + eta * eta));
  if (rho <= 1.0) {
    double u = std::acos(rho), t, v;
    tau_omega(u, -u, xi, eta, phi, t, v);
    if ((t >= 0.) & (v <= 0.)) return {true, t, u, v};
  }
  return kNo;
}
// :349-364
Tuv f_lrlrp(double x, double y, double phi) {
  double xi = x + std::sin(phi);
  double eta = y - 1.0 - std::cos(phi);
  double rho = (20. - xi * xi - eta * eta) / 16.0;
  if ((0.0 <= rho) & (rho <= 1.0)) {
    double u = -std::acos(rho);
    if (u >= -0.5 * kPi) {
      double t, v;
      tau_omega(u, u, xi, eta, phi, t, v);
      if ((t >= 0.) & (v >= 0.)) return {true, t, u, v};
    }
  }
  return kNo;
}
// :408-422
Tuv f_lrsr(double x, double y, double phi) {
  double xi = x + std::sin(phi);
  double eta = y - 1.0 - std::cos(phi);
  double rho, theta;
  polar(-eta, xi, rho, theta);
  if (rho >= 2.0) {
    double t = theta;
    double u = 2.0 - rho;
    double v = mod2pi(t + 0.5 * kPi - phi);
    if ((t >= 0.0) & (u <= 0.0) & (v <= 0.0)) return {true, t, u, v};
  }
  return kNo;
}
// :423-438
Tuv f_lrsl(double x, double y, double phi) {
  double xi = x - std::sin(phi);
  double eta = y - 1.0 + std::cos(phi);
  double rho, theta;
  polar(xi, eta, rho, theta);
  if (rho >= 2.0) {
    double r = std::sqrt(rho * rho - 4.0);
    double u = 2.0 - r;
    double t = mod2pi(theta + std::atan2(r, -2.0));
    double v = mod2pi(phi - 0.5 * kPi - t);
    if ((t >= 0.0) & (u <= 0.0) & (v <= 0.0)) return {true, t, u, v};
  }
  return kNo;
}
// :526-542
Tuv f_lrslr(double x, double y, double phi) {
  double xi = x + std::sin(phi);
  double eta = y - 1.0 - std::cos(phi);
  double rho, theta;
  polar(xi, eta, rho, theta);
  if (rho >= 2.0) {
    double u = 4.0 - std::sqrt(rho * rho - 4.0);
    if (u <= 0.) {
      double t = mod2pi(std::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
      double v = mod2pi(t - phi);
      if ((t >= 0.0) & (v >= 0.0)) return {true, t, u, v};
    }
  }
  return kNo;
}

// One accepted word, lengths still in normalised (curvature) units.
struct Word {
  int n;
  double len[5];
  char ct[5];
  double total;
};

// reeds_shepp.rs:130-159.  Returns false when the assert! at :157 would fire.
bool set_path(std::vector<Word>& paths, const double* len, const char* ct, int n) {
  for (const Word& e : paths) {
    if (e.n == n && std::memcmp(e.ct, ct, n) == 0) {
      double s = 0.0;
      for (int i = 0; i < n; ++i) s += e.len[i] - len[i];
      if (s <= 0.01) return true;  // an equal word exists: not inserted
    }
  }
  double total = 0.0;
  for (int i = 0; i < n; ++i) total += std::fabs(len[i]);
  if (total >= 1000.0) return true;  // MAX_LENGTH, :6
  if (!(total >= 0.01)) return false;
  Word w{};
  w.n = n;
  for (int i = 0; i < n; ++i) {
    w.len[i] = len[i];
    w.ct[i] = ct[i];
  }
  w.total = total;
  paths.push_back(w);
  return true;
}

// The 46 call sites of scs/csc/ccc/cccc/ccsc/ccscc (:202-211, :231-316, :366-406, :440-524,
// :544-580) follow one scheme: a base formula is evaluated on four mirror images of (x, y, phi),
//   k=0 ( x,  y,  phi)  lengths +, word as written
//   k=1 (-x,  y, -phi)  lengths -, word as written          ("time flip")
//   k=2 ( x, -y, -phi)  lengths +, word with L<->R swapped  ("reflect")
//   k=3 (-x, -y,  phi)  lengths -, word with L<->R swapped
// optionally on the "backwards" point (xb, yb) (:294-295, :482-483), and (t, u, v) is laid out
// into 3-5 segment lengths by one of seven patterns.
enum Formula { F_SLS, F_LSL, F_LSR, F_LRL, F_LRLRN, F_LRLRP, F_LRSL, F_LRSR, F_LRSLR };
enum Layout { TUV, VUT, T_U_NU_V, T_U_U_V, T_H_U_V, V_U_H_T, T_H_U_H_V };
struct Group {
  Formula f;
  bool backwards;
  Layout lay;
  const char* word;  // for k = 0, 1
  int kmask;         // which mirror images are generated
};
const Group kGroups[] = {
    {F_SLS, false, TUV, "SLS", 0b0101},       // scs :202-211 (only k=0 and k=2)
    {F_LSL, false, TUV, "LSL", 0b1111},       // csc :232-249
    {F_LSR, false, TUV, "LSR", 0b1111},       // csc :251-269
    {F_LRL, false, TUV, "LRL", 0b1111},       // ccc :273-291
    {F_LRL, true, VUT, "LRL", 0b1111},        // ccc :297-315
    {F_LRLRN, false, T_U_NU_V, "LRLR", 0b1111},  // cccc :367-385
    {F_LRLRP, false, T_U_U_V, "LRLR", 0b1111},   // cccc :387-405
    {F_LRSL, false, T_H_U_V, "LRSL", 0b1111},    // ccsc :441-459
    {F_LRSR, false, T_H_U_V, "LRSR", 0b1111},    // ccsc :461-479
    {F_LRSL, true, V_U_H_T, "LSRL", 0b1111},     // ccsc :485-503
    {F_LRSR, true, V_U_H_T, "RSRL", 0b1111},     // ccsc :505-523
    {F_LRSLR, false, T_H_U_H_V, "LRSLR", 0b1111},  // ccscc :545-579
};

Tuv eval_formula(Formula f, double x, double y, double phi) {
  switch (f) {
    case F_SLS: return f_sls(x, y, phi);
    case F_LSL: return f_lsl(x, y, phi);
    case F_LSR: return f_lsr(x, y, phi);
    case F_LRL: return f_lrl(x, y, phi);
    case F_LRLRN: return f_lrlrn(x, y, phi);
    case F_LRLRP: return f_lrlrp(x, y, phi);
    case F_LRSL: return f_lrsl(x, y, phi);
    case F_LRSR: return f_lrsr(x, y, phi);
    case F_LRSLR: return f_lrslr(x, y, phi);
  }
  return kNo;
}

// reeds_shepp.rs:707-725.  Returns false if the :157 assert would fire.
bool generate_path(const double q0[3], const double q1[3], double maxc, std::vector<Word>& paths) {
  double dx = q1[0] - q0[0];
  double dy = q1[1] - q0[1];
  double dth = q1[2] - q0[2];
  double c = std::cos(q0[2]);
  double s = std::sin(q0[2]);
  double x = (c * dx + s * dy) * maxc;
  double y = (-s * dx + c * dy) * maxc;
  double phi = dth;
  double xb = x * std::cos(phi) + y * std::sin(phi);
  double yb = x * std::sin(phi) - y * std::cos(phi);
  const double H = 0.5 * kPi;

  paths.clear();
  for (const Group& g : kGroups) {
    double bx = g.backwards ? xb : x;
    double by = g.backwards ? yb : y;
    for (int k = 0; k < 4; ++k) {
      if (!((g.kmask >> k) & 1)) continue;
      double ax = (k & 1) ? -bx : bx;
      double ay = (k & 2) ? -by : by;
      double ap = (k == 1 || k == 2) ? -phi : phi;
      Tuv r = eval_formula(g.f, ax, ay, ap);
      if (!r.ok) continue;
      double len[5];
      int n = 0;
      switch (g.lay) {
        case TUV: len[0] = r.t; len[1] = r.u; len[2] = r.v; n = 3; break;
        case VUT: len[0] = r.v; len[1] = r.u; len[2] = r.t; n = 3; break;
        case T_U_NU_V: len[0] = r.t; len[1] = r.u; len[2] = -r.u; len[3] = r.v; n = 4; break;
        case T_U_U_V: len[0] = r.t; len[1] = r.u; len[2] = r.u; len[3] = r.v; n = 4; break;
        case T_H_U_V: len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = r.v; n = 4; break;
        case V_U_H_T: len[0] = r.v; len[1] = r.u; len[2] = -H; len[3] = r.t; n = 4; break;
        case T_H_U_H_V: len[0] = r.t; len[1] = -H; len[2] = r.u; len[3] = -H; len[4] = r.v; n = 5; break;
      }
      if (k & 1)
        for (int i = 0; i < n; ++i) len[i] = -len[i];
      char ct[5];
      for (int i = 0; i < n; ++i) {
        char ch = g.word[i];
        if (k & 2) ch = (ch == 'L') ? 'R' : (ch == 'R') ? 'L' : ch;
        ct[i] = ch;
      }
      if (!set_path(paths, len, ct, n)) return false;
    }
  }
  return true;
}

// reeds_shepp.rs:665-705
inline void interpolate(double l, char m, double maxc, double ox, double oy, double oyaw,
                        double& px, double& py, double& pyaw, int64_t& dir) {
  if (m == 'S') {
    px = ox + l / maxc * std::cos(oyaw);
    py = oy + l / maxc * std::sin(oyaw);
    pyaw = oyaw;
  } else {
    double ldx = std::sin(l) / maxc;
    double ldy = (m == 'L') ? (1.0 - std::cos(l)) / maxc : -(1.0 - std::cos(l)) / maxc;
    double gdx = std::cos(-oyaw) * ldx + std::sin(-oyaw) * ldy;
    double gdy = -std::sin(-oyaw) * ldx + std::cos(-oyaw) * ldy;
    px = ox + gdx;
    py = oy + gdy;
  }
  if (m == 'L')
    pyaw = oyaw + l;
  else if (m == 'R')
    pyaw = oyaw - l;
  dir = (l > 0.0) ? 1 : -1;
}

// reeds_shepp.rs:581-663
void local_course(const Word& w, double maxc, double step, std::vector<double>& px,
                  std::vector<double>& py, std::vector<double>& pyaw, std::vector<int64_t>& dirs) {
  size_t point_num = (size_t)sat_usize(w.total / step) + (size_t)w.n + 3;
  px.assign(point_num, 0.0);
  py.assign(point_num, 0.0);
  pyaw.assign(point_num, 0.0);
  dirs.assign(point_num, 0);
  size_t ind = 1;
  dirs[0] = (w.len[0] > 0.0) ? 1 : -1;
  double ll = 0.0;
  for (int i = 0; i < w.n; ++i) {
    char m = w.ct[i];
    double l = w.len[i];
    double d = (l > 0.0) ? step : -step;
    double ox = px[ind], oy = py[ind], oyaw = pyaw[ind];
    ind -= 1;
    double pd;
    if (i >= 1 && w.len[i - 1] * w.len[i] > 0.)
      pd = -d - ll;
    else
      pd = d - ll;
    while (std::fabs(pd) <= std::fabs(l)) {
      ind += 1;
      interpolate(pd, m, maxc, ox, oy, oyaw, px[ind], py[ind], pyaw[ind], dirs[ind]);
      pd += d;
    }
    ll = l - pd - d;
    ind += 1;
    interpolate(l, m, maxc, ox, oy, oyaw, px[ind], py[ind], pyaw[ind], dirs[ind]);
  }
  while (!px.empty() && px.back() == 0.0) {  // :656-661 (Vec::last().unwrap() panics if emptied)
    px.pop_back();
    py.pop_back();
    pyaw.pop_back();
    dirs.pop_back();
  }
}

struct FullPath {
  Word w;  // lengths de-normalised (metres)
  std::vector<double> x, y, yaw;
  std::vector<int64_t> dir;
};

// reeds_shepp.rs:87-128
bool calc_all_paths(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                    double maxc, double step_size, std::vector<FullPath>& out, bool want_traj) {
  double q0[3] = {sx, sy, syaw}, q1[3] = {gx, gy, gyaw};
  std::vector<Word> words;
  if (!generate_path(q0, q1, maxc, words)) return false;
  out.clear();
  out.resize(words.size());
  std::vector<double> lx, ly, lyaw;
  for (size_t k = 0; k < words.size(); ++k) {
    FullPath& p = out[k];
    if (want_traj) {
      local_course(words[k], maxc, step_size * maxc, lx, ly, lyaw, p.dir);
      size_t n = lx.size();
      p.x.resize(n);
      p.y.resize(n);
      p.yaw.resize(n);
      for (size_t i = 0; i < n; ++i) {
        p.x[i] = lx[i] * std::cos(-q0[2]) + ly[i] * std::sin(-q0[2]) + q0[0];
        p.y[i] = -lx[i] * std::sin(-q0[2]) + ly[i] * std::cos(-q0[2]) + q0[1];
        p.yaw[i] = pi_2_pi(lyaw[i] + q0[2]);
      }
    }
    p.w = words[k];
    for (int i = 0; i < p.w.n; ++i) p.w.len[i] = p.w.len[i] / maxc;
    p.w.total = p.w.total / maxc;
  }
  return true;
}

void to_abi(const Word& w, pp_rs_word* o) {
  std::memset(o, 0, sizeof(*o));
  for (int i = 0; i < w.n; ++i) {
    o->lengths[i] = w.len[i];
    o->ctypes[i] = (uint8_t)w.ct[i];
  }
  o->nseg = (uint8_t)w.n;
  o->total_length = w.total;
}

// hybrid_astar.rs:450-489
double rs_path_cost(const double* len, const uint8_t* ct, int n, const pp_config& conf) {
  double cost = 0.0;
  for (int i = 0; i < n; ++i) {
    if (len[i] >= 0.0)
      cost += 1.0;
    else
      cost += std::fabs(len[i]) * conf.backward_cost;
  }
  for (int i = 0; i + 1 < n; ++i)
    if (len[i] * len[i + 1] < 0.0) cost += conf.gear_cost;
  for (int i = 0; i < n; ++i)
    if (ct[i] != 'S') cost += conf.steer_angle_cost * std::fabs(conf.max_steer);
  double ulist[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (ct[i] == 'R')
      ulist[i] = -conf.max_steer;
    else if (ct[i] == 'L')
      ulist[i] = conf.max_steer;
  }
  for (int i = 0; i + 1 < n; ++i) cost += conf.steer_change_cost * std::fabs(ulist[i + 1] - ulist[i]);
  return cost;
}

}  // namespace

// ------------------------------------------------------------------ the KD-tree form
// The reference finds an obstacle's neighbours with the `kdtree` crate (Cargo.toml:7, pinned to
// 0.7.0 in Cargo.lock; not vendored): `KdTree::new(2)` + `add` per obstacle (hybrid_astar.rs:136-139)
// and `within(&[cx, cy], bound_radius, &euclid)` per pose (:426-431).  This is a restatement of that
// crate's published algorithm, for the CPU baseline only: leaves of up to 16 points; a full leaf
// splits on the dimension of widest bounds at the midpoint of those bounds (points below the split
// value go left); `within` walks the tree best-first over the distance from the query to each
// node's bounding box, keeps every point whose distance is <= radius and returns them sorted by
// distance in a freshly allocated vector.  The flags do not depend on any of this (the radius query
// is a pre-filter for the rectangle test), which the CPU tests check against the other two forms.
namespace {
class KdTree2 {
 public:
  void add(double px, double py, uint32_t data) {
    if (nodes_.empty()) nodes_.emplace_back();
    int cur = 0;
    for (;;) {
      Node& n = nodes_[cur];
      n.minb[0] = std::min(n.minb[0], px);
      n.maxb[0] = std::max(n.maxb[0], px);
      n.minb[1] = std::min(n.minb[1], py);
      n.maxb[1] = std::max(n.maxb[1], py);
      ++n.size;
      if (n.left < 0) break;
      cur = ((n.dim == 0 ? px : py) < n.split) ? n.left : n.right;
    }
    nodes_[cur].pts.push_back({px, py, data});
    if (nodes_[cur].pts.size() > kCapacity) split(cur);
  }
  struct Hit {
    double dist;
    uint32_t data;
  };
  // sorted by distance, as the crate returns them
  std::vector<Hit> within(double qx, double qy, double radius) const {
    std::vector<Hit> out;
    if (nodes_.empty()) return out;
    typedef std::pair<double, int> Pend;  // (distance to the node's box, node)
    std::priority_queue<Pend, std::vector<Pend>, std::greater<Pend>> pending;
    pending.push({0.0, 0});
    while (!pending.empty() && pending.top().first <= radius) {
      int cur = pending.top().second;
      pending.pop();
      while (nodes_[cur].left >= 0) {
        const Node& n = nodes_[cur];
        const bool go_left = (n.dim == 0 ? qx : qy) < n.split;
        const int near = go_left ? n.left : n.right, far = go_left ? n.right : n.left;
        const double d = box_distance(qx, qy, nodes_[far]);
        if (d <= radius) pending.push({d, far});
        cur = near;
      }
      for (const Pt& p : nodes_[cur].pts) {
        const double d = std::sqrt((qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y));  // :428-430
        if (d <= radius) out.push_back({d, p.data});
      }
    }
    std::sort(out.begin(), out.end(), [](const Hit& a, const Hit& b) { return a.dist < b.dist; });
    return out;
  }
  bool empty() const { return nodes_.empty(); }

 private:
  static constexpr size_t kCapacity = 16;
  struct Pt {
    double x, y;
    uint32_t data;
  };
  struct Node {
    int left = -1, right = -1, dim = 0;
    double split = 0.0;
    double minb[2] = {kInf, kInf}, maxb[2] = {-kInf, -kInf};
    size_t size = 0;
    std::vector<Pt> pts;
  };
  static double box_distance(double qx, double qy, const Node& n) {
    const double cx = std::min(std::max(qx, n.minb[0]), n.maxb[0]);
    const double cy = std::min(std::max(qy, n.minb[1]), n.maxb[1]);
    return std::sqrt((qx - cx) * (qx - cx) + (qy - cy) * (qy - cy));
  }
  void split(int cur) {
    Node& n = nodes_[cur];
    const int dim = (n.maxb[1] - n.minb[1]) > (n.maxb[0] - n.minb[0]) ? 1 : 0;
    if (!(n.maxb[dim] > n.minb[dim])) return;  // all points coincide (the demo map's doubled corners): stays a leaf
    const double sv = n.minb[dim] + (n.maxb[dim] - n.minb[dim]) / 2.0;
    std::vector<Pt> pts;
    pts.swap(n.pts);
    const int l = (int)nodes_.size();
    nodes_.emplace_back();
    nodes_.emplace_back();
    Node& m = nodes_[cur];  // the vector may have moved
    m.dim = dim;
    m.split = sv;
    m.left = l;
    m.right = l + 1;
    for (const Pt& p : pts) {
      Node& c = nodes_[((dim == 0 ? p.x : p.y) < sv) ? l : l + 1];
      c.minb[0] = std::min(c.minb[0], p.x);
      c.maxb[0] = std::max(c.maxb[0], p.x);
      c.minb[1] = std::min(c.minb[1], p.y);
      c.maxb[1] = std::max(c.maxb[1], p.y);
      ++c.size;
      c.pts.push_back(p);
    }
  }
  std::vector<Node> nodes_;
};
}  // namespace

// ------------------------------------------------------------------ obstacle map
struct orc_map {
  KdTree2 kd;  // built by orc_map_create, used by mode 2 of the collision entry points
  std::vector<double> ox, oy;
  // bucket grid over unit squares: bucket (bx, by) holds the indices of points with
  // floor(x) == bx0 + bx, floor(y) == by0 + by.
  int64_t bx0 = 0, by0 = 0, nbx = 0, nby = 0;
  std::vector<uint32_t> start;  // CSR
  std::vector<uint32_t> items;
  bool finite = true;
};

orc_map* orc_map_create(const double* ox, const double* oy, uint64_t n) {
  orc_map* m = new orc_map();
  m->ox.assign(ox, ox + n);
  m->oy.assign(oy, oy + n);
  double mnx = kInf, mny = kInf, mxx = -kInf, mxy = -kInf;
  for (uint64_t i = 0; i < n; ++i) {
    if (!std::isfinite(ox[i]) || !std::isfinite(oy[i])) m->finite = false;
    mnx = std::min(mnx, ox[i]);
    mxx = std::max(mxx, ox[i]);
    mny = std::min(mny, oy[i]);
    mxy = std::max(mxy, oy[i]);
  }
  if (n == 0 || !m->finite) return m;
  for (uint64_t i = 0; i < n; ++i) m->kd.add(ox[i], oy[i], (uint32_t)i);  // hybrid_astar.rs:136-139
  m->bx0 = (int64_t)std::floor(mnx);
  m->by0 = (int64_t)std::floor(mny);
  m->nbx = (int64_t)std::floor(mxx) - m->bx0 + 1;
  m->nby = (int64_t)std::floor(mxy) - m->by0 + 1;
  size_t nb = (size_t)m->nbx * (size_t)m->nby;
  m->start.assign(nb + 1, 0);
  auto bucket = [&](uint64_t i) {
    int64_t bx = (int64_t)std::floor(ox[i]) - m->bx0;
    int64_t by = (int64_t)std::floor(oy[i]) - m->by0;
    return (size_t)by * (size_t)m->nbx + (size_t)bx;
  };
  for (uint64_t i = 0; i < n; ++i) m->start[bucket(i) + 1]++;
  for (size_t b = 0; b < nb; ++b) m->start[b + 1] += m->start[b];
  m->items.resize(n);
  std::vector<uint32_t> fill(m->start.begin(), m->start.end() - 1);
  for (uint64_t i = 0; i < n; ++i) m->items[fill[bucket(i)]++] = (uint32_t)i;
  return m;
}
void orc_map_destroy(orc_map* m) { delete m; }

// hybrid_astar.rs:522-555
int32_t orc_map_param(const orc_map* m, const pp_config* conf, pp_param* out) {
  if (!m || !conf || !out) return PP_ERR_INVALID;
  double mnx = kInf, mny = kInf, mxx = -kInf, mxy = -kInf;
  for (size_t i = 0; i < m->ox.size(); ++i) {
    mnx = std::fmin(mnx, m->ox[i]);  // f64::min ignores NaN like fmin
    mxx = std::fmax(mxx, m->ox[i]);
    mny = std::fmin(mny, m->oy[i]);
    mxy = std::fmax(mxy, m->oy[i]);
  }
  double xyreso = conf->xy_reso, yawreso = conf->yaw_reso;
  out->minx = sat_usize(std::round(mnx / xyreso));
  out->miny = sat_usize(std::round(mny / xyreso));
  out->maxx = sat_usize(std::round(mxx / xyreso));
  out->maxy = sat_usize(std::round(mxy / xyreso));
  out->xw = out->maxx - out->minx;
  out->yw = out->maxy - out->miny;
  out->minyaw = sat_isize(std::round(-kPi / yawreso)) - 1;
  out->xyreso = xyreso;
  out->yawreso = yawreso;
  return PP_OK;
}

namespace {

struct Footprint {
  double cp2center, bound_len, bound_wid, bound_radius;
};
Footprint footprint(const pp_config& conf) {  // hybrid_astar.rs:418-421
  Footprint f;
  f.cp2center = (conf.car.center2front - conf.car.center2back) / 2.0;
  f.bound_len = (conf.car.center2front + conf.car.center2back) / 2.0 + conf.extend_bound;
  f.bound_wid = conf.car.width / 2.0 + conf.extend_bound;
  f.bound_radius = std::hypot(f.bound_len, f.bound_wid);
  return f;
}

// util.rs:1-5
inline void rot2d(double p0, double p1, double th, double& x, double& y) {
  x = p0 * std::cos(th) - p1 * std::sin(th);
  y = p0 * std::sin(th) + p1 * std::cos(th);
}

// One obstacle against one pose: the kdtree closure (:428-430) with `<= radius`, then :437-443.
inline bool obstacle_hits(double obx, double oby, double cx, double cy, double iyaw,
                          const Footprint& f) {
  double dist = std::sqrt((cx - obx) * (cx - obx) + (cy - oby) * (cy - oby));
  if (!(dist <= f.bound_radius)) return false;
  double xo = obx - cx;
  double yo = oby - cy;
  double dx, dy;
  rot2d(xo, yo, -iyaw, dx, dy);
  return (std::fabs(dx) < f.bound_len) && (std::fabs(dy) < f.bound_wid);
}

// hybrid_astar.rs:422-445 for one pose.  status: PP_ERR_NONFINITE when kdtree.within would Err.
inline bool pose_collides(const orc_map& m, const Footprint& f, double ix, double iy, double iyaw,
                          int mode, int32_t* status) {
  double cx = ix + f.cp2center * std::cos(iyaw);
  double cy = iy + f.cp2center * std::sin(iyaw);
  if (!std::isfinite(cx) || !std::isfinite(cy)) {
    if (status) *status = PP_ERR_NONFINITE;
    return false;
  }
  if (mode == 2) {  // hybrid_astar.rs:426-445 as written: radius query, then the rotated-box test per id
    const std::vector<KdTree2::Hit> ids = m.kd.within(cx, cy, f.bound_radius);
    for (const KdTree2::Hit& h : ids) {
      const double xo = m.ox[h.data] - cx, yo = m.oy[h.data] - cy;
      double dx, dy;
      rot2d(xo, yo, -iyaw, dx, dy);  // cos and sin of -yaw once per obstacle, as util.rs:1-5 does
      if (std::fabs(dx) < f.bound_len && std::fabs(dy) < f.bound_wid) return true;
    }
    return false;
  }
  if (mode == 0 || m.nbx == 0) {
    for (size_t i = 0; i < m.ox.size(); ++i)
      if (obstacle_hits(m.ox[i], m.oy[i], cx, cy, iyaw, f)) return true;
    return false;
  }
  // conservative bucket window: every point with dist <= R has |x-cx| <= R (+1 bucket of slack)
  double R = f.bound_radius;
  int64_t bx_lo = (int64_t)std::floor(cx - R) - 1 - m.bx0, bx_hi = (int64_t)std::floor(cx + R) + 1 - m.bx0;
  int64_t by_lo = (int64_t)std::floor(cy - R) - 1 - m.by0, by_hi = (int64_t)std::floor(cy + R) + 1 - m.by0;
  bx_lo = std::max<int64_t>(bx_lo, 0);
  by_lo = std::max<int64_t>(by_lo, 0);
  bx_hi = std::min<int64_t>(bx_hi, m.nbx - 1);
  by_hi = std::min<int64_t>(by_hi, m.nby - 1);
  for (int64_t by = by_lo; by <= by_hi; ++by) {
    size_t row = (size_t)by * (size_t)m.nbx;
    for (int64_t bx = bx_lo; bx <= bx_hi; ++bx) {
      for (uint32_t k = m.start[row + bx]; k < m.start[row + bx + 1]; ++k) {
        uint32_t i = m.items[k];
        if (obstacle_hits(m.ox[i], m.oy[i], cx, cy, iyaw, f)) return true;
      }
    }
  }
  return false;
}

}  // namespace

int32_t orc_collision_flags(const orc_map* m, const pp_config* conf, uint64_t n, const double* x,
                            const double* y, const double* yaw, uint8_t* flags, int32_t mode) {
  if (!m || !conf || (n && (!x || !y || !yaw || !flags))) return PP_ERR_INVALID;
  if (!m->finite) return PP_ERR_NONFINITE;
  Footprint f = footprint(*conf);
  int32_t st = PP_OK;
  for (uint64_t i = 0; i < n; ++i) flags[i] = pose_collides(*m, f, x[i], y[i], yaw[i], mode, &st) ? 1 : 0;
  return st;
}

int32_t orc_collision_flags_mt_mode(const orc_map* m, const pp_config* conf, uint64_t n, const double* x,
                                    const double* y, const double* yaw, uint8_t* flags, int32_t threads,
                                    int32_t mode) {
  if (!m || !conf || (n && (!x || !y || !yaw || !flags))) return PP_ERR_INVALID;
  if (!m->finite) return PP_ERR_NONFINITE;
  Footprint f = footprint(*conf);
  int32_t st = PP_OK;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    int32_t s = PP_OK;
    flags[i] = pose_collides(*m, f, x[i], y[i], yaw[i], mode, &s) ? 1 : 0;
    if (s != PP_OK) st = s;
  }
  return st;
}

int32_t orc_collision_flags_mt(const orc_map* m, const pp_config* conf, uint64_t n, const double* x,
                               const double* y, const double* yaw, uint8_t* flags, int32_t threads) {
  if (!m || !conf || (n && (!x || !y || !yaw || !flags))) return PP_ERR_INVALID;
  if (!m->finite) return PP_ERR_NONFINITE;
  Footprint f = footprint(*conf);
  int32_t st = PP_OK;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    int32_t s = PP_OK;
    flags[i] = pose_collides(*m, f, x[i], y[i], yaw[i], 1, &s) ? 1 : 0;
    if (s != PP_OK) st = s;
  }
  return st;
}

// hybrid_astar.rs:495-514
int32_t orc_motion_set(const pp_config* conf, double* steer, double* direc, uint64_t cap,
                       uint64_t* n_out) {
  if (!conf || !n_out) return PP_ERR_INVALID;
  std::vector<double> s;
  for (uint64_t i = 1; i < conf->n_steer; ++i)
    s.push_back((double)i * conf->max_steer / (double)conf->n_steer);
  std::vector<double> st(s);
  st.push_back(0.0);
  for (double v : s) st.push_back(-v);
  size_t half = st.size();
  *n_out = 2 * half;
  if (cap < 2 * half) return PP_ERR_CAPACITY;
  for (size_t i = 0; i < half; ++i) {
    steer[i] = st[i];
    steer[half + i] = st[i];
    direc[i] = 1.0;
    direc[half + i] = -1.0;
  }
  return PP_OK;
}

namespace {

struct Node {  // data_struct.rs:2-14
  uint64_t xind, yind;
  int64_t yawind, direction;
  std::vector<double> x, y, yaw;
  std::vector<int64_t> directions;
  double steer, cost;
  int64_t pind;
};

// hybrid_astar.rs:252-320 (+ is_index_ok :322-343).  Returns false for None.
bool calc_next_node(const orc_map& m, const pp_param& P, const pp_config& conf, const Footprint& fp,
                    double curr_x, double curr_y, double curr_yaw, int64_t curr_dir,
                    double curr_steer, double curr_cost, uint64_t c_id, double u, double d,
                    Node* node, uint64_t* poses_checked, int32_t* status) {
  double step = conf.xy_reso * 2.0;
  size_t nlist = (size_t)sat_usize(std::ceil(step / conf.move_step));
  std::vector<double> xl(nlist + 1, curr_x), yl(nlist + 1, curr_y), yawl(nlist + 1, curr_yaw);
  for (size_t i = 0; i < nlist; ++i) {
    xl[i + 1] = xl[i] + d * conf.move_step * std::cos(yawl[i]);
    yl[i + 1] = yl[i] + d * conf.move_step * std::sin(yawl[i]);
    yawl[i + 1] = yawl[i] + d * conf.move_step / conf.car.wheel_base * std::tan(u);
  }
  uint64_t xind = sat_usize(std::round(xl.back() / P.xyreso));
  uint64_t yind = sat_usize(std::round(yl.back() / P.xyreso));
  int64_t yawind = sat_isize(std::round(yawl.back() / P.yawreso));

  // is_index_ok
  if (xind <= P.minx || xind >= P.maxx || yind <= P.miny || yind >= P.maxy) return false;
  size_t stepc = (size_t)conf.collision_check_step;
  // poses_checked counts the poses handed to is_collision (the length of the picked lists),
  // not the ones evaluated before its early return
  if (poses_checked) *poses_checked += (xl.size() + stepc - 1) / stepc;
  for (size_t k = 0; k < xl.size(); k += stepc)
    if (pose_collides(m, fp, xl[k], yl[k], yawl[k], 1, status)) return false;

  double cost = 0.0;
  int64_t direction;
  if (d > 0.0) {
    direction = 1;
    cost += std::fabs(step);
  } else {
    direction = -1;
    cost += std::fabs(step) * conf.backward_cost;
  }
  if (direction != curr_dir) cost += conf.gear_cost;
  cost += conf.steer_angle_cost * std::fabs(u);
  cost += conf.steer_change_cost * std::fabs(curr_steer - u);
  cost += curr_cost;

  node->xind = xind;
  node->yind = yind;
  node->yawind = yawind;
  node->direction = direction;
  node->x = std::move(xl);
  node->y = std::move(yl);
  node->yaw = std::move(yawl);
  node->directions.assign(node->x.size(), direction);
  node->steer = u;
  node->cost = cost;
  node->pind = (int64_t)c_id;
  return true;
}

}  // namespace

int32_t orc_rollout_points(const pp_config* conf, double x0, double y0, double yaw0, double u,
                           double d, double* x, double* y, double* yaw, uint64_t cap,
                           uint64_t* n_out) {
  if (!conf || !n_out) return PP_ERR_INVALID;
  double step = conf->xy_reso * 2.0;
  size_t nlist = (size_t)sat_usize(std::ceil(step / conf->move_step));
  *n_out = nlist + 1;
  if (cap < nlist + 1) return PP_ERR_CAPACITY;
  x[0] = x0;
  y[0] = y0;
  yaw[0] = yaw0;
  for (size_t i = 0; i < nlist; ++i) {
    x[i + 1] = x[i] + d * conf->move_step * std::cos(yaw[i]);
    y[i + 1] = y[i] + d * conf->move_step * std::sin(yaw[i]);
    yaw[i + 1] = yaw[i] + d * conf->move_step / conf->car.wheel_base * std::tan(u);
  }
  return PP_OK;
}

int32_t orc_expand_batch(const orc_map* m, const pp_config* conf, uint64_t n, const double* px,
                         const double* py, const double* pyaw, const int64_t* pdir,
                         const double* psteer, const double* pcost, pp_expand_out* out) {
  if (!m || !conf || !out) return PP_ERR_INVALID;
  if (!m->finite) return PP_ERR_NONFINITE;
  pp_param P;
  orc_map_param(m, conf, &P);
  Footprint fp = footprint(*conf);
  uint64_t np = 0;
  orc_motion_set(conf, nullptr, nullptr, 0, &np);
  std::vector<double> steer(np), direc(np);
  orc_motion_set(conf, steer.data(), direc.data(), np, &np);
  int32_t st = PP_OK;
  for (uint64_t p = 0; p < n; ++p) {
    for (uint64_t i = 0; i < np; ++i) {
      Node nd;
      bool ok = calc_next_node(*m, P, *conf, fp, px[p], py[p], pyaw[p], pdir[p], psteer[p],
                               pcost[p], 0, steer[i], direc[i], &nd, nullptr, &st);
      uint64_t o = p * np + i;
      out->valid[o] = ok ? 1 : 0;
      if (ok) {
        out->xind[o] = nd.xind;
        out->yind[o] = nd.yind;
        out->yawind[o] = nd.yawind;
        out->cost[o] = nd.cost;
        out->end_x[o] = nd.x.back();
        out->end_y[o] = nd.y.back();
        out->end_yaw[o] = nd.yaw.back();
      } else {
        out->xind[o] = out->yind[o] = 0;
        out->yawind[o] = 0;
        out->cost[o] = 0.0;
        out->end_x[o] = out->end_y[o] = out->end_yaw[o] = 0.0;
      }
    }
  }
  return st;
}

// ---------------------------------------------------------------------- astar.rs
namespace {

struct GridDims {
  uint64_t minx, miny, maxx, maxy, xw, yw;
};

// astar.rs:232-242 on the scaled obstacle lists
GridDims grid_dims(const std::vector<double>& oxf, const std::vector<double>& oyf) {
  double mnx = kInf, mny = kInf, mxx = -kInf, mxy = -kInf;
  for (size_t i = 0; i < oxf.size(); ++i) {
    mnx = std::fmin(mnx, oxf[i]);
    mxx = std::fmax(mxx, oxf[i]);
    mny = std::fmin(mny, oyf[i]);
    mxy = std::fmax(mxy, oyf[i]);
  }
  GridDims g;
  g.minx = sat_usize(std::round(mnx));
  g.miny = sat_usize(std::round(mny));
  g.maxx = sat_usize(std::round(mxx));
  g.maxy = sat_usize(std::round(mxy));
  g.xw = g.maxx - g.minx;
  g.yw = g.maxy - g.miny;
  return g;
}

// astar.rs:244-261; obsmap[x*yw + y]
void calc_obsmap(const std::vector<double>& oxf, const std::vector<double>& oyf, double rr,
                 double reso, const GridDims& g, std::vector<uint8_t>& obs, int mode) {
  obs.assign((size_t)g.xw * (size_t)g.yw, 0);
  double thr = rr / reso;
  if (mode == 0) {
    for (uint64_t x = 0; x < g.xw; ++x) {
      uint64_t xx = x + g.minx;
      for (uint64_t y = 0; y < g.yw; ++y) {
        uint64_t yy = y + g.miny;
        for (size_t i = 0; i < oxf.size(); ++i) {
          if (std::hypot(oxf[i] - (double)xx, oyf[i] - (double)yy) <= thr) {
            obs[x * g.yw + y] = 1;
            break;
          }
        }
      }
    }
    return;
  }
  // scatter: cells farther than thr + 1 from the obstacle in x or y cannot satisfy the predicate
  for (size_t i = 0; i < oxf.size(); ++i) {
    double lo_x = std::floor(oxf[i] - thr) - 1, hi_x = std::ceil(oxf[i] + thr) + 1;
    double lo_y = std::floor(oyf[i] - thr) - 1, hi_y = std::ceil(oyf[i] + thr) + 1;
    int64_t x0 = std::max<int64_t>((int64_t)lo_x, (int64_t)g.minx);
    int64_t x1 = std::min<int64_t>((int64_t)hi_x, (int64_t)g.maxx - 1);
    int64_t y0 = std::max<int64_t>((int64_t)lo_y, (int64_t)g.miny);
    int64_t y1 = std::min<int64_t>((int64_t)hi_y, (int64_t)g.maxy - 1);
    for (int64_t xx = x0; xx <= x1; ++xx)
      for (int64_t yy = y0; yy <= y1; ++yy)
        if (std::hypot(oxf[i] - (double)xx, oyf[i] - (double)yy) <= thr)
          obs[(size_t)(xx - (int64_t)g.minx) * g.yw + (size_t)(yy - (int64_t)g.miny)] = 1;
  }
}

struct HeapEnt {
  double cost;
  uint64_t ind;
};
struct HeapLess {  // max-heap on (Reverse(cost), ind): top = min cost, ties -> largest ind
  bool operator()(const HeapEnt& a, const HeapEnt& b) const {
    if (a.cost != b.cost) return a.cost > b.cost;
    return a.ind < b.ind;
  }
};

// astar.rs:153-198 with dense arrays in place of the two HashMaps (keys are unique cell
// indices, so the maps are only associative storage).  (gx, gy) absolute cell coordinates.
int32_t dijkstra(const GridDims& g, const std::vector<uint8_t>& obs, uint64_t gx, uint64_t gy,
                 double* hmap) {
  size_t ncell = (size_t)g.xw * (size_t)g.yw;
  for (size_t i = 0; i < ncell; ++i) hmap[i] = kInf;
  // the goal's own hmap write (astar.rs:195) indexes hmap[gx-minx][gy-miny]: out of range panics
  if (gx < g.minx || gx - g.minx >= g.xw || gy < g.miny || gy - g.miny >= g.yw) return PP_ERR_BOUNDS;
  static const int mdx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};  // astar.rs:297-308
  static const int mdy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  double ucost[8];
  for (int k = 0; k < 8; ++k) ucost[k] = std::hypot((double)mdx[k], (double)mdy[k]);  // :213-215
  std::vector<uint8_t> state(ncell, 0);  // 0 unseen, 1 open, 2 closed
  std::vector<double> cost(ncell, 0.0);
  std::priority_queue<HeapEnt, std::vector<HeapEnt>, HeapLess> pq;
  auto key = [&](uint64_t x, uint64_t y) { return (y - g.miny) * g.xw + (x - g.minx); };  // :228-230
  uint64_t gk = key(gx, gy);
  state[gk] = 1;
  cost[gk] = 0.0;
  pq.push({0.0, gk});
  size_t n_open = 1;
  while (n_open > 0) {
    HeapEnt e = pq.top();
    pq.pop();
    uint64_t ind = e.ind;
    state[ind] = 2;
    --n_open;
    uint64_t cx = ind % g.xw + g.minx, cy = ind / g.xw + g.miny;
    double ccost = cost[ind];
    for (int k = 0; k < 8; ++k) {
      uint64_t nx = (uint64_t)((int64_t)cx + mdx[k]);
      uint64_t ny = (uint64_t)((int64_t)cy + mdy[k]);
      double ncost = ccost + ucost[k];
      // check_node :200-211
      if (nx <= g.minx || nx >= g.maxx || ny <= g.miny || ny >= g.maxy) continue;
      if (obs[(nx - g.minx) * g.yw + (ny - g.miny)]) continue;
      uint64_t nk = key(nx, ny);
      if (state[nk] == 2) continue;
      if (state[nk] == 1) {
        if (cost[nk] > ncost) cost[nk] = ncost;
      } else {
        state[nk] = 1;
        cost[nk] = ncost;
        ++n_open;
        pq.push({ncost, nk});
      }
    }
  }
  for (size_t k = 0; k < ncell; ++k) {
    if (state[k] == 2) {
      uint64_t x = k % g.xw, y = k / g.xw;
      hmap[x * g.yw + y] = cost[k];
    }
  }
  return PP_OK;
}

}  // namespace

int32_t orc_obsmap(const double* ox, const double* oy, uint64_t n, double reso, double rr,
                   uint8_t* obsmap, uint64_t cap, uint64_t dims[6], int32_t mode) {
  if (n == 0) return PP_ERR_EMPTY_MAP;
  std::vector<double> oxf(n), oyf(n);
  for (uint64_t i = 0; i < n; ++i) {
    oxf[i] = ox[i] / reso;
    oyf[i] = oy[i] / reso;
  }
  GridDims g = grid_dims(oxf, oyf);
  uint64_t d[6] = {g.minx, g.miny, g.maxx, g.maxy, g.xw, g.yw};
  if (dims) std::memcpy(dims, d, sizeof(d));
  if (cap < g.xw * g.yw) return PP_ERR_CAPACITY;
  std::vector<uint8_t> obs;
  calc_obsmap(oxf, oyf, rr, reso, g, obs, mode);
  std::memcpy(obsmap, obs.data(), obs.size());
  return PP_OK;
}

int32_t orc_heuristic(const double* ox, const double* oy, uint64_t n, double gx, double gy,
                      double reso, double rr, double* hmap, uint64_t cap, uint64_t dims[6]) {
  if (n == 0) return PP_ERR_EMPTY_MAP;
  uint64_t gxi = sat_usize(std::round(gx / reso));  // astar.rs:141-146
  uint64_t gyi = sat_usize(std::round(gy / reso));
  std::vector<double> oxf(n), oyf(n);
  for (uint64_t i = 0; i < n; ++i) {
    oxf[i] = ox[i] / reso;
    oyf[i] = oy[i] / reso;
  }
  GridDims g = grid_dims(oxf, oyf);
  uint64_t d[6] = {g.minx, g.miny, g.maxx, g.maxy, g.xw, g.yw};
  if (dims) std::memcpy(dims, d, sizeof(d));
  if (cap < g.xw * g.yw) return PP_ERR_CAPACITY;
  std::vector<uint8_t> obs;
  calc_obsmap(oxf, oyf, rr, reso, g, obs, 1);
  return dijkstra(g, obs, gxi, gyi, hmap);
}

int32_t orc_wavefront(uint64_t xw, uint64_t yw, const uint8_t* obs, uint64_t goal_x,
                      uint64_t goal_y, double* cost) {
  GridDims g{0, 0, xw, yw, xw, yw};
  std::vector<uint8_t> o(obs, obs + (size_t)xw * (size_t)yw);
  return dijkstra(g, o, goal_x, goal_y, cost);
}

// ------------------------------------------------------------------ plain grid A* (astar.rs:53-132)
namespace {
struct GridNode {  // HolonomicNode, astar.rs:9-15
  uint64_t x, y;
  double cost;
  int64_t pind;
};
}  // namespace

int32_t orc_astar(const double* ox, const double* oy, uint64_t n, double sx, double sy, double gx,
                  double gy, double reso, double rr, double* path_x, double* path_y, uint64_t cap,
                  uint64_t* len, uint64_t* closed_order, uint64_t closed_cap, uint64_t* expansions) {
  if (n == 0) return PP_ERR_EMPTY_MAP;
  *len = 0;
  const GridNode n_start{sat_usize(std::round(sx / reso)), sat_usize(std::round(sy / reso)), 0.0, -1};
  const GridNode n_goal{sat_usize(std::round(gx / reso)), sat_usize(std::round(gy / reso)), 0.0, -1};
  std::vector<double> oxf(n), oyf(n);
  for (uint64_t i = 0; i < n; ++i) {  // :77-78
    oxf[i] = ox[i] / reso;
    oyf[i] = oy[i] / reso;
  }
  const GridDims g = grid_dims(oxf, oyf);  // calc_parameters :232-242
  std::vector<uint8_t> obs;
  calc_obsmap(oxf, oyf, rr, reso, g, obs, 1);
  auto calc_index = [&](const GridNode& nd) { return (nd.y - g.miny) * g.xw + (nd.x - g.minx); };  // :228-230
  auto fvalue = [&](const GridNode& nd) {  // :217-226
    return nd.cost + std::hypot((double)nd.x - (double)n_goal.x, (double)nd.y - (double)n_goal.y);
  };
  static const int mdx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};  // :297-308
  static const int mdy[8] = {0, 1, 1, 1, 0, -1, -1, -1};

  const uint64_t goal_ind = calc_index(n_goal);
  std::unordered_map<uint64_t, GridNode> open_set, closed_set;
  open_set.emplace(calc_index(n_start), n_start);
  // (Reverse(f), ind) in a max-heap: lowest f first, ties to the larger ind (HeapLess)
  std::priority_queue<HeapEnt, std::vector<HeapEnt>, HeapLess> q;
  q.push({fvalue(n_start), calc_index(n_start)});
  uint64_t n_closed = 0;
  bool found = false;
  while (!open_set.empty()) {  // :93
    const uint64_t ind = q.top().ind;
    q.pop();
    const GridNode cur = open_set.at(ind);
    open_set.erase(ind);
    closed_set[ind] = cur;
    if (closed_order && n_closed < closed_cap) closed_order[n_closed] = ind;
    ++n_closed;
    if (goal_ind == ind) {
      found = true;
      break;
    }
    for (int k = 0; k < 8; ++k) {
      GridNode nd{(uint64_t)((int64_t)cur.x + mdx[k]), (uint64_t)((int64_t)cur.y + mdy[k]),
                  cur.cost + std::hypot((double)mdx[k], (double)mdy[k]), (int64_t)ind};
      // check_node :200-211
      if (nd.x <= g.minx || nd.x >= g.maxx || nd.y <= g.miny || nd.y >= g.maxy) continue;
      if (obs[(nd.x - g.minx) * g.yw + (nd.y - g.miny)]) continue;
      const uint64_t n_ind = calc_index(nd);
      if (closed_set.count(n_ind)) continue;
      auto it = open_set.find(n_ind);
      if (it != open_set.end()) {
        if (it->second.cost > nd.cost) {
          it->second.cost = nd.cost;
          it->second.pind = (int64_t)ind;
        }
      } else {
        open_set.emplace(n_ind, nd);
        q.push({fvalue(nd), n_ind});
      }
    }
  }
  if (expansions) *expansions = n_closed;
  if (!found) return PP_NO_PATH;
  // extract_path :263-295
  std::vector<double> px{(double)n_goal.x}, py{(double)n_goal.y};
  uint64_t n_ind = goal_ind;
  for (;;) {
    auto it = closed_set.find(n_ind);
    if (it == closed_set.end()) break;
    const GridNode& nd = it->second;
    px.push_back((double)nd.x);
    py.push_back((double)nd.y);
    n_ind = (uint64_t)nd.pind;
    if (nd.x == n_start.x && nd.y == n_start.y && nd.cost == n_start.cost && nd.pind == n_start.pind) break;
  }
  std::reverse(px.begin(), px.end());
  std::reverse(py.begin(), py.end());
  *len = px.size();
  if (px.size() > cap) return PP_ERR_CAPACITY;
  for (size_t i = 0; i < px.size(); ++i) {
    path_x[i] = px[i] * reso;
    path_y[i] = py[i] * reso;
  }
  return PP_OK;
}

// ------------------------------------------------------------------ Reeds-Shepp ABI
int32_t orc_rs_words(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                     double maxc, int32_t* n_paths, pp_rs_word* words) {
  std::vector<FullPath> paths;
  if (!calc_all_paths(sx, sy, syaw, gx, gy, gyaw, maxc, 0.0, paths, false)) return PP_ERR_DEGENERATE_RS;
  *n_paths = (int32_t)paths.size();
  for (size_t k = 0; k < paths.size(); ++k) to_abi(paths[k].w, &words[k]);
  return PP_OK;
}

int32_t orc_rs_all_paths(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                         double maxc, double step_size, int32_t* n_paths, pp_rs_word* words,
                         uint64_t* traj_off, double* traj_x, double* traj_y, double* traj_yaw,
                         int64_t* traj_dir, uint64_t traj_cap) {
  std::vector<FullPath> paths;
  if (!calc_all_paths(sx, sy, syaw, gx, gy, gyaw, maxc, step_size, paths, true))
    return PP_ERR_DEGENERATE_RS;
  *n_paths = (int32_t)paths.size();
  uint64_t off = 0;
  for (size_t k = 0; k < paths.size(); ++k) {
    to_abi(paths[k].w, &words[k]);
    traj_off[k] = off;
    off += paths[k].x.size();
  }
  traj_off[paths.size()] = off;
  if (off > traj_cap) return PP_ERR_CAPACITY;
  for (size_t k = 0; k < paths.size(); ++k) {
    uint64_t o = traj_off[k];
    for (size_t i = 0; i < paths[k].x.size(); ++i) {
      traj_x[o + i] = paths[k].x[i];
      traj_y[o + i] = paths[k].y[i];
      traj_yaw[o + i] = paths[k].yaw[i];
      traj_dir[o + i] = paths[k].dir[i];
    }
  }
  return PP_OK;
}

int32_t orc_rs_batch(uint64_t n, const double* sx, const double* sy, const double* syaw,
                     const double* gx, const double* gy, const double* gyaw, double maxc,
                     int32_t* n_paths, int32_t* best_idx, pp_rs_word* best, uint8_t* flags,
                     int32_t threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    std::vector<FullPath> paths;
    bool ok = calc_all_paths(sx[i], sy[i], syaw[i], gx[i], gy[i], gyaw[i], maxc, 0.0, paths, false);
    flags[i] = ok ? 0 : 1;
    n_paths[i] = ok ? (int32_t)paths.size() : 0;
    best_idx[i] = -1;
    std::memset(&best[i], 0, sizeof(pp_rs_word));
    if (!ok || paths.empty()) continue;
    double min_l = paths[0].w.total;  // reeds_shepp.rs:72-82
    size_t mini = 0;
    for (size_t k = 0; k < paths.size(); ++k) {
      if (paths[k].w.total <= min_l) {
        min_l = paths[k].w.total;
        mini = k;
      }
    }
    best_idx[i] = (int32_t)mini;
    to_abi(paths[mini].w, &best[i]);
  }
  return PP_OK;
}

int32_t orc_rs_words_batch(uint64_t n, const double* sx, const double* sy, const double* syaw,
                           const double* gx, const double* gy, const double* gyaw, double maxc,
                           int32_t* n_paths, pp_rs_word* words, int32_t threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    std::vector<FullPath> paths;
    pp_rs_word* out = words + (size_t)i * PP_RS_MAX_PATHS;
    std::memset(out, 0, sizeof(pp_rs_word) * PP_RS_MAX_PATHS);
    if (!calc_all_paths(sx[i], sy[i], syaw[i], gx[i], gy[i], gyaw[i], maxc, 0.0, paths, false)) {
      n_paths[i] = -1;  // the :157 assert
      continue;
    }
    n_paths[i] = (int32_t)paths.size();
    for (size_t k = 0; k < paths.size(); ++k) to_abi(paths[k].w, &out[k]);
  }
  return PP_OK;
}

double orc_rs_path_cost(const pp_rs_word* w, const pp_config* conf) {
  return rs_path_cost(w->lengths, w->ctypes, w->nseg, *conf);
}

namespace {

// hybrid_astar.rs:380-415.  Returns index of the accepted path or -1.
int analytic_expansion(const orc_map& m, const pp_config& conf, const Footprint& fp, double sx,
                       double sy, double syaw, double gx, double gy, double gyaw,
                       std::vector<FullPath>& paths, double* rs_cost, uint64_t* poses_checked,
                       int32_t* status) {
  double maxc = std::tan(conf.max_steer) / conf.car.wheel_base;
  if (!calc_all_paths(sx, sy, syaw, gx, gy, gyaw, maxc, conf.move_step, paths, true)) {
    *status = PP_ERR_DEGENERATE_RS;
    return -1;
  }
  std::priority_queue<HeapEnt, std::vector<HeapEnt>, HeapLess> pq;
  for (size_t i = 0; i < paths.size(); ++i) {
    double c = rs_path_cost(paths[i].w.len, (const uint8_t*)paths[i].w.ct, paths[i].w.n, conf);
    if (c != c) {
      *status = PP_ERR_NAN_COST;
      return -1;
    }
    pq.push({c, (uint64_t)i});
  }
  size_t stepc = (size_t)conf.collision_check_step;
  while (!pq.empty()) {
    HeapEnt e = pq.top();
    pq.pop();
    const FullPath& p = paths[e.ind];
    bool hit = false;
    if (poses_checked) *poses_checked += (p.x.size() + stepc - 1) / stepc;
    for (size_t k = 0; k < p.x.size(); k += stepc) {
      if (pose_collides(m, fp, p.x[k], p.y[k], p.yaw[k], 1, status)) {
        hit = true;
        break;
      }
    }
    if (!hit) {
      if (rs_cost) *rs_cost = e.cost;
      return (int)e.ind;
    }
  }
  return -1;
}

}  // namespace

int32_t orc_analytic_batch(const orc_map* m, const pp_config* conf, uint64_t n, const double* sx,
                           const double* sy, const double* syaw, const double* gx,
                           const double* gy, const double* gyaw, uint8_t* found,
                           int32_t* path_idx, int32_t* n_paths, pp_rs_word* word,
                           double* rs_cost) {
  if (!m || !conf) return PP_ERR_INVALID;
  Footprint fp = footprint(*conf);
  for (uint64_t i = 0; i < n; ++i) {
    std::vector<FullPath> paths;
    int32_t st = PP_OK;
    double c = 0.0;
    int k = analytic_expansion(*m, *conf, fp, sx[i], sy[i], syaw[i], gx[i], gy[i], gyaw[i], paths,
                               &c, nullptr, &st);
    found[i] = (k >= 0 ? 1 : 0) | (st == PP_ERR_DEGENERATE_RS ? 4 : 0);
    path_idx[i] = k;
    n_paths[i] = (int32_t)paths.size();
    std::memset(&word[i], 0, sizeof(pp_rs_word));
    rs_cost[i] = 0.0;
    if (k >= 0) {
      to_abi(paths[k].w, &word[i]);
      rs_cost[i] = c;
    }
  }
  return PP_OK;
}

// ------------------------------------------------------------------ the planner
namespace {

// hybrid_astar.rs:516-520, release-mode wrapping arithmetic
inline uint64_t calc_index(const Node& n, const pp_param& P) {
  return (uint64_t)(n.yawind - P.minyaw) * P.xw * P.yw + (n.yind - P.miny) * P.xw + (n.xind - P.minx);
}

}  // namespace

static uint64_t g_max_expansions = 0;
void orc_set_max_expansions(uint64_t n) { g_max_expansions = n; }

int32_t orc_plan(const orc_map* m, const pp_config* conf_, const double start[3],
                 const double goal[3], pp_path* out, pp_plan_stats* stats) {
  if (!m || !conf_ || !start || !goal || !out) return PP_ERR_INVALID;
  if (m->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!m->finite) return PP_ERR_NONFINITE;
  const pp_config& conf = *conf_;
  double sx = start[0], sy = start[1], syaw = start[2];
  double gx = goal[0], gy = goal[1], gyaw = goal[2];
  double xyreso = conf.xy_reso, yawreso = conf.yaw_reso;
  Node nstart, ngoal;
  nstart.xind = sat_usize(std::round(sx / xyreso));
  nstart.yind = sat_usize(std::round(sy / xyreso));
  nstart.yawind = sat_isize(std::round(pi_2_pi(syaw) / yawreso));
  nstart.direction = 1;
  nstart.x = {sx};
  nstart.y = {sy};
  nstart.yaw = {syaw};
  nstart.directions = {1};
  nstart.steer = 0.0;
  nstart.cost = 0.0;
  nstart.pind = -1;
  ngoal = nstart;
  ngoal.xind = sat_usize(std::round(gx / xyreso));
  ngoal.yind = sat_usize(std::round(gy / xyreso));
  ngoal.yawind = sat_isize(std::round(pi_2_pi(gyaw) / yawreso));
  ngoal.x = {gx};
  ngoal.y = {gy};
  ngoal.yaw = {gyaw};

  pp_param P;
  orc_map_param(m, &conf, &P);
  Footprint fp = footprint(conf);

  // heuristic (hybrid_astar.rs:142-143): reso = xyreso, rr = 1.0
  std::vector<double> hmap((size_t)P.xw * (size_t)P.yw);
  uint64_t dims[6];
  int32_t hst = orc_heuristic(m->ox.data(), m->oy.data(), m->ox.size(), gx, gy, P.xyreso, 1.0,
                              hmap.data(), hmap.size(), dims);
  if (hst != PP_OK) return hst;
  uint64_t hxw = dims[4], hyw = dims[5];

  uint64_t np = 0;
  orc_motion_set(&conf, nullptr, nullptr, 0, &np);
  std::vector<double> steer_set(np), direc_set(np);
  orc_motion_set(&conf, steer_set.data(), direc_set.data(), np, &np);

  int32_t status = PP_OK;
  auto hybrid_cost = [&](const Node& n, double* c) -> bool {  // :491-493
    uint64_t hx = n.xind - P.minx, hy = n.yind - P.miny;
    if (hx >= hxw || hy >= hyw) {
      status = PP_ERR_BOUNDS;
      return false;
    }
    *c = n.cost + conf.h_cost * hmap[hx * hyw + hy];
    if (*c != *c) {
      status = PP_ERR_NAN_COST;
      return false;
    }
    return true;
  };

  std::unordered_map<uint64_t, Node> open_set, closed_set;
  std::priority_queue<HeapEnt, std::vector<HeapEnt>, HeapLess> pq;
  uint64_t st_idx = calc_index(nstart, P);
  open_set.emplace(st_idx, nstart);
  double c0;
  if (!hybrid_cost(nstart, &c0)) return status;
  pq.push({c0, st_idx});

  pp_plan_stats local{};
  pp_plan_stats* S = stats ? stats : &local;
  uint64_t closed_cap = S->closed_cap;
  uint64_t* closed_order = S->closed_order;
  S->expansions = S->stale_pops = S->rollouts = S->poses_checked = S->rs_paths = S->ambiguous = 0;
  S->rs_final_length = 0.0;
  S->first_ambiguous = UINT64_MAX;  // the oracle takes no guard-band decisions

  while (!open_set.empty()) {
    if (g_max_expansions && S->expansions >= g_max_expansions) return PP_ERR_LIMIT;
    HeapEnt e = pq.top();
    pq.pop();
    uint64_t ind = e.ind;
    auto it = open_set.find(ind);
    if (it == open_set.end()) {
      ++S->stale_pops;
      continue;
    }
    Node n_curr = std::move(it->second);
    open_set.erase(it);
    closed_set[ind] = n_curr;
    if (closed_order && S->expansions < closed_cap) closed_order[S->expansions] = ind;
    ++S->expansions;

    // update_node_with_analystic_expansion :345-378
    {
      std::vector<FullPath> paths;
      double rs_cost = 0.0;
      int k = analytic_expansion(*m, conf, fp, n_curr.x.back(), n_curr.y.back(), n_curr.yaw.back(),
                                 ngoal.x.back(), ngoal.y.back(), ngoal.yaw.back(), paths, &rs_cost,
                                 &S->poses_checked, &status);
      if (status != PP_OK) return status;
      S->rs_paths += paths.size();
      if (k >= 0) {
        Node f;
        f.xind = n_curr.xind;
        f.yind = n_curr.yind;
        f.yawind = n_curr.yawind;
        f.direction = n_curr.direction;
        f.x = paths[k].x;
        f.y = paths[k].y;
        f.yaw = paths[k].yaw;
        f.directions = paths[k].dir;
        f.steer = 0.0;
        f.cost = n_curr.cost + rs_cost;
        f.pind = (int64_t)calc_index(n_curr, P);
        S->rs_final_length = paths[k].w.total;
        // extract_path :218-250
        std::vector<double> rx, ry, ryaw;
        std::vector<int64_t> direc;
        double cost = 0.0;
        Node node = f;
        while (node.pind >= 0) {
          for (size_t i = node.x.size(); i-- > 1;) {
            rx.push_back(node.x[i]);
            ry.push_back(node.y[i]);
            ryaw.push_back(node.yaw[i]);
            direc.push_back(node.directions[i]);
          }
          cost += node.cost;
          auto pit = closed_set.find((uint64_t)node.pind);
          if (pit == closed_set.end()) return PP_ERR_BOUNDS;
          node = pit->second;
        }
        for (size_t i = nstart.x.size(); i-- > 0;) {
          rx.push_back(nstart.x[i]);
          ry.push_back(nstart.y[i]);
          ryaw.push_back(nstart.yaw[i]);
          direc.push_back(nstart.directions[i]);
        }
        cost += nstart.cost;
        out->len = rx.size();
        out->cost = cost;
        if (out->cap < rx.size()) return PP_ERR_CAPACITY;
        size_t L = rx.size();
        for (size_t i = 0; i < L; ++i) {
          out->x[i] = rx[L - 1 - i];
          out->y[i] = ry[L - 1 - i];
          out->yaw[i] = ryaw[L - 1 - i];
          out->direction[i] = direc[L - 1 - i];
        }
        return PP_OK;
      }
    }

    for (uint64_t i = 0; i < np; ++i) {
      Node node;
      ++S->rollouts;
      if (!calc_next_node(*m, P, conf, fp, n_curr.x.back(), n_curr.y.back(), n_curr.yaw.back(),
                          n_curr.direction, n_curr.steer, n_curr.cost, ind, steer_set[i],
                          direc_set[i], &node, &S->poses_checked, &status))
        continue;
      if (status != PP_OK) return status;
      uint64_t node_ind = calc_index(node, P);
      if (closed_set.count(node_ind)) continue;
      auto oit = open_set.find(node_ind);
      if (oit == open_set.end()) {
        double c;
        if (!hybrid_cost(node, &c)) return status;
        pq.push({c, node_ind});
        open_set.emplace(node_ind, std::move(node));
      } else if (oit->second.cost > node.cost) {
        double c;
        if (!hybrid_cost(node, &c)) return status;
        pq.push({c, node_ind});
        oit->second = std::move(node);
      }
    }
    if (status != PP_OK) return status;
  }
  out->len = 0;
  out->cost = 0.0;
  return PP_NO_PATH;
}
