// san_check.cpp -- walks the CPU oracle through every entry point under AddressSanitizer and
// UndefinedBehaviorSanitizer (host code only; `make san` builds it with the sanitizer runtime linked
// statically).  Test infrastructure like the rest of this directory: it checks the checker.
// Inputs are the reference's demo (src/main.rs:46-84, :215-234) plus seeded random batches; the
// expected numbers are the demo's known results.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pp_oracle.h"

namespace {
int fails = 0;
void expect(bool ok, const char* what) {
  if (!ok) {
    std::fprintf(stderr, "FAIL: %s\n", what);
    ++fails;
  }
}
uint64_t rng_state = 0x5EED0001ull;
double uni() {  // splitmix64 -> [0, 1)
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
}  // namespace

int main() {
  const double kPi = 3.14159265358979323846;
  pp_config conf{2.0, 15.0 * kPi / 180.0, 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0, 15.0, {4.5, 1.0, 3.0, 3.5}, 0.6};
  std::vector<double> ox, oy;
  for (int i = 0; i < 51; ++i) { ox.push_back(i); oy.push_back(0.0); }
  for (int i = 0; i < 51; ++i) { ox.push_back(i); oy.push_back(30.0); }
  for (int i = 0; i < 31; ++i) { ox.push_back(0.0); oy.push_back(i); }
  for (int i = 0; i < 31; ++i) { ox.push_back(50.0); oy.push_back(i); }
  for (int i = 10; i < 21; ++i) { ox.push_back(i); oy.push_back(15.0); }
  for (int i = 0; i < 15; ++i) { ox.push_back(20.0); oy.push_back(i); }
  for (int i = 15; i < 30; ++i) { ox.push_back(30.0); oy.push_back(i); }
  for (int i = 0; i < 16; ++i) { ox.push_back(40.0); oy.push_back(i); }
  const uint64_t no = ox.size();
  orc_map* m = orc_map_create(ox.data(), oy.data(), no);
  expect(m != nullptr, "map");

  pp_param P{};
  expect(orc_map_param(m, &conf, &P) == PP_OK && P.xw == 25 && P.yw == 15 && P.minyaw == -13, "param");

  // planner on the demo query
  {
    const double start[3] = {10.0, 7.0, 120.0 * kPi / 180.0}, goal[3] = {45.0, 20.0, 90.0 * kPi / 180.0};
    std::vector<double> x(4096), y(4096), yaw(4096);
    std::vector<int64_t> dir(4096);
    std::vector<uint64_t> closed(256);
    pp_path out{x.data(), y.data(), yaw.data(), dir.data(), 4096, 0, 0.0};
    pp_plan_stats st{};
    st.closed_cap = closed.size();
    st.closed_order = closed.data();
    const int32_t rc = orc_plan(m, &conf, start, goal, &out, &st);
    expect(rc == PP_OK, "plan status");
    expect(out.len == 184, "plan points");
    expect(std::fabs(out.cost - 561.4479147109321) < 1e-9 * 561.45, "plan cost");
    expect(st.expansions == 24 && st.stale_pops == 11 && st.poses_checked == 8951, "plan counters");
    expect(closed[0] == 7980 && closed[1] == 8004 && closed[2] == 7304, "closing order");
    out.cap = 10;  // too small: the required length is reported
    expect(orc_plan(m, &conf, start, goal, &out, nullptr) == PP_ERR_CAPACITY && out.len == 184, "plan capacity");
  }
  // heuristic grid and raster
  {
    uint64_t dims[6];
    std::vector<double> h(25 * 15);
    expect(orc_heuristic(ox.data(), oy.data(), no, 45.0, 20.0, 2.0, 1.0, h.data(), h.size(), dims) == PP_OK, "heuristic");
    int finite = 0;
    double sum = 0.0, mx = 0.0;
    for (double v : h)
      if (std::isfinite(v)) { ++finite; sum += v; mx = std::fmax(mx, v); }
    expect(finite == 302 && std::fabs(sum - 4588.116374840029) < 1e-6 && std::fabs(mx - 31.384776310850242) < 1e-12, "heuristic values");
    std::vector<uint8_t> a(25 * 15), b(25 * 15);
    expect(orc_obsmap(ox.data(), oy.data(), no, 2.0, 1.0, a.data(), a.size(), dims, 0) == PP_OK, "obsmap 0");
    expect(orc_obsmap(ox.data(), oy.data(), no, 2.0, 1.0, b.data(), b.size(), dims, 1) == PP_OK && a == b, "obsmap 1");
    expect(orc_obsmap(ox.data(), oy.data(), no, 2.0, 1.0, b.data(), 3, dims, 1) == PP_ERR_CAPACITY, "obsmap capacity");
    std::vector<double> w(25 * 15);
    expect(orc_wavefront(25, 15, a.data(), 22, 10, w.data()) == PP_OK, "wavefront");
  }
  // grid A*
  {
    std::vector<double> px(512), py(512);
    std::vector<uint64_t> closed(1024);
    uint64_t len = 0, ex = 0;
    const int32_t rc = orc_astar(ox.data(), oy.data(), no, 10.0, 7.0, 45.0, 20.0, 2.0, 1.0, px.data(), py.data(), 512,
                                 &len, closed.data(), closed.size(), &ex);
    expect(rc == PP_OK && len > 2 && ex > 0, "astar");
  }
  // collision flags: bucket candidates equal the full scan
  {
    const uint64_t n = 4000;
    std::vector<double> x(n), y(n), yaw(n);
    for (uint64_t i = 0; i < n; ++i) { x[i] = uni() * 50.0; y[i] = uni() * 30.0; yaw[i] = (uni() * 2.0 - 1.0) * kPi; }
    std::vector<uint8_t> f0(n), f1(n), f2(n);
    expect(orc_collision_flags(m, &conf, n, x.data(), y.data(), yaw.data(), f0.data(), 0) == PP_OK, "flags 0");
    expect(orc_collision_flags(m, &conf, n, x.data(), y.data(), yaw.data(), f1.data(), 1) == PP_OK && f0 == f1, "flags 1");
    expect(orc_collision_flags_mt(m, &conf, n, x.data(), y.data(), yaw.data(), f2.data(), 4) == PP_OK && f0 == f2, "flags mt");
  }
  // motion set, expansion, rollout points
  {
    std::vector<double> steer(78), direc(78);
    uint64_t np = 0;
    expect(orc_motion_set(&conf, steer.data(), direc.data(), 78, &np) == PP_OK && np == 78, "motion set");
    expect(orc_motion_set(&conf, steer.data(), direc.data(), 10, &np) == PP_ERR_CAPACITY, "motion set capacity");
    const uint64_t n = 50;
    std::vector<double> px(n), py(n), pyaw(n), ps(n, 0.0), pc(n, 0.0);
    std::vector<int64_t> pd(n, 1);
    for (uint64_t i = 0; i < n; ++i) { px[i] = 3.0 + uni() * 44.0; py[i] = 3.0 + uni() * 24.0; pyaw[i] = (uni() * 2.0 - 1.0) * kPi; }
    const uint64_t it = n * 78;
    std::vector<uint8_t> valid(it);
    std::vector<uint64_t> xi(it), yi(it);
    std::vector<int64_t> wi(it);
    std::vector<double> c(it), ex(it), ey(it), ew(it);
    pp_expand_out eo{valid.data(), xi.data(), yi.data(), wi.data(), c.data(), ex.data(), ey.data(), ew.data()};
    expect(orc_expand_batch(m, &conf, n, px.data(), py.data(), pyaw.data(), pd.data(), ps.data(), pc.data(), &eo) == PP_OK, "expand");
    std::vector<double> rx(16), ry(16), rw(16);
    uint64_t nr = 0;
    expect(orc_rollout_points(&conf, 10.0, 7.0, 1.0, steer[3], 1.0, rx.data(), ry.data(), rw.data(), 16, &nr) == PP_OK && nr == 11, "rollout");
  }
  // Reeds-Shepp: demo leg 0 (src/main.rs:93-137), batches, degenerate pair
  {
    int32_t np = 0;
    std::vector<pp_rs_word> words(PP_RS_MAX_PATHS);
    expect(orc_rs_words(0.0, 0.0, 0.0, 10.0, 10.0, -90.0 * kPi / 180.0, 0.1, &np, words.data()) == PP_OK && np == 5, "rs words");
    std::vector<uint64_t> off(PP_RS_MAX_PATHS + 1);
    std::vector<double> tx(1 << 14), ty(1 << 14), tw(1 << 14);
    std::vector<int64_t> td(1 << 14);
    expect(orc_rs_all_paths(0.0, 0.0, 0.0, 10.0, 10.0, -90.0 * kPi / 180.0, 0.1, 0.2, &np, words.data(), off.data(),
                            tx.data(), ty.data(), tw.data(), td.data(), tx.size()) == PP_OK && np == 5, "rs all paths");
    expect(orc_rs_words(1.0, 2.0, 0.5, 1.0, 2.0, 0.5, 0.1, &np, words.data()) == PP_ERR_DEGENERATE_RS, "rs degenerate");
    const uint64_t n = 2000;
    std::vector<double> q[6];
    for (auto& v : q) v.resize(n);
    for (uint64_t i = 0; i < n; ++i)
      for (int k = 0; k < 6; ++k) q[k][i] = (k % 3 == 2) ? (uni() * 2.0 - 1.0) * kPi : 3.0 + uni() * (k % 3 ? 24.0 : 44.0);
    std::vector<int32_t> npaths(n), best(n), pidx(n);
    std::vector<pp_rs_word> bw(n), aw(n);
    std::vector<uint8_t> fl(n), found(n);
    std::vector<double> cost(n);
    const double maxc = std::tan(0.6) / 3.5;
    expect(orc_rs_batch(n, q[0].data(), q[1].data(), q[2].data(), q[3].data(), q[4].data(), q[5].data(), maxc,
                        npaths.data(), best.data(), bw.data(), fl.data(), 2) == PP_OK, "rs batch");
    expect(orc_rs_path_cost(&bw[0], &conf) > 0.0, "rs cost");
    expect(orc_analytic_batch(m, &conf, 300, q[0].data(), q[1].data(), q[2].data(), q[3].data(), q[4].data(),
                              q[5].data(), found.data(), pidx.data(), npaths.data(), aw.data(), cost.data()) == PP_OK,
           "analytic batch");
  }
  orc_map_destroy(m);
  if (fails) return 1;
  std::puts("oracle sanitizer walk: ok");
  return 0;
}
