// path_planner.hpp -- the reference's own function names and data types on top of the C ABI of
// pp.h, for a C++ host that wants to swap the reference's planner calls one for one.
//
// Every function keeps the reference's argument order and meaning; `Option::None` is
// std::nullopt and every condition under which the reference panics throws path_planner::Panic
// with the status code of pp.h.  The functions that take obstacle lists rebuild the device map on
// every call, as the reference rebuilds its KD-tree and rasters on every call; a caller that plans
// repeatedly against one map keeps a Session instead.  Header-only; link with libpp_hip.so.
#ifndef PATH_PLANNER_HPP_
#define PATH_PLANNER_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pp.h"

namespace path_planner {

struct Panic : std::runtime_error {  // the reference's unwrap() / assert! / index panics
  int32_t status;
  Panic(int32_t st, const char* where)
      : std::runtime_error(std::string(where) + ": " + pp_strerror(st)), status(st) {}
};

// reeds_shepp.rs:41-58
struct Car {
  double center2front, center2back, width, wheel_base;
  Car(double c2f, double c2b, double w, double wb) : center2front(c2f), center2back(c2b), width(w), wheel_base(wb) {}
};

// hybrid_astar.rs:13-62, constructor arguments in the reference's order
struct Config {
  double xy_reso, yaw_reso, move_step;
  size_t n_steer, collision_check_step;
  double extend_bound, gear_cost, backward_cost, steer_change_cost, steer_angle_cost, h_cost;
  Car car;
  double max_steer;
  Config(double xy_reso, double yaw_reso, double move_step, size_t n_steer, size_t collision_check_step,
         double extend_bound, double gear_cost, double backward_cost, double steer_change_cost,
         double steer_angle_cost, double h_cost, Car car, double max_steer)
      : xy_reso(xy_reso), yaw_reso(yaw_reso), move_step(move_step), n_steer(n_steer),
        collision_check_step(collision_check_step), extend_bound(extend_bound), gear_cost(gear_cost),
        backward_cost(backward_cost), steer_change_cost(steer_change_cost),
        steer_angle_cost(steer_angle_cost), h_cost(h_cost), car(car), max_steer(max_steer) {}
  pp_config abi() const {
    return pp_config{xy_reso, yaw_reso, move_step, (uint64_t)n_steer, (uint64_t)collision_check_step,
                     extend_bound, gear_cost, backward_cost, steer_change_cost, steer_angle_cost, h_cost,
                     pp_car{car.center2front, car.center2back, car.width, car.wheel_base}, max_steer};
  }
};

// hybrid_astar.rs:80-87
struct HybridAstarPath {
  std::vector<double> x, y, yaw;
  std::vector<int64_t> direction;
  double cost = 0.0;
};

// reeds_shepp.rs:8-39
struct RSPath {
  std::vector<double> lengths;
  std::vector<char> ctypes;
  double total_length = 0.0;
  std::vector<double> x, y, yaw;
  std::vector<int64_t> directions;
};

// One device context (and optionally one obstacle map) kept across calls.
class Session {
 public:
  explicit Session(int device = 0) {
    int32_t st = pp_ctx_create(device, nullptr, &ctx_);
    if (st != PP_OK) throw Panic(st, "pp_ctx_create");
  }
  Session(const std::vector<double>& ox, const std::vector<double>& oy, int device = 0) : Session(device) {
    set_obstacles(ox, oy);
  }
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;
  ~Session() {
    if (map_) pp_map_destroy(map_);
    if (ctx_) pp_ctx_destroy(ctx_);
  }
  void set_obstacles(const std::vector<double>& ox, const std::vector<double>& oy) {
    if (map_) pp_map_destroy(map_);
    map_ = nullptr;
    // the reference zips the two lists (hybrid_astar.rs:137, astar.rs:251)
    int32_t st = pp_map_create(ctx_, ox.data(), oy.data(), std::min(ox.size(), oy.size()), &map_);
    if (st != PP_OK) throw Panic(st, "pp_map_create");
  }
  pp_ctx* ctx() const { return ctx_; }
  pp_map* map() const { return map_; }

  // hybrid_astar_planning, hybrid_astar.rs:89-216 (rec = None)
  std::optional<HybridAstarPath> hybrid_astar_planning(double sx, double sy, double syaw, double gx, double gy,
                                                       double gyaw, const Config& conf) const {
    const pp_config c = conf.abi();
    const double start[3] = {sx, sy, syaw}, goal[3] = {gx, gy, gyaw};
    HybridAstarPath p;
    size_t cap = 4096;
    for (;;) {
      p.x.resize(cap);
      p.y.resize(cap);
      p.yaw.resize(cap);
      p.direction.resize(cap);
      pp_path out{p.x.data(), p.y.data(), p.yaw.data(), p.direction.data(), cap, 0, 0.0};
      const int32_t st = pp_plan(ctx_, map_, &c, start, goal, &out, nullptr);
      if (st == PP_NO_PATH) return std::nullopt;
      if (st == PP_ERR_CAPACITY) {
        cap = out.len;
        continue;
      }
      if (st != PP_OK) throw Panic(st, "hybrid_astar_planning");
      p.x.resize(out.len);
      p.y.resize(out.len);
      p.yaw.resize(out.len);
      p.direction.resize(out.len);
      p.cost = out.cost;
      return p;
    }
  }

  // astar_planning, astar.rs:53-132 (rec = None)
  std::optional<std::pair<std::vector<double>, std::vector<double>>> astar_planning(double sx, double sy, double gx,
                                                                                   double gy, double reso,
                                                                                   double rr) const {
    std::vector<double> px(1024), py(1024);
    for (;;) {
      uint64_t n = 0;
      const int32_t st = pp_astar_plan(ctx_, map_, sx, sy, gx, gy, reso, rr, px.data(), py.data(), px.size(), &n,
                                       nullptr, 0, nullptr);
      if (st == PP_NO_PATH) return std::nullopt;
      if (st == PP_ERR_CAPACITY) {
        px.resize(n);
        py.resize(n);
        continue;
      }
      if (st != PP_OK) throw Panic(st, "astar_planning");
      px.resize(n);
      py.resize(n);
      return std::make_pair(std::move(px), std::move(py));
    }
  }

  // calc_holonomic_heuristic_with_obstacle, astar.rs:134-198: hmap[x - minx][y - miny]
  std::vector<std::vector<double>> calc_holonomic_heuristic_with_obstacle(double goal_x, double goal_y, double reso,
                                                                          double rr) const {
    uint64_t dims[6];
    int32_t st = pp_heuristic(ctx_, map_, goal_x, goal_y, reso, rr, nullptr, 0, dims);
    if (st != PP_OK && st != PP_ERR_CAPACITY) throw Panic(st, "calc_holonomic_heuristic_with_obstacle");
    std::vector<double> flat(dims[4] * dims[5]);
    st = pp_heuristic(ctx_, map_, goal_x, goal_y, reso, rr, flat.data(), flat.size(), dims);
    if (st != PP_OK) throw Panic(st, "calc_holonomic_heuristic_with_obstacle");
    std::vector<std::vector<double>> hmap(dims[4]);
    for (uint64_t x = 0; x < dims[4]; ++x) hmap[x].assign(flat.begin() + x * dims[5], flat.begin() + (x + 1) * dims[5]);
    return hmap;
  }

  // is_collision, hybrid_astar.rs:417-448
  bool is_collision(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& yaw,
                    const Config& conf) const {
    const pp_config c = conf.abi();
    int32_t hit = 0;
    const uint64_t n = std::min(x.size(), std::min(y.size(), yaw.size()));
    const int32_t st = pp_is_collision(ctx_, map_, &c, n, x.data(), y.data(), yaw.data(), &hit, nullptr);
    if (st != PP_OK) throw Panic(st, "is_collision");
    return hit != 0;
  }

  // calc_all_paths, reeds_shepp.rs:87-128
  std::vector<RSPath> calc_all_paths(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                                     double max_curvature, double step_size) const {
    std::vector<pp_rs_word> words(PP_RS_MAX_PATHS);
    std::vector<uint64_t> off(PP_RS_MAX_PATHS + 1);
    size_t cap = 1 << 16;
    for (;;) {
      std::vector<double> tx(cap), ty(cap), tyaw(cap);
      std::vector<int64_t> td(cap);
      int32_t n = 0;
      const int32_t st = pp_rs_all_paths(ctx_, sx, sy, syaw, gx, gy, gyaw, max_curvature, step_size, &n,
                                         words.data(), off.data(), tx.data(), ty.data(), tyaw.data(), td.data(), cap);
      if (st == PP_ERR_CAPACITY) {
        cap = std::max<size_t>(2 * cap, off[n > 0 ? n : 0]);
        continue;
      }
      if (st != PP_OK) throw Panic(st, "calc_all_paths");
      std::vector<RSPath> paths((size_t)n);
      for (int32_t k = 0; k < n; ++k) fill(paths[k], words[k], tx, ty, tyaw, td, off[k], off[k + 1]);
      return paths;
    }
  }

  // calc_optimal_path, reeds_shepp.rs:60-85
  RSPath calc_optimal_path(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                           double max_curvature, double step_size) const {
    size_t cap = 1 << 14;
    for (;;) {
      std::vector<double> tx(cap), ty(cap), tyaw(cap);
      std::vector<int64_t> td(cap);
      pp_rs_word w;
      uint64_t len = 0;
      const int32_t st = pp_rs_optimal_path(ctx_, sx, sy, syaw, gx, gy, gyaw, max_curvature, step_size, &w,
                                            tx.data(), ty.data(), tyaw.data(), td.data(), cap, &len);
      if (st == PP_ERR_CAPACITY) {
        cap = std::max<size_t>(2 * cap, len);
        continue;
      }
      if (st != PP_OK) throw Panic(st, "calc_optimal_path");
      RSPath p;
      fill(p, w, tx, ty, tyaw, td, 0, len);
      return p;
    }
  }

 private:
  static void fill(RSPath& p, const pp_rs_word& w, const std::vector<double>& tx, const std::vector<double>& ty,
                   const std::vector<double>& tyaw, const std::vector<int64_t>& td, uint64_t a, uint64_t b) {
    p.lengths.assign(w.lengths, w.lengths + w.nseg);
    p.ctypes.assign(w.ctypes, w.ctypes + w.nseg);
    p.total_length = w.total_length;
    p.x.assign(tx.begin() + a, tx.begin() + b);
    p.y.assign(ty.begin() + a, ty.begin() + b);
    p.yaw.assign(tyaw.begin() + a, tyaw.begin() + b);
    p.directions.assign(td.begin() + a, td.begin() + b);
  }
  pp_ctx* ctx_ = nullptr;
  pp_map* map_ = nullptr;
};

// ---- the reference's free functions, one device session per call

// hybrid_astar.rs:89-100
inline std::optional<HybridAstarPath> hybrid_astar_planning(double sx, double sy, double syaw, double gx, double gy,
                                                            double gyaw, const std::vector<double>& ox,
                                                            const std::vector<double>& oy, const Config& conf) {
  return Session(ox, oy).hybrid_astar_planning(sx, sy, syaw, gx, gy, gyaw, conf);
}
// astar.rs:53-63
inline std::optional<std::pair<std::vector<double>, std::vector<double>>> astar_planning(
    double sx, double sy, double gx, double gy, const std::vector<double>& ox, const std::vector<double>& oy,
    double reso, double rr) {
  return Session(ox, oy).astar_planning(sx, sy, gx, gy, reso, rr);
}
// astar.rs:134-140 (the reference passes a Node and reads its last x / y)
inline std::vector<std::vector<double>> calc_holonomic_heuristic_with_obstacle(
    double goal_x, double goal_y, const std::vector<double>& ox, const std::vector<double>& oy, double reso,
    double rr) {
  return Session(ox, oy).calc_holonomic_heuristic_with_obstacle(goal_x, goal_y, reso, rr);
}
// reeds_shepp.rs:87-96
inline std::vector<RSPath> calc_all_paths(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                                          double max_curvature, double step_size) {
  return Session().calc_all_paths(sx, sy, syaw, gx, gy, gyaw, max_curvature, step_size);
}
// reeds_shepp.rs:60-69
inline RSPath calc_optimal_path(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                                double max_curvature, double step_size) {
  return Session().calc_optimal_path(sx, sy, syaw, gx, gy, gyaw, max_curvature, step_size);
}
// hybrid_astar.rs:450-489
inline double calc_rs_path_cost(const RSPath& rspath, const Config& conf) {
  pp_rs_word w{};
  w.nseg = (uint8_t)rspath.lengths.size();
  for (size_t i = 0; i < rspath.lengths.size() && i < PP_RS_MAX_SEG; ++i) {
    w.lengths[i] = rspath.lengths[i];
    w.ctypes[i] = (uint8_t)rspath.ctypes[i];
  }
  w.total_length = rspath.total_length;
  const pp_config c = conf.abi();
  return pp_rs_path_cost(&w, &c);
}

}  // namespace path_planner

#endif  // PATH_PLANNER_HPP_
