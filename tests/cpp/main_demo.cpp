// The three demos of the reference's main (src/main.rs:86-296) through the C++ mirror of its
// functions, printed as one JSON object for tests/test_gpu_cpp_mirror.py.
#include <cstdio>
#include <vector>

#include "path_planner.hpp"

namespace pp = path_planner;
static const double kPi = 3.14159265358979323846;
static double to_rad(double deg) { return deg * kPi / 180.0; }

static void setup_astar_env(std::vector<double>& ox, std::vector<double>& oy) {  // main.rs:14-44
  for (int i = 0; i < 60; ++i) { ox.push_back(i); oy.push_back(0.0); }
  for (int i = 0; i < 60; ++i) { ox.push_back(60.0); oy.push_back(i); }
  for (int i = 0; i < 61; ++i) { ox.push_back(i); oy.push_back(60.0); }
  for (int i = 0; i < 61; ++i) { ox.push_back(0.0); oy.push_back(i); }
  for (int i = 0; i < 40; ++i) { ox.push_back(20.0); oy.push_back(i); }
  for (int i = 0; i < 40; ++i) { ox.push_back(40.0); oy.push_back(60.0 - i); }
}

static void design_obstacles(int x, int y, std::vector<double>& ox, std::vector<double>& oy) {  // main.rs:46-84
  for (int i = 0; i < x; ++i) { ox.push_back(i); oy.push_back(0.0); }
  for (int i = 0; i < x; ++i) { ox.push_back(i); oy.push_back(y - 1); }
  for (int i = 0; i < y; ++i) { ox.push_back(0.0); oy.push_back(i); }
  for (int i = 0; i < y; ++i) { ox.push_back(x - 1); oy.push_back(i); }
  for (int i = 10; i < 21; ++i) { ox.push_back(i); oy.push_back(15.0); }
  for (int i = 0; i < 15; ++i) { ox.push_back(20.0); oy.push_back(i); }
  for (int i = 15; i < 30; ++i) { ox.push_back(30.0); oy.push_back(i); }
  for (int i = 0; i < 16; ++i) { ox.push_back(40.0); oy.push_back(i); }
}

int main() {
  try {
    printf("{");
    {  // Reeds-Shepp legs, main.rs:93-137
      const double st[8][3] = {{0, 0, 0}, {10, 10, -90}, {20, 5, 60}, {30, 10, 120},
                               {35, -5, 30}, {25, -10, -120}, {15, -15, 100}, {0, -10, -90}};
      pp::Session s;
      printf("\"rs_total\": [");
      for (int i = 0; i < 7; ++i) {
        pp::RSPath p = s.calc_optimal_path(st[i][0], st[i][1], to_rad(st[i][2]), st[i + 1][0], st[i + 1][1],
                                           to_rad(st[i + 1][2]), 0.1, 0.2);
        printf("%s[%.17g, %zu, \"%s\"]", i ? ", " : "", p.total_length, p.x.size(),
               std::string(p.ctypes.begin(), p.ctypes.end()).c_str());
      }
      printf("], ");
    }
    {  // plain grid A*, main.rs:157-191
      std::vector<double> ox, oy;
      setup_astar_env(ox, oy);
      auto path = pp::astar_planning(10.0, 10.0, 50.0, 50.0, ox, oy, 1.0, 2.0);
      printf("\"astar_len\": %zu, ", path ? path->first.size() : (size_t)0);
      if (path) printf("\"astar_first\": [%.17g, %.17g], \"astar_last\": [%.17g, %.17g], ", path->first.front(),
                       path->second.front(), path->first.back(), path->second.back());
    }
    {  // hybrid A*, main.rs:215-269
      std::vector<double> ox, oy;
      design_obstacles(51, 31, ox, oy);
      pp::Car car(4.5, 1.0, 3.0, 3.5);
      pp::Config conf(2.0, to_rad(15.0), 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0, 15.0, car, 0.6);
      auto path = pp::hybrid_astar_planning(10.0, 7.0, to_rad(120.0), 45.0, 20.0, to_rad(90.0), ox, oy, conf);
      printf("\"hybrid_points\": %zu, \"hybrid_cost\": %.17g, ", path ? path->x.size() : (size_t)0,
             path ? path->cost : -1.0);
      pp::Session s(ox, oy);
      auto h = s.calc_holonomic_heuristic_with_obstacle(45.0, 20.0, 2.0, 1.0);
      printf("\"hmap_dims\": [%zu, %zu], \"hmap_start\": %.17g, ", h.size(), h[0].size(), h[5][4]);
      printf("\"collides_at_wall\": %d, \"collides_at_start\": %d, ",
             (int)s.is_collision({20.0}, {5.0}, {0.0}, conf), (int)s.is_collision({10.0}, {7.0}, {to_rad(120.0)}, conf));
      // a goal outside the map is the reference's hmap index panic
      bool panicked = false;
      try {
        s.hybrid_astar_planning(10.0, 7.0, 0.0, 60.0, 20.0, 0.0, conf);
      } catch (const pp::Panic& e) {
        panicked = e.status == PP_ERR_BOUNDS;
      }
      printf("\"panics_on_goal_outside\": %d", (int)panicked);
    }
    printf("}\n");
  } catch (const std::exception& e) {
    fprintf(stderr, "main_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
