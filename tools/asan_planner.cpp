// Host-side memory checking of the planner: a two-lane batch with failing and capped queries, twice.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "pp.h"
static void design_obstacles(int x, int y, std::vector<double>& ox, std::vector<double>& oy) {
  for (int i = 0; i < x; ++i) { ox.push_back(i); oy.push_back(0.0); }
  for (int i = 0; i < x; ++i) { ox.push_back(i); oy.push_back(y - 1); }
  for (int i = 0; i < y; ++i) { ox.push_back(0.0); oy.push_back(i); }
  for (int i = 0; i < y; ++i) { ox.push_back(x - 1); oy.push_back(i); }
  for (int i = 10; i < 21; ++i) { ox.push_back(i); oy.push_back(15.0); }
  for (int i = 0; i < 15; ++i) { ox.push_back(20.0); oy.push_back(i); }
  for (int i = 15; i < 30; ++i) { ox.push_back(30.0); oy.push_back(i); }
  for (int i = 0; i < 16; ++i) { ox.push_back(40.0); oy.push_back(i); }
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int n = argc > 1 ? atoi(argv[1]) : 1100;
  const int cap = argc > 2 ? atoi(argv[2]) : 150;
  pp_ctx* ctx = nullptr;
  if (pp_ctx_create(0, nullptr, &ctx) != 0) return 2;
  std::vector<double> ox, oy;
  design_obstacles(51, 31, ox, oy);
  pp_map* map = nullptr;
  if (pp_map_create(ctx, ox.data(), oy.data(), ox.size(), &map) != 0) return 3;
  pp_config conf{2.0, 15.0 * 3.14159265358979323846 / 180.0, 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0, 15.0,
                 {4.5, 1.0, 3.0, 3.5}, 0.6};
  pp_ctx_set_max_expansions(ctx, (uint64_t)cap);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> ux(4, 46), uy(4, 26), ua(-3, 3);
  for (int rep = 0; rep < 3; ++rep) {
    const int nq = rep == 2 ? n / 3 : n;
    std::vector<double> st(3 * nq), go(3 * nq);
    for (int i = 0; i < nq; ++i) {
      st[3 * i] = ux(rng); st[3 * i + 1] = uy(rng); st[3 * i + 2] = ua(rng);
      go[3 * i] = ux(rng); go[3 * i + 1] = uy(rng); go[3 * i + 2] = ua(rng);
      if (i % 7 == 0) { go[3 * i] = 60.0; go[3 * i + 1] = 20.0; }
      if (i % 11 == 3) { go[3 * i] = st[3 * i]; go[3 * i + 1] = st[3 * i + 1]; go[3 * i + 2] = st[3 * i + 2]; }
      if (i % 5 == 2) { st[3 * i] = 25.0; st[3 * i + 1] = 8.0; st[3 * i + 2] = 0.0; go[3 * i] = 45.0; go[3 * i + 1] = 28.5; go[3 * i + 2] = 0.0; }
    }
    const uint64_t pc = 1024;
    std::vector<double> px((size_t)nq * pc), py((size_t)nq * pc), pyaw((size_t)nq * pc);
    std::vector<int64_t> pd((size_t)nq * pc);
    std::vector<pp_path> out(nq);
    std::vector<pp_plan_stats> stats(nq);
    std::vector<uint64_t> closed((size_t)nq * cap);
    std::vector<int32_t> status(nq);
    for (int i = 0; i < nq; ++i) {
      out[i] = pp_path{px.data() + (size_t)i * pc, py.data() + (size_t)i * pc, pyaw.data() + (size_t)i * pc,
                       pd.data() + (size_t)i * pc, pc, 0, 0.0};
      stats[i] = pp_plan_stats{};
      stats[i].closed_cap = (uint64_t)cap;
      stats[i].closed_order = closed.data() + (size_t)i * cap;
    }
    const int32_t rc = pp_plan_batch(ctx, map, &conf, (uint64_t)nq, st.data(), go.data(), out.data(), stats.data(), status.data());
    long ok = 0, lim = 0; unsigned long long ex = 0;
    for (int i = 0; i < nq; ++i) { ok += status[i] == 0; lim += status[i] == -11; ex += stats[i].expansions; }
    printf("rep %d rc %d: %d queries, %ld found, %ld at the cap, %llu expansions\n", rep, rc, nq, ok, lim, ex);
  }
  {  // the heuristic grid in column slabs over three contexts, a host thread each
    const uint64_t xw = 301, yw = 257;
    std::vector<uint8_t> obs(xw * yw, 0);
    for (uint64_t i = 0; i < xw * yw; ++i) obs[i] = (rng() % 100) < 22;
    obs[200 * yw + 90] = 0;
    std::vector<double> one(xw * yw), three(xw * yw);
    pp_ctx *c2 = nullptr, *c3 = nullptr;
    if (pp_ctx_create(0, nullptr, &c2) != 0 || pp_ctx_create(0, nullptr, &c3) != 0) return 4;
    pp_ctx* all[3] = {ctx, c2, c3};
    uint64_t r1 = 0, r3 = 0;
    const int32_t a = pp_wavefront_tiled(all, 1, xw, yw, obs.data(), 200, 90, one.data(), 0.0, &r1);
    const int32_t b = pp_wavefront_tiled(all, 3, xw, yw, obs.data(), 200, 90, three.data(), 8.0, &r3);
    size_t differ = 0;
    for (uint64_t i = 0; i < xw * yw; ++i) differ += !(one[i] == three[i]);
    printf("tiled rc %d %d: %llu and %llu exchange rounds, %zu cells differ\n", a, b, (unsigned long long)r1,
           (unsigned long long)r3, differ);
    pp_ctx_destroy(c3);
    pp_ctx_destroy(c2);
  }
  pp_map_destroy(map);
  pp_ctx_destroy(ctx);
  printf("done\n");
  // leave without the exit handlers: with this ROCm the sanitizer's own device-allocator check trips
  // inside the HIP runtime's unload (operator delete after the device runtime is gone), whatever ran
  _Exit(0);
}
