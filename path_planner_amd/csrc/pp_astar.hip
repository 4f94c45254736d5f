// pp_astar.hip -- the plain grid A* planner (astar_planning, astar.rs:53-132).
//
// The obstacle raster is built on the device (calc_obsmap, astar.rs:244-261: the reference's
// O(xw * yw * n_obstacles) triple loop is where its time goes on any map of size); the heap search
// itself is serial by construction -- which parent a cell ends up with depends on the order the
// cells are popped, ties included -- and runs on the host, one thread per query of a batch.
#include "pp_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// Heap entry.  The reference's BinaryHeap<(Reverse<NotNan<f64>>, usize)> (astar.rs:86-90) pops the
// lowest f, and among equal f the LARGEST cell index.
struct Ent {
  double f;
  uint64_t key;
};
inline bool before(const Ent& a, const Ent& b) {  // a pops before b
  return a.f < b.f || (a.f == b.f && a.key > b.key);
}

class MinHeap {
 public:
  bool empty() const { return v_.empty(); }
  void push(Ent e) {
    size_t i = v_.size();
    v_.push_back(e);
    while (i > 0) {
      size_t p = (i - 1) / 2;
      if (!before(v_[i], v_[p])) break;
      std::swap(v_[i], v_[p]);
      i = p;
    }
  }
  Ent pop() {
    Ent top = v_[0];
    v_[0] = v_.back();
    v_.pop_back();
    size_t i = 0, n = v_.size();
    for (;;) {
      size_t l = 2 * i + 1, r = l + 1, m = i;
      if (l < n && before(v_[l], v_[m])) m = l;
      if (r < n && before(v_[r], v_[m])) m = r;
      if (m == i) break;
      std::swap(v_[i], v_[m]);
      i = m;
    }
    return top;
  }

 private:
  std::vector<Ent> v_;
};

struct Grid {
  uint64_t minx, miny, maxx, maxy, xw, yw;
  const uint8_t* obs;  // obs[(x - minx) * yw + (y - miny)]
};

// One query.  Cells are kept in dense arrays indexed by calc_index; the only node whose index can
// fall outside the grid (or alias another cell's) is the start, which the reference inserts
// without check_node (astar.rs:84): it gets slot `ncell` when its index is out of range, and
// whichever slot holds it reports the start's own coordinates.
struct Search {
  const Grid& g;
  uint64_t ncell;
  std::vector<uint8_t> state;  // 0 unseen, 1 open, 2 closed
  std::vector<double> cost;
  std::vector<int64_t> pind;
  uint64_t sx, sy, skey;

  explicit Search(const Grid& grid) : g(grid), ncell(grid.xw * grid.yw) {
    state.assign(ncell + 1, 0);
    cost.assign(ncell + 1, 0.0);
    pind.assign(ncell + 1, -1);
  }
  // calc_index, astar.rs:228-230 (usize arithmetic wraps in a release build)
  uint64_t index(uint64_t x, uint64_t y) const { return (y - g.miny) * g.xw + (x - g.minx); }
  bool known(uint64_t key) const { return key < ncell || key == skey; }
  uint64_t slot(uint64_t key) const { return key < ncell ? key : ncell; }
  void coords(uint64_t key, uint64_t& x, uint64_t& y) const {
    if (key == skey) {
      x = sx;
      y = sy;
    } else {
      x = key % g.xw + g.minx;
      y = key / g.xw + g.miny;
    }
  }
};

int32_t astar_search(const Grid& g, double sxm, double sym, double gxm, double gym, double reso,
                     std::vector<double>& px, std::vector<double>& py, uint64_t* closed_order,
                     uint64_t closed_cap, uint64_t* expansions) {
  static const int mdx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};  // get_motion, astar.rs:297-308
  static const int mdy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  double ucost[8];
  for (int k = 0; k < 8; ++k) ucost[k] = std::hypot((double)mdx[k], (double)mdy[k]);  // :213-215

  Search s(g);
  s.sx = pp_sat_usize(std::round(sxm / reso));  // astar.rs:64-69
  s.sy = pp_sat_usize(std::round(sym / reso));
  const uint64_t gx = pp_sat_usize(std::round(gxm / reso));  // :70-75
  const uint64_t gy = pp_sat_usize(std::round(gym / reso));
  s.skey = s.index(s.sx, s.sy);
  const uint64_t goal_key = s.index(gx, gy);
  auto fvalue = [&](uint64_t x, uint64_t y, double c) {  // :217-226
    return c + std::hypot((double)x - (double)gx, (double)y - (double)gy);
  };

  MinHeap heap;
  s.state[s.slot(s.skey)] = 1;
  s.cost[s.slot(s.skey)] = 0.0;
  s.pind[s.slot(s.skey)] = -1;
  heap.push({fvalue(s.sx, s.sy, 0.0), s.skey});
  uint64_t n_open = 1, n_closed = 0;
  bool found = false;
  while (n_open > 0) {  // :93
    const Ent e = heap.pop();
    const uint64_t ind = e.key, sl = s.slot(ind);
    s.state[sl] = 2;
    --n_open;
    if (closed_order && n_closed < closed_cap) closed_order[n_closed] = ind;
    ++n_closed;
    if (ind == goal_key) {  // :101
      found = true;
      break;
    }
    uint64_t cx, cy;
    s.coords(ind, cx, cy);
    const double ccost = s.cost[sl];
    for (int k = 0; k < 8; ++k) {
      const uint64_t nx = (uint64_t)((int64_t)cx + mdx[k]);  // (x as isize + dx) as usize
      const uint64_t ny = (uint64_t)((int64_t)cy + mdy[k]);
      const double ncost = ccost + ucost[k];
      // check_node, :200-211
      if (nx <= g.minx || nx >= g.maxx || ny <= g.miny || ny >= g.maxy) continue;
      if (g.obs[(nx - g.minx) * g.yw + (ny - g.miny)]) continue;
      const uint64_t nk = s.index(nx, ny);  // inside the grid, so also its slot
      if (s.state[nk] == 2) continue;
      if (s.state[nk] == 1) {
        if (s.cost[nk] > ncost) {  // :120-123; the heap entry keeps its old priority
          s.cost[nk] = ncost;
          s.pind[nk] = (int64_t)ind;
        }
      } else {
        s.state[nk] = 1;
        s.cost[nk] = ncost;
        s.pind[nk] = (int64_t)ind;
        ++n_open;
        heap.push({fvalue(nx, ny, ncost), nk});
      }
    }
  }
  if (expansions) *expansions = n_closed;
  if (!found) return PP_NO_PATH;  // :131

  // extract_path, :263-295: the goal's coordinates, then every closed node from the goal back to
  // the start (so the goal appears twice), reversed and scaled by reso
  px.clear();
  py.clear();
  px.push_back((double)gx);
  py.push_back((double)gy);
  uint64_t k = goal_key;
  while (s.known(k) && s.state[s.slot(k)] == 2) {
    uint64_t x, y;
    s.coords(k, x, y);
    px.push_back((double)x);
    py.push_back((double)y);
    const uint64_t sl = s.slot(k);
    const bool is_start = x == s.sx && y == s.sy && s.cost[sl] == 0.0 && s.pind[sl] == -1;
    k = (uint64_t)s.pind[sl];
    if (is_start) break;
  }
  std::reverse(px.begin(), px.end());
  std::reverse(py.begin(), py.end());
  for (double& v : px) v *= reso;
  for (double& v : py) v *= reso;
  return PP_OK;
}

int32_t build_raster(pp_ctx* ctx, const pp_map* map, double reso, double rr, uint64_t dims[6],
                     std::vector<uint8_t>& obs) {
  int32_t st = pp_heuristic_checks(ctx, map, reso, dims);
  if (st != PP_OK) return st;
  const uint64_t ncells = dims[4] * dims[5];
  obs.assign(ncells, 0);
  if (ncells == 0) return PP_OK;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  st = pp_scratch_reserve(ctx, ncells);
  if (st != PP_OK) return st;
  uint8_t* d_obs = (uint8_t*)ctx->scratch;
  st = pp_obsmap_dev(ctx, map, reso, rr, dims, d_obs);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipMemcpyAsync(obs.data(), d_obs, ncells, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

}  // namespace

extern "C" int32_t pp_astar_plan(pp_ctx* ctx, const pp_map* map, double sx, double sy, double gx,
                                 double gy, double reso, double rr, double* path_x, double* path_y,
                                 uint64_t cap, uint64_t* len, uint64_t* closed_order,
                                 uint64_t closed_cap, uint64_t* expansions) {
  if (!len) return PP_ERR_INVALID;
  *len = 0;
  uint64_t dims[6];
  std::vector<uint8_t> obs;
  int32_t st = build_raster(ctx, map, reso, rr, dims, obs);
  if (st != PP_OK) return st;
  Grid g{dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], obs.data()};
  std::vector<double> px, py;
  st = astar_search(g, sx, sy, gx, gy, reso, px, py, closed_order, closed_cap, expansions);
  if (st != PP_OK) return st;
  *len = px.size();
  if (px.size() > cap) return PP_ERR_CAPACITY;
  if (!path_x || !path_y) return PP_ERR_INVALID;
  std::memcpy(path_x, px.data(), px.size() * sizeof(double));
  std::memcpy(path_y, py.data(), py.size() * sizeof(double));
  return PP_OK;
}

extern "C" int32_t pp_astar_plan_batch(pp_ctx* ctx, const pp_map* map, uint64_t n,
                                       const double* starts, const double* goals, double reso,
                                       double rr, double* path_x, double* path_y, uint64_t cap,
                                       uint64_t* path_off, int32_t* status, uint64_t* expansions) {
  if (n && (!starts || !goals || !path_off || !status)) return PP_ERR_INVALID;
  uint64_t dims[6];
  std::vector<uint8_t> obs;
  int32_t st = build_raster(ctx, map, reso, rr, dims, obs);
  if (st != PP_OK) return st;
  Grid g{dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], obs.data()};
  std::vector<std::vector<double>> px(n), py(n);
  std::atomic<uint64_t> next{0};
  auto work = [&]() {
    for (uint64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      uint64_t ex = 0;
      status[i] = astar_search(g, starts[2 * i], starts[2 * i + 1], goals[2 * i], goals[2 * i + 1],
                               reso, px[i], py[i], nullptr, 0, &ex);
      if (expansions) expansions[i] = ex;
    }
  };
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = (unsigned)std::min<uint64_t>(std::min<uint64_t>(n, hw), 16);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  uint64_t off = 0;
  for (uint64_t i = 0; i < n; ++i) {
    path_off[i] = off;
    if (status[i] == PP_OK) off += px[i].size();
  }
  path_off[n] = off;
  if (off > cap) return PP_ERR_CAPACITY;
  if (off && (!path_x || !path_y)) return PP_ERR_INVALID;
  for (uint64_t i = 0; i < n; ++i) {
    if (status[i] != PP_OK) continue;
    std::memcpy(path_x + path_off[i], px[i].data(), px[i].size() * sizeof(double));
    std::memcpy(path_y + path_off[i], py[i].data(), py[i].size() * sizeof(double));
  }
  return PP_OK;
}
