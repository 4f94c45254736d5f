// pp_planner.hip -- hybrid_astar_planning (hybrid_astar.rs:89-250) with the per-expansion work on
// the device.
//
// The best-first search stays serial per query on the host, exactly as the reference runs it:
// open/closed maps keyed by calc_index, a lazy-duplicate heap ordered (f ascending, key
// descending), stale entries skipped.  What each expansion needs -- the analytic-expansion shot
// from the node just closed and its 2*(2*n_steer-1) motion primitives, each with footprint checks
// and the heuristic lookup -- is handed to the device.  One query alone gives the device 78 lanes
// of work per step, so queries are advanced in lock-step rounds: every live query closes one
// node, one launch set expands all of them, and the host applies the results to each query's
// heap (over a small thread pool, queries being independent).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <unordered_map>

#include "pp_expand.h"
#include "pp_internal.h"
#include "pp_rs.h"

extern "C" int32_t pp_wavefront_dev(pp_ctx*, uint64_t, uint64_t, const uint8_t*, uint64_t, uint64_t,
                                    double*, uint64_t*);

namespace {

const double kPi = 3.14159265358979323846;

double pi_2_pi(double theta) {  // reeds_shepp.rs:727-738
  if (!(std::fabs(theta) < 1.0e9)) return theta;
  while (theta > kPi) theta -= 2.0 * kPi;
  while (theta < -kPi) theta += 2.0 * kPi;
  return theta;
}

struct SNode {  // data_struct.rs:2-14 without the point lists (regenerated for the final path)
  uint64_t xind, yind;
  int64_t yawind, direction;
  double x, y, yaw;  // last element of the node's lists
  double steer, cost;
  int64_t pind;      // parent's calc_index key, -1 for the start node
  int32_t parent;    // parent's slot in the node store
  int32_t prim;      // primitive index that produced it
};

struct HeapEnt {
  double f;
  uint64_t key;
};
struct HeapLess {  // BinaryHeap<(Reverse(f), key)>: top = min f, ties -> largest key
  bool operator()(const HeapEnt& a, const HeapEnt& b) const {
    if (a.f != b.f) return a.f > b.f;
    return a.key < b.key;
  }
};

struct Query {
  double start[3], goal[3];
  int32_t hmap_slot = -1;
  std::vector<SNode> nodes;
  std::unordered_map<uint64_t, int32_t> open, closed;
  std::priority_queue<HeapEnt, std::vector<HeapEnt>, HeapLess> heap;
  int32_t status = PP_OK;
  bool live = false;
  int32_t curr = -1;       // node closed in this round
  uint64_t curr_key = 0;
  // outcome
  bool found = false;
  int32_t final_parent = -1;
  int32_t final_rs_idx = -1;
  double final_rs_cost = 0.0;
  pp_plan_stats st{};
};

// calc_index, hybrid_astar.rs:516-520 (release-mode wrapping arithmetic)
inline uint64_t calc_index(uint64_t xind, uint64_t yind, int64_t yawind, const pp_param& P) {
  return (uint64_t)(yawind - P.minyaw) * P.xw * P.yw + (yind - P.miny) * P.xw + (xind - P.minx);
}

// A small fork-join pool: run(n, fn) calls fn(i) for i in [0, n) on the workers and the caller.
class Pool {
 public:
  explicit Pool(unsigned workers) {
    for (unsigned i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    if (th_.empty() || n < 64) {
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> l(m_);
      fn_ = &fn;
      n_ = n;
      next_.store(0);
      pending_ = th_.size();
      ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      size_t i = next_.fetch_add(16);
      if (i >= n_) break;
      size_t e = std::min(n_, i + 16);
      for (; i < e; ++i) (*fn_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
      {
        std::lock_guard<std::mutex> l(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  size_t pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
struct PinBuf {
  void* p = nullptr;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int32_t pp_plan_batch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t nq,
                                 const double* starts, const double* goals, pp_path* out,
                                 pp_plan_stats* stats, int32_t* status) {
  if (!ctx || !map || !conf || (nq && (!starts || !goals || !out))) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (nq == 0) return PP_OK;
  if (nq > (1u << 24)) return PP_ERR_UNSUPPORTED;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;

  PpExpandPlan plan;
  st = pp_expand_prepare(ctx, map, conf, &plan);
  if (st != PP_OK) return st;
  const pp_param P = plan.param;
  const uint32_t NP = plan.n_prim;
  const double xyreso = conf->xy_reso, yawreso = conf->yaw_reso;
  const double maxc = std::tan(conf->max_steer) / conf->car.wheel_base;  // hybrid_astar.rs:388
  const double t_begin = now_ms();

  // ---- heuristic grids, one per distinct goal cell (hybrid_astar.rs:142-143: reso = xyreso, rr = 1.0)
  uint64_t dims[6];
  st = pp_heuristic_checks(ctx, map, xyreso, dims);
  if (st != PP_OK) return st;
  const uint64_t hxw = dims[4], hyw = dims[5], hcells = hxw * hyw;
  std::vector<Query> Q(nq);
  std::unordered_map<uint64_t, int32_t> goal_slot;
  std::vector<std::pair<uint64_t, uint64_t>> goal_cells;
  for (uint64_t i = 0; i < nq; ++i) {
    Query& q = Q[i];
    std::memcpy(q.start, starts + 3 * i, 24);
    std::memcpy(q.goal, goals + 3 * i, 24);
    q.st = pp_plan_stats{};
    if (stats) {
      q.st.closed_cap = stats[i].closed_cap;
      q.st.closed_order = stats[i].closed_order;
    }
    uint64_t gxi = pp_sat_usize(std::round(q.goal[0] / xyreso));  // astar.rs:141-146
    uint64_t gyi = pp_sat_usize(std::round(q.goal[1] / xyreso));
    if (hcells == 0 || gxi < dims[0] || gxi - dims[0] >= hxw || gyi < dims[1] || gyi - dims[1] >= hyw) {
      q.status = PP_ERR_BOUNDS;  // the goal's own hmap write panics (astar.rs:195)
      continue;
    }
    uint64_t gk = (gxi - dims[0]) * hyw + (gyi - dims[1]);
    auto it = goal_slot.find(gk);
    if (it == goal_slot.end()) {
      it = goal_slot.emplace(gk, (int32_t)goal_cells.size()).first;
      goal_cells.emplace_back(gxi - dims[0], gyi - dims[1]);
    }
    q.hmap_slot = it->second;
    q.live = true;
  }
  const size_t ngoals = goal_cells.size();
  DevBuf hm_buf, obs_buf, tab_buf;
  std::vector<double*> hm_ptr(ngoals);
  if (ngoals) {
    PP_HIP(ctx, hipMalloc(&hm_buf.p, ngoals * al256(hcells * 8)));
    PP_HIP(ctx, hipMalloc(&obs_buf.p, al256(hcells)));
    PP_HIP(ctx, hipMalloc(&tab_buf.p, ngoals * sizeof(double*)));
    st = pp_obsmap_dev(ctx, map, xyreso, 1.0, dims, (uint8_t*)obs_buf.p);
    if (st != PP_OK) return st;
    for (size_t g = 0; g < ngoals; ++g) {
      hm_ptr[g] = (double*)((char*)hm_buf.p + g * al256(hcells * 8));
      st = pp_wavefront_dev(ctx, hxw, hyw, (const uint8_t*)obs_buf.p, goal_cells[g].first,
                            goal_cells[g].second, hm_ptr[g], nullptr);
      if (st != PP_OK) return st;
    }
    PP_HIP(ctx, hipMemcpyAsync(tab_buf.p, hm_ptr.data(), ngoals * sizeof(double*), hipMemcpyHostToDevice, s));
    PP_HIP(ctx, hipStreamSynchronize(s));
  }
  const double t_heur = now_ms();

  // ---- start nodes (hybrid_astar.rs:103-154)
  {
    // hmap values under the start cells: one small gather per query
    for (uint64_t i = 0; i < nq; ++i) {
      Query& q = Q[i];
      if (!q.live) continue;
      SNode n{};
      n.xind = pp_sat_usize(std::round(q.start[0] / xyreso));
      n.yind = pp_sat_usize(std::round(q.start[1] / xyreso));
      n.yawind = pp_sat_isize(std::round(pi_2_pi(q.start[2]) / yawreso));
      n.direction = 1;
      n.x = q.start[0];
      n.y = q.start[1];
      n.yaw = q.start[2];
      n.steer = 0.0;
      n.cost = 0.0;
      n.pind = -1;
      n.parent = -1;
      n.prim = -1;
      uint64_t hx = n.xind - P.minx, hy = n.yind - P.miny;
      if (hx >= hxw || hy >= hyw) {  // hmap[..][..] index panic, hybrid_astar.rs:492
        q.status = PP_ERR_BOUNDS;
        q.live = false;
        continue;
      }
      double h = 0.0;
      PP_HIP(ctx, hipMemcpy(&h, hm_ptr[q.hmap_slot] + hx * hyw + hy, 8, hipMemcpyDeviceToHost));
      double f = n.cost + conf->h_cost * h;
      if (f != f) {
        q.status = PP_ERR_NAN_COST;
        q.live = false;
        continue;
      }
      uint64_t key = calc_index(n.xind, n.yind, n.yawind, P);
      q.nodes.push_back(n);
      q.open.emplace(key, 0);
      q.heap.push({f, key});
    }
  }

  // ---- device / pinned staging for the rounds
  const size_t B = nq;
  const size_t items = B * NP;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += al256(bytes);
    return o;
  };
  // inputs: q[6][B] (start pose + goal pose), pdir, psteer, pcost, hmap_id, motion
  const size_t o_q = take(6 * B * 8), o_dir = take(B * 8), o_steer = take(B * 8), o_cost = take(B * 8);
  const size_t o_hid = take(B * 4), o_mot = take(plan.motion.size() * 8);
  const size_t in_bytes = off;
  // outputs
  const size_t o_shot = take(B * sizeof(PpShotOut));
  const size_t o_valid = take(items), o_xind = take(items * 8), o_yind = take(items * 8), o_yaw = take(items * 8);
  const size_t o_c = take(items * 8), o_ex = take(items * 8), o_ey = take(items * 8), o_eyaw = take(items * 8);
  const size_t o_f = take(items * 8);
  const size_t out_bytes = off - in_bytes;
  const size_t o_cand = take(B * PP_RS_MAX_PATHS * sizeof(PpRsCand)), o_pair = take(B * sizeof(PpRsPair));
  DevBuf dev;
  PinBuf pin;
  PP_HIP(ctx, hipMalloc(&dev.p, off));
  PP_HIP(ctx, hipHostMalloc(&pin.p, in_bytes + out_bytes, hipHostMallocDefault));
  char* d = (char*)dev.p;
  char* h = (char*)pin.p;
  std::memcpy(h + o_mot, plan.motion.data(), plan.motion.size() * 8);

  PpExpandDev eo;
  eo.valid = (uint8_t*)(d + o_valid);
  eo.xind = (uint64_t*)(d + o_xind);
  eo.yind = (uint64_t*)(d + o_yind);
  eo.yawind = (int64_t*)(d + o_yaw);
  eo.cost = (double*)(d + o_c);
  eo.end_x = (double*)(d + o_ex);
  eo.end_y = (double*)(d + o_ey);
  eo.end_yaw = (double*)(d + o_eyaw);
  eo.fcost = (double*)(d + o_f);
  eo.hmaps = (const double* const*)tab_buf.p;
  eo.hmap_id = (const int32_t*)(d + o_hid);
  eo.hxw = hxw;
  eo.hyw = hyw;
  eo.h_cost = conf->h_cost;

  unsigned hw = std::thread::hardware_concurrency();
  unsigned workers = nq >= 128 ? std::min(15u, hw > 1 ? hw - 1 : 0u) : 0u;
  Pool pool(workers);
  std::vector<uint32_t> live_ids;
  live_ids.reserve(nq);

  // ---- lock-step rounds
  for (;;) {
    // (1) every live query closes its next node (hybrid_astar.rs:157-161)
    live_ids.clear();
    for (uint64_t i = 0; i < nq; ++i) {
      Query& q = Q[i];
      if (!q.live) continue;
      q.curr = -1;
      if (ctx->max_expansions && q.st.expansions >= ctx->max_expansions) {
        q.live = false;
        q.status = PP_ERR_LIMIT;
        continue;
      }
      while (!q.open.empty()) {
        HeapEnt e = q.heap.top();
        q.heap.pop();
        auto it = q.open.find(e.key);
        if (it == q.open.end()) {
          ++q.st.stale_pops;
          continue;
        }
        q.curr = it->second;
        q.curr_key = e.key;
        q.open.erase(it);
        q.closed[e.key] = q.curr;
        if (q.st.closed_order && q.st.expansions < q.st.closed_cap) q.st.closed_order[q.st.expansions] = e.key;
        ++q.st.expansions;
        break;
      }
      if (q.curr < 0) {  // open set exhausted: None
        q.live = false;
        q.status = PP_NO_PATH;
        continue;
      }
      live_ids.push_back((uint32_t)i);
    }
    const size_t nl = live_ids.size();
    if (nl == 0) break;

    // (2) stage parents, launch, fetch
    double* hq = (double*)(h + o_q);
    int64_t* hdir = (int64_t*)(h + o_dir);
    double* hsteer = (double*)(h + o_steer);
    double* hcost = (double*)(h + o_cost);
    int32_t* hhid = (int32_t*)(h + o_hid);
    for (size_t k = 0; k < nl; ++k) {
      const Query& q = Q[live_ids[k]];
      const SNode& n = q.nodes[q.curr];
      hq[0 * nl + k] = n.x;
      hq[1 * nl + k] = n.y;
      hq[2 * nl + k] = n.yaw;
      hq[3 * nl + k] = q.goal[0];
      hq[4 * nl + k] = q.goal[1];
      hq[5 * nl + k] = q.goal[2];
      hdir[k] = n.direction;
      hsteer[k] = n.steer;
      hcost[k] = n.cost;
      hhid[k] = q.hmap_slot;
    }
    PP_HIP(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    const double* dq = (const double*)(d + o_q);
    pp_rs_words_launch(ctx, nl, dq, maxc, (PpRsCand*)(d + o_cand), (PpRsPair*)(d + o_pair));
    pp_rs_shot_launch(ctx, map, conf, (uint32_t)nl, dq, maxc, (const PpRsCand*)(d + o_cand),
                      (const PpRsPair*)(d + o_pair), (PpShotOut*)(d + o_shot));
    pp_expand_launch(ctx, map, plan, nl, dq, dq + nl, dq + 2 * nl, (const int64_t*)(d + o_dir),
                     (const double*)(d + o_steer), (const double*)(d + o_cost),
                     (const double*)(d + o_mot), eo);
    PP_HIP(ctx, hipGetLastError());
    {  // fetch only what this round produced
      const size_t it8 = nl * NP * 8;
      PP_HIP(ctx, hipMemcpyAsync(h + o_shot, d + o_shot, nl * sizeof(PpShotOut), hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipMemcpyAsync(h + o_valid, d + o_valid, nl * NP, hipMemcpyDeviceToHost, s));
      const size_t offs[8] = {o_xind, o_yind, o_yaw, o_c, o_ex, o_ey, o_eyaw, o_f};
      for (size_t o : offs) PP_HIP(ctx, hipMemcpyAsync(h + o, d + o, it8, hipMemcpyDeviceToHost, s));
    }
    PP_HIP(ctx, hipStreamSynchronize(s));

    // (3) apply (hybrid_astar.rs:176-209)
    const PpShotOut* shot = (const PpShotOut*)(h + o_shot);
    const uint8_t* v_valid = (const uint8_t*)(h + o_valid);
    const uint64_t* v_x = (const uint64_t*)(h + o_xind);
    const uint64_t* v_y = (const uint64_t*)(h + o_yind);
    const int64_t* v_yaw = (const int64_t*)(h + o_yaw);
    const double* v_c = (const double*)(h + o_c);
    const double* v_ex = (const double*)(h + o_ex);
    const double* v_ey = (const double*)(h + o_ey);
    const double* v_eyaw = (const double*)(h + o_eyaw);
    const double* v_f = (const double*)(h + o_f);
    const double* mot = plan.motion.data();
    pool.run(nl, [&](size_t k) {
      Query& q = Q[live_ids[k]];
      const PpShotOut& so = shot[k];
      q.st.rs_paths += (uint64_t)so.n_paths;
      q.st.poses_checked += so.poses_handed;
      if (so.flags & 2u) ++q.st.ambiguous;
      if (so.flags & 4u) {
        q.status = PP_ERR_DEGENERATE_RS;
        q.live = false;
        return;
      }
      if (so.flags & 8u) {
        q.status = PP_ERR_NAN_COST;
        q.live = false;
        return;
      }
      if (so.flags & 16u) {
        q.status = PP_ERR_NONFINITE;
        q.live = false;
        return;
      }
      if (so.flags & 1u) {  // update_node_with_analystic_expansion returned Some
        q.found = true;
        q.final_parent = q.curr;
        q.final_rs_idx = so.path_idx;
        q.final_rs_cost = so.cost;
        q.live = false;
        return;
      }
      const size_t base = k * NP;
      const uint32_t checks = plan.prm.nlist / plan.prm.check_step + 1;
      for (uint32_t i = 0; i < NP; ++i) {
        ++q.st.rollouts;
        const uint8_t vf = v_valid[base + i];
        if (vf & 8u) q.st.poses_checked += checks;
        if (vf & 2u) ++q.st.ambiguous;
        if (vf & 4u) {
          q.status = PP_ERR_NONFINITE;
          q.live = false;
          return;
        }
        if (!(vf & 1u)) continue;
        const uint64_t key = calc_index(v_x[base + i], v_y[base + i], v_yaw[base + i], P);
        if (q.closed.count(key)) continue;
        const double cost = v_c[base + i];
        auto it = q.open.find(key);
        if (it != q.open.end() && !(q.nodes[it->second].cost > cost)) continue;
        const double f = v_f[base + i];
        if (f != f) {  // NotNan::new(..).unwrap()
          q.status = PP_ERR_NAN_COST;
          q.live = false;
          return;
        }
        SNode n;
        n.xind = v_x[base + i];
        n.yind = v_y[base + i];
        n.yawind = v_yaw[base + i];
        n.direction = mot[NP + i] > 0.0 ? 1 : -1;
        n.x = v_ex[base + i];
        n.y = v_ey[base + i];
        n.yaw = v_eyaw[base + i];
        n.steer = mot[i];
        n.cost = cost;
        n.pind = (int64_t)q.curr_key;
        n.parent = q.curr;
        n.prim = (int32_t)i;
        q.nodes.push_back(n);
        int32_t id = (int32_t)q.nodes.size() - 1;
        q.heap.push({f, key});
        if (it == q.open.end())
          q.open.emplace(key, id);
        else
          it->second = id;
      }
    });
  }
  const double t_search = now_ms();

  // ---- extract_path (hybrid_astar.rs:218-250) for the queries that found one
  int32_t ret = PP_OK;
  for (uint64_t i = 0; i < nq; ++i) {
    Query& q = Q[i];
    pp_path& po = out[i];
    po.len = 0;
    po.cost = 0.0;
    if (q.found) {
      // chain: final RS node <- n_curr <- ... <- start
      std::vector<int32_t> chain;
      for (int32_t id = q.final_parent; id >= 0; id = q.nodes[id].parent) chain.push_back(id);
      // chain.back() is the start node
      const SNode& fp = q.nodes[q.final_parent];
      double qq[6] = {fp.x, fp.y, fp.yaw, q.goal[0], q.goal[1], q.goal[2]};
      int32_t np = 0;
      pp_rs_word words[PP_RS_MAX_PATHS];
      uint64_t toff[PP_RS_MAX_PATHS + 1];
      size_t cap_rs = 1 << 16;
      std::vector<double> rx(cap_rs), ry(cap_rs), ryaw(cap_rs);
      std::vector<int64_t> rdir(cap_rs);
      int32_t s2 = pp_rs_single(ctx, qq, maxc, conf->move_step, q.final_rs_idx, &np, words, toff,
                                rx.data(), ry.data(), ryaw.data(), rdir.data(), cap_rs, nullptr);
      if (s2 != PP_OK) {
        q.status = s2;
      } else {
        uint64_t a = toff[q.final_rs_idx], b = toff[q.final_rs_idx + 1];
        q.st.rs_final_length = words[q.final_rs_idx].total_length;
        // The reference walks `while node.pind >= 0` with pind an isize copy of the parent's
        // key (:226, :372): a key that wrapped below zero (yaw index under minyaw, the rollout
        // never wraps yaw) ends the walk early and the start point is appended regardless.
        const bool with_final =
            (int64_t)calc_index(fp.xind, fp.yind, fp.yawind, P) >= 0;
        uint32_t nroll = 0, nlist = 0;
        if (with_final)
          while (nroll + 1 < chain.size() && q.nodes[chain[nroll]].pind >= 0) ++nroll;
        std::vector<double> in5((size_t)nroll * 5);
        for (uint32_t k = 0; k < nroll; ++k) {
          const SNode& n = q.nodes[chain[k]];
          const SNode& pa = q.nodes[n.parent];
          in5[k * 5 + 0] = pa.x;
          in5[k * 5 + 1] = pa.y;
          in5[k * 5 + 2] = pa.yaw;
          in5[k * 5 + 3] = n.steer;
          in5[k * 5 + 4] = plan.motion[NP + n.prim];
        }
        uint32_t pts = plan.prm.nlist + 1;
        std::vector<double> lx((size_t)nroll * pts), ly(lx.size()), lyaw(lx.size());
        s2 = pp_rollout_points_batch(ctx, conf, nroll, in5.data(), lx.data(), ly.data(), lyaw.data(), &nlist);
        if (s2 != PP_OK) {
          q.status = s2;
        } else {
          // reversed accumulation exactly as the reference builds it, then one final reverse
          std::vector<double> X, Y, YAW;
          std::vector<int64_t> D;
          double cost = 0.0;
          if (with_final) {
            for (uint64_t j = b; j-- > a + 1;) {  // node.x[1..].rev() of the RS node
              X.push_back(rx[j]);
              Y.push_back(ry[j]);
              YAW.push_back(ryaw[j]);
              D.push_back(rdir[j]);
            }
            cost += fp.cost + q.final_rs_cost;  // fcost = n_curr.cost + rs cost (:357)
          }
          for (uint32_t k = 0; k < nroll; ++k) {
            const SNode& n = q.nodes[chain[k]];
            for (uint32_t j = pts; j-- > 1;) {
              X.push_back(lx[(size_t)k * pts + j]);
              Y.push_back(ly[(size_t)k * pts + j]);
              YAW.push_back(lyaw[(size_t)k * pts + j]);
              D.push_back(n.direction);
            }
            cost += n.cost;
          }
          const SNode& ns = q.nodes[chain.back()];
          X.push_back(ns.x);
          Y.push_back(ns.y);
          YAW.push_back(ns.yaw);
          D.push_back(1);
          cost += ns.cost;
          po.len = X.size();
          po.cost = cost;
          if (po.cap < X.size() || !po.x || !po.y || !po.yaw || !po.direction) {
            q.status = PP_ERR_CAPACITY;
          } else {
            size_t L = X.size();
            for (size_t j = 0; j < L; ++j) {
              po.x[j] = X[L - 1 - j];
              po.y[j] = Y[L - 1 - j];
              po.yaw[j] = YAW[L - 1 - j];
              po.direction[j] = D[L - 1 - j];
            }
            q.status = PP_OK;
          }
        }
      }
    }
    q.st.t_heuristic_ms = t_heur - t_begin;
    q.st.t_search_ms = t_search - t_heur;
    if (stats) {
      uint64_t cc = stats[i].closed_cap;
      uint64_t* co = stats[i].closed_order;
      stats[i] = q.st;
      stats[i].closed_cap = cc;
      stats[i].closed_order = co;
    }
    if (status) status[i] = q.status;
    if (q.status < 0 && ret == PP_OK && !status) ret = q.status;  // per-query codes go to status[]
  }
  return ret;
}

extern "C" int32_t pp_plan(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                           const double start[3], const double goal[3], pp_path* out,
                           pp_plan_stats* stats) {
  if (!start || !goal || !out) return PP_ERR_INVALID;
  int32_t status = PP_OK;
  int32_t st = pp_plan_batch(ctx, map, conf, 1, start, goal, out, stats, &status);
  if (st != PP_OK) return st;
  return status;
}
