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
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <unordered_map>

#include "pp_expand_device.h"
#include "pp_internal.h"
#include "pp_pool.h"
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

// calc_index key -> the query's node of that cell, for the open and the closed set alike
// (hybrid_astar.rs:147-149 keeps two hash maps).  A rollout is looked up once: closed -> dropped
// (:187), open -> kept only if cheaper than the open node (:199), whose cost sits in the slot so the
// node itself is not touched.  Closing a node (:161-175) flips its slot in place, so nothing is
// ever erased.  Open addressing with linear probing over 32-byte slots, two to a cache line.
class NodeTable {
 public:
  enum : int32_t { kOpen = 1, kClosed = 2 };
  struct Slot {
    uint64_t key;
    double cost;    // of the open node
    int32_t node;   // its index in the query's node store
    uint32_t tag;   // 4 * generation + kOpen / kClosed; a slot of another generation is empty
    int32_t state() const { return (int32_t)(tag & 3u); }
  };
  void clear() {  // keeps the table: its slots become empty by belonging to an older generation
    if (++gen_ == (1u << 30)) {
      for (Slot& s : tab_) s.tag = 0;
      gen_ = 1;
    }
    used_ = open_ = 0;
  }
  size_t open_count() const { return open_; }
  size_t size() const { return used_; }  // open and closed
  template <class F>
  void for_each(F&& f) const {
    for (const Slot& s : tab_)
      if (live(s)) f(s);
  }
  Slot* find(uint64_t key) {
    if (cap_ == 0) return nullptr;
    for (size_t i = hash(key) & (cap_ - 1);; i = (i + 1) & (cap_ - 1)) {
      Slot& s = tab_[i];
      if (!live(s)) return nullptr;
      if (s.key == key) return &s;
    }
  }
  // Asks for the cache line a lookup of `key` starts at (a batch of thousands of queries finds
  // none of its tables in cache when its turn comes).
  void prefetch(uint64_t key) const {
    if (cap_ == 0) return;
    __builtin_prefetch(&tab_[hash(key) & (cap_ - 1)]);
  }
  // A key that is not in the table enters the open set.  May move the table: slots found earlier
  // are no longer valid.
  void insert_open(uint64_t key, int32_t node, double cost) {
    if ((used_ + 1) * 2 > cap_) grow();
    size_t i = hash(key) & (cap_ - 1);
    while (live(tab_[i])) i = (i + 1) & (cap_ - 1);
    tab_[i] = Slot{key, cost, node, gen_ * 4u + (uint32_t)kOpen};
    ++used_;
    ++open_;
  }
  void close(Slot* s) {
    s->tag = gen_ * 4u + (uint32_t)kClosed;
    --open_;
  }

 private:
  bool live(const Slot& s) const { return (s.tag >> 2) == gen_; }
  static size_t hash(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (size_t)k;
  }
  void grow() {
    std::vector<Slot> old;
    old.swap(tab_);
    cap_ = std::max<size_t>(64, cap_ * 2);
    tab_.assign(cap_, Slot{0, 0.0, 0, 0u});
    for (const Slot& o : old)
      if (live(o)) {
        size_t j = hash(o.key) & (cap_ - 1);
        while (live(tab_[j])) j = (j + 1) & (cap_ - 1);
        tab_[j] = o;
      }
  }
  std::vector<Slot> tab_;
  size_t cap_ = 0, used_ = 0, open_ = 0;
  uint32_t gen_ = 1;
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

// The open list's priority queue (hybrid_astar.rs:150-160): pop order = HeapLess's top, a total
// order on (f, key), so any heap gives the reference's sequence.  Four children per node, stored so
// that the four share one 64-byte line: a pop on a heap that has left the cache (thousands of
// queries take turns) walks half as many lines as a binary heap.
class OpenHeap {
 public:
  OpenHeap() = default;
  OpenHeap(const OpenHeap&) = delete;
  OpenHeap& operator=(const OpenHeap&) = delete;
  OpenHeap(OpenHeap&& o) noexcept : a_(o.a_), n_(o.n_), cap_(o.cap_) { o.a_ = nullptr, o.n_ = o.cap_ = 0; }
  OpenHeap& operator=(OpenHeap&& o) noexcept {
    if (this != &o) {
      std::free(a_);
      a_ = o.a_, n_ = o.n_, cap_ = o.cap_;
      o.a_ = nullptr, o.n_ = o.cap_ = 0;
    }
    return *this;
  }
  ~OpenHeap() { std::free(a_); }
  void clear() { n_ = 0; }  // keeps the storage
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  const HeapEnt& operator[](size_t i) const { return a_[i + kPad]; }  // i < size(), heap order
  void push(HeapEnt e) {
    if (n_ + kPad >= cap_) grow();
    size_t i = n_++;
    while (i > 0) {
      const size_t p = (i - 1) / 4;
      if (!before(e, at(p))) break;
      at(i) = at(p);
      i = p;
    }
    at(i) = e;
  }
  // The lines a pop and the pushes after it start with: the first three levels and the last entry
  // with its parent (asked for when a query's turn begins, used a few hundred cycles later).
  void prefetch_ends() const {
    if (!a_) return;
    for (size_t o = 0; o < 24 && o < n_ + kPad; o += 4) __builtin_prefetch(a_ + o);
    __builtin_prefetch(a_ + n_ + kPad);
    if (n_) __builtin_prefetch(a_ + (n_ - 1) / 4 + kPad);
  }
  HeapEnt pop() {  // size() > 0
    const HeapEnt top = at(0), e = at(--n_);
    size_t i = 0;
    for (;;) {
      const size_t c0 = 4 * i + 1;
      if (c0 >= n_) break;
      // the sixteen grandchildren are contiguous (four lines): whichever child wins, its children
      // are on their way while the four children are compared
      const size_t g0 = 4 * c0 + 1;
      if (g0 < n_)
        for (size_t o = 0; o < 16; o += 4) __builtin_prefetch(a_ + g0 + kPad + o);
      size_t m = c0;
      const size_t ce = std::min(c0 + 4, n_);
      for (size_t c = c0 + 1; c < ce; ++c)
        if (before(at(c), at(m))) m = c;
      if (!before(at(m), e)) break;
      at(i) = at(m);
      i = m;
    }
    if (n_) at(i) = e;
    return top;
  }

 private:
  static constexpr size_t kPad = 3;  // entry k sits in slot k + 3: the children 4i+1 .. 4i+4 fill line i + 1
  static bool before(const HeapEnt& a, const HeapEnt& b) { return a.f < b.f || (a.f == b.f && a.key > b.key); }
  HeapEnt& at(size_t i) { return a_[i + kPad]; }
  void grow() {
    const size_t cap = cap_ ? cap_ * 2 : 256;
    HeapEnt* a = (HeapEnt*)std::aligned_alloc(64, cap * sizeof(HeapEnt));
    if (!a) throw std::bad_alloc();
    if (a_) std::memcpy((void*)a, (const void*)a_, (n_ + kPad) * sizeof(HeapEnt));
    std::free(a_);
    a_ = a;
    cap_ = cap;
  }
  HeapEnt* a_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// What the device returns for one node: the analytic shot from its end pose and its P primitive
// rollouts.  It is a function of the node alone, so it can be fetched before the node is closed.
struct Expansion {
  PpShotOut shot;
  std::unique_ptr<uint64_t[]> words;  // P records of 8 (PpExpandDev::rec), those of the valid primitives filled in,
                                      // and behind them the P valid bytes
  const uint8_t* valid(uint32_t np) const { return (const uint8_t*)(words.get() + 8 * (size_t)np); }
  uint8_t* valid(uint32_t np) { return (uint8_t*)(words.get() + 8 * (size_t)np); }
};

// A query's expansions fetched ahead of their node's turn, by node id.  A few hundred entries at
// most, filled and emptied every round: entries live in a slab that is reused (an entry's record
// block is allocated once), found through a small open-addressing table.
class ExpansionCache {
 public:
  size_t size() const { return n_; }
  void clear() {
    if (n_ == 0) return;
    std::fill(tab_.begin(), tab_.end(), Ent{-1, -1});
    free_.clear();
    for (int32_t s = (int32_t)slab_.size(); s-- > 0;) free_.push_back(s);
    n_ = 0;
  }
  Expansion* find(int32_t id) {
    if (n_ == 0) return nullptr;
    for (size_t i = home(id);; i = (i + 1) & (tab_.size() - 1)) {
      if (tab_[i].id == id) return &slab_[tab_[i].slot];
      if (tab_[i].id < 0) return nullptr;
    }
  }
  // the entry for `id` (not present so far); np = primitives per node
  Expansion& insert(int32_t id, uint32_t np) {
    if (np != np_) {  // another motion set than the slab was built for (between batches: the cache is empty)
      slab_.clear();
      free_.clear();
      std::fill(tab_.begin(), tab_.end(), Ent{-1, -1});
      n_ = 0;
      np_ = np;
    }
    if (tab_.empty() || 2 * (n_ + 1) > tab_.size()) grow();
    int32_t slot;
    if (free_.empty()) {
      slot = (int32_t)slab_.size();
      slab_.emplace_back();
      slab_.back().words.reset(new uint64_t[8 * (size_t)np + (np + 7) / 8]);
    } else {
      slot = free_.back();
      free_.pop_back();
    }
    size_t i = home(id);
    while (tab_[i].id >= 0) i = (i + 1) & (tab_.size() - 1);
    tab_[i] = Ent{id, slot};
    ++n_;
    return slab_[slot];
  }
  void erase(int32_t id) {
    if (n_ == 0) return;
    const size_t mask = tab_.size() - 1;
    size_t i = home(id);
    while (tab_[i].id != id) {
      if (tab_[i].id < 0) return;
      i = (i + 1) & mask;
    }
    free_.push_back(tab_[i].slot);
    --n_;
    // close the gap: move back the entries of the run behind it that may sit closer to their home
    for (size_t j = (i + 1) & mask; tab_[j].id >= 0; j = (j + 1) & mask) {
      const size_t h = home(tab_[j].id);
      if (((j - h) & mask) >= ((j - i) & mask)) {
        tab_[i] = tab_[j];
        i = j;
      }
    }
    tab_[i] = Ent{-1, -1};
  }

 private:
  struct Ent {
    int32_t id, slot;
  };
  size_t home(int32_t id) const { return ((uint32_t)id * 2654435761u >> 7) & (tab_.size() - 1); }
  void grow() {
    std::vector<Ent> old;
    old.swap(tab_);
    tab_.assign(old.empty() ? 64 : 2 * old.size(), Ent{-1, -1});
    for (const Ent& e : old)
      if (e.id >= 0) {
        size_t i = home(e.id);
        while (tab_[i].id >= 0) i = (i + 1) & (tab_.size() - 1);
        tab_[i] = e;
      }
  }
  std::vector<Ent> tab_;
  std::vector<Expansion> slab_;  // an entry's address is only used until the next insert
  std::vector<int32_t> free_;
  size_t n_ = 0;
  uint32_t np_ = 0;
};
// The same data in place, in the round's staging buffer or in a cache entry.
struct ExpansionView {
  const PpShotOut* shot;
  const uint8_t* valid;
  const uint64_t* rec;  // primitive i's record at rec + 8 i, meaningful where valid[i] has bit 0
  uint64_t xind(uint32_t i) const { return rec[8 * i]; }
  uint64_t yind(uint32_t i) const { return rec[8 * i + 1]; }
  int64_t yawind(uint32_t i) const { return (int64_t)rec[8 * i + 2]; }
  double cost(uint32_t i) const { return f64(rec[8 * i + 3]); }
  double ex(uint32_t i) const { return f64(rec[8 * i + 4]); }
  double ey(uint32_t i) const { return f64(rec[8 * i + 5]); }
  double eyaw(uint32_t i) const { return f64(rec[8 * i + 6]); }
  double f(uint32_t i) const { return f64(rec[8 * i + 7]); }
  static double f64(uint64_t b) {
    double v;
    std::memcpy(&v, &b, 8);
    return v;
  }
};

struct Query {
  double start[3], goal[3];
  int32_t hmap_slot = -1;
  std::vector<SNode> nodes;
  NodeTable tab;  // open and closed set
  OpenHeap heap;
  ExpansionCache cache;  // expansions fetched ahead of their node's turn
  std::vector<int32_t> want;                     // nodes to expand in this round, want[0] = curr
  int32_t status = PP_OK;
  bool live = false;
  int32_t ftab = -1;               // the query's table of known cells on the device (pp_expand.h), -1 = none
  std::vector<PpFtabUpd> upd;      // what that table has not been told yet
  int32_t curr = -1;       // node closed last and not yet expanded
  uint64_t curr_key = 0;
  // outcome
  bool found = false;
  int32_t final_parent = -1;
  int32_t final_rs_idx = -1;
  double final_rs_cost = 0.0;
  // Back to the state of a fresh query; the containers keep what they have allocated.
  void reset() {
    hmap_slot = -1;
    nodes.clear();
    tab.clear();
    heap.clear();
    cache.clear();
    want.clear();
    status = PP_OK;
    live = false;
    ftab = -1;
    upd.clear();
    curr = -1;
    curr_key = 0;
    found = false;
    final_parent = final_rs_idx = -1;
    final_rs_cost = 0.0;
  }
  pp_plan_stats st{};
};

// calc_index, hybrid_astar.rs:516-520 (release-mode wrapping arithmetic)
inline uint64_t calc_index(uint64_t xind, uint64_t yind, int64_t yawind, const pp_param& P) {
  return (uint64_t)(yawind - P.minyaw) * P.xw * P.yw + (yind - P.miny) * P.xw + (xind - P.minx);
}

// What a context keeps between batches: the query objects and the worker threads.
struct PlanState {
  std::vector<Query> queries;
  std::unique_ptr<Pool> pool;
  unsigned pool_workers = 0;
  std::unique_ptr<Pool> shared_pool;  // the workers the parts of a large batch share (pp_plan_batch)
  unsigned shared_pool_asked = 0;     // the size it was created for (it may have got fewer threads)
  hipEvent_t lane_done[2] = {nullptr, nullptr};  // a lane's results have reached the pinned area
  hipStream_t lane_stream = nullptr;             // the second lane's launches (the first lane uses the context's stream)
  ~PlanState() {
    for (hipEvent_t e : lane_done)
      if (e) (void)hipEventDestroy(e);
    if (lane_stream) (void)hipStreamDestroy(lane_stream);
  }
};

static PlanState& plan_state_of(pp_ctx* ctx) {
  if (!ctx->plan_state) {
    ctx->plan_state = new PlanState();
    ctx->plan_state_free = [](void* p) { delete (PlanState*)p; };
  }
  return *(PlanState*)ctx->plan_state;
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }


__global__ void k_gather_hmap(uint32_t n, const double* const* __restrict__ tab,
                              const int32_t* __restrict__ slot, const uint64_t* __restrict__ off,
                              double* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tab[slot[i]][off[i]];
}

__global__ void k_gather_blocked(uint32_t n, PpBlockedGrids g, uint64_t hyw, const int32_t* __restrict__ slot,
                                 const uint64_t* __restrict__ off, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t hx = off[i] / hyw, hy = off[i] - hx * hyw;
    out[i] = pp_blocked_value(g.blk, g.stamp, g.nty, (uint64_t)slot[i] * g.plane_cols + hx, hy);
  }
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

static int32_t plan_batch_rounds(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t nq,
                                 const double* starts, const double* goals, pp_path* out,
                                 pp_plan_stats* stats, int32_t* status);

// One batch on one context.  Nothing may unwind through the C boundary or out of a part's thread:
// an exception from the host side of the rounds (in practice: no memory for a query's containers)
// ends the batch with PP_ERR_UNSUPPORTED once the device has finished what was in flight.
static int32_t plan_batch_core(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t nq,
                               const double* starts, const double* goals, pp_path* out,
                               pp_plan_stats* stats, int32_t* status) {
  try {
    return plan_batch_rounds(ctx, map, conf, nq, starts, goals, out, stats, status);
  } catch (...) {
    if (ctx) {
      (void)hipSetDevice(ctx->device);
      (void)hipDeviceSynchronize();  // kernels of either lane may still be using the staging buffers
      // the query objects may be half-built: the next batch starts from new ones (the rest of the
      // state stays -- the other parts of a batch may be using the worker pool it holds)
      if (ctx->plan_state) ((PlanState*)ctx->plan_state)->queries.clear();
    }
    return PP_ERR_UNSUPPORTED;
  }
}

// The lock-step rounds described at the top of the file.
static int32_t plan_batch_rounds(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t nq,
                                 const double* starts, const double* goals, pp_path* out,
                                 pp_plan_stats* stats, int32_t* status) {
  const double t_entry = now_ms();
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
  // the query objects live in the context between calls: thousands of small containers are
  // cheaper to clear than to free and allocate again
  PlanState& state = plan_state_of(ctx);
  std::vector<Query>& Q = state.queries;
  if (Q.size() < nq) Q.resize(nq);
  std::unordered_map<uint64_t, int32_t> goal_slot;
  std::vector<std::pair<uint64_t, uint64_t>> goal_cells;
  for (uint64_t i = 0; i < nq; ++i) {
    Query& q = Q[i];
    q.reset();
    std::memcpy(q.start, starts + 3 * i, 24);
    std::memcpy(q.goal, goals + 3 * i, 24);
    q.st = pp_plan_stats{};
    q.st.first_ambiguous = UINT64_MAX;
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
  double h_limit = __builtin_inf();
  bool h_bounded = false, keep_blocked = false;
  PpBlockedGrids blocked;
  // heuristic grids, raster and pointer table share one buffer kept in the context
  struct {
    void* p = nullptr;
  } hm_buf, obs_buf, tab_buf;
  std::vector<double*> hm_ptr(ngoals);
  if (ngoals) {
    // Relaxed in one pass (the usual case), the grids stay in the relaxation's blocked working array
    // and the searches read them there; otherwise they are written out as dense arrays.
    keep_blocked = pp_wavefront_multi_single_pass(ctx, hxw, hyw, (uint32_t)ngoals);
    const size_t b_hm = keep_blocked ? 0 : ngoals * al256(hcells * 8), b_obs = al256(hcells),
                 b_tab = al256(ngoals * sizeof(double*));
    if (b_hm + b_obs + b_tab > ctx->hmap_bytes) {
      if (ctx->hmap_buf) PP_HIP(ctx, hipFree(ctx->hmap_buf));
      ctx->hmap_buf = nullptr;
      ctx->hmap_bytes = 0;
      PP_HIP(ctx, hipMalloc(&ctx->hmap_buf, b_hm + b_obs + b_tab));
      ctx->hmap_bytes = b_hm + b_obs + b_tab;
    }
    hm_buf.p = ctx->hmap_buf;
    obs_buf.p = (char*)ctx->hmap_buf + b_hm;
    tab_buf.p = (char*)ctx->hmap_buf + b_hm + b_obs;
    st = pp_obsmap_dev(ctx, map, xyreso, 1.0, dims, (uint8_t*)obs_buf.p);
    if (st != PP_OK) return st;
    std::vector<uint64_t> ggx(ngoals), ggy(ngoals);
    for (size_t g = 0; g < ngoals; ++g) {
      hm_ptr[g] = (double*)((char*)hm_buf.p + g * al256(hcells * 8));
      ggx[g] = goal_cells[g].first;
      ggy[g] = goal_cells[g].second;
    }
    // The searches only look at the grids near their queries: the relaxation stops at twice the
    // longest start-goal distance (in cells) plus a margin, where every value up to that limit is
    // final.  Should a search meet a value beyond it, the grids are completed then (below).
    double far = 0.0;
    for (uint64_t i = 0; i < nq; ++i)
      if (Q[i].live) far = std::max(far, std::hypot(Q[i].start[0] - Q[i].goal[0], Q[i].start[1] - Q[i].goal[1]));
    h_limit = 2.0 * far / xyreso + 32.0;
    if (ctx->heuristic_limit >= 0.0) h_limit = ctx->heuristic_limit;  // the caller's choice: any value is safe
    if (!(h_limit >= 0.0)) h_limit = __builtin_inf();
    st = pp_wavefront_multi_dev(ctx, hxw, hyw, (const uint8_t*)obs_buf.p, (uint32_t)ngoals, ggx.data(),
                                ggy.data(), keep_blocked ? nullptr : (double*)hm_buf.p, al256(hcells * 8) / 8,
                                h_limit, &h_bounded, keep_blocked ? &blocked : nullptr);
    if (st != PP_OK) return st;
    if (!keep_blocked) {
      PP_HIP(ctx, hipMemcpyAsync(tab_buf.p, hm_ptr.data(), ngoals * sizeof(double*), hipMemcpyHostToDevice, s));
      PP_HIP(ctx, hipStreamSynchronize(s));
    }
  }
  const double t_heur = now_ms();

  // ---- start nodes (hybrid_astar.rs:103-154)
  {
    std::vector<SNode> sn(nq);
    std::vector<int32_t> g_slot;
    std::vector<uint64_t> g_off;
    std::vector<uint32_t> g_q;
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
      sn[i] = n;
      g_slot.push_back(q.hmap_slot);
      g_off.push_back(hx * hyw + hy);
      g_q.push_back((uint32_t)i);
    }
    // hmap values under the start cells: one gather for all queries
    const size_t ng = g_q.size();
    std::vector<double> hval(ng);
    const size_t b4 = al256(ng * 4), b8 = al256(ng * 8);
    // slots and offsets are already in the scratch buffer; read the grids where they live
    auto gather_start_values = [&]() -> int32_t {
      char* b = (char*)ctx->scratch;
      const dim3 grid((unsigned)((ng + 255) / 256));
      if (keep_blocked)
        hipLaunchKernelGGL(k_gather_blocked, grid, dim3(256), 0, s, (uint32_t)ng, blocked, hyw, (const int32_t*)b,
                           (const uint64_t*)(b + b4), (double*)(b + b4 + b8));
      else
        hipLaunchKernelGGL(k_gather_hmap, grid, dim3(256), 0, s, (uint32_t)ng, (const double* const*)tab_buf.p,
                           (const int32_t*)b, (const uint64_t*)(b + b4), (double*)(b + b4 + b8));
      PP_HIP(ctx, hipGetLastError());
      PP_HIP(ctx, hipMemcpyAsync(hval.data(), b + b4 + b8, ng * 8, hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipStreamSynchronize(s));
      return PP_OK;
    };
    if (ng) {
      st = pp_scratch_reserve(ctx, b4 + 2 * b8);
      if (st != PP_OK) return st;
      char* b = (char*)ctx->scratch;
      PP_HIP(ctx, hipMemcpyAsync(b, g_slot.data(), ng * 4, hipMemcpyHostToDevice, s));
      PP_HIP(ctx, hipMemcpyAsync(b + b4, g_off.data(), ng * 8, hipMemcpyHostToDevice, s));
      st = gather_start_values();
      if (st != PP_OK) return st;
    }
    if (h_bounded) {
      // a start cell beyond the limit (blocked cells included: the host does not hold the raster)
      bool beyond = false;
      for (size_t k = 0; k < ng; ++k) beyond |= hval[k] > h_limit;
      if (beyond) {
        st = pp_wavefront_multi_finish(ctx);
        if (st != PP_OK) return st;
        h_bounded = false;
        st = gather_start_values();
        if (st != PP_OK) return st;
      }
    }
    for (size_t k = 0; k < ng; ++k) {
      Query& q = Q[g_q[k]];
      const SNode& n = sn[g_q[k]];
      double f = n.cost + conf->h_cost * hval[k];
      if (f != f) {
        q.status = PP_ERR_NAN_COST;
        q.live = false;
        continue;
      }
      uint64_t key = calc_index(n.xind, n.yind, n.yawind, P);
      q.nodes.push_back(n);
      q.tab.insert_open(key, 0, n.cost);
      q.heap.push({f, key});
    }
  }

  // ---- device / pinned staging for the rounds
  // a round expands one node per live query, plus nodes near the top of the open list while few
  // queries are left (kSpecItems in all, at most kSpecDepth per query)
#ifndef PP_SPEC_ITEMS
#define PP_SPEC_ITEMS 128
#endif
#ifndef PP_SPEC_WALK
#define PP_SPEC_WALK 3
#endif
#ifndef PP_SPEC_DEPTH
#define PP_SPEC_DEPTH 8
#endif
  constexpr size_t kSpecItems = PP_SPEC_ITEMS, kSpecDepth = PP_SPEC_DEPTH, kCacheMax = 256;
  constexpr size_t kRoundFusedMax = 1024;  // nodes; below this a round is one launch
  constexpr size_t kRoundDirectMax = 128;  // nodes; below this the kernels work on the pinned staging area
#ifndef PP_ROUND_OUT_HOST_MIN
#define PP_ROUND_OUT_HOST_MIN 128
#endif
  constexpr size_t kRoundOutHostMin = PP_ROUND_OUT_HOST_MIN;  // nodes; from this on results go straight to it
  // a large batch runs as two lanes (below), each with its own half of the staging buffers
  constexpr size_t kTwoLanesMin = 1024;
  // a lane's staging area holds Bl nodes (below): a round with look-ahead sends up to kSpecItems
  static_assert(kSpecItems + kSpecDepth <= kTwoLanesMin / 2, "the look-ahead must fit a lane's staging area");
  const bool two_lanes = nq >= kTwoLanesMin;
  const size_t Bl = two_lanes ? (nq + 1) / 2 : std::max<size_t>(nq, kSpecItems + kSpecDepth);  // items per lane
  const size_t B = two_lanes ? 2 * Bl : Bl;
  const size_t items = Bl * NP;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += al256(bytes);
    return o;
  };
  // per-round inputs (start pose + goal pose, pdir, psteer, pcost, hmap_id) and outputs share one
  // device buffer and one pinned mirror, laid out per round; the motion table and the
  // Reeds-Shepp work arrays sit behind them
  const size_t io_lane = al256(6 * Bl * 8) + 3 * al256(Bl * 8) + 2 * al256(Bl * 4) + al256(Bl * sizeof(PpShotOut)) +
                         al256(items) + al256(items * 64);
  const size_t io_bytes = (two_lanes ? 2 : 1) * io_lane;
  // pinned only, behind the mirror: per lane, the news of a round for the tables of known cells
  const size_t upd_cap = Bl * (NP + 2);
  const size_t upd_lane = al256(upd_cap * sizeof(PpFtabUpd));
  off = io_bytes;
  const size_t o_mot = take(plan.motion.size() * 8);
  const size_t o_cand = take(B * PP_RS_MAX_PATHS * sizeof(PpRsCand)), o_pair = take(B * sizeof(PpRsPair));
  // both live in the context and are reused by the next batch of the same size
  if (off > ctx->plan_bytes) {
    if (ctx->plan_buf) PP_HIP(ctx, hipFree(ctx->plan_buf));
    ctx->plan_buf = nullptr;
    ctx->plan_bytes = 0;
    PP_HIP(ctx, hipMalloc(&ctx->plan_buf, off));
    ctx->plan_bytes = off;
  }
  st = pp_pinned_reserve(ctx, io_bytes + 2 * upd_lane);
  if (st != PP_OK) return st;
  char* d = (char*)ctx->plan_buf;
  char* h = (char*)ctx->pinned;
  const double* d_mot = (const double*)(d + o_mot);
  PP_HIP(ctx, hipMemcpyAsync(d + o_mot, plan.motion.data(), plan.motion.size() * 8, hipMemcpyHostToDevice, s));

  PpExpandDev eo{};
  eo.hmaps = (const double* const*)tab_buf.p;
  if (keep_blocked) {
    eo.hblk = blocked.blk;
    eo.hblk_stamp = blocked.stamp;
    eo.hblk_nty = blocked.nty;
    eo.hblk_plane_cols = blocked.plane_cols;
  }
  eo.hxw = hxw;
  eo.hyw = hyw;
  eo.h_cost = conf->h_cost;
  if (h_bounded) {
    eo.hobs = (const uint8_t*)obs_buf.p;
    eo.h_limit = h_limit;
  }
  // set by any query whose expansion met a heuristic value beyond the limit: that expansion is
  // not applied, the grids are completed after the round and the node is expanded again
  std::atomic<bool> need_grids{false};
  auto beyond_limit = [&](const ExpansionView& v) {  // the mark is only ever set by a launch on bounded grids
    for (uint32_t i = 0; i < NP; ++i)
      if ((v.valid[i] & 17u) == 17u) return true;
    return false;
  };

  unsigned hw = std::thread::hardware_concurrency();
  unsigned workers = nq >= 128 ? std::min((unsigned)ctx->host_threads, hw > 1 ? hw - 1 : 0u) : 0u;
  // the worker threads are started once per context (they sleep between batches); a small batch
  // runs on the calling thread alone
  // (the parts of a large batch share one set of workers, handed in by pp_plan_batch)
  Pool serial(0);
  Pool* const lent = nq >= 128 ? (Pool*)ctx->lent_pool : nullptr;
  if (lent) workers = 0;
  if (workers && (!state.pool || state.pool_workers != workers)) {
    state.pool.reset(new Pool(workers));
    state.pool_workers = workers;
  }
  Pool& pool = lent ? *lent : workers ? *state.pool : serial;
  for (hipEvent_t& e : state.lane_done)
    if (!e) PP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // what the device needs of the node a query wants expanded (one record per live query)
  struct Emit {
    double x, y, yaw, gx, gy, gyaw;
    int64_t dir;
    double steer, cost;
    int32_t hid, fid;
    uint8_t alive;
  };
  auto f64_bits = [](double v) {
    unsigned long long b;
    std::memcpy(&b, &v, 8);
    return b;
  };

  // ---- lock-step rounds
  const double* mot = plan.motion.data();
  const uint32_t checks = plan.prm.nlist / plan.prm.check_step + 1;
  // hybrid_astar.rs:176-209 for the node q.curr from its expansion
  auto apply = [&](Query& q, const ExpansionView& v) {
    const PpShotOut& so = *v.shot;
    q.st.rs_paths += (uint64_t)so.n_paths;
    q.st.poses_checked += so.poses_handed;
    // the node being expanded is the last one closed: position expansions - 1 of the closing order
    auto guard_band = [&q] {
      if (q.st.ambiguous++ == 0) q.st.first_ambiguous = q.st.expansions - 1;
    };
    if (so.flags & 2u) guard_band();
    if (so.flags & (4u | 8u | 16u)) {
      q.status = (so.flags & 4u) ? PP_ERR_DEGENERATE_RS : (so.flags & 8u) ? PP_ERR_NAN_COST : PP_ERR_NONFINITE;
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
    // where the pushes will write and the pop after them will start
    q.heap.prefetch_ends();
    __builtin_prefetch(q.nodes.data() + q.nodes.size());
    // keys of the valid children first, with their table lines requested, then the lookups
    uint64_t ckey[256];
    const bool keyed = NP <= 256;
    for (uint32_t i = 0; i < NP; ++i)
      if (v.valid[i] & 1u) __builtin_prefetch(v.rec + 8 * i);
    if (keyed)
      for (uint32_t i = 0; i < NP; ++i)
        if (v.valid[i] & 1u) {
          ckey[i] = calc_index(v.xind(i), v.yind(i), v.yawind(i), P);
          q.tab.prefetch(ckey[i]);
        }
    for (uint32_t i = 0; i < NP; ++i) {
      ++q.st.rollouts;
      const uint8_t vf = v.valid[i];
      if (vf & 8u) q.st.poses_checked += checks;
      if (vf & 2u) guard_band();
      if (vf & 4u) {
        q.status = PP_ERR_NONFINITE;
        q.live = false;
        return;
      }
      if (!(vf & 1u)) continue;
      const uint64_t key = keyed ? ckey[i] : calc_index(v.xind(i), v.yind(i), v.yawind(i), P);
      const double cost = v.cost(i);
      NodeTable::Slot* sl = q.tab.find(key);
      if (sl && (sl->state() == NodeTable::kClosed || !(sl->cost > cost))) continue;
      const double f = v.f(i);
      if (f != f) {  // NotNan::new(..).unwrap()
        q.status = PP_ERR_NAN_COST;
        q.live = false;
        return;
      }
      SNode n;
      n.xind = v.xind(i);
      n.yind = v.yind(i);
      n.yawind = v.yawind(i);
      n.direction = mot[NP + i] > 0.0 ? 1 : -1;
      n.x = v.ex(i);
      n.y = v.ey(i);
      n.yaw = v.eyaw(i);
      n.steer = mot[i];
      n.cost = cost;
      n.pind = (int64_t)q.curr_key;
      n.parent = q.curr;
      n.prim = (int32_t)i;
      q.nodes.push_back(n);
      const int32_t id = (int32_t)q.nodes.size() - 1;
      q.heap.push({f, key});
      if (!sl) {
        q.tab.insert_open(key, id, cost);
      } else {
        sl->node = id;
        sl->cost = cost;
      }
      if (q.ftab >= 0) q.upd.push_back(PpFtabUpd{key, f64_bits(cost), q.ftab, 0});
    }
  };
  auto view_of = [&](const Expansion& e) { return ExpansionView{&e.shot, e.valid(NP), e.words.get()}; };
  // Closes nodes (hybrid_astar.rs:157-161) for as long as their expansions are already at hand;
  // returns with q.curr = the node the device has to expand next, or with the query finished.
  auto advance = [&](Query& q) {
    if (!q.live) return;
    for (;;) {
      if (q.curr >= 0) {
        const Expansion* c = q.cache.find(q.curr);
        if (!c) return;
        if (beyond_limit(view_of(*c))) {  // fetched before the grids were completed
          q.cache.erase(q.curr);
          return;
        }
        apply(q, view_of(*c));  // (the entry stays put meanwhile: apply inserts nothing)
        q.cache.erase(q.curr);
        q.curr = -1;
        if (!q.live) return;
      }
      if (ctx->max_expansions && q.st.expansions >= ctx->max_expansions) {
        q.live = false;
        q.status = PP_ERR_LIMIT;
        return;
      }
      while (q.tab.open_count() != 0) {
        const HeapEnt e = q.heap.pop();
        NodeTable::Slot* sl = q.tab.find(e.key);
        if (!sl || sl->state() != NodeTable::kOpen) {
          ++q.st.stale_pops;
          continue;
        }
        q.curr = sl->node;
        q.curr_key = e.key;
        q.tab.close(sl);
        if (q.ftab >= 0) q.upd.push_back(PpFtabUpd{e.key, 0ull, q.ftab, 0});
        if (q.st.closed_order && q.st.expansions < q.st.closed_cap) q.st.closed_order[q.st.expansions] = e.key;
        ++q.st.expansions;
        break;
      }
      if (q.curr < 0) {  // open set exhausted: None
        q.live = false;
        q.status = PP_NO_PATH;
        return;
      }
    }
  };
  // Open nodes likely to be closed soon: the entries the next pops return if nothing cheaper is
  // pushed meanwhile, found by walking the heap down from its root (the best entry not yet taken is
  // always a child of one that was).
  auto speculate = [&](Query& q, size_t depth) {
    size_t frontier[3 * PP_SPEC_WALK * kSpecDepth + 8], nf = 0;  // a step takes one entry off and adds up to four
    if (q.heap.size()) frontier[nf++] = 0;
    // (a walk of three times the nodes a query may send: whatever lies deeper is several rounds away,
    // and a walk over entries that are all fetched already is the most expensive kind)
    for (size_t step = 0; step < PP_SPEC_WALK * depth && nf && q.want.size() < depth; ++step) {
      size_t b = 0;
      for (size_t j = 1; j < nf; ++j)
        if (HeapLess()(q.heap[frontier[b]], q.heap[frontier[j]])) b = j;
      const size_t i = frontier[b];
      frontier[b] = frontier[--nf];
      for (size_t c = 4 * i + 1; c <= 4 * i + 4 && c < q.heap.size(); ++c) frontier[nf++] = c;
      const NodeTable::Slot* sl = q.tab.find(q.heap[i].key);
      if (!sl || sl->state() != NodeTable::kOpen) continue;  // a stale entry
      const int32_t id = sl->node;
      bool dup = q.cache.find(id) != nullptr;
      for (size_t j = 1; j < q.want.size() && !dup; ++j) dup = q.want[j] == id;  // a key's older entries name the same node
      if (!dup) q.want.push_back(id);
    }
  };

  const bool prof = getenv("PP_PLAN_PROFILE") != nullptr;
  double p_adv[2] = {0, 0}, p_stage[2] = {0, 0}, p_dev[2] = {0, 0}, p_apply[2] = {0, 0};
  uint64_t p_rounds[2] = {0, 0}, p_items[2] = {0, 0}, p_live[2] = {0, 0};
  uint64_t c_rounds = 0, c_small = 0, c_items = 0;  // always counted: pp_plan_batch_counters

  // A lane is a set of queries that go through their rounds together.  A large batch runs as two
  // lanes that take turns: while the device works on one lane's nodes, the host applies the other
  // lane's results, pops its next nodes and stages them.  Both use the one stream, so a lane's
  // launches queue up behind the other lane's and its event says when its own results are in.
  struct Lane {
    std::vector<uint32_t> ids;  // live queries
    // What the device needs of the nodes ids[k] wants expanded: `stride` records per query while
    // the queries write them (emit[k * stride] = the node to close next, behind it emit_n[k] - 1
    // nodes near the top of its open list), packed by launch() into one record per item, query k's
    // first at item0[k].
    std::vector<Emit> emit;
    std::vector<uint8_t> emit_n;
    std::vector<uint32_t> item0;
    size_t stride = 1;
    size_t io_base = 0;         // the lane's part of the staging buffers (pinned and device), bytes
    size_t item_base = 0;       // the lane's part of the Reeds-Shepp work arrays, items
    hipEvent_t done = nullptr;
    bool in_flight = false;
    int pc = 0;
    hipStream_t stream = nullptr;
    size_t nl = 0, ni = 0, depth = 1;
    size_t r_shot = 0, r_valid = 0, r_rec = 0;
    // news for the tables of known cells, written by the workers as they apply a round
    PpFtabUpd* upd = nullptr;
    std::atomic<uint32_t> upd_n{0}, upd_fail{UINT32_MAX};  // records reserved; lowest reservation that did not fit
    uint32_t rounds = 0;
    bool unassigned = true;  // some live query may still be without a table
  };
  const size_t nlanes = two_lanes ? 2 : 1;
  Lane lanes[2];
  // The second lane has a stream of its own: its kernels then run beside the first lane's instead of
  // behind them, so a lane waits one device latency for its results however many lanes are in flight.
  // (The lanes hold different queries: they share no staging area, work array or table.)
  if (two_lanes && !state.lane_stream)
    PP_HIP(ctx, hipStreamCreateWithFlags(&state.lane_stream, hipStreamNonBlocking));
  lanes[0].stream = lanes[1].stream = s;
  if (two_lanes) {
    lanes[1].stream = state.lane_stream;
    PP_HIP(ctx, hipStreamSynchronize(s));  // the grids and tables queued on the context's stream are in
  }
  // the launch helpers use the context's stream: it is the lane's for the length of a lane's launches
  struct StreamScope {
    pp_ctx* c;
    hipStream_t keep;
    StreamScope(pp_ctx* c_, hipStream_t now) : c(c_), keep(c_->stream) { c->stream = now; }
    ~StreamScope() { c->stream = keep; }
  };
  auto sync_lanes = [&]() -> hipError_t {
    hipError_t e = hipStreamSynchronize(s);
    if (two_lanes && e == hipSuccess) e = hipStreamSynchronize(state.lane_stream);
    return e;
  };
  for (size_t l = 0; l < nlanes; ++l) {
    lanes[l].io_base = l * io_lane;
    lanes[l].item_base = l * Bl;
    lanes[l].done = state.lane_done[l];
    lanes[l].upd = (PpFtabUpd*)(h + io_bytes + l * upd_lane);
  }
  for (uint64_t i = 0, k = 0; i < nq; ++i)
    if (Q[i].live) lanes[k++ % nlanes].ids.push_back((uint32_t)i);

  // (1) A query closes nodes until one needs the device, and leaves what the device needs of that
  // node in a flat record: the loops of a round that run on this thread alone, and the staging
  // of a large round, then never touch the query objects (thousands of them, a cache miss apiece).
  // After the first round this runs right behind the query's apply step, while its tables are
  // still in cache.
  auto advance_emit = [&](Lane& ln, size_t k) {
    Query& q = Q[ln.ids[k]];
    advance(q);
    if (const size_t nu = q.upd.size()) {
      const uint32_t at = ln.upd_n.fetch_add((uint32_t)nu, std::memory_order_relaxed);
      if (at + nu <= upd_cap) {
        std::memcpy((void*)(ln.upd + at), q.upd.data(), nu * sizeof(PpFtabUpd));
      } else {  // no room (a query ran through many expansions it had in its cache): the tables miss this news
        uint32_t lo = ln.upd_fail.load(std::memory_order_relaxed);
        while (at < lo && !ln.upd_fail.compare_exchange_weak(lo, at, std::memory_order_relaxed)) {
        }
      }
      q.upd.clear();
    }
    q.want.clear();
    q.want.push_back(q.curr);
    Emit* const e0 = &ln.emit[k * ln.stride];
    e0->alive = q.live;
    ln.emit_n[k] = 1;
    if (!q.live) return;
    // the look-ahead is chosen here too, while the query's heap and tables are in this thread's cache
    if (ln.stride > 1) speculate(q, ln.stride);
    ln.emit_n[k] = (uint8_t)q.want.size();
    for (size_t j = 0; j < q.want.size(); ++j) {
      const SNode& n = q.nodes[q.want[j]];
      Emit& e = e0[j];
      e.x = n.x;
      e.y = n.y;
      e.yaw = n.yaw;
      e.gx = q.goal[0];
      e.gy = q.goal[1];
      e.gyaw = q.goal[2];
      e.dir = n.direction;
      e.steer = n.steer;
      e.cost = n.cost;
      e.hid = q.hmap_slot;
      e.fid = q.ftab;
      e.alive = 1;
    }
  };

  // Tables of known cells (pp_expand.h).  A search that is still running after kFtabStartRound
  // rounds gets a table on the device, filled with what it has closed and holds open so far and
  // kept up to date round by round; from then on the device drops the rollouts the search would
  // drop anyway, which is most of them (on the search-heavy benchmark 82 % of the valid ones), and
  // the host never looks at them.  Short searches -- most of a typical batch -- never pay for it.
#ifndef PP_FTAB_START
#define PP_FTAB_START 64
#endif
  constexpr uint32_t kFtabStartRound = PP_FTAB_START;
#ifndef PP_FTAB_MIN_LIVE
#define PP_FTAB_MIN_LIVE 4
#endif
  constexpr size_t kFtabMinLive = PP_FTAB_MIN_LIVE;
  const size_t kFtabBudget = ctx->ftab_budget;  // bytes of device memory for all tables (pp_ctx_set_table_budget)
  // the cost bit patterns order like the costs only if no cost term can be negative
  bool ftab_off = !(plan.prm.backward_cost >= 0.0 && plan.prm.gear_cost >= 0.0 &&
                    plan.prm.steer_angle_cost >= 0.0 && plan.prm.steer_change_cost >= 0.0);
  uint64_t ftab_slots = 0;
  uint32_t ftab_tables = 0, ftab_next = 0;
  // the records the workers left in the lane's buffer; the buffer starts over
  auto take_updates = [&](Lane& ln) {
    const uint32_t n = std::min(ln.upd_n.load(std::memory_order_relaxed), ln.upd_fail.load(std::memory_order_relaxed));
    ln.upd_n.store(0, std::memory_order_relaxed);
    ln.upd_fail.store(UINT32_MAX, std::memory_order_relaxed);
    return n;
  };
  auto ftab_assign = [&](Lane& ln) -> int32_t {
    ln.unassigned = false;
    // (runs on the context's stream with both lanes' streams idle before and after: the other lane's
    // kernel in flight reads the tables, and this lane's next launch is on a stream of its own)
    PP_HIP(ctx, sync_lanes());
    struct SyncAtExit {
      pp_ctx* c;
      hipStream_t s;
      ~SyncAtExit() { (void)hipStreamSynchronize(s); }
    } sync_at_exit{ctx, s};
    if (ftab_slots == 0) {  // first time: tables for every query still alive
      size_t live = 0;
      for (size_t l = 0; l < nlanes; ++l) live += lanes[l].ids.size();
      uint64_t slots = (uint64_t)1 << 17;
      while (slots >= 4096 && live * slots * 16 > kFtabBudget) slots >>= 1;
      if (slots < 4096 || live == 0) {
        ftab_off = true;
        return PP_OK;
      }
      // The tables only spare the host work: a smaller table rules out less, no table nothing, and
      // the plans are the same.  So a device that has no room for them is not an error -- the
      // tables shrink until they fit, or the batch goes on without them.
      size_t need = live * slots * 16;
      if (need > ctx->ftab_bytes) {
        if (ctx->ftab_buf) PP_HIP(ctx, hipFree(ctx->ftab_buf));
        ctx->ftab_buf = nullptr;
        ctx->ftab_bytes = 0;
        for (;;) {
          const hipError_t me = hipMalloc(&ctx->ftab_buf, need);
          if (me == hipSuccess) break;
          ctx->ftab_buf = nullptr;
          (void)hipGetLastError();  // the failed allocation must not surface at the next launch check
          if (me != hipErrorOutOfMemory) PP_HIP(ctx, me);
          slots >>= 1;
          need = live * slots * 16;
          if (slots < 4096) {
            ftab_off = true;
            return PP_OK;
          }
        }
        ctx->ftab_bytes = need;
      }
      PP_HIP(ctx, hipMemsetAsync(ctx->ftab_buf, 0xFF, need, s));  // empty keys, values above every cost
      ftab_slots = slots;
      ftab_tables = (uint32_t)live;
      eo.ftab_slots = slots;
      eo.f_minyaw = P.minyaw;
      eo.f_xw = P.xw;
      eo.f_yw = P.yw;
    }
    std::vector<uint32_t> fresh;
    std::vector<size_t> base;
    size_t total = 0;
    for (size_t k = 0; k < ln.ids.size(); ++k) {
      Query& q = Q[ln.ids[k]];
      if (q.ftab >= 0 || ftab_next >= ftab_tables) continue;
      q.ftab = (int32_t)ftab_next++;
      for (size_t it = ln.item0[k], end = k + 1 < ln.ids.size() ? ln.item0[k + 1] : ln.ni; it < end; ++it)
        ln.emit[it].fid = q.ftab;
      fresh.push_back(ln.ids[k]);
      base.push_back(total);
      total += q.tab.size();
    }
    if (total == 0) return PP_OK;
    if (total > UINT32_MAX) return PP_ERR_UNSUPPORTED;
    std::vector<PpFtabUpd> all(total);
    pool.run(fresh.size(), [&](size_t j) {
      const Query& q = Q[fresh[j]];
      PpFtabUpd* o = all.data() + base[j];
      q.tab.for_each([&](const NodeTable::Slot& sl) {
        *o++ = PpFtabUpd{sl.key, sl.state() == NodeTable::kClosed ? 0ull : f64_bits(sl.cost), q.ftab, 0};
      });
    });
    // What the searches know so far goes over in pieces of whatever staging memory there is; with
    // none at all the tables start empty and learn from the rounds to come.
    size_t piece = total;
    void* tmp = nullptr;
    for (;;) {
      const hipError_t me = hipMalloc(&tmp, piece * sizeof(PpFtabUpd));
      if (me == hipSuccess) break;
      tmp = nullptr;
      (void)hipGetLastError();
      if (me != hipErrorOutOfMemory) PP_HIP(ctx, me);
      if (piece <= 4096) return PP_OK;
      piece = (piece + 1) / 2;
    }
    hipError_t e = hipSuccess;
    for (size_t at = 0; at < total && e == hipSuccess; at += piece) {
      const size_t m = std::min(piece, total - at);
      e = hipMemcpyAsync(tmp, all.data() + at, m * sizeof(PpFtabUpd), hipMemcpyHostToDevice, s);
      if (e != hipSuccess) break;
      pp_ftab_update_launch(ctx, (uint32_t)m, (const PpFtabUpd*)tmp, (unsigned long long*)ctx->ftab_buf, ftab_slots);
      e = hipStreamSynchronize(s);  // tmp is reused by the next piece, `all` and tmp go after the last
    }
    (void)hipFree(tmp);
    PP_HIP(ctx, e);
    return PP_OK;
  };

  // (2) stage the lane's nodes and launch.  Inputs and outputs are laid out for this round's
  // number of items, so each direction is one copy of exactly what is used.
  auto launch = [&](Lane& ln) -> int32_t {
    const double p1 = prof ? now_ms() : 0.0;
    size_t nl = 0, ni = 0;  // one node per live query, plus what the queries chose to send along
    ln.item0.resize(ln.ids.size());
    for (size_t k = 0; k < ln.ids.size(); ++k)
      if (ln.emit[k * ln.stride].alive) {
        ln.ids[nl] = ln.ids[k];
        ln.item0[nl] = (uint32_t)ni;
        for (size_t j = 0, n = ln.emit_n[k]; j < n; ++j) ln.emit[ni++] = ln.emit[k * ln.stride + j];  // ni <= k * stride
        ++nl;
      }
    ln.ids.resize(nl);
    ln.nl = nl;
    ln.ni = ni;
    if (nl == 0) {  // the lane is done
      ln.emit.clear();
      ln.emit_n.clear();
      return PP_OK;
    }
    ++ln.rounds;
    // a round of few nodes is latency, not host work: it runs without the tables (what they miss
    // of its news only makes them rule out less)
    const bool tables = !ftab_off && nl >= kFtabMinLive;
    if (tables && ln.unassigned && ln.rounds > kFtabStartRound) {
      const int32_t fst = ftab_assign(ln);
      if (fst != PP_OK) return fst;
    }
    // this round's news for the tables: with the round's own launch when that is one kernel
    uint32_t n_upd = take_updates(ln);
    if (!(tables && ftab_slots != 0)) n_upd = 0;
    eo.ftab = tables && ftab_slots != 0 ? (const unsigned long long*)ctx->ftab_buf : nullptr;
    ln.pc = nl < 64 ? 1 : 0;
    ln.depth = ln.stride;
    size_t ro = ln.io_base;
    auto rtake = [&](size_t bytes) {
      size_t o = ro;
      ro += al256(bytes);
      return o;
    };
    const size_t r_q = rtake(6 * ni * 8), r_dir = rtake(ni * 8), r_steer = rtake(ni * 8), r_cost = rtake(ni * 8);
    const size_t r_hid = rtake(ni * 4), r_fid = rtake(ni * 4);
    const size_t r_in = ro;
    const size_t itn = ni * NP;
    ln.r_shot = rtake(ni * sizeof(PpShotOut));
    ln.r_valid = rtake(itn);
    ln.r_rec = rtake(itn * 64);  // one 64-byte record per primitive (PpExpandDev::rec)
    const size_t r_out = ro - r_in;
    double* hq = (double*)(h + r_q);
    int64_t* hdir = (int64_t*)(h + r_dir);
    double* hsteer = (double*)(h + r_steer);
    double* hcost = (double*)(h + r_cost);
    int32_t* hhid = (int32_t*)(h + r_hid);
    int32_t* hfid = (int32_t*)(h + r_fid);
    {
      pool.run(ni, [&](size_t k) {  // straight from the records
        const Emit& e = ln.emit[k];
        hq[0 * ni + k] = e.x;
        hq[1 * ni + k] = e.y;
        hq[2 * ni + k] = e.yaw;
        hq[3 * ni + k] = e.gx;
        hq[4 * ni + k] = e.gy;
        hq[5 * ni + k] = e.gyaw;
        hdir[k] = e.dir;
        hsteer[k] = e.steer;
        hcost[k] = e.cost;
        hhid[k] = e.hid;
        hfid[k] = e.fid;
      }, 64, 512);
    }
    // The next round's records: while few queries are left each also sends nodes it may close soon
    // (kSpecItems nodes in all, by this round's count of live queries -- the next one's is not higher).
    ln.stride = std::min<size_t>(kSpecDepth, std::max<size_t>(1, kSpecItems / nl));
    ln.emit.resize(nl * ln.stride);
    ln.emit_n.resize(nl);
    const double p2 = prof ? now_ms() : 0.0;
    ++c_rounds;
    if (nl < 64) ++c_small;
    c_items += ni;
    if (prof) {
      p_stage[ln.pc] += p2 - p1;
      ++p_rounds[ln.pc];
      p_items[ln.pc] += ni;
      p_live[ln.pc] += nl;
    }
    // A round of few nodes skips both staging copies (each a blit kernel of its own, with its
    // launch gap): the kernels read and write the pinned area itself.  Only for few nodes -- every
    // primitive reads its parent's pose, and over PCIe that does not scale.
    char* const dw = d;  // the Reeds-Shepp work arrays stay in device memory
    const bool direct = ni < kRoundDirectMax;
    char* const dd = direct ? h : dw;
    const hipStream_t ls = ln.stream;
    StreamScope on_lane_stream(ctx, ls);
    if (!direct) PP_HIP(ctx, hipMemcpyAsync(dd + ln.io_base, h + ln.io_base, r_in - ln.io_base, hipMemcpyHostToDevice, ls));
    // A large round writes its results straight into the pinned area as well: full-wave stores
    // cross PCIe while the kernels run, where a copy afterwards takes as long again as the kernels
    // (the inputs stay in device memory: they are read once per primitive).
    const bool out_host = !direct && ni >= kRoundOutHostMin;
    char* const od = out_host ? h : dd;
    eo.sparse = out_host ? 1 : 0;
    eo.valid = (uint8_t*)(od + ln.r_valid);
    eo.rec = (uint64_t*)(od + ln.r_rec);
    eo.hmap_id = (const int32_t*)(dd + r_hid);
    eo.ftab_id = (const int32_t*)(dd + r_fid);
    const double* dq = (const double*)(dd + r_q);
    if (ni < kRoundFusedMax) {
      pp_round_launch(ctx, map, conf, plan, (uint32_t)ni, dq, maxc, (PpShotOut*)(od + ln.r_shot),
                      (const int64_t*)(dd + r_dir), (const double*)(dd + r_steer), (const double*)(dd + r_cost),
                      (const double*)d_mot, eo, n_upd, ln.upd, (unsigned long long*)ctx->ftab_buf,
                      direct ? ln.done : nullptr);  // nothing follows the kernel: it carries the lane's event
    } else {
      pp_ftab_update_launch(ctx, n_upd, ln.upd, (unsigned long long*)ctx->ftab_buf, ftab_slots);
      PpRsCand* cands = (PpRsCand*)(dw + o_cand) + ln.item_base * PP_RS_MAX_PATHS;
      PpRsPair* pairs = (PpRsPair*)(dw + o_pair) + ln.item_base;
      pp_rs_words_launch(ctx, ni, dq, maxc, cands, pairs);
      pp_rs_shot_launch(ctx, map, conf, (uint32_t)ni, dq, maxc, cands, pairs, (PpShotOut*)(od + ln.r_shot));
      pp_expand_launch(ctx, map, plan, ni, dq, dq + ni, dq + 2 * ni, (const int64_t*)(dd + r_dir),
                       (const double*)(dd + r_steer), (const double*)(dd + r_cost), (const double*)d_mot, eo);
    }
    PP_HIP(ctx, hipGetLastError());
    if (!direct && !out_host) PP_HIP(ctx, hipMemcpyAsync(h + r_in, dd + r_in, r_out, hipMemcpyDeviceToHost, ls));
    if (!direct) PP_HIP(ctx, hipEventRecord(ln.done, ls));  // (a direct round is one launch: see above)
    ln.in_flight = true;
    if (prof) p_dev[ln.pc] += now_ms() - p2;
    return PP_OK;
  };

  // (3) wait for the lane's results; the closed node is applied straight from the staging buffer,
  // the nodes fetched ahead of their turn go into the query's cache
  auto collect = [&](Lane& ln) -> int32_t {
    if (!ln.in_flight) return PP_OK;
    const double p2 = now_ms();
    if (lent) {
      // the thread that drives a part does not sit out its wait for the device: meanwhile it takes
      // items of the other parts' jobs, a chunk at a time
      hipError_t qe;
      while ((qe = hipEventQuery(ln.done)) == hipErrorNotReady)
        if (!lent->help_once()) PP_POOL_PAUSE();
      PP_HIP(ctx, qe);
    } else {
      PP_HIP(ctx, hipEventSynchronize(ln.done));
    }
    ln.in_flight = false;
    const double p3 = now_ms();
    if (prof) p_dev[ln.pc] += p3 - p2;
    const PpShotOut* shot = (const PpShotOut*)(h + ln.r_shot);
    const uint8_t* v_valid = (const uint8_t*)(h + ln.r_valid);
    const uint64_t* v_rec = (const uint64_t*)(h + ln.r_rec);
    const size_t depth = ln.depth;
    pool.run(ln.nl, [&](size_t k) {
      Query& q = Q[ln.ids[k]];
      const size_t first = ln.item0[k];
      // entries of nodes that were replaced in the open list are never asked for again: a long
      // search drops the lot now and then (whatever is still wanted is simply fetched again)
      if (q.cache.size() > kCacheMax) q.cache.clear();
      for (size_t j = 1; j < q.want.size(); ++j) {
        const size_t it = first + j;
        Expansion& e = q.cache.insert(q.want[j], NP);
        e.shot = shot[it];
        std::memcpy(e.valid(NP), v_valid + it * NP, NP);
        for (uint32_t i = 0; i < NP; ++i)
          if (v_valid[it * NP + i] & 1u) std::memcpy(e.words.get() + 8 * (size_t)i, v_rec + 8 * (it * NP + i), 64);
      }
      // the next query's results (written by the device: in no cache) and the query itself
      if (depth == 1 && k + 1 < ln.nl) {
        const size_t nb = (k + 1) * NP;
        __builtin_prefetch(&shot[k + 1]);
        for (size_t o = 0; o < NP; o += 64) __builtin_prefetch(v_valid + nb + o);
        const char* qn = (const char*)&Q[ln.ids[k + 1]];
        for (size_t o = 0; o < sizeof(Query); o += 64) __builtin_prefetch(qn + o);
      }
      const size_t it = first, b = it * NP;
      const ExpansionView v{&shot[it], v_valid + b, v_rec + 8 * b};
      if (beyond_limit(v)) {
        need_grids.store(true, std::memory_order_relaxed);  // q.curr stays: the node goes to the device again
      } else {
        apply(q, v);
        q.curr = -1;
      }
      advance_emit(ln, k);  // the next node to expand, while the query's tables are in cache
    }, depth > 1 ? 1 : 8, depth > 1 ? 2 : 64);  // with the look-ahead a query is several nodes' worth of work
    if (need_grids.load(std::memory_order_relaxed)) {
      // (the other lane's round in flight was launched on the bounded grids: whatever of it looked
      // beyond the limit carries the mark and comes back here at that lane's turn)
      const double pf = prof ? now_ms() : 0.0;
      PP_HIP(ctx, sync_lanes());  // the other lane's kernel in flight reads the grids
      int32_t fst = pp_wavefront_multi_finish(ctx);
      if (fst != PP_OK) return fst;
      PP_HIP(ctx, hipStreamSynchronize(s));  // complete before either lane launches again
      if (prof) fprintf(stderr, "[plan] heuristic grids completed during the search: %.1f ms\n", now_ms() - pf);
      h_bounded = false;
      eo.hobs = nullptr;
      need_grids.store(false, std::memory_order_relaxed);
      for (size_t l = 0; l < nlanes; ++l)
        for (uint32_t id : lanes[l].ids) Q[id].cache.clear();  // f costs from the bounded grids
    }
    if (prof) p_apply[ln.pc] += now_ms() - p3;
    return PP_OK;
  };

  // leaving with an error: the other lane's round may still be running on the staging buffers
  auto drain = [&](int32_t code) {
    (void)sync_lanes();
    return code;
  };
  for (size_t l = 0; l < nlanes; ++l) {
    Lane& ln = lanes[l];
    const double p0 = prof ? now_ms() : 0.0;
    ln.emit.resize(ln.ids.size());
    ln.emit_n.resize(ln.ids.size());
    pool.run(ln.ids.size(), [&](size_t k) { advance_emit(ln, k); });
    if (prof) p_adv[ln.ids.size() < 64 ? 1 : 0] += now_ms() - p0;
  }
  for (size_t l = 0; l < nlanes; ++l) {
    st = launch(lanes[l]);
    if (st != PP_OK) return drain(st);
  }
  // Two lanes pay off while the host is the slower side -- one lane's results are applied and its
  // next nodes chosen while the other's kernels run -- and cost nothing when the device is: on
  // streams of their own both lanes' kernels are in flight together, so each lane still waits one
  // device latency per round.  (On one stream the second lane's kernel ran behind the first's, and
  // the lanes had to be merged once the host started waiting.)
  while (lanes[0].in_flight || lanes[1].in_flight) {
    for (size_t l = 0; l < nlanes; ++l) {
      Lane& ln = lanes[l];
      st = collect(ln);
      if (st != PP_OK) return drain(st);
      st = launch(ln);
      if (st != PP_OK) return drain(st);
    }
  }
  const double t_search = now_ms();
  if (prof)
    for (int c = 0; c < 2; ++c)
      fprintf(stderr, "[plan] %s rounds %llu live-sum %llu items %llu | advance %.1f stage %.1f device %.1f apply %.1f ms\n",
              c ? "small" : "large", (unsigned long long)p_rounds[c], (unsigned long long)p_live[c],
              (unsigned long long)p_items[c], p_adv[c], p_stage[c], p_dev[c], p_apply[c]);

  // ---- extract_path (hybrid_astar.rs:218-250) for the queries that found one: the analytic
  // shots' trajectories and the rollout lists of every node on a final path are regenerated in
  // two batched launches.
  int32_t ret = PP_OK;
  struct Found {
    uint32_t q;
    std::vector<int32_t> chain;  // final RS node's parent <- ... <- start
    bool with_final;
    uint32_t nroll, roll0;       // rollout lists used, first one's index in the batch
  };
  std::vector<Found> found;
  for (uint64_t i = 0; i < nq; ++i) {
    out[i].len = 0;
    out[i].cost = 0.0;
    if (!Q[i].found) continue;
    Found f;
    f.q = (uint32_t)i;
    found.push_back(std::move(f));
  }
  pool.run(found.size(), [&](size_t k) {
    Found& f = found[k];
    const Query& q = Q[f.q];
    for (int32_t id = q.final_parent; id >= 0; id = q.nodes[id].parent) f.chain.push_back(id);
    const SNode& fp = q.nodes[q.final_parent];
    // The reference walks `while node.pind >= 0` with pind an isize copy of the parent's key
    // (:226, :372): a key that wrapped below zero (yaw index under minyaw, the rollout never wraps
    // yaw) ends the walk early and the start point is appended regardless.
    f.with_final = (int64_t)calc_index(fp.xind, fp.yind, fp.yawind, P) >= 0;
    f.nroll = 0;
    if (f.with_final)
      while (f.nroll + 1 < f.chain.size() && q.nodes[f.chain[f.nroll]].pind >= 0) ++f.nroll;
  });
  uint32_t nroll_sum = 0;
  for (Found& f : found) {
    f.roll0 = nroll_sum;
    nroll_sum += f.nroll;
  }
  std::vector<double> in5((size_t)nroll_sum * 5);
  pool.run(found.size(), [&](size_t k) {
    const Found& f = found[k];
    const Query& q = Q[f.q];
    for (uint32_t r = 0; r < f.nroll; ++r) {
      const SNode& n = q.nodes[f.chain[r]];
      const SNode& pa = q.nodes[n.parent];
      double* o = in5.data() + (size_t)(f.roll0 + r) * 5;
      o[0] = pa.x;
      o[1] = pa.y;
      o[2] = pa.yaw;
      o[3] = n.steer;
      o[4] = plan.motion[NP + n.prim];
    }
  });
  const double x0 = now_ms();
  const uint32_t nf = (uint32_t)found.size();
  const uint32_t pts = plan.prm.nlist + 1;
  const uint32_t nroll_all = (uint32_t)(in5.size() / 5);
  std::vector<double> lx((size_t)nroll_all * pts), ly(lx.size()), lyaw(lx.size());
  if (nroll_all) {
    uint32_t nlist = 0;
    st = pp_rollout_points_batch(ctx, conf, nroll_all, in5.data(), lx.data(), ly.data(), lyaw.data(), &nlist);
    if (st != PP_OK) return st;
  }
  const double x1 = now_ms();
  // analytic-shot trajectories of all found queries, packed
  const double *rx = nullptr, *ry = nullptr, *ryaw = nullptr;
  const int64_t* rdir = nullptr;
  std::vector<double> rtotal;
  std::vector<uint32_t> rfl;
  std::vector<uint64_t> roff;
  {
    std::vector<double> q6((size_t)6 * nf);
    std::vector<int32_t> sel(nf);
    for (uint32_t k = 0; k < nf; ++k) {
      const Query& q = Q[found[k].q];
      const SNode& fp = q.nodes[q.final_parent];
      q6[k] = fp.x;
      q6[nf + k] = fp.y;
      q6[2 * (size_t)nf + k] = fp.yaw;
      q6[3 * (size_t)nf + k] = q.goal[0];
      q6[4 * (size_t)nf + k] = q.goal[1];
      q6[5 * (size_t)nf + k] = q.goal[2];
      sel[k] = q.final_rs_idx;
    }
    st = pp_rs_selected_batch(ctx, nf, q6.data(), maxc, conf->move_step, sel.data(), roff, rtotal, rfl, &rx,
                              &ry, &ryaw, &rdir);
    if (st != PP_OK) return st;
  }
  const double x2 = now_ms();
  pool.run(nf, [&](size_t k) {
    const Found& f = found[k];
    Query& q = Q[f.q];
    pp_path& po = out[f.q];
    const uint64_t rnp_k = roff[k + 1] - roff[k];
    if ((rfl[k] & (4u | 8u | 16u)) || rnp_k == 0) {
      q.status = PP_ERR_DEGENERATE_RS;
      return;
    }
    q.st.rs_final_length = rtotal[k];
    const SNode& fp = q.nodes[q.final_parent];
    // The reference accumulates the path backwards (the analytic node's points, then each
    // rollout list, then the start point) and reverses it once; here every point goes straight to
    // its final slot.
    const uint64_t n_rs = f.with_final ? rnp_k - 1 : 0;
    const size_t L = (size_t)n_rs + (size_t)f.nroll * (pts - 1) + 1;
    double cost = 0.0;
    if (f.with_final) cost += fp.cost + q.final_rs_cost;  // fcost = n_curr.cost + rs cost (:357)
    for (uint32_t r = 0; r < f.nroll; ++r) cost += q.nodes[f.chain[r]].cost;
    const SNode& ns = q.nodes[f.chain.back()];
    cost += ns.cost;
    po.len = L;
    po.cost = cost;
    if (po.cap < L || !po.x || !po.y || !po.yaw || !po.direction) {
      q.status = PP_ERR_CAPACITY;
      return;
    }
    size_t at = L;  // slots [at, L) are filled
    if (f.with_final) {
      const size_t a = roff[k];
      for (uint64_t j = rnp_k; j-- > 1;) {  // node.x[1..] of the RS node
        --at;
        po.x[at] = rx[a + j];
        po.y[at] = ry[a + j];
        po.yaw[at] = ryaw[a + j];
        po.direction[at] = rdir[a + j];
      }
    }
    for (uint32_t r = 0; r < f.nroll; ++r) {
      const SNode& n = q.nodes[f.chain[r]];
      const size_t a = (size_t)(f.roll0 + r) * pts;
      for (uint32_t j = pts; j-- > 1;) {
        --at;
        po.x[at] = lx[a + j];
        po.y[at] = ly[a + j];
        po.yaw[at] = lyaw[a + j];
        po.direction[at] = n.direction;
      }
    }
    po.x[0] = ns.x;
    po.y[0] = ns.y;
    po.yaw[0] = ns.yaw;
    po.direction[0] = 1;
    q.status = PP_OK;
  });
  const double t_extract = now_ms();
  if (getenv("PP_PLAN_PROFILE"))
    fprintf(stderr, "[plan] extract: chains %.2f rollouts %.2f rs %.2f assemble %.2f ms (%u rollouts)\n", x0 - t_search,
            x1 - x0, x2 - x1, t_extract - x2, nroll_all);
  if (getenv("PP_PLAN_PROFILE"))
    fprintf(stderr, "[plan] entry->begin %.1f heur %.1f search(incl. staging alloc) %.1f extract %.1f ms\n", t_begin - t_entry, t_heur - t_begin, t_search - t_heur, t_extract - t_search);
  {
    uint64_t longest = 0, closed = 0;
    for (uint64_t i = 0; i < nq; ++i) {
      longest = std::max<uint64_t>(longest, Q[i].st.expansions);
      closed += Q[i].st.expansions;
    }
    uint64_t* c = ctx->plan_counters;
    c[0] = c_rounds;
    c[1] = c_small;
    c[2] = c_items;
    c[3] = ngoals;
    c[4] = longest;
    c[5] = closed;
    c[6] = nq;
    c[7] = (uint64_t)nlanes;
  }
  for (uint64_t i = 0; i < nq; ++i) {
    Query& q = Q[i];
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
  // a query that grew far beyond the usual gives its memory back; the rest keep theirs
  for (uint64_t i = 0; i < nq; ++i)
    if (Q[i].nodes.capacity() > (1u << 16)) Q[i] = Query();
  if (getenv("PP_PLAN_PROFILE")) fprintf(stderr, "[plan] teardown %.1f ms\n", now_ms() - t_extract);
  return ret;
}

extern "C" int32_t pp_plan_batch_counters(pp_ctx* ctx, uint64_t out[8]) {
  if (!ctx || !out) return PP_ERR_INVALID;
  std::memcpy(out, ctx->plan_counters, sizeof(ctx->plan_counters));
  return PP_OK;
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

// ---------------------------------------------------------------------------------------------
// A batch dealt over several contexts: context k takes a share of the queries, a host thread per
// context runs the batch on it, and every result lands at its query's place in the caller's
// arrays.  The queries are dealt by goal: ordered by the heuristic-grid cell of their goal
// (astar.rs:141-146) and cut into n_ctx runs of equal length, so a context relaxes the grids of
// its own goals only -- about (distinct goals) / n_ctx + 1 of them -- and no data travels between
// contexts.  A query's result does not depend on the batch it is planned in, so the outcome equals
// one batch over the whole list on one context.  `core`: run plan_batch_core on each share (the
// contexts are the parts of one split batch) instead of pp_plan_batch (which may split again).
static int32_t plan_batch_dealt(pp_ctx* const* ctxs, const pp_map* const* maps, uint32_t n_ctx,
                                const pp_config* conf, uint64_t nq, const double* starts,
                                const double* goals, pp_path* out, pp_plan_stats* stats, int32_t* status,
                                bool core) {
  if (!(conf->xy_reso > 0.0)) return PP_ERR_INVALID;
  // Order by goal cell, ties by position (queries of one goal keep their order), the goal groups
  // taken in turn: group r of the sorted cells goes to the front of share r mod n_ctx.  Neighbouring
  // goals tend to be equally hard (the same cluttered corner of the map), and a contiguous run of
  // them would make one share the slowest.
  std::vector<uint64_t> order(nq);
  std::vector<std::pair<uint64_t, uint64_t>> cell(nq);
  for (uint64_t i = 0; i < nq; ++i) {
    order[i] = i;
    cell[i] = {pp_sat_usize(std::round(goals[3 * i] / conf->xy_reso)),
               pp_sat_usize(std::round(goals[3 * i + 1] / conf->xy_reso))};
  }
  std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return cell[a] < cell[b]; });
  {
    std::vector<uint64_t> rank(nq);  // rank of a query's goal cell among the distinct cells
    uint64_t r = 0;
    for (uint64_t k = 0; k < nq; ++k) {
      if (k && cell[order[k]] != cell[order[k - 1]]) ++r;
      rank[order[k]] = r;
    }
    std::stable_sort(order.begin(), order.end(),
                     [&](uint64_t a, uint64_t b) { return rank[a] % n_ctx < rank[b] % n_ctx; });
  }
  struct Share {
    std::vector<uint64_t> ids;
    std::vector<double> s, g;
    std::vector<pp_path> paths;
    std::vector<pp_plan_stats> stats;
    std::vector<int32_t> status;
    int32_t ret = PP_OK;
  };
  std::vector<Share> sh(n_ctx);
  for (uint32_t k = 0; k < n_ctx; ++k) {
    Share& h = sh[k];
    const uint64_t lo = nq * k / n_ctx, hi = nq * (k + 1) / n_ctx;
    h.ids.assign(order.begin() + lo, order.begin() + hi);
    std::sort(h.ids.begin(), h.ids.end());
    const size_t m = h.ids.size();
    h.s.resize(3 * m);
    h.g.resize(3 * m);
    h.paths.resize(m);
    h.status.assign(m, PP_OK);
    if (stats) h.stats.resize(m);
    for (size_t j = 0; j < m; ++j) {
      const uint64_t i = h.ids[j];
      std::memcpy(&h.s[3 * j], &starts[3 * i], 24);
      std::memcpy(&h.g[3 * j], &goals[3 * i], 24);
      h.paths[j] = out[i];               // the caller's buffers and capacities
      if (stats) h.stats[j] = stats[i];  // closed_order / closed_cap are inputs
    }
  }
  auto run = [&](uint32_t k) {
    Share& h = sh[k];
    if (h.ids.empty()) {
      std::memset(ctxs[k]->plan_counters, 0, sizeof(ctxs[k]->plan_counters));
      return;
    }
    h.ret = (core ? plan_batch_core : pp_plan_batch)(ctxs[k], maps[k], conf, h.ids.size(), h.s.data(), h.g.data(),
                                                     h.paths.data(), stats ? h.stats.data() : nullptr,
                                                     h.status.data());
  };
  // a thread per share, the calling thread taking the first; a share whose thread cannot be started
  // (no exception may cross the C boundary) is planned by the calling thread afterwards
  std::vector<std::thread> th;
  std::vector<uint32_t> left;
  for (uint32_t k = 1; k < n_ctx; ++k) {
    try {
      th.emplace_back(run, k);
    } catch (...) {
      left.push_back(k);
    }
  }
  run(0);
  for (uint32_t k : left) run(k);
  for (auto& t : th) t.join();
  int32_t ret = PP_OK;
  for (uint32_t k = 0; k < n_ctx; ++k) {
    Share& h = sh[k];
    if (h.ret != PP_OK && ret == PP_OK) ret = h.ret;  // the batch as a whole failed on that context
    for (size_t j = 0; j < h.ids.size(); ++j) {
      const uint64_t i = h.ids[j];
      out[i].len = h.paths[j].len;
      out[i].cost = h.paths[j].cost;
      if (stats) stats[i] = h.stats[j];
      if (status) status[i] = h.ret == PP_OK ? h.status[j] : h.ret;
      else if (h.ret == PP_OK && h.status[j] < 0 && ret == PP_OK) ret = h.status[j];
    }
  }
  return ret;
}

// hybrid_astar_planning for n queries on one context.  A large batch runs as several independent
// parts on streams of their own (helper contexts on the same device, kept in the context): the
// host side of a round has stretches only the calling thread works on, and the tables of
// thousands of queries do not fit the host's caches; parts run those stretches side by side on
// smaller working sets (4096 queries as four parts: 23.6 -> 20.5 ms, the search-heavy variant
// 0.67 -> 0.55 s; since the parts share their worker threads, six).  Which part a query is planned
// in changes nothing about its result.
constexpr uint64_t kSplitMinQueries = 2048;
static uint32_t batch_parts(const pp_ctx* ctx, uint64_t nq) {
  if (ctx->is_helper) return 1;
  if (ctx->batch_parts) return (uint32_t)std::min<uint64_t>(ctx->batch_parts, std::max<uint64_t>(1, nq));
  if (nq < kSplitMinQueries || ctx->host_threads < 3) return 1;
  // parts of about 700 queries, six at most (they share the workers, so a part costs one driving
  // thread): 4096 queries in six parts against four -- the light benchmark batch 16.1-16.4 ms against
  // 19.0-19.5, the search-heavy one 268-271 against 277-285; eight parts are slower again
  const uint64_t by_size = (nq + 767) / 768;
  return (uint32_t)std::min<uint64_t>({6, (ctx->host_threads + 1) / 2, by_size});
}

extern "C" int32_t pp_plan_batch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint64_t nq,
                                 const double* starts, const double* goals, pp_path* out,
                                 pp_plan_stats* stats, int32_t* status) {
  if (!ctx || !map || !conf || (nq && (!starts || !goals || !out))) return PP_ERR_INVALID;
  const uint32_t parts = nq ? batch_parts(ctx, nq) : 1;
  if (parts <= 1) return plan_batch_core(ctx, map, conf, nq, starts, goals, out, stats, status);
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (nq > (1u << 24)) return PP_ERR_UNSUPPORTED;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  while (ctx->helpers.size() + 1 < parts) {
    pp_ctx* h = nullptr;
    st = pp_ctx_create(ctx->device, nullptr, &h);
    if (st != PP_OK) return st;
    h->is_helper = true;
    ctx->helpers.push_back(h);
  }
  // The parts share the context's settings and its allowance of threads and memory.  The threads:
  // one drives each part, the rest are one pool that serves all of them -- a part that waits for the
  // device, or is done, leaves the workers to the others (a fixed share per part left a quarter of
  // the cores idle in the tail of a batch).
  const uint32_t own_workers = ctx->host_threads;
  const unsigned hw = std::thread::hardware_concurrency();
  const uint32_t budget = std::min<uint32_t>(ctx->host_threads + 1, hw ? hw : 1);  // threads in all
  const uint32_t workers = budget > parts ? budget - parts : 0;
  PlanState& own_state = plan_state_of(ctx);
  if (workers && (!own_state.shared_pool || own_state.shared_pool_asked != workers)) {
    try {
      own_state.shared_pool.reset(new Pool(workers));
      own_state.shared_pool_asked = workers;
    } catch (...) {
      own_state.shared_pool.reset();  // no threads to be had: every part works on its own thread alone
    }
  }
  Pool* const lend = workers ? own_state.shared_pool.get() : nullptr;
  const size_t own_budget = ctx->ftab_budget, part_budget = own_budget / parts;
  std::vector<pp_ctx*> cs{ctx};
  for (uint32_t k = 1; k < parts; ++k) cs.push_back(ctx->helpers[k - 1]);
  for (pp_ctx* c : cs) {
    c->max_expansions = ctx->max_expansions;
    c->heuristic_limit = ctx->heuristic_limit;
    c->heuristic_pass_blocks = ctx->heuristic_pass_blocks;
    c->wave_band = ctx->wave_band;
    c->ftab_budget = part_budget;
    c->host_threads = 0;
    c->lent_pool = lend;
  }
  std::vector<const pp_map*> ms(parts, map);
  st = plan_batch_dealt(cs.data(), ms.data(), parts, conf, nq, starts, goals, out, stats, status, true);
  for (pp_ctx* c : cs) c->lent_pool = nullptr;
  ctx->host_threads = own_workers;
  ctx->ftab_budget = own_budget;
  // the batch's counters: what bounds its time is the part with the most rounds
  uint64_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (pp_ctx* p : cs) {
    const uint64_t* q = p->plan_counters;
    c[0] = std::max(c[0], q[0]);
    c[1] = std::max(c[1], q[1]);
    c[2] += q[2];
    c[3] += q[3];
    c[4] = std::max(c[4], q[4]);
    c[5] += q[5];
    c[6] += q[6];
    c[7] += q[7];
    if (p != ctx && p->last_hip_error) ctx->last_hip_error = p->last_hip_error;
  }
  std::memcpy(ctx->plan_counters, c, sizeof(c));
  return st;
}

// The batch over several devices of one process: one context and one map per device.
extern "C" int32_t pp_plan_batch_multi(pp_ctx* const* ctxs, const pp_map* const* maps, uint32_t n_ctx,
                                       const pp_config* conf, uint64_t nq, const double* starts,
                                       const double* goals, pp_path* out, pp_plan_stats* stats,
                                       int32_t* status) {
  if (!ctxs || !maps || n_ctx == 0 || !conf || (nq && (!starts || !goals || !out))) return PP_ERR_INVALID;
  for (uint32_t k = 0; k < n_ctx; ++k) {
    if (!ctxs[k] || !maps[k]) return PP_ERR_INVALID;
    for (uint32_t j = 0; j < k; ++j)
      if (ctxs[j] == ctxs[k]) return PP_ERR_INVALID;  // a context runs one batch at a time
  }
  if (nq == 0) return PP_OK;
  if (n_ctx == 1) return pp_plan_batch(ctxs[0], maps[0], conf, nq, starts, goals, out, stats, status);
  return plan_batch_dealt(ctxs, maps, n_ctx, conf, nq, starts, goals, out, stats, status, false);
}
