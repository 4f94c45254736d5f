// pp_heuristic.hip -- holonomic heuristic: obstacle raster (calc_obsmap, astar.rs:244-261) and
// cost-to-go grid (the Dijkstra of astar.rs:153-198) as a bucketed wavefront relaxation.
//
// The cost-to-go is the unique fixpoint d[v] = min over valid neighbours n of fl(d[n] + w),
// w in {1, hypot(1,1)}, d[goal] = 0: every addition the serial Dijkstra performs along a
// shortest path is performed here too, in fp64, and min is order-free, so the grid is
// bit-identical to the reference's whatever the schedule.  Edge weights are >= 1, hence every
// cell whose cost lies in [k, k+1) is final once all cells below k have been expanded
// (fl(c + w) >= k + 1 for c >= k): the wavefront processes unit-width cost buckets in order,
// each cell expanded once with its final cost.  Relaxations are 64-bit atomic min on the cost
// bits (non-negative doubles order like unsigned integers).
#include <algorithm>

#include "pp_internal.h"

namespace {

// ------------------------------------------------------------------ exact `hypot(a,b) <= thr`

struct Expansion {  // non-overlapping expansion, increasing magnitude (Shewchuk)
  double e[12];
  int n;
};
__device__ inline void two_sum(double a, double b, double& s, double& err) {
  s = a + b;
  double bb = s - a;
  err = (a - (s - bb)) + (b - bb);
}
__device__ inline void grow(Expansion& x, double b) {
  double q = b;
  int m = 0;
  for (int i = 0; i < x.n; ++i) {
    double s, err;
    two_sum(q, x.e[i], s, err);
    if (err != 0.0) x.e[m++] = err;
    q = s;
  }
  if (q != 0.0 || m == 0) x.e[m++] = q;
  x.n = m;
}
// sign of (a^2 + b^2) - (t + h)^2 computed exactly; h = half an ulp of t (a power of two)
__device__ __noinline__ int sumsq_vs_mid(double a, double b, double t, double h) {
  Expansion x;
  x.n = 0;
  double p, e;
  p = a * a; e = __builtin_fma(a, a, -p); grow(x, e); grow(x, p);
  p = b * b; e = __builtin_fma(b, b, -p); grow(x, e); grow(x, p);
  p = t * t; e = __builtin_fma(t, t, -p); grow(x, -e); grow(x, -p);
  grow(x, -2.0 * t * h);  // exact: h is a power of two
  grow(x, -h * h);
  double top = x.e[x.n - 1];
  return top > 0.0 ? 1 : (top < 0.0 ? -1 : 0);
}
// hypot(a, b) <= thr for a correctly rounded hypot (round to nearest even), decided without
// forming hypot: RN(sqrt(S)) <= thr  <=>  sqrt(S) < thr + ulp/2, or == with thr's mantissa even.
__device__ inline bool hypot_le(double a, double b, double thr) {
  if (!(thr >= 0.0)) return false;  // negative or NaN threshold
  a = fabs(a);
  b = fabs(b);
  if (!(a <= 1.0e150 && b <= 1.0e150 && thr <= 1.0e150)) return hypot(a, b) <= thr;  // no overflow care
  double s = a * a + b * b, t2 = thr * thr;
  if (s < t2 * (1.0 - 1.0e-14)) return true;
  if (s > t2 * (1.0 + 1.0e-14)) return false;
  if (thr < 1.0e-140) return hypot(a, b) <= thr;  // products would underflow
  double up = __longlong_as_double(__double_as_longlong(thr) + 1);
  double h = (up - thr) * 0.5;
  int sg = sumsq_vs_mid(a, b, thr, h);
  if (sg < 0) return true;
  if (sg > 0) return false;
  return (__double_as_longlong(thr) & 1) == 0;  // tie rounds to the even neighbour
}

// ------------------------------------------------------------------ K5: obstacle raster

// One lane per obstacle point: marks the coarse cells within rr/reso of it.  A cell (xx, yy) is
// blocked iff hypot(ox/reso - xx, oy/reso - yy) <= rr/reso for SOME obstacle (astar.rs:251-255);
// cells farther than thr + 1 from the obstacle in x or y cannot satisfy it.
__global__ void k_obsmap(uint64_t n, const double* __restrict__ ox, const double* __restrict__ oy,
                         double reso, double thr, int64_t minx, int64_t miny, int64_t maxx,
                         int64_t maxy, uint64_t yw, uint8_t* __restrict__ obs) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double fx = ox[i] / reso, fy = oy[i] / reso;
    double lx = fmax(floor(fx - thr) - 1.0, (double)minx), hx = fmin(ceil(fx + thr) + 1.0, (double)(maxx - 1));
    double ly = fmax(floor(fy - thr) - 1.0, (double)miny), hy = fmin(ceil(fy + thr) + 1.0, (double)(maxy - 1));
    if (!(lx <= hx) || !(ly <= hy)) continue;
    for (int64_t xx = (int64_t)lx; xx <= (int64_t)hx; ++xx)
      for (int64_t yy = (int64_t)ly; yy <= (int64_t)hy; ++yy)
        if (hypot_le(fx - (double)xx, fy - (double)yy, thr))
          obs[(uint64_t)(xx - minx) * yw + (uint64_t)(yy - miny)] = 1;
  }
}

// ------------------------------------------------------------------ K6: bucketed wavefront

struct WaveGrid {
  double* cost;          // column-major in x: cost[lx * yw + y]
  const uint8_t* obs;    // same indexing
  uint32_t yw;
  uint32_t ncols;        // columns stored (including halo columns for a slab)
  int64_t gx_off;        // global x of stored column 0
  int64_t xw;            // global grid width (valid cells: 0 < gx < xw, 0 < y < yw)
  uint32_t tgt_lo, tgt_hi;  // stored columns [tgt_lo, tgt_hi) may be relaxed (owned columns)
};

struct WaveQueues {
  uint32_t* cell[3];
  double* qcost[3];
  uint32_t cap;
  uint32_t* count;     // per absolute bucket
  uint32_t nbuckets;
  uint32_t* status;    // [0] max bucket pushed, [1] overflow flag, [2] changed_lo, [3] changed_hi
  // seeds expanded when the wavefront reaches their bucket (halo cells lowered by the caller)
  const uint32_t* seed_cell;
  const uint32_t* seed_count;
};

__device__ inline void wave_push(const WaveQueues& q, uint32_t cell, double c) {
  double fb = floor(c);
  uint32_t b = fb < 4.0e9 ? (uint32_t)fb : 0xFFFFFFFFu;
  if (b >= q.nbuckets) {
    atomicOr(&q.status[1], 2u);
    return;
  }
  uint32_t slot = atomicAdd(&q.count[b], 1u);
  if (slot >= q.cap) {
    atomicOr(&q.status[1], 1u);
    return;
  }
  q.cell[b % 3][slot] = cell;
  q.qcost[b % 3][slot] = c;
  atomicMax(&q.status[0], b);
}

// Relax the 8 neighbours of one final cell.  The loads, the atomic mins and the queue-slot
// reservations are each issued as a group so their latencies overlap; a cell in bucket k only
// ever pushes into buckets k+1 and k+2 (1 <= w < 1.5), so two reservations per lane suffice.
__device__ inline void wave_expand(const WaveGrid& g, const WaveQueues& q, uint32_t cell, double c,
                                   double diag, uint32_t k) {
  const uint32_t lx = cell / g.yw, y = cell - lx * g.yw;
  uint32_t ncell[8];
  double nc[8];
  bool ok[8];
  uint8_t ob[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // astar.rs:297-308
    const int dx = (j == 0 || j == 1 || j == 7) ? -1 : ((j >= 3 && j <= 5) ? 1 : 0);
    const int dy = (j >= 1 && j <= 3) ? 1 : ((j >= 5) ? -1 : 0);
    const int64_t nlx = (int64_t)lx + dx, ny = (int64_t)y + dy, ngx = nlx + g.gx_off;
    // owned column, and check_node's bounds (astar.rs:200-204)
    ok[j] = nlx >= (int64_t)g.tgt_lo && nlx < (int64_t)g.tgt_hi && ngx > 0 && ngx < g.xw && ny > 0 &&
            ny < (int64_t)g.yw;
    ncell[j] = ok[j] ? (uint32_t)nlx * g.yw + (uint32_t)ny : cell;
    nc[j] = c + ((dx != 0 && dy != 0) ? diag : 1.0);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) ob[j] = g.obs[ncell[j]];
  unsigned long long old[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ok[j] = ok[j] && ob[j] == 0;  // astar.rs:206-208
    const unsigned long long nb = (unsigned long long)__double_as_longlong(nc[j]);
    old[j] = ok[j] ? atomicMin((unsigned long long*)&g.cost[ncell[j]], nb) : 0ull;
  }
  uint32_t n1 = 0, n2 = 0;
  const double k2 = (double)k + 2.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ok[j] = ok[j] && (unsigned long long)__double_as_longlong(nc[j]) < old[j];
    if (ok[j]) {
      if (nc[j] < k2) ++n1; else ++n2;
    }
  }
  if (n1 + n2 == 0) return;
  const uint32_t b1 = k + 1, b2 = k + 2;
  if (b2 >= q.nbuckets) {
    atomicOr(&q.status[1], 2u);
    return;
  }
  uint32_t s1 = n1 ? atomicAdd(&q.count[b1], n1) : 0u;
  uint32_t s2 = n2 ? atomicAdd(&q.count[b2], n2) : 0u;
  if ((n1 && s1 + n1 > q.cap) || (n2 && s2 + n2 > q.cap)) {
    atomicOr(&q.status[1], 1u);
    return;
  }
  uint32_t* c1 = q.cell[b1 % 3];
  uint32_t* c2 = q.cell[b2 % 3];
  double* v1 = q.qcost[b1 % 3];
  double* v2 = q.qcost[b2 % 3];
  bool edge_lo = false, edge_hi = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!ok[j]) continue;
    if (nc[j] < k2) {
      c1[s1] = ncell[j];
      v1[s1++] = nc[j];
    } else {
      c2[s2] = ncell[j];
      v2[s2++] = nc[j];
    }
    const uint32_t nlx = ncell[j] / g.yw;
    edge_lo |= nlx == g.tgt_lo;
    edge_hi |= nlx + 1 == g.tgt_hi;
  }
  atomicMax(&q.status[0], n2 ? b2 : b1);
  if (edge_lo) q.status[2] = 1;
  if (edge_hi) q.status[3] = 1;
}

__global__ void __launch_bounds__(256)
k_wave_step(WaveGrid g, WaveQueues q, uint32_t k, double diag) {
  const uint32_t nq = min(q.count[k], q.cap);
  const uint32_t ns = q.seed_count ? *q.seed_count : 0u;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t* qc = q.cell[k % 3];
  const double* qv = q.qcost[k % 3];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nq + ns; i += stride) {
    uint32_t cell;
    double c;
    if (i < nq) {
      cell = qc[i];
      c = qv[i];
      // stale entry: the cell was lowered again after this entry was queued
      if (__double_as_longlong(g.cost[cell]) != __double_as_longlong(c)) continue;
    } else {
      cell = q.seed_cell[i - nq];
      c = g.cost[cell];
      if (!(c >= (double)k && c < (double)k + 1.0)) continue;
    }
    wave_expand(g, q, cell, c, diag, k);
  }
}

__global__ void k_fill_inf(double* p, uint64_t n) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = __longlong_as_double(0x7FF0000000000000ll);
}

__global__ void k_seed_goal(WaveGrid g, WaveQueues q, uint32_t cell) {
  g.cost[cell] = 0.0;
  wave_push(q, cell, 0.0);
}

// Halo columns handed in by the neighbours: where the new value is lower than the stored one,
// store it and list the cell as a seed; track the lowest / highest seed bucket.
__global__ void k_take_halo(WaveGrid g, const double* __restrict__ lo_new,
                            const double* __restrict__ hi_new, uint32_t* seed_cell,
                            uint32_t* seed_count, uint32_t* seed_range) {
  uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= g.yw) return;
  for (int side = 0; side < 2; ++side) {
    const double* src = side ? hi_new : lo_new;
    if (!src) continue;
    uint32_t cell = (side ? (g.ncols - 1) : 0u) * g.yw + y;
    double v = src[y];
    if (v < g.cost[cell]) {
      g.cost[cell] = v;
      uint32_t slot = atomicAdd(seed_count, 1u);
      seed_cell[slot] = cell;
      double fb = floor(v);
      uint32_t b = fb < 4.0e9 ? (uint32_t)fb : 0xFFFFFFFEu;
      atomicMin(&seed_range[0], b);
      atomicMax(&seed_range[1], b);
    }
  }
}

struct WaveWork {
  WaveQueues q{};
  uint32_t* seed_cell = nullptr;
  uint32_t* seed_count = nullptr;  // [0] count, [1] min bucket, [2] max bucket
  void* base = nullptr;
};

int32_t wave_alloc(pp_ctx* ctx, uint64_t ncells, uint32_t yw, uint32_t cap, WaveWork* w) {
  uint32_t nb = (uint32_t)std::min<uint64_t>(ncells + 16, (uint64_t)1 << 26);
  size_t sz = 0;
  auto take = [&](size_t bytes) {
    size_t o = sz;
    sz += (bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_cell[3], o_cost[3];
  for (int i = 0; i < 3; ++i) {
    o_cell[i] = take((size_t)cap * 4);
    o_cost[i] = take((size_t)cap * 8);
  }
  size_t o_cnt = take((size_t)nb * 4), o_st = take(64), o_seed = take((size_t)2 * yw * 4 + 16), o_sc = take(64);
  PP_HIP(ctx, hipMalloc(&w->base, sz));
  char* b = (char*)w->base;
  for (int i = 0; i < 3; ++i) {
    w->q.cell[i] = (uint32_t*)(b + o_cell[i]);
    w->q.qcost[i] = (double*)(b + o_cost[i]);
  }
  w->q.cap = cap;
  w->q.count = (uint32_t*)(b + o_cnt);
  w->q.nbuckets = nb;
  w->q.status = (uint32_t*)(b + o_st);
  w->seed_cell = (uint32_t*)(b + o_seed);
  w->seed_count = (uint32_t*)(b + o_sc);
  w->q.seed_cell = w->seed_cell;
  w->q.seed_count = nullptr;
  PP_HIP(ctx, hipMemsetAsync(w->q.count, 0, (size_t)nb * 4, ctx->stream));
  PP_HIP(ctx, hipMemsetAsync(w->q.status, 0, 64, ctx->stream));
  uint32_t sc[3] = {0u, 0xFFFFFFFFu, 0u};
  PP_HIP(ctx, hipMemcpyAsync(w->seed_count, sc, 12, hipMemcpyHostToDevice, ctx->stream));
  return PP_OK;
}

// Runs bucket steps from k0 until no bucket at or above the current one holds work.
int32_t wave_run(pp_ctx* ctx, const WaveGrid& g, WaveWork& w, uint32_t k0, uint32_t seed_max,
                 uint64_t* iters, uint32_t status_out[4]) {
  const double diag = std::hypot(1.0, 1.0);  // u_cost((+-1, +-1)), astar.rs:213-215
  const unsigned block = 256;
  const unsigned grid = (unsigned)ctx->num_cus * 4;
  uint32_t k = k0;
  uint32_t st[4] = {0, 0, 0, 0};
  uint64_t steps = 0;
  const uint32_t batch = 128;
  for (;;) {
    for (uint32_t j = 0; j < batch; ++j, ++k) {
      if (k >= w.q.nbuckets) break;
      hipLaunchKernelGGL(k_wave_step, dim3(grid), dim3(block), 0, ctx->stream, g, w.q, k, diag);
      ++steps;
    }
    PP_HIP(ctx, hipGetLastError());
    PP_HIP(ctx, hipMemcpyAsync(st, w.q.status, 16, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st[1]) break;  // overflow
    uint32_t last = std::max(st[0], seed_max);
    if (k > last || k >= w.q.nbuckets) break;
  }
  if (iters) *iters = steps;
  std::memcpy(status_out, st, sizeof(st));
  if (st[1] & 2u) return PP_ERR_UNSUPPORTED;  // cost beyond the bucket table
  if (st[1] & 1u) return PP_ERR_CAPACITY;
  return PP_OK;
}

uint32_t queue_cap_guess(uint64_t xw, uint64_t yw) {
  uint64_t c = 64 * (xw + yw) + (1u << 16);
  return (uint32_t)std::min<uint64_t>(c, xw * yw + 16);
}

}  // namespace

// ---------------------------------------------------------------------------- entry points

extern "C" int32_t pp_wavefront_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, const uint8_t* d_obs,
                                    uint64_t goal_x, uint64_t goal_y, double* d_cost,
                                    uint64_t* iters) {
  if (!ctx || !d_obs || !d_cost) return PP_ERR_INVALID;
  if (xw == 0 || yw == 0 || xw * yw >= ((uint64_t)1 << 32) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  // the goal's own hmap write (astar.rs:195) indexes hmap[gx - minx][gy - miny]
  if (goal_x >= xw || goal_y >= yw) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t ncells = xw * yw;
  WaveGrid g{d_cost, d_obs, (uint32_t)yw, (uint32_t)xw, 0, (int64_t)xw, 0u, (uint32_t)xw};
  uint32_t cap = queue_cap_guess(xw, yw);
  for (int attempt = 0; attempt < 2; ++attempt) {
    WaveWork w;
    int32_t st = wave_alloc(ctx, ncells, (uint32_t)yw, cap, &w);
    if (st != PP_OK) return st;
    hipLaunchKernelGGL(k_fill_inf, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, d_cost, ncells);
    hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, g, w.q,
                       (uint32_t)(goal_x * yw + goal_y));
    uint32_t status[4];
    st = wave_run(ctx, g, w, 0, 0, iters, status);
    (void)hipFree(w.base);
    if (st != PP_ERR_CAPACITY) return st;
    cap = (uint32_t)std::min<uint64_t>(ncells + 16, 0xFFFFFFF0u);  // frontier larger than guessed
  }
  return PP_ERR_CAPACITY;
}

extern "C" int32_t pp_wavefront_slab_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, uint64_t x_begin,
                                         uint64_t x_end, const uint8_t* d_obs_slab,
                                         double* d_cost_slab, int32_t seed_goal, uint64_t goal_x,
                                         uint64_t goal_y, const double* d_halo_lo_new,
                                         const double* d_halo_hi_new, int32_t* changed_lo,
                                         int32_t* changed_hi, uint64_t* iters) {
  if (!ctx || !d_obs_slab || !d_cost_slab) return PP_ERR_INVALID;
  if (x_begin >= x_end || x_end > xw) return PP_ERR_INVALID;
  uint64_t ncols = x_end - x_begin + 2;
  if (ncols * yw >= ((uint64_t)1 << 32) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  if (seed_goal && (goal_x >= xw || goal_y >= yw)) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  WaveGrid g{d_cost_slab, d_obs_slab, (uint32_t)yw, (uint32_t)ncols, (int64_t)x_begin - 1,
             (int64_t)xw, 1u, (uint32_t)(ncols - 1)};
  uint32_t cap = queue_cap_guess(ncols, yw);
  WaveWork w;
  int32_t st = wave_alloc(ctx, ncols * yw, (uint32_t)yw, cap, &w);
  if (st != PP_OK) return st;
  hipLaunchKernelGGL(k_take_halo, dim3((unsigned)((yw + 255) / 256)), dim3(256), 0, ctx->stream, g,
                     d_halo_lo_new, d_halo_hi_new, w.seed_cell, w.seed_count, w.seed_count + 1);
  bool goal_here = seed_goal && goal_x >= x_begin && goal_x < x_end;
  if (goal_here)
    hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, g, w.q,
                       (uint32_t)((goal_x - x_begin + 1) * yw + goal_y));
  uint32_t sc[3];
  PP_HIP(ctx, hipMemcpyAsync(sc, w.seed_count, 12, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t status[4] = {0, 0, 0, 0};
  uint64_t steps = 0;
  if (sc[0] > 0 || goal_here) {
    uint32_t k0 = goal_here ? 0u : sc[1];
    w.q.seed_count = w.seed_count;
    st = wave_run(ctx, g, w, k0, sc[0] ? sc[2] : 0u, &steps, status);
  }
  (void)hipFree(w.base);
  if (changed_lo) *changed_lo = (int32_t)status[2];
  if (changed_hi) *changed_hi = (int32_t)status[3];
  if (iters) *iters = steps;
  return st;
}

// calc_parameters of astar.rs:232-242 on the obstacle lists scaled by 1/reso.
static void heuristic_dims(const pp_map* map, double reso, uint64_t dims[6]) {
  // min/max commute with the division by a positive reso
  double mnx = map->minx / reso, mxx = map->maxx / reso, mny = map->miny / reso, mxy = map->maxy / reso;
  dims[0] = pp_sat_usize(std::round(mnx));
  dims[1] = pp_sat_usize(std::round(mny));
  dims[2] = pp_sat_usize(std::round(mxx));
  dims[3] = pp_sat_usize(std::round(mxy));
  dims[4] = dims[2] - dims[0];
  dims[5] = dims[3] - dims[1];
}

int32_t pp_obsmap_dev(pp_ctx* ctx, const pp_map* map, double reso, double rr,
                          const uint64_t dims[6], uint8_t* d_obs) {
  uint64_t ncells = dims[4] * dims[5];
  PP_HIP(ctx, hipMemsetAsync(d_obs, 0, ncells, ctx->stream));
  uint64_t n = map->ox.size();
  hipLaunchKernelGGL(k_obsmap, dim3(pp_stream_grid(ctx, n, 256)), dim3(256), 0, ctx->stream, n,
                     map->d_ox, map->d_oy, reso, rr / reso, (int64_t)dims[0], (int64_t)dims[1],
                     (int64_t)dims[2], (int64_t)dims[3], dims[5], d_obs);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

int32_t pp_heuristic_checks(pp_ctx* ctx, const pp_map* map, double reso, uint64_t dims[6]) {
  if (!ctx || !map) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (!(reso > 0.0)) return PP_ERR_INVALID;
  heuristic_dims(map, reso, dims);
  if (dims[2] < dims[0] || dims[3] < dims[1]) return PP_ERR_BOUNDS;  // usize underflow panics in debug
  if (dims[4] >= ((uint64_t)1 << 31) || dims[5] >= ((uint64_t)1 << 31) ||
      dims[4] * dims[5] >= ((uint64_t)1 << 32))
    return PP_ERR_UNSUPPORTED;
  return PP_OK;
}

extern "C" int32_t pp_obsmap(pp_ctx* ctx, const pp_map* map, double reso, double rr,
                             uint8_t* obsmap, uint64_t cap, uint64_t dims_out[6]) {
  uint64_t dims[6];
  int32_t st = pp_heuristic_checks(ctx, map, reso, dims);
  if (st != PP_OK) return st;
  if (dims_out) std::memcpy(dims_out, dims, sizeof(dims));
  uint64_t ncells = dims[4] * dims[5];
  if (cap < ncells) return PP_ERR_CAPACITY;
  if (ncells == 0) return PP_OK;
  if (!obsmap) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  st = pp_scratch_reserve(ctx, ncells);
  if (st != PP_OK) return st;
  uint8_t* d_obs = (uint8_t*)ctx->scratch;
  st = pp_obsmap_dev(ctx, map, reso, rr, dims, d_obs);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipMemcpyAsync(obsmap, d_obs, ncells, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

// Device-resident form used by the planner: d_hmap (xw*yw doubles) and d_obs (xw*yw bytes) are
// caller-provided device buffers.
int32_t pp_heuristic_dev(pp_ctx* ctx, const pp_map* map, double gx, double gy, double reso,
                         double rr, const uint64_t dims[6], uint8_t* d_obs, double* d_hmap) {
  uint64_t gxi = pp_sat_usize(std::round(gx / reso));  // astar.rs:141-146
  uint64_t gyi = pp_sat_usize(std::round(gy / reso));
  if (gxi < dims[0] || gxi - dims[0] >= dims[4] || gyi < dims[1] || gyi - dims[1] >= dims[5])
    return PP_ERR_BOUNDS;
  int32_t st = pp_obsmap_dev(ctx, map, reso, rr, dims, d_obs);
  if (st != PP_OK) return st;
  return pp_wavefront_dev(ctx, dims[4], dims[5], d_obs, gxi - dims[0], gyi - dims[1], d_hmap, nullptr);
}

extern "C" int32_t pp_heuristic(pp_ctx* ctx, const pp_map* map, double gx, double gy, double reso,
                                double rr, double* hmap, uint64_t cap, uint64_t dims_out[6]) {
  uint64_t dims[6];
  int32_t st = pp_heuristic_checks(ctx, map, reso, dims);
  if (st != PP_OK) return st;
  if (dims_out) std::memcpy(dims_out, dims, sizeof(dims));
  uint64_t ncells = dims[4] * dims[5];
  if (cap < ncells) return PP_ERR_CAPACITY;
  if (ncells == 0) return PP_ERR_BOUNDS;
  if (!hmap) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  size_t cost_b = (ncells * 8 + 255) & ~(size_t)255;
  st = pp_scratch_reserve(ctx, cost_b + ncells);
  if (st != PP_OK) return st;
  double* d_hmap = (double*)ctx->scratch;
  uint8_t* d_obs = (uint8_t*)ctx->scratch + cost_b;
  st = pp_heuristic_dev(ctx, map, gx, gy, reso, rr, dims, d_obs, d_hmap);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipMemcpyAsync(hmap, d_hmap, ncells * 8, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}
