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

// ------------------------------------------------------------------ K6: tiled relaxation
//
// The reference's Dijkstra (astar.rs:153-198) computes the unique fixpoint of
//   d[goal] = 0,  d[v] = min over valid neighbours n of fl(d[n] + w(n, v)),  w in {1, hypot(1,1)}
// (each fl(d + w) is monotone in d, so relaxing from any upper bounds in any order converges to the
// same bits).  The grid is cut into 32x32-cell tiles; a round relaxes every active tile to its
// own fixpoint against the current values of the surrounding ring of cells, and a tile whose edge
// cells moved activates the neighbours that see them.  One wave owns one tile: each lane keeps a
// 4x4 patch in registers and sweeps it forward and backward (a two-pass chamfer transform), the
// patches trade edge values through LDS, and the wave repeats until no cell moves.

constexpr int kTile = 32;                    // cells per tile side
constexpr int kPps = 8;                      // 4x4 patches per tile side
constexpr int kRing = kPps + 2;              // patch grid plus the ring outside the tile
constexpr int kSlot = kRing * kRing;         // doubles per patch-cell slot
constexpr int kWaveLds = 16 * kSlot;         // doubles of LDS per wave
constexpr int kWavesPerBlock = 4;

struct TileGrid {
  double* cost;          // cost[lx * yw + y]: x-major, one x-column of yw cells contiguous
  const uint8_t* obs;    // same indexing
  uint64_t yw;
  uint64_t ncols;        // stored columns (a slab includes its two halo columns)
  int64_t gx_off;        // global x of stored column 0
  int64_t xw;            // global grid width
  uint64_t own_lo, own_hi;  // stored columns [own_lo, own_hi) are relaxed, the rest only read
  uint32_t ntx, nty;     // tiles over the owned columns
  int64_t goal_cell;     // index of the goal cell in cost[], -1 when it is not stored here
};

struct TileWork {
  uint32_t* list[2];     // tiles to relax in even / odd rounds
  uint32_t* flag[2];     // "already listed" per tile, per parity
  uint32_t* count;       // [3]: list lengths of rounds r, r+1, r+2 (mod 3)
  uint32_t* status;      // [0] tile visits, [1] sweep pairs, [2] first owned column moved, [3] last
};

__device__ inline double pp_min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline double pp_max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ inline void tile_activate(const TileGrid& g, const TileWork& w, int64_t tx, int64_t ty,
                                     uint32_t round) {
  if (tx < 0 || ty < 0 || tx >= (int64_t)g.ntx || ty >= (int64_t)g.nty) return;
  const uint32_t t = (uint32_t)tx * g.nty + (uint32_t)ty;
  if (atomicExch(&w.flag[round & 1][t], 1u) == 0u) {
    const uint32_t s = atomicAdd(&w.count[round % 3], 1u);
    w.list[round & 1][s] = t;
  }
}

__device__ inline double tile_ring_value(const TileGrid& g, int64_t x, int64_t y) {
  if (x < 0 || y < 0 || x >= (int64_t)g.ncols || y >= (int64_t)g.yw)
    return __longlong_as_double(0x7FF0000000000000ll);
  return g.cost[(uint64_t)x * g.yw + (uint64_t)y];
}

__global__ void __launch_bounds__(64 * kWavesPerBlock)
k_tile_round(TileGrid g, TileWork w, uint32_t round, double diag) {
  __shared__ double lds_all[kWavesPerBlock * kWaveLds];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  double* lds = lds_all + wv * kWaveLds;
  const uint32_t n = w.count[round % 3];
  if (blockIdx.x == 0 && threadIdx.x == 0) w.count[(round + 2) % 3] = 0;
  const uint32_t* list = w.list[round & 1];
  const int px = (int)(lane >> 3), py = (int)(lane & 7u);
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const int own_slot = (px + 1) * kRing + (py + 1);

  for (uint32_t it = blockIdx.x * kWavesPerBlock + wv; it < n; it += gridDim.x * kWavesPerBlock) {
    const uint32_t t = list[it];
    if (lane == 0) w.flag[round & 1][t] = 0;
    const uint32_t tx = t / g.nty, ty = t - tx * g.nty;
    const int64_t x0 = (int64_t)g.own_lo + (int64_t)tx * kTile, y0 = (int64_t)ty * kTile;

    // e[1..4][1..4] = the lane's patch, the rest of e = the cells around it.  floor_[][] is the
    // value a cell may not go below: 0 for a free cell, +inf for a blocked one (check_node,
    // astar.rs:200-211), its own value for a stored cell this rank does not own.
    double e[6][6], floor_[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t x = x0 + 4 * px + i, y = y0 + 4 * py + j;
        double v = INF, fl = INF;
        if (x < (int64_t)g.ncols && y < (int64_t)g.yw) {
          const uint64_t idx = (uint64_t)x * g.yw + (uint64_t)y;
          v = g.cost[idx];
          if (x >= (int64_t)g.own_hi) {
            fl = v;
          } else {
            const bool blocked = g.obs[idx] != 0 || x + g.gx_off <= 0 || y <= 0;
            fl = (blocked && (int64_t)idx != g.goal_cell) ? INF : 0.0;
          }
        }
        e[i + 1][j + 1] = v;
        floor_[i][j] = fl;
        lds[(i * 4 + j) * kSlot + own_slot] = v;
      }
    }
    // the ring of cells just outside the tile
    if (lane < 34) {
      const int o = (int)lane - 1, q = (o + 4) >> 2, j = (o + 4) & 3;
      lds[(3 * 4 + j) * kSlot + 0 * kRing + q] = tile_ring_value(g, x0 - 1, y0 + o);
      lds[(0 * 4 + j) * kSlot + (kRing - 1) * kRing + q] = tile_ring_value(g, x0 + kTile, y0 + o);
    }
    if (lane < 32) {
      const int o = (int)lane, prow = (o >> 2) + 1, i = o & 3;
      lds[(i * 4 + 3) * kSlot + prow * kRing + 0] = tile_ring_value(g, x0 + o, y0 - 1);
      lds[(i * 4 + 0) * kSlot + prow * kRing + (kRing - 1)] = tile_ring_value(g, x0 + o, y0 + kTile);
    }

    uint32_t sweeps = 0;
    for (;;) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ii = 0; ii < 6; ++ii) {
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
          if (ii >= 1 && ii <= 4 && jj >= 1 && jj <= 4) continue;
          const int dx = ii == 0 ? -1 : (ii == 5 ? 1 : 0), dy = jj == 0 ? -1 : (jj == 5 ? 1 : 0);
          const int si = ii == 0 ? 3 : (ii == 5 ? 0 : ii - 1), sj = jj == 0 ? 3 : (jj == 5 ? 0 : jj - 1);
          e[ii][jj] = lds[(si * 4 + sj) * kSlot + own_slot + dx * kRing + dy];
        }
      }
#pragma unroll
      for (int i = 1; i <= 4; ++i) {
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
          double v = e[i][j];
          v = pp_min_f64(v, e[i - 1][j - 1] + diag);
          v = pp_min_f64(v, e[i - 1][j] + 1.0);
          v = pp_min_f64(v, e[i - 1][j + 1] + diag);
          v = pp_min_f64(v, e[i][j - 1] + 1.0);
          e[i][j] = pp_max_f64(v, floor_[i - 1][j - 1]);
        }
      }
#pragma unroll
      for (int i = 4; i >= 1; --i) {
#pragma unroll
        for (int j = 4; j >= 1; --j) {
          double v = e[i][j];
          v = pp_min_f64(v, e[i + 1][j + 1] + diag);
          v = pp_min_f64(v, e[i + 1][j] + 1.0);
          v = pp_min_f64(v, e[i + 1][j - 1] + diag);
          v = pp_min_f64(v, e[i][j + 1] + 1.0);
          e[i][j] = pp_max_f64(v, floor_[i - 1][j - 1]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bool moved = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double* slot = &lds[(i * 4 + j) * kSlot + own_slot];
          const double v = e[i + 1][j + 1];
          moved |= __double_as_longlong(*slot) != __double_as_longlong(v);
          *slot = v;
        }
      }
      ++sweeps;
      if (__ballot(moved) == 0ull) break;
    }

    // write back what moved; cells on the tile's edge wake the neighbours that read them
    bool m_xlo = false, m_xhi = false, m_ylo = false, m_yhi = false;
    bool c00 = false, c01 = false, c10 = false, c11 = false, first_col = false, last_col = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int li = 4 * px + i, lj = 4 * py + j;
        const int64_t x = x0 + li, y = y0 + lj;
        if (x < (int64_t)g.own_hi && y < (int64_t)g.yw) {
          const uint64_t idx = (uint64_t)x * g.yw + (uint64_t)y;
          const double v = e[i + 1][j + 1];
          if (__double_as_longlong(g.cost[idx]) != __double_as_longlong(v)) {
            g.cost[idx] = v;
            m_xlo |= li == 0;
            m_xhi |= li == kTile - 1;
            m_ylo |= lj == 0;
            m_yhi |= lj == kTile - 1;
            c00 |= li == 0 && lj == 0;
            c01 |= li == 0 && lj == kTile - 1;
            c10 |= li == kTile - 1 && lj == 0;
            c11 |= li == kTile - 1 && lj == kTile - 1;
            first_col |= x == (int64_t)g.own_lo;
            last_col |= x + 1 == (int64_t)g.own_hi;
          }
        }
      }
    }
    const bool b_xlo = __ballot(m_xlo) != 0ull, b_xhi = __ballot(m_xhi) != 0ull;
    const bool b_ylo = __ballot(m_ylo) != 0ull, b_yhi = __ballot(m_yhi) != 0ull;
    const bool b00 = __ballot(c00) != 0ull, b01 = __ballot(c01) != 0ull;
    const bool b10 = __ballot(c10) != 0ull, b11 = __ballot(c11) != 0ull;
    const bool b_first = __ballot(first_col) != 0ull, b_last = __ballot(last_col) != 0ull;
    if (lane == 0) {
      const int64_t X = tx, Y = ty;
      if (b_xlo) tile_activate(g, w, X - 1, Y, round + 1);
      if (b_xhi) tile_activate(g, w, X + 1, Y, round + 1);
      if (b_ylo) tile_activate(g, w, X, Y - 1, round + 1);
      if (b_yhi) tile_activate(g, w, X, Y + 1, round + 1);
      if (b00) tile_activate(g, w, X - 1, Y - 1, round + 1);
      if (b01) tile_activate(g, w, X - 1, Y + 1, round + 1);
      if (b10) tile_activate(g, w, X + 1, Y - 1, round + 1);
      if (b11) tile_activate(g, w, X + 1, Y + 1, round + 1);
      if (b_first) w.status[2] = 1;
      if (b_last) w.status[3] = 1;
      atomicAdd(&w.status[0], 1u);
      atomicAdd(&w.status[1], sweeps);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void k_fill_inf(double* p, uint64_t n) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = __longlong_as_double(0x7FF0000000000000ll);
}

// The goal is inserted without check_node (astar.rs:155): cost 0 even on an obstacle or border.
__global__ void k_seed_goal(TileGrid g, TileWork w) {
  g.cost[g.goal_cell] = 0.0;
  const uint64_t lx = (uint64_t)g.goal_cell / g.yw, y = (uint64_t)g.goal_cell - lx * g.yw;
  tile_activate(g, w, (int64_t)((lx - g.own_lo) / kTile), (int64_t)(y / kTile), 0);
}

// Halo columns handed in by the neighbour ranks: where the new value is lower than the stored one,
// store it and wake the tiles whose ring contains the cell.
__global__ void k_take_halo(TileGrid g, TileWork w, const double* __restrict__ lo_new,
                            const double* __restrict__ hi_new) {
  const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= g.yw) return;
  for (int side = 0; side < 2; ++side) {
    const double* src = side ? hi_new : lo_new;
    if (!src) continue;
    const uint64_t col = side ? g.own_hi : g.own_lo - 1;
    const uint64_t cell = col * g.yw + y;
    const double v = src[y];
    if (v < g.cost[cell]) {
      g.cost[cell] = v;
      // an owned tile column holds `col` itself when the owned width is not a multiple of the
      // tile size; otherwise the last tile column sees it in its ring
      const int64_t tx = side ? (int64_t)g.ntx - 1 : 0;
      const int64_t ty = (int64_t)(y / kTile);
      tile_activate(g, w, tx, ty, 0);
      if (y % kTile == 0) tile_activate(g, w, tx, ty - 1, 0);
      if (y % kTile == kTile - 1) tile_activate(g, w, tx, ty + 1, 0);
    }
  }
}

struct TileRun {
  TileWork w{};
  uint32_t ntiles = 0;
};

// Work lists live in a context-owned buffer (grown on demand) so repeated calls do not allocate.
int32_t tile_prepare(pp_ctx* ctx, TileGrid& g, TileRun* run) {
  g.ntx = (uint32_t)((g.own_hi - g.own_lo + kTile - 1) / kTile);
  g.nty = (uint32_t)((g.yw + kTile - 1) / kTile);
  const uint64_t ntiles = (uint64_t)g.ntx * g.nty;
  if (ntiles == 0 || ntiles >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  const size_t arr = ((size_t)ntiles * 4 + 255) & ~(size_t)255;
  const size_t need = 4 * arr + 256;
  if (need > ctx->wave_bytes) {
    if (ctx->wave_buf) PP_HIP(ctx, hipFree(ctx->wave_buf));
    ctx->wave_buf = nullptr;
    ctx->wave_bytes = 0;
    PP_HIP(ctx, hipMalloc(&ctx->wave_buf, need));
    ctx->wave_bytes = need;
  }
  char* b = (char*)ctx->wave_buf;
  run->ntiles = (uint32_t)ntiles;
  run->w.list[0] = (uint32_t*)b;
  run->w.list[1] = (uint32_t*)(b + arr);
  run->w.flag[0] = (uint32_t*)(b + 2 * arr);
  run->w.flag[1] = (uint32_t*)(b + 3 * arr);
  run->w.count = (uint32_t*)(b + 4 * arr);
  run->w.status = (uint32_t*)(b + 4 * arr + 64);
  PP_HIP(ctx, hipMemsetAsync(b + 2 * arr, 0, 2 * arr + 256, ctx->stream));
  return PP_OK;
}

// Rounds until no tile is listed.  The host looks at the list length once per batch of rounds;
// a round over an empty list costs one empty launch.
int32_t tile_run(pp_ctx* ctx, const TileGrid& g, TileRun& run, uint64_t* rounds_out,
                 uint32_t status_out[4]) {
  const double diag = std::hypot(1.0, 1.0);  // u_cost((+-1, +-1)), astar.rs:213-215
  const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)ctx->num_cus * 3,
                                                     (run.ntiles + kWavesPerBlock - 1) / kWavesPerBlock);
  uint32_t round = 0;
  uint32_t counts[3];
  uint32_t batch = 8;
  for (;;) {
    for (uint32_t j = 0; j < batch; ++j, ++round)
      hipLaunchKernelGGL(k_tile_round, dim3(grid), dim3(64 * kWavesPerBlock), 0, ctx->stream, g, run.w,
                         round, diag);
    PP_HIP(ctx, hipGetLastError());
    PP_HIP(ctx, hipMemcpyAsync(counts, run.w.count, 12, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (counts[round % 3] == 0) break;
    if (round > (1u << 30)) return PP_ERR_UNSUPPORTED;
    batch = std::min<uint32_t>(batch * 2, 64);
  }
  PP_HIP(ctx, hipMemcpyAsync(status_out, run.w.status, 16, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (rounds_out) *rounds_out = round;
  return PP_OK;
}

}  // namespace

// ---------------------------------------------------------------------------- entry points

extern "C" int32_t pp_wavefront_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, const uint8_t* d_obs,
                                    uint64_t goal_x, uint64_t goal_y, double* d_cost,
                                    uint64_t* iters) {
  if (!ctx || !d_obs || !d_cost) return PP_ERR_INVALID;
  if (xw == 0 || yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  // the goal's own hmap write (astar.rs:195) indexes hmap[gx - minx][gy - miny]
  if (goal_x >= xw || goal_y >= yw) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  TileGrid g{d_cost, d_obs, yw, xw, 0, (int64_t)xw, 0, xw, 0, 0, (int64_t)(goal_x * yw + goal_y)};
  TileRun run;
  int32_t st = tile_prepare(ctx, g, &run);
  if (st != PP_OK) return st;
  hipLaunchKernelGGL(k_fill_inf, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, d_cost, xw * yw);
  hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, g, run.w);
  uint32_t status[4];
  return tile_run(ctx, g, run, iters, status);
}

extern "C" int32_t pp_wavefront_slab_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, uint64_t x_begin,
                                         uint64_t x_end, const uint8_t* d_obs_slab,
                                         double* d_cost_slab, int32_t seed_goal, uint64_t goal_x,
                                         uint64_t goal_y, const double* d_halo_lo_new,
                                         const double* d_halo_hi_new, int32_t* changed_lo,
                                         int32_t* changed_hi, uint64_t* iters) {
  if (!ctx || !d_obs_slab || !d_cost_slab) return PP_ERR_INVALID;
  if (x_begin >= x_end || x_end > xw) return PP_ERR_INVALID;
  if (yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  if (seed_goal && (goal_x >= xw || goal_y >= yw)) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t ncols = x_end - x_begin + 2;
  const bool goal_here = seed_goal && goal_x >= x_begin && goal_x < x_end;
  TileGrid g{d_cost_slab, d_obs_slab, yw, ncols, (int64_t)x_begin - 1, (int64_t)xw, 1, ncols - 1, 0, 0,
             goal_x >= x_begin && goal_x < x_end ? (int64_t)((goal_x - x_begin + 1) * yw + goal_y) : -1};
  TileRun run;
  int32_t st = tile_prepare(ctx, g, &run);
  if (st != PP_OK) return st;
  if (d_halo_lo_new || d_halo_hi_new)
    hipLaunchKernelGGL(k_take_halo, dim3((unsigned)((yw + 255) / 256)), dim3(256), 0, ctx->stream, g,
                       run.w, d_halo_lo_new, d_halo_hi_new);
  if (goal_here) hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, g, run.w);
  uint32_t status[4] = {0, 0, 0, 0};
  st = tile_run(ctx, g, run, iters, status);
  if (changed_lo) *changed_lo = (int32_t)status[2];
  if (changed_hi) *changed_hi = (int32_t)status[3];
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
