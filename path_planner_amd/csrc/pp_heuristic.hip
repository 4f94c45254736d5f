// pp_heuristic.hip -- holonomic heuristic: obstacle raster (calc_obsmap, astar.rs:244-261) and
// cost-to-go grid (the Dijkstra of astar.rs:153-198) as a tiled relaxation.
//
// The cost-to-go is the unique fixpoint d[v] = min over valid neighbours n of fl(d[n] + w),
// w in {1, hypot(1,1)}, d[goal] = 0: every addition the serial Dijkstra performs along a
// shortest path is performed here too, in fp64, and min is order-free, so the grid is
// bit-identical to the reference's whatever the schedule.  The schedule here: 64x64-cell tiles
// relaxed by one workgroup each (register patches, chamfer sweeps, rims traded through LDS), in
// rounds ordered by the lowest value waiting at each tile.
#include <cstdio>
#include <algorithm>
#include <system_error>
#include <thread>
#include <vector>
#include <cstring>

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
// same bits).  The grid is cut into 64x64-cell tiles; a round relaxes every listed tile to its
// own fixpoint against the current values of the surrounding ring of cells, and a tile wakes a
// neighbour when one of its edge cells moved to a value that can still lower a cell of that
// neighbour.  One workgroup of four waves owns one tile, a
// wave one 32x32 quadrant of it: each lane keeps a 4x4 patch in registers and sweeps it forward
// and backward (a two-pass chamfer transform), the patches trade rim values through LDS, and the
// workgroup repeats until no cell moves.  A value crosses one patch per sweep, so the cost of a
// visit is the number of patches a front has to cross times the f64 operations of a sweep (the
// sweep is bound by VALU issue, not by latency); the larger tile halves the number of rounds a
// front needs to cross the grid.
//
// Rounds are ordered by cost, Dijkstra-like at tile granularity: a woken tile carries a key (the
// lowest value that newly appeared next to it) and is listed only once the key is within `band`
// of the lowest pending key.  Relaxing a tile before its inputs have settled would push
// provisional values downstream that every later correction has to chase.
//
// Working layout: costs live in a blocked array, one contiguous 8 KiB block per 32x32 quadrant
// with the 16 cells of a lane's patch interleaved across lanes (so every load is a coalesced 1 KiB), and
// the obstacle raster is folded into one mask word per lane.  The cells of a block's first and
// last row sit in 32 different cache lines there, so each block also keeps a compact copy of those
// two rows for the tiles above and below to read.  The caller's dense x-major arrays are packed
// into that layout before the rounds and unpacked after them.

constexpr int kTile = 32;                    // cells per block side (one wave's quadrant of a tile)
constexpr int kPps = 8;                      // 4x4 patches per block side
constexpr int kBlk = kTile * kTile;          // doubles per block
constexpr unsigned long long kInfBits = 0x7FF0000000000000ull;
#ifndef PP_TILE_WAVES
#define PP_TILE_WAVES 3      // waves per SIMD the register allocation of k_tile_round aims at
#endif

// LDS slot of patch cell (i, j); only cells on the patch rim are ever read by other lanes.
__device__ constexpr int rim_slot(int i, int j) {
  return i == 0 ? j : (i == 3 ? 4 + j : (j == 0 ? 7 + i : 9 + i));
}
__device__ constexpr bool on_rim(int i, int j) { return i == 0 || i == 3 || j == 0 || j == 3; }
// Offset of patch cell k = 4i + j of `lane` inside a tile block: cells 2m, 2m+1 of one lane are
// adjacent (one 16-byte access), the 64 lanes of a pair are consecutive.
__device__ __host__ constexpr uint32_t blk_off(uint32_t k, uint32_t lane) {
  return ((k >> 1) * 64u + lane) * 2u + (k & 1u);
}

struct DenseView {       // the caller's arrays
  double* cost;          // cost[lx * yw + y]: x-major, one x-column of yw cells contiguous
  const uint8_t* obs;    // same indexing
  uint64_t yw;
  uint64_t ncols;        // stored columns (a slab includes its two halo columns)
  int64_t gx_off;        // global x of stored column 0
  uint64_t own_lo, own_hi;  // stored columns [own_lo, own_hi) are relaxed, the rest only read
  int64_t goal_x, goal_y;   // stored column / row of the goal, -1 when it is not in an owned column
  // Several independent grids over the same obstacle raster, relaxed together: plane p occupies
  // stored columns [p * plane_cols, p * plane_cols + plane_cols - 1) followed by one blocked
  // separator column, its costs start at cost + p * cost_stride and its goal is goals[2p], [2p+1].
  uint32_t planes;          // 0: a single grid, none of the fields below is used
  uint64_t plane_cols;
  uint64_t cost_stride;
  const int64_t* goals;
};

// Stored column -> (plane, column inside the plane's dense arrays); false for a separator column.
__device__ inline bool dense_column(const DenseView& d, int64_t x, uint32_t& plane, int64_t& xl) {
  if (d.planes == 0) {
    plane = 0;
    xl = x;
    return true;
  }
  plane = (uint32_t)((uint64_t)x / d.plane_cols);
  xl = x - (int64_t)(plane * d.plane_cols);
  return xl + 1 < (int64_t)d.plane_cols;
}

struct TileGrid {
  double* blk;           // tile t at blk + t * kBlk
  uint32_t* mask;        // mask[t * 64 + lane]: bit k = cell k blocked, bit 16 + k = cell k fixed
  double* yrim;          // yrim[(t * 2 + s) * 32 + lx]: a copy of block t's rows ly = 0 (s = 0) and ly = 31 (s = 1),
                         // kept current for the rows on a scheduling tile's edge (even block rows: s = 0, odd: s = 1).
                         // In the block itself the 32 cells of such a row sit in 32 different cache lines;
                         // the tiles above and below read them from here
  uint32_t* stamp;       // stamp[t]: 0 = block t has not been written yet: every cell of it is +inf and neither its
                         // costs nor its row copies are initialised memory; 1 = written before the rounds; r + 2 =
                         // first written by round r.  A round treats a block first written in the same launch
                         // as still unwritten (+inf is what the block held when the round began), so it never
                         // reads a block that is half way through its first write
  uint32_t ntx, nty;     // 32x32 blocks over the stored columns
  uint32_t nsx, nsy;     // scheduling tiles: 2x2 blocks = 64x64 cells, one workgroup each
};
constexpr uint32_t kAfterRounds = 0xFFFFFFF0u;  // "round" for readers that run when no round is in flight
__device__ inline bool tile_block_live(const TileGrid& g, uint32_t t, uint32_t round) {
  const uint32_t s = g.stamp[t];
  return s != 0u && s <= round + 1u;
}

struct TileWork {
  uint32_t* list;        // tiles to relax this round
  uint32_t* bitmap;      // woken tiles
  uint32_t* count;       // [3]: list lengths of rounds r, r+1, r+2 (mod 3)
  unsigned long long* key;      // per tile: bit pattern of the lowest pending value
  unsigned long long* minkey;   // [3]: lowest pending key for rounds r, r+1, r+2 (mod 3)
  uint32_t* status;      // [0] tile visits, [1] sweep pairs, [2] first owned column moved, [3] last
  uint32_t ntiles;
  double band;
  double limit;          // woken tiles whose key is above it are left pending (+inf: none)
  uint64_t own_first, own_last;  // stored columns whose change is reported in status[2], [3]
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

// Wakes tile (tx, ty) with key `kbits` (bit patterns of non-negative doubles order like the
// doubles).  Neither atomic returns a value, so the caller does not wait on them; the caller folds
// kbits into minkey[] of the next round.
__device__ inline void tile_wake(const TileGrid& g, const TileWork& w, int64_t tx, int64_t ty,
                                 unsigned long long kbits) {
  if (tx < 0 || ty < 0 || tx >= (int64_t)g.nsx || ty >= (int64_t)g.nsy) return;
  const uint32_t t = (uint32_t)tx * g.nsy + (uint32_t)ty;
  atomicMin(&w.key[t], kbits);
  atomicOr(&w.bitmap[t >> 5], 1u << (t & 31u));
}

// Address of cell (x, y) (stored column, row) in the blocked array; +inf outside the tile grid.
__device__ inline double tile_cell_value(const TileGrid& g, int64_t x, int64_t y) {
  if (x < 0 || y < 0 || x >= (int64_t)g.ntx * kTile || y >= (int64_t)g.nty * kTile)
    return __longlong_as_double((long long)kInfBits);
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  if (!tile_block_live(g, t, kAfterRounds)) return __longlong_as_double((long long)kInfBits);
  return g.blk[(uint64_t)t * kBlk + blk_off((lx & 3u) * 4u + (ly & 3u), (lx >> 2) * 8u + (ly >> 2))];
}

// The same, and whether a visit of the cell's own tile could lower it (neither blocked nor a
// source-only cell, inside the grid).
__device__ inline double tile_cell_value(const TileGrid& g, int64_t x, int64_t y, bool& relaxable,
                                         uint32_t round) {
  relaxable = false;
  if (x < 0 || y < 0 || x >= (int64_t)g.ntx * kTile || y >= (int64_t)g.nty * kTile)
    return __longlong_as_double((long long)kInfBits);
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  const uint32_t k = (lx & 3u) * 4u + (ly & 3u), ln = (lx >> 2) * 8u + (ly >> 2);
  relaxable = ((g.mask[(uint64_t)t * 64 + ln] >> k) & 0x10001u) == 0u;
  if (!tile_block_live(g, t, round)) return __longlong_as_double((long long)kInfBits);
  return g.blk[(uint64_t)t * kBlk + blk_off(k, ln)];
}

// Every store to a cell of row ly = 0 or 31 of a block is repeated in the block's row copy.
__device__ inline void tile_yrim_store(const TileGrid& g, uint32_t t, uint32_t lx, uint32_t ly, double v) {
  if (ly == 0u) g.yrim[((uint64_t)t * 2) * 32 + lx] = v;
  if (ly == 31u) g.yrim[((uint64_t)t * 2 + 1) * 32 + lx] = v;
}
// tile_cell_value for a cell of such a row, read from the copy.
__device__ inline double tile_yrim_value(const TileGrid& g, int64_t x, int64_t y, bool& relaxable,
                                         uint32_t round) {
  relaxable = false;
  if (x < 0 || y < 0 || x >= (int64_t)g.ntx * kTile || y >= (int64_t)g.nty * kTile)
    return __longlong_as_double((long long)kInfBits);
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  const uint32_t k = (lx & 3u) * 4u + (ly & 3u), ln = (lx >> 2) * 8u + (ly >> 2);
  relaxable = ((g.mask[(uint64_t)t * 64 + ln] >> k) & 0x10001u) == 0u;
  if (!tile_block_live(g, t, round)) return __longlong_as_double((long long)kInfBits);
  return g.yrim[((uint64_t)t * 2 + (ly ? 1u : 0u)) * 32 + lx];
}

// Dense -> blocked.  One wave per tile, consecutive waves take tiles that are consecutive in y so
// the strided rows of the dense arrays are read in long runs.  A lane's patch is four runs of four
// consecutive cells of a dense column: a run that lies inside the grid and is suitably aligned is
// moved in one piece (a dword of the raster, two 16-byte pieces of the costs).
// With init_inf the costs all start at +inf, and a block of +inf is not written at all: its stamp
// says so and the first visit that moves one of its cells writes it whole.  Blocks that hold
// another rank's columns are the exception (the halo intake compares against their cells).
__global__ void __launch_bounds__(256)
k_tile_pack(DenseView d, TileGrid g, uint32_t ntiles, int init_inf) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t px = lane >> 3, py = lane & 7u;
  const double INF = __longlong_as_double((long long)kInfBits);
  for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += gridDim.x * 4) {
    const uint32_t tx = t / g.nty, ty = t - tx * g.nty;
    const int64_t y0 = (int64_t)ty * kTile + 4 * py;
    const uint64_t bx0 = (uint64_t)tx * kTile, bx1 = std::min<uint64_t>(bx0 + kTile, d.ncols);
    const bool eager = !init_inf || bx0 < d.own_lo || bx1 > d.own_hi;
    uint32_t m = 0;
    double v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t x = (int64_t)tx * kTile + 4 * px + i;
      uint32_t plane = 0;
      int64_t xl = 0;
      const bool col = x < (int64_t)d.ncols && y0 < (int64_t)d.yw && dense_column(d, x, plane, xl);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i * 4 + j] = INF;
      if (!col) {
        m |= 0xFu << (4 * i);
        continue;
      }
      const uint64_t idx = (uint64_t)xl * d.yw + (uint64_t)y0;
      const bool whole = y0 + 3 < (int64_t)d.yw;
      const bool other = x < (int64_t)d.own_lo || x >= (int64_t)d.own_hi;  // another rank's column
      if (!init_inf) {
        const double* src = d.cost + plane * d.cost_stride + idx;
        if (whole && (((uintptr_t)src) & 15u) == 0u) {
          const double2 p0 = *(const double2*)src, p1 = *(const double2*)(src + 2);
          v[i * 4] = p0.x;
          v[i * 4 + 1] = p0.y;
          v[i * 4 + 2] = p1.x;
          v[i * 4 + 3] = p1.y;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (y0 + j < (int64_t)d.yw) v[i * 4 + j] = src[j];
        }
      }
      uint32_t ob = 0;  // byte j: the raster's cell (x, y0 + j)
      if (!other) {
        const uint8_t* src = d.obs + idx;
        if (whole && (((uintptr_t)src) & 3u) == 0u) {
          ob = *(const uint32_t*)src;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (y0 + j < (int64_t)d.yw) ob |= (uint32_t)src[j] << (8 * j);
        }
      }
      const int64_t gx = d.planes ? d.goals[2 * plane] : d.goal_x;
      const int64_t gy = d.planes ? d.goals[2 * plane + 1] : d.goal_y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = i * 4 + j;
        const int64_t y = y0 + j;
        if (y >= (int64_t)d.yw) {
          m |= 1u << k;
        } else if (other) {
          m |= 1u << (16 + k);  // a source, never relaxed here
        } else {
          // check_node (astar.rs:200-211); the goal is inserted without it (astar.rs:155)
          const bool blocked = ((ob >> (8 * j)) & 0xFFu) != 0u || xl + d.gx_off <= 0 || y <= 0;
          if (blocked && !(xl == gx && y == gy)) m |= 1u << k;
        }
      }
    }
    g.mask[(uint64_t)t * 64 + lane] = m;
    if (lane == 0u) g.stamp[t] = eager ? 1u : 0u;
    if (!eager) continue;
    double* b = g.blk + (uint64_t)t * kBlk;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
      *(double2*)(b + blk_off(2 * k2, lane)) = make_double2(v[2 * k2], v[2 * k2 + 1]);
    if (py == 0u || py == 7u) {  // the block's first / last row
      double* yr = g.yrim + ((uint64_t)t * 2 + (py ? 1u : 0u)) * 32 + 4 * px;
      const int j = py ? 3 : 0;
      *(double2*)yr = make_double2(v[0 * 4 + j], v[1 * 4 + j]);
      *(double2*)(yr + 2) = make_double2(v[2 * 4 + j], v[3 * 4 + j]);
    }
  }
}

// Blocked -> dense, owned and halo columns alike.
__global__ void __launch_bounds__(256) k_tile_unpack(DenseView d, TileGrid g, uint32_t ntiles) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t px = lane >> 3, py = lane & 7u;
  for (uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += gridDim.x * 4) {
    const uint32_t tx = t / g.nty, ty = t - tx * g.nty;
    const int64_t y0 = (int64_t)ty * kTile + 4 * py;
    const double* b = g.blk + (uint64_t)t * kBlk;
    const bool live = g.stamp[t] != 0u;  // a block nobody wrote is +inf throughout
    double v[16];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      double2 p = make_double2(__longlong_as_double((long long)kInfBits), __longlong_as_double((long long)kInfBits));
      if (live) p = *(const double2*)(b + blk_off(2 * k2, lane));
      v[2 * k2] = p.x;
      v[2 * k2 + 1] = p.y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t x = (int64_t)tx * kTile + 4 * px + i;
      uint32_t plane;
      int64_t xl;
      if (!(x < (int64_t)d.ncols && y0 < (int64_t)d.yw && dense_column(d, x, plane, xl))) continue;
      double* dst = d.cost + plane * d.cost_stride + (uint64_t)xl * d.yw + (uint64_t)y0;
      if (y0 + 3 < (int64_t)d.yw && (((uintptr_t)dst) & 15u) == 0u) {
        *(double2*)dst = make_double2(v[i * 4], v[i * 4 + 1]);
        *(double2*)(dst + 2) = make_double2(v[i * 4 + 2], v[i * 4 + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (y0 + j < (int64_t)d.yw) dst[j] = v[i * 4 + j];
      }
    }
  }
}

// Lists the woken tiles whose key is within the band of the lowest pending key; the others stay
// woken and their keys go into the next round's minimum.  Tiles whose key is above w.limit are
// never listed: a key is the lowest value that newly appeared next to the tile, values only grow
// along a path, so when the rounds run dry every cell whose final value is at most the limit has
// it, and a later run with a higher limit picks up from the pending tiles.  One lane per tile, so the key loads of
// a crowded stretch of the bitmap are all in flight together.
// kListThreads: 1024 for large grids (fewer atomics in the crowded rounds: 16384^2 20.1 -> 19.3 ms),
// 256 for small ones, whose rounds are short and pay for the wider workgroup's barriers instead
// (4096^2: 3.38 against 3.44 ms).
#ifdef PP_LAB_TRACE
__device__ double lab_trace[8192][2];  // per round: lowest pending key, tiles listed
#endif
template <int kListThreads>
__global__ void __launch_bounds__(kListThreads) k_tile_list(TileGrid g, TileWork w, uint32_t round) {
  // one list-space reservation and one waiting-key minimum per workgroup: the atomics of a crowded
  // round all hit the same two words
  __shared__ uint32_t s_take[kListThreads / 64], s_base;
  __shared__ unsigned long long s_wait[kListThreads / 64];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const double theta = __longlong_as_double((long long)w.minkey[round % 3]) + w.band;
  if (t == 0) {
    w.count[(round + 1) % 3] = 0;
    w.minkey[(round + 2) % 3] = kInfBits;
#ifdef PP_LAB_TRACE
    if (round < 8192) {
      lab_trace[round][0] = __longlong_as_double((long long)w.minkey[round % 3]);
      if (round > 0) lab_trace[round - 1][1] = (double)w.count[(round + 2) % 3];
    }
#endif
  }
  // the key is fetched alongside the bitmap word, not after it: a round waits for this kernel
  const uint32_t word = t < w.ntiles ? w.bitmap[t >> 5] : 0u;
  unsigned long long kb = t < w.ntiles ? w.key[t] : kInfBits;
  const bool woken = (word >> (t & 31u)) & 1u;
  bool take = false;
  if (woken) {
    const double kv = __longlong_as_double((long long)kb);
    take = !(kv > theta) && !(kv > w.limit);
  } else {
    kb = kInfBits;
  }
  // the 32 lanes that share a bitmap word clear their taken bits together
  const unsigned long long tb = __ballot(take);
  if ((lane & 31u) == 0u) {
    const uint32_t cleared = (uint32_t)(tb >> (lane & 32u));
    if (cleared) w.bitmap[t >> 5] = word & ~cleared;
  }
  if (take) w.key[t] = kInfBits;
  // keys left waiting
  unsigned long long wait = (woken && !take) ? kb : kInfBits;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(wait, off);
    wait = o < wait ? o : wait;
  }
  if (lane == 0) {
    s_take[wv] = (uint32_t)__popcll(tb);
    s_wait[wv] = wait;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    unsigned long long wmin = kInfBits;
    for (int k = 0; k < kListThreads / 64; ++k) {
      const uint32_t c = s_take[k];
      s_take[k] = total;  // the wave's offset inside the workgroup's reservation
      total += c;
      wmin = s_wait[k] < wmin ? s_wait[k] : wmin;
    }
    s_base = total ? atomicAdd(&w.count[round % 3], total) : 0u;
    if (wmin != kInfBits) atomicMin(&w.minkey[(round + 1) % 3], wmin);
  }
  __syncthreads();
  if (take) w.list[s_base + s_take[wv] + (uint32_t)__popcll(tb & ((1ull << lane) - 1ull))] = t;
}

// One workgroup relaxes one 64x64 scheduling tile: its four waves own the four 32x32 blocks, the
// patches of all four trade rim values through one LDS array, and the sweeps repeat (two block
// barriers each) until no cell of the tile moves.
constexpr int kSPps = 2 * kPps;              // patches per scheduling-tile side
constexpr int kSRing = kSPps + 2;
#ifndef PP_LDS_STRIDE
#define PP_LDS_STRIDE 24
#endif
constexpr int kSRow = PP_LDS_STRIDE;         // doubles from one patch row to the next in a rim slot
constexpr int kSSlot = kSRing * kSRow;       // doubles per rim slot
constexpr int kSTile = 2 * kTile;
#ifndef PP_SWEEP_CAP
#define PP_SWEEP_CAP 12
#endif
constexpr uint32_t kSweepCap = PP_SWEEP_CAP;  // sweeps per visit (the mean visit takes 10.6)
#ifndef PP_CAP_MIN_TILES
#define PP_CAP_MIN_TILES 64
#endif
#ifndef PP_TILE_GRID
#define PP_TILE_GRID 4                       // workgroups per CU in a round's launch
#endif


// Which neighbours the cells on one side of a lane's patch can still reach; called for patches
// that moved (every cell of such a patch counts as moved).  The lane's patch
// lies on the tile's edge SIDE (0 x-lo, 1 x-hi, 2 y-lo, 3 y-hi); its four cells there have the six
// ring cells r = 0..5 across the edge (tile coordinate o0 + r along the edge, -1 .. 64) as
// neighbours, cell c next to r = c, c + 1, c + 2.  A cell v can lower ring cell r only if the
// owner of r may relax it and fl(v + step) is below r's value -- as read at the start of the
// visit, and the owner can only have lowered it since, so a "no" is final.  reach[] holds, per ring
// cell, that value, or -1 where the owner may not relax the cell.  The lowest such v per neighbour
// tile goes to wake_key[] (0..3 the sides, 4..7 the corners (lo,lo) (lo,hi) (hi,lo) (hi,hi)).
template <int SIDE>
__device__ __forceinline__ void edge_reach(const double (&e)[6][6], const double (*reach)[68],
                                           int o0, double diag, unsigned long long* wake_key) {
  const double INF = __longlong_as_double((long long)kInfBits);
  double ring[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) ring[r] = reach[SIDE][o0 + r + 1];
  double k_lo = INF, k_mid = INF, k_hi = INF;  // towards ring cell 0, 1..4, 5
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double v = SIDE == 0 ? e[1][1 + c] : SIDE == 1 ? e[4][1 + c] : SIDE == 2 ? e[1 + c][1] : e[1 + c][4];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int r = c + d;
      if (v + (d == 1 ? 1.0 : diag) < ring[r]) {
        if (r == 0) k_lo = pp_min_f64(k_lo, v);
        else if (r == 5) k_hi = pp_min_f64(k_hi, v);
        else k_mid = pp_min_f64(k_mid, v);
      }
    }
  }
  // ring cells 0 and 5 belong to a corner tile when they lie beyond the tile's own extent
  const int corner_lo = SIDE == 0 ? 4 : SIDE == 1 ? 6 : SIDE == 2 ? 4 : 5;
  const int corner_hi = SIDE == 0 ? 5 : SIDE == 1 ? 7 : SIDE == 2 ? 6 : 7;
  if (k_mid < INF) atomicMin(&wake_key[SIDE], (unsigned long long)__double_as_longlong(k_mid));
  if (k_lo < INF) atomicMin(&wake_key[o0 < 0 ? corner_lo : SIDE], (unsigned long long)__double_as_longlong(k_lo));
  if (k_hi < INF) atomicMin(&wake_key[o0 + 5 > kSTile - 1 ? corner_hi : SIDE], (unsigned long long)__double_as_longlong(k_hi));
}

__global__ void __launch_bounds__(256, PP_TILE_WAVES)
k_tile_round(TileGrid g, TileWork w, uint32_t round, double diag) {
  __shared__ double lds[12 * kSSlot];
  __shared__ uint32_t edge_bits;                 // whether the slab's first / last owned column moved
  __shared__ unsigned long long wake_key[8];     // per neighbour: lowest moved value that can still lower one of its cells
  __shared__ double reach[4][68];                // per side: the ring cells' values where their owner may relax them, else -1 (index: offset along the edge + 1)
  __shared__ unsigned long long self_key;        // set when the sweeps were cut short
  __shared__ uint32_t act[3];                    // waves with work in sweep s at act[s % 3]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const int qx = (int)(wv >> 1), qy = (int)(wv & 1u);
  const uint32_t n = w.count[round % 3];
  // sweeps per visit: a round with many tiles is cut at its slowest ones, a round with a handful of
  // tiles (a front threading a corridor) gains nothing from that and relaxes them to the end
  const uint32_t cap = n >= PP_CAP_MIN_TILES ? kSweepCap : 0xFFFFFFFFu;
  const int px = (int)(lane >> 3), py = (int)(lane & 7u);
  const double INF = __longlong_as_double((long long)kInfBits);
  const int own_slot = (qx * kPps + px + 1) * kSRow + (qy * kPps + py + 1);
  unsigned long long blk_min = kInfBits;  // lowest key this thread handed out (threads 0..7)
  uint32_t visits = 0, sweeps_total = 0;

#ifdef PP_LAB_PHASES
  unsigned long long ph[4] = {0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define PP_PHASE(k) do { t1 = wall_clock64(); ph[k] += t1 - t0; t0 = t1; } while (0)
#else
#define PP_PHASE(k) do { } while (0)
#endif
  for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
    const uint32_t t = w.list[it];
    const uint32_t tx = t / g.nsy, ty = t - tx * g.nsy;
    const int64_t X0 = (int64_t)tx * kSTile, Y0 = (int64_t)ty * kSTile;
    const uint32_t bx = 2 * tx + (uint32_t)qx, by = 2 * ty + (uint32_t)qy;
    const bool have = bx < g.ntx && by < g.nty;  // the grid may end inside the scheduling tile
    const uint32_t blk_id = have ? bx * g.nty + by : 0u;
    double* blk = g.blk + (uint64_t)blk_id * kBlk;
    if (tid < 8) wake_key[tid] = kInfBits;
    if (tid == 0) {
      edge_bits = 0;
      self_key = kInfBits;
      act[0] = 0;
      act[1] = 0;
      act[2] = 0;
    }

    // e[1..4][1..4] = the lane's patch, the rest of e = the cells around it.  floor_[][] is the
    // value a cell may not go below: 0 for a free cell, +inf for a blocked one, its own value for
    // a cell that is only a source.
    double e[6][6], floor_[4][4];
    const uint32_t m = have ? g.mask[(uint64_t)blk_id * 64 + lane] : 0xFFFFu;
    // a block no visit has written yet is +inf throughout (its stamp cannot be this round's: a tile
    // is listed once per round)
    const bool fresh = have && g.stamp[blk_id] == 0u;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      double2 p = make_double2(INF, INF);
      if (have && !fresh) p = *(const double2*)(blk + blk_off(2 * k2, lane));
      e[(2 * k2) / 4 + 1][(2 * k2) % 4 + 1] = p.x;
      e[(2 * k2 + 1) / 4 + 1][(2 * k2 + 1) % 4 + 1] = p.y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = i * 4 + j;
        const double v = e[i + 1][j + 1];
        floor_[i][j] = ((m >> k) & 1u) ? INF : (((m >> (16 + k)) & 1u) ? v : 0.0);
        if (on_rim(i, j)) lds[rim_slot(i, j) * kSSlot + own_slot] = v;
      }
    }
    // the ring of cells just outside the tile: 64 on each side, one per thread, and the four corner
    // cells, which threads 0..3 fetch alongside their side cell (both loads in flight together: a
    // second pass over the ring for four cells would cost every visit a memory round trip)
    {
      const int r = (int)tid;
      const bool xside = r < 2 * kSTile;
      const int rr = xside ? r : r - 2 * kSTile;
      const int side = rr >= kSTile, o = rr & (kSTile - 1);  // o = 0 .. 63
      bool rel, crel = false;
      double cv = INF;
      const int cside = (int)(tid >> 1), co = (tid & 1u) ? kSTile : -1;
      if (tid < 4) cv = tile_cell_value(g, cside ? X0 + kSTile : X0 - 1, Y0 + co, crel, round);
      const double v = xside ? tile_cell_value(g, side ? X0 + kSTile : X0 - 1, Y0 + o, rel, round)
                             : tile_yrim_value(g, X0 + o, side ? Y0 + kSTile : Y0 - 1, rel, round);
      if (xside) {
        const int q = (o + 4) >> 2, j = (o + 4) & 3;
        if (!side)
          lds[(4 + j) * kSSlot + 0 * kSRow + q] = v;                // slot (3, j)
        else
          lds[j * kSSlot + (kSRing - 1) * kSRow + q] = v;           // slot (0, j)
        reach[side][o + 1] = rel ? v : -1.0;
      } else {
        const int prow = (o >> 2) + 1, i = o & 3;
        const int s3 = i == 0 ? 3 : (i == 3 ? 7 : 9 + i), s0 = i == 0 ? 0 : (i == 3 ? 4 : 7 + i);
        if (!side)
          lds[s3 * kSSlot + prow * kSRow + 0] = v;                  // slot (i, 3)
        else
          lds[s0 * kSSlot + prow * kSRow + (kSRing - 1)] = v;       // slot (i, 0)
        reach[2 + side][o + 1] = rel ? v : -1.0;
      }
      if (tid < 4) {
        const int q = (co + 4) >> 2, j = (co + 4) & 3;
        if (!cside)
          lds[(4 + j) * kSSlot + 0 * kSRow + q] = cv;
        else
          lds[j * kSSlot + (kSRing - 1) * kSRow + q] = cv;
        const double rv = crel ? cv : -1.0;
        reach[cside][co + 1] = rv;
        // the corner cells are also the ends of the y sides' rows
        reach[co < 0 ? 2 : 3][cside ? kSTile + 1 : 0] = rv;
      }
    }
    __syncthreads();
    PP_PHASE(0);

    // A wave sits a sweep out when nothing it reads has changed since its last one: none of its
    // own lanes moved and no neighbouring wave moved a patch on the shared edge.
    const unsigned long long facing_x = qx ? 0x00000000000000FFull : 0xFF00000000000000ull;
    const unsigned long long facing_y = qy ? 0x0101010101010101ull : 0x8080808080808080ull;
    // One barrier per sweep.  A wave may read a rim its neighbour has already rewritten in the same
    // sweep: any mix of old and new values is a state the relaxation may pass through (an 8-byte
    // LDS access is not torn), and the loop only ends after a sweep in which nobody wrote.
    uint32_t sweeps = 0, todo = 0xFu;
    bool moved_any = false;  // this lane's patch changed in some sweep of the visit
    for (;;) {
      const bool active = (todo >> wv) & 1u;
      const uint32_t nxt = (sweeps + 1u) % 3u;
      bool moved = false;  // values only ever go down
      if (tid == 0) act[(sweeps + 2u) % 3u] = 0;  // last read before the previous barrier, filled after the next
      if (active) {
#define PP_FLOOR(v, i, j) pp_max_f64(v, floor_[(i) - 1][(j) - 1])
        // the cells around the patch, read for the pass that uses them (PASS 0: the forward pass
        // reads row 0, column 0 and rows 0..3 of column 5; PASS 1: the backward pass the rest and
        // again what both need, which may have moved on meanwhile -- any mix of old and new is fine).
        // Eight reads more per sweep than fetching the whole ring once, but twelve values fewer
        // held across the forward pass (16384^2: 19.1 -> 18.7 ms)
#ifndef PP_HALO_WHOLE
#define PP_HALO_WANTED(ii, jj, PASS) ((PASS) == 0 ? ((ii) == 0 || ((jj) == 0 && (ii) <= 4) || ((jj) == 5 && (ii) <= 3)) \
                                                  : ((ii) == 5 || ((jj) == 5 && (ii) >= 1) || ((jj) == 0 && (ii) >= 2)))
#else
#define PP_HALO_WANTED(ii, jj, PASS) ((PASS) == 0)
#endif
#define PP_HALO_READ(PASS)                                                                                   \
  _Pragma("unroll") for (int ii = 0; ii < 6; ++ii) {                                                         \
    _Pragma("unroll") for (int jj = 0; jj < 6; ++jj) {                                                       \
      if (ii >= 1 && ii <= 4 && jj >= 1 && jj <= 4) continue;                                                \
      if (!PP_HALO_WANTED(ii, jj, PASS)) continue;                                                           \
      const int dx = ii == 0 ? -1 : (ii == 5 ? 1 : 0), dy = jj == 0 ? -1 : (jj == 5 ? 1 : 0);                \
      const int si = ii == 0 ? 3 : (ii == 5 ? 0 : ii - 1), sj = jj == 0 ? 3 : (jj == 5 ? 0 : jj - 1);        \
      e[ii][jj] = lds[rim_slot(si, sj) * kSSlot + own_slot + dx * kSRow + dy];                               \
    }                                                                                                        \
  }
        PP_HALO_READ(0)
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
#pragma unroll
          for (int j = 1; j <= 4; ++j) {
            double v = e[i][j];
            v = pp_min_f64(v, e[i - 1][j - 1] + diag);
            v = pp_min_f64(v, e[i - 1][j] + 1.0);
            v = pp_min_f64(v, e[i - 1][j + 1] + diag);
            v = pp_min_f64(v, e[i][j - 1] + 1.0);
            v = PP_FLOOR(v, i, j);
            moved |= v < e[i][j];
            e[i][j] = v;
          }
        }
#ifndef PP_HALO_WHOLE
        PP_HALO_READ(1)
#endif
#pragma unroll
        for (int i = 4; i >= 1; --i) {
#pragma unroll
          for (int j = 4; j >= 1; --j) {
            double v = e[i][j];
            v = pp_min_f64(v, e[i + 1][j + 1] + diag);
            v = pp_min_f64(v, e[i + 1][j] + 1.0);
            v = pp_min_f64(v, e[i + 1][j - 1] + diag);
            v = pp_min_f64(v, e[i][j + 1] + 1.0);
            v = PP_FLOOR(v, i, j);
            moved |= v < e[i][j];
            e[i][j] = v;
          }
        }
        moved_any |= moved;
        if (moved) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (on_rim(i, j)) lds[rim_slot(i, j) * kSSlot + own_slot] = e[i + 1][j + 1];
          }
        }
        const unsigned long long bal = __ballot(moved);
        if (bal != 0ull && lane == 0) {
          uint32_t hit = 1u << wv;
          if (bal & facing_x) hit |= 1u << (wv ^ 2u);
          if (bal & facing_y) hit |= 1u << (wv ^ 1u);
          if (bal & facing_x & facing_y) hit |= 1u << (wv ^ 3u);
          atomicOr(&act[nxt], hit);
        }
      }
      ++sweeps;
      __syncthreads();  // rims and the activity word are published
      todo = act[nxt];
      if (todo == 0) break;
      if (sweeps >= cap) {
        // A front that winds through the tile needs many sweeps, and the round lasts as long as
        // its slowest tile.  Such a tile stops here and wakes itself, with the lowest value of the
        // waves that still had work as its key, so the round ends and the rest of the front moves
        // on (4096^2: 4.3 -> 3.7 ms, 16384^2: 27.4 -> 26.7 ms).  (Keying the tile by the patches that
        // moved in the last sweep alone, a tighter bound, changes neither the rounds nor the time.)
        double mv = INF;
        if ((todo >> wv) & 1u) {
#pragma unroll
          for (int i = 1; i <= 4; ++i)
#pragma unroll
            for (int j = 1; j <= 4; ++j) mv = pp_min_f64(mv, e[i][j]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mv = pp_min_f64(mv, __shfl_xor(mv, off));
        if (lane == 0 && mv < INF) atomicMin(&self_key, (unsigned long long)__double_as_longlong(mv));
        break;
      }
    }

    PP_PHASE(1);
    // Write back the patches that moved, whole, and let cells on the tile's edge wake the
    // neighbours that read them.  Which cells of a moved patch changed is not looked up (that takes
    // reading the old values back, 2 us of latency at the end of every visit, for 4 % fewer bytes
    // written): all sixteen count as moved.  For the wake test below an edge cell that did not
    // change can only repeat a wake its neighbour has had already -- if the neighbour has relaxed
    // since, its cells are low enough for the test to say no.
    // A block written for the first time is written whole, the patches still at +inf included,
    // and stamped; it stays unwritten when none of its cells moved.
    const bool moved_here = have && moved_any;
    const bool first_write = fresh && __ballot(moved_here) != 0ull;
    const bool wrote = moved_here || first_write;
    if (wrote) {
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2)
        *(double2*)(blk + blk_off(2 * k2, lane)) =
            make_double2(e[(2 * k2) / 4 + 1][(2 * k2) % 4 + 1], e[(2 * k2 + 1) / 4 + 1][(2 * k2 + 1) % 4 + 1]);
    }
    if (first_write && lane == 0u) g.stamp[blk_id] = round + 2u;
    if (moved_here) {
      // a slab's first / last owned column (reported to the caller) runs through this patch
      const uint64_t c0 = (uint64_t)(X0 + qx * kTile + 4 * px);
      if (w.own_first >= c0 && w.own_first < c0 + 4) atomicOr(&edge_bits, 256u);
      if (w.own_last >= c0 && w.own_last < c0 + 4) atomicOr(&edge_bits, 512u);
    }
    // the copies of the rows on the tile's y edges (the only ones other tiles read; the two rows
    // where the tile's blocks meet are left as they are): the lane's four cells in one piece
    if (qy == 0 && py == 0 && wrote) {
      double* yr = g.yrim + ((uint64_t)blk_id * 2) * 32 + 4 * px;
      *(double2*)yr = make_double2(e[1][1], e[2][1]);
      *(double2*)(yr + 2) = make_double2(e[3][1], e[4][1]);
    }
    if (qy == 1 && py == kPps - 1 && wrote) {
      double* yr = g.yrim + ((uint64_t)blk_id * 2 + 1) * 32 + 4 * px;
      *(double2*)yr = make_double2(e[1][4], e[2][4]);
      *(double2*)(yr + 2) = make_double2(e[3][4], e[4][4]);
    }
    // moved cells on the tile's edge: which neighbours can they still reach
    if (__ballot(moved_here) != 0ull) {
      if (qx == 0) {
        if (px == 0 && moved_here) edge_reach<0>(e, reach, qy * kTile + 4 * py - 1, diag, wake_key);
      } else {
        if (px == kPps - 1 && moved_here) edge_reach<1>(e, reach, qy * kTile + 4 * py - 1, diag, wake_key);
      }
      if (qy == 0) {
        if (py == 0 && moved_here) edge_reach<2>(e, reach, qx * kTile + 4 * px - 1, diag, wake_key);
      } else {
        if (py == kPps - 1 && moved_here) edge_reach<3>(e, reach, qx * kTile + 4 * px - 1, diag, wake_key);
      }
    }
    __syncthreads();
    // threads 0..7 look after one neighbour each
    if (tid < 8) {
      const int ddx = tid < 3 ? -1 : (tid < 5 ? 0 : 1);
      const int ddy = (tid == 0 || tid == 3 || tid == 5) ? -1 : ((tid == 2 || tid == 4 || tid == 7) ? 1 : 0);
      const int nb = tid == 0 ? 4 : tid == 1 ? 0 : tid == 2 ? 5 : tid == 3 ? 2 : tid == 4 ? 3 : tid == 5 ? 6 : tid == 6 ? 1 : 7;
      const unsigned long long wk = wake_key[nb];
      if (wk != kInfBits) {
        tile_wake(g, w, (int64_t)tx + ddx, (int64_t)ty + ddy, wk);
        blk_min = min(blk_min, wk);
      }
    }
    if (tid == 0 && self_key != kInfBits) {
      tile_wake(g, w, (int64_t)tx, (int64_t)ty, self_key);
      blk_min = min(blk_min, self_key);
    }
    if (tid == 0) {
      const uint32_t eb = edge_bits;
      if (eb & 256u) w.status[2] = 1;
      if (eb & 512u) w.status[3] = 1;
      ++visits;
      sweeps_total += sweeps;
    }
    __syncthreads();  // the shared words and the LDS rims are reused by the next tile
    PP_PHASE(2);
  }
#ifdef PP_LAB_PHASES
  if (tid == 0 && visits) {
    unsigned long long* prof = (unsigned long long*)((char*)w.count + 208);
    for (int k = 0; k < 3; ++k) atomicAdd(&prof[k], ph[k]);
  }
#endif
  // one minimum and one pair of counters per workgroup: every workgroup of a round adds to the
  // same two words
  if (wv == 0) {
    unsigned long long m = blk_min;  // held by threads 0..7
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off);
      m = o < m ? o : m;
    }
    if (tid == 0 && m != kInfBits) atomicMin(&w.minkey[(round + 1) % 3], m);
    if (tid == 0 && visits)  // status[0] += visits, status[1] += sweeps_total (adjacent words, 8-byte aligned)
      atomicAdd((unsigned long long*)&w.status[0], ((unsigned long long)sweeps_total << 32) | visits);
  }
}

__global__ void k_fill_inf(double* p, uint64_t n) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = __longlong_as_double((long long)kInfBits);
}

// Goal cells are stored before the rounds, so their blocks have to hold real values by then: one
// wave per goal writes +inf over a block that is still unwritten.  The stamps are left alone (two
// goals of neighbouring planes may share a block; both waves then write the same values) and are
// set by the seeding kernel that follows.  goals == nullptr: the one goal (sx, sy).
__global__ void __launch_bounds__(64)
k_seed_fill(TileGrid g, uint32_t n, uint64_t plane_cols, const int64_t* __restrict__ goals, int64_t sx, int64_t sy) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  if (p >= n) return;
  const int64_t x = goals ? (int64_t)(p * plane_cols) + goals[2 * p] : sx, y = goals ? goals[2 * p + 1] : sy;
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  if (g.stamp[t] != 0u) return;
  const double INF = __longlong_as_double((long long)kInfBits);
  double* b = g.blk + (uint64_t)t * kBlk;
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) *(double2*)(b + blk_off(2 * k2, lane)) = make_double2(INF, INF);
  g.yrim[(uint64_t)t * 64 + lane] = INF;  // both row copies
}

// The goal is inserted without check_node (astar.rs:155): cost 0 even on an obstacle or border.
__global__ void k_seed_goal(TileGrid g, TileWork w, int64_t x, int64_t y) {
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  g.blk[(uint64_t)t * kBlk + blk_off((lx & 3u) * 4u + (ly & 3u), (lx >> 2) * 8u + (ly >> 2))] = 0.0;
  tile_yrim_store(g, t, lx, ly, 0.0);
  g.stamp[t] = 1u;
  tile_wake(g, w, x >> 6, y >> 6, 0ull);
  atomicMin(&w.minkey[0], 0ull);
}

__global__ void k_seed_goals(TileGrid g, TileWork w, uint32_t planes, uint64_t plane_cols,
                             const int64_t* __restrict__ goals) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= planes) return;
  const int64_t x = (int64_t)(p * plane_cols) + goals[2 * p], y = goals[2 * p + 1];
  const uint32_t t = (uint32_t)(x >> 5) * g.nty + (uint32_t)(y >> 5);
  const uint32_t lx = (uint32_t)x & 31u, ly = (uint32_t)y & 31u;
  g.blk[(uint64_t)t * kBlk + blk_off((lx & 3u) * 4u + (ly & 3u), (lx >> 2) * 8u + (ly >> 2))] = 0.0;
  tile_yrim_store(g, t, lx, ly, 0.0);
  g.stamp[t] = 1u;
  tile_wake(g, w, x >> 6, y >> 6, 0ull);
  if (p == 0) atomicMin(&w.minkey[0], 0ull);
}

// Halo columns handed in by the neighbour ranks (dense arrays, before packing): where the new value
// is lower than the stored one, store it and wake every tile that holds or borders the cell.
__global__ void k_take_halo(DenseView d, TileGrid g, TileWork w, const double* __restrict__ lo_new,
                            const double* __restrict__ hi_new) {
  const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= d.yw) return;
  for (int side = 0; side < 2; ++side) {
    const double* src = side ? hi_new : lo_new;
    if (!src) continue;
    const uint64_t col = side ? d.own_hi : d.own_lo - 1;
    const uint64_t cell = col * d.yw + y;
    const double v = src[y];
    if (v < d.cost[cell]) {
      d.cost[cell] = v;
      const unsigned long long kb = (unsigned long long)__double_as_longlong(v);
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
          const int64_t nx = (int64_t)col + dx, ny = (int64_t)y + dy;
          if (nx >= 0 && ny >= 0) tile_wake(g, w, nx >> 6, ny >> 6, kb);
        }
      atomicMin(&w.minkey[0], kb);
    }
  }
}

// The same for a slab whose working copy stays in the blocked layout between calls: the halo
// columns are stored columns col_lo / col_hi of the blocked array, and `round` is the next round to
// run (its list kernel reads minkey[round % 3]).
__global__ void k_slab_take_halo(TileGrid g, TileWork w, uint64_t yw, uint64_t col_lo, uint64_t col_hi,
                                 const double* __restrict__ lo_new, const double* __restrict__ hi_new,
                                 uint32_t round) {
  const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= yw) return;
  for (int side = 0; side < 2; ++side) {
    const double* src = side ? hi_new : lo_new;
    if (!src) continue;
    const uint64_t col = side ? col_hi : col_lo;
    const uint32_t t = (uint32_t)(col >> 5) * g.nty + (uint32_t)(y >> 5);
    const uint32_t lx = (uint32_t)col & 31u, ly = (uint32_t)y & 31u;
    double* cell = g.blk + (uint64_t)t * kBlk + blk_off((lx & 3u) * 4u + (ly & 3u), (lx >> 2) * 8u + (ly >> 2));
    const double v = src[y];
    if (v < *cell) {
      *cell = v;
      tile_yrim_store(g, t, lx, ly, v);
      const unsigned long long kb = (unsigned long long)__double_as_longlong(v);
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
          const int64_t nx = (int64_t)col + dx, ny = (int64_t)y + dy;
          if (nx >= 0 && ny >= 0) tile_wake(g, w, nx >> 6, ny >> 6, kb);
        }
      atomicMin(&w.minkey[round % 3], kb);
    }
  }
}

// First and last owned column of a slab, blocked -> the neighbours' halo buffers.  edge_lo / edge_hi
// hold what was handed out before; *moved_min receives the lowest value that is new.
__global__ void k_slab_edges(TileGrid g, uint64_t yw, uint64_t col_first, uint64_t col_last,
                             double* __restrict__ edge_lo, double* __restrict__ edge_hi,
                             unsigned long long* moved_min) {
  const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= yw) return;
  for (int side = 0; side < 2; ++side) {
    double* dst = side ? edge_hi : edge_lo;
    if (!dst) continue;
    const double v = tile_cell_value(g, (int64_t)(side ? col_last : col_first), (int64_t)y);
    if (v < dst[y]) {
      dst[y] = v;
      atomicMin(moved_min, (unsigned long long)__double_as_longlong(v));
    }
  }
}

// Cells (x[i], y[i]) in global coordinates from the blocked array; stored column = x - gx_off.
__global__ void k_slab_gather(TileGrid g, uint64_t n, int64_t gx_off, uint64_t ncols, uint64_t yw,
                              const uint64_t* __restrict__ x, const uint64_t* __restrict__ y,
                              double* __restrict__ out, uint32_t* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t col = (int64_t)x[i] - gx_off;
  if (col < 0 || col >= (int64_t)ncols || y[i] >= yw) {
    out[i] = __longlong_as_double((long long)kInfBits);
    *bad = 1u;
    return;
  }
  out[i] = tile_cell_value(g, col, (int64_t)y[i]);
}

struct TileRun {
  TileGrid g{};
  TileWork w{};
  uint32_t nblocks = 0;  // 32x32 blocks (pack / unpack work items)
  uint32_t round = 0;    // rounds run so far (the work slots rotate with it, so a later run continues the count)
};

// A relaxation stopped at a cost limit, to be completed later from the same working buffer.
struct WaveResume {
  TileRun run;
  DenseView d;
  unsigned sg;
  bool unpack;  // the caller reads the dense arrays (false: it reads the blocked array itself)
};
void wave_resume_drop(pp_ctx* ctx) {
  delete (WaveResume*)ctx->wave_resume;
  ctx->wave_resume = nullptr;
}

// Blocked costs, masks and work lists live in a context-owned buffer (grown on demand) so
// repeated calls do not allocate.
int32_t tile_prepare(pp_ctx* ctx, const DenseView& d, TileRun* run, void** own_buf = nullptr,
                     size_t* own_bytes = nullptr) {
  void*& wbuf = own_buf ? *own_buf : ctx->wave_buf;
  size_t& wbytes = own_bytes ? *own_bytes : ctx->wave_bytes;
  if (!own_buf) wave_resume_drop(ctx);  // the context's buffer is about to be overwritten
  const uint64_t ntx = (d.ncols + kTile - 1) / kTile, nty = (d.yw + kTile - 1) / kTile;
  const uint64_t nblocks = ntx * nty;
  const uint64_t nsx = (ntx + 1) / 2, nsy = (nty + 1) / 2, ntiles = nsx * nsy;  // scheduling tiles
  if (nblocks == 0 || nblocks >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_blk = al(nblocks * kBlk * 8), b_mask = al(nblocks * 64 * 4), b_list = al(ntiles * 4);
  const size_t b_yrim = al(nblocks * 2 * 32 * 8), b_stamp = al(nblocks * 4);
  const size_t b_bits = al(((ntiles + 63) / 64) * 8), b_key = al(ntiles * 8), b_misc = 256;
  const size_t need = b_blk + b_mask + b_yrim + b_stamp + b_list + b_bits + b_key + b_misc;
  if (need > wbytes) {
    if (wbuf) PP_HIP(ctx, hipFree(wbuf));
    wbuf = nullptr;
    wbytes = 0;
    PP_HIP(ctx, hipMalloc(&wbuf, need));
    wbytes = need;
  }
  char* b = (char*)wbuf;
  run->g.blk = (double*)b;
  b += b_blk;
  run->g.mask = (uint32_t*)b;
  b += b_mask;
  run->g.yrim = (double*)b;
  b += b_yrim;
  run->g.stamp = (uint32_t*)b;
  b += b_stamp;
  run->g.ntx = (uint32_t)ntx;
  run->g.nty = (uint32_t)nty;
  run->g.nsx = (uint32_t)nsx;
  run->g.nsy = (uint32_t)nsy;
  run->nblocks = (uint32_t)nblocks;
  run->w.list = (uint32_t*)b;
  b += b_list;
  run->w.bitmap = (uint32_t*)b;
  b += b_bits;
  run->w.key = (unsigned long long*)b;
  b += b_key;
  run->w.count = (uint32_t*)b;
  run->w.status = (uint32_t*)(b + 64);
  run->w.minkey = (unsigned long long*)(b + 128);
  run->w.ntiles = (uint32_t)ntiles;
  run->w.band = ctx->wave_band;
  run->w.limit = __builtin_inf();
  run->w.own_first = d.own_lo;
  run->w.own_last = d.own_hi - 1;
  PP_HIP(ctx, hipMemsetAsync(run->w.bitmap, 0, b_bits, ctx->stream));
  PP_HIP(ctx, hipMemsetAsync(run->w.count, 0, 256, ctx->stream));
  // keys and minkey[3] start at +inf
  hipLaunchKernelGGL(k_fill_inf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, (double*)run->w.key,
                     (uint64_t)ntiles);
  hipLaunchKernelGGL(k_fill_inf, dim3(1), dim3(64), 0, ctx->stream, (double*)run->w.minkey, (uint64_t)3);
  return PP_OK;
}

// Rounds until no tile is woken.  The host reads the list lengths once per batch of rounds, and the
// device is kept busy meanwhile: two rounds of the next batch are queued before the host waits for
// the read-back (a wait plus the relaunch costs about as long as one or two rounds take), the rest
// of the batch follows if the last batch ended with tiles still listed.  A round over an empty
// list costs two empty launches.
int32_t tile_run(pp_ctx* ctx, TileRun& run, uint64_t* rounds_out, uint32_t status_out[4]) {
  const double diag = std::hypot(1.0, 1.0);  // u_cost((+-1, +-1)), astar.rs:213-215
  const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)ctx->num_cus * PP_TILE_GRID, run.w.ntiles);
  const bool wide_list = run.w.ntiles >= 16384u;
  const unsigned lgrid = (run.w.ntiles + (wide_list ? 1023u : 255u)) / (wide_list ? 1024u : 256u);
  if (!ctx->wave_counts) PP_HIP(ctx, hipHostMalloc((void**)&ctx->wave_counts, 64, hipHostMallocDefault));
  if (!ctx->wave_event) PP_HIP(ctx, hipEventCreateWithFlags(&ctx->wave_event, hipEventDisableTiming));
  uint32_t round = run.round;
  auto enqueue = [&](uint32_t k) {
    for (uint32_t j = 0; j < k; ++j, ++round) {
      if (wide_list)
        hipLaunchKernelGGL(k_tile_list<1024>, dim3(lgrid), dim3(1024), 0, ctx->stream, run.g, run.w, round);
      else
        hipLaunchKernelGGL(k_tile_list<256>, dim3(lgrid), dim3(256), 0, ctx->stream, run.g, run.w, round);
      hipLaunchKernelGGL(k_tile_round, dim3(grid), dim3(256), 0, ctx->stream, run.g, run.w, round, diag);
    }
  };
#ifndef PP_RUN_AHEAD
#define PP_RUN_AHEAD 2
#endif
#ifndef PP_RUN_BATCH_MAX
#define PP_RUN_BATCH_MAX 64
#endif
  constexpr uint32_t kAhead = PP_RUN_AHEAD;
  uint32_t batch = 2;  // a slab session often has a round or two of work per call
  enqueue(batch);
  for (;;) {
    const uint32_t last = round - 1;  // the round whose list length decides
    PP_HIP(ctx, hipGetLastError());
    PP_HIP(ctx, hipMemcpyAsync(ctx->wave_counts, run.w.count, 12, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIP(ctx, hipEventRecord(ctx->wave_event, ctx->stream));
    enqueue(kAhead);
    PP_HIP(ctx, hipEventSynchronize(ctx->wave_event));
    if (ctx->wave_counts[last % 3] == 0) break;  // that round found nothing woken, nor do the two after it
    if (round > (1u << 30)) {
      (void)hipStreamSynchronize(ctx->stream);
      return PP_ERR_UNSUPPORTED;
    }
    batch = std::min<uint32_t>(batch * 2, PP_RUN_BATCH_MAX);
    enqueue(batch > kAhead ? batch - kAhead : 0);
  }
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipMemcpyAsync(status_out, run.w.status, 16, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
#ifdef PP_LAB_TRACE
  {
    static double tr[8192][2];
    PP_HIP(ctx, hipMemcpyFromSymbol(tr, HIP_SYMBOL(lab_trace), sizeof(tr)));
    for (uint32_t r = 0; r < round && r < 8192; ++r)
      fprintf(stderr, "[trace] round %u minkey %.1f listed %.0f\n", r, tr[r][0], tr[r][1]);
  }
#endif
#ifdef PP_LAB_PHASES
  {
    unsigned long long prof[3];
    PP_HIP(ctx, hipMemcpy(prof, (char*)run.w.count + 208, 24, hipMemcpyDeviceToHost));
    fprintf(stderr, "[phases] %u visits: load %.2f sweeps %.2f store+wake %.2f us per visit\n", status_out[0],
            prof[0] / 100.0 / status_out[0], prof[1] / 100.0 / status_out[0], prof[2] / 100.0 / status_out[0]);
  }
#endif
  run.round = round;
  if (rounds_out) *rounds_out = round;
  ctx->wave_stats[0] = round;
  ctx->wave_stats[1] = status_out[0];
  ctx->wave_stats[2] = status_out[1];
  ctx->wave_stats[3] = run.w.ntiles;
  return PP_OK;
}

unsigned tile_stream_grid(const pp_ctx* ctx, uint32_t ntiles) {
  return (unsigned)std::min<uint64_t>((uint64_t)ctx->num_cus * 8, ((uint64_t)ntiles + 3) / 4);
}

}  // namespace

// ---------------------------------------------------------------------------- entry points

extern "C" int32_t pp_wavefront_stats(pp_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return PP_ERR_INVALID;
  std::memcpy(out, ctx->wave_stats, sizeof(ctx->wave_stats));
  return PP_OK;
}

extern "C" int32_t pp_wavefront_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, const uint8_t* d_obs,
                                    uint64_t goal_x, uint64_t goal_y, double* d_cost,
                                    uint64_t* iters) {
  if (!ctx || !d_obs || !d_cost) return PP_ERR_INVALID;
  if (xw == 0 || yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  // the goal's own hmap write (astar.rs:195) indexes hmap[gx - minx][gy - miny]
  if (goal_x >= xw || goal_y >= yw) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  DenseView d{d_cost, d_obs, yw, xw, 0, 0, xw, (int64_t)goal_x, (int64_t)goal_y, 0, 0, 0, nullptr};
  TileRun run;
  int32_t st = tile_prepare(ctx, d, &run);
  if (st != PP_OK) return st;
  const unsigned sg = tile_stream_grid(ctx, run.nblocks);
  hipLaunchKernelGGL(k_tile_pack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks, 1);
  hipLaunchKernelGGL(k_seed_fill, dim3(1), dim3(64), 0, ctx->stream, run.g, 1u, (uint64_t)0, (const int64_t*)nullptr,
                     (int64_t)goal_x, (int64_t)goal_y);
  hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, run.g, run.w, (int64_t)goal_x,
                     (int64_t)goal_y);
  uint32_t status[4];
  st = tile_run(ctx, run, iters, status);
  if (st != PP_OK) return st;
  hipLaunchKernelGGL(k_tile_unpack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

// n grids over one obstacle raster, one goal each, relaxed in the same rounds (the planner's
// heuristic maps for the distinct goal cells of a batch of queries).  d_cost holds the n dense
// grids cost_stride doubles apart; goal_x / goal_y are host arrays.
//
// With a finite `limit` (and all grids in one pass) the relaxation stops at cells of that value:
// every cell whose final value is at most the limit has it, the others hold an upper bound or
// +inf.  The working state stays in the context and pp_wavefront_multi_finish completes the grids;
// *bounded tells the caller whether that is the case.
static uint32_t multi_per_pass(const pp_ctx* ctx, uint64_t xw, uint64_t yw, uint32_t n) {
  const uint64_t plane_tiles = ((xw + 1 + kTile - 1) / kTile + 1) * ((yw + kTile - 1) / kTile);
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n, ctx->heuristic_pass_blocks / plane_tiles));
}
bool pp_wavefront_multi_single_pass(const pp_ctx* ctx, uint64_t xw, uint64_t yw, uint32_t n) {
  return multi_per_pass(ctx, xw, yw, n) >= n;
}

int32_t pp_wavefront_multi_finish(pp_ctx* ctx) {
  if (!ctx) return PP_ERR_INVALID;
  WaveResume* r = (WaveResume*)ctx->wave_resume;
  if (!r) return PP_OK;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  r->run.w.limit = __builtin_inf();
  uint32_t status[4];
  int32_t st = tile_run(ctx, r->run, nullptr, status);
  if (st == PP_OK) {
    if (r->unpack)
      hipLaunchKernelGGL(k_tile_unpack, dim3(r->sg), dim3(256), 0, ctx->stream, r->d, r->run.g, r->run.nblocks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = PP_ERR_HIP;
  }
  wave_resume_drop(ctx);
  return st;
}

int32_t pp_wavefront_multi_dev(pp_ctx* ctx, uint64_t xw, uint64_t yw, const uint8_t* d_obs, uint32_t n,
                               const uint64_t* goal_x, const uint64_t* goal_y, double* d_cost,
                               uint64_t cost_stride, double limit, bool* bounded, PpBlockedGrids* keep_blocked) {
  if (bounded) *bounded = false;
  if (!ctx || !d_obs || (!d_cost && !keep_blocked) || !goal_x || !goal_y) return PP_ERR_INVALID;
  if (keep_blocked && !pp_wavefront_multi_single_pass(ctx, xw, yw, n)) return PP_ERR_INVALID;
  if (n == 0) return PP_OK;
  if (xw == 0 || yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  // planes per pass: keep the blocked working copy under ~8 GiB
  const uint32_t per_pass = multi_per_pass(ctx, xw, yw, n);
  std::vector<int64_t> hg(2 * (size_t)per_pass);
  for (uint32_t p0 = 0; p0 < n; p0 += per_pass) {
    const uint32_t np = std::min(per_pass, n - p0);
    for (uint32_t p = 0; p < np; ++p) {
      if (goal_x[p0 + p] >= xw || goal_y[p0 + p] >= yw) return PP_ERR_BOUNDS;
      hg[2 * p] = (int64_t)goal_x[p0 + p];
      hg[2 * p + 1] = (int64_t)goal_y[p0 + p];
    }
    int32_t st = pp_scratch_reserve(ctx, 2 * (size_t)per_pass * 8);
    if (st != PP_OK) return st;
    int64_t* d_goals = (int64_t*)ctx->scratch;
    PP_HIP(ctx, hipMemcpyAsync(d_goals, hg.data(), 2 * (size_t)np * 8, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t ncols = (uint64_t)np * (xw + 1);
    DenseView d{d_cost + (uint64_t)p0 * cost_stride, d_obs, yw, ncols, 0, 0, ncols, -1, -1,
                np, xw + 1, cost_stride, d_goals};
    TileRun run;
    st = tile_prepare(ctx, d, &run);
    if (st != PP_OK) return st;
    const bool bound = limit < __builtin_inf() && per_pass >= n;
    if (bound) run.w.limit = limit;
    const unsigned sg = tile_stream_grid(ctx, run.nblocks);
    hipLaunchKernelGGL(k_tile_pack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks, 1);
    hipLaunchKernelGGL(k_seed_fill, dim3(np), dim3(64), 0, ctx->stream, run.g, np, xw + 1, (const int64_t*)d_goals,
                       (int64_t)0, (int64_t)0);
    hipLaunchKernelGGL(k_seed_goals, dim3((np + 63) / 64), dim3(64), 0, ctx->stream, run.g, run.w, np,
                       xw + 1, (const int64_t*)d_goals);
    uint32_t status[4];
    st = tile_run(ctx, run, nullptr, status);
    if (st != PP_OK) return st;
    if (!keep_blocked)
      hipLaunchKernelGGL(k_tile_unpack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks);
    else
      *keep_blocked = PpBlockedGrids{run.g.blk, run.g.stamp, run.g.nty, xw + 1};
    PP_HIP(ctx, hipGetLastError());
    PP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // d_goals is reused by the next pass
    if (bound) {
      ctx->wave_resume = new WaveResume{run, d, sg, keep_blocked == nullptr};
      ctx->wave_resume_free = [](void* p) { delete (WaveResume*)p; };
      if (bounded) *bounded = true;
    }
  }
  return PP_OK;
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
  const bool goal_owned = goal_x >= x_begin && goal_x < x_end && goal_y < yw;
  DenseView d{d_cost_slab, d_obs_slab, yw, ncols, (int64_t)x_begin - 1, 1, ncols - 1,
              goal_owned ? (int64_t)(goal_x - x_begin + 1) : -1, goal_owned ? (int64_t)goal_y : -1,
              0, 0, 0, nullptr};
  TileRun run;
  int32_t st = tile_prepare(ctx, d, &run);
  if (st != PP_OK) return st;
  if (d_halo_lo_new || d_halo_hi_new)
    hipLaunchKernelGGL(k_take_halo, dim3((unsigned)((yw + 255) / 256)), dim3(256), 0, ctx->stream, d,
                       run.g, run.w, d_halo_lo_new, d_halo_hi_new);
  const unsigned sg = tile_stream_grid(ctx, run.nblocks);
  hipLaunchKernelGGL(k_tile_pack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks, 0);
  if (seed_goal && goal_owned)  // this entry point packs every block with the caller's values
    hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, run.g, run.w, d.goal_x, d.goal_y);
  uint32_t status[4] = {0, 0, 0, 0};
  st = tile_run(ctx, run, iters, status);
  if (st != PP_OK) return st;
  hipLaunchKernelGGL(k_tile_unpack, dim3(sg), dim3(256), 0, ctx->stream, d, run.g, run.nblocks);
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (changed_lo) *changed_lo = (int32_t)status[2];
  if (changed_hi) *changed_hi = (int32_t)status[3];
  return PP_OK;
}

// ------------------------------------------------------------------ slab sessions
// A slab whose blocked working copy lives from pp_slab_create to pp_slab_destroy: the raster is
// packed once, every pp_slab_relax continues from the state the last one left, and the dense costs
// are written only when asked for.
struct pp_slab {
  pp_ctx* ctx = nullptr;                    // null once the context is gone: every call then fails, destroy still works
  pp_slab *prev = nullptr, *next = nullptr;  // the context's list of live sessions
  TileRun run;
  DenseView d{};
  unsigned sg = 0;
  void* buf = nullptr;
  size_t bytes = 0;
  unsigned long long* moved_min = nullptr;  // device word inside buf
};

extern "C" int32_t pp_slab_reset(pp_slab* s, const uint8_t* d_obs_slab);

// A context that is destroyed before its slab sessions takes their device memory with it; the
// handles stay valid for pp_slab_destroy and refuse everything else.
static void slabs_orphan(pp_ctx* ctx) {
  for (pp_slab* s = (pp_slab*)ctx->slabs; s;) {
    pp_slab* nx = s->next;
    if (s->buf) (void)hipFree(s->buf);
    s->buf = nullptr;
    s->bytes = 0;
    s->ctx = nullptr;
    s->prev = s->next = nullptr;
    s = nx;
  }
  ctx->slabs = nullptr;
}

extern "C" int32_t pp_slab_create(pp_ctx* ctx, uint64_t xw, uint64_t yw, uint64_t x_begin, uint64_t x_end,
                                  const uint8_t* d_obs_slab, uint64_t goal_x, uint64_t goal_y, pp_slab** out) {
  if (!ctx || !d_obs_slab || !out) return PP_ERR_INVALID;
  *out = nullptr;
  if (x_begin >= x_end || x_end > xw) return PP_ERR_INVALID;
  if (yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  if (goal_x >= xw || goal_y >= yw) return PP_ERR_BOUNDS;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t ncols = x_end - x_begin + 2;
  const bool goal_owned = goal_x >= x_begin && goal_x < x_end;
  pp_slab* s = new pp_slab();
  s->ctx = ctx;
  s->d = DenseView{nullptr, d_obs_slab, yw, ncols, (int64_t)x_begin - 1, 1, ncols - 1,
                   goal_owned ? (int64_t)(goal_x - x_begin + 1) : -1, goal_owned ? (int64_t)goal_y : -1,
                   0, 0, 0, nullptr};
  int32_t st = pp_slab_reset(s, d_obs_slab);
  if (st != PP_OK) {
    if (s->buf) (void)hipFree(s->buf);
    delete s;
    return st;
  }
  s->next = (pp_slab*)ctx->slabs;
  if (s->next) s->next->prev = s;
  ctx->slabs = s;
  ctx->slabs_orphan = slabs_orphan;
  *out = s;
  return PP_OK;
}

extern "C" int32_t pp_slab_reset(pp_slab* s, const uint8_t* d_obs_slab) {
  if (!s || !s->ctx || !d_obs_slab) return PP_ERR_INVALID;
  pp_ctx* ctx = s->ctx;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  s->d.obs = d_obs_slab;
  s->run = TileRun{};
  int32_t st = tile_prepare(ctx, s->d, &s->run, &s->buf, &s->bytes);
  if (st != PP_OK) return st;
  s->moved_min = (unsigned long long*)((char*)s->run.w.count + 192);
  s->sg = tile_stream_grid(ctx, s->run.nblocks);
  hipLaunchKernelGGL(k_tile_pack, dim3(s->sg), dim3(256), 0, ctx->stream, s->d, s->run.g, s->run.nblocks, 1);
  if (s->d.goal_x >= 0) {
    hipLaunchKernelGGL(k_seed_fill, dim3(1), dim3(64), 0, ctx->stream, s->run.g, 1u, (uint64_t)0,
                       (const int64_t*)nullptr, s->d.goal_x, s->d.goal_y);
    hipLaunchKernelGGL(k_seed_goal, dim3(1), dim3(1), 0, ctx->stream, s->run.g, s->run.w, s->d.goal_x, s->d.goal_y);
  }
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the raster may be released after the call
  return PP_OK;
}

extern "C" void pp_slab_destroy(pp_slab* s) {
  if (!s) return;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->buf) (void)hipFree(s->buf);
    if (s->prev) s->prev->next = s->next; else s->ctx->slabs = s->next;
    if (s->next) s->next->prev = s->prev;
  }
  delete s;
}

extern "C" int32_t pp_slab_relax(pp_slab* s, const double* d_halo_lo_new, const double* d_halo_hi_new,
                                 double limit, double* d_edge_lo, double* d_edge_hi, double* pending,
                                 uint64_t* iters) {
  if (!s || !s->ctx || limit != limit) return PP_ERR_INVALID;
  pp_ctx* ctx = s->ctx;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t yw = s->d.yw;
  const unsigned yg = (unsigned)((yw + 255) / 256);
  if (d_halo_lo_new || d_halo_hi_new)
    hipLaunchKernelGGL(k_slab_take_halo, dim3(yg), dim3(256), 0, ctx->stream, s->run.g, s->run.w, yw,
                       (uint64_t)0, s->d.ncols - 1, d_halo_lo_new, d_halo_hi_new, s->run.round);
  s->run.w.limit = limit;
  const uint64_t round0 = s->run.round;
  uint64_t rounds = 0;
  uint32_t status[4];
  int32_t st = tile_run(ctx, s->run, &rounds, status);
  if (st != PP_OK) return st;
  hipLaunchKernelGGL(k_fill_inf, dim3(1), dim3(64), 0, ctx->stream, (double*)s->moved_min, (uint64_t)1);
  hipLaunchKernelGGL(k_slab_edges, dim3(yg), dim3(256), 0, ctx->stream, s->run.g, yw, (uint64_t)1, s->d.ncols - 2,
                     d_edge_lo, d_edge_hi, s->moved_min);
  PP_HIP(ctx, hipGetLastError());
  // what is left: tiles woken above the limit, and edge values the neighbours have not seen yet
  double left[2];
  PP_HIP(ctx, hipMemcpyAsync(&left[0], s->run.w.minkey + s->run.round % 3, 8, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipMemcpyAsync(&left[1], s->moved_min, 8, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (pending) *pending = std::min(left[0], left[1]);
  if (iters) *iters = rounds - round0;
  return PP_OK;
}

extern "C" int32_t pp_slab_gather(pp_slab* s, uint64_t n, const uint64_t* d_x, const uint64_t* d_y, double* d_out) {
  if (!s || !s->ctx) return PP_ERR_INVALID;
  if (n == 0) return PP_OK;
  if (!d_x || !d_y || !d_out) return PP_ERR_INVALID;
  pp_ctx* ctx = s->ctx;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t* bad = (uint32_t*)((char*)s->run.w.count + 200);  // a spare word of the session's counters
  PP_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_slab_gather, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 30)), dim3(256), 0,
                     ctx->stream, s->run.g, n, s->d.gx_off, s->d.ncols, s->d.yw, d_x, d_y, d_out, bad);
  PP_HIP(ctx, hipGetLastError());
  uint32_t flag = 0;
  PP_HIP(ctx, hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return flag ? PP_ERR_BOUNDS : PP_OK;
}

extern "C" int32_t pp_slab_read(pp_slab* s, double* d_cost_slab) {
  if (!s || !s->ctx || !d_cost_slab) return PP_ERR_INVALID;
  pp_ctx* ctx = s->ctx;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  DenseView d = s->d;
  d.cost = d_cost_slab;
  hipLaunchKernelGGL(k_tile_unpack, dim3(s->sg), dim3(256), 0, ctx->stream, d, s->run.g, s->run.nblocks);
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

// ------------------------------------------------------------------ slabs of one process, driven together
// The exchange loop a multi-GPU driver runs over its ranks' slab sessions, for the sessions of one
// process (one context per device): every round each slab relaxes what it has pending up to the
// limit, on a host thread of its own, neighbours' boundary columns are copied across, and the
// lowest value pending anywhere plus `band` is the next limit.  The fixpoint does not depend on
// the schedule, so the slabs end up holding the costs a single relaxation of the whole raster
// gives.
extern "C" int32_t pp_slab_relax_all(pp_slab* const* slabs, uint32_t n, double band, uint64_t* rounds_out) {
  if (!slabs || n == 0 || band != band) return PP_ERR_INVALID;
  if (!(band > 0.0)) band = 1024.0;
  for (uint32_t k = 0; k < n; ++k) {
    if (!slabs[k] || !slabs[k]->ctx) return PP_ERR_INVALID;
    for (uint32_t j = 0; j < k; ++j)
      if (slabs[j] == slabs[k] || slabs[j]->ctx == slabs[k]->ctx) return PP_ERR_INVALID;  // a context drives one slab
    if (slabs[k]->d.yw != slabs[0]->d.yw) return PP_ERR_INVALID;
    // column order, no gap: slab k's first owned column follows slab k - 1's last
    if (k && slabs[k]->d.gx_off + 1 != slabs[k - 1]->d.gx_off + (int64_t)slabs[k - 1]->d.ncols - 1) return PP_ERR_INVALID;
  }
  const uint64_t yw = slabs[0]->d.yw;
  const double inf = __builtin_inf();
  struct Side {
    double *edge_lo = nullptr, *edge_hi = nullptr, *halo_lo = nullptr, *halo_hi = nullptr;
  };
  std::vector<Side> side(n);
  auto release = [&] {
    for (uint32_t k = 0; k < n; ++k) {
      (void)hipSetDevice(slabs[k]->ctx->device);
      for (double* p : {side[k].edge_lo, side[k].edge_hi, side[k].halo_lo, side[k].halo_hi})
        if (p) (void)hipFree(p);
    }
  };
  auto fail = [&](int32_t code) {
    release();
    return code;
  };
  for (uint32_t k = 0; k < n; ++k) {
    pp_ctx* ctx = slabs[k]->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(PP_ERR_HIP);
    // a side with no neighbour has no edge buffer: its values are nobody's halo
    for (double** p : {k > 0 ? &side[k].edge_lo : nullptr, k + 1 < n ? &side[k].edge_hi : nullptr,
                       k > 0 ? &side[k].halo_lo : nullptr, k + 1 < n ? &side[k].halo_hi : nullptr}) {
      if (!p) continue;
      const hipError_t e = hipMalloc((void**)p, yw * 8);
      if (e != hipSuccess) {
        ctx->last_hip_error = (int32_t)e;
        (void)hipGetLastError();
        return fail(PP_ERR_HIP);
      }
      hipLaunchKernelGGL(k_fill_inf, dim3((unsigned)std::min<uint64_t>((yw + 255) / 256, 1024)), dim3(256), 0,
                         ctx->stream, *p, yw);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(PP_ERR_HIP);
  }
  std::vector<double> pending(n, inf);
  std::vector<int32_t> status(n, PP_OK);
  double limit = band;  // the goal's cost is 0
  bool first = true;
  uint64_t rounds = 0;
  for (;; ++rounds) {
    if (rounds > ((uint64_t)1 << 24)) return fail(PP_ERR_UNSUPPORTED);
    auto relax = [&](uint32_t k) {
      try {
        status[k] = pp_slab_relax(slabs[k], first ? nullptr : side[k].halo_lo, first ? nullptr : side[k].halo_hi, limit,
                                  side[k].edge_lo, side[k].edge_hi, &pending[k], nullptr);
      } catch (...) {
        status[k] = PP_ERR_UNSUPPORTED;
      }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 1; k < n; ++k) {
      try {
        th.emplace_back(relax, k);
      } catch (const std::system_error&) {
        relax(k);  // no thread to be had: this slab's turn runs here
      }
    }
    relax(0);
    for (std::thread& t : th) t.join();
    for (uint32_t k = 0; k < n; ++k)
      if (status[k] != PP_OK) return fail(status[k]);
    first = false;
    const double lowest = *std::min_element(pending.begin(), pending.end());
    if (lowest == inf) {
      ++rounds;
      break;
    }
    limit = lowest + band;
    // every relax has waited for its stream, so the edge columns are in place: across they go
    for (uint32_t k = 0; k + 1 < n; ++k) {
      pp_ctx *a = slabs[k]->ctx, *b = slabs[k + 1]->ctx;
      hipError_t e;
      if (a->device == b->device) {
        (void)hipSetDevice(a->device);
        e = hipMemcpy(side[k + 1].halo_lo, side[k].edge_hi, yw * 8, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(side[k].halo_hi, side[k + 1].edge_lo, yw * 8, hipMemcpyDeviceToDevice);
      } else {
        e = hipMemcpyPeer(side[k + 1].halo_lo, b->device, side[k].edge_hi, a->device, yw * 8);
        if (e == hipSuccess) e = hipMemcpyPeer(side[k].halo_hi, a->device, side[k + 1].edge_lo, b->device, yw * 8);
      }
      if (e != hipSuccess) {
        a->last_hip_error = (int32_t)e;
        (void)hipGetLastError();
        return fail(PP_ERR_HIP);
      }
    }
  }
  release();
  if (rounds_out) *rounds_out = rounds;
  return PP_OK;
}

// Columns [begin, end) of slab `rank` out of `world` over xw columns: contiguous, sizes differ by at most one.
static void slab_bounds(uint64_t xw, uint32_t world, uint32_t rank, uint64_t* begin, uint64_t* end) {
  const uint64_t base = xw / world, rem = xw % world;
  *begin = rank * base + std::min<uint64_t>(rank, rem);
  *end = *begin + base + (rank < rem ? 1 : 0);
}

extern "C" int32_t pp_wavefront_tiled(pp_ctx* const* ctxs, uint32_t n_ctx, uint64_t xw, uint64_t yw,
                                      const uint8_t* obs, uint64_t goal_x, uint64_t goal_y, double* cost,
                                      double band, uint64_t* rounds) {
  if (!ctxs || n_ctx == 0 || !obs || !cost) return PP_ERR_INVALID;
  for (uint32_t k = 0; k < n_ctx; ++k) {
    if (!ctxs[k]) return PP_ERR_INVALID;
    for (uint32_t j = 0; j < k; ++j)
      if (ctxs[j] == ctxs[k]) return PP_ERR_INVALID;
  }
  if (xw == 0 || yw == 0 || xw >= ((uint64_t)1 << 31) || yw >= ((uint64_t)1 << 31)) return PP_ERR_UNSUPPORTED;
  if (goal_x >= xw || goal_y >= yw) return PP_ERR_BOUNDS;
  if (n_ctx > xw) return PP_ERR_INVALID;  // a slab owns at least one column
  std::vector<pp_slab*> slabs(n_ctx, nullptr);
  auto leave = [&](int32_t code) {
    for (pp_slab* s : slabs) pp_slab_destroy(s);
    return code;
  };
  std::vector<uint8_t> hs;
  for (uint32_t k = 0; k < n_ctx; ++k) {
    pp_ctx* ctx = ctxs[k];
    uint64_t x0, x1;
    slab_bounds(xw, n_ctx, k, &x0, &x1);
    const uint64_t ncols = x1 - x0 + 2;
    // the slab's raster with one halo column each side, blocked outside the grid
    hs.assign(ncols * yw, 1);
    const uint64_t lo = x0 ? x0 - 1 : 0, hi = std::min(x1 + 1, xw);
    std::memcpy(hs.data() + (lo - (x0 - 1)) * yw, obs + lo * yw, (hi - lo) * yw);
    if (hipSetDevice(ctx->device) != hipSuccess) return leave(PP_ERR_HIP);
    uint8_t* d_obs = nullptr;
    hipError_t e = hipMalloc((void**)&d_obs, hs.size());
    if (e == hipSuccess) e = hipMemcpy(d_obs, hs.data(), hs.size(), hipMemcpyHostToDevice);
    int32_t st = PP_ERR_HIP;
    if (e == hipSuccess) st = pp_slab_create(ctx, xw, yw, x0, x1, d_obs, goal_x, goal_y, &slabs[k]);
    else ctx->last_hip_error = (int32_t)e, (void)hipGetLastError();
    if (d_obs) (void)hipFree(d_obs);  // packed: the session holds its own copy
    if (st != PP_OK) return leave(st);
  }
  int32_t st = pp_slab_relax_all(slabs.data(), n_ctx, band, rounds);
  if (st != PP_OK) return leave(st);
  for (uint32_t k = 0; k < n_ctx; ++k) {
    pp_ctx* ctx = ctxs[k];
    uint64_t x0, x1;
    slab_bounds(xw, n_ctx, k, &x0, &x1);
    const uint64_t ncols = x1 - x0 + 2;
    if (hipSetDevice(ctx->device) != hipSuccess) return leave(PP_ERR_HIP);
    double* d_cost = nullptr;
    hipError_t e = hipMalloc((void**)&d_cost, ncols * yw * 8);
    if (e != hipSuccess) {
      ctx->last_hip_error = (int32_t)e;
      (void)hipGetLastError();
      return leave(PP_ERR_HIP);
    }
    st = pp_slab_read(slabs[k], d_cost);
    if (st == PP_OK)  // the owned columns only
      e = hipMemcpy(cost + x0 * yw, d_cost + yw, (x1 - x0) * yw * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_cost);
    if (st != PP_OK) return leave(st);
    if (e != hipSuccess) {
      ctx->last_hip_error = (int32_t)e;
      return leave(PP_ERR_HIP);
    }
  }
  return leave(PP_OK);
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
