// pp_rs.hip -- Reeds-Shepp kernels: word generation + optimal-path selection for batches of
// pairs (generate_path / calc_optimal_path, reeds_shepp.rs:60-85, :707-725), trajectory sampling
// (generate_local_course / interpolate / global transform, :87-128, :581-705) and the
// analytic-expansion shot of the planner (hybrid_astar.rs:380-415, :450-489).
#include <algorithm>
#include <mutex>

#include "pp_device.h"
#include <vector>

#include "pp_rs.h"
#include <hip/hip_ext.h>
#include "pp_rs_device.h"
#include "pp_expand_device.h"

#ifdef PP_LAB_STAMPS
__device__ long long lab_stamp[32];  // [0, 8): phases of workgroup 0; [16, 28): when each formula wave of k_round was done
#define PP_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) lab_stamp[k] = wall_clock64(); } while (0)
extern "C" int32_t pp_lab_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lab_stamp), sizeof(lab_stamp)) == hipSuccess ? 0 : -1;
}
__device__ inline void lab_stamp_end() { lab_stamp[7] = wall_clock64(); }
#else
#define PP_STAMP(k) do {} while (0)
__device__ inline void lab_stamp_end() {}
#endif
namespace {

// ------------------------------------------------------------------ K3: words, one lane per pair

struct BestSink {
  double maxc;
  double best_total = 0.0;
  int best_gen = -1;
  uint32_t best_word = 0;
  int best_n = 0;
  bool near_tie = false;  // another word's total lies within the guard band of the best one
  double best_len[5] = {0, 0, 0, 0, 0};
  __device__ __forceinline__ void operator()(int gen, uint32_t word, int n, const double* len, double total) {
    double td = total / maxc;  // the selection compares de-normalised totals (:125, :77)
    // min total, ties -> the later path in generation order (fold with `<=`, :72-82)
    bool take = best_gen < 0 || td < best_total || (td == best_total && gen > best_gen);
    if (best_gen >= 0) {
      double eps = PP_RS_GUARD * fmax(td, best_total);
      if (fabs(td - best_total) <= eps) near_tie = true;   // mirror-image words tie mathematically
      else if (take) near_tie = false;                     // clearly better: earlier ties are moot
    }
    if (take) {
      best_total = td;
      best_gen = gen;
      best_word = word;
      best_n = n;
#pragma unroll
      for (int i = 0; i < 5; ++i) best_len[i] = len[i];
    }
  }
};

__device__ inline void write_word(pp_rs_word* o, uint32_t word, int n, const double* len_norm,
                                  double total_denorm, double maxc) {
  for (int i = 0; i < 5; ++i) {
    o->lengths[i] = i < n ? len_norm[i] / maxc : 0.0;  // :124
    o->ctypes[i] = i < n ? pp_word_letter(word, i) : 0;
  }
  o->total_length = total_denorm;
  o->nseg = (uint8_t)n;
  o->pad_[0] = o->pad_[1] = 0;
}

// Two waves per SIMD measured best for this register-hungry kernel (1.68 G pairs/s; the compiler's
// own choice of one wave gives 1.1, three waves spill and give 1.4, four 0.97).
#ifndef PP_RS_BATCH_WAVES
#define PP_RS_BATCH_WAVES 3
#endif
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(PP_RS_BATCH_WAVES, PP_RS_BATCH_WAVES)))
k_rs_batch(uint64_t n, const double* __restrict__ sx, const double* __restrict__ sy,
           const double* __restrict__ syaw, const double* __restrict__ gx,
           const double* __restrict__ gy, const double* __restrict__ gyaw, double maxc,
           int32_t* __restrict__ n_paths, int32_t* __restrict__ best_idx,
           pp_rs_word* __restrict__ best, uint8_t* __restrict__ flags) {
  // each wave's areas for the formulas it evaluates apart (pp_rs_generate_wave)
  __shared__ __attribute__((aligned(16))) unsigned char area[2][(kRsDeferredBytes + 15) & ~(size_t)15];
  const uint32_t lane = threadIdx.x & 63u;
  const PpRsDeferred D = pp_deferred_areas(area[threadIdx.x >> 6]);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // a wave goes round as long as any of its lanes has a pair: the lanes work on one another's candidates
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
    const bool have = i0 + lane < n;
    const uint64_t i = have ? i0 + lane : n - 1;  // a lane without a pair repeats the last one and stores nothing
    PpRsState S;
    BestSink sink;
    sink.maxc = maxc;
    const PpRsInput in = pp_rs_input(sx[i], sy[i], syaw[i], gx[i], gy[i], gyaw[i], maxc);
    pp_rs_generate_wave(in, S, D, lane, sink);
    if (!have) continue;
    int np = __popcll(S.accepted);
    if (S.degenerate) {  // the reference panics inside generate_path: no paths reported
      n_paths[i] = 0;
      best_idx[i] = -1;
      pp_rs_word z{};
      best[i] = z;
      flags[i] = 1 | (S.A.amb ? 2 : 0);
      continue;
    }
    n_paths[i] = np;
    if (np == 0) {
      best_idx[i] = -1;
      pp_rs_word z{};
      best[i] = z;
    } else {
      best_idx[i] = __popcll(S.accepted & ((1ull << sink.best_gen) - 1ull));
      write_word(&best[i], sink.best_word, sink.best_n, sink.best_len, sink.best_total, maxc);
    }
    flags[i] = (S.A.amb ? 2 : 0) | (sink.near_tie ? 8 : 0);
  }
}

// ------------------------------------------------------------------ all words of a pair

struct AllSink {
  PpRsCand* slots;  // 46 entries indexed by generation index
  __device__ __forceinline__ void operator()(int gen, uint32_t word, int n, const double* len, double total) {
    PpRsCand c;
#pragma unroll
    for (int i = 0; i < 5; ++i) c.len[i] = len[i];
    c.total = total;
    c.word = word;
    c.nseg = n;
    slots[gen] = c;
  }
};

// Words of n pairs, every accepted candidate kept.  A block takes 64 pairs; its seven waves each
// evaluate one candidate family for all of them (the families' duplicate tests are independent),
// so a small batch -- one analytic shot per planner round -- waits for the longest family, not
// for all 46 candidates in a row.
__global__ void __launch_bounds__(64 * kRsFamilies)
k_rs_words(uint64_t n, const double* __restrict__ sx, const double* __restrict__ sy,
           const double* __restrict__ syaw, const double* __restrict__ gx,
           const double* __restrict__ gy, const double* __restrict__ gyaw, double maxc,
           PpRsCand* __restrict__ cands, PpRsPair* __restrict__ pairs) {
  __shared__ unsigned long long acc[kRsFamilies][64];
  __shared__ uint32_t fl[kRsFamilies][64];
  const uint32_t lane = threadIdx.x & 63u, fam = threadIdx.x >> 6;
  for (uint64_t b0 = (uint64_t)blockIdx.x * 64; b0 < n; b0 += (uint64_t)gridDim.x * 64) {
    const uint64_t i = b0 + lane;
    PpRsState S;
    if (i < n) {
      const PpRsInput in = pp_rs_input(sx[i], sy[i], syaw[i], gx[i], gy[i], gyaw[i], maxc);
      AllSink sink{cands + i * PP_RS_MAX_PATHS};
      switch (fam) {  // wave-uniform
        case 0: pp_rs_family<0>(in, S, sink); break;
        case 1: pp_rs_family<1>(in, S, sink); break;
        case 2: pp_rs_family<2>(in, S, sink); break;
        case 3: pp_rs_family<3>(in, S, sink); break;
        case 4: pp_rs_family<4>(in, S, sink); break;
        case 5: pp_rs_family<5>(in, S, sink); break;
        default: pp_rs_family<6>(in, S, sink); break;
      }
    }
    acc[fam][lane] = S.accepted;
    fl[fam][lane] = (S.degenerate ? 4u : 0u) | (S.A.amb ? 2u : 0u);
    __syncthreads();
    if (fam == 0 && i < n) {
      unsigned long long a = 0;
      uint32_t f = 0;
#pragma unroll
      for (int k = 0; k < kRsFamilies; ++k) {
        a |= acc[k][lane];
        f |= fl[k][lane];
      }
      PpRsPair p;
      p.accepted = (f & 4u) ? 0ull : a;
      p.flags = f;
      p.n_paths = __popcll(p.accepted);
      pairs[i] = p;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ local course bookkeeping

// One segment of one path, as generate_local_course walks it (:601-655).
struct Seg {
  double pd0, d, l;        // first sample offset, signed step, segment length (normalised)
  double ox, oy, oyaw;     // segment origin = end of the previous segment (start frame, metres)
  double co, so;           // cos / sin of oyaw
  int32_t first;           // trajectory index of the first sample
  int32_t n;               // number of samples (loop iterations of :625-640)
  int32_t type;            // 1 L, 2 S, 3 R
  int32_t pad_;
};
struct PathPlan {
  Seg seg[5];
  int32_t nseg;
  int32_t npts;            // after the trailing-zero trim (:656-661)
  int32_t dir0;
  uint32_t amb;
};

// interpolate, :665-705, with the segment origin's cos/sin precomputed.
__device__ inline void interp(const Seg& s, double pd, double maxc, double& px, double& py,
                              double& pyaw, int& dir) {
  if (s.type == 2) {
    px = s.ox + pd / maxc * s.co;
    py = s.oy + pd / maxc * s.so;
    pyaw = s.oyaw;
  } else {
    double sl, cl;
    sincos(pd, &sl, &cl);
    double ldx = sl / maxc;
    double ldy = (s.type == 1) ? (1.0 - cl) / maxc : -(1.0 - cl) / maxc;
    // cos(-oyaw) = co, sin(-oyaw) = -so
    double gdx = s.co * ldx + (-s.so) * ldy;
    double gdy = s.so * ldx + s.co * ldy;
    px = s.ox + gdx;
    py = s.oy + gdy;
    pyaw = (s.type == 1) ? s.oyaw + pd : s.oyaw - pd;
  }
  dir = pd > 0.0 ? 1 : -1;
}

// Index bookkeeping of generate_local_course without walking the samples: the loop
// `while |pd| <= |l| { pd += d }` runs n = floor((l - pd0)/d) + 1 times when it runs at all.
// The count is verified against the loop condition at both ends; a verification inside the
// guard band is reported.
__device__ inline void plan_path(const PpRsCand& c, double maxc, double step, PathPlan& P) {
  P.nseg = c.nseg;
  P.dir0 = c.len[0] > 0.0 ? 1 : -1;
  P.amb = 0;
  double ll = 0.0;
  double ox = 0.0, oy = 0.0, oyaw = 0.0;
  int32_t ind = 1;
  for (int i = 0; i < c.nseg; ++i) {
    Seg& s = P.seg[i];
    double l = c.len[i];
    double d = l > 0.0 ? step : -step;
    s.type = (int32_t)((c.word >> (4 * i)) & 15u);
    s.l = l;
    s.d = d;
    s.ox = ox;
    s.oy = oy;
    s.oyaw = oyaw;
    sincos(oyaw, &s.so, &s.co);
    double pd0;
    if (i >= 1 && c.len[i - 1] * c.len[i] > 0.)
      pd0 = -d - ll;
    else
      pd0 = d - ll;
    s.pd0 = pd0;
    int32_t first = ind - 1 + 1;  // ind -= 1, then the first write goes to ind + 1
    s.first = first;
    int32_t n = 0;
    const double al = fabs(l);
    if (fabs(pd0) <= al) {
      double mm = floor((l - pd0) / d);
      if (!(mm >= 0.0)) mm = 0.0;
      if (mm > 1.0e7) mm = 1.0e7;
      while (fabs(pd0 + (mm + 1.0) * d) <= al) mm += 1.0;
      while (mm > 0.0 && fabs(pd0 + mm * d) > al) mm -= 1.0;
      n = (int32_t)mm + 1;
      if (fabs(fabs(pd0 + mm * d) - al) < PP_RS_GUARD || fabs(fabs(pd0 + (mm + 1.0) * d) - al) < PP_RS_GUARD)
        P.amb = 1;
    } else if (fabs(fabs(pd0) - al) < PP_RS_GUARD) {
      P.amb = 1;
    }
    s.n = n;
    double pd_exit = pd0 + (double)n * d;
    ll = l - pd_exit - d;
    ind = first + n;  // index of the segment's end point
    // the segment's end point is the next segment's origin
    double ex, ey, eyaw;
    int dir;
    interp(s, l, maxc, ex, ey, eyaw, dir);
    ox = ex;
    oy = ey;
    oyaw = eyaw;
  }
  P.npts = ind + 1;
}

// plan_path in two steps, for kernels that have lanes to spare.  The heading at a segment's
// origin is a sum of signed segment lengths and its end point is the origin plus a displacement
// that depends on that heading and the length only, so all sines and cosines of a path can be
// taken side by side (one lane per segment); what is left for the path's own lane is additions
// and the sample counts.  Same operations on the same values as plan_path, hence the same bits.
__device__ inline void plan_segment(const PpRsCand& c, int i, double maxc, Seg& s) {
  double oyaw = 0.0;
  for (int k = 0; k < i; ++k) {
    const uint32_t t = (c.word >> (4 * k)) & 15u;
    if (t == 1u) oyaw = oyaw + c.len[k];
    else if (t != 2u) oyaw = oyaw - c.len[k];
  }
  s.type = (int32_t)((c.word >> (4 * i)) & 15u);
  s.l = c.len[i];
  s.oyaw = oyaw;
  sincos(oyaw, &s.so, &s.co);
  // displacement of the end point, parked in ox / oy until plan_chain turns it into origins
  double px, py, pyaw;
  int dir;
  s.ox = 0.0;
  s.oy = 0.0;
  interp(s, s.l, maxc, px, py, pyaw, dir);
  s.ox = px;
  s.oy = py;
}

__device__ inline void plan_chain(const PpRsCand& c, double step, PathPlan& P) {
  P.nseg = c.nseg;
  P.dir0 = c.len[0] > 0.0 ? 1 : -1;
  P.amb = 0;
  double ll = 0.0;
  double ox = 0.0, oy = 0.0;
  int32_t ind = 1;
  for (int i = 0; i < c.nseg; ++i) {
    Seg& s = P.seg[i];
    const double l = c.len[i];
    const double d = l > 0.0 ? step : -step;
    const double gdx = s.ox, gdy = s.oy;
    s.d = d;
    s.ox = ox;
    s.oy = oy;
    double pd0;
    if (i >= 1 && c.len[i - 1] * c.len[i] > 0.)
      pd0 = -d - ll;
    else
      pd0 = d - ll;
    s.pd0 = pd0;
    const int32_t first = ind;
    s.first = first;
    int32_t n = 0;
    const double al = fabs(l);
    if (fabs(pd0) <= al) {
      double mm = floor((l - pd0) / d);
      if (!(mm >= 0.0)) mm = 0.0;
      if (mm > 1.0e7) mm = 1.0e7;
      while (fabs(pd0 + (mm + 1.0) * d) <= al) mm += 1.0;
      while (mm > 0.0 && fabs(pd0 + mm * d) > al) mm -= 1.0;
      n = (int32_t)mm + 1;
      if (fabs(fabs(pd0 + mm * d) - al) < PP_RS_GUARD || fabs(fabs(pd0 + (mm + 1.0) * d) - al) < PP_RS_GUARD)
        P.amb = 1;
    } else if (fabs(fabs(pd0) - al) < PP_RS_GUARD) {
      P.amb = 1;
    }
    s.n = n;
    const double pd_exit = pd0 + (double)n * d;
    ll = l - pd_exit - d;
    ind = first + n;
    ox = ox + gdx;
    oy = oy + gdy;
  }
  P.npts = ind + 1;
}

// Trajectory point j of a planned path in the start frame.
__device__ inline void local_point(const PathPlan& P, int j, double maxc, double& px, double& py,
                                   double& pyaw, int& dir) {
  if (j == 0) {
    px = py = pyaw = 0.0;
    dir = P.dir0;
    return;
  }
  int i = 0;
#pragma unroll
  for (int k = 1; k < 5; ++k)
    if (k < P.nseg && P.seg[k].first <= j) i = k;
  const Seg& s = P.seg[i];
  int m = j - s.first;
  double pd = (m < s.n) ? s.pd0 + (double)m * s.d : s.l;
  interp(s, pd, maxc, px, py, pyaw, dir);
}

// :656-661 -- drop trailing points whose local x is exactly 0.0
__device__ inline void trim_path(PathPlan& P, double maxc) {
  while (P.npts > 0) {
    double px, py, pyaw;
    int dir;
    local_point(P, P.npts - 1, maxc, px, py, pyaw, dir);
    if (px != 0.0) break;
    P.npts--;
  }
}

// global transform, :111-122
__device__ inline void to_global(double lx, double ly, double lyaw, double x0, double y0, double yaw0,
                                 double c0, double s0, double& X, double& Y, double& YAW) {
  X = lx * c0 + ly * (-s0) + x0;   // ix cos(-yaw0) + iy sin(-yaw0) + x0
  Y = lx * s0 + ly * c0 + y0;      // -ix sin(-yaw0) + iy cos(-yaw0) + y0
  YAW = pp_pi_2_pi(lyaw + yaw0);
}

// Pose j of a planned path for the collision test only: position and heading DIRECTION in the
// global frame.  The heading is the segment origin's heading turned by the arc angle and by the
// start yaw (two complex products instead of a second sin/cos), and 1/maxc is multiplied instead
// of divided by: the pose moves by a few ulp (1e-15 m), nine orders below the guard bands of the
// collision predicate, which report any decision that close to an edge as ambiguous.
__device__ inline void global_pose_for_test(const PathPlan& P, int j, double rmaxc, double x0, double y0,
                                            double c0, double s0, double& X, double& Y, double& C, double& S) {
  double lx = 0.0, ly = 0.0, hc = 1.0, hs = 0.0;
  if (j != 0) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k)
      if (k < P.nseg && P.seg[k].first <= j) i = k;
    const Seg& g = P.seg[i];
    const int m = j - g.first;
    const double pd = (m < g.n) ? g.pd0 + (double)m * g.d : g.l;
    if (g.type == 2) {
      const double r = pd * rmaxc;
      lx = g.ox + r * g.co;
      ly = g.oy + r * g.so;
      hc = g.co;
      hs = g.so;
    } else {
      double sl, cl;
      sincos(pd, &sl, &cl);
      const double ldx = sl * rmaxc;
      double ldy = (1.0 - cl) * rmaxc;
      if (g.type != 1) {  // R: mirror image of L
        ldy = -ldy;
        sl = -sl;
      }
      lx = g.ox + (g.co * ldx - g.so * ldy);
      ly = g.oy + (g.so * ldx + g.co * ldy);
      hc = g.co * cl - g.so * sl;
      hs = g.so * cl + g.co * sl;
    }
  }
  X = lx * c0 - ly * s0 + x0;
  Y = lx * s0 + ly * c0 + y0;
  C = hc * c0 - hs * s0;
  S = hs * c0 + hc * s0;
}

// hybrid_astar.rs:450-489 on the de-normalised lengths
__host__ __device__ inline double rs_cost(const double* len, uint32_t word, int n, const PpCostParams& cp) {
  double cost = 0.0;
  for (int i = 0; i < n; ++i) {
    if (len[i] >= 0.0)
      cost += 1.0;
    else
      cost += fabs(len[i]) * cp.backward_cost;
  }
  for (int i = 0; i + 1 < n; ++i)
    if (len[i] * len[i + 1] < 0.0) cost += cp.gear_cost;
  for (int i = 0; i < n; ++i)
    if (((word >> (4 * i)) & 15u) != 2u) cost += cp.steer_angle_cost * fabs(cp.max_steer);
  double ul[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    uint32_t c = (word >> (4 * i)) & 15u;
    if (c == 3u) ul[i] = -cp.max_steer;
    else if (c == 1u) ul[i] = cp.max_steer;
  }
  for (int i = 0; i + 1 < n; ++i) cost += cp.steer_change_cost * fabs(ul[i + 1] - ul[i]);
  return cost;
}

// ------------------------------------------------------------------ K4: analytic shot, one workgroup per pair
// (the waves of a workgroup share the sampled poses of a pair: four waves finish a lone shot -- the
// tail of a planner batch -- in a quarter of the time; once the batch alone fills the chip one wave
// per pair keeps every resident wave busy through the steps that only the path lanes take part in:
// 3.5 ms against 6.4 ms for 262144 pairs)

//
// A pair has at most 46 paths but rarely more than ten, and the per-path plans in LDS decide how
// many workgroups a CU holds.  A bulk batch therefore runs twice over the pairs: a pass whose LDS
// holds kShotFewPaths plans takes the pairs that fit, a pass with room for all 46 takes the rest;
// a workgroup whose pair belongs to the other pass leaves at once.

#ifndef PP_SHOT_BULK_THREADS
#define PP_SHOT_BULK_THREADS 64
#endif
#ifndef PP_SHOT_FEW_PATHS
#define PP_SHOT_FEW_PATHS 16
#endif
constexpr int kShotThreadsFew = 256;
constexpr int kShotThreadsBulk = PP_SHOT_BULK_THREADS;
constexpr int kShotFewPaths = PP_SHOT_FEW_PATHS;
constexpr uint32_t kShotSettled = 0x80000000u;  // in hit[]: a pose of the path collides beyond doubt
constexpr uint32_t kShotBulkMin = 16384;  // pairs; above this every CU holds many workgroups anyway

// handles the pairs with np_above < n_paths <= kMaxPaths
template <int kShotThreads, int kMaxPaths>
__device__ __forceinline__ void
pp_shot_pair(const uint32_t pair, int np_above, const double x0, const double y0, const double yaw0, double maxc,
             double step_size, uint32_t check_step,
             const PpCostParams& cp, const PpRsCand* pc /* the pair's 46 candidate slots */, const PpRsPair pr,
             const PpFootprint& f, const PpLattice& lat, const PpBuckets& bk, int use_lattice,
             PpShotOut* __restrict__ out, pp_rs_word* __restrict__ word_out /* optional: the accepted path's word */) {
  __shared__ PathPlan plans[kMaxPaths];
  __shared__ double costs[kMaxPaths];
  __shared__ uint32_t hit[kMaxPaths];
  __shared__ uint8_t gen_of[kMaxPaths];
  __shared__ uint32_t n_samples[kMaxPaths + 1];  // sampled poses per path; [kMaxPaths]: their maximum
  __shared__ uint32_t n_settled;
  const int lane = threadIdx.x;
  const int np = pr.n_paths;
  if (np <= np_above || np > kMaxPaths) return;  // uniform over the workgroup
  // path k is the k-th set bit of the accepted mask
  if (lane < 64 && ((pr.accepted >> lane) & 1ull))
    gen_of[__builtin_popcountll(pr.accepted & ((1ull << lane) - 1ull))] = (uint8_t)lane;
  __syncthreads();
  PP_STAMP(3);
  const double step = step_size * maxc;
  // the paths live on the first np lanes of wave 0 (np <= 46) from here to the end
  for (int t = lane; t < np * 5; t += kShotThreads) {
    const int p = t / 5, i = t - 5 * p;
    const PpRsCand c = pc[gen_of[p]];
    if (i < c.nseg) plan_segment(c, i, maxc, plans[p].seg[i]);
  }
  __syncthreads();
  PP_STAMP(4);
  uint32_t my_tasks = 0;
  if (lane < np) {
    const PpRsCand c = pc[gen_of[lane]];
    plan_chain(c, step, plans[lane]);
    trim_path(plans[lane], maxc);
    if (kShotThreads < 128) {
      double len[5];
      for (int i = 0; i < 5; ++i) len[i] = c.len[i] / maxc;
      costs[lane] = rs_cost(len, c.word, c.nseg, cp);
    }
    hit[lane] = 0;
    my_tasks = (uint32_t)((plans[lane].npts + (int)check_step - 1) / (int)check_step);
  }
  // sampled poses per path (read by the task loop) and their maximum
  if (lane < np) n_samples[lane] = my_tasks;
  if (kShotThreads >= 128 && lane >= 64 && lane < 64 + np) {  // the costs meanwhile, on the second wave
    const PpRsCand c = pc[gen_of[lane - 64]];
    double len[5];
    for (int i = 0; i < 5; ++i) len[i] = c.len[i] / maxc;
    costs[lane - 64] = rs_cost(len, c.word, c.nseg, cp);
  }
  if (lane < 64) {
    uint32_t mx = my_tasks;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 0) {
      n_samples[kMaxPaths] = mx;
      n_settled = 0;
    }
  }
  __syncthreads();
  PP_STAMP(5);
  // Tasks in pose-major order: slot t is sample t / np of path t % np (slots past a path's last
  // sample are empty), so the early samples of every path come first.  is_collision returns at a
  // path's first colliding pose, and a path with a clear hit is settled whatever its other poses
  // say; once every path is settled the remaining samples cannot change the outcome and the
  // workgroup stops testing.  (A stale count only means a few tests too many.)
  const uint32_t nslots = np ? n_samples[kMaxPaths] * (uint32_t)np : 0u;
  const double rmaxc = 1.0 / maxc;
  double s0, c0;
  sincos(yaw0, &s0, &c0);
  for (uint32_t t = lane; t < nslots; t += kShotThreads) {
    if (*(volatile uint32_t*)&n_settled >= (uint32_t)np) break;
    const uint32_t k = t / (uint32_t)np;
    const int p = (int)(t - k * (uint32_t)np);
    // a settled path needs no more tests (a candidate can be thousands of metres long, and every
    // lane of the workgroup would otherwise walk it to its end)
    if (k >= n_samples[p] || (*(volatile uint32_t*)&hit[p] & kShotSettled)) continue;
    const int j = (int)k * (int)check_step;
    double X, Y, c, s;
    global_pose_for_test(plans[p], j, rmaxc, x0, y0, c0, s0, X, Y, c, s);
    unsigned r = use_lattice ? pp_pose_vs_lattice(X, Y, c, s, f, lat) : pp_pose_vs_buckets(X, Y, c, s, f, bk);
    if (r) {
      const bool clear_hit = (r & PP_F_HIT) && !(r & PP_F_AMBIG);
      const unsigned old = atomicOr(&hit[p], clear_hit ? (r | kShotSettled) : r);
      if (clear_hit && !(old & kShotSettled)) atomicAdd(&n_settled, 1u);
    }
  }
  __syncthreads();
  PP_STAMP(6);
  if (lane >= 64) return;
  // Pop order of BinaryHeap<(Reverse(cost), i)>: cost ascending, ties -> larger i.  Each lane ranks
  // its own path; the accepted path is the collision-free one of lowest rank, and the paths popped
  // on the way to it are those of rank up to its rank (all of them when none is free).
  const bool mine = lane < np;
  const double myc = mine ? costs[lane] : 0.0;
  int rank = 0;
  bool nan_cost = mine && myc != myc;  // NotNan::new(..).unwrap() panics while the heap is filled
  for (int q = 0; q < np; ++q) {
    const double cq = costs[q];
    if (cq < myc || (cq == myc && q > lane)) ++rank;
  }
  const uint32_t myhit = mine ? hit[lane] : 0u;
  const int key = (mine && !(myhit & PP_F_HIT)) ? rank : 0x7fffffff;
  int best = key;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = min(best, __shfl_xor(best, d, 64));
  const bool popped = mine && rank <= best;
  uint32_t handed = popped ? my_tasks : 0u;
  uint32_t fl = 0;
  if (popped) {
    // a guard-band decision on some pose does not matter to a path that has a clear hit elsewhere
    if (plans[lane].amb || ((myhit & PP_F_AMBIG) && !(myhit & kShotSettled))) fl |= 2u;
    if (myhit & PP_F_NONFINITE) fl |= 16u;
  }
  if (nan_cost) fl |= 8u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    handed += (uint32_t)__shfl_xor((int)handed, d, 64);
    fl |= (uint32_t)__shfl_xor((int)fl, d, 64);
  }
  const uint64_t winners = __ballot(key == best && best != 0x7fffffff);
  const int writer = winners ? (int)__builtin_ctzll(winners) : 0;
  if (lane == writer) {
    PpShotOut o{};
    o.n_paths = np;
    o.flags = pr.flags | fl;
    o.path_idx = -1;
    o.poses_handed = handed;
    if (winners) {
      o.path_idx = lane;
      o.gen = gen_of[lane];
      o.cost = myc;
      o.flags |= 1u;
    }
    out[pair] = o;
    if (blockIdx.x == 0) lab_stamp_end();
    if (word_out) {
      pp_rs_word wz{};
      if (o.path_idx >= 0) {
        const PpRsCand c = pc[o.gen];
        write_word(&wz, c.word, c.nseg, c.len, c.total / maxc, maxc);
      }
      word_out[pair] = wz;
    }
  }
}

template <int kShotThreads, int kMaxPaths>
__global__ void __launch_bounds__(kShotThreads)
k_rs_shot(uint32_t n, int np_above, const double* __restrict__ sx, const double* __restrict__ sy,
          const double* __restrict__ syaw, double maxc, double step_size, uint32_t check_step,
          PpCostParams cp, const PpRsCand* __restrict__ cands, const PpRsPair* __restrict__ pairs,
          PpFootprint f, PpLattice lat, PpBuckets bk, int use_lattice, PpShotOut* __restrict__ out,
          pp_rs_word* __restrict__ word_out) {
  if (blockIdx.x >= n) return;
  pp_shot_pair<kShotThreads, kMaxPaths>(blockIdx.x, np_above, sx[blockIdx.x], sy[blockIdx.x], syaw[blockIdx.x], maxc,
                                        step_size, check_step, cp,
                                        cands + (size_t)blockIdx.x * PP_RS_MAX_PATHS, pairs[blockIdx.x], f, lat, bk,
                                        use_lattice, out, word_out);
}

// One launch for a planner round of few nodes: workgroups [0, n) take the nodes' Reeds-Shepp
// words and analytic shots, the workgroups behind them the nodes' motion primitives (which do not
// depend on the shots), so a round is one kernel and the primitives cost it no time of their own.
struct PpRoundExpand {
  const double *px, *py, *pyaw, *psteer, *pcost, *motion;
  const int64_t* pdir;
  uint32_t n_prim;
  PpExpandParams prm;
  PpExpandDev out;
  // optional: news for the tables of known cells (pp_expand.h), applied by the first `ub`
  // workgroups of the launch -- by the time the primitives reach their lookups, at the end of
  // their rollouts, it is in; a lookup that comes too early only rules out less
  uint32_t ub = 0, n_upd = 0;
  const PpFtabUpd* upd = nullptr;
  unsigned long long* ftab_rw = nullptr;
};

constexpr int kRoundThreads = 1024;  // sixteen waves: the twelve candidate groups side by side

__global__ void __launch_bounds__(kRoundThreads)
k_round(uint32_t n, const double* __restrict__ q6 /* sx sy syaw gx gy gyaw, n each */, double maxc,
        double step_size, uint32_t check_step, PpCostParams cp, PpFootprint f, PpLattice lat, PpBuckets bk,
        int use_lattice, PpShotOut* __restrict__ out, PpRoundExpand e) {
  PP_STAMP(0);
  if (blockIdx.x < e.ub) {
    const uint32_t t = blockIdx.x * kRoundThreads + threadIdx.x;
    if (t < e.n_upd) pp_ftab_update_one(e.upd[t], e.ftab_rw, e.out.ftab_slots);
    return;
  }
  const uint32_t blk = blockIdx.x - e.ub;
  if (blk < n) {
    // The pair's words first: wave g < 12 evaluates formula g, its lanes 0..3 the four mirror
    // images; then lane s < 20 of wave 0 runs set_path's duplicate test over the candidates of
    // word s in the reference's order.  Candidates and the pair record stay in LDS for the shot.
    __shared__ PpRsCand s_cands[PP_RS_MAX_PATHS];
    __shared__ PpRsRaw s_raw[PP_RS_MAX_PATHS];
    __shared__ unsigned long long s_acc;
    __shared__ uint32_t s_fl;
    __shared__ PpRsPair s_pair;
    __shared__ double s_start[3];
    const uint32_t i = blk, wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
    if (threadIdx.x == 0) {
      s_acc = 0ull;
      s_fl = 0u;
    }
    if (wave < 12u && wl < 4u) {
      // the start pose comes over PCIe when the round runs on the pinned area: read once, kept in LDS
      const double qx = q6[i], qy = q6[n + i], qyaw = q6[2 * (size_t)n + i];
      if (threadIdx.x == 0) {
        s_start[0] = qx;
        s_start[1] = qy;
        s_start[2] = qyaw;
      }
      const PpRsInput in = pp_rs_input(qx, qy, qyaw, q6[3 * (size_t)n + i], q6[4 * (size_t)n + i],
                                       q6[5 * (size_t)n + i], maxc);
      PpAmb A{false};
      pp_rs_eval_wave(in, (int)wave, (int)wl, A, s_raw);
#ifdef PP_LAB_STAMPS
      if (wl == 0 && blockIdx.x == 0) lab_stamp[16 + wave] = wall_clock64();
#endif
      if (A.amb) atomicOr(&s_fl, 2u);
    }
    __syncthreads();
    PP_STAMP(1);
    if (threadIdx.x < (uint32_t)kRsWordSets) {
      uint32_t fl = 0;
      const unsigned long long acc = pp_rs_commit_set((int)threadIdx.x, s_raw, s_cands, &fl);
      if (acc) atomicOr(&s_acc, acc);
      if (fl) atomicOr(&s_fl, fl);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      PpRsPair p;
      p.accepted = (s_fl & 4u) ? 0ull : s_acc;
      p.flags = s_fl;
      p.n_paths = __popcll(p.accepted);
      s_pair = p;
    }
    __syncthreads();
    PP_STAMP(2);
    pp_shot_pair<kRoundThreads, PP_RS_MAX_PATHS>(i, -1, s_start[0], s_start[1], s_start[2], maxc, step_size, check_step,
                                                 cp, s_cands, s_pair, f, lat, bk, use_lattice, out, nullptr);
    return;
  }
  const uint64_t items = (uint64_t)n * e.n_prim;
  const uint64_t stride = (uint64_t)(gridDim.x - e.ub - n) * kRoundThreads;
  for (uint64_t t = (uint64_t)(blk - n) * kRoundThreads + threadIdx.x; t < items; t += stride)
    pp_expand_item(t, e.n_prim, e.px, e.py, e.pyaw, e.pdir, e.psteer, e.pcost, e.motion, e.prm, f, lat, bk,
                   use_lattice, nullptr, e.out);
}

// ------------------------------------------------------------------ full trajectories of one pair

__global__ void __launch_bounds__(64)
k_rs_traj(double x0, double y0, double yaw0, double maxc, double step_size,
          const PpRsCand* __restrict__ cands, const PpRsPair* __restrict__ pairs, int only_rank,
          uint64_t* __restrict__ traj_off, double* __restrict__ tx, double* __restrict__ ty,
          double* __restrict__ tyaw, int64_t* __restrict__ tdir, uint64_t cap,
          uint32_t* __restrict__ status) {
  __shared__ PathPlan plans[PP_RS_MAX_PATHS];
  __shared__ uint8_t gen_of[PP_RS_MAX_PATHS];
  __shared__ uint64_t off[PP_RS_MAX_PATHS + 1];
  const int lane = threadIdx.x;
  const PpRsPair pr = pairs[0];
  const int np = pr.n_paths;
  if (lane == 0) {
    uint64_t m = pr.accepted;
    int k = 0;
    while (m) {
      gen_of[k++] = (uint8_t)__builtin_ctzll(m);
      m &= m - 1;
    }
  }
  __syncthreads();
  const double step = step_size * maxc;
  if (lane < np) {
    plan_path(cands[gen_of[lane]], maxc, step, plans[lane]);
    trim_path(plans[lane], maxc);
  }
  __syncthreads();
  if (lane == 0) {
    uint64_t acc = 0;
    uint32_t amb = 0, empty = 0;
    for (int p = 0; p < np; ++p) {
      off[p] = acc;
      if (only_rank < 0 || only_rank == p) acc += (uint64_t)plans[p].npts;
      amb |= plans[p].amb;
      if (plans[p].npts == 0) empty = 1;  // px.last().unwrap() on an emptied Vec panics
    }
    off[np] = acc;
    for (int p = 0; p <= np; ++p) traj_off[p] = off[p];
    status[0] = (acc > cap) ? 1u : 0u;
    status[1] = amb;
    status[2] = empty;
  }
  __syncthreads();
  if (off[np] > cap) return;
  double s0, c0;
  sincos(yaw0, &s0, &c0);
  for (int p = 0; p < np; ++p) {
    if (only_rank >= 0 && only_rank != p) continue;
    for (int j = lane; j < plans[p].npts; j += 64) {
      double lx, ly, lyaw, X, Y, YAW;
      int dir;
      local_point(plans[p], j, maxc, lx, ly, lyaw, dir);
      to_global(lx, ly, lyaw, x0, y0, yaw0, c0, s0, X, Y, YAW);
      uint64_t o = off[p] + (uint64_t)j;
      tx[o] = X;
      ty[o] = Y;
      tyaw[o] = YAW;
      tdir[o] = dir;
    }
  }
}

// ------------------------------------------------------------------ one chosen path per pair

// Trajectory of the path with rank sel[pair] (generation order among the accepted words) of each
// pair.  One wave per pair.  With offs == nullptr only the point counts are produced; otherwise
// pair p's points go to [offs[p], offs[p + 1]) of the trajectory arrays.
__global__ void __launch_bounds__(64)
k_rs_traj_sel(uint32_t n, const double* __restrict__ sx, const double* __restrict__ sy,
              const double* __restrict__ syaw, double maxc, double step_size,
              const PpRsCand* __restrict__ cands, const PpRsPair* __restrict__ pairs,
              const int32_t* __restrict__ sel, const uint64_t* __restrict__ offs,
              uint32_t* __restrict__ npts, double* __restrict__ total, uint32_t* __restrict__ flags,
              double* __restrict__ tx, double* __restrict__ ty, double* __restrict__ tyaw,
              int64_t* __restrict__ tdir) {
  __shared__ PathPlan plan;
  __shared__ int ok;
  const uint32_t pair = blockIdx.x;
  if (pair >= n) return;
  const int lane = threadIdx.x;
  if (lane == 0) {
    const PpRsPair pr = pairs[pair];
    const int rank = sel[pair];
    uint32_t fl = pr.flags;
    ok = 0;
    uint32_t np = 0;
    double tot = 0.0;
    if (rank >= 0 && rank < pr.n_paths) {
      uint64_t m = pr.accepted;
      for (int k = 0; k < rank; ++k) m &= m - 1;
      const PpRsCand c = cands[(size_t)pair * PP_RS_MAX_PATHS + __builtin_ctzll(m)];
      plan_path(c, maxc, step_size * maxc, plan);
      trim_path(plan, maxc);
      np = (uint32_t)plan.npts;
      tot = c.total / maxc;  // reeds_shepp.rs:125
      if (plan.amb) fl |= 2u;
      if (plan.npts == 0) fl |= 8u;  // px.last().unwrap() on an emptied Vec panics
      ok = offs != nullptr && plan.npts > 0 && (uint64_t)plan.npts == offs[pair + 1] - offs[pair];
    } else {
      fl |= 16u;
    }
    if (!offs) {
      npts[pair] = np;
      total[pair] = tot;
      flags[pair] = fl;
    }
  }
  __syncthreads();
  if (!ok) return;
  const double x0 = sx[pair], y0 = sy[pair], yaw0 = syaw[pair];
  double s0, c0;
  sincos(yaw0, &s0, &c0);
  const uint64_t base = offs[pair];
  for (int j = lane; j < plan.npts; j += 64) {
    double lx, ly, lyaw, X, Y, YAW;
    int dir;
    local_point(plan, j, maxc, lx, ly, lyaw, dir);
    to_global(lx, ly, lyaw, x0, y0, yaw0, c0, s0, X, Y, YAW);
    const uint64_t o = base + (uint64_t)j;
    tx[o] = X;
    ty[o] = Y;
    tyaw[o] = YAW;
    tdir[o] = dir;
  }
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

PpCostParams cost_params(const pp_config* conf) {
  PpCostParams c;
  c.backward_cost = conf->backward_cost;
  c.gear_cost = conf->gear_cost;
  c.steer_angle_cost = conf->steer_angle_cost;
  c.steer_change_cost = conf->steer_change_cost;
  c.max_steer = conf->max_steer;
  return c;
}

}  // namespace

// ---------------------------------------------------------------------------- device-side API

void pp_rs_words_launch(pp_ctx* ctx, uint64_t n, const double* d_q /* 6 arrays of n */, double maxc,
                        PpRsCand* d_cands, PpRsPair* d_pairs) {
  unsigned grid = pp_stream_grid(ctx, n, 64);
  hipLaunchKernelGGL(k_rs_words, dim3(grid), dim3(64 * kRsFamilies), 0, ctx->stream, n, d_q, d_q + n, d_q + 2 * n,
                     d_q + 3 * n, d_q + 4 * n, d_q + 5 * n, maxc, d_cands, d_pairs);
}

void pp_rs_shot_launch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint32_t n,
                       const double* d_q, double maxc, const PpRsCand* d_cands,
                       const PpRsPair* d_pairs, PpShotOut* d_out, pp_rs_word* d_words) {
  if (n == 0) return;
  PpFootprint f = pp_make_footprint(conf);
  uint32_t cs = (uint32_t)std::min<uint64_t>(conf->collision_check_step, 0x7fffffffu);
#define PP_SHOT_LAUNCH(T, MAXP, ABOVE)                                                                          \
  hipLaunchKernelGGL((k_rs_shot<T, MAXP>), dim3(n), dim3(T), 0, ctx->stream, n, ABOVE, d_q, d_q + n,               \
                     d_q + 2 * (size_t)n, maxc, conf->move_step, cs, cost_params(conf), d_cands, d_pairs, f,       \
                     map->lat, map->bk, map->lattice ? 1 : 0, d_out, d_words)
  if (n >= kShotBulkMin) {
    PP_SHOT_LAUNCH(kShotThreadsBulk, kShotFewPaths, -1);
    PP_SHOT_LAUNCH(kShotThreadsBulk, PP_RS_MAX_PATHS, kShotFewPaths);
  } else {
    PP_SHOT_LAUNCH(kShotThreadsFew, PP_RS_MAX_PATHS, -1);
  }
#undef PP_SHOT_LAUNCH
}

void pp_round_launch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, const PpExpandPlan& plan, uint32_t n,
                     const double* d_q, double maxc, PpShotOut* d_out, const int64_t* d_pdir,
                     const double* d_psteer, const double* d_pcost, const double* d_motion, PpExpandDev eo,
                     uint32_t n_upd, const PpFtabUpd* upd, unsigned long long* ftab, hipEvent_t done) {
  if (n == 0) {
    if (done) (void)hipEventRecord(done, ctx->stream);
    return;
  }
  PpFootprint f = pp_make_footprint(conf);
  uint32_t cs = (uint32_t)std::min<uint64_t>(conf->collision_check_step, 0x7fffffffu);
  PpRoundExpand e;
  e.px = d_q;
  e.py = d_q + n;
  e.pyaw = d_q + 2 * (size_t)n;
  e.pdir = d_pdir;
  e.psteer = d_psteer;
  e.pcost = d_pcost;
  e.motion = d_motion;
  e.n_prim = plan.n_prim;
  e.prm = plan.prm;
  e.out = eo;
  const uint64_t items = (uint64_t)n * plan.n_prim;
  const unsigned eb = (unsigned)((items + kRoundThreads - 1) / kRoundThreads);
  if (n_upd && upd && ftab) {
    e.n_upd = n_upd;
    e.ub = (n_upd + kRoundThreads - 1) / kRoundThreads;
    e.upd = upd;
    e.ftab_rw = ftab;
  }
  // `done` rides on the kernel's own completion signal: the host learns of the results 2 us sooner
  // than through an event recorded behind the kernel (35.0 against 37.0 us for the round trip of a
  // 20 us kernel)
  if (done)
    hipExtLaunchKernelGGL(k_round, dim3(e.ub + n + eb), dim3(kRoundThreads), 0, ctx->stream, nullptr, done, 0, n, d_q,
                          maxc, conf->move_step, cs, cost_params(conf), f, map->lat, map->bk, map->lattice ? 1 : 0,
                          d_out, e);
  else
    hipLaunchKernelGGL(k_round, dim3(e.ub + n + eb), dim3(kRoundThreads), 0, ctx->stream, n, d_q, maxc, conf->move_step,
                       cs, cost_params(conf), f, map->lat, map->bk, map->lattice ? 1 : 0, d_out, e);
}

extern "C" int32_t pp_rs_batch_dev(pp_ctx* ctx, uint64_t n, const double* d_sx, const double* d_sy,
                                   const double* d_syaw, const double* d_gx, const double* d_gy,
                                   const double* d_gyaw, double maxc, int32_t* d_n_paths,
                                   int32_t* d_best_idx, pp_rs_word* d_best, uint8_t* d_flags) {
  if (!ctx) return PP_ERR_INVALID;
  if (n == 0) return PP_OK;
  if (!d_sx || !d_sy || !d_syaw || !d_gx || !d_gy || !d_gyaw || !d_n_paths || !d_best_idx || !d_best || !d_flags)
    return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  unsigned grid = pp_stream_grid(ctx, n, 128);
  hipLaunchKernelGGL(k_rs_batch, dim3(grid), dim3(128), 0, ctx->stream, n, d_sx, d_sy, d_syaw, d_gx,
                     d_gy, d_gyaw, maxc, d_n_paths, d_best_idx, d_best, d_flags);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

extern "C" int32_t pp_rs_batch(pp_ctx* ctx, uint64_t n, const double* sx, const double* sy,
                               const double* syaw, const double* gx, const double* gy,
                               const double* gyaw, double maxc, int32_t* n_paths, int32_t* best_idx,
                               pp_rs_word* best, uint8_t* flags) {
  if (!ctx) return PP_ERR_INVALID;
  if (n == 0) return PP_OK;
  if (!sx || !sy || !syaw || !gx || !gy || !gyaw || !n_paths || !best_idx || !best || !flags) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  size_t a8 = al256(n * 8), a4 = al256(n * 4), aw = al256(n * sizeof(pp_rs_word)), a1 = al256(n);
  int32_t st = pp_scratch_reserve(ctx, 6 * a8 + 2 * a4 + aw + a1);
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  double* dq[6];
  const double* hq[6] = {sx, sy, syaw, gx, gy, gyaw};
  hipStream_t s = ctx->stream;
  for (int k = 0; k < 6; ++k) {
    dq[k] = (double*)b;
    b += a8;
    PP_HIP(ctx, hipMemcpyAsync(dq[k], hq[k], n * 8, hipMemcpyHostToDevice, s));
  }
  int32_t* d_np = (int32_t*)b; b += a4;
  int32_t* d_bi = (int32_t*)b; b += a4;
  pp_rs_word* d_w = (pp_rs_word*)b; b += aw;
  uint8_t* d_f = (uint8_t*)b;
  st = pp_rs_batch_dev(ctx, n, dq[0], dq[1], dq[2], dq[3], dq[4], dq[5], maxc, d_np, d_bi, d_w, d_f);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipMemcpyAsync(n_paths, d_np, n * 4, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(best_idx, d_bi, n * 4, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(best, d_w, n * sizeof(pp_rs_word), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(flags, d_f, n, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  return PP_OK;
}

// generate_path for n pairs with every word kept (host pointers): pair i's words in generation
// order at words[i * PP_RS_MAX_PATHS ..], n_paths[i] of them.
extern "C" int32_t pp_rs_words_batch(pp_ctx* ctx, uint64_t n, const double* sx, const double* sy,
                                     const double* syaw, const double* gx, const double* gy,
                                     const double* gyaw, double maxc, int32_t* n_paths, pp_rs_word* words,
                                     uint8_t* flags) {
  if (!ctx) return PP_ERR_INVALID;
  if (n == 0) return PP_OK;
  if (n > 0x7fffffffu) return PP_ERR_UNSUPPORTED;
  if (!sx || !sy || !syaw || !gx || !gy || !gyaw || !n_paths || !words || !flags) return PP_ERR_INVALID;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t a_q = al256(6 * n * 8), a_c = al256(n * PP_RS_MAX_PATHS * sizeof(PpRsCand));
  int32_t st = pp_scratch_reserve(ctx, a_q + a_c + al256(n * sizeof(PpRsPair)));
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  double* d_q = (double*)b;
  PpRsCand* d_c = (PpRsCand*)(b + a_q);
  PpRsPair* d_p = (PpRsPair*)(b + a_q + a_c);
  const double* hq[6] = {sx, sy, syaw, gx, gy, gyaw};
  hipStream_t s = ctx->stream;
  for (int k = 0; k < 6; ++k) PP_HIP(ctx, hipMemcpyAsync(d_q + k * n, hq[k], n * 8, hipMemcpyHostToDevice, s));
  pp_rs_words_launch(ctx, n, d_q, maxc, d_c, d_p);
  PP_HIP(ctx, hipGetLastError());
  std::vector<PpRsCand> hc(n * PP_RS_MAX_PATHS);
  std::vector<PpRsPair> hp(n);
  PP_HIP(ctx, hipMemcpyAsync(hc.data(), d_c, hc.size() * sizeof(PpRsCand), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(hp.data(), d_p, n * sizeof(PpRsPair), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  for (uint64_t i = 0; i < n; ++i) {
    pp_rs_word* out = words + i * PP_RS_MAX_PATHS;
    std::memset(out, 0, sizeof(pp_rs_word) * PP_RS_MAX_PATHS);
    const bool degenerate = (hp[i].flags & 4u) != 0;
    flags[i] = (uint8_t)((degenerate ? 1u : 0u) | (hp[i].flags & 2u));
    n_paths[i] = degenerate ? 0 : hp[i].n_paths;
    if (degenerate) continue;
    uint64_t m = hp[i].accepted;
    int k = 0;
    while (m) {
      const int g = __builtin_ctzll(m);
      m &= m - 1;
      const PpRsCand& c = hc[i * PP_RS_MAX_PATHS + g];
      pp_rs_word& w = out[k++];
      for (int j = 0; j < c.nseg; ++j) {
        w.lengths[j] = c.len[j] / maxc;  // :124
        w.ctypes[j] = pp_word_letter(c.word, j);
      }
      w.nseg = (uint8_t)c.nseg;
      w.total_length = c.total / maxc;   // :125
    }
  }
  return PP_OK;
}

// The chosen path (rank sel[i]) of n pairs with its trajectory, tightly packed: pair i's points
// are [offs[i], offs[i + 1]) of the output vectors.  flags: the pair's word flags, | 8 trajectory
// emptied by the trim, | 16 rank out of range.
int32_t pp_rs_selected_batch(pp_ctx* ctx, uint32_t n, const double* q6 /* 6 arrays of n */, double maxc,
                             double step_size, const int32_t* sel, std::vector<uint64_t>& offs,
                             std::vector<double>& total, std::vector<uint32_t>& flags,
                             const double** tx, const double** ty, const double** tyaw,
                             const int64_t** tdir) {
  *tx = *ty = *tyaw = nullptr;
  *tdir = nullptr;
  offs.assign((size_t)n + 1, 0);
  total.assign(n, 0.0);
  flags.assign(n, 0);
  if (n == 0) return PP_OK;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t b_q = al256((size_t)n * 48), b_c = al256((size_t)n * PP_RS_MAX_PATHS * sizeof(PpRsCand));
  const size_t b_p = al256((size_t)n * sizeof(PpRsPair)), b_4 = al256((size_t)n * 4), b_8 = al256((size_t)n * 8 + 8);
  const size_t fixed = b_q + b_c + b_p + 3 * b_4 + 2 * b_8;
  int32_t st = pp_scratch_reserve(ctx, fixed);
  if (st != PP_OK) return st;
  auto carve = [&](char* b, double*& d_q, PpRsCand*& d_c, PpRsPair*& d_p, int32_t*& d_sel, uint32_t*& d_np,
                   uint32_t*& d_fl, double*& d_tot, uint64_t*& d_off) {
    d_q = (double*)b; b += b_q;
    d_c = (PpRsCand*)b; b += b_c;
    d_p = (PpRsPair*)b; b += b_p;
    d_sel = (int32_t*)b; b += b_4;
    d_np = (uint32_t*)b; b += b_4;
    d_fl = (uint32_t*)b; b += b_4;
    d_tot = (double*)b; b += b_8;
    d_off = (uint64_t*)b;
  };
  double* d_q; PpRsCand* d_c; PpRsPair* d_p; int32_t* d_sel; uint32_t *d_np, *d_fl; double* d_tot; uint64_t* d_off;
  carve((char*)ctx->scratch, d_q, d_c, d_p, d_sel, d_np, d_fl, d_tot, d_off);
  PP_HIP(ctx, hipMemcpyAsync(d_q, q6, (size_t)n * 48, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_sel, sel, (size_t)n * 4, hipMemcpyHostToDevice, s));
  pp_rs_words_launch(ctx, n, d_q, maxc, d_c, d_p);
  // pass 1: point counts
  hipLaunchKernelGGL(k_rs_traj_sel, dim3(n), dim3(64), 0, s, n, d_q, d_q + n, d_q + 2 * (size_t)n, maxc,
                     step_size, d_c, d_p, d_sel, (const uint64_t*)nullptr, d_np, d_tot, d_fl,
                     (double*)nullptr, (double*)nullptr, (double*)nullptr, (int64_t*)nullptr);
  PP_HIP(ctx, hipGetLastError());
  std::vector<uint32_t> np(n);
  PP_HIP(ctx, hipMemcpyAsync(np.data(), d_np, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(flags.data(), d_fl, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(total.data(), d_tot, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  for (uint32_t i = 0; i < n; ++i) offs[i + 1] = offs[i] + np[i];
  const uint64_t tot_pts = offs[n];
  if (tot_pts == 0) return PP_OK;
  // pass 2: the points, packed.  The words are generated again if the scratch buffer has to move.
  const size_t b_t = al256(tot_pts * 8);
  // (the buffer's size tells whether it was replaced; its address does not, a new allocation can be
  // handed the address that was just freed, with other contents)
  const size_t bytes_before = ctx->scratch_bytes;
  st = pp_scratch_reserve(ctx, fixed + 4 * b_t);
  if (st != PP_OK) return st;
  if (ctx->scratch_bytes != bytes_before) {
    carve((char*)ctx->scratch, d_q, d_c, d_p, d_sel, d_np, d_fl, d_tot, d_off);
    PP_HIP(ctx, hipMemcpyAsync(d_q, q6, (size_t)n * 48, hipMemcpyHostToDevice, s));
    PP_HIP(ctx, hipMemcpyAsync(d_sel, sel, (size_t)n * 4, hipMemcpyHostToDevice, s));
    pp_rs_words_launch(ctx, n, d_q, maxc, d_c, d_p);
  }
  char* tb = (char*)ctx->scratch + fixed;
  double* d_tx = (double*)tb;
  double* d_ty = (double*)(tb + b_t);
  double* d_tyaw = (double*)(tb + 2 * b_t);
  int64_t* d_td = (int64_t*)(tb + 3 * b_t);
  PP_HIP(ctx, hipMemcpyAsync(d_off, offs.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_rs_traj_sel, dim3(n), dim3(64), 0, s, n, d_q, d_q + n, d_q + 2 * (size_t)n, maxc,
                     step_size, d_c, d_p, d_sel, (const uint64_t*)d_off, d_np, d_tot, d_fl, d_tx, d_ty, d_tyaw,
                     d_td);
  PP_HIP(ctx, hipGetLastError());
  // the four packed arrays come back in one copy into the context's pinned buffer
  st = pp_pinned_reserve(ctx, 4 * b_t);
  if (st != PP_OK) return st;
  char* hb = (char*)ctx->pinned;
  PP_HIP(ctx, hipMemcpyAsync(hb, tb, 4 * b_t, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  *tx = (const double*)hb;
  *ty = (const double*)(hb + b_t);
  *tyaw = (const double*)(hb + 2 * b_t);
  *tdir = (const int64_t*)(hb + 3 * b_t);
  return PP_OK;
}

// Words (+ optionally trajectories) of one pair.  only_rank < 0: every path.
int32_t pp_rs_single(pp_ctx* ctx, const double q[6], double maxc, double step_size, int only_rank,
                     int32_t* n_paths, pp_rs_word* words, uint64_t* traj_off, double* traj_x,
                     double* traj_y, double* traj_yaw, int64_t* traj_dir, uint64_t traj_cap,
                     uint32_t* flags_out) {
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const bool want_traj = traj_x != nullptr;
  size_t o_q = 0, o_c = o_q + al256(48), o_p = o_c + al256(sizeof(PpRsCand) * PP_RS_MAX_PATHS);
  size_t o_off = o_p + al256(sizeof(PpRsPair)), o_st = o_off + al256(8 * (PP_RS_MAX_PATHS + 1));
  size_t o_t = o_st + 256;
  size_t tb = want_traj ? al256(traj_cap * 8) : 0;
  int32_t st = pp_scratch_reserve(ctx, o_t + 4 * tb);
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  double* d_q = (double*)(b + o_q);
  PpRsCand* d_c = (PpRsCand*)(b + o_c);
  PpRsPair* d_p = (PpRsPair*)(b + o_p);
  uint64_t* d_off = (uint64_t*)(b + o_off);
  uint32_t* d_st = (uint32_t*)(b + o_st);
  hipStream_t s = ctx->stream;
  PP_HIP(ctx, hipMemcpyAsync(d_q, q, 48, hipMemcpyHostToDevice, s));
  pp_rs_words_launch(ctx, 1, d_q, maxc, d_c, d_p);
  PP_HIP(ctx, hipGetLastError());
  double *d_tx = nullptr, *d_ty = nullptr, *d_tyaw = nullptr;
  int64_t* d_td = nullptr;
  if (want_traj) {
    d_tx = (double*)(b + o_t);
    d_ty = (double*)(b + o_t + tb);
    d_tyaw = (double*)(b + o_t + 2 * tb);
    d_td = (int64_t*)(b + o_t + 3 * tb);
    hipLaunchKernelGGL(k_rs_traj, dim3(1), dim3(64), 0, s, q[0], q[1], q[2], maxc, step_size, d_c, d_p,
                       only_rank, d_off, d_tx, d_ty, d_tyaw, d_td, traj_cap, d_st);
    PP_HIP(ctx, hipGetLastError());
  }
  PpRsCand hc[PP_RS_MAX_PATHS];
  PpRsPair hp;
  uint64_t hoff[PP_RS_MAX_PATHS + 1] = {0};
  uint32_t hst[4] = {0, 0, 0, 0};
  PP_HIP(ctx, hipMemcpyAsync(hc, d_c, sizeof(hc), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(&hp, d_p, sizeof(hp), hipMemcpyDeviceToHost, s));
  if (want_traj) {
    PP_HIP(ctx, hipMemcpyAsync(hoff, d_off, sizeof(hoff), hipMemcpyDeviceToHost, s));
    PP_HIP(ctx, hipMemcpyAsync(hst, d_st, 16, hipMemcpyDeviceToHost, s));
  }
  PP_HIP(ctx, hipStreamSynchronize(s));
  if (flags_out) *flags_out = hp.flags | (hst[1] ? 2u : 0u);
  if (hp.flags & 4u) return PP_ERR_DEGENERATE_RS;
  *n_paths = hp.n_paths;
  uint64_t m = hp.accepted;
  int k = 0;
  while (m) {
    int g = __builtin_ctzll(m);
    m &= m - 1;
    const PpRsCand& c = hc[g];
    pp_rs_word& w = words[k++];
    std::memset(&w, 0, sizeof(w));
    for (int i = 0; i < c.nseg; ++i) {
      w.lengths[i] = c.len[i] / maxc;  // :124
      w.ctypes[i] = pp_word_letter(c.word, i);
    }
    w.nseg = (uint8_t)c.nseg;
    w.total_length = c.total / maxc;   // :125
  }
  if (want_traj) {
    for (int p = 0; p <= hp.n_paths; ++p) traj_off[p] = hoff[p];
    if (hst[0]) return PP_ERR_CAPACITY;
    if (hst[2]) return PP_ERR_DEGENERATE_RS;
    uint64_t tot = hoff[hp.n_paths];
    if (tot) {
      PP_HIP(ctx, hipMemcpyAsync(traj_x, d_tx, tot * 8, hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipMemcpyAsync(traj_y, d_ty, tot * 8, hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipMemcpyAsync(traj_yaw, d_tyaw, tot * 8, hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipMemcpyAsync(traj_dir, d_td, tot * 8, hipMemcpyDeviceToHost, s));
      PP_HIP(ctx, hipStreamSynchronize(s));
    }
  }
  return PP_OK;
}

extern "C" int32_t pp_rs_all_paths(pp_ctx* ctx, double sx, double sy, double syaw, double gx,
                                   double gy, double gyaw, double maxc, double step_size,
                                   int32_t* n_paths, pp_rs_word* words, uint64_t* traj_off,
                                   double* traj_x, double* traj_y, double* traj_yaw,
                                   int64_t* traj_dir, uint64_t traj_cap) {
  if (!ctx || !n_paths || !words) return PP_ERR_INVALID;
  if (traj_x && (!traj_off || !traj_y || !traj_yaw || !traj_dir)) return PP_ERR_INVALID;
  double q[6] = {sx, sy, syaw, gx, gy, gyaw};
  return pp_rs_single(ctx, q, maxc, step_size, -1, n_paths, words, traj_off, traj_x, traj_y, traj_yaw,
                      traj_dir, traj_cap, nullptr);
}

extern "C" int32_t pp_rs_optimal_path(pp_ctx* ctx, double sx, double sy, double syaw, double gx,
                                      double gy, double gyaw, double maxc, double step_size,
                                      pp_rs_word* word, double* traj_x, double* traj_y,
                                      double* traj_yaw, int64_t* traj_dir, uint64_t traj_cap,
                                      uint64_t* traj_len) {
  if (!ctx || !word) return PP_ERR_INVALID;
  double q[6] = {sx, sy, syaw, gx, gy, gyaw};
  int32_t np = 0;
  pp_rs_word words[PP_RS_MAX_PATHS];
  int32_t st = pp_rs_single(ctx, q, maxc, step_size, -1, &np, words, nullptr, nullptr, nullptr, nullptr,
                            nullptr, 0, nullptr);
  if (st != PP_OK) return st;
  if (np == 0) return PP_ERR_EMPTY_RS;  // paths[0] panics, :76
  int mini = 0;
  double min_l = words[0].total_length;
  for (int i = 0; i < np; ++i)
    if (words[i].total_length <= min_l) {
      min_l = words[i].total_length;
      mini = i;
    }
  *word = words[mini];
  if (traj_x) {
    uint64_t off[PP_RS_MAX_PATHS + 1];
    st = pp_rs_single(ctx, q, maxc, step_size, mini, &np, words, off, traj_x, traj_y, traj_yaw, traj_dir,
                      traj_cap, nullptr);
    if (traj_len) *traj_len = off[np];
    if (st != PP_OK) return st;
  }
  return PP_OK;
}

// calc_rs_path_cost, hybrid_astar.rs:450-489
extern "C" double pp_rs_path_cost(const pp_rs_word* w, const pp_config* conf) {
  if (!w || !conf) return 0.0;
  uint32_t word = 0;
  for (int i = 0; i < w->nseg && i < 5; ++i)
    word |= (uint32_t)(w->ctypes[i] == 'L' ? 1 : (w->ctypes[i] == 'S' ? 2 : 3)) << (4 * i);
  return rs_cost(w->lengths, word, w->nseg, cost_params(conf));
}

// PpShotOut records -> the entry point's separate arrays.
__global__ void k_shot_unpack(uint32_t n, const PpShotOut* __restrict__ o, uint8_t* __restrict__ found,
                              int32_t* __restrict__ path_idx, int32_t* __restrict__ n_paths,
                              double* __restrict__ rs_cost) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PpShotOut r = o[i];
  found[i] = (uint8_t)(r.flags & 31u);
  path_idx[i] = r.path_idx;
  n_paths[i] = r.n_paths;
  rs_cost[i] = r.path_idx >= 0 ? r.cost : 0.0;
}

extern "C" int32_t pp_analytic_batch_dev(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                         uint64_t n, const double* d_q6, uint8_t* d_found,
                                         int32_t* d_path_idx, int32_t* d_n_paths, pp_rs_word* d_word,
                                         double* d_rs_cost) {
  if (!ctx || !map || !conf) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (n == 0) return PP_OK;
  if (n > 0x7fffffffu) return PP_ERR_UNSUPPORTED;
  if (!d_q6 || !d_found || !d_path_idx || !d_n_paths || !d_word || !d_rs_cost) return PP_ERR_INVALID;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ac = al256(n * PP_RS_MAX_PATHS * sizeof(PpRsCand)), ap = al256(n * sizeof(PpRsPair));
  st = pp_scratch_reserve(ctx, ac + ap + al256(n * sizeof(PpShotOut)));
  if (st != PP_OK) return st;
  PpRsCand* d_c = (PpRsCand*)ctx->scratch;
  PpRsPair* d_p = (PpRsPair*)((char*)d_c + ac);
  PpShotOut* d_o = (PpShotOut*)((char*)d_p + ap);
  const double maxc = std::tan(conf->max_steer) / conf->car.wheel_base;  // hybrid_astar.rs:388
  pp_rs_words_launch(ctx, n, d_q6, maxc, d_c, d_p);
  pp_rs_shot_launch(ctx, map, conf, (uint32_t)n, d_q6, maxc, d_c, d_p, d_o, d_word);
  hipLaunchKernelGGL(k_shot_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (uint32_t)n, d_o, d_found, d_path_idx, d_n_paths, d_rs_cost);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

extern "C" int32_t pp_analytic_batch(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                     uint64_t n, const double* sx, const double* sy,
                                     const double* syaw, const double* gx, const double* gy,
                                     const double* gyaw, uint8_t* found, int32_t* path_idx,
                                     int32_t* n_paths, pp_rs_word* word, double* rs_cost_out) {
  if (!ctx || !map || !conf) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (n == 0) return PP_OK;
  if (n > 0x7fffffffu) return PP_ERR_UNSUPPORTED;
  if (!sx || !sy || !syaw || !gx || !gy || !gyaw || !found || !path_idx || !n_paths || !word || !rs_cost_out)
    return PP_ERR_INVALID;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  size_t ac = al256(n * PP_RS_MAX_PATHS * sizeof(PpRsCand)), ap = al256(n * sizeof(PpRsPair));
  size_t ao = al256(n * sizeof(PpShotOut)), aw = al256(n * sizeof(pp_rs_word));
  const double* hq[6] = {sx, sy, syaw, gx, gy, gyaw};
  hipStream_t s = ctx->stream;
  // the six arrays go back to back, as the launch helpers expect
  std::vector<double> packed(6 * n);
  for (int k = 0; k < 6; ++k) std::memcpy(packed.data() + k * n, hq[k], n * 8);
  st = pp_scratch_reserve(ctx, al256(6 * n * 8) + ac + ap + ao + aw);
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  double* d_q = (double*)b;
  PpRsCand* d_c = (PpRsCand*)(b + al256(6 * n * 8));
  PpRsPair* d_p = (PpRsPair*)((char*)d_c + ac);
  PpShotOut* d_o = (PpShotOut*)((char*)d_p + ap);
  pp_rs_word* d_w = (pp_rs_word*)((char*)d_o + ao);
  PP_HIP(ctx, hipMemcpyAsync(d_q, packed.data(), 6 * n * 8, hipMemcpyHostToDevice, s));
  double maxc = std::tan(conf->max_steer) / conf->car.wheel_base;  // hybrid_astar.rs:388
  pp_rs_words_launch(ctx, n, d_q, maxc, d_c, d_p);
  pp_rs_shot_launch(ctx, map, conf, (uint32_t)n, d_q, maxc, d_c, d_p, d_o, d_w);
  PP_HIP(ctx, hipGetLastError());
  std::vector<PpShotOut> ho(n);
  PP_HIP(ctx, hipMemcpyAsync(ho.data(), d_o, n * sizeof(PpShotOut), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(word, d_w, n * sizeof(pp_rs_word), hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  int32_t ret = PP_OK;
  for (uint64_t i = 0; i < n; ++i) {
    const PpShotOut& o = ho[i];
    found[i] = (uint8_t)((o.flags & 1u) | (o.flags & 2u) | (o.flags & 4u));
    path_idx[i] = o.path_idx;
    n_paths[i] = o.n_paths;
    rs_cost_out[i] = o.path_idx >= 0 ? o.cost : 0.0;
    if (o.flags & 8u) ret = PP_ERR_NAN_COST;
    if (o.flags & 16u) ret = PP_ERR_NONFINITE;
  }
  return ret;
}

// ---------------------------------------------------------------------------------------------
// pp_analytic_batch_dev for callers that must not allocate (under stream capture, in a serving
// loop, in a harness that forbids hipMalloc).  Its intermediate records -- the candidate words of
// every pair, the per-pair summaries and the shot results the unpack kernel reads -- live in one
// array of device memory that comes with the library instead of in the context's grown-on-demand
// buffer.  Every GPU the library runs on holds its own copy of the array for the life of the
// process; the host finds a device's copy once, by symbol, and hands that address to the kernels.

namespace {

constexpr uint64_t kStaticMaxPairs = 4096;
constexpr size_t kStaticCandBytes = (kStaticMaxPairs * PP_RS_MAX_PATHS * sizeof(PpRsCand) + 255) & ~(size_t)255;
constexpr size_t kStaticPairBytes = (kStaticMaxPairs * sizeof(PpRsPair) + 255) & ~(size_t)255;
constexpr size_t kStaticOutBytes = (kStaticMaxPairs * sizeof(PpShotOut) + 255) & ~(size_t)255;
constexpr size_t kStaticBytes = kStaticCandBytes + kStaticPairBytes + kStaticOutBytes;  // 10.3 MiB

constexpr int kStaticMaxDevices = 64;
void* g_static_ptr[kStaticMaxDevices];  // device address of `scratch` per HIP ordinal, 0 = not looked up
std::mutex g_static_mu;

}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticBytes];

// Device address of this device's copy of `scratch` (the device must be current).
static int32_t static_scratch(pp_ctx* ctx, unsigned char** out) {
  if (ctx->device < 0 || ctx->device >= kStaticMaxDevices) return PP_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[ctx->device]) {
    void* p = nullptr;
    PP_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return PP_ERR_HIP;
    g_static_ptr[ctx->device] = p;
  }
  *out = (unsigned char*)g_static_ptr[ctx->device];
  return PP_OK;
}

extern "C" uint64_t pp_analytic_static_max(void) { return kStaticMaxPairs; }

extern "C" int32_t pp_analytic_batch_dev_static(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                                uint64_t n, const double* d_q6, uint8_t* d_found,
                                                int32_t* d_path_idx, int32_t* d_n_paths,
                                                pp_rs_word* d_word, double* d_rs_cost) {
  if (!ctx || !map || !conf) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (n == 0) return PP_OK;
  if (n > kStaticMaxPairs) return PP_ERR_CAPACITY;  // the array holds the records of 4096 pairs
  if (!d_q6 || !d_found || !d_path_idx || !d_n_paths || !d_word || !d_rs_cost) return PP_ERR_INVALID;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  unsigned char* base = nullptr;
  st = static_scratch(ctx, &base);
  if (st != PP_OK) return st;
  // Nothing in the array has to start at zero: the word kernel writes a pair's summary and the
  // candidates its `accepted` bits name before the shot kernel reads them, and the shot kernel
  // writes every result record the unpack kernel reads, all in stream order.
  PpRsCand* d_c = (PpRsCand*)base;
  PpRsPair* d_p = (PpRsPair*)(base + kStaticCandBytes);
  PpShotOut* d_o = (PpShotOut*)(base + kStaticCandBytes + kStaticPairBytes);
  const double maxc = std::tan(conf->max_steer) / conf->car.wheel_base;  // hybrid_astar.rs:388
  pp_rs_words_launch(ctx, n, d_q6, maxc, d_c, d_p);
  pp_rs_shot_launch(ctx, map, conf, (uint32_t)n, d_q6, maxc, d_c, d_p, d_o, d_word);
  hipLaunchKernelGGL(k_shot_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (uint32_t)n, d_o, d_found, d_path_idx, d_n_paths, d_rs_cost);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}
