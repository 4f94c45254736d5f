// pp_expand.hip -- motion-primitive rollouts with footprint checks
// (calc_next_node + is_index_ok, hybrid_astar.rs:252-343) for a batch of parent nodes.
//
// One lane per (parent, primitive).  The yaw recurrence uses tan(u) from a host-computed table
// (calc_motion_set yields only 2*n_steer-1 distinct steering angles), so every yaw of a rollout
// is bit-identical to the reference.  The direction (cos yaw, sin yaw) is evaluated once, at the
// parent's pose, and turned by the primitive's yaw step from pose to pose (the step's cos and sin
// come with the motion table): after k steps it is off the libm value of the reference's rounded
// yaw by at most k (ulp(yaw) + 3 ulp(1)), about 1e-14 over a primitive at |yaw| < 16, which moves
// a pose by less than 1e-14 m -- five orders below the guard bands of every discrete decision.
#include "pp_expand_device.h"

// The first pose of every primitive is the parent's own (hybrid_astar.rs:266-268 start the lists
// with it and is_index_ok samples index 0): for a large batch it is tested once per parent here
// instead of once per primitive.
__global__ void __launch_bounds__(256)
k_parent_pose(uint64_t n, const double* __restrict__ px, const double* __restrict__ py,
              const double* __restrict__ pyaw, PpFootprint f, PpLattice lat, PpBuckets bk, int use_lattice,
              uint8_t* __restrict__ flag) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double s, c;
  sincos(pyaw[p], &s, &c);
  const unsigned r = use_lattice ? pp_pose_vs_lattice(px[p], py[p], c, s, f, lat)
                                 : pp_pose_vs_buckets(px[p], py[p], c, s, f, bk);
  flag[p] = (uint8_t)r;
}

// (four waves per SIMD: the table lookup of pp_expand_item would otherwise cost one register too many)
__global__ void __launch_bounds__(256, 4)
k_expand(uint64_t n_items, uint32_t n_prim, const double* __restrict__ px,
         const double* __restrict__ py, const double* __restrict__ pyaw,
         const int64_t* __restrict__ pdir, const double* __restrict__ psteer,
         const double* __restrict__ pcost, const double* __restrict__ motion, PpExpandParams prm,
         PpFootprint f, PpLattice lat, PpBuckets bk, int use_lattice, const uint8_t* __restrict__ pflag,
         PpExpandDev out) {
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += stride) {
    pp_expand_item(t, n_prim, px, py, pyaw, pdir, psteer, pcost, motion, prm, f, lat, bk, use_lattice, pflag, out);
  }
}

// The (nlist+1)-point lists of individual primitives (hybrid_astar.rs:266-274), for the nodes
// extract_path walks.  One lane per primitive; out[(j * (nlist+1) + k)].
__global__ void k_rollout_points(uint32_t n, uint32_t nlist, const double* __restrict__ in,
                                 double move_step, double wheel_base, double* __restrict__ ox,
                                 double* __restrict__ oy, double* __restrict__ oyaw) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double x = in[j * 7 + 0], y = in[j * 7 + 1], yaw = in[j * 7 + 2], tanu = in[j * 7 + 3],
         d = in[j * 7 + 4];
  const double cd = in[j * 7 + 5], sd = in[j * 7 + 6];
  double dm = d * move_step;
  double dyaw = dm / wheel_base * tanu;
  size_t o = (size_t)j * (nlist + 1);
  ox[o] = x;
  oy[o] = y;
  oyaw[o] = yaw;
  // the same arithmetic as k_expand, so a node's last point is the pose the search stored for it
  double s, c;
  sincos(yaw, &s, &c);
  for (uint32_t k = 0; k < nlist; ++k) {
    double ax = dm * c, ay = dm * s;
    x = x + ax;
    y = y + ay;
    yaw = yaw + dyaw;
    pp_turn(c, s, cd, sd);
    ox[o + k + 1] = x;
    oy[o + k + 1] = y;
    oyaw[o + k + 1] = yaw;
  }
}

// ---------------------------------------------------------------------------- host side

// calc_motion_set, hybrid_astar.rs:495-514
extern "C" int32_t pp_motion_set(const pp_config* conf, double* steer, double* direc, uint64_t cap,
                                 uint64_t* n_out) {
  if (!conf || !n_out || conf->n_steer == 0) return PP_ERR_INVALID;
  uint64_t half = 2 * (conf->n_steer - 1) + 1;
  *n_out = 2 * half;
  if (cap < 2 * half) return PP_ERR_CAPACITY;
  if (!steer || !direc) return PP_ERR_INVALID;
  uint64_t ns = conf->n_steer - 1;
  for (uint64_t i = 1; i < conf->n_steer; ++i) {
    double v = (double)i * conf->max_steer / (double)conf->n_steer;
    steer[i - 1] = v;
    steer[ns + i] = -v;
  }
  steer[ns] = 0.0;
  for (uint64_t i = 0; i < half; ++i) {
    steer[half + i] = steer[i];
    direc[i] = 1.0;
    direc[half + i] = -1.0;
  }
  return PP_OK;
}

int32_t pp_expand_prepare(pp_ctx* ctx, const pp_map* map, const pp_config* conf, PpExpandPlan* plan) {
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  uint64_t np = 0;
  pp_motion_set(conf, nullptr, nullptr, 0, &np);
  plan->n_prim = (uint32_t)np;
  // steer | direction | tan(steer) | cos, sin of the yaw step d * move_step / wheel_base * tan(steer)
  plan->motion.resize(5 * np);
  pp_motion_set(conf, plan->motion.data(), plan->motion.data() + np, np, &np);
  for (uint64_t i = 0; i < np; ++i) {
    const double tanu = std::tan(plan->motion[i]);
    const double dyaw = plan->motion[np + i] * conf->move_step / conf->car.wheel_base * tanu;
    plan->motion[2 * np + i] = tanu;
    plan->motion[3 * np + i] = std::cos(dyaw);
    plan->motion[4 * np + i] = std::sin(dyaw);
  }
  pp_param P;
  pp_map_param(map, conf, &P);
  plan->param = P;
  PpExpandParams& q = plan->prm;
  q.step = conf->xy_reso * 2.0;
  double nl = std::ceil(q.step / conf->move_step);
  if (!(nl >= 0.0) || nl > 1.0e6) return PP_ERR_INVALID;
  q.nlist = (uint32_t)pp_sat_usize(nl);
  q.check_step = (uint32_t)std::min<uint64_t>(conf->collision_check_step, 0x7fffffffu);
  q.move_step = conf->move_step;
  q.wheel_base = conf->car.wheel_base;
  q.xyreso = P.xyreso;
  q.yawreso = P.yawreso;
  q.minx = P.minx;
  q.miny = P.miny;
  q.maxx = P.maxx;
  q.maxy = P.maxy;
  q.backward_cost = conf->backward_cost;
  q.gear_cost = conf->gear_cost;
  q.steer_angle_cost = conf->steer_angle_cost;
  q.steer_change_cost = conf->steer_change_cost;
  plan->fp = pp_make_footprint(conf);
  (void)ctx;
  return PP_OK;
}

// One lane per record.
__global__ void k_ftab_update(uint32_t n, const PpFtabUpd* __restrict__ upd, unsigned long long* __restrict__ ftab,
                              uint64_t slots) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) pp_ftab_update_one(upd[t], ftab, slots);
}

void pp_ftab_update_launch(pp_ctx* ctx, uint32_t n, const PpFtabUpd* upd, unsigned long long* ftab, uint64_t slots) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_ftab_update, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, upd, ftab, slots);
}

void pp_expand_launch(pp_ctx* ctx, const pp_map* map, const PpExpandPlan& plan, uint64_t n,
                      const double* d_px, const double* d_py, const double* d_pyaw,
                      const int64_t* d_pdir, const double* d_psteer, const double* d_pcost,
                      const double* d_motion, PpExpandDev out, uint8_t* d_parent_flag) {
  uint64_t items = n * plan.n_prim;
  const unsigned block = 128;  // with 128 blocks per CU: 714 us on 20 M primitives, 256 x 32 per CU: 729 us
  const unsigned grid = (unsigned)std::max<uint64_t>(
      1, std::min<uint64_t>((items + block - 1) / block, (uint64_t)ctx->num_cus * 128));
  if (d_parent_flag)
    hipLaunchKernelGGL(k_parent_pose, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, ctx->stream, n,
                       d_px, d_py, d_pyaw, plan.fp, map->lat, map->bk, map->lattice ? 1 : 0, d_parent_flag);
  hipLaunchKernelGGL(k_expand, dim3(grid), dim3(block), 0, ctx->stream, items, plan.n_prim, d_px,
                     d_py, d_pyaw, d_pdir, d_psteer, d_pcost, d_motion, plan.prm, plan.fp, map->lat,
                     map->bk, map->lattice ? 1 : 0, (const uint8_t*)d_parent_flag, out);
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int32_t pp_expand_batch(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                   uint64_t n, const double* px, const double* py,
                                   const double* pyaw, const int64_t* pdir, const double* psteer,
                                   const double* pcost, pp_expand_out* out) {
  if (!ctx || !map || !conf || !out) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (n == 0) return PP_OK;
  if (!px || !py || !pyaw || !pdir || !psteer || !pcost) return PP_ERR_INVALID;
  PpExpandPlan plan;
  int32_t st = pp_expand_prepare(ctx, map, conf, &plan);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t items = n * plan.n_prim;
  size_t in_b = al256(n * 8), mot_b = al256(plan.motion.size() * 8);
  size_t o8 = al256(items * 8), o1 = al256(items);
  st = pp_scratch_reserve(ctx, 6 * in_b + mot_b + 7 * o8 + o1 + al256(n));
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  uint8_t* d_pf = n >= kParentPassMin ? (uint8_t*)(b + 6 * in_b + mot_b + 7 * o8 + o1) : nullptr;
  double* d_px = (double*)b; b += in_b;
  double* d_py = (double*)b; b += in_b;
  double* d_pyaw = (double*)b; b += in_b;
  int64_t* d_pdir = (int64_t*)b; b += in_b;
  double* d_ps = (double*)b; b += in_b;
  double* d_pc = (double*)b; b += in_b;
  double* d_mot = (double*)b; b += mot_b;
  PpExpandDev o;
  o.xind = (uint64_t*)b; b += o8;
  o.yind = (uint64_t*)b; b += o8;
  o.yawind = (int64_t*)b; b += o8;
  o.cost = (double*)b; b += o8;
  o.end_x = (double*)b; b += o8;
  o.end_y = (double*)b; b += o8;
  o.end_yaw = (double*)b; b += o8;
  o.valid = (uint8_t*)b;
  hipStream_t s = ctx->stream;
  PP_HIP(ctx, hipMemcpyAsync(d_px, px, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_py, py, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_pyaw, pyaw, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_pdir, pdir, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_ps, psteer, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_pc, pcost, n * 8, hipMemcpyHostToDevice, s));
  PP_HIP(ctx, hipMemcpyAsync(d_mot, plan.motion.data(), plan.motion.size() * 8, hipMemcpyHostToDevice, s));
  pp_expand_launch(ctx, map, plan, n, d_px, d_py, d_pyaw, d_pdir, d_ps, d_pc, d_mot, o, d_pf);
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipMemcpyAsync(out->valid, o.valid, items, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->xind, o.xind, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->yind, o.yind, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->yawind, o.yawind, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->cost, o.cost, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->end_x, o.end_x, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->end_y, o.end_y, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(out->end_yaw, o.end_yaw, items * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  for (uint64_t t = 0; t < items; ++t)
    if (out->valid[t] & 4) return PP_ERR_NONFINITE;
  return PP_OK;
}

extern "C" int32_t pp_expand_batch_dev(pp_ctx* ctx, const pp_map* map, const pp_config* conf,
                                       uint64_t n, const double* d_px, const double* d_py,
                                       const double* d_pyaw, const int64_t* d_pdir,
                                       const double* d_psteer, const double* d_pcost,
                                       const pp_expand_out* d_out) {
  if (!ctx || !map || !conf || !d_out) return PP_ERR_INVALID;
  if (map->ox.empty()) return PP_ERR_EMPTY_MAP;
  if (!map->finite) return PP_ERR_NONFINITE;
  if (n == 0) return PP_OK;
  if (!d_px || !d_py || !d_pyaw || !d_pdir || !d_psteer || !d_pcost) return PP_ERR_INVALID;
  if (!d_out->valid || !d_out->xind || !d_out->yind || !d_out->yawind || !d_out->cost || !d_out->end_x ||
      !d_out->end_y || !d_out->end_yaw)
    return PP_ERR_INVALID;
  PpExpandPlan plan;
  int32_t st = pp_expand_prepare(ctx, map, conf, &plan);
  if (st != PP_OK) return st;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  // the motion table is the only thing staged by the library
  st = pp_scratch_reserve(ctx, al256(plan.motion.size() * 8) + al256(n));
  if (st != PP_OK) return st;
  double* d_mot = (double*)ctx->scratch;
  uint8_t* d_pf = n >= kParentPassMin ? (uint8_t*)ctx->scratch + al256(plan.motion.size() * 8) : nullptr;
  PP_HIP(ctx, hipMemcpyAsync(d_mot, plan.motion.data(), plan.motion.size() * 8, hipMemcpyHostToDevice,
                             ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // plan.motion is a local
  PpExpandDev o{};
  o.valid = d_out->valid;
  o.xind = d_out->xind;
  o.yind = d_out->yind;
  o.yawind = d_out->yawind;
  o.cost = d_out->cost;
  o.end_x = d_out->end_x;
  o.end_y = d_out->end_y;
  o.end_yaw = d_out->end_yaw;
  pp_expand_launch(ctx, map, plan, n, d_px, d_py, d_pyaw, d_pdir, d_psteer, d_pcost, d_mot, o, d_pf);
  PP_HIP(ctx, hipGetLastError());
  return PP_OK;
}

int32_t pp_rollout_points_batch(pp_ctx* ctx, const pp_config* conf, uint32_t n,
                                const double* in5 /* x,y,yaw,u,d per item */, double* x, double* y,
                                double* yaw, uint32_t* nlist_out) {
  double step = conf->xy_reso * 2.0;
  uint32_t nlist = (uint32_t)pp_sat_usize(std::ceil(step / conf->move_step));
  if (nlist_out) *nlist_out = nlist;
  if (n == 0) return PP_OK;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<double> in((size_t)n * 7);  // x, y, yaw, tan(u), d, cos and sin of the yaw step
  for (uint32_t j = 0; j < n; ++j) {
    const double* a = in5 + (size_t)j * 5;
    const double tanu = std::tan(a[3]);
    const double dyaw = a[4] * conf->move_step / conf->car.wheel_base * tanu;
    const double v[7] = {a[0], a[1], a[2], tanu, a[4], std::cos(dyaw), std::sin(dyaw)};
    std::copy(v, v + 7, in.begin() + (size_t)j * 7);
  }
  size_t pts = (size_t)n * (nlist + 1);
  size_t in_b = al256(in.size() * 8), o_b = al256(pts * 8);
  int32_t st = pp_scratch_reserve(ctx, in_b + 3 * o_b);
  if (st != PP_OK) return st;
  char* b = (char*)ctx->scratch;
  double* d_in = (double*)b; b += in_b;
  double* d_x = (double*)b; b += o_b;
  double* d_y = (double*)b; b += o_b;
  double* d_yaw = (double*)b;
  hipStream_t s = ctx->stream;
  PP_HIP(ctx, hipMemcpyAsync(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_rollout_points, dim3((n + 63) / 64), dim3(64), 0, s, n, nlist, d_in,
                     conf->move_step, conf->car.wheel_base, d_x, d_y, d_yaw);
  PP_HIP(ctx, hipGetLastError());
  PP_HIP(ctx, hipMemcpyAsync(x, d_x, pts * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(y, d_y, pts * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipMemcpyAsync(yaw, d_yaw, pts * 8, hipMemcpyDeviceToHost, s));
  PP_HIP(ctx, hipStreamSynchronize(s));
  return PP_OK;
}

extern "C" int32_t pp_rollout_points(pp_ctx* ctx, const pp_config* conf, double x0, double y0,
                                     double yaw0, double u, double d, double* x, double* y,
                                     double* yaw, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !conf || !n_out) return PP_ERR_INVALID;
  int32_t st = pp_check_config(conf);
  if (st != PP_OK) return st;
  double step = conf->xy_reso * 2.0;
  uint64_t nlist = pp_sat_usize(std::ceil(step / conf->move_step));
  *n_out = nlist + 1;
  if (cap < nlist + 1) return PP_ERR_CAPACITY;
  if (!x || !y || !yaw) return PP_ERR_INVALID;
  double in5[5] = {x0, y0, yaw0, u, d};
  return pp_rollout_points_batch(ctx, conf, 1, in5, x, y, yaw, nullptr);
}
