// pp_ctx.hip -- context and obstacle-map lifecycle.
#include <algorithm>
#include <limits>

#include "pp_internal.h"

extern "C" const char* pp_version(void) { return "path_planner_amd 0.1 (gfx950)"; }

extern "C" const char* pp_strerror(int32_t st) {
  switch (st) {
    case PP_OK: return "ok";
    case PP_NO_PATH: return "no path (Option::None)";
    case PP_ERR_INVALID: return "invalid argument";
    case PP_ERR_HIP: return "HIP runtime error";
    case PP_ERR_NAN_COST: return "NaN cost (NotNan::new(..).unwrap() would panic)";
    case PP_ERR_DEGENERATE_RS: return "Reeds-Shepp word shorter than 0.01 (assert would fire)";
    case PP_ERR_EMPTY_RS: return "no Reeds-Shepp path (paths[0] would panic)";
    case PP_ERR_BOUNDS: return "index out of bounds (hmap / closed-set lookup would panic)";
    case PP_ERR_CAPACITY: return "caller buffer too small";
    case PP_ERR_NONFINITE: return "non-finite coordinate (kdtree would return Err)";
    case PP_ERR_EMPTY_MAP: return "empty obstacle list";
    case PP_ERR_UNSUPPORTED: return "unsupported";
    case PP_ERR_LIMIT: return "expansion limit reached (the reference would keep searching)";
  }
  return "unknown status";
}

extern "C" int32_t pp_ctx_create(int32_t device, void* stream, pp_ctx** out) {
  if (!out) return PP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || device < 0 || device >= ndev) return PP_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return PP_ERR_HIP;
  pp_ctx* ctx = new pp_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return PP_ERR_HIP;
    }
    ctx->own_stream = true;
  }
  *out = ctx;
  return PP_OK;
}

extern "C" void pp_ctx_destroy(pp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (pp_ctx* h : ctx->helpers) pp_ctx_destroy(h);
  ctx->helpers.clear();
  (void)hipSetDevice(ctx->device);
  if (ctx->slabs && ctx->slabs_orphan) ctx->slabs_orphan(ctx);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->wave_resume && ctx->wave_resume_free) ctx->wave_resume_free(ctx->wave_resume);
  if (ctx->wave_buf) (void)hipFree(ctx->wave_buf);
  if (ctx->wave_counts) (void)hipHostFree(ctx->wave_counts);
  if (ctx->wave_event) (void)hipEventDestroy(ctx->wave_event);
  if (ctx->ftab_buf) (void)hipFree(ctx->ftab_buf);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->plan_buf) (void)hipFree(ctx->plan_buf);
  if (ctx->hmap_buf) (void)hipFree(ctx->hmap_buf);
  if (ctx->plan_state && ctx->plan_state_free) ctx->plan_state_free(ctx->plan_state);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" void* pp_ctx_stream(pp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int32_t pp_ctx_synchronize(pp_ctx* ctx) {
  if (!ctx) return PP_ERR_INVALID;
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_relax_band(pp_ctx* ctx, double band) {
  if (!ctx || !(band >= 0.0)) return PP_ERR_INVALID;
  ctx->wave_band = band;
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_max_expansions(pp_ctx* ctx, uint64_t max_expansions) {
  if (!ctx) return PP_ERR_INVALID;
  ctx->max_expansions = max_expansions;
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_heuristic_limit(pp_ctx* ctx, double limit) {
  if (!ctx || limit != limit) return PP_ERR_INVALID;
  ctx->heuristic_limit = limit >= 0.0 ? limit : -1.0;
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_heuristic_workspace(pp_ctx* ctx, uint64_t bytes) {
  if (!ctx) return PP_ERR_INVALID;
  ctx->heuristic_pass_blocks = std::max<uint64_t>(1, bytes / 8192);
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_table_budget(pp_ctx* ctx, uint64_t bytes) {
  if (!ctx) return PP_ERR_INVALID;
  ctx->ftab_budget = (size_t)bytes;
  if (ctx->ftab_buf && ctx->ftab_bytes > ctx->ftab_budget) {  // give back what no longer fits
    (void)hipSetDevice(ctx->device);
    PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PP_HIP(ctx, hipFree(ctx->ftab_buf));
    ctx->ftab_buf = nullptr;
    ctx->ftab_bytes = 0;
  }
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_batch_parts(pp_ctx* ctx, uint32_t parts) {
  if (!ctx || parts > 64) return PP_ERR_INVALID;
  ctx->batch_parts = parts;
  return PP_OK;
}

extern "C" int32_t pp_ctx_set_host_threads(pp_ctx* ctx, uint32_t n) {
  if (!ctx) return PP_ERR_INVALID;
  ctx->host_threads = std::min<uint32_t>(n, 255);
  return PP_OK;
}

extern "C" uint64_t pp_ctx_device_bytes(const pp_ctx* ctx) {
  if (!ctx) return 0;
  uint64_t b = (uint64_t)ctx->scratch_bytes + ctx->wave_bytes + ctx->ftab_bytes + ctx->plan_bytes + ctx->hmap_bytes;
  for (const pp_ctx* h : ctx->helpers) b += pp_ctx_device_bytes(h);
  return b;
}

extern "C" int32_t pp_last_hip_error(pp_ctx* ctx) { return ctx ? ctx->last_hip_error : 0; }

int32_t pp_scratch_reserve(pp_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return PP_OK;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->scratch) PP_HIP(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  size_t want = std::max(bytes, (size_t)1 << 20);
  PP_HIP(ctx, hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return PP_OK;
}

int32_t pp_pinned_reserve(pp_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_bytes) return PP_OK;
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->pinned) PP_HIP(ctx, hipHostFree(ctx->pinned));
  ctx->pinned = nullptr;
  ctx->pinned_bytes = 0;
  size_t want = std::max(bytes, (size_t)1 << 16);
  PP_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
  ctx->pinned_bytes = want;
  return PP_OK;
}

int32_t pp_check_config(const pp_config* c) {
  if (!c) return PP_ERR_INVALID;
  if (!(c->xy_reso > 0.0) || !(c->yaw_reso > 0.0) || !(c->move_step > 0.0)) return PP_ERR_INVALID;
  if (c->n_steer == 0 || c->n_steer > 4096 || c->collision_check_step == 0) return PP_ERR_INVALID;
  if (!(c->car.wheel_base != 0.0)) return PP_ERR_INVALID;
  return PP_OK;
}

PpFootprint pp_make_footprint(const pp_config* conf) {
  PpFootprint f;
  f.cp2center = (conf->car.center2front - conf->car.center2back) / 2.0;
  f.bound_len = (conf->car.center2front + conf->car.center2back) / 2.0 + conf->extend_bound;
  f.bound_wid = conf->car.width / 2.0 + conf->extend_bound;
  f.bound_radius = std::hypot(f.bound_len, f.bound_wid);
  // largest t with sqrt(t) <= R: start at R*R and walk by ulps (host sqrt is correctly rounded)
  double t = f.bound_radius * f.bound_radius;
  const double inf = std::numeric_limits<double>::infinity();
  if (std::isfinite(t)) {
    while (std::sqrt(t) > f.bound_radius) t = std::nextafter(t, -inf);
    while (std::sqrt(std::nextafter(t, inf)) <= f.bound_radius) t = std::nextafter(t, inf);
  }
  f.r2_max = t;
  f.guard = 1.0e-11;
  const double inflate = 2.5e-4;
  f.scan_lo = std::nextafter((float)(f.bound_len + inflate), 1e30f);
  f.scan_wo = std::nextafter((float)(f.bound_wid + inflate), 1e30f);
  f.scan_li = std::nextafter((float)(f.bound_len - inflate), -1e30f);
  f.scan_wi = std::nextafter((float)(f.bound_wid - inflate), -1e30f);
  f.scan_ok = (std::isfinite(f.bound_radius) && f.bound_radius + 2.0 * inflate <= 6.0 &&
               f.bound_len >= 0.0 && f.bound_wid >= 0.0) ? 1 : 0;
  f.scan_small = (f.scan_ok && f.bound_radius + 2.0 * inflate <= 3.0) ? 1 : 0;
#ifdef PP_LAB_NO_SCAN_SMALL
  f.scan_small = 0;
#endif
  return f;
}

// ------------------------------------------------------------------------ map build

static void map_free_device(pp_map* m) {
  if (m->d_tiles) (void)hipFree(m->d_tiles);
  if (m->d_bstart) (void)hipFree(m->d_bstart);
  if (m->d_bpx) (void)hipFree(m->d_bpx);
  if (m->d_bpy) (void)hipFree(m->d_bpy);
  if (m->d_ox) (void)hipFree(m->d_ox);
  if (m->d_oy) (void)hipFree(m->d_oy);
  m->d_tiles = nullptr;
  m->d_bstart = nullptr;
  m->d_bpx = m->d_bpy = m->d_ox = m->d_oy = nullptr;
}

// Build the tiled bitmap from lattice cells given by callback-free arrays of integer coordinates.
static int32_t build_lattice(pp_ctx* ctx, pp_map* m, const std::vector<int64_t>& ix,
                             const std::vector<int64_t>& iy, int64_t lo_x, int64_t lo_y,
                             int64_t hi_x, int64_t hi_y) {
  // one ring of empty tiles around the occupied extent
  int64_t x0 = (lo_x & ~(int64_t)31) - 32, y0 = (lo_y & ~(int64_t)31) - 32;
  int64_t nx = ((hi_x - x0) / 32 + 2) * 32, ny = ((hi_y - y0) / 32 + 2) * 32;
  if (nx > (1 << 24) || ny > (1 << 24)) return PP_ERR_UNSUPPORTED;
  const int32_t ntx = (int32_t)(nx / 16), nty = (int32_t)(ny / 16);
  const size_t nwords = (size_t)ntx * (size_t)nty * 32;
  if (nwords > ((size_t)1 << 33)) return PP_ERR_UNSUPPORTED;
  std::vector<uint32_t> words(nwords, 0u);
  for (size_t i = 0; i < ix.size(); ++i) {
    const int64_t gx = ix[i] - x0, gy = iy[i] - y0;
    // the (up to) four overlapping tiles that hold the cell
    for (int64_t tx = (gx >> 4) - 1; tx <= (gx >> 4); ++tx)
      for (int64_t ty = (gy >> 4) - 1; ty <= (gy >> 4); ++ty) {
        if (tx < 0 || ty < 0) continue;
        // column c of a tile row sits at bit c + kLatticeShift; the filter never looks past column 27
        const int64_t col = gx - 16 * tx;
        if (col + kLatticeShift < 32)
          words[((size_t)tx * (size_t)nty + (size_t)ty) * 32 + (size_t)(gy - 16 * ty)] |= 1u << (col + kLatticeShift);
      }
  }
  PP_HIP(ctx, hipMalloc((void**)&m->d_tiles, nwords * sizeof(uint32_t)));
  PP_HIP(ctx, hipMemcpy(m->d_tiles, words.data(), nwords * sizeof(uint32_t), hipMemcpyHostToDevice));
  m->lat.tiles = m->d_tiles;
  m->lat.ntx = ntx;
  m->lat.nty = nty;
  m->lat.nx = (int32_t)nx;
  m->lat.ny = (int32_t)ny;
  m->lat.x0 = x0;
  m->lat.y0 = y0;
  m->lattice = true;
  return PP_OK;
}

static int32_t build_buckets(pp_ctx* ctx, pp_map* m) {
  size_t n = m->ox.size();
  int64_t bx0 = (int64_t)std::floor(m->minx), by0 = (int64_t)std::floor(m->miny);
  int64_t nbx = (int64_t)std::floor(m->maxx) - bx0 + 1, nby = (int64_t)std::floor(m->maxy) - by0 + 1;
  if (nbx <= 0 || nby <= 0 || nbx > (1 << 20) || nby > (1 << 20) ||
      (size_t)nbx * (size_t)nby > ((size_t)1 << 31))
    return PP_ERR_UNSUPPORTED;
  size_t nb = (size_t)nbx * (size_t)nby;
  std::vector<uint32_t> start(nb + 1, 0u);
  auto bucket = [&](size_t i) {
    int64_t bx = (int64_t)std::floor(m->ox[i]) - bx0, by = (int64_t)std::floor(m->oy[i]) - by0;
    return (size_t)by * (size_t)nbx + (size_t)bx;
  };
  for (size_t i = 0; i < n; ++i) start[bucket(i) + 1]++;
  for (size_t b = 0; b < nb; ++b) start[b + 1] += start[b];
  std::vector<uint32_t> fill(start.begin(), start.end() - 1);
  std::vector<double> px(n), py(n);
  for (size_t i = 0; i < n; ++i) {
    uint32_t k = fill[bucket(i)]++;
    px[k] = m->ox[i];
    py[k] = m->oy[i];
  }
  PP_HIP(ctx, hipMalloc((void**)&m->d_bstart, (nb + 1) * sizeof(uint32_t)));
  PP_HIP(ctx, hipMalloc((void**)&m->d_bpx, std::max<size_t>(n, 1) * sizeof(double)));
  PP_HIP(ctx, hipMalloc((void**)&m->d_bpy, std::max<size_t>(n, 1) * sizeof(double)));
  PP_HIP(ctx, hipMemcpy(m->d_bstart, start.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
  PP_HIP(ctx, hipMemcpy(m->d_bpx, px.data(), n * sizeof(double), hipMemcpyHostToDevice));
  PP_HIP(ctx, hipMemcpy(m->d_bpy, py.data(), n * sizeof(double), hipMemcpyHostToDevice));
  m->bk.start = m->d_bstart;
  m->bk.px = m->d_bpx;
  m->bk.py = m->d_bpy;
  m->bk.bx0 = bx0;
  m->bk.by0 = by0;
  m->bk.nbx = (int32_t)nbx;
  m->bk.nby = (int32_t)nby;
  return PP_OK;
}

static int32_t finish_map(pp_ctx* ctx, pp_map* m, pp_map** out) {
  size_t n = m->ox.size();
  const double inf = std::numeric_limits<double>::infinity();
  m->minx = m->miny = inf;
  m->maxx = m->maxy = -inf;
  bool lattice = n > 0;
  for (size_t i = 0; i < n; ++i) {
    double x = m->ox[i], y = m->oy[i];
    if (!std::isfinite(x) || !std::isfinite(y)) m->finite = false;
    m->minx = std::fmin(m->minx, x);
    m->maxx = std::fmax(m->maxx, x);
    m->miny = std::fmin(m->miny, y);
    m->maxy = std::fmax(m->maxy, y);
    if (x != std::floor(x) || y != std::floor(y) || std::fabs(x) > 8.0e6 || std::fabs(y) > 8.0e6)
      lattice = false;
  }
  int32_t st = PP_OK;
  if (n > 0 && m->finite) {
    if (lattice) {
      std::vector<int64_t> ix(n), iy(n);
      for (size_t i = 0; i < n; ++i) {
        ix[i] = (int64_t)m->ox[i];
        iy[i] = (int64_t)m->oy[i];
      }
      st = build_lattice(ctx, m, ix, iy, (int64_t)m->minx, (int64_t)m->miny, (int64_t)m->maxx,
                         (int64_t)m->maxy);
      if (st == PP_ERR_UNSUPPORTED) st = PP_OK;  // too large to rasterise: use the bucket path
    }
    if (st == PP_OK && !m->lattice) st = build_buckets(ctx, m);
    if (st == PP_OK) {
      hipError_t e = hipMalloc((void**)&m->d_ox, n * sizeof(double));
      if (e == hipSuccess) e = hipMalloc((void**)&m->d_oy, n * sizeof(double));
      if (e == hipSuccess) e = hipMemcpy(m->d_ox, m->ox.data(), n * sizeof(double), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(m->d_oy, m->oy.data(), n * sizeof(double), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        ctx->last_hip_error = (int)e;
        st = PP_ERR_HIP;
      }
    }
  }
  if (st != PP_OK) {
    map_free_device(m);
    delete m;
    return st;
  }
  *out = m;
  return PP_OK;
}

extern "C" int32_t pp_map_create(pp_ctx* ctx, const double* ox, const double* oy, uint64_t n,
                                 pp_map** out) {
  if (!ctx || !out || (n && (!ox || !oy))) return PP_ERR_INVALID;
  *out = nullptr;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  pp_map* m = new pp_map();
  m->ctx = ctx;
  m->device = ctx->device;
  m->ox.assign(ox, ox + n);
  m->oy.assign(oy, oy + n);
  return finish_map(ctx, m, out);
}

extern "C" int32_t pp_map_create_raster(pp_ctx* ctx, const uint8_t* occ, uint64_t width,
                                        uint64_t height, int64_t x0, int64_t y0, pp_map** out) {
  if (!ctx || !out || !occ || width == 0 || height == 0) return PP_ERR_INVALID;
  *out = nullptr;
  PP_HIP(ctx, hipSetDevice(ctx->device));
  pp_map* m = new pp_map();
  m->ctx = ctx;
  m->device = ctx->device;
  for (uint64_t y = 0; y < height; ++y)
    for (uint64_t x = 0; x < width; ++x)
      if (occ[y * width + x]) {
        m->ox.push_back((double)((int64_t)x + x0));
        m->oy.push_back((double)((int64_t)y + y0));
      }
  return finish_map(ctx, m, out);
}

extern "C" void pp_map_destroy(pp_map* m) {
  if (!m) return;
  // the context may already be gone; hipFree waits for work that still uses the buffers
  (void)hipSetDevice(m->device);
  map_free_device(m);
  delete m;
}

extern "C" uint64_t pp_map_num_obstacles(const pp_map* m) { return m ? m->ox.size() : 0; }
extern "C" int32_t pp_map_is_lattice(const pp_map* m) { return (m && m->lattice) ? 1 : 0; }

// calc_parameters, hybrid_astar.rs:522-555
extern "C" int32_t pp_map_param(const pp_map* m, const pp_config* conf, pp_param* out) {
  if (!m || !conf || !out) return PP_ERR_INVALID;
  const double kPi = 3.14159265358979323846;
  double xyreso = conf->xy_reso, yawreso = conf->yaw_reso;
  out->minx = pp_sat_usize(std::round(m->minx / xyreso));
  out->miny = pp_sat_usize(std::round(m->miny / xyreso));
  out->maxx = pp_sat_usize(std::round(m->maxx / xyreso));
  out->maxy = pp_sat_usize(std::round(m->maxy / xyreso));
  out->xw = out->maxx - out->minx;
  out->yw = out->maxy - out->miny;
  out->minyaw = pp_sat_isize(std::round(-kPi / yawreso)) - 1;
  out->xyreso = xyreso;
  out->yawreso = yawreso;
  return PP_OK;
}
