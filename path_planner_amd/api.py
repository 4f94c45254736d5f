"""Host-side mirror of the reference planner's functions over the C ABI (include/pp.h).

Function names and argument meaning follow the reference (src/hybrid_astar.rs,
src/reeds_shepp.rs, src/astar.rs); every call goes through libpp_hip.so.  numpy arrays are
host data (copied by the library); torch CUDA tensors are passed as device pointers.
"""
import ctypes as C
import math

import numpy as np

from . import _abi as A


class PlannerError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = A.load_library().pp_strerror(status).decode()
        super().__init__(f"{where}: {msg} (status {status})")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _chk(st, where, ok=(A.PP_OK,)):
    if st not in ok:
        raise PlannerError(st, where)
    return st


class Context:
    """One device + one stream (pp_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = A.load_library()
        h = C.c_void_p()
        _chk(self.lib.pp_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)),
             "pp_ctx_create")
        self.h = h
        self.device = device

    def set_max_expansions(self, n):
        _chk(self.lib.pp_ctx_set_max_expansions(self.h, n), "pp_ctx_set_max_expansions")

    def set_heuristic_limit(self, limit):
        """Cost limit (cells) of plan_batch's first heuristic relaxation; negative = automatic."""
        _chk(self.lib.pp_ctx_set_heuristic_limit(self.h, float(limit)), "pp_ctx_set_heuristic_limit")

    def set_heuristic_workspace(self, nbytes):
        """Memory budget of plan_batch's joint heuristic relaxation (default 8 GiB)."""
        _chk(self.lib.pp_ctx_set_heuristic_workspace(self.h, int(nbytes)), "pp_ctx_set_heuristic_workspace")

    def set_host_threads(self, n):
        """Worker threads plan_batch may start beside the calling thread (default 15)."""
        _chk(self.lib.pp_ctx_set_host_threads(self.h, int(n)), "pp_ctx_set_host_threads")

    def set_batch_parts(self, parts):
        """Parts a large plan_batch runs in on this context (0 = automatic, 1 = never split)."""
        _chk(self.lib.pp_ctx_set_batch_parts(self.h, int(parts)), "pp_ctx_set_batch_parts")

    def set_table_budget(self, nbytes):
        """Device memory plan_batch may take for its per-query tables of known cells (default 16 GiB,
        0 = no tables); the plans do not depend on it."""
        _chk(self.lib.pp_ctx_set_table_budget(self.h, int(nbytes)), "pp_ctx_set_table_budget")

    def device_bytes(self):
        """Device memory the context holds in its grown-on-demand buffers."""
        return int(self.lib.pp_ctx_device_bytes(self.h))

    def synchronize(self):
        _chk(self.lib.pp_ctx_synchronize(self.h), "pp_ctx_synchronize")

    def close(self):
        if getattr(self, "h", None):
            self.lib.pp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class ObstacleMap:
    """The reference's (ox, oy) obstacle lists resident on the device (pp_map)."""

    def __init__(self, ctx, ox=None, oy=None, raster=None, origin=(0, 0)):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        if raster is not None:
            occ = np.ascontiguousarray(raster, dtype=np.uint8)
            _chk(self.lib.pp_map_create_raster(ctx.h, _p(occ, C.c_uint8), occ.shape[1], occ.shape[0],
                                               origin[0], origin[1], C.byref(h)), "pp_map_create_raster")
        else:
            ox, oy = _f64(ox), _f64(oy)
            assert ox.shape == oy.shape
            _chk(self.lib.pp_map_create(ctx.h, _p(ox), _p(oy), len(ox), C.byref(h)), "pp_map_create")
        self.h = h

    @property
    def num_obstacles(self):
        return int(self.lib.pp_map_num_obstacles(self.h))

    @property
    def is_lattice(self):
        return bool(self.lib.pp_map_is_lattice(self.h))

    def param(self, conf):
        P = A.Param()
        _chk(self.lib.pp_map_param(self.h, C.byref(conf), C.byref(P)), "pp_map_param")
        return P

    def close(self):
        if getattr(self, "h", None):
            self.lib.pp_map_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


# ------------------------------------------------------------------ is_collision
def collision_flags(ctx, omap, conf, x, y, yaw):
    """Per-pose is_collision (hybrid_astar.rs:417-448).  Returns uint8 flags
    (bit0 hit, bit1 guard-band, bit2 non-finite)."""
    x, y, yaw = _f64(x), _f64(y), _f64(yaw)
    flags = np.zeros(len(x), np.uint8)
    _chk(ctx.lib.pp_collision_batch(ctx.h, omap.h, C.byref(conf), len(x), _p(x), _p(y), _p(yaw),
                                    _p(flags, C.c_uint8)), "pp_collision_batch")
    return flags


def collision_flags_dev(ctx, omap, conf, x, y, yaw, flags):
    """Same on torch CUDA tensors (float64 x3, uint8 out); asynchronous on the context stream."""
    n = x.numel()
    _chk(ctx.lib.pp_collision_batch_dev(ctx.h, omap.h, C.byref(conf), n, x.data_ptr(), y.data_ptr(),
                                        yaw.data_ptr(), flags.data_ptr()), "pp_collision_batch_dev")


def is_collision(ctx, omap, conf, x, y, yaw):
    """is_collision(&x, &y, &yaw, param, conf) -> bool, as the reference calls it."""
    x, y, yaw = _f64(x), _f64(y), _f64(yaw)
    hit = C.c_int32(0)
    amb = C.c_uint64(0)
    _chk(ctx.lib.pp_is_collision(ctx.h, omap.h, C.byref(conf), len(x), _p(x), _p(y), _p(yaw),
                                 C.byref(hit), C.byref(amb)), "pp_is_collision")
    return bool(hit.value)


# ------------------------------------------------------------------ calc_next_node
def calc_motion_set(conf):
    n = C.c_uint64(0)
    cap = 4 * int(conf.n_steer)
    steer, direc = np.zeros(cap), np.zeros(cap)
    _chk(A.load_library().pp_motion_set(C.byref(conf), _p(steer), _p(direc), cap, C.byref(n)),
         "pp_motion_set")
    return steer[: n.value].copy(), direc[: n.value].copy()


def expand(ctx, omap, conf, px, py, pyaw, pdir, psteer, pcost):
    """All primitives of each parent (hybrid_astar.rs:180-183 -> :252-320).  Arrays of n*P."""
    px, py, pyaw, psteer, pcost = map(_f64, (px, py, pyaw, psteer, pcost))
    pdir = np.ascontiguousarray(pdir, dtype=np.int64)
    steer, _ = calc_motion_set(conf)
    m = len(px) * len(steer)
    out = {"valid": np.zeros(m, np.uint8), "xind": np.zeros(m, np.uint64),
           "yind": np.zeros(m, np.uint64), "yawind": np.zeros(m, np.int64), "cost": np.zeros(m),
           "end_x": np.zeros(m), "end_y": np.zeros(m), "end_yaw": np.zeros(m)}
    s = A.ExpandOut(_p(out["valid"], C.c_uint8), _p(out["xind"], C.c_uint64),
                    _p(out["yind"], C.c_uint64), _p(out["yawind"], C.c_int64), _p(out["cost"]),
                    _p(out["end_x"]), _p(out["end_y"]), _p(out["end_yaw"]))
    _chk(ctx.lib.pp_expand_batch(ctx.h, omap.h, C.byref(conf), len(px), _p(px), _p(py), _p(pyaw),
                                 _p(pdir, C.c_int64), _p(psteer), _p(pcost), C.byref(s)),
         "pp_expand_batch")
    return out


def expand_dev(ctx, omap, conf, px, py, pyaw, pdir, psteer, pcost, out):
    """expand() on torch CUDA tensors: parents (float64 / int64 pdir) and out = dict of n*P tensors
    valid (uint8), xind, yind (int64 storage of u64), yawind (int64), cost, end_x, end_y, end_yaw."""
    vp = C.c_void_p
    s = A.ExpandOut(C.cast(vp(out["valid"].data_ptr()), A.c_u8_p),
                    C.cast(vp(out["xind"].data_ptr()), A.c_u64_p), C.cast(vp(out["yind"].data_ptr()), A.c_u64_p),
                    C.cast(vp(out["yawind"].data_ptr()), A.c_i64_p), C.cast(vp(out["cost"].data_ptr()), A.c_double_p),
                    C.cast(vp(out["end_x"].data_ptr()), A.c_double_p),
                    C.cast(vp(out["end_y"].data_ptr()), A.c_double_p),
                    C.cast(vp(out["end_yaw"].data_ptr()), A.c_double_p))
    _chk(ctx.lib.pp_expand_batch_dev(ctx.h, omap.h, C.byref(conf), px.numel(), px.data_ptr(), py.data_ptr(),
                                     pyaw.data_ptr(), pdir.data_ptr(), psteer.data_ptr(), pcost.data_ptr(),
                                     C.byref(s)), "pp_expand_batch_dev")


def rollout_points(ctx, conf, x0, y0, yaw0, u, d):
    cap = 1 << 16
    x, y, yaw = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    n = C.c_uint64(0)
    _chk(ctx.lib.pp_rollout_points(ctx.h, C.byref(conf), x0, y0, yaw0, u, d, _p(x), _p(y), _p(yaw),
                                   cap, C.byref(n)), "pp_rollout_points")
    return x[: n.value].copy(), y[: n.value].copy(), yaw[: n.value].copy()


# ------------------------------------------------------------------ heuristic
def calc_obsmap(ctx, omap, reso, rr):
    dims = np.zeros(6, np.uint64)
    st = ctx.lib.pp_obsmap(ctx.h, omap.h, reso, rr, None, 0, _p(dims, C.c_uint64))
    _chk(st, "pp_obsmap", (A.PP_OK, A.PP_ERR_CAPACITY))
    out = np.zeros(int(dims[4] * dims[5]), np.uint8)
    _chk(ctx.lib.pp_obsmap(ctx.h, omap.h, reso, rr, _p(out, C.c_uint8), len(out),
                           _p(dims, C.c_uint64)), "pp_obsmap")
    return out.reshape(int(dims[4]), int(dims[5])), dims


def calc_holonomic_heuristic_with_obstacle(ctx, omap, gx, gy, reso, rr):
    """astar.rs:134-198: hmap[x - minx][y - miny], inf where unreachable."""
    dims = np.zeros(6, np.uint64)
    st = ctx.lib.pp_heuristic(ctx.h, omap.h, gx, gy, reso, rr, None, 0, _p(dims, C.c_uint64))
    _chk(st, "pp_heuristic", (A.PP_OK, A.PP_ERR_CAPACITY))
    out = np.zeros(int(dims[4] * dims[5]))
    _chk(ctx.lib.pp_heuristic(ctx.h, omap.h, gx, gy, reso, rr, _p(out), len(out),
                              _p(dims, C.c_uint64)), "pp_heuristic")
    return out.reshape(int(dims[4]), int(dims[5])), dims


def wavefront_dev(ctx, obs, goal_x, goal_y, cost):
    """astar.rs:153-198 on a device raster obs[x, y] (torch uint8) into cost[x, y] (float64)."""
    xw, yw = obs.shape
    iters = C.c_uint64(0)
    _chk(ctx.lib.pp_wavefront_dev(ctx.h, xw, yw, obs.data_ptr(), goal_x, goal_y, cost.data_ptr(),
                                  C.byref(iters)), "pp_wavefront_dev")
    return iters.value


def wavefront_tiled(ctxs, obs, goal_x, goal_y, band=0.0):
    """astar.rs:153-198 on a host raster obs[x, y] (uint8, non-zero = blocked) cut into column slabs
    over the contexts of this process (pp_wavefront_tiled).  Returns (cost[x, y] float64, exchange
    rounds)."""
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    xw, yw = obs.shape
    cost = np.empty((xw, yw), np.float64)
    rounds = C.c_uint64(0)
    hc = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    _chk(ctxs[0].lib.pp_wavefront_tiled(hc, len(ctxs), xw, yw, _p(obs, C.c_uint8), goal_x, goal_y, _p(cost),
                                        float(band), C.byref(rounds)), "pp_wavefront_tiled")
    return cost, rounds.value


def wavefront_stats(ctx):
    out = np.zeros(4, np.uint64)
    _chk(ctx.lib.pp_wavefront_stats(ctx.h, _p(out, C.c_uint64)), "pp_wavefront_stats")
    return dict(zip(("rounds", "tile_visits", "sweep_pairs", "tiles"), (int(v) for v in out)))


def plan_batch_counters(ctx):
    """Work counters of the context's last plan_batch (pp_plan_batch_counters)."""
    out = np.zeros(8, np.uint64)
    _chk(ctx.lib.pp_plan_batch_counters(ctx.h, _p(out, C.c_uint64)), "pp_plan_batch_counters")
    return dict(zip(("rounds", "rounds_under_64_live", "nodes_sent", "goal_grids", "longest_query_expansions",
                     "expansions", "queries", "lanes"), (int(v) for v in out)))


# ------------------------------------------------------------------ Reeds-Shepp
def _cols(q):
    q = _f64(q).reshape(-1, 3)
    return [np.ascontiguousarray(q[:, k]) for k in range(3)]


def rs_batch(ctx, q0, q1, maxc):
    """generate_path + calc_optimal_path's selection for n pairs (reeds_shepp.rs:60-85, :707-725)."""
    s, g = _cols(q0), _cols(q1)
    n = len(s[0])
    npaths = np.zeros(n, np.int32)
    best = np.zeros(n, np.int32)
    words = (A.RsWord * n)()
    flags = np.zeros(n, np.uint8)
    _chk(ctx.lib.pp_rs_batch(ctx.h, n, *[_p(c) for c in s + g], maxc, _p(npaths, C.c_int32),
                             _p(best, C.c_int32), words, _p(flags, C.c_uint8)), "pp_rs_batch")
    return npaths, best, words, flags


def rs_words_batch(ctx, q0, q1, maxc):
    """generate_path (reeds_shepp.rs:707-725) for n pairs, every word kept: (n_paths[n], RsWord array
    of n * PP_RS_MAX_PATHS with pair i's words from i * PP_RS_MAX_PATHS on, flags[n])."""
    s, g = _cols(q0), _cols(q1)
    n = len(s[0])
    npaths = np.zeros(n, np.int32)
    words = (A.RsWord * (n * A.PP_RS_MAX_PATHS))()
    flags = np.zeros(n, np.uint8)
    _chk(ctx.lib.pp_rs_words_batch(ctx.h, n, *[_p(c) for c in s + g], maxc, _p(npaths, C.c_int32), words,
                                   _p(flags, C.c_uint8)), "pp_rs_words_batch")
    return npaths, words, flags


def calc_all_paths(ctx, sx, sy, syaw, gx, gy, gyaw, maxc, step_size, traj_cap=400000):
    """reeds_shepp.rs:87-128.  Returns a list of dicts (ctypes, lengths, total, x, y, yaw, dir)."""
    n = C.c_int32(0)
    words = (A.RsWord * A.PP_RS_MAX_PATHS)()
    off = np.zeros(A.PP_RS_MAX_PATHS + 1, np.uint64)
    tx, ty, tyaw = np.zeros(traj_cap), np.zeros(traj_cap), np.zeros(traj_cap)
    td = np.zeros(traj_cap, np.int64)
    _chk(ctx.lib.pp_rs_all_paths(ctx.h, sx, sy, syaw, gx, gy, gyaw, maxc, step_size, C.byref(n),
                                 words, _p(off, C.c_uint64), _p(tx), _p(ty), _p(tyaw),
                                 _p(td, C.c_int64), traj_cap), "pp_rs_all_paths")
    paths = []
    for k in range(n.value):
        a, b = int(off[k]), int(off[k + 1])
        paths.append({"ctypes": words[k].word(), "lengths": words[k].lens(),
                      "total": words[k].total_length, "x": tx[a:b].copy(), "y": ty[a:b].copy(),
                      "yaw": tyaw[a:b].copy(), "dir": td[a:b].copy()})
    return paths


def calc_optimal_path(ctx, sx, sy, syaw, gx, gy, gyaw, maxc, step_size, traj_cap=100000):
    """reeds_shepp.rs:60-85."""
    w = A.RsWord()
    tx, ty, tyaw = np.zeros(traj_cap), np.zeros(traj_cap), np.zeros(traj_cap)
    td = np.zeros(traj_cap, np.int64)
    n = C.c_uint64(0)
    _chk(ctx.lib.pp_rs_optimal_path(ctx.h, sx, sy, syaw, gx, gy, gyaw, maxc, step_size, C.byref(w),
                                    _p(tx), _p(ty), _p(tyaw), _p(td, C.c_int64), traj_cap,
                                    C.byref(n)), "pp_rs_optimal_path")
    k = n.value
    return {"ctypes": w.word(), "lengths": w.lens(), "total": w.total_length, "x": tx[:k].copy(),
            "y": ty[:k].copy(), "yaw": tyaw[:k].copy(), "dir": td[:k].copy()}


def calc_rs_path_cost(word, conf):
    return A.load_library().pp_rs_path_cost(C.byref(word), C.byref(conf))


def analytic_expansion(ctx, omap, conf, starts, goals):
    """analystic_expansion (hybrid_astar.rs:380-415) for n (start, goal) pose pairs."""
    s, g = _cols(starts), _cols(goals)
    n = len(s[0])
    found = np.zeros(n, np.uint8)
    idx = np.zeros(n, np.int32)
    npaths = np.zeros(n, np.int32)
    words = (A.RsWord * n)()
    cost = np.zeros(n)
    _chk(ctx.lib.pp_analytic_batch(ctx.h, omap.h, C.byref(conf), n, *[_p(c) for c in s + g],
                                   _p(found, C.c_uint8), _p(idx, C.c_int32), _p(npaths, C.c_int32),
                                   words, _p(cost)), "pp_analytic_batch")
    return found, idx, npaths, words, cost


def analytic_expansion_dev(ctx, omap, conf, q6, found, idx, npaths, words, cost):
    """pp_analytic_batch_dev on torch device tensors: q6 float64 (6, n) rows sx, sy, syaw, gx, gy,
    gyaw; found uint8 (n), idx / npaths int32 (n), words uint8 (n, 56), cost float64 (n)."""
    n = q6.shape[1]
    _chk(ctx.lib.pp_analytic_batch_dev(ctx.h, omap.h, C.byref(conf), n, q6.data_ptr(), found.data_ptr(),
                                       idx.data_ptr(), npaths.data_ptr(), words.data_ptr(),
                                       cost.data_ptr()), "pp_analytic_batch_dev")


def analytic_expansion_dev_static(ctx, omap, conf, q6, found, idx, npaths, words, cost):
    """analytic_expansion_dev through the entry point that allocates nothing (at most
    pp_analytic_static_max() pairs; PP_ERR_CAPACITY above)."""
    n = q6.shape[1]
    _chk(ctx.lib.pp_analytic_batch_dev_static(ctx.h, omap.h, C.byref(conf), n, q6.data_ptr(),
                                              found.data_ptr(), idx.data_ptr(), npaths.data_ptr(),
                                              words.data_ptr(), cost.data_ptr()),
         "pp_analytic_batch_dev_static")


# ------------------------------------------------------------------ the planner
def hybrid_astar_planning(ctx, omap, conf, sx, sy, syaw, gx, gy, gyaw, cap=200000,
                          closed_cap=1 << 20):
    """hybrid_astar_planning (hybrid_astar.rs:89-216) with rec = None.  Returns a dict with
    x, y, yaw, direction, cost (HybridAstarPath) plus search statistics, or None for no path."""
    s, g = _f64([sx, sy, syaw]), _f64([gx, gy, gyaw])
    x, y, yaw = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    dirs = np.zeros(cap, np.int64)
    closed = np.zeros(closed_cap, np.uint64)
    path = A.Path(_p(x), _p(y), _p(yaw), _p(dirs, C.c_int64), cap, 0, 0.0)
    stats = A.PlanStats()
    stats.closed_cap = closed_cap
    stats.closed_order = _p(closed, C.c_uint64)
    st = ctx.lib.pp_plan(ctx.h, omap.h, C.byref(conf), _p(s), _p(g), C.byref(path), C.byref(stats))
    _chk(st, "pp_plan", (A.PP_OK, A.PP_NO_PATH, A.PP_ERR_LIMIT))
    n = int(path.len) if st == A.PP_OK else 0
    return {"status": st, "x": x[:n].copy(), "y": y[:n].copy(), "yaw": yaw[:n].copy(),
            "dir": dirs[:n].copy(), "cost": path.cost, "stats": stats,
            "closed": closed[: min(int(stats.expansions), closed_cap)].copy()}


def plan_batch(ctx, omap, conf, starts, goals, cap=4096, closed_cap=0, out=None):
    """n independent hybrid_astar_planning calls advanced together (pp_plan_batch).  Returns
    status[n], a list-like of paths (x, y, yaw, dir arrays of shape (n, cap) with len[n], cost[n])
    and the per-query statistics.  `out` = the result of an earlier call of the same shape: its
    path arrays are written again instead of allocating (and page-faulting in) new ones.
    With lists of contexts and maps (maps[k] created on ctxs[k]) the queries are dealt over the
    contexts by goal, one host thread each (pp_plan_batch_multi)."""
    starts = np.ascontiguousarray(_f64(starts).reshape(-1, 3))
    goals = np.ascontiguousarray(_f64(goals).reshape(-1, 3))
    n = len(starts)
    if out is not None and out["x"].shape == (n, cap):
        X, Y, YAW, D = out["x"], out["y"], out["yaw"], out["dir"]
    else:
        X, Y, YAW = np.zeros((n, cap)), np.zeros((n, cap)), np.zeros((n, cap))
        D = np.zeros((n, cap), np.int64)
    closed = np.zeros((n, max(closed_cap, 1)), np.uint64)
    # the pp_path array is filled as a structured numpy array (one vector operation per field)
    pt = np.zeros(n, dtype=[("x", "u8"), ("y", "u8"), ("yaw", "u8"), ("direction", "u8"), ("cap", "u8"),
                            ("len", "u8"), ("cost", "f8")])
    assert pt.itemsize == C.sizeof(A.Path)
    row = np.arange(n, dtype=np.uint64) * np.uint64(cap * 8)
    for name, arr in (("x", X), ("y", Y), ("yaw", YAW), ("direction", D)):
        pt[name] = np.uint64(arr.ctypes.data) + row
    pt["cap"] = cap
    paths = C.cast(pt.ctypes.data, C.POINTER(A.Path))
    stats = (A.PlanStats * n)()
    if closed_cap:
        cs, up = closed.ctypes.data, A.c_u64_p
        for i in range(n):
            stats[i].closed_cap = closed_cap
            stats[i].closed_order = C.cast(cs + i * closed_cap * 8, up)
    status = np.zeros(n, np.int32)
    if isinstance(ctx, (list, tuple)):
        k = len(ctx)
        assert k == len(omap) and k >= 1
        hc = (C.c_void_p * k)(*[c.h for c in ctx])
        hm = (C.c_void_p * k)(*[m.h for m in omap])
        st = ctx[0].lib.pp_plan_batch_multi(hc, hm, k, C.byref(conf), n, _p(starts), _p(goals), paths, stats,
                                            _p(status, C.c_int32))
        _chk(st, "pp_plan_batch_multi")
    else:
        st = ctx.lib.pp_plan_batch(ctx.h, omap.h, C.byref(conf), n, _p(starts), _p(goals), paths, stats,
                                   _p(status, C.c_int32))
        _chk(st, "pp_plan_batch")
    lens = pt["len"].astype(np.int64)
    cost = pt["cost"].copy()
    return {"status": status, "x": X, "y": Y, "yaw": YAW, "dir": D, "len": lens, "cost": cost,
            "stats": stats, "closed": closed}


def astar_planning(ctx, omap, sx, sy, gx, gy, reso, rr, cap=0, closed_cap=0):
    """astar.rs:53-132 (rec = None): (pathx, pathy) or None; with closed_cap > 0 also the
    calc_index keys in closing order."""
    if cap == 0:  # size the path first; a query without a path is already answered by this call
        closed = np.zeros(closed_cap, np.uint64)
        n, ex = C.c_uint64(0), C.c_uint64(0)
        st = ctx.lib.pp_astar_plan(ctx.h, omap.h, sx, sy, gx, gy, reso, rr, None, None, 0,
                                   C.byref(n), _p(closed, C.c_uint64) if closed_cap else None,
                                   closed_cap, C.byref(ex))
        if st == A.PP_NO_PATH:
            return None if closed_cap == 0 else (None, closed[: min(ex.value, closed_cap)].copy())
        _chk(st, "pp_astar_plan", (A.PP_ERR_CAPACITY,))
        cap = n.value
    px, py = np.zeros(cap), np.zeros(cap)
    closed = np.zeros(closed_cap, np.uint64)
    n, ex = C.c_uint64(0), C.c_uint64(0)
    st = ctx.lib.pp_astar_plan(ctx.h, omap.h, sx, sy, gx, gy, reso, rr, _p(px), _p(py), cap,
                               C.byref(n), _p(closed, C.c_uint64) if closed_cap else None,
                               closed_cap, C.byref(ex))
    order = closed[: min(ex.value, closed_cap)].copy()
    if st == A.PP_NO_PATH:
        return None if closed_cap == 0 else (None, order)
    _chk(st, "pp_astar_plan")
    path = (px[: n.value].copy(), py[: n.value].copy())
    return path if closed_cap == 0 else (path, order)


def astar_planning_batch(ctx, omap, starts, goals, reso, rr, cap=0):
    """n queries {x, y} against one raster: list of (pathx, pathy) or None, and expansions."""
    starts, goals = _f64(starts).reshape(-1, 2), _f64(goals).reshape(-1, 2)
    n = len(starts)
    off = np.zeros(n + 1, np.uint64)
    status = np.zeros(n, np.int32)
    ex = np.zeros(n, np.uint64)
    if cap == 0:
        st = ctx.lib.pp_astar_plan_batch(ctx.h, omap.h, n, _p(starts), _p(goals), reso, rr, None,
                                         None, 0, _p(off, C.c_uint64), _p(status, C.c_int32),
                                         _p(ex, C.c_uint64))
        _chk(st, "pp_astar_plan_batch", (A.PP_OK, A.PP_ERR_CAPACITY))
        cap = max(1, int(off[n]))
    px, py = np.zeros(cap), np.zeros(cap)
    _chk(ctx.lib.pp_astar_plan_batch(ctx.h, omap.h, n, _p(starts), _p(goals), reso, rr, _p(px),
                                     _p(py), cap, _p(off, C.c_uint64), _p(status, C.c_int32),
                                     _p(ex, C.c_uint64)), "pp_astar_plan_batch")
    out = []
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        out.append((px[a:b].copy(), py[a:b].copy()) if status[i] == A.PP_OK else None)
    return out, ex


def demo_maxc(conf):
    return math.tan(conf.max_steer) / conf.car.wheel_base
