"""ctypes wrapper over oracle/liborc.so -- the CPU checker used by the tests.

Builds the oracle with `make -C oracle` if the shared object is missing (gcc is
present both in the build container and on the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from path_planner_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_PATH = os.path.join(ROOT, "oracle", "liborc.so")

_lib = None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORC_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(ORC_PATH)
    vp = C.c_void_p
    d, u8, i32, i64, u64 = A.c_double_p, A.c_u8_p, A.c_i32_p, A.c_i64_p, A.c_u64_p
    cfg = C.POINTER(A.Config)
    W = C.POINTER(A.RsWord)
    sig = {
        "orc_map_create": (vp, [d, d, C.c_uint64]),
        "orc_map_destroy": (None, [vp]),
        "orc_map_param": (C.c_int32, [vp, cfg, C.POINTER(A.Param)]),
        "orc_collision_flags": (C.c_int32, [vp, cfg, C.c_uint64, d, d, d, u8, C.c_int32]),
        "orc_collision_flags_mt": (C.c_int32, [vp, cfg, C.c_uint64, d, d, d, u8, C.c_int32]),
        "orc_collision_flags_mt_mode": (C.c_int32, [vp, cfg, C.c_uint64, d, d, d, u8, C.c_int32, C.c_int32]),
        "orc_motion_set": (C.c_int32, [cfg, d, d, C.c_uint64, u64]),
        "orc_expand_batch": (C.c_int32, [vp, cfg, C.c_uint64, d, d, d, i64, d, d,
                                         C.POINTER(A.ExpandOut)]),
        "orc_rollout_points": (C.c_int32, [cfg] + [C.c_double] * 5 + [d, d, d, C.c_uint64, u64]),
        "orc_obsmap": (C.c_int32, [d, d, C.c_uint64, C.c_double, C.c_double, u8, C.c_uint64, u64,
                                   C.c_int32]),
        "orc_heuristic": (C.c_int32, [d, d, C.c_uint64] + [C.c_double] * 4 + [d, C.c_uint64, u64]),
        "orc_wavefront": (C.c_int32, [C.c_uint64, C.c_uint64, u8, C.c_uint64, C.c_uint64, d]),
        "orc_rs_words": (C.c_int32, [C.c_double] * 7 + [i32, W]),
        "orc_rs_all_paths": (C.c_int32, [C.c_double] * 8 + [i32, W, u64, d, d, d, i64, C.c_uint64]),
        "orc_rs_batch": (C.c_int32, [C.c_uint64, d, d, d, d, d, d, C.c_double, i32, i32, W, u8,
                                     C.c_int32]),
        "orc_rs_words_batch": (C.c_int32, [C.c_uint64, d, d, d, d, d, d, C.c_double, i32, W, C.c_int32]),
        "orc_rs_path_cost": (C.c_double, [W, cfg]),
        "orc_analytic_batch": (C.c_int32, [vp, cfg, C.c_uint64, d, d, d, d, d, d, u8, i32, i32, W, d]),
        "orc_plan": (C.c_int32, [vp, cfg, d, d, C.POINTER(A.Path), C.POINTER(A.PlanStats)]),
        "orc_set_max_expansions": (None, [C.c_uint64]),
        "orc_astar": (C.c_int32, [d, d, C.c_uint64] + [C.c_double] * 6 + [d, d, C.c_uint64, u64, u64,
                                                                           C.c_uint64, u64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class OracleMap:
    def __init__(self, ox, oy):
        self.ox, self.oy = _f64(ox), _f64(oy)
        self.h = lib().orc_map_create(_p(self.ox), _p(self.oy), len(self.ox))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # module globals are gone at interpreter exit
            lib().orc_map_destroy(self.h)
            self.h = None

    def param(self, conf):
        P = A.Param()
        st = lib().orc_map_param(self.h, C.byref(conf), C.byref(P))
        assert st == 0
        return P

    def collision_flags(self, conf, x, y, yaw, mode=1, threads=0):
        x, y, yaw = _f64(x), _f64(y), _f64(yaw)
        flags = np.zeros(len(x), dtype=np.uint8)
        if threads:
            st = lib().orc_collision_flags_mt_mode(self.h, C.byref(conf), len(x), _p(x), _p(y), _p(yaw),
                                                   _p(flags, C.c_uint8), threads, mode)
        else:
            st = lib().orc_collision_flags(self.h, C.byref(conf), len(x), _p(x), _p(y), _p(yaw),
                                           _p(flags, C.c_uint8), mode)
        return st, flags

    def expand(self, conf, px, py, pyaw, pdir, psteer, pcost):
        px, py, pyaw, psteer, pcost = map(_f64, (px, py, pyaw, psteer, pcost))
        pdir = np.ascontiguousarray(pdir, dtype=np.int64)
        steer, _ = motion_set(conf)
        n, P = len(px), len(steer)
        out = alloc_expand(n * P)
        st = lib().orc_expand_batch(self.h, C.byref(conf), n, _p(px), _p(py), _p(pyaw),
                                    _p(pdir, C.c_int64), _p(psteer), _p(pcost),
                                    C.byref(out["struct"]))
        return st, out

    def analytic(self, conf, s, g):
        s, g = _f64(s).reshape(-1, 3), _f64(g).reshape(-1, 3)
        n = len(s)
        cols = [np.ascontiguousarray(s[:, k]) for k in range(3)] + \
               [np.ascontiguousarray(g[:, k]) for k in range(3)]
        found = np.zeros(n, np.uint8)
        idx = np.zeros(n, np.int32)
        npaths = np.zeros(n, np.int32)
        words = (A.RsWord * n)()
        cost = np.zeros(n)
        st = lib().orc_analytic_batch(self.h, C.byref(conf), n, *[_p(c) for c in cols],
                                      _p(found, C.c_uint8), _p(idx, C.c_int32),
                                      _p(npaths, C.c_int32), words, _p(cost))
        return st, found, idx, npaths, words, cost

    def plan(self, conf, start, goal, cap=100000, closed_cap=100000):
        return run_plan(lambda *a: lib().orc_plan(self.h, *a), conf, start, goal, cap, closed_cap)


def set_max_expansions(n):
    lib().orc_set_max_expansions(n)


def run_plan(fn, conf, start, goal, cap=100000, closed_cap=100000):
    """Shared driver for orc_plan / pp_plan: fn(conf*, start*, goal*, Path*, PlanStats*)."""
    s, g = _f64(start), _f64(goal)
    x, y, yaw = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    dirs = np.zeros(cap, np.int64)
    closed = np.zeros(closed_cap, np.uint64)
    path = A.Path(_p(x), _p(y), _p(yaw), _p(dirs, C.c_int64), cap, 0, 0.0)
    stats = A.PlanStats()
    stats.closed_cap = closed_cap
    stats.closed_order = _p(closed, C.c_uint64)
    st = fn(C.byref(conf), _p(s), _p(g), C.byref(path), C.byref(stats))
    n = int(path.len) if st == 0 else 0
    return {"status": st, "x": x[:n].copy(), "y": y[:n].copy(), "yaw": yaw[:n].copy(),
            "dir": dirs[:n].copy(), "cost": path.cost, "stats": stats,
            "closed": closed[: min(int(stats.expansions), closed_cap)].copy()}


def alloc_expand(m):
    arrs = {"valid": np.zeros(m, np.uint8), "xind": np.zeros(m, np.uint64),
            "yind": np.zeros(m, np.uint64), "yawind": np.zeros(m, np.int64),
            "cost": np.zeros(m), "end_x": np.zeros(m), "end_y": np.zeros(m), "end_yaw": np.zeros(m)}
    arrs["struct"] = A.ExpandOut(_p(arrs["valid"], C.c_uint8), _p(arrs["xind"], C.c_uint64),
                                 _p(arrs["yind"], C.c_uint64), _p(arrs["yawind"], C.c_int64),
                                 _p(arrs["cost"]), _p(arrs["end_x"]), _p(arrs["end_y"]),
                                 _p(arrs["end_yaw"]))
    return arrs


def motion_set(conf):
    n = C.c_uint64(0)
    cap = 2 * (2 * int(conf.n_steer))
    steer, direc = np.zeros(cap), np.zeros(cap)
    st = lib().orc_motion_set(C.byref(conf), _p(steer), _p(direc), cap, C.byref(n))
    assert st == 0
    return steer[: n.value].copy(), direc[: n.value].copy()


def rollout_points(conf, x0, y0, yaw0, u, d):
    cap = 4096
    x, y, yaw = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    n = C.c_uint64(0)
    st = lib().orc_rollout_points(C.byref(conf), x0, y0, yaw0, u, d, _p(x), _p(y), _p(yaw), cap,
                                  C.byref(n))
    assert st == 0
    return x[: n.value].copy(), y[: n.value].copy(), yaw[: n.value].copy()


def obsmap(ox, oy, reso, rr, mode=1):
    ox, oy = _f64(ox), _f64(oy)
    dims = np.zeros(6, np.uint64)
    st = lib().orc_obsmap(_p(ox), _p(oy), len(ox), reso, rr, None, 0, _p(dims, C.c_uint64), mode)
    assert st == A.PP_ERR_CAPACITY or (st == 0 and dims[4] * dims[5] == 0), st
    out = np.zeros(int(dims[4] * dims[5]), np.uint8)
    st = lib().orc_obsmap(_p(ox), _p(oy), len(ox), reso, rr, _p(out, C.c_uint8), len(out),
                          _p(dims, C.c_uint64), mode)
    return st, out.reshape(int(dims[4]), int(dims[5])), dims


def heuristic(ox, oy, gx, gy, reso, rr):
    ox, oy = _f64(ox), _f64(oy)
    dims = np.zeros(6, np.uint64)
    st = lib().orc_heuristic(_p(ox), _p(oy), len(ox), gx, gy, reso, rr, None, 0,
                             _p(dims, C.c_uint64))
    out = np.zeros(int(dims[4] * dims[5]))
    st = lib().orc_heuristic(_p(ox), _p(oy), len(ox), gx, gy, reso, rr, _p(out), len(out),
                             _p(dims, C.c_uint64))
    return st, out.reshape(int(dims[4]), int(dims[5])), dims


def astar(ox, oy, sx, sy, gx, gy, reso, rr, cap=100000, closed_cap=1000000):
    """(status, (pathx, pathy) or None, closing order)."""
    ox, oy = _f64(ox), _f64(oy)
    px, py = np.zeros(cap), np.zeros(cap)
    closed = np.zeros(closed_cap, np.uint64)
    n, ex = C.c_uint64(0), C.c_uint64(0)
    st = lib().orc_astar(_p(ox), _p(oy), len(ox), sx, sy, gx, gy, reso, rr, _p(px), _p(py), cap,
                         C.byref(n), _p(closed, C.c_uint64), closed_cap, C.byref(ex))
    order = closed[: min(ex.value, closed_cap)].copy()
    if st != 0:
        return st, None, order
    return st, (px[: n.value].copy(), py[: n.value].copy()), order


def wavefront(obs, goal_x, goal_y):
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    xw, yw = obs.shape
    cost = np.zeros((xw, yw))
    st = lib().orc_wavefront(xw, yw, _p(obs, C.c_uint8), goal_x, goal_y, _p(cost))
    return st, cost


def rs_all_paths(q0, q1, maxc, step, traj_cap=200000):
    n = C.c_int32(0)
    words = (A.RsWord * A.PP_RS_MAX_PATHS)()
    off = np.zeros(A.PP_RS_MAX_PATHS + 1, np.uint64)
    tx, ty, tyaw = np.zeros(traj_cap), np.zeros(traj_cap), np.zeros(traj_cap)
    td = np.zeros(traj_cap, np.int64)
    st = lib().orc_rs_all_paths(*q0, *q1, maxc, step, C.byref(n), words, _p(off, C.c_uint64),
                                _p(tx), _p(ty), _p(tyaw), _p(td, C.c_int64), traj_cap)
    return unpack_paths(st, n.value, words, off, tx, ty, tyaw, td)


def unpack_paths(st, n, words, off, tx, ty, tyaw, td):
    paths = []
    if st == 0:
        for k in range(n):
            a, b = int(off[k]), int(off[k + 1])
            paths.append({"ctypes": words[k].word(), "lengths": words[k].lens(),
                          "total": words[k].total_length, "x": tx[a:b].copy(), "y": ty[a:b].copy(),
                          "yaw": tyaw[a:b].copy(), "dir": td[a:b].copy()})
    return st, paths


def rs_words(q0, q1, maxc):
    """generate_path's words for one pair, in generation order: [(ctypes, lengths, total)] with
    lengths in metres."""
    n = C.c_int32(0)
    words = (A.RsWord * A.PP_RS_MAX_PATHS)()
    st = lib().orc_rs_words(*map(float, q0), *map(float, q1), maxc, C.byref(n), words)
    assert st == 0, st
    return [(words[k].word(), words[k].lens(), words[k].total_length) for k in range(n.value)]


def rs_batch(s, g, maxc, threads=1):
    s, g = _f64(s).reshape(-1, 3), _f64(g).reshape(-1, 3)
    n = len(s)
    cols = [np.ascontiguousarray(s[:, k]) for k in range(3)] + \
           [np.ascontiguousarray(g[:, k]) for k in range(3)]
    npaths = np.zeros(n, np.int32)
    best = np.zeros(n, np.int32)
    words = (A.RsWord * n)()
    flags = np.zeros(n, np.uint8)
    st = lib().orc_rs_batch(n, *[_p(c) for c in cols], maxc, _p(npaths, C.c_int32),
                            _p(best, C.c_int32), words, _p(flags, C.c_uint8), threads)
    return st, npaths, best, words, flags


def rs_words_batch(s, g, maxc, threads=8):
    """generate_path's words for n pairs: (n_paths[n] (-1 = degenerate), RsWord array of n * 46)."""
    s, g = _f64(s).reshape(-1, 3), _f64(g).reshape(-1, 3)
    n = len(s)
    cols = [np.ascontiguousarray(s[:, k]) for k in range(3)] + \
           [np.ascontiguousarray(g[:, k]) for k in range(3)]
    npaths = np.zeros(n, np.int32)
    words = (A.RsWord * (n * A.PP_RS_MAX_PATHS))()
    st = lib().orc_rs_words_batch(n, *[_p(c) for c in cols], maxc, _p(npaths, C.c_int32), words, threads)
    assert st == 0, st
    return npaths, words


def rs_path_cost(word, conf):
    return lib().orc_rs_path_cost(C.byref(word), C.byref(conf))


def words_to_numpy(words, n):
    """(lengths[n,5], total[n], ctypes[n,5], nseg[n]) from an RsWord array."""
    buf = np.frombuffer(words, dtype=np.uint8).reshape(n, 56)
    lengths = buf[:, :40].copy().view(np.float64).reshape(n, 5)
    total = buf[:, 40:48].copy().view(np.float64).reshape(n)
    return lengths, total, buf[:, 48:53].copy(), buf[:, 53].copy()
