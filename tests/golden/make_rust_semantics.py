#!/usr/bin/env python3
"""Fixtures for the parts of the hybrid A* path where the reference's Python twin and its Rust
crate differ, generated from the twin itself.

Run in the build container only (needs /root/reference):
    python tests/golden/make_rust_semantics.py

Two groups of files are written next to this script.

A. UN-PATCHED twin, on inputs where the twin and the crate provably agree
   twin_next_node.json   calc_next_node (hybrid_astar.py:188-235 == hybrid_astar.rs:252-320) for
                         parents whose ten-step rollout keeps yaw strictly inside (-pi, pi) -- so the
                         twin's pi_2_pi wrap (hybrid_astar.py:201-203) is the identity -- and whose
                         end-pose quotients x/reso, y/reso, yaw/yawreso stay 1e-6 away from a .5
                         tie -- so numpy's half-even rounding (hybrid_astar.py:205-207) equals
                         Rust's half-away f64::round (hybrid_astar.rs:276-278).  The steer values
                         are passed in explicitly (the crate's (i * max_steer) / n_steer).
   twin_analytic.json    analystic_expantion (hybrid_astar.py:292-320 == hybrid_astar.rs:380-415)
                         for pairs whose Reeds-Shepp path costs are pairwise distinct (by > 1e-6),
                         so the pop order of any correct priority queue is the crate's
                         (cost ascending; the tie rule never applies).

B. The twin PATCHED to the crate's semantics (every patch cites both sides), for what no input
   selection can reach: the search loop's pop order and everything downstream of it.
   astar_rust.json       the plain grid planner (astar.py:32-104 == astar.rs:53-132 under D1 and H):
                         closing order and path for the demo environment of src/main.rs:14-44,
                         157-171 and twelve random maps.
   rust_semantics.json   the demo query of src/main.rs:215-269 and 24 random queries on the demo
                         map: Param, heuristic-grid summary, closed keys in expansion order, stale
                         pops, rollouts, poses handed to is_collision, Reeds-Shepp paths generated,
                         the accepted word, HybridAstarPath.cost and the path itself.

   patch  twin (path_planner_py/src/path_planner_py)          crate (src)
   D1     np.round, half to even: hybrid_astar.py:105-108,    f64::round, half away from zero:
          205-207, 417-424; astar.py:55-56, 111, 189-190      hybrid_astar.rs:103-108, 276-278,
                                                              529-534; astar.rs:65-72, 142-143, 233-236
   D2     rollout yaw wrapped: hybrid_astar.py:201-203        not wrapped: hybrid_astar.rs:273
   D3     heapdict with decrease-key (ties -> smallest key):  BinaryHeap of (Reverse(f), key) with
          hybrid_astar.py:78-89, 120-153                      duplicates, stale entries skipped,
                                                              ties -> largest key: hybrid_astar.rs:150-160,
                                                              190-208
   D4     steer set from np.arange: hybrid_astar.py:391-395   (i as f64) * max_steer / n_steer:
                                                              hybrid_astar.rs:496-498
   D5     M() with floored %: reeds_shepp.py:610              truncated %: reeds_shepp.rs:748
   D6     extract_path slices [:1:-1], direc[0] patched:      [1..] reversed, nothing patched:
          hybrid_astar.py:163-166, 182                        hybrid_astar.rs:227-230
   D8     calc_index on Python big ints:                      usize arithmetic, wrapping in release
          hybrid_astar.py:404-411                             builds: hybrid_astar.rs:516-520; pind is
                                                              the key as isize (:316), the parent walk
                                                              stops at pind < 0 (:226)
   D10    shot queue = heapdict keyed by RSPath objects:      BinaryHeap of (Reverse(cost), i): cost
          hybrid_astar.py:304-306                             ascending, ties -> larger i:
                                                              hybrid_astar.rs:391-397
   H      heuristic heap ties -> smallest index:              ties -> largest index: astar.rs:157-163
          astar.py:123-126, 150                               (the grid values do not depend on it)
   (D7 float -> usize saturation and D9 the signature need no patch: a negative index fails the
   twin's bounds test exactly as the saturated 0 fails the crate's, hybrid_astar.py:246 /
   hybrid_astar.rs:331.)

The twin imports two packages that are not installed here: `rerun` (drawing only) and `heapdict`
(its priority queue).  `rerun` is bound to an inert placeholder.  `heapdict` is bound to a small
priority dictionary: in group A ties cannot occur, in group B the twin's QueuePrior is replaced by
the crate's queue discipline anyway (D3, D10).
"""
import heapq as _heapq
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/path_planner_py/src"
U64 = 1 << 64


# --------------------------------------------------------------------------- import of the twin
class _PriorityDict(dict):
    """Stand-in for heapdict.heapdict: d[key] = priority; popitem() -> (key, lowest priority).
    Ties go to the entry set first (never reached by the group-A inputs)."""

    def __init__(self):
        super().__init__()
        self._h, self._n = [], 0

    def __setitem__(self, key, prio):
        super().__setitem__(key, prio)
        self._n += 1
        _heapq.heappush(self._h, (prio, self._n, key))

    def popitem(self):
        while True:
            prio, _, key = _heapq.heappop(self._h)
            if key in self and dict.__getitem__(self, key) == prio:
                dict.__delitem__(self, key)
                return key, prio


def import_twin():
    rr = types.ModuleType("rerun")
    for name in ("log", "init", "spawn", "LineStrips2D", "Points2D", "Arrows2D", "Boxes2D"):
        setattr(rr, name, lambda *a, **k: None)
    sys.modules.setdefault("rerun", rr)
    hd = types.ModuleType("heapdict")
    hd.heapdict = _PriorityDict
    sys.modules.setdefault("heapdict", hd)
    sys.path.insert(0, REF)
    import path_planner_py.astar as astar
    import path_planner_py.hybrid_astar as ha
    import path_planner_py.reeds_shepp as rs
    return rs, ha, astar


def demo_obstacles(x=51, y=31):
    """The obstacle layout of src/main.rs:46-84 (data, regenerated from its description)."""
    ox, oy = [], []
    for i in range(x):
        ox.append(float(i)); oy.append(0.0)
    for i in range(x):
        ox.append(float(i)); oy.append(float(y - 1))
    for i in range(y):
        ox.append(0.0); oy.append(float(i))
    for i in range(y):
        ox.append(float(x - 1)); oy.append(float(i))
    for i in range(10, 21):
        ox.append(float(i)); oy.append(15.0)
    for i in range(15):
        ox.append(20.0); oy.append(float(i))
    for i in range(15, 30):
        ox.append(30.0); oy.append(float(i))
    for i in range(16):
        ox.append(40.0); oy.append(float(i))
    return ox, oy


def rust_steer_set(max_steer=0.6, n_steer=20):
    """hybrid_astar.rs:495-514."""
    s = [(float(i) * max_steer) / float(n_steer) for i in range(1, n_steer)]
    steer = s + [0.0] + [-v for v in s]
    direc = [1.0] * len(steer) + [-1.0] * len(steer)
    return steer + steer, direc


def demo_conf(rs, ha):
    """src/main.rs:215-230."""
    return ha.HybridAstarConfig(2.0, math.radians(15.0), 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0,
                                15.0, rs.Car(4.5, 1.0, 3.0, 3.5), 0.6)


def far_from_tie(v, eps=1e-6):
    f = abs(v) - math.floor(abs(v))
    return abs(f - 0.5) > eps


# --------------------------------------------------------------------------- group A
def group_a(rs, ha, rng):
    from scipy.spatial import KDTree
    conf = demo_conf(rs, ha)
    ox, oy = demo_obstacles()
    P = ha.calc_parameters(ox, oy, 2.0, conf.YAW_RESO, KDTree(np.c_[ox, oy]))
    assert (P.minx, P.miny, P.minyaw, P.maxx, P.maxy) == (0, 0, -13, 25, 15)
    steer, direc = rust_steer_set()
    # the largest yaw change of a rollout: 10 steps of move_step / wheel_base * tan(max steer value)
    dyaw_max = 10 * 0.4 / 3.5 * math.tan(max(steer))
    parents = []
    while len(parents) < 160:
        x, y = float(rng.uniform(2, 49)), float(rng.uniform(2, 29))
        yaw = float(rng.uniform(-math.pi + dyaw_max + 0.01, math.pi - dyaw_max - 0.01))
        pdir = int(rng.choice([1, -1]))
        psteer = float(rng.choice(steer))
        pcost = float(rng.uniform(0, 300))
        parent = ha.Node(0, 0, 0, pdir, [x], [y], [yaw], [pdir], psteer, pcost, -1)
        prims, ok = [], True
        for u, d in zip(steer, direc):
            prims.append(ha.calc_next_node(parent, 12345, u, d, P, conf))
        # vetting needs the end pose of every primitive, also of those the twin rejects: the same
        # twin function with a Param whose bounds accept everything and whose one obstacle is far away
        Pfree = ha.Param(-10**9, -10**9, P.minyaw, 10**9, 10**9, P.xw, P.yw, 2.0, conf.YAW_RESO,
                         [1e9], [1e9], KDTree(np.array([[1e9, 1e9]])))
        for u, d in zip(steer, direc):
            n = ha.calc_next_node(parent, 12345, u, d, Pfree, conf)
            ex, ey, eyaw = n.x[-1], n.y[-1], n.yaw[-1]
            if not (far_from_tie(ex / 2.0) and far_from_tie(ey / 2.0) and
                    far_from_tie(eyaw / conf.YAW_RESO)):
                ok = False
            if not all(-math.pi < v < math.pi for v in n.yaw):
                ok = False
        if not ok:
            continue
        rec = {"x": x, "y": y, "yaw": yaw, "dir": pdir, "steer": psteer, "cost": pcost, "prims": []}
        for n in prims:
            if n is None:
                rec["prims"].append(None)
            else:
                rec["prims"].append({"xind": int(n.xind), "yind": int(n.yind), "yawind": int(n.yawind),
                                     "cost": float(n.cost), "dir": int(n.direction),
                                     "end": [float(n.x[-1]), float(n.y[-1]), float(n.yaw[-1])],
                                     "npts": len(n.x)})
        parents.append(rec)
    some = sum(1 for p in parents for n in p["prims"] if n is not None)
    json.dump({"steer": steer, "direc": direc, "parents": parents},
              open(os.path.join(HERE, "twin_next_node.json"), "w"))
    print("twin_next_node.json: %d parents, %d of %d primitives Some" % (len(parents), some, 78 * len(parents)))

    # analystic_expantion with pairwise distinct path costs
    goal_demo = (45.0, 20.0, math.radians(90.0))
    shots = []
    maxc = math.tan(conf.MAX_STEER) / conf.car.wheel_base
    tries = 0
    while len(shots) < 240:
        tries += 1
        s = (float(rng.uniform(3, 47)), float(rng.uniform(3, 27)), float(rng.uniform(-math.pi, math.pi)))
        if len(shots) % 4 == 0:
            g = goal_demo
        elif len(shots) % 4 < 3:     # nearby goals: a fair share of collision-free shots
            g = (min(47.0, max(3.0, s[0] + float(rng.uniform(-9, 9)))),
                 min(27.0, max(3.0, s[1] + float(rng.uniform(-9, 9)))), float(rng.uniform(-math.pi, math.pi)))
        else:
            g = (float(rng.uniform(3, 47)), float(rng.uniform(3, 27)), float(rng.uniform(-math.pi, math.pi)))
        paths = rs.calc_all_paths(*s, *g, maxc, conf.MOVE_STEP)
        costs = sorted(float(ha.calc_rs_path_cost(p, conf)) for p in paths)
        if any(b - a <= 1e-6 for a, b in zip(costs, costs[1:])):
            continue
        node = ha.Node(0, 0, 0, 1, [s[0]], [s[1]], [s[2]], [1], 0.0, 0.0, -1)
        ngoal = ha.Node(0, 0, 0, 1, [g[0]], [g[1]], [g[2]], [1], 0.0, 0.0, -1)
        got = ha.analystic_expantion(node, ngoal, P, conf)
        rec = {"s": list(s), "g": list(g), "n_paths": len(paths), "found": got is not None}
        if got is not None:
            # index of the returned path in calc_all_paths order: the twin returns the object of
            # its own second calc_all_paths call, so match by word and lengths
            again = rs.calc_all_paths(*s, *g, maxc, conf.MOVE_STEP)
            idx = [i for i, p in enumerate(again) if p.ctypes == got.ctypes and p.move_length == got.move_length]
            assert len(idx) == 1
            rec.update({"idx": idx[0], "ctypes": "".join(got.ctypes),
                        "lengths": [float(v) for v in got.move_length],
                        "cost": float(ha.calc_rs_path_cost(got, conf)), "npts": len(got.traj_x)})
        shots.append(rec)
    nf = sum(1 for r in shots if r["found"])
    json.dump({"shots": shots}, open(os.path.join(HERE, "twin_analytic.json"), "w"))
    print("twin_analytic.json: %d shots (%d found a path), %d candidates drawn" % (len(shots), nf, tries))


# --------------------------------------------------------------------------- group B patches
def round_half_away(v):
    """f64::round: to nearest, halves away from zero (v - trunc(v) is exact in binary64)."""
    v = float(v)
    if not math.isfinite(v):
        return v
    t = float(math.trunc(v))
    if abs(v - t) >= 0.5:
        t += math.copysign(1.0, v)
    return t


class NpHalfAway:
    """numpy with `round` replaced (D1); everything else the twin uses passes through."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def round(v):
        return round_half_away(v)


class HeapLargestIndexFirst:
    """heapq for (cost, index) entries with ties -> the larger index (H)."""

    @staticmethod
    def heappush(q, item):
        _heapq.heappush(q, (item[0], -item[1]))

    @staticmethod
    def heappop(q):
        c, i = _heapq.heappop(q)
        return c, -i


class Counters:
    def __init__(self):
        self.reset()

    def reset(self):
        self.closed, self.stale, self.rollouts, self.poses, self.rs_paths = [], 0, 0, 0, 0
        self.final = None


class LimitReached(Exception):
    pass


def apply_rust_semantics(rs, ha, astar, K, limit):
    # D1
    ha.np = NpHalfAway()
    astar.np = NpHalfAway()
    # H
    astar.heapq = HeapLargestIndexFirst()

    # D5
    def m_trunc(theta):
        phi = math.fmod(theta, 2.0 * math.pi)
        if phi < -math.pi:
            phi += 2.0 * math.pi
        if phi > math.pi:
            phi -= 2.0 * math.pi
        return phi
    rs.M = m_trunc

    # D4
    ha.calc_motion_set = lambda conf: rust_steer_set(conf.MAX_STEER, conf.N_STEER)

    # D8
    def calc_index_u64(node, P):
        a = (node.yawind - P.minyaw) % U64
        ind = (a * P.xw) % U64
        ind = (ind * P.yw) % U64
        ind = (ind + ((node.yind - P.miny) % U64) * P.xw) % U64
        return (ind + (node.xind - P.minx)) % U64
    ha.calc_index = calc_index_u64

    # D3 + D10: the crate's queue discipline in place of the twin's QueuePrior
    class RustQueue:
        def __init__(self):
            self.h, self.live, self.n = [], set(), 0

        def empty(self):
            return len(self.live) == 0

        def put(self, item, priority):
            if isinstance(item, int):        # open list: (Reverse(f), key), duplicates allowed
                tie = item
            else:                            # shot queue: (Reverse(cost), i), i = insertion order
                tie = self.n
                self.n += 1
            self.live.add(id(item) if not isinstance(item, int) else item)
            _heapq.heappush(self.h, (priority, -tie, self.n if isinstance(item, int) else 0, item))

        def get(self):
            while True:
                _, _, _, item = _heapq.heappop(self.h)
                key = item if isinstance(item, int) else id(item)
                if key in self.live:
                    self.live.discard(key)
                    return item
                K.stale += 1                 # hybrid_astar.rs:160: the entry's node is gone
    ha.QueuePrior = RustQueue

    # D2 (and the rollout counter)
    twin_next = ha.calc_next_node
    twin_wrap = ha.pi_2_pi

    def next_node_unwrapped(n_curr, c_id, u, d, P, conf):
        K.rollouts += 1
        ha.pi_2_pi = lambda th: th
        try:
            return twin_next(n_curr, c_id, u, d, P, conf)
        finally:
            ha.pi_2_pi = twin_wrap
    ha.calc_next_node = next_node_unwrapped

    # counters on the seams (no arithmetic replaced)
    twin_coll = ha.is_collision

    def counted_collision(x, y, yaw, P, conf):
        K.poses += len(x)
        return twin_coll(x, y, yaw, P, conf)
    ha.is_collision = counted_collision

    twin_all = ha.calc_all_paths

    def counted_all_paths(*a, **k):
        paths = twin_all(*a, **k)
        K.rs_paths += len(paths)
        return paths
    ha.calc_all_paths = counted_all_paths

    twin_update = ha.update_node_with_analystic_expantion

    def counted_update(n_curr, ngoal, P, conf):
        K.closed.append(ha.calc_index(n_curr, P))
        if len(K.closed) > limit:
            raise LimitReached()
        f = twin_update(n_curr, ngoal, P, conf)
        if f:
            K.final = f
        return f
    ha.update_node_with_analystic_expantion = counted_update

    # D6 + D8
    def extract_path_rust(closed, ngoal, nstart):
        rx, ry, ryaw, direc = [], [], [], []
        cost = 0.0
        node = ngoal

        def as_isize(v):
            return v - U64 if v >= (1 << 63) else v
        while as_isize(node.pind) >= 0:
            rx += node.x[1:][::-1]
            ry += node.y[1:][::-1]
            ryaw += node.yaw[1:][::-1]
            direc += list(node.directions[1:][::-1])
            cost += node.cost
            node = closed[node.pind]
        rx += nstart.x[::-1]
        ry += nstart.y[::-1]
        ryaw += nstart.yaw[::-1]
        direc += nstart.directions[::-1]
        cost += nstart.cost
        return ha.HybridAstarPath(rx[::-1], ry[::-1], ryaw[::-1], direc[::-1], cost)
    ha.extract_path = extract_path_rust


def group_b(rs, ha, astar, rng):
    from scipy.spatial import KDTree
    K = Counters()
    LIMIT = 400
    apply_rust_semantics(rs, ha, astar, K, LIMIT)
    conf = demo_conf(rs, ha)
    ox, oy = demo_obstacles()
    P = ha.calc_parameters(ox, oy, 2.0, conf.YAW_RESO, KDTree(np.c_[ox, oy]))

    class G:
        def __init__(self, x, y):
            self.x, self.y = [x], [y]
    hm = np.array(astar.calc_holonomic_heuristic_with_obstacle(G(45.0, 20.0), ox, oy, 2.0, 1.0))
    fin = np.isfinite(hm)
    out = {"param": [P.minx, P.miny, P.minyaw, P.maxx, P.maxy, P.xw, P.yw],
           "hmap_demo_goal": {"finite": int(fin.sum()), "max": float(hm[fin].max()),
                              "sum": float(math.fsum(hm[fin])),
                              "rows": [[(float(v) if math.isfinite(v) else "inf") for v in row] for row in hm]},
           "limit": LIMIT, "queries": []}

    def free(x, y, yaw):
        Pq = ha.Param(0, 0, 0, 0, 0, 0, 0, 2.0, 0.0, ox, oy, KDTree(np.c_[ox, oy]))
        K.poses = 0
        return not ha.is_collision([x], [y], [yaw], Pq, conf)

    queries = [((10.0, 7.0, math.radians(120.0)), (45.0, 20.0, math.radians(90.0)))]   # src/main.rs:233-234
    while len(queries) < 25:
        s = (float(rng.uniform(4, 46)), float(rng.uniform(4, 26)), float(rng.uniform(-math.pi, math.pi)))
        g = (float(rng.uniform(4, 46)), float(rng.uniform(4, 26)), float(rng.uniform(-math.pi, math.pi)))
        if free(*s) and free(*g):
            queries.append((s, g))
    for qi, (s, g) in enumerate(queries):
        K.reset()
        try:
            path = ha.hybrid_astar_planning(*s, *g, ox, oy, 2.0, conf.YAW_RESO, conf)
            status = "ok" if path is not None else "none"
        except LimitReached:
            path, status = None, "limit"
            K.closed = K.closed[:LIMIT]
        rec = {"start": list(s), "goal": list(g), "status": status,
               "closed": [int(v) for v in K.closed], "stale_pops": K.stale, "rollouts": K.rollouts,
               "poses_checked": K.poses, "rs_paths": K.rs_paths}
        if status == "ok":
            f = K.final
            rec.update({"cost": float(path.cost), "npts": len(path.x),
                        "x": [float(v) for v in path.x], "y": [float(v) for v in path.y],
                        "yaw": [float(v) for v in path.yaw], "dir": [int(v) for v in path.direction],
                        "final_npts": len(f.x), "final_cost": float(f.cost)})
        out["queries"].append(rec)
        print("query %2d: %-5s closed %3d stale %3d rollouts %5d poses %6d rs_paths %4d%s" % (
            qi, status, len(rec["closed"]), K.stale, K.rollouts, K.poses, K.rs_paths,
            "  cost %.10f npts %d" % (rec["cost"], rec["npts"]) if status == "ok" else ""), flush=True)
    json.dump(out, open(os.path.join(HERE, "rust_semantics.json"), "w"))


# --------------------------------------------------------------------------- group C
def astar_env():
    """The obstacle layout of src/main.rs:14-44 (data, regenerated from its description)."""
    ox, oy = [], []
    for i in range(60):
        ox.append(float(i)); oy.append(0.0)
    for i in range(60):
        ox.append(60.0); oy.append(float(i))
    for i in range(61):
        ox.append(float(i)); oy.append(60.0)
    for i in range(61):
        ox.append(0.0); oy.append(float(i))
    for i in range(40):
        ox.append(20.0); oy.append(float(i))
    for i in range(40):
        ox.append(40.0); oy.append(60.0 - i)
    return ox, oy


def group_c(astar, rng):
    """astar_rust.json: the plain grid planner, astar_planning (astar.py:32-104 == astar.rs:53-132
    once patches D1 and H are in: rounding of the start/goal/grid bounds, and the heap's tie rule
    -- lowest f, then the LARGEST calc_index).  The closing order is read off the cells the twin
    hands to its drawing call every iteration (astar.py:75-76)."""
    seen = []
    astar.rr.Points2D = lambda arr: seen.append(np.array(arr, copy=True))
    out = []

    def run(ox, oy, sx, sy, gx, gy, reso, rr_, keep_obstacles):
        del seen[:]
        res = astar.astar_planning(sx, sy, gx, gy, ox, oy, reso, rr_)
        P, _ = astar.calc_parameters([x / reso for x in ox], [y / reso for y in oy], rr_, reso)
        cells = seen[-1] if seen else np.zeros((0, 2))
        order = [int((int(c[1]) - P.miny) * P.xw + (int(c[0]) - P.minx)) for c in cells]
        rec = {"s": [sx, sy], "g": [gx, gy], "reso": reso, "rr": rr_, "closed": order,
               "path": None if res is None else [[float(v) for v in res[0]], [float(v) for v in res[1]]]}
        if keep_obstacles:
            rec["ox"], rec["oy"] = [float(v) for v in ox], [float(v) for v in oy]
        out.append(rec)
        return rec

    ox, oy = astar_env()
    r = run(ox, oy, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0, False)      # src/main.rs:157-171
    r["map"] = "astar_env"
    r = run(ox, oy, 30.0, 30.0, 30.0, 30.0, 1.0, 2.0, False)      # start == goal
    r["map"] = "astar_env"
    r = run(ox, oy, 10.0, 10.0, 20.0, 20.0, 1.0, 2.0, False)      # goal inside an obstacle disc: None
    r["map"] = "astar_env"
    for trial in range(12):
        n = int(rng.integers(20, 60))
        reso = [1.0, 2.0, 0.5][trial % 3]
        pts = [(x, y) for x in range(n + 1) for y in range(n + 1)
               if x in (0, n) or y in (0, n) or rng.random() < 0.18]
        mox, moy = [p[0] * reso for p in pts], [p[1] * reso for p in pts]
        s = rng.uniform(1, n - 1, 2) * reso
        g = rng.uniform(1, n - 1, 2) * reso
        run(mox, moy, float(s[0]), float(s[1]), float(g[0]), float(g[1]), reso, 0.4 * reso, True)
    json.dump(out, open(os.path.join(HERE, "astar_rust.json"), "w"))
    print("astar_rust.json: %d queries, %d with a path, closed cells %s" % (
        len(out), sum(1 for r in out if r["path"]), [len(r["closed"]) for r in out]))


def main():
    rs, ha, astar = import_twin()
    rng = np.random.default_rng(20251111)
    group_a(rs, ha, rng)
    group_b(rs, ha, astar, rng)
    group_c(astar, np.random.default_rng(20251112))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
