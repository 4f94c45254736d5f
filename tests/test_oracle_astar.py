"""CPU checks of the oracle's plain grid A* (astar_planning, astar.rs:53-132).

Pinned by tests/golden/astar_rust.json: closing order and path from the reference's Python twin
with its rounding and its heap's tie rule patched to the crate's (tests/golden/make_rust_semantics.py,
group C).  The pure-Python loop below is a second restatement used on further random maps.  The
raster and grid dimensions underneath are pinned by the twin's fixtures (test_oracle_golden.py)."""
import json
import os
import heapq
import math

import numpy as np

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import workloads as W


def rust_round(v):
    return math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)


def py_astar(ox, oy, sx, sy, gx, gy, reso, rr):
    """astar.rs:53-132 with dicts and heapq; keys (f, -index) pop lowest f, then largest index."""
    st, obs, dims = O.obsmap(ox, oy, reso, rr)
    minx, miny, maxx, maxy, xw, yw = (int(v) for v in dims)
    s = (int(rust_round(sx / reso)), int(rust_round(sy / reso)))
    g = (int(rust_round(gx / reso)), int(rust_round(gy / reso)))
    index = lambda x, y: (y - miny) * xw + (x - minx)
    motion = [(-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)]
    open_set = {index(*s): [s[0], s[1], 0.0, -1]}
    closed, order = {}, []
    heap = [(math.hypot(s[0] - g[0], s[1] - g[1]), -index(*s))]
    while open_set:
        _, negk = heapq.heappop(heap)
        ind = -negk
        cur = open_set.pop(ind)
        closed[ind] = cur
        order.append(ind)
        if ind == index(*g):
            px, py = [float(g[0])], [float(g[1])]
            k = ind
            while k in closed:
                nd = closed[k]
                px.append(float(nd[0])); py.append(float(nd[1]))
                k = nd[3]
                if nd == [s[0], s[1], 0.0, -1]:
                    break
            return ([v * reso for v in reversed(px)], [v * reso for v in reversed(py)]), order
        for dx, dy in motion:
            x, y, c = cur[0] + dx, cur[1] + dy, cur[2] + math.hypot(dx, dy)
            if x <= minx or x >= maxx or y <= miny or y >= maxy or obs[x - minx, y - miny]:
                continue
            k = index(x, y)
            if k in closed:
                continue
            if k in open_set:
                if open_set[k][2] > c:
                    open_set[k][2], open_set[k][3] = c, ind
            else:
                open_set[k] = [x, y, c, ind]
                heapq.heappush(heap, (c + math.hypot(x - g[0], y - g[1]), -k))
    return None, order


def astar_env():
    """Obstacle layout of src/main.rs:14-44 (data, regenerated from its description)."""
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


def check_path(path, ox, oy, s, g, reso, rr):
    px, py = path
    st, obs, dims = O.obsmap(ox, oy, reso, rr)
    cx = np.round(np.asarray(px) / reso).astype(int)
    cy = np.round(np.asarray(py) / reso).astype(int)
    assert (cx[0], cy[0]) == s and (cx[-1], cy[-1]) == g and (cx[-2], cy[-2]) == g  # goal twice
    assert (np.abs(np.diff(cx[:-1])) <= 1).all() and (np.abs(np.diff(cy[:-1])) <= 1).all()
    assert not obs[cx[1:] - int(dims[0]), cy[1:] - int(dims[1])].any()
    return float(np.hypot(np.diff(cx[:-1]), np.diff(cy[:-1])).sum())


def check_astar_fixture(plan):
    """tests/golden/astar_rust.json against plan(ox, oy, sx, sy, gx, gy, reso, rr) ->
    (status, (pathx, pathy) or None, closing order)."""
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "astar_rust.json")))
    env = astar_env()
    for i, q in enumerate(data):
        ox, oy = env if q.get("map") == "astar_env" else (q["ox"], q["oy"])
        st, path, order = plan(ox, oy, *q["s"], *q["g"], q["reso"], q["rr"])
        assert list(order) == q["closed"], i
        if q["path"] is None:
            assert st == A.PP_NO_PATH and path is None, i
        else:
            assert st == 0 and list(path[0]) == q["path"][0] and list(path[1]) == q["path"][1], i
    assert sum(1 for q in data if q["path"] is None) >= 2 and len(data) >= 15


def test_against_the_patched_twin_fixture():
    check_astar_fixture(O.astar)


def test_demo_environment():
    ox, oy = astar_env()
    st, path, order = O.astar(ox, oy, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0)  # main.rs:157-171
    assert st == 0
    length = check_path(path, ox, oy, (10, 10), (50, 50), 1.0, 2.0)
    ref_path, ref_order = py_astar(ox, oy, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0)
    assert list(order) == ref_order
    assert list(path[0]) == ref_path[0] and list(path[1]) == ref_path[1]
    # never shorter than the true optimum (the holonomic heuristic from the goal)
    st, hm, dims = O.heuristic(ox, oy, 50.0, 50.0, 1.0, 2.0)
    assert length >= hm[10 - int(dims[0]), 10 - int(dims[1])] - 1e-9


def test_random_maps_against_the_python_restatement():
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(20, 70))
        reso = [1.0, 2.0, 0.5][trial % 3]
        ox, oy = W.raster_to_points(W.occupancy_grid(n, 0.18, seed=100 + trial))
        ox, oy = ox * reso, oy * reso
        s = rng.uniform(1, n - 1, 2) * reso
        g = rng.uniform(1, n - 1, 2) * reso
        st, path, order = O.astar(ox, oy, s[0], s[1], g[0], g[1], reso, 0.4 * reso)
        ref_path, ref_order = py_astar(ox, oy, s[0], s[1], g[0], g[1], reso, 0.4 * reso)
        assert list(order) == ref_order, trial
        if ref_path is None:
            assert st == A.PP_NO_PATH and path is None
        else:
            assert st == 0 and list(path[0]) == ref_path[0] and list(path[1]) == ref_path[1]


def test_edge_cases():
    ox, oy = astar_env()
    # start == goal: the start is closed first and is the goal (astar.rs:101): goal, start
    st, path, order = O.astar(ox, oy, 30.0, 30.0, 30.0, 30.0, 1.0, 2.0)
    assert st == 0 and list(path[0]) == [30.0, 30.0] and len(order) == 1
    # goal inside an obstacle disc: never pushed, the reachable region is exhausted -> None
    st, path, order = O.astar(ox, oy, 10.0, 10.0, 20.0, 20.0, 1.0, 2.0)
    assert st == A.PP_NO_PATH and path is None and len(order) > 100
    # start outside the grid: inserted unchecked (astar.rs:84); its neighbours all fail check_node
    st, path, order = O.astar(ox, oy, 100.0, 100.0, 50.0, 50.0, 1.0, 2.0)
    assert st == A.PP_NO_PATH and len(order) == 1
