"""The reference's three demos (src/main.rs) driven from compiled C++ through the mirror of its
function names (include/path_planner.hpp) on top of the C ABI; results against the oracle and the
golden fixtures."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import workloads as W
from test_oracle_astar import astar_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "main_demo")


def test_main_demo_matches_oracle_and_goldens():
    assert os.path.exists(EXE), "tests/cpp/main_demo is built by __graft_entry__.build()"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    # Reeds-Shepp legs: the twin's fixtures
    legs = json.load(open(os.path.join(ROOT, "tests", "golden", "rs_demo_legs.json")))
    assert len(r["rs_total"]) == 7
    for got, leg in zip(r["rs_total"], legs):
        best = leg["paths"][leg["best"]]
        assert abs(got[0] - best["total"]) <= 1e-9 * best["total"]
        assert got[1] == best["npts"] and got[2] == best["ctypes"]
    # grid A*: the oracle
    ox, oy = astar_env()
    st, path, order = O.astar(ox, oy, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0)
    assert r["astar_len"] == len(path[0])
    assert r["astar_first"] == [path[0][0], path[1][0]] and r["astar_last"] == [path[0][-1], path[1][-1]]
    # hybrid A*: the oracle's planner on the demo query
    conf = A.demo_config()
    dox, doy = W.demo_obstacles()
    (sx, sy, syaw), (gx, gy, gyaw) = W.demo_query()
    e = O.OracleMap(dox, doy).plan(conf, [sx, sy, syaw], [gx, gy, gyaw])
    assert r["hybrid_points"] == len(e["x"]) == 184
    assert abs(r["hybrid_cost"] - e["cost"]) <= 1e-9 * e["cost"]
    st, hm, dims = O.heuristic(dox, doy, 45.0, 20.0, 2.0, 1.0)
    assert r["hmap_dims"] == [int(dims[4]), int(dims[5])] and r["hmap_start"] == hm[5, 4]
    assert r["collides_at_wall"] == 1 and r["collides_at_start"] == 0
    assert r["panics_on_goal_outside"] == 1
