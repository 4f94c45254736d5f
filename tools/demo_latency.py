"""Latency of the reference's demo query (src/main.rs:215-269) through pp_plan, next to the CPU oracle."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as O
from path_planner_amd import _abi as A, api, workloads as W
ctx = api.Context(0); conf = A.demo_config()
ox, oy = W.demo_obstacles(); (sx, sy, syaw), (gx, gy, gyaw) = W.demo_query()
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    m = api.ObstacleMap(ctx, ox, oy)
    r = api.hybrid_astar_planning(ctx, m, conf, sx, sy, syaw, gx, gy, gyaw)
    ts.append(time.perf_counter() - t0); m.close()
print("device: map build + plan %.2f ms (best of 5 after warm-up); heuristic %.2f ms, search %.2f ms, cost %.6f, %d points"
      % (min(ts[1:]) * 1e3, r["stats"].t_heuristic_ms, r["stats"].t_search_ms, r["cost"], len(r["x"])))
tc = []
for _ in range(5):
    t0 = time.perf_counter()
    om = O.OracleMap(ox, oy); e = om.plan(conf, [sx, sy, syaw], [gx, gy, gyaw])
    tc.append(time.perf_counter() - t0)
print("oracle (1 core): %.2f ms, cost %.6f, %d points" % (min(tc) * 1e3, e["cost"], len(e["x"])))
