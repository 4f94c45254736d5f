"""Plain grid A* (astar.rs:53-132): the demo query of src/main.rs:157-191 and a batch on a large
map through pp_astar_plan(_batch), next to the CPU oracle (whose raster uses the same scatter; the
reference's own raster loop, oracle mode 0, is timed separately on the demo map)."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as O
from path_planner_amd import api, workloads as W
from test_oracle_astar import astar_env

def best(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3, r

ctx = api.Context(0)
ox, oy = astar_env()
m = api.ObstacleMap(ctx, ox, oy)
t, p = best(lambda: api.astar_planning(ctx, m, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0, cap=4096))
to, (st, po, order) = best(lambda: O.astar(ox, oy, 10.0, 10.0, 50.0, 50.0, 1.0, 2.0))
t0, _ = best(lambda: O.obsmap(ox, oy, 1.0, 2.0, mode=0))
print("demo query: device raster + host search %.3f ms; oracle %.3f ms (%d cells closed, %d path points, same path: %s); "
      "reference-style raster loop alone %.3f ms" % (t, to, len(order), len(po[0]), bool((p[0] == po[0]).all()), t0))

n = 2048
occ = W.occupancy_grid(n, 0.1, seed=4242)
oxb, oyb = W.raster_to_points(occ)
mb = api.ObstacleMap(ctx, oxb, oyb)
rng = np.random.default_rng(1)
starts, goals = rng.uniform(1, n - 1, (64, 2)), rng.uniform(1, n - 1, (64, 2))
t_r, _ = best(lambda: api.calc_obsmap(ctx, mb, 1.0, 0.45), 3)
t_b, (paths, ex) = best(lambda: api.astar_planning_batch(ctx, mb, starts, goals, 1.0, 0.45), 2)
t_o0 = time.perf_counter(); so = O.astar(oxb, oyb, *starts[0], *goals[0], 1.0, 0.45, cap=100000, closed_cap=1); t_o = (time.perf_counter() - t_o0) * 1e3
print("%dx%d map, %d obstacles: raster (device, incl. copy back) %.2f ms; 64 queries %.1f ms (%d found, %.2f M cells closed); "
      "oracle, query 0 alone (raster + search, 1 core): %.1f ms" %
      (n, n, len(oxb), t_r, t_b, sum(p is not None for p in paths), ex.sum() / 1e6, t_o))
