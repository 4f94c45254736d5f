"""PCIe-inclusive rate of the host-pointer collision entry point (pp_collision_batch): the same
16 M poses handed over as host arrays, copies included.  Not the bench value."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from path_planner_amd import _abi as A, api, workloads as W
ctx = api.Context(0)
conf = A.demo_config()
omap = api.ObstacleMap(ctx, raster=W.occupancy_grid(2048, 0.05))
n = 1 << 24
x, y, yaw = W.random_poses(n, 2048)
api.collision_flags(ctx, omap, conf, x[:1000], y[:1000], yaw[:1000])
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    f = api.collision_flags(ctx, omap, conf, x, y, yaw)
    ts.append(time.perf_counter() - t0)
t = min(ts[1:])
print(f"pp_collision_batch host path: {t*1e3:.1f} ms for {n} poses = {n/t/1e9:.2f} G poses/s (pageable host arrays)")
