"""Development: how search-heavy the many-queries workload gets with the obstacle density."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_planner_amd import _abi as A, api, workloads as W
ctx = api.Context(0)
conf = A.demo_config()
size, nq = 4096, int(os.environ.get("NQ", "1024"))
cap = int(os.environ.get("CAP", "3000"))
ctx.set_max_expansions(cap)
for dens in [float(v) for v in os.environ.get("DENS", "0.006 0.015 0.03 0.05").split()]:
    for dist in ((30.0, 200.0), (100.0, 400.0)):
        occ = W.plan_map(size, dens)
        m = api.ObstacleMap(ctx, raster=occ)
        cs, cg = W.plan_queries(6 * nq, size, 64, min_dist=dist[0], max_dist=dist[1])
        inside = (cs[:, 0] > 8) & (cs[:, 0] < size - 9) & (cs[:, 1] > 8) & (cs[:, 1] < size - 9)
        fs = api.collision_flags(ctx, m, conf, cs[:, 0], cs[:, 1], cs[:, 2])
        fg = api.collision_flags(ctx, m, conf, cg[:, 0], cg[:, 1], cg[:, 2])
        keep = np.nonzero(inside & (fs == 0) & (fg == 0))[0][:nq]
        if len(keep) < nq:
            print("density", dens, "only", len(keep), "free candidates"); continue
        r = api.plan_batch(ctx, m, conf, cs[keep], cg[keep], cap=8192)
        t0 = time.perf_counter()
        r = api.plan_batch(ctx, m, conf, cs[keep], cg[keep], cap=8192)
        dt = time.perf_counter() - t0
        ex = np.array([int(s.expansions) for s in r["stats"]])
        st = r["status"]
        print("density %.3f (%.4f) dist %s: %.1f ms  found %d  none %d  cap %d  other %d | expansions median %d mean %.1f p90 %d max %d" % (
            dens, occ.mean(), dist, dt * 1e3, (st == 0).sum(), (st == 1).sum(), (st == A.PP_ERR_LIMIT).sum(),
            ((st < 0) & (st != A.PP_ERR_LIMIT)).sum(), np.median(ex), ex.mean(), np.percentile(ex, 90), ex.max()), flush=True)
        m.close()
