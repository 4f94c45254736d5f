import time, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from path_planner_amd import _abi as A, api, workloads as W
ctx = api.Context(0); conf = A.demo_config()
occ = W.plan_map(4096); omap = api.ObstacleMap(ctx, raster=occ); ctx.set_max_expansions(3000)
cs, cg = W.plan_queries(3*4096, 4096, 64)
inside = (cs[:,0]>8)&(cs[:,0]<4087)&(cs[:,1]>8)&(cs[:,1]<4087)
fs = api.collision_flags(ctx, omap, conf, cs[:,0], cs[:,1], cs[:,2]); fg = api.collision_flags(ctx, omap, conf, cg[:,0], cg[:,1], cg[:,2])
keep = np.nonzero(inside&(fs==0)&(fg==0))[0][:4096]
starts, goals = cs[keep], cg[keep]
for rep in range(3):
    t0=time.perf_counter(); r = api.plan_batch(ctx, omap, conf, starts, goals, cap=4096); t1=time.perf_counter()
    st=r["stats"]
    ex=np.array([st[i].expansions for i in range(4096)])
    print("total %.1f ms heur %.1f search %.1f  max exp %d  mean %.1f  >100: %d  >300: %d" % ((t1-t0)*1e3, st[0].t_heuristic_ms, st[0].t_search_ms, ex.max(), ex.mean(), (ex>100).sum(), (ex>300).sum()))
