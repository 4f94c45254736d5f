"""Phase times of the planner's round kernel for the demo query (experiments only).

Needs a library built with -DPP_LAB_STAMPS=1 on pp_rs.hip (SRC=pp_rs tools/variant.sh stamps
-DPP_LAB_STAMPS=1) and PP_HIP_LIB pointing at it: workgroup 0 of k_round records the 100 MHz wall
clock at its phase boundaries and pp_lab_stamps() copies the stamps out."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from path_planner_amd import _abi as A, api, workloads as W
ctx = api.Context(0)
ox, oy = W.demo_obstacles()
m = api.ObstacleMap(ctx, ox, oy)
(sx, sy, syaw), (gx, gy, gyaw) = W.demo_query()
conf = A.demo_config()
lib = ctx.lib
lib.pp_lab_stamps.argtypes = [C.POINTER(C.c_longlong)]
for rep in range(4):
    r = api.hybrid_astar_planning(ctx, m, conf, sx, sy, syaw, gx, gy, gyaw)
    buf = (C.c_longlong * 32)()
    lib.pp_lab_stamps(buf)
    t = np.array(buf[:8], dtype=np.int64)
    print("us since kernel start (100 MHz clock):", [(int(v - t[0])) / 100.0 for v in t[1:8]])
    print("   formula waves 0..11 done at:", [(int(v - t[0])) / 100.0 for v in buf[16:28]])
