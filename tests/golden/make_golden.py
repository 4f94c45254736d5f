#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference's Python twin.

Run in the build container only (needs /root/reference; the GPU box has no
reference):  python tests/golden/make_golden.py

The twin (path_planner_py/src/path_planner_py) imports two packages that are
not installed here: `rerun` (a visualisation SDK, used only by drawing code
that none of the functions below reach) and `heapdict` (the open list of the
twin's own search loop, which is NOT used for any fixture because its
tie-breaking differs from the Rust crate's BinaryHeap).  Both names are bound to
inert placeholder modules so that `import` succeeds; no arithmetic of the twin
is replaced.  Only functions whose semantics coincide with the Rust source are
sampled:

  reeds_shepp.calc_all_paths / calc_optimal_path      (src/reeds_shepp.rs:60-128)
  hybrid_astar.is_collision (per pose)                (src/hybrid_astar.rs:417-448)
  hybrid_astar.calc_rs_path_cost                      (src/hybrid_astar.rs:450-489)
  astar.calc_obsmap / calc_holonomic_heuristic_...    (src/astar.rs:134-261)

Inputs are chosen so the known twin-vs-crate differences do not bite: no
coordinate/resolution quotient lands on a .5 tie (numpy rounds half-to-even,
Rust half-away), and the twin's floored `%` in M() differs from Rust's truncated
`%` only by < 1e-10 relative in lengths (tests use 1e-9).
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/path_planner_py/src"


def _import_twin():
    rr = types.ModuleType("rerun")
    for name in ("log", "init", "spawn", "LineStrips2D", "Points2D", "Arrows2D", "Boxes2D"):
        setattr(rr, name, lambda *a, **k: None)
    sys.modules.setdefault("rerun", rr)
    hd = types.ModuleType("heapdict")

    class heapdict(dict):  # never exercised: the twin's search loop is not sampled
        def popitem(self):
            raise NotImplementedError("placeholder")

    hd.heapdict = heapdict
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


def paths_to_json(paths, with_traj):
    out = []
    for p in paths:
        d = {
            "ctypes": "".join(p.ctypes),
            "lengths": [float(v) for v in p.move_length],
            "total": float(p.total_length),
            "npts": len(p.traj_x),
        }
        if with_traj:
            d["x"] = [float(v) for v in p.traj_x]
            d["y"] = [float(v) for v in p.traj_y]
            d["yaw"] = [float(v) for v in p.traj_yaw]
            d["dir"] = [int(v) for v in p.traj_dirs]
        else:
            d["sum_x"] = float(math.fsum(p.traj_x))
            d["sum_y"] = float(math.fsum(p.traj_y))
            d["sum_yaw"] = float(math.fsum(p.traj_yaw))
            d["last"] = [float(p.traj_x[-1]), float(p.traj_y[-1]), float(p.traj_yaw[-1])]
            d["ndir_fwd"] = int(sum(1 for v in p.traj_dirs if v > 0))
        out.append(d)
    return out


def main():
    rs, ha, astar = _import_twin()
    rng = np.random.default_rng(20250711)

    # ---- (i) the seven Reeds-Shepp demo legs of src/main.rs:93-137 (maxc 0.1, step 0.2)
    states = [
        (0.0, 0.0, 0.0), (10.0, 10.0, -90.0), (20.0, 5.0, 60.0), (30.0, 10.0, 120.0),
        (35.0, -5.0, 30.0), (25.0, -10.0, -120.0), (15.0, -15.0, 100.0), (0.0, -10.0, -90.0),
    ]
    legs = []
    for a, b in zip(states[:-1], states[1:]):
        q0 = (a[0], a[1], math.radians(a[2]))
        q1 = (b[0], b[1], math.radians(b[2]))
        paths = rs.calc_all_paths(*q0, *q1, 0.1, 0.2)
        best = rs.calc_optimal_path(*q0, *q1, 0.1, 0.2)
        bi = max(i for i, p in enumerate(paths) if p.total_length == best.total_length
                 and p.ctypes == best.ctypes)
        legs.append({"q0": list(q0), "q1": list(q1), "maxc": 0.1, "step": 0.2,
                     "best": bi, "paths": paths_to_json(paths, True)})
    json.dump(legs, open(os.path.join(HERE, "rs_demo_legs.json"), "w"))

    # ---- (ii) random pairs at the hybrid A* demo curvature
    maxc = math.tan(0.6) / 3.5
    conf = ha.HybridAstarConfig(2.0, math.radians(15.0), 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0,
                                15.0, rs.Car(4.5, 1.0, 3.0, 3.5), 0.6)
    pairs = []
    for k in range(400):
        if k < 300:
            q0 = (rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(-math.pi, math.pi))
            q1 = (rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(-math.pi, math.pi))
        else:  # short-range pairs exercise the CCC / CCCC families
            q0 = (rng.uniform(0, 100), rng.uniform(0, 100), rng.uniform(-math.pi, math.pi))
            q1 = (q0[0] + rng.uniform(-8, 8), q0[1] + rng.uniform(-8, 8),
                  rng.uniform(-math.pi, math.pi))
        paths = rs.calc_all_paths(*q0, *q1, maxc, 0.4)
        costs = [float(ha.calc_rs_path_cost(p, conf)) for p in paths]
        best = rs.calc_optimal_path(*q0, *q1, maxc, 0.4)
        bi = max(i for i, p in enumerate(paths) if p.total_length == best.total_length
                 and p.ctypes == best.ctypes)
        pairs.append({"q0": [float(v) for v in q0], "q1": [float(v) for v in q1], "best": bi,
                      "costs": costs, "paths": paths_to_json(paths, k % 40 == 0)})
    json.dump({"maxc": maxc, "step": 0.4, "pairs": pairs},
              open(os.path.join(HERE, "rs_random.json"), "w"))

    # ---- (iii) per-pose collision flags: demo map, and a random NON-lattice point set
    from scipy.spatial import KDTree

    def flags_for(ox, oy, poses):
        P = ha.Param(0, 0, 0, 0, 0, 0, 0, 2.0, 0.0, ox, oy, KDTree(np.c_[ox, oy]))
        return [int(bool(ha.is_collision([x], [y], [yaw], P, conf))) for x, y, yaw in poses]

    ox, oy = demo_obstacles()
    poses = [(float(rng.uniform(-2, 53)), float(rng.uniform(-2, 33)),
              float(rng.uniform(-math.pi, math.pi))) for _ in range(3000)]
    col = {"demo": {"poses": poses, "flags": flags_for(ox, oy, poses)}}
    rox = [float(v) for v in rng.uniform(0, 60, 60)]
    roy = [float(v) for v in rng.uniform(0, 60, 60)]
    poses2 = [(float(rng.uniform(0, 60)), float(rng.uniform(0, 60)),
               float(rng.uniform(-4 * math.pi, 4 * math.pi))) for _ in range(3000)]
    col["scatter"] = {"ox": rox, "oy": roy, "poses": poses2, "flags": flags_for(rox, roy, poses2)}
    json.dump(col, open(os.path.join(HERE, "collision.json"), "w"))

    # ---- (iv) obstacle raster + holonomic heuristic.  Goals avoid .5 ties after /reso.
    class G:  # the twin only reads node.x[-1], node.y[-1]
        def __init__(self, x, y):
            self.x, self.y = [x], [y]

    heur = []
    for (gx, gy, reso, rr) in [(44.0, 20.0, 2.0, 1.0), (10.0, 7.0, 2.0, 1.0), (45.2, 20.1, 1.0, 1.0),
                               (25.0, 8.0, 1.0, 2.0)]:
        hm = astar.calc_holonomic_heuristic_with_obstacle(G(gx, gy), ox, oy, reso, rr)
        oxs, oys = [v / reso for v in ox], [v / reso for v in oy]
        P, obsmap = astar.calc_parameters(oxs, oys, rr, reso)
        heur.append({"gx": gx, "gy": gy, "reso": reso, "rr": rr,
                     "dims": [P.minx, P.miny, P.maxx, P.maxy, P.xw, P.yw],
                     "obsmap": [[int(b) for b in row] for row in obsmap],
                     "hmap": [[(v if math.isfinite(v) else "inf") for v in row] for row in hm]})
    # a random 40x36 map at reso 1 (lattice obstacles, density 0.25, full border)
    W, H = 40, 36
    mox, moy = [], []
    for x in range(W + 1):
        for y in range(H + 1):
            if x in (0, W) or y in (0, H) or rng.random() < 0.25:
                mox.append(float(x)); moy.append(float(y))
    hm = astar.calc_holonomic_heuristic_with_obstacle(G(20.2, 17.9), mox, moy, 1.0, 0.5)
    P, obsmap = astar.calc_parameters(mox, moy, 0.5, 1.0)
    heur.append({"gx": 20.2, "gy": 17.9, "reso": 1.0, "rr": 0.5, "ox": mox, "oy": moy,
                 "dims": [P.minx, P.miny, P.maxx, P.maxy, P.xw, P.yw],
                 "obsmap": [[int(b) for b in row] for row in obsmap],
                 "hmap": [[(v if math.isfinite(v) else "inf") for v in row] for row in hm]})
    json.dump(heur, open(os.path.join(HERE, "heuristic.json"), "w"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
