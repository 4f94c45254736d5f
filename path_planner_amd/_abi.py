"""ctypes mirror of include/pp.h: plain-data structs and the library loader.

The product library (libpp_hip.so) is hand-written HIP for gfx950; there is no
CPU fallback.  `load_library()` raises if the shared object is missing so that a
GPU run can never silently pass on something else.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PP_HIP_LIB") or os.path.join(PKG_DIR, "csrc", "libpp_hip.so")

PP_OK = 0
PP_NO_PATH = 1
PP_ERR_INVALID = -1
PP_ERR_HIP = -2
PP_ERR_NAN_COST = -3
PP_ERR_DEGENERATE_RS = -4
PP_ERR_EMPTY_RS = -5
PP_ERR_BOUNDS = -6
PP_ERR_CAPACITY = -7
PP_ERR_NONFINITE = -8
PP_ERR_EMPTY_MAP = -9
PP_ERR_UNSUPPORTED = -10
PP_ERR_LIMIT = -11

PP_RS_MAX_SEG = 5
PP_RS_MAX_PATHS = 46

c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_u64_p = C.POINTER(C.c_uint64)


class Car(C.Structure):
    """reeds_shepp.rs:42-47"""
    _fields_ = [("center2front", C.c_double), ("center2back", C.c_double),
                ("width", C.c_double), ("wheel_base", C.c_double)]


class Config(C.Structure):
    """hybrid_astar.rs:14-28 (constructor order :31-45)"""
    _fields_ = [("xy_reso", C.c_double), ("yaw_reso", C.c_double), ("move_step", C.c_double),
                ("n_steer", C.c_uint64), ("collision_check_step", C.c_uint64),
                ("extend_bound", C.c_double), ("gear_cost", C.c_double),
                ("backward_cost", C.c_double), ("steer_change_cost", C.c_double),
                ("steer_angle_cost", C.c_double), ("h_cost", C.c_double), ("car", Car),
                ("max_steer", C.c_double)]


class Param(C.Structure):
    """hybrid_astar.rs:65-78 minus obstacle copies and KD-tree"""
    _fields_ = [("minx", C.c_uint64), ("miny", C.c_uint64), ("minyaw", C.c_int64),
                ("maxx", C.c_uint64), ("maxy", C.c_uint64), ("xw", C.c_uint64),
                ("yw", C.c_uint64), ("xyreso", C.c_double), ("yawreso", C.c_double)]


class RsWord(C.Structure):
    """reeds_shepp.rs:9-17 without the trajectory vectors"""
    _fields_ = [("lengths", C.c_double * PP_RS_MAX_SEG), ("total_length", C.c_double),
                ("ctypes", C.c_uint8 * PP_RS_MAX_SEG), ("nseg", C.c_uint8), ("pad_", C.c_uint8 * 2)]

    def word(self):
        return bytes(self.ctypes[: self.nseg]).decode()

    def lens(self):
        return [self.lengths[i] for i in range(self.nseg)]


class Path(C.Structure):
    """hybrid_astar.rs:81-87 over caller-owned arrays"""
    _fields_ = [("x", c_double_p), ("y", c_double_p), ("yaw", c_double_p),
                ("direction", c_i64_p), ("cap", C.c_uint64), ("len", C.c_uint64),
                ("cost", C.c_double)]


class PlanStats(C.Structure):
    _fields_ = [("expansions", C.c_uint64), ("stale_pops", C.c_uint64), ("rollouts", C.c_uint64),
                ("poses_checked", C.c_uint64), ("rs_paths", C.c_uint64), ("ambiguous", C.c_uint64),
                ("closed_cap", C.c_uint64), ("closed_order", c_u64_p),
                ("rs_final_length", C.c_double), ("t_heuristic_ms", C.c_double),
                ("t_search_ms", C.c_double), ("first_ambiguous", C.c_uint64)]


class ExpandOut(C.Structure):
    _fields_ = [("valid", c_u8_p), ("xind", c_u64_p), ("yind", c_u64_p), ("yawind", c_i64_p),
                ("cost", c_double_p), ("end_x", c_double_p), ("end_y", c_double_p),
                ("end_yaw", c_double_p)]


assert C.sizeof(RsWord) == 56
assert C.sizeof(Config) == 128


def demo_config():
    """The Config of src/main.rs:215-230."""
    import math
    return Config(2.0, math.radians(15.0), 0.4, 20, 5, 1.4, 100.0, 5.0, 5.0, 1.0, 15.0,
                  Car(4.5, 1.0, 3.0, 3.5), 0.6)


_lib = None


def load_library(path=None):
    """Load libpp_hip.so and declare every entry point of include/pp.h.  Raises if missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build it with `python -m path_planner_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(p)
    vp = C.c_void_p
    d, u8, i32, i64, u64 = c_double_p, c_u8_p, c_i32_p, c_i64_p, c_u64_p
    cfg = C.POINTER(Config)
    sigs = {
        "pp_ctx_create": (C.c_int32, [C.c_int32, vp, C.POINTER(vp)]),
        "pp_ctx_destroy": (None, [vp]),
        "pp_ctx_stream": (vp, [vp]),
        "pp_ctx_synchronize": (C.c_int32, [vp]),
        "pp_strerror": (C.c_char_p, [C.c_int32]),
        "pp_last_hip_error": (C.c_int32, [vp]),
        "pp_version": (C.c_char_p, []),
        "pp_ctx_set_max_expansions": (C.c_int32, [vp, C.c_uint64]),
        "pp_ctx_set_heuristic_limit": (C.c_int32, [vp, C.c_double]),
        "pp_ctx_set_heuristic_workspace": (C.c_int32, [vp, C.c_uint64]),
        "pp_ctx_set_host_threads": (C.c_int32, [vp, C.c_uint32]),
        "pp_ctx_set_table_budget": (C.c_int32, [vp, C.c_uint64]),
        "pp_ctx_set_batch_parts": (C.c_int32, [vp, C.c_uint32]),
        "pp_ctx_device_bytes": (C.c_uint64, [vp]),
        "pp_plan_batch_counters": (C.c_int32, [vp, C.POINTER(C.c_uint64)]),
        "pp_wavefront_stats": (C.c_int32, [vp, c_u64_p]),
        "pp_ctx_set_relax_band": (C.c_int32, [vp, C.c_double]),
        "pp_map_create": (C.c_int32, [vp, d, d, C.c_uint64, C.POINTER(vp)]),
        "pp_map_create_raster": (C.c_int32, [vp, u8, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64,
                                             C.POINTER(vp)]),
        "pp_map_destroy": (None, [vp]),
        "pp_map_num_obstacles": (C.c_uint64, [vp]),
        "pp_map_is_lattice": (C.c_int32, [vp]),
        "pp_map_param": (C.c_int32, [vp, cfg, C.POINTER(Param)]),
        "pp_collision_batch": (C.c_int32, [vp, vp, cfg, C.c_uint64, d, d, d, u8]),
        "pp_collision_batch_dev": (C.c_int32, [vp, vp, cfg, C.c_uint64, vp, vp, vp, vp]),
        "pp_is_collision": (C.c_int32, [vp, vp, cfg, C.c_uint64, d, d, d, i32, u64]),
        "pp_motion_set": (C.c_int32, [cfg, d, d, C.c_uint64, u64]),
        "pp_expand_batch": (C.c_int32, [vp, vp, cfg, C.c_uint64, d, d, d, i64, d, d,
                                        C.POINTER(ExpandOut)]),
        "pp_expand_batch_dev": (C.c_int32, [vp, vp, cfg, C.c_uint64, vp, vp, vp, vp, vp, vp,
                                            C.POINTER(ExpandOut)]),
        "pp_rollout_points": (C.c_int32, [vp, cfg, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, d, d, d, C.c_uint64, u64]),
        "pp_heuristic": (C.c_int32, [vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, d,
                                     C.c_uint64, u64]),
        "pp_obsmap": (C.c_int32, [vp, vp, C.c_double, C.c_double, u8, C.c_uint64, u64]),
        "pp_wavefront_dev": (C.c_int32, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64,
                                         vp, u64]),
        "pp_wavefront_slab_dev": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                              vp, vp, C.c_int32, C.c_uint64, C.c_uint64, vp, vp,
                                              i32, i32, u64]),
        "pp_slab_create": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint64,
                                       C.c_uint64, C.POINTER(vp)]),
        "pp_slab_relax": (C.c_int32, [vp, vp, vp, C.c_double, vp, vp, d, u64]),
        "pp_slab_read": (C.c_int32, [vp, vp]),
        "pp_slab_gather": (C.c_int32, [vp, C.c_uint64, vp, vp, vp]),
        "pp_slab_reset": (C.c_int32, [vp, vp]),
        "pp_slab_destroy": (None, [vp]),
        "pp_slab_relax_all": (C.c_int32, [C.POINTER(vp), C.c_uint32, C.c_double, u64]),
        "pp_wavefront_tiled": (C.c_int32, [C.POINTER(vp), C.c_uint32, C.c_uint64, C.c_uint64, u8, C.c_uint64,
                                           C.c_uint64, d, C.c_double, u64]),
        "pp_rs_batch": (C.c_int32, [vp, C.c_uint64, d, d, d, d, d, d, C.c_double, i32, i32,
                                    C.POINTER(RsWord), u8]),
        "pp_rs_words_batch": (C.c_int32, [vp, C.c_uint64, d, d, d, d, d, d, C.c_double, i32,
                                          C.POINTER(RsWord), u8]),
        "pp_rs_batch_dev": (C.c_int32, [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp,
                                        vp, vp]),
        "pp_rs_all_paths": (C.c_int32, [vp] + [C.c_double] * 8 + [i32, C.POINTER(RsWord), u64, d, d,
                                                                  d, i64, C.c_uint64]),
        "pp_rs_optimal_path": (C.c_int32, [vp] + [C.c_double] * 8 + [C.POINTER(RsWord), d, d, d, i64,
                                                                     C.c_uint64, u64]),
        "pp_rs_path_cost": (C.c_double, [C.POINTER(RsWord), cfg]),
        "pp_analytic_batch": (C.c_int32, [vp, vp, cfg, C.c_uint64, d, d, d, d, d, d, u8, i32, i32,
                                          C.POINTER(RsWord), d]),
        "pp_analytic_batch_dev": (C.c_int32, [vp, vp, cfg, C.c_uint64, vp, vp, vp, vp, vp, vp]),
        "pp_analytic_batch_dev_static": (C.c_int32, [vp, vp, cfg, C.c_uint64, vp, vp, vp, vp, vp, vp]),
        "pp_analytic_static_max": (C.c_uint64, []),
        "pp_plan": (C.c_int32, [vp, vp, cfg, d, d, C.POINTER(Path), C.POINTER(PlanStats)]),
        "pp_plan_batch": (C.c_int32, [vp, vp, cfg, C.c_uint64, d, d, C.POINTER(Path),
                                      C.POINTER(PlanStats), i32]),
        "pp_plan_batch_multi": (C.c_int32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, cfg,
                                            C.c_uint64, d, d, C.POINTER(Path), C.POINTER(PlanStats), i32]),
        "pp_astar_plan": (C.c_int32, [vp, vp] + [C.c_double] * 6 + [d, d, C.c_uint64, u64, u64,
                                                                    C.c_uint64, u64]),
        "pp_astar_plan_batch": (C.c_int32, [vp, vp, C.c_uint64, d, d, C.c_double, C.c_double, d, d,
                                            C.c_uint64, u64, i32, u64]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._pp_symbols = sorted(sigs)
    if path is None:
        _lib = lib
    return lib
