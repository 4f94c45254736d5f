"""MI355X (gfx950) implementation of the hybrid A* hot path of ishidur/path_planner.

Host-side mirror of the reference's planner entry points over the C ABI in
include/pp.h.  The compute lives in path_planner_amd/csrc (hand-written HIP).
"""
from . import _abi  # noqa: F401
from ._abi import Car, Config, demo_config, load_library  # noqa: F401

__all__ = ["Car", "Config", "demo_config", "load_library"]
