"""The planner's fork-join pool (path_planner_amd/csrc/pp_pool.h) under several callers at once, as
the parts of a large batch use it: compiled on its own and stressed on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "pool_stress.cpp")


@pytest.mark.parametrize("callers,workers", [(4, 6), (12, 3), (1, 7), (3, 0)])
def test_every_item_runs_once_under_concurrent_callers(tmp_path, callers, workers):
    cxx = shutil.which("g++")
    assert cxx, "g++ not found"
    exe = str(tmp_path / "pool_stress")
    subprocess.run([cxx, "-O2", "-std=c++17", "-pthread", "-o", exe, SRC], check=True)
    r = subprocess.run([exe, str(callers), str(workers), "1.0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 errors" in r.stdout, r.stdout


def test_pool_is_clean_under_thread_sanitizer(tmp_path):
    cxx = shutil.which("g++")
    exe = str(tmp_path / "pool_stress_tsan")
    b = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe, SRC],
                       capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("no thread sanitizer runtime for g++ here: " + b.stderr[-200:])
    r = subprocess.run([exe, "4", "4", "2.0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-2000:]
