"""The CPU oracle is the checker of every parity test, so it is itself checked for memory errors
and undefined behaviour: `make -C oracle san` builds it with AddressSanitizer and
UndefinedBehaviorSanitizer (host code, runtime linked statically) and walks every entry point on
the reference's demo inputs (src/main.rs:46-84, :215-234) and seeded random batches."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_oracle_walk_under_asan_and_ubsan():
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle sanitizer walk: ok" in r.stdout
