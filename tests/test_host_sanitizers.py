"""Host-side AddressSanitizer + UndefinedBehaviorSanitizer run of the library's host logic.

`scripts/build_host_asan.sh` builds every translation unit with `-Xarch_host -fsanitize=address,undefined` (the device pass as
always) into scratch/libyolo4hip_hostasan.so; this test re-runs the host-only ABI / plan tests -- y4_create, y4_layer_info,
y4_set_workspace_aliasing + y4_workspace_bytes, y4_set_tiles with shipped and malformed schedules, y4_launch_counts, the error paths --
in a child process with the sanitizer runtime preloaded.  A report aborts the child (`-fno-sanitize-recover`, ASan's default).  Skipped
when the sanitizer build is absent (it takes three minutes: built on demand, not by `build()`); GPU ASan / XNACK do not exist on this pool.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "scratch", "libyolo4hip_hostasan.so")
RT_FILE = os.path.join(ROOT, "scratch", "hostasan_runtime.txt")


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(RT_FILE)), reason="run scripts/build_host_asan.sh first (3 min)")
def test_host_logic_under_asan_and_ubsan():
    rt = open(RT_FILE).read().strip()
    assert os.path.exists(rt), rt
    out = subprocess.run(["nm", "-D", LIB], capture_output=True, text=True).stdout
    assert "__asan_init" in out and "__ubsan_handle" in out, "the library is not instrumented"
    env = dict(os.environ, LD_PRELOAD=rt, YOLO4HIP_LIB=LIB, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_lib_abi.py"), os.path.join(ROOT, "tests", "test_plan.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
