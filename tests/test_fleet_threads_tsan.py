"""SURVEY 8(b) threading: libnavgpu's host side under ThreadSanitizer (CPU box, no GPU).

The four host translation units of libnavgpu.so are compiled with g++ -fsanitize=thread against tests/tsan/hip_host_stub.cpp
(host-memory stand-ins for the HIP runtime calls and no-op kernel launchers: only the bookkeeping, the staging mirrors and
the locking are under test) and driven by tests/tsan/fleet_threads.cpp: one thread runs control cycles on a 3-robot fleet
while three others reconfigure the planner (re-allocating its tables), the layers and the footprint, and read state back,
with no lock of their own.  ThreadSanitizer reports any data race; the harness itself checks that every cycle's result
comes from one configuration.  The same build with the per-fleet mutex compiled out (NAVGPU_TEST_NO_FLEET_LOCK) is the
negative control: TSAN must flag it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "navigation_amd", "csrc", f) for f in ("navgpu_host.cpp", "navgpu_local_planner.cpp", "navgpu_tp.cpp", "navgpu_navfn.cpp")]
SRC += [os.path.join(ROOT, "tests", "tsan", f) for f in ("hip_host_stub.cpp", "fleet_threads.cpp")]
HIP_INC = "/opt/rocm/include"


def build(tmp, name, extra):
    exe = os.path.join(tmp, name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I", HIP_INC] + extra + SRC + ["-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(not os.path.isdir(os.path.join(HIP_INC, "hip")), reason="HIP headers not installed")
def test_fleet_calls_from_four_threads_are_race_free(tmp_path):
    exe = build(str(tmp_path), "fleet_threads", [])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "ThreadSanitizer" not in r.stderr
    assert "mixed_results 0" in r.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(HIP_INC, "hip")), reason="HIP headers not installed")
def test_tsan_flags_the_build_without_the_fleet_mutex(tmp_path):
    exe = build(str(tmp_path), "fleet_threads_nolock", ["-DNAVGPU_TEST_NO_FLEET_LOCK"])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode != 0 and "ThreadSanitizer: data race" in r.stderr, "the harness cannot see a race it should see"
