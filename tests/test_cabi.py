"""CPU-side checks of the drop-in boundary: libnavgpu.so loads, exports every symbol that
include/navgpu.h declares, struct layouts agree with the header, and — with no GPU — the product
fails loudly instead of falling back to any CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nav():
    import navigation_amd as nav
    if not os.path.exists(nav.lib_path()):
        nav.build()
    return nav


def header_functions():
    src = open(os.path.join(ROOT, "include", "navgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(navgpu_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(nav):
    from navigation_amd import _lib
    L = nav.lib()
    declared = header_functions()
    assert len(declared) >= 35
    bound = {n for n, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(L, name), f"{name} declared in navgpu.h but not exported"
        assert name in bound, f"{name} has no ctypes signature in navigation_amd/_lib.py"


def test_struct_sizes_match_header(nav, tmp_path):
    """sizeof() of every POD struct as the C compiler sees navgpu.h == the ctypes mirrors
    (navigation_amd/_lib.py and oracle/pyoracle.py)."""
    from navigation_amd import _lib
    from oracle import pyoracle
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "navgpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(navgpu_fleet_desc),sizeof(navgpu_observation),sizeof(navgpu_obstacle_params),'
                   'sizeof(navgpu_inflation_params),sizeof(navgpu_dwa_config),sizeof(navgpu_robot_state),'
                   'sizeof(navgpu_plan_result),sizeof(navgpu_local_limits),sizeof(navgpu_robot_input),sizeof(navgpu_cmd_result),'
                   'sizeof(navgpu_tp_config),sizeof(navgpu_tp_state),sizeof(navgpu_tp_result),sizeof(navgpu_tp_sample));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [_lib.FleetDesc, _lib.Observation, _lib.ObstacleParams, _lib.InflationParams, _lib.DwaConfig, _lib.RobotState,
               _lib.PlanResult, _lib.LocalLimits, _lib.RobotInput, _lib.CmdResult, _lib.TpConfig, _lib.TpState, _lib.TpResult,
               _lib.TpSample]
    assert sizes == [C.sizeof(m) for m in mirrors]
    assert C.sizeof(pyoracle.DwaConfig) == C.sizeof(_lib.DwaConfig)
    assert C.sizeof(pyoracle.PlanResult) == C.sizeof(_lib.PlanResult)
    assert [f[0] for f in pyoracle.DwaConfig._fields_] == [f[0] for f in _lib.DwaConfig._fields_]


def test_no_cpu_fallback_without_gpu(nav):
    L = nav.lib()
    if L.navgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nav.NavgpuError) as e:
        nav.Fleet(1, 64, 64, 0.05)
    assert "no usable HIP device" in str(e.value)


def test_error_strings(nav):
    L = nav.lib()
    assert L.navgpu_strerror(0) == b"ok"
    assert b"fallback" in L.navgpu_strerror(-2)
    assert L.navgpu_kernel_name(4) == b"k_score"
    assert b"gfx950" in L.navgpu_version()


def test_product_never_imports_oracle():
    """The product package must not import, include, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "navigation_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+oracle|pyoracle|liboracle|oracle/|oracle_capi|_oracle\.hpp|orc_[a-z_]+\()")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                m = bad.search(text)
                assert m is None, f"{os.path.join(dirpath, fn)} references the oracle: {m.group(0)}"
