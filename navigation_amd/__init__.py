"""navigation_amd — MI355X-native costmap + DWA hot path (libnavgpu.so) and its Python binding.

The product is the C-ABI shared library built from navigation_amd/csrc (HIP kernels for gfx950);
this package is the ctypes binding the tests, bench.py and the fleet harness use.  It never
imports the CPU oracle and has no CPU fallback: loading fails loudly when libnavgpu.so is missing.
"""
from ._lib import (TpConfig, DwaConfig, FleetDesc, InflationParams, NavgpuError, Observation, ObstacleParams, PlanResult,
                   RobotState, build, lib, lib_path)
from .fleet import Fleet
from .navfn import NavFn

__all__ = ["Fleet", "NavFn", "TpConfig", "DwaConfig", "FleetDesc", "InflationParams", "ObstacleParams", "Observation", "PlanResult",
           "RobotState", "NavgpuError", "build", "lib", "lib_path"]
