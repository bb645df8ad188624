"""Device sincos (as k_score evaluates headings) against the host's libm, bit for bit:  python3 tools/probe_sincos.py"""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import navigation_amd as nav  # noqa: E402

L = nav.lib()
rs = np.random.RandomState(7)
n = 1 << 20
th = np.concatenate([rs.uniform(-2 * math.pi, 2 * math.pi, n).astype(np.float32).astype(np.float64),       # rollout headings: floats
                     math.pi / 2 + rs.uniform(-math.pi, math.pi, n // 4).astype(np.float32).astype(np.float64)])  # M_PI_2 + theta
sn, cs = np.empty_like(th), np.empty_like(th)
rc = L.navgpu_device_sincos(0, th.ctypes.data_as(C.c_void_p), len(th), sn.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
hs = np.array([math.sin(v) for v in th])
hc = np.array([math.cos(v) for v in th])
for name, d, h in (("sin", sn, hs), ("cos", cs, hc)):
    ulp = np.abs(d.view(np.int64) - h.view(np.int64))
    print(name, "different:", int((ulp != 0).sum()), "of", len(th), "(%.3g)" % ((ulp != 0).mean()), "max ulp", int(ulp.max()))
