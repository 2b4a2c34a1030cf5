#!/usr/bin/env python3
"""Developer tool: wall time of a timed region of n step launches, T(n) = a + b n (jxs_step_repeat_timed):
how much of a short benchmark region is launch + synchronisation latency."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JAXSIM_AMD_SPECIALIZE", "1")
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

model = bench.build_model("icub23")
N = 1024
data = bench.synthetic_state(model, N, seed=0, dtype=np.float32)
lib = _lib.load()
stream = runtime.Stream()
runtime.set_stream(stream)
dm = runtime.device_model(model, np.float32)
ptr = C.c_void_p(data._state.ptr)
sec = C.c_double()
rows = []
for n in (2, 5, 10, 20, 40, 80, 160):
    for _ in range(3):
        _lib.check(lib.jxs_step_repeat_timed(dm.handle, ptr, None, None, 2, N, n, stream.handle, C.byref(sec)), "timed")
    ts = []
    for _ in range(15):
        _lib.check(lib.jxs_step_repeat_timed(dm.handle, ptr, None, None, 2, N, n, stream.handle, C.byref(sec)), "timed")
        ts.append(sec.value * 1e6)
    rows.append((n, float(np.median(ts)), float(np.min(ts))))
    print(f"n = {n:4d}: median {rows[-1][1]:8.2f} us  min {rows[-1][2]:8.2f} us  -> {rows[-1][1] / n:6.2f} us per launch")
A = np.array([[1.0, r[0]] for r in rows])
a, b = np.linalg.lstsq(A, np.array([r[1] for r in rows]), rcond=None)[0]
print(f"fit: T(n) = {a:.1f} us + {b:.2f} us * n")
