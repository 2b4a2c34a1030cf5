#!/usr/bin/env python3
"""Developer tool: per-phase cycle breakdown of the step kernel (needs the library built with
-DJXS_PHASE_TIMING, e.g. JAXSIM_AMD_LIB=.../libjaxsim_amd_timing.so)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = bench.build_model("icub23")
data = bench.synthetic_state(model, N, seed=0, dtype=np.float32)
lib = _lib.load()
dm = runtime.device_model(model, np.float32)
blocks = (N + 1) // 2
buf = C.c_void_p()
lib.jxs_malloc(C.byref(buf), blocks * 32 * 8)
lib.jxs_memset(buf, 0, blocks * 32 * 8, None)
lib.jxs_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.jxs_debug_set_stamp_buffer(buf)
ptr = C.c_void_p(data._state.ptr)
for _ in range(20):
    lib.jxs_step(dm.handle, ptr, ptr, None, None, 2, N, None)
runtime.synchronize()
out = np.zeros((blocks, 32), dtype=np.int64)
lib.jxs_memcpy_d2h(out.ctypes.data_as(C.c_void_p), buf, out.nbytes, None)
d = np.diff(out[:, :11], axis=1)
names = ["loads arrive", "actuation+local xform", "FK (pointer jumping)", "velocities", "contacts", "inertia+bias",
         "pass 2", "base solve", "pass 3", "integrate+stores"]
print(f"N={N}: mean cycles per phase over {blocks} waves (s_memtime ticks)")
for n, m, mx in zip(names, d.mean(axis=0), d.max(axis=0)):
    print(f"  {n:24s} {m:9.0f}  (max {mx})")
print(f"  {'total':24s} {(out[:, 10] - out[:, 0]).mean():9.0f}")
if out[:, 20].any():
    for i, n in ((20, "index tables arrived"), (21, "base state rows arrived"), (22, "point tables arrived"), (23, "joint state rows arrived")):
        print(f"  since wave start: {n:28s} {(out[:, i] - out[:, 0]).mean():9.0f}  (max {(out[:, i] - out[:, 0]).max()})")
if out[:, 24].any():  # row-distributed pass 2: one stamp per tree level, deepest level first
    lv = out[:, 31:23:-1]
    dl = np.diff(np.concatenate([out[:, 6:7], lv], axis=1), axis=1)
    print("  pass 2 per level (deepest first): " + " ".join(f"{x:.0f}" for x in dl.mean(axis=0)))
span = out[:, 10].max() - out[:, 0].min()
print(f"  first start -> last end: {span} ticks")
