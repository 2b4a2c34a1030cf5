#!/usr/bin/env python3
"""Developer tool: cycle breakdown of the RigidContacts step (library built with -DJXS_PHASE_TIMING,
JAXSIM_AMD_LIB=.../libjaxsim_amd_timing.so).  python tools/phase_timing_rigid.py [points] [N]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import jaxsim_amd.api as js  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kind = sys.argv[3] if len(sys.argv) > 3 else "rigid"       # rigid | relaxed
standing = len(sys.argv) > 4 and sys.argv[4] == "standing"
zoo = helpers.ModelZoo()
robot = "icub" if pts == 32 else "anymal"
idx = helpers.ANYMAL_FEET_4 if pts == 4 else helpers.ANYMAL_FEET_16 if pts == 16 else list(range(32))
if kind == "rigid":
    model = helpers.rigid_model(zoo(robot), idx, K=1e4, D=2e2)
else:
    model = helpers.relaxed_model(zoo(robot), idx, mu=0.5)
if standing:
    d = helpers.standing_data(model, N, seed=0, dtype=np.float32, noise=0.003)
else:
    d = zoo.random_data(robot, N, seed=0, dtype=np.float32)
data = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), 2)
lib = _lib.load()
dm = runtime.device_model(model, np.float32)
blocks = (N + dm.layout.tile - 1) // dm.layout.tile
buf = C.c_void_p()
lib.jxs_malloc(C.byref(buf), blocks * 64 * 8)
lib.jxs_debug_set_stamp_buffer.argtypes = [C.c_void_p]
ptr = C.c_void_p(data._state.ptr)
for _ in range(60):
    lib.jxs_step(dm.handle, ptr, ptr, None, None, 2, N, None)
lib.jxs_memset(buf, 0, blocks * 64 * 8, None)
lib.jxs_debug_set_stamp_buffer(buf)
lib.jxs_step(dm.handle, ptr, ptr, None, None, 2, N, None)
runtime.synchronize()
out = np.zeros((blocks, 64), dtype=np.int64)
lib.jxs_memcpy_d2h(out.ctypes.data_as(C.c_void_p), buf, out.nbytes, None)
if kind == "relaxed":
    has = (out[:, 11] > 0) & (out[:, 13] > 0)
    o = out[has]
    tot = o[:, 10] - o[:, 0]
    print(f"points={pts} N={N} relaxed{' standing' if standing else ''}: {blocks} waves, {has.sum()} with contacts")
    print("  total                 %9.0f (max %d)" % (tot.mean(), tot.max()))
    # [round 5] solved in the tree: no Delassus matrix; the second figure is the points' terms, W into the link lanes and
    # the articulated-inertia recursion of the augmented tree (ta_factor)
    print("  delassus (dense path) %9.0f" % (o[:, 12] - o[:, 11]).mean())
    print("  terms + W + tree factor / H + cholesky %9.0f" % (o[:, 14] - o[:, 12]).mean())
    print("  first solve           %9.0f" % (o[:, 15] - o[:, 14]).mean())
    print("  refinement            %9.0f" % (o[:, 13] - o[:, 15]).mean())
    print("  everything else       %9.0f" % (tot - (o[:, 13] - o[:, 11])).mean())
    sys.exit(0)
has = (out[:, 13] > out[:, 12]) & (out[:, 15] > out[:, 14]) & (out[:, 14] > out[:, 13])
print(f"points={pts} N={N}: {blocks} waves, {has.sum()} with contacts in both stages")
o = out[has]
tot = o[:, 10] - o[:, 0]
print("  total                 %9.0f (max %d)" % (tot.mean(), tot.max()))
print("  stage0 delassus       %9.0f" % (o[:, 12] - o[:, 11]).mean())
print("  stage0 QP             %9.0f (max %d)" % ((o[:, 13] - o[:, 12]).mean(), (o[:, 13] - o[:, 12]).max()))
print("  stage1 delassus+impact%9.0f" % (o[:, 15] - o[:, 14]).mean())
print("  everything else       %9.0f" % (tot - (o[:, 13] - o[:, 11]) - (o[:, 15] - o[:, 14])).mean())
print("  QP iterations (max over the wave): mean %.1f  p90 %d  max %d" % (o[:, 16].mean(), np.percentile(o[:, 16], 90), o[:, 16].max()))
print("  impact CG iterations:              mean %.1f  max %d" % (o[:, 17].mean(), o[:, 17].max()))
worst = np.argsort(-tot)[:5]
print("  slowest waves: total / QP / impact / QP iterations:", [(int(tot[i]), int(o[i, 13] - o[i, 12]), int(o[i, 15] - o[i, 14]), int(o[i, 16])) for i in worst])
free = out[~has & (out[:, 10] > 0)]
if len(free):
    print("  waves without contact: total %9.0f" % (free[:, 10] - free[:, 0]).mean())
