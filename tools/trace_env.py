#!/usr/bin/env python3
"""Developer tool: single-step trajectory of one environment on the GPU (state after every step)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import jaxsim_amd.api as js  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

model = bench.build_model("icub23")
init = np.load(sys.argv[1]).astype(np.float32)
steps = int(sys.argv[2])
data = js.data.JaxSimModelData.from_state_block(model, np.repeat(init, 2, axis=1), 2)
lib = _lib.load()
dm = runtime.device_model(model, np.float32)
ptr = C.c_void_p(data._state.ptr)
traj = [init[:, 0].copy()]
for k in range(steps):
    _lib.check(lib.jxs_step(dm.handle, ptr, ptr, None, None, 2, 2, None), "step")
    b = data.state_block()
    traj.append(b[:, 0].copy())
    if not np.isfinite(b).all():
        print("non-finite after step", k + 1)
        break
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/traj.npy", np.array(traj))
print("saved", len(traj))
