#!/usr/bin/env python3
"""Developer tool: which environment of the benchmark batch leaves the finite range, and when."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

model = bench.build_model("icub23")
N = 1024
data = bench.synthetic_state(model, N, seed=0, dtype=np.float32)
init = data.state_block().copy()
lib = _lib.load()
stream = runtime.Stream()
runtime.set_stream(stream)
dm = runtime.device_model(model, np.float32)
ptr = C.c_void_p(data._state.ptr)
prev = init
for k in range(1, 121):
    _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, 100, stream.handle), "repeat")
    stream.synchronize()
    blk = data.state_block()
    bad = np.where(~np.isfinite(blk).all(axis=0))[0]
    if len(bad):
        print(f"after {100 * k} steps: environments {bad.tolist()} are not finite")
        os.makedirs("gpurun_out", exist_ok=True)
        np.save("gpurun_out/diverged_init.npy", init[:, bad[:1]])
        np.save("gpurun_out/diverged_prev.npy", prev[:, bad[:1]])
        print("state 100 steps before (max |v|):", float(np.abs(prev[:, bad[0]]).max()))
        break
    prev = blk
else:
    print("all finite after 12000 steps")
for k in range(120):
    _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, 100, stream.handle), "repeat")
stream.synchronize()
blk = data.state_block()
print("after 12000 further steps: not finite", np.where(~np.isfinite(blk).all(axis=0))[0].tolist())
