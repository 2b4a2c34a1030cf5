#!/usr/bin/env python3
"""Developer tool: time the step kernel for several batch sizes (HIP events, in place)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024,2048,4096,8192,65536")
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--model", default="icub23")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rollout", action="store_true", help="also time the same steps fused in one launch (jxs_rollout) and with the state of every step recorded (jxs_rollout_recorded)")
args = ap.parse_args()
model = bench.build_model(args.model)
dtype = np.dtype(args.dtype)
lib = _lib.load()
stream = runtime.Stream()
for N in [int(x) for x in args.sizes.split(",")]:
    data = bench.synthetic_state(model, N, seed=0, dtype=dtype)
    runtime.set_stream(stream)
    dm = runtime.device_model(model, dtype)
    ptr = C.c_void_p(data._state.ptr)
    def run(k):  # the launch path of bench.py: hipGraph replays of single-step launches
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, k, stream.handle), "jxs_step_repeat")

    run(args.steps)
    stream.synchronize()
    best = []
    for _ in range(args.reps):
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record(stream)
        run(args.steps)
        e1.record(stream)
        stream.synchronize()
        best.append(e0.elapsed_ms(e1) / args.steps * 1e3)
    us = float(np.median(best))
    fin = np.isfinite(data.state_block()).all(axis=0).mean()
    print(f"lib={os.environ.get('JAXSIM_AMD_LIB', 'default')} model={args.model} {dtype.name} N={N:7d}  {us:9.2f} us/step  "
          f"{N / us:9.2f} M env-steps/s  finite={fin:.4f}", flush=True)
    if args.rollout:
        K = min(args.steps, 200)
        fresh = bench.synthetic_state(model, N, seed=0, dtype=dtype)
        fp = C.c_void_p(fresh._state.ptr)
        start = fresh._state.copy()
        traj = runtime.DeviceArray(K * fresh._state.shape[0], N, dtype, tile=fresh._state.tile)
        out = []
        for name, call in (("jxs_rollout", lambda: lib.jxs_rollout(dm.handle, fp, None, None, 2, N, K, stream.handle)),
                           ("jxs_rollout_recorded", lambda: lib.jxs_rollout_recorded(dm.handle, fp, None, 0, None, 2, N, K, C.c_void_p(traj.ptr), stream.handle))):
            ts = []
            for _ in range(args.reps + 1):
                _lib.check(lib.jxs_memcpy_d2d(fp, C.c_void_p(start.ptr), start.nbytes, stream.handle), "jxs_memcpy_d2d")
                e0, e1 = runtime.Event(), runtime.Event()
                e0.record(stream)
                _lib.check(call(), name)
                e1.record(stream)
                stream.synchronize()
                ts.append(e0.elapsed_ms(e1) / K * 1e3)
            u = float(np.median(ts[1:]))
            out.append(f"{name} {u:8.2f} us/step {N / u:8.1f} M env-steps/s")
        print(f"    fused, {K} steps per launch: " + "   ".join(out), flush=True)
        del traj, fresh, start
    runtime.set_stream(None)
