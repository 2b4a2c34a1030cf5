#!/usr/bin/env python3
"""[round 4] Divergence-rate audit at scale.

The saturated bench figure steps 65 536 environments of the headline workload 1000 times; round 3 saw ~0.1 % of them end
non-finite and argued "chaos at the stability limit of the explicit contact model, triggered by fp32 rounding" without
a rate for any other implementation of the same arithmetic.  This tool steps THE SAME initial states through

  * the HIP kernel in fp32 (the product path, hardware rcp / rsq / sincos) and in fp64,
  * the oracle's C port (oracle/cport, the reference's formulation, IEEE libm) in fp32 and in fp64 on the host cores,

and reports the fraction of environments whose state is non-finite (or has left the scene: |x| > 1e6) after `--steps`
steps, how the sets overlap, and when the GPU's environments leave.  Test / measurement infrastructure: the oracle is
the checker here, never the product.

    python tools/divergence_audit.py [--envs 65536] [--steps 1000] [--threads 128] > profiles/r04_divergence_audit.txt
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--threads", type=int, default=min(128, os.cpu_count() or 8))
ap.add_argument("--model", default="icub23")
ap.add_argument("--chunk", type=int, default=100, help="GPU: steps between two looks at the state (when do environments leave?)")
args = ap.parse_args()

model = bench.build_model(args.model)
N = args.envs
lib = _lib.load()
stream = runtime.Stream()
runtime.set_stream(stream)
data32 = bench.synthetic_state(model, N, seed=0, dtype=np.float32)
block0 = data32.state_block().copy()  # [rows, N] fp32: the one set of initial states everybody starts from


def gone(block):
    with np.errstate(invalid="ignore"):
        return ~np.isfinite(block).all(axis=0) | (np.abs(np.nan_to_num(block, nan=0.0, posinf=0.0, neginf=0.0)).max(axis=0) > 1e6)


def gpu(dtype):
    import jaxsim_amd.api as js

    d = js.data.JaxSimModelData.from_state_block(model, block0.astype(dtype))
    dm = runtime.device_model(model, dtype)
    ptr = C.c_void_p(d._state.ptr)
    when = np.full(N, -1, dtype=np.int64)
    done = 0
    t0 = time.time()
    while done < args.steps:
        k = min(args.chunk, args.steps - done)
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, k, stream.handle), "jxs_step_repeat")
        stream.synchronize()
        done += k
        g = gone(d.state_block())
        when[(when < 0) & g] = done
    return gone(d.state_block()), when, time.time() - t0


def cpu(dtype):
    from oracle import cport

    t0 = time.time()
    out = cport.step(model, np.ascontiguousarray(block0.astype(dtype)), n_steps=args.steps, n_threads=args.threads)
    return gone(out), time.time() - t0


print(f"# divergence audit: {args.model} synthetic humanoid, soft contacts (bench.py build_model / synthetic_state seed 0), N = {N}, {args.steps} steps of dt = {model.time_step}")
print(f"# host: {os.cpu_count()} CPUs, C port on {args.threads} threads")
res = {}
g32, when32, t = gpu(np.float32)
res["gpu_fp32"] = g32
print(f"HIP kernel fp32      gone {g32.sum():6d} / {N} = {g32.mean():.5f}   ({t:.1f} s)")
g64, when64, t = gpu(np.float64)
res["gpu_fp64"] = g64
print(f"HIP kernel fp64      gone {g64.sum():6d} / {N} = {g64.mean():.5f}   ({t:.1f} s)")
for name, dt in (("cport_fp32", np.float32), ("cport_fp64", np.float64)):
    m, t = cpu(dt)
    res[name] = m
    print(f"C port (oracle) {name[-4:]} gone {m.sum():6d} / {N} = {m.mean():.5f}   ({t:.1f} s)")
names = list(res)
print("# overlap of the sets (environments gone in both):")
print("               " + "".join(f"{n:>12s}" for n in names))
for a in names:
    print(f"{a:>14s} " + "".join(f"{int((res[a] & res[b]).sum()):12d}" for b in names))
print("# when the HIP fp32 / fp64 environments leave (steps, cumulative):")
for s in range(args.chunk, args.steps + 1, args.chunk):
    print(f"  <= {s:5d}: fp32 {int(((when32 > 0) & (when32 <= s)).sum()):6d}   fp64 {int(((when64 > 0) & (when64 <= s)).sum()):6d}")
