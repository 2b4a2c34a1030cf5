#!/usr/bin/env python3
"""Developer experiment: how much of the model-specialised kernel's gain comes from the feature switches and how
much from the tree-shape constants.  Builds jxs_spec.hip with SUBSETS of the description as constants (the
description string itself stays the model's, so that the library accepts the object) and times the step."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["JAXSIM_AMD_SPECIALIZE"] = "0"
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime, specialize as sp  # noqa: E402

FEATURES = ("floating", "any_suc", "seg_dpp_ok", "row_mode", "flat", "enable_friction", "pq_half", "anchored", "rigid", "rk4fast", "n_chunks")
SHAPE = ("n_rounds", "max_depth", "seg_steps", "row_cross_levels", "row_ppull_levels", "row_pull_counts", "nonadj_levels", "maxch_nib")

model = bench.build_model("icub23")
N = 1024
text = sp.spec(model, np.float32)
head, assign = text.rsplit(";", 1)
items = assign.split(",")
lib = _lib.load()
stream = runtime.Stream()
runtime.set_stream(stream)


def pick(names):
    return ",".join(it for it in items if any(it.startswith("P." + n) for n in names))


def timed(tag, sub):
    model.__dict__.pop("_device", None)
    dm = runtime.device_model(model, np.float32)
    if sub is not None:
        out = "/tmp/spec_" + "".join(ch if ch.isalnum() else "_" for ch in tag) + ".so"
        cmd = [sp._HIPCC, *sp._FLAGS, "-DJXS_SPEC_T=float", "-DJXS_SPEC_G=32", "-DJXS_SPEC_MODE=0", f"-DJXS_SPEC_ASSIGN={sub}",
               f'-DJXS_SPEC_STRING="{text}"', "jxs_spec.hip", "-o", out]
        subprocess.run(cmd, cwd=sp._CSRC, check=True, capture_output=True)
        _lib.check(lib.jxs_model_attach_specialized(dm.handle, 0, out.encode()), "attach")
    data = bench.synthetic_state(model, N, seed=0, dtype=np.float32)
    ptr = C.c_void_p(data._state.ptr)
    best = []
    for _ in range(4):
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, 2000, stream.handle), "repeat")
        stream.synchronize()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record(stream)
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, 2000, stream.handle), "repeat")
        e1.record(stream)
        stream.synchronize()
        best.append(e0.elapsed_ms(e1) / 2000 * 1e3)
    print(f"{tag:34s} {min(best):6.2f} us per step", flush=True)


timed("library kernel (KV_COMMON where it applies; JXS_DISABLE_COMMON_VARIANT=1: nothing constant)", None)
timed("feature switches only", pick(FEATURES))
timed("tree-shape constants only", pick(SHAPE))
timed("features + rounds / depth / seg steps", pick(FEATURES + ("n_rounds", "max_depth", "seg_steps")))
timed("features + row-layout level masks", pick(FEATURES + ("row_cross_levels", "row_ppull_levels", "row_pull_counts")))
timed("everything (the product's build)", assign)
