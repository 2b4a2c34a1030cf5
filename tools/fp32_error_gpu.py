#!/usr/bin/env python3
"""fp32 error of one GPU step against the fp64 oracle, per model (the states of tests/test_gpu_parity.py):
worst element and distribution over the environments.  Next to it the reference formulation evaluated in fp32
(the oracle run with float32 arrays).   python tools/fp32_error_gpu.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import helpers  # noqa: E402
import jaxsim_amd.api as js  # noqa: E402
import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
zoo = helpers.ModelZoo()
for name in ["box", "sphere", "pendulum", "double_pendulum", "cartpole", "chain5", "chain9f", "anymal", "icub", "icub16", "planar_biped", "planar10f"]:
    model = zoo(name)
    for seed in (4, 5):
        d = zoo.random_data(name, N, seed=seed, dtype=np.float32)
        tau, f = helpers.random_inputs(model, N, seed + 1, np.float32)
        truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d), link_forces=f.astype(np.float64), joint_force_references=tau.astype(np.float64)))
        g = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d))
        out = js.model.step(model, g, link_forces=f, joint_force_references=tau).state_block()
        ref32 = helpers.odata_to_block(model, oracle.step(model, d, link_forces=f, joint_force_references=tau))
        e = np.max(np.abs(out.astype(np.float64) - truth) / np.maximum(1.0, np.abs(truth)), axis=0)
        r = np.max(np.abs(ref32.astype(np.float64) - truth) / np.maximum(1.0, np.abs(truth)), axis=0)
        print(f"{name:16s} seed {seed}: GPU median {np.median(e):.1e} p99 {np.percentile(e, 99):.1e} worst {e.max():.1e} | "
              f"reference formulation in fp32: median {np.median(r):.1e} p99 {np.percentile(r, 99):.1e} worst {r.max():.1e}", flush=True)

# [round 5] the rigid contact models in fp32 (the cases of tests/test_gpu_parity.py), same truth: the fp64 oracle on the same state
from test_gpu_parity import RELAXED_CASES, RIGID_CASES  # noqa: E402

for kind, table, make in (("relaxed", RELAXED_CASES, helpers.relaxed_model), ("rigid", RIGID_CASES, helpers.rigid_model)):
    for key, (name, idx, params) in table.items():
        model = make(zoo(name), idx, **params)
        for seed in (4, 5):
            d = zoo.random_data(name, N, seed=seed, dtype=np.float32)
            try:
                truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d, model)))
                out = js.model.step(model, js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d))).state_block()
            except Exception as exc:  # (a singular oracle state, a refused model)
                print(f"{kind}/{key:14s} seed {seed}: skipped ({exc!r})"[:160], flush=True)
                continue
            e = np.max(np.abs(out.astype(np.float64) - truth) / np.maximum(1.0, np.abs(truth)), axis=0)
            print(f"{kind}/{key:14s} seed {seed}: GPU median {np.median(e):.1e} p99 {np.percentile(e, 99):.1e} worst {e.max():.1e}", flush=True)
