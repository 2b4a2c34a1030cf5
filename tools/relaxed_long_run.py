#!/usr/bin/env python3
"""[round 5] Long runs of the contact solve in the tree on the GPU: the humanoid (32 points), the quadruped (16 points) and
the 200-point quadruped with RelaxedRigidContacts (estimated parameters), standing states with noise, `steps` in-place steps:
the fraction of environments that stay finite and the solves that were discarded (jxs_solver_fault_counts) -- next to the
same run through the triangles (JXS_DISABLE_CT_TREE=1) where they exist.   python tools/relaxed_long_run.py [steps] [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import helpers  # noqa: E402
import jaxsim_amd as ja  # noqa: E402
import jaxsim_amd.api as js  # noqa: E402
from jaxsim_amd import robots  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
zoo = helpers.ModelZoo()
cases = [("humanoid 32 points", zoo("icub"), list(range(32))), ("quadruped 16 points", zoo("anymal"), helpers.ANYMAL_FEET_16),
         ("quadruped 200 points", ja.JaxSimModel.build_from_model_description(robots.anymal12_urdf(foot_shape="sphere")), list(range(200)))]
for name, base, idx in cases:
    model = helpers.relaxed_model(base, idx)
    model = helpers.with_params(model, contact_params=js.contact.estimate_good_contact_parameters(model))
    for dtype in (np.float32, np.float64):
        d = helpers.standing_data(model, N, seed=0, dtype=dtype, noise=0.05)
        data = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), 2)
        js.model.solver_fault_counts(model, dtype, reset=True)
        for _ in range(steps):
            data = js.model.step(model, data, inplace=True)
        blk = data.state_block()
        fin = np.isfinite(blk).all(axis=0)
        z = blk[2, fin]
        print(f"{name:22s} {np.dtype(dtype).name}: {steps} steps x {N} envs: finite {fin.mean():.5f}, discarded solves {js.model.solver_fault_counts(model, dtype)}, "
              f"base height of the finite ones min {z.min():.3f} median {np.median(z):.3f} max {z.max():.3f}, tree={os.environ.get('JXS_DISABLE_CT_TREE') is None}", flush=True)
