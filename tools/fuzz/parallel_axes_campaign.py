#!/usr/bin/env python3
"""[round 5, VERDICT r4 weak #1] The model class round 4's link-space contact solve got wrong, at scale, in the host
emulation of the kernel sources against the fp64 oracle: a planar biped (six parallel pitch joints between the feet)
and planar serial chains (8 .. 14 links, every joint revolute about x), RelaxedRigidContacts (mu = 0.5) in fp32 and
fp64, RigidContacts in fp64; random and standing states.  usage: python tools/fuzz/parallel_axes_campaign.py [states per model]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
per_model = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
models = [("planar_biped", ja.JaxSimModel.build_from_model_description(robots.planar_biped_urdf()), ((-1, -1, 0.78), (1, 1, 0.95)))]
for n in range(8, 15):
    models.append((f"planar_chain{n}", ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n, fixed_base=False, seed=40 + n, max_back=1, parallel_axes="all")), ((-1, -1, 0.0), (1, 1, 0.3))))
tot = {}
for name, base, bounds in models:
    for kind, dtype in (("relaxed", np.float32), ("relaxed", np.float64), ("rigid", np.float64)):
        model = helpers.relaxed_model(base, range(16), mu=0.5) if kind == "relaxed" else helpers.rigid_model(base, range(16), K=1e4, D=1e2)
        # (RigidContacts takes the tree by default only where the triangles do not fit the LDS; the campaign runs it everywhere)
        if kind == "rigid" and os.environ.get("JXS_DISABLE_CT_TREE") is None:
            os.environ["JXS_CT_TREE_RIGID"] = "1"
        else:
            os.environ.pop("JXS_CT_TREE_RIGID", None)
        errs = []
        for part, seed in (("random", 0), ("standing", 1)):
            N = per_model // 2
            if part == "random":
                d = oracle.random_model_data(model, batch_size=N, seed=seed, dtype=dtype, base_pos_bounds=bounds, base_rpy_bounds=((-0.3, -0.3, -3), (0.3, 0.3, 3)))
            else:
                d = helpers.standing_data(model, N, seed=seed, dtype=dtype, noise=0.3)
            out = eb.run(model, eb.MODE_STEP, helpers.odata_to_block(model, d))
            ref = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d, model) if dtype == np.float32 else d))
            errs.append((np.abs(out - ref) / np.maximum(1, np.abs(ref))).max(axis=0))
        e = np.concatenate(errs)
        key = (kind, np.dtype(dtype).name)
        t = tot.setdefault(key, [0, 0.0, 0, 0])
        gate = 3e-4 if dtype == np.float32 else (1e-9 if kind == "relaxed" else 5e-5)
        t[0] += e.size; t[1] = max(t[1], float(e.max())); t[2] += int((e > gate).sum()); t[3] += int((e > 3e-3).sum())
        print(f"{name:16s} {kind:8s} {np.dtype(dtype).name}: {e.size} states, worst {e.max():.2e}, median {np.median(e):.1e}, p99 {np.quantile(e, 0.99):.1e}, above gate ({gate:g}) {int((e > gate).sum())}")
for k, t in tot.items():
    print("TOTAL", k, f"{t[0]} states, worst {t[1]:.2e}, above the gate {t[2]}, above 3e-3 {t[3]}")
