#!/usr/bin/env python3
"""[round 4] Fuzz of the link-space contact solve (host emulation against the oracle): random floating trees with the two
collision boxes on random links SIX OR MORE joints apart (the eligibility rule of csrc/jxs_pack.h), RelaxedRigidContacts
fp64 / fp32 and RigidContacts fp64.  usage: python tools/fuzz/fuzz_linkspace.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
rng = np.random.default_rng(91)
worst = {}; n_ls = 0
for trial in range(120):
    n_links = int(rng.integers(8, 25)); seed = 900 + trial
    a, b = sorted(int(v) for v in rng.choice(np.arange(0, n_links), size=2, replace=False))
    base = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=False, seed=seed, max_back=int(rng.integers(1,3)), collision_links=(a, b)))
    model0 = helpers.relaxed_model(base, list(range(16)), mu=0.5)
    if helpers.contact_link_separation(model0) < 6: continue
    n_ls += 1
    for kind, dtype, tol in (("relaxed", np.float64, 1e-9), ("relaxed", np.float32, 2e-3), ("rigid", np.float64, 5e-5)):
        model = model0 if kind=="relaxed" else helpers.rigid_model(base, list(range(16)), K=1e4, D=1e2)
        d = oracle.random_model_data(model, batch_size=6, seed=seed, dtype=dtype, base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.3)), base_rpy_bounds=((-0.4, -0.4, -3), (0.4, 0.4, 3)))
        blk = helpers.odata_to_block(model, d); truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d) if dtype==np.float32 else d))
        out = eb.run(model, eb.MODE_STEP, blk)
        e = helpers.rel_err(out, truth); key=(kind, np.dtype(dtype).name)
        worst[key]=max(worst.get(key,0), e)
        if not (e < tol): print('FAIL', trial, n_links, (a,b), key, '%.2e'%e)
print(n_ls, 'link-space trees;', worst)
