#!/usr/bin/env python3
"""[round 4] Fuzz campaign in the HOST EMULATION of the kernel core (tests/emul: the kernel sources compiled for the CPU with a
lane-array backend) against the oracle, on random trees (jaxsim_amd/robots.py chain_urdf: 1 to 40 links, serial to
bushy, fixed / floating base, collision boxes on random links).  No GPU.  usage: python tools/fuzz/fuzz_rigid.py [seed] [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nfail = 0; worst = {}; compared = refused = oracle_failed = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n_links = int(rng.integers(1, 25)); seed = 5000 + trial; fixed = bool(rng.integers(0, 4) == 0) and n_links > 1
    mb = int(rng.integers(1, 4))
    ncl = int(rng.integers(1, 4))
    cl = tuple(sorted(set(int(v) for v in rng.integers(0, n_links, size=ncl))))
    base = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=mb, collision_links=cl,
                                                                         base_offset=(0.0, 0.0, 0.0) if trial % 4 else (0.1, -0.2, 0.5)))  # (a base-link offset is refused by the rigid models: one fixed tree in four keeps it, to cover the refusal)
    npts = 8 * len(cl)
    k = int(rng.integers(1, npts + 1))
    idx = sorted(int(v) for v in rng.choice(npts, size=k, replace=False))
    kind = ['relaxed', 'rigid'][int(rng.integers(0, 2))]
    integ = int(rng.integers(0, 3) == 0)
    try:
        model = helpers.relaxed_model(base, idx, mu=float(rng.choice([0.005, 0.3, 0.8]))) if kind == 'relaxed' else helpers.rigid_model(base, idx, K=1e4, D=1e2, build=dict(solver_options={"solver_tol": 1e-9}))
        if integ:
            model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4)
        N = 4
        d = oracle.random_model_data(model, batch_size=N, seed=seed, base_pos_bounds=((-1,-1,0.0),(1,1,0.3)), base_rpy_bounds=((-0.4,-0.4,-3),(0.4,0.4,3)))
        blk = helpers.odata_to_block(model, d)
        ref = oracle.step(model, d)
        out = eb.run(model, eb.MODE_STEP, blk)
    except RuntimeError as ex:
        refused += 1; print('refused', trial, n_links, fixed, cl, len(idx), kind, str(ex)[:80]); continue
    except np.linalg.LinAlgError as ex:
        oracle_failed += 1; print('oracle failed', trial, kind); continue
    compared += 1
    e = helpers.rel_err(out, helpers.odata_to_block(model, ref))
    key = (kind, 'rk4' if integ else 'euler')
    worst[key] = max(worst.get(key, 0), e)
    tol = 1e-8 if kind == 'relaxed' else 1e-5
    if not (e < tol):
        nfail += 1; print('FAIL', trial, 'nL', n_links, 'fixed', fixed, 'mb', mb, 'coll', cl, 'points', idx, key, '%.2e'%e)
print('fails', nfail, 'compared', compared, 'refused', refused, 'oracle_failed', oracle_failed, worst)
