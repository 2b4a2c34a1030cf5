#!/usr/bin/env python3
"""[round 5] Second fuzz campaign of the contact solve in the tree (host emulation against the oracle): what the first one
(fuzz_contact_tree.py) leaves out -- up to EIGHT contact links, random subsets of the points (single points, pairs:
rank-deficient W), RungeKutta4, deep serial chains, and the chunked form (a random lane-group cap, so that the points go
through in two to eight chunks).  RelaxedRigidContacts fp64 / fp32 (mu 0.3 .. 0.8).  No GPU.
usage: python tools/fuzz/fuzz_contact_tree2.py [seed] [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 17)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst, count, nfail, refused, oracle_failed = {}, {}, 0, 0, 0
only = set(int(v) for v in os.environ["FUZZ_ONLY"].split(",")) if os.environ.get("FUZZ_ONLY") else None  # re-run single trials of a campaign (same random stream)
for trial in range(trials):
    n_links = int(rng.integers(2, 41)); seed = 9000 + trial
    fixed = bool(rng.integers(0, 5) == 0)
    axes = [None, "all", "aligned", None][trial % 4]
    ncl = int(rng.integers(1, min(8, n_links) + 1))
    cl = tuple(sorted(set(int(v) for v in rng.choice(np.arange(0, n_links), size=ncl, replace=False))))
    if fixed and cl == (0,):
        cl = (n_links - 1,)
    base = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=int(rng.integers(1, 4)),
                                                                         collision_links=cl, parallel_axes=axes, base_offset=(0.0, 0.0, 0.0)))
    npts = 8 * len(cl)
    k = int(rng.integers(1, npts + 1)) if trial % 2 else npts
    idx = sorted(int(v) for v in rng.choice(npts, size=k, replace=False))
    rk4 = bool(rng.integers(0, 4) == 0)
    cap = [None, None, "8", "16"][int(rng.integers(0, 4))]
    os.environ.pop("JXS_CT_CHUNK_LANES", None)
    if cap is not None and not rk4:
        os.environ["JXS_CT_CHUNK_LANES"] = cap
    for dtype, tol in ((np.float64, 1e-8), (np.float32, 3e-3)):
        key = ("rk4" if rk4 else "euler", "chunk" + str(cap) if (cap and not rk4) else "one", np.dtype(dtype).name)
        mu = float(rng.choice([0.3, 0.5, 0.8]))
        if only is not None and trial not in only:
            continue
        try:
            model = helpers.relaxed_model(base, idx, mu=mu)
            if rk4:
                model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4)
            d = oracle.random_model_data(model, batch_size=4, seed=seed, dtype=dtype, base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.3)), base_rpy_bounds=((-0.4, -0.4, -3), (0.4, 0.4, 3)))
            truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d, model) if dtype == np.float32 else d))
            out = eb.run(model, eb.MODE_STEP, helpers.odata_to_block(model, d))
        except RuntimeError as ex:
            refused += 1; print('refused', trial, n_links, fixed, cl, key, str(ex)[:90]); continue
        except np.linalg.LinAlgError:
            oracle_failed += 1; print('oracle failed', trial, key); continue
        e = helpers.rel_err(out, truth)
        worst[key] = max(worst.get(key, 0), e); count[key] = count.get(key, 0) + 1
        if not (e < tol):
            nfail += 1; print('FAIL', trial, 'nL', n_links, 'fixed', fixed, 'axes', axes, 'coll', cl, 'points', len(idx), key, '%.2e' % e)
os.environ.pop("JXS_CT_CHUNK_LANES", None)
print('fails', nfail, 'compared', sum(count.values()), 'refused', refused, 'oracle_failed', oracle_failed, {k: (count[k], float('%.2e' % worst[k])) for k in sorted(worst)})
