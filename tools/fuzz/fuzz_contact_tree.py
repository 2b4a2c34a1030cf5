#!/usr/bin/env python3
"""[round 5] Fuzz of the contact solve IN THE TREE (csrc/jxs_rigid.inc ta_*; host emulation of the kernel sources against
the oracle): random trees with collision boxes on ONE TO FOUR random links -- neighbours, far apart, on the base -- a
third of the trees with every joint axis parallel (planar mechanisms: the model class whose link-space matrix is
singular in every configuration), a third with axis-aligned joints (runs of parallel axes, as in real
robots), floating and fixed base (the fixed ones without a base-link offset, so that they are compared and not refused).
RelaxedRigidContacts fp64 / fp32 (mu = 0.5) and RigidContacts fp64.  No GPU.
usage: python tools/fuzz/fuzz_contact_tree.py [seed] [trials] [states]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots, specialize
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 91)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N = int(sys.argv[3]) if len(sys.argv) > 3 else 6
worst, count, nfail, refused, oracle_failed = {}, {}, 0, 0, 0
for trial in range(trials):
    n_links = int(rng.integers(2, 25)); seed = 7000 + trial
    fixed = bool(rng.integers(0, 5) == 0)
    axes = [None, "all", "aligned"][trial % 3]
    ncl = int(rng.integers(1, 5))
    cl = tuple(sorted(set(int(v) for v in rng.choice(np.arange(0, n_links), size=min(ncl, n_links), replace=False))))
    if fixed and cl == (0,):
        cl = (n_links - 1,)
    base = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=int(rng.integers(1, 4)),
                                                                         collision_links=cl, parallel_axes=axes, base_offset=(0.0, 0.0, 0.0)))
    idx = list(range(8 * len(cl)))
    for kind, dtype, tol in (("relaxed", np.float64, 1e-9), ("relaxed", np.float32, 2e-3), ("rigid", np.float64, 5e-5)):
        key = (kind, np.dtype(dtype).name)
        # (RigidContacts takes the tree by default only where the triangles do not fit the LDS; the campaign runs it everywhere)
        if kind == "rigid" and os.environ.get("JXS_DISABLE_CT_TREE") is None:
            os.environ["JXS_CT_TREE_RIGID"] = "1"
        else:
            os.environ.pop("JXS_CT_TREE_RIGID", None)
        try:
            model = helpers.relaxed_model(base, idx, mu=0.5) if kind == "relaxed" else helpers.rigid_model(base, idx, K=1e4, D=1e2)
            assert ("P.ct_tree=1" in specialize.spec(model, dtype, specialize.MODE_STEP_RIGID)) == (os.environ.get("JXS_DISABLE_CT_TREE") is None), (trial, key)
            d = oracle.random_model_data(model, batch_size=N, seed=seed, dtype=dtype, base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.3)), base_rpy_bounds=((-0.4, -0.4, -3), (0.4, 0.4, 3)))
            blk = helpers.odata_to_block(model, d)
            truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d, model) if dtype == np.float32 else d))
            out = eb.run(model, eb.MODE_STEP, blk)
        except RuntimeError as ex:
            refused += 1; print('refused', trial, n_links, fixed, cl, key, str(ex)[:90]); continue
        except np.linalg.LinAlgError:
            oracle_failed += 1; print('oracle failed', trial, key); continue
        e = helpers.rel_err(out, truth)
        worst[key] = max(worst.get(key, 0), e); count[key] = count.get(key, 0) + 1
        if not (e < tol):
            nfail += 1; print('FAIL', trial, 'nL', n_links, 'fixed', fixed, 'axes', axes, 'coll', cl, key, '%.2e' % e)
print('fails', nfail, 'compared', sum(count.values()), 'refused', refused, 'oracle_failed', oracle_failed, {k: (count[k], float('%.2e' % worst[k])) for k in worst})
