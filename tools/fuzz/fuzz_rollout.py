#!/usr/bin/env python3
"""[round 4] Fuzz of the controlled / recorded rollouts (jxs_rollout_controlled, jxs_rollout_recorded) in the host
emulation against the oracle stepping with tau[k]: random trees, Euler (fused, or unfused with several point chunks /
disabled points), RungeKutta4 and the rigid contact models (one launch per step).  usage: python tools/fuzz/fuzz_rollout.py [seed] [trials]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import emul_binding as eb  # noqa: E402
import helpers  # noqa: E402
import jaxsim_amd as ja  # noqa: E402
import oracle  # noqa: E402
from jaxsim_amd import robots  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nfail, worst = 0, {}
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n_links = int(rng.integers(2, 30))
    seed = 9000 + trial
    ncl = int(rng.integers(0, 4))
    cl = tuple(sorted(set(int(v) for v in rng.integers(0, n_links, size=ncl))))
    model = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=False, seed=seed, max_back=int(rng.integers(1, 4)), collision_links=cl))
    kind = ["euler", "rk4", "rigid", "disabled"][int(rng.integers(0, 4))]
    npts = 8 * len(cl)
    if kind == "rk4":
        model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4, contact_params=ja.SoftContactsParams.build(K=2e4, D=60.0, mu=0.6))
    if kind == "rigid":
        if npts == 0:
            continue
        model = helpers.rigid_model(model, list(range(min(npts, 4))), K=1e4, D=2e2, build=dict(solver_options={"solver_tol": 1e-9}))
    if kind == "disabled":
        if npts < 2:
            continue
        model = helpers.enable_points(model, sorted(int(v) for v in rng.choice(npts, size=npts // 2, replace=False)))
    n = model.dofs()
    K, N = int(rng.integers(2, 6)), 3
    seq = bool(rng.integers(0, 2))
    d = oracle.random_model_data(model, batch_size=N, seed=seed, base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.4)), base_rpy_bounds=((-0.4, -0.4, -3), (0.4, 0.4, 3)))
    tau = rng.uniform(-3, 3, size=(K, N, n))
    if not seq:
        tau[:] = tau[0]
    blk = helpers.odata_to_block(model, d)
    arg = np.ascontiguousarray(tau.transpose(0, 2, 1).reshape(K * n, N)) if seq else np.ascontiguousarray(tau[0].T)
    try:
        final, states = eb.run(model, eb.MODE_STEP, blk, tau=arg if n else None, n_steps=K, tau_seq=seq and n > 0, record=True, force_repr=2)
    except RuntimeError as ex:
        print("refused", trial, kind, str(ex)[:80])
        continue
    # two criteria: every recorded state equals the single-step launches BITWISE (the rollout machinery adds nothing),
    # and stays near the oracle (a switch of a contact's stick / slip state between the two amplifies 1e-13 to 1e-8
    # within one step: the first criterion is the sharp one)
    ref, cur, e, same = d, blk, 0.0, True
    for k in range(K):
        try:
            ref = oracle.step(model, ref, joint_force_references=tau[k])
        except np.linalg.LinAlgError:  # (the oracle's own Cholesky gave up on a diverged state)
            break
        truth = helpers.odata_to_block(model, ref)
        ok_env = np.isfinite(truth).all(axis=0) & (np.abs(truth).max(axis=0) < 1e3)  # (an environment that blows up -- stiff contacts on light random links -- is compared bitwise only)
        if ok_env.any():
            e = max(e, helpers.rel_err(states[k][:, ok_env], truth[:, ok_env]))
        cur = eb.run(model, eb.MODE_STEP, cur, tau=np.ascontiguousarray(tau[k].T) if n else None, force_repr=2)
        same = same and np.array_equal(cur, states[k], equal_nan=True)
    same = same and np.array_equal(states[-1], final, equal_nan=True)
    worst[kind] = max(worst.get(kind, 0), e)
    if not (e < 1e-5) or not same:
        nfail += 1
        print("FAIL", trial, n_links, cl, kind, "seq" if seq else "const", K, "%.2e" % e, same)
print("fails", nfail, {k: "%.1e" % v for k, v in worst.items()})
