#!/usr/bin/env python3
"""[round 4] Fuzz campaign in the HOST EMULATION of the kernel core (tests/emul: the kernel sources compiled for the CPU with a
lane-array backend) against the oracle, on random trees (jaxsim_amd/robots.py chain_urdf: 1 to 40 links, serial to
bushy, fixed / floating base, collision boxes on random links).  No GPU.  usage: python tools/fuzz/fuzz_step.py [seed] [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
from oracle import VelRepr
REPR_CODE = {VelRepr.Inertial: 0, VelRepr.Body: 1, VelRepr.Mixed: 2}
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nfail = 0; worst = {}
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n_links = int(rng.integers(1, 41)); seed = 3000 + trial; fixed = bool(rng.integers(0, 2)) and n_links > 1
    mb = int(rng.integers(1, 5))
    ncl = int(rng.integers(0, 5))
    cl = tuple(sorted(set(int(v) for v in rng.integers(0, n_links, size=ncl))))
    urdf = robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=mb, collision_links=cl)
    try:
        model = ja.JaxSimModel.build_from_model_description(urdf)
    except Exception as e:
        print('build failed', trial, n_links, fixed, cl, repr(e)[:100]); continue
    if rng.integers(0, 3) == 0:  # a tilted ground plane (PlaneTerrain)
        model = helpers.with_params(model, terrain=ja.PlaneTerrain.build(height=float(rng.uniform(-0.05, 0.05)), normal=[float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), 1.0]))
    if rng.integers(0, 4) == 0 and model.kin_dyn_parameters.number_of_collidable_points() > 1:  # soft-contact exponents other than 1/2, a different friction
        model = helpers.with_params(model, contact_params=ja.SoftContactsParams.build(K=5e5, D=1.5e3, mu=0.9, p=0.7, q=0.3))
    integ = int(rng.integers(0, 2))
    if integ:
        model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4, contact_params=ja.SoftContactsParams.build(K=2e4, D=60.0, mu=0.6))
    N = 5
    rep = [VelRepr.Inertial, VelRepr.Body, VelRepr.Mixed][int(rng.integers(0,3))]
    d = oracle.random_model_data(model, batch_size=N, seed=seed, velocity_representation=rep, base_pos_bounds=((-1,-1,0.0),(1,1,0.4)), base_rpy_bounds=((-0.5,-0.5,-3),(0.5,0.5,3)))
    d.tangential_deformation[:] = 1e-3 * rng.normal(size=d.tangential_deformation.shape)
    tau, f = helpers.random_inputs(model, N, seed, np.float64)
    try:
        blk = helpers.odata_to_block(model, d)
        kw = dict(tau=tau.T, link_forces=f.reshape(N, -1).T, force_repr=REPR_CODE[rep])
        ref = oracle.step(model, d, link_forces=f, joint_force_references=tau)
        out = eb.run(model, eb.MODE_STEP, blk, **kw)
        e = helpers.rel_err(out, helpers.odata_to_block(model, ref))
        acc = eb.run(model, eb.MODE_FD, blk, tau=tau.T, link_forces=f.reshape(N, -1).T, force_repr=REPR_CODE[rep])
    except RuntimeError as ex:
        print('refused', trial, n_links, fixed, cl, 'rk4' if integ else 'euler', str(ex)[:90]); continue
    key = ('rk4' if integ else 'euler')
    worst[key] = max(worst.get(key, 0), e)
    if not (e < 1e-8):
        nfail += 1; print('FAIL', trial, 'nL', n_links, 'fixed', fixed, 'mb', mb, 'coll', cl, key, rep, '%.2e'%e)
print('fails', nfail, worst)
