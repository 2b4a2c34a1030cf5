#!/usr/bin/env python3
"""[round 4] Fuzz campaign in the HOST EMULATION of the kernel core (tests/emul: the kernel sources compiled for the CPU with a
lane-array backend) against the oracle, on random trees (jaxsim_amd/robots.py chain_urdf: 1 to 40 links, serial to
bushy, fixed / floating base, collision boxes on random links).  No GPU.  usage: python tools/fuzz/fuzz_query.py [seed] [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emul_binding as eb, helpers, oracle
import jaxsim_amd as ja
from jaxsim_amd import robots
from oracle import VelRepr, refrigid
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nfail = 0; worst = {}
def rec(key, e, tol, info):
    global nfail
    worst[key] = max(worst.get(key, 0), e)
    if not (e < tol):
        nfail += 1; print('FAIL', key, '%.2e'%e, info)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n_links = int(rng.integers(1, 41)); seed = 7000 + trial; fixed = bool(rng.integers(0, 2)) and n_links > 1
    mb = int(rng.integers(1, 5))
    model = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=mb))
    N = 3; nL = model.number_of_links(); nv = 6 + model.dofs()
    info = (trial, n_links, fixed, mb)
    d = oracle.random_model_data(model, batch_size=N, seed=seed, velocity_representation=VelRepr.Inertial)
    blk = helpers.odata_to_block(model, d)
    tau, f = helpers.random_inputs(model, N, seed, np.float64)
    vd, sdd = oracle.forward_dynamics_aba(model, d, joint_forces=tau, link_forces=f)
    out = eb.run(model, eb.MODE_FD, blk, tau=tau.T, link_forces=f.reshape(N, -1).T, force_repr=0)
    rec('FD', helpers.rel_err(out.T, np.concatenate([vd, sdd], -1)), 1e-8, info)
    acc = rng.uniform(-2, 2, size=(N, nv))
    fB, tq = oracle.inverse_dynamics(model, d, joint_accelerations=acc[:, 6:], base_acceleration=acc[:, :6], link_forces=f)
    out = eb.run(model, eb.MODE_ID, blk, link_forces=f.reshape(N, -1).T, force_repr=0, in_acc=acc.T)
    ref = np.concatenate([fB, tq], -1)
    if not model.floating_base(): ref[:, :6] = out.T[:, :6]
    rec('ID', float(np.abs(out.T - ref).max()) / max(1.0, float(np.abs(ref).max())), 1e-9, info)
    H, V = eb.run(model, eb.MODE_KIN, blk)
    dc = d.update_caches(model)
    rec('KIN_H', helpers.rel_err(H.T.reshape(N, nL, 3, 4), dc.link_transforms[:, :, :3, :]), 1e-10, info)
    rec('KIN_V', helpers.rel_err(V.T.reshape(N, nL, 6), dc.link_velocities), 1e-10, info)
    dm = oracle.random_model_data(model, batch_size=N, seed=seed)
    bm = helpers.odata_to_block(model, dm)
    M = refrigid.free_floating_mass_matrix_mixed(model, dm)
    out = eb.run(model, eb.MODE_CRBA, bm).T.reshape(N, nv, nv)
    rec('CRBA', np.abs(out - M).max() / max(1.0, np.abs(M).max()), 1e-10, info)
    out = eb.run(model, eb.MODE_MINV, bm).T.reshape(N, nv, nv)
    rec('MINV', np.abs(M @ out - np.eye(nv)).max(), 1e-6, info)
    JJ, BH = eb.run(model, eb.MODE_JAC, bm)
    J = JJ[: 6 * nv].T.reshape(N, 6, nv); Jd = JJ[6 * nv :].T.reshape(N, 6, nv)
    J_ref, BH_ref = refrigid.jacobian_full_doubly_left(model, dm.joint_positions)
    Jd_ref = refrigid.jacobian_derivative_full_doubly_left(model, dm.joint_positions, dm.joint_velocities)
    rec('JAC', helpers.rel_err(J, J_ref), 1e-10, info); rec('JACD', helpers.rel_err(Jd, Jd_ref), 1e-10, info)
    g = eb.run(model, eb.MODE_GRAV, bm).T[:, 6:]
    if g.size: rec('GRAV', float(np.abs(g - oracle.free_floating_gravity_forces(model, dm)[:, 6:]).max()) / max(1.0, float(np.abs(g).max())), 1e-10, info)
print('fails', nfail, {k: '%.1e'%v for k, v in worst.items()})
