"""Oracle checks for the RigidContacts restatement (oracle/refrigid.py): the known answers and
identities the reference's own tests hold for this model, re-expressed
for the NumPy restatement.  No GPU, no kernel code."""

import dataclasses

import numpy as np
import pytest

import helpers
import oracle
from oracle import refrigid as rr
from oracle import refstep as rs
from oracle import VelRepr


def _mixed(d):
    return dataclasses.replace(d, velocity_representation=VelRepr.Mixed)


@pytest.mark.parametrize("name,idx", [("box", [0, 1, 2, 3]), ("anymal", helpers.ANYMAL_FEET_16), ("icub16", list(range(16)))])
def test_contact_jacobian_maps_velocity_to_point_velocity(models, name, idx):
    """``J nu`` = velocity of the collidable points (reference: tests/test_api_contact.py:59-102)."""
    model = helpers.rigid_model(models(name), idx)
    d = _mixed(models.random_data(name, 4, seed=3))
    J = rr.contact_jacobian_mixed(model, d)
    nu = d.generalized_velocity(VelRepr.Mixed)
    _, v = rs.collidable_points_pos_vel(model, link_transforms=d.link_transforms, link_velocities=d.link_velocities)
    np.testing.assert_allclose(np.einsum("ncij,nj->nci", J, nu)[..., :3], v, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,idx", [("box", [0, 1, 2, 3]), ("anymal", helpers.ANYMAL_FEET_4), ("chain9f", [0, 1, 8, 9])])
def test_free_contact_acceleration_is_the_point_acceleration(models, name, idx):
    """``Jdot nu + J nudot`` is W_pdd_C (rigid.py:503-521): finite difference of the point velocities
    along the free motion."""
    model = helpers.rigid_model(models(name), idx)
    d = models.random_data(name, 3, seed=5)
    pb = rr.rigid_problem(model, d)
    dm = _mixed(d)
    nu, nud, h = dm.generalized_velocity(VelRepr.Mixed), pb["nud_free"], 1e-7
    qd = oracle.refmath.quaternion_derivative(d.base_quaternion, nu[:, 3:6], False, K=0.0)
    d2 = rs.OracleData.build(
        model, base_position=d.base_position + h * nu[:, :3], base_quaternion=d.base_quaternion + h * qd,
        joint_positions=d.joint_positions + h * d.joint_velocities, base_linear_velocity=nu[:, :3] + h * nud[:, :3],
        base_angular_velocity=nu[:, 3:6] + h * nud[:, 3:6], joint_velocities=d.joint_velocities + h * nud[:, 6:],
        velocity_representation=VelRepr.Mixed)  # fmt: skip
    _, v1 = rs.collidable_points_pos_vel(model, link_transforms=d.link_transforms, link_velocities=d.link_velocities)
    _, v2 = rs.collidable_points_pos_vel(model, link_transforms=d2.link_transforms, link_velocities=d2.link_velocities)
    np.testing.assert_allclose(((v2 - v1) / h).reshape(3, -1), pb["a_free"], rtol=0, atol=2e-5)


def test_mass_matrix_inverse_and_delassus(models):
    model = helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_16)
    d = _mixed(models.random_data("anymal", 3, seed=7))
    M, Mi = rr.free_floating_mass_matrix_mixed(model, d), rr.free_floating_mass_matrix_inverse_mixed(model, d)
    np.testing.assert_allclose(M @ Mi, np.broadcast_to(np.eye(18), (3, 18, 18)), atol=1e-10)
    G = rr.rigid_problem(model, d)["delassus"]
    np.testing.assert_allclose(G, np.swapaxes(G, -1, -2), atol=1e-12)
    assert np.linalg.eigvalsh(G).min() > -1e-10
    # four coplanar corners of one rigid foot: 12 rows of rank <= 6 per foot
    assert np.linalg.matrix_rank(G[0][:12, :12], tol=1e-9) <= 6


def test_qp_solution_satisfies_kkt_and_both_forms_agree(models):
    rng = np.random.default_rng(0)
    for n_cp, n_inactive in ((4, 0), (4, 2), (8, 3)):
        A = rng.normal(size=(3 * n_cp, 3 * n_cp + 2))
        Q = A @ A.T / (3 * n_cp) + 1e-6 * np.eye(3 * n_cp)
        q = rng.normal(size=3 * n_cp) - np.tile([0, 0, 3.0], n_cp)
        inactive = np.zeros(n_cp, dtype=bool)
        inactive[:n_inactive] = True
        G = rr.ineq_constraint_matrix(inactive, 0.5, np.float64)
        x, s, z, it, ok = rr.solve_qp_pdip(Q, q, G, np.zeros(6 * n_cp), solver_tol=1e-10)
        assert ok and it < rr.QP_MAX_ITER
        assert np.max(G @ x) < 1e-8 and np.min(z) > -1e-12  # primal / dual feasibility
        assert np.max(np.abs(Q @ x + q + G.T @ z)) < 1e-8  # stationarity
        assert abs(z @ (G @ x)) < 1e-7  # complementarity
        np.testing.assert_allclose(x.reshape(n_cp, 3)[inactive], 0, atol=1e-8)
        # reduced statement (the one the HIP kernel solves): inactive points removed, 5 rows per point
        act = np.flatnonzero(~inactive)
        rows = (3 * act[:, None] + np.arange(3)).reshape(-1)
        Gr = rr.ineq_constraint_matrix(np.zeros(act.size, dtype=bool), 0.5, np.float64)
        Gr = Gr[np.arange(6 * act.size) % 6 != 5]
        xr, *_ = rr.solve_qp_pdip(Q[np.ix_(rows, rows)], q[rows], Gr, np.zeros(5 * act.size), solver_tol=1e-10)
        np.testing.assert_allclose(xr, x[rows], atol=1e-7)


def test_default_tolerance_forms_agree_to_solver_tol(models):
    """At the reference's solver_tol = 1e-3 the two statements of the QP stop at different iterates;
    the step results agree to the level that tolerance implies."""
    model = helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_16, K=1e4, D=1e2)
    d = models.random_data("anymal", 6, seed=5)
    a = helpers.odata_to_block(model, oracle.step(model, d))
    try:
        rr.REDUCED_QP = True
        b = helpers.odata_to_block(model, oracle.step(model, d))
    finally:
        rr.REDUCED_QP = False
    assert helpers.rel_err(b, a) < 1e-4


def test_impact_removes_the_velocity_of_the_points_in_contact(models):
    model = helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_4)
    d = models.random_data("anymal", 8, seed=5)
    out = rr.update_velocity_after_impact(model, d)
    p, v0 = rs.collidable_points_pos_vel(model, link_transforms=d.link_transforms, link_velocities=d.link_velocities)
    _, v1 = rs.collidable_points_pos_vel(model, link_transforms=out.link_transforms, link_velocities=out.link_velocities)
    active = p[..., 2] < 0
    assert active.any() and (~active.any(axis=1)).any()
    assert np.abs(v1[active]).max() < 1e-9 and np.abs(v0[active]).max() > 1e-2
    untouched = ~active.any(axis=1)  # no contact: the velocity is unchanged
    np.testing.assert_allclose(out.joint_velocities[untouched], d.joint_velocities[untouched], atol=1e-12)


def test_box_settles_on_rigid_ground(models):
    """reference: tests/test_simulations.py:245-292 -- box dropped from z = 0.2 with four corner points,
    K = 1e5, solver_tol = 1e-3, rests with (almost) no penetration at z = box_height / 2."""
    model = helpers.rigid_model(models("box"), [0, 1, 2, 3], build=dict(solver_options={"solver_tol": 1e-3}), K=1e5)
    d = oracle.OracleData.build(model, base_position=[0.0, 0.0, 0.2], velocity_representation=VelRepr.Inertial)
    for _ in range(400):
        d = oracle.step(model, d)
    np.testing.assert_allclose(d.base_position[0, :2], 0.0, atol=1e-9)
    assert d.base_position[0, 2] == pytest.approx(0.05, abs=1e-4)
    assert np.abs(d.base_linear_velocity).max() < 1e-3


def test_rk4fast_link_forces_do_not_depend_on_the_data_representation():
    """RungeKutta4Fast with a contact model without contact state and non-zero external link forces: the
    step of Mixed data with the forces given in Mixed equals the step of the same state held as Inertial data
    with the same wrenches converted to Inertial.  The reference hands the already inertial wrenches to a
    contact solve that re-reads them in the data's representation (api/integrators.py:175-187); the
    restatement -- and the kernel checked against it -- deliberately does not (oracle/refstep.py docstring)."""
    import jaxsim_amd as ja

    zoo = helpers.ModelZoo()
    rr.REDUCED_QP = True
    try:
        model = helpers.with_params(helpers.rigid_model(zoo("anymal"), helpers.ANYMAL_FEET_4), integrator=ja.IntegratorType.RungeKutta4Fast)
        N = 6
        d_mixed = zoo.random_data("anymal", N, seed=5)
        assert d_mixed.velocity_representation == VelRepr.Mixed
        tau, f_mixed = helpers.random_inputs(model, N, 7, np.float64)
        f_inertial = rs.other_representation_to_inertial(f_mixed, VelRepr.Mixed, d_mixed.link_transforms, is_force=True)
        assert np.abs(f_inertial - f_mixed).max() > 1e-3  # the two readings of the numbers differ
        d_inertial = dataclasses.replace(d_mixed, velocity_representation=VelRepr.Inertial).update_caches(model)
        a = oracle.step(model, d_mixed, link_forces=f_mixed, joint_force_references=tau)
        b = oracle.step(model, d_inertial, link_forces=f_inertial, joint_force_references=tau)
        assert helpers.rel_err(helpers.odata_to_block(model, a), helpers.odata_to_block(model, b)) < 1e-12
        # and the forces matter in this sample
        c = oracle.step(model, d_mixed, joint_force_references=tau)
        assert helpers.rel_err(helpers.odata_to_block(model, a), helpers.odata_to_block(model, c)) > 1e-6
    finally:
        rr.REDUCED_QP = False
