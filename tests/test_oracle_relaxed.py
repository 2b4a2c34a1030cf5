"""Oracle checks for the RelaxedRigidContacts restatement (oracle/refrelaxed.py): the known answer
the reference's own test holds for this model (tests/test_simulations.py:295-346) and the
identities of its regularised linear system.  No GPU, no kernel code."""

import numpy as np
import pytest

import helpers
import oracle
from oracle import refrelaxed as rx
from oracle import refstep as rs
from oracle import VelRepr


def test_box_settles_known_answer(models):
    """reference tests/test_simulations.py:295-346: box dropped from z = 0.2 with the default
    parameters; after 1 s x, y are unchanged (atol 1e-5) and z = box_height / 2 (atol 1e-4)."""
    model = helpers.relaxed_model(models("box"), [0, 1, 2, 3], build=dict(solver_options={"tol": 1e-3}))
    d = oracle.OracleData.build(model, base_position=[0.0, 0.0, 0.2], velocity_representation=VelRepr.Inertial)
    for _ in range(1000):
        d = oracle.step(model, d)
    np.testing.assert_allclose(d.base_position[0, :2], [0.0, 0.0], atol=1e-5)
    assert d.base_position[0, 2] == pytest.approx(0.05, abs=1e-4)
    # at rest the four points carry the weight
    W_f, _ = rx.compute_contact_forces(model, d)
    assert W_f[0, :, 2].sum() == pytest.approx(9.80665 * float(np.sum(model.kin_dyn_parameters.link_mass)), rel=1e-3)


def test_regularizers_hand_values(models):
    """Impedance curve of ``_regularizers`` (relaxed_rigid.py:540-591) at hand-computed points of the
    default parameters (width 1e-3, midpoint 0.5, power 2, d in [0.9, 0.95])."""
    model = helpers.relaxed_model(models("box"), [0, 1, 2, 3], mu=0.5)
    cp = model.contact_params
    m = float(model.kin_dyn_parameters.link_mass[0])
    # penetrations: none, width / 4 (first branch), 3 width / 4 (second branch), 2 width (saturated)
    delta = np.array([0.0, 0.25e-3, 0.75e-3, 2e-3])
    pos = np.zeros((1, 4, 3))
    pos[0, :, 2] = -delta
    vel = np.tile(np.array([0.1, -0.2, 0.3]), (1, 4, 1))
    a_ref, r = rx.regularizers(model, pos, vel, cp)
    a_ref, r = a_ref.reshape(4, 3), r.reshape(4, 3)
    xi_z = np.array([np.nan, 0.9 + 0.05 * (2 * 0.25**2), 0.9 + 0.05 * (1 - 2 * 0.25**2), 0.95])
    K, D = 1 / (0.95 * 0.02 * 1.0) ** 2, 2 / (0.95 * 0.02)
    coef = 2 * 0.25 * 1.25 / m
    assert np.all(a_ref[0] == 0) and np.all(r[0] == 0)  # inactive point
    for c in (1, 2, 3):
        np.testing.assert_allclose(r[c, :2], coef * 0.1 / (0.9 + 1e-12), rtol=1e-12)  # tangential axes: d_min
        np.testing.assert_allclose(r[c, 2], coef * (1 - xi_z[c]) / (xi_z[c] + 1e-12), rtol=1e-12)
        np.testing.assert_allclose(a_ref[c, :2], -D * vel[0, c, :2], rtol=1e-12)
        np.testing.assert_allclose(a_ref[c, 2], -(D * 0.3 + K * xi_z[c] * (-delta[c])), rtol=1e-12)


@pytest.mark.parametrize("name,idx,params", [
    ("box", [0, 1, 2, 3], dict()),
    ("anymal", helpers.ANYMAL_FEET_16, dict(mu=0.5)),
    ("chain9f", [0, 1, 2, 3, 8, 9], dict(mu=0.8)),
])  # fmt: skip
def test_forces_solve_the_regularised_system(models, name, idx, params):
    """``A x = -b`` on the active rows, zero force on inactive points (relaxed_rigid.py:383-397,459-465);
    the contact acceleration the forces produce is the regularised reference acceleration:
    ``a_free + G x = a_ref - R x``."""
    model = helpers.relaxed_model(models(name), idx, **params)
    d = models.random_data(name, 8, seed=5)
    tau, f = helpers.random_inputs(model, 8, 7, np.float64)
    W_f, aux = rx.compute_contact_forces(model, d, link_forces=f, joint_torques=tau)
    pb, x = aux["problem"], aux["forces"].reshape(8, -1)
    assert pb["active"].any() and (~pb["active"]).any()
    res = np.einsum("nij,nj->ni", pb["A"], x) + pb["b"]
    scale = np.abs(pb["b"]).max()
    assert np.abs(res).max() < 1e-9 * scale
    assert np.all(aux["forces"][~pb["active"]] == 0)
    np.testing.assert_allclose(W_f[..., 3:], np.cross(pb["position"], W_f[..., :3]), atol=1e-12)


def test_null_space_forces_carry_no_generalised_force(models):
    """Several points on one rigid body: ``J M^-1 J^T`` is singular and force components in its null
    space produce no generalised force (they matter only through the non-uniform regulariser, see the
    header of oracle/refrelaxed.py)."""
    model = helpers.relaxed_model(models("box"), [0, 1, 2, 3], mu=0.5)
    d = models.random_data("box", 8, seed=5)
    _, aux = rx.compute_contact_forces(model, d)
    pb = aux["problem"]
    for e in range(8):
        act = np.repeat(pb["active"][e], 3)
        if act.sum() < 9:
            continue
        A, J = pb["A"][e][np.ix_(act, act)], pb["J_lin"][e][act]
        w, V = np.linalg.eigh(A - np.diag(pb["r"][e][act]))
        null = V[:, w < 1e-9 * w.max()]
        assert null.shape[1] >= 3  # 9+ rows, rank <= 6
        assert np.abs(J.T @ null).max() < 1e-6
        break
    else:
        pytest.fail("no sample with three active points")


def test_dispatch_has_no_impact_stage(models):
    """``update_velocity_after_impact`` is the identity for this model (relaxed_rigid.py:265-281)."""
    model = helpers.relaxed_model(models("box"), [0, 1, 2, 3])
    assert rs.is_relaxed_rigid_contact_model(model) and not rs.is_rigid_contact_model(model)
