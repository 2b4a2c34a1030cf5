"""Pins of the oracle: the known-answer and self-consistency tests the reference's own
suite holds for the step path, re-expressed for the NumPy oracle.

The reference cannot run here (no JAX), so these analytic pins are what anchors the oracle;
``oracle/__init__.py`` states "parity unpinned by execution".  Tolerances are the reference's
defaults, rtol 1e-7 / atol 1e-9 in fp64 (``tests/utils.py:14-26``), unless noted.
"""

import dataclasses

import numpy as np
import pytest

import helpers
import jaxsim_amd as ja
import oracle
from oracle import VelRepr

RTOL, ATOL = 1e-7, 1e-9


def run_simulation(model, data, tf):
    for _ in range(int(round(tf / model.time_step))):
        data = oracle.step(model, data)
    return data


# reference: tests/test_simulations.py:194-242 (box fixture tests/conftest.py:207-243)
def test_box_settles_on_soft_ground(models):
    model = models("box")
    max_penetration = 0.001
    params = oracle.estimate_good_contact_parameters(
        model,
        number_of_active_collidable_points_steady_state=4,
        static_friction_coefficient=1.0,
        damping_ratio=1.0,
        max_penetration=max_penetration,
    )
    model = helpers.with_params(model, contact_params=ja.SoftContactsParams.build(**params))
    model = helpers.enable_points(model, [0, 1, 2, 3])
    assert int(np.sum(model.kin_dyn_parameters.contact_enabled)) == 4
    box_height = 0.1
    d0 = oracle.OracleData.build(model, base_position=[0.0, 0.0, box_height * 2], velocity_representation=VelRepr.Inertial)
    df = run_simulation(model, d0, tf=1.0)
    np.testing.assert_allclose(df.base_position[0, :2], d0.base_position[0, :2], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(df.base_position[0, 2] + max_penetration, box_height / 2, rtol=RTOL, atol=ATOL)
    # analytic steady state: 4 K delta^1.5 = m g with K = m g / 4 / delta_max^1.5
    assert np.isclose(params["K"], 1.0 * 9.81 / 4 / max_penetration**1.5)


# reference: tests/test_simulations.py:15-85 (hover under a gravity-cancelling wrench, 3 representations)
@pytest.mark.parametrize("rep", [VelRepr.Inertial, VelRepr.Body, VelRepr.Mixed])
def test_box_hovers_under_gravity_cancelling_wrench(models, rep):
    model = helpers.with_params(models("box"), kin_dyn_parameters=_strip_points(models("box")))
    d = oracle.OracleData.build(
        model,
        base_position=[0.3, -0.2, 0.5],
        base_quaternion=[0.9238795, 0.0, 0.3826834, 0.0],
        velocity_representation=rep,
    )
    mg = -model.gravity * model.total_mass()
    W_f = np.zeros((1, 1, 6))
    W_f[0, 0, 2] = mg  # pure force through the CoM (link origin)
    # express the world-aligned force at the CoM in the data's representation
    f_rep = oracle.inertial_to_other_representation(
        np.concatenate([W_f[..., :3], np.cross(d.base_position[:, None], W_f[..., :3])], -1),
        rep, d.link_transforms, is_force=True,
    )  # fmt: skip
    p0, q0 = d.base_position.copy(), d.base_quaternion.copy()
    for _ in range(500):
        d = oracle.step(model, d, link_forces=f_rep)
    np.testing.assert_allclose(d.base_position, p0, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(d.base_quaternion, q0, rtol=RTOL, atol=ATOL)


# reference: tests/test_simulations.py:88-167 (zero gravity, constant force -> p0 + 1/2 f/m t^2, atol 1e-3)
def test_box_constant_force_zero_gravity(models):
    model = helpers.with_params(models("box"), kin_dyn_parameters=_strip_points(models("box")), gravity=0.0)
    d = oracle.OracleData.build(model, base_position=[0, 0, 1.0], velocity_representation=VelRepr.Inertial)
    f = np.zeros((1, 1, 6))
    f[0, 0, :3] = [1.0, -2.0, 0.5]
    # inertial-fixed wrench of a force applied at the (moving) CoM: keep the moment consistent
    n_steps = 10
    p0 = d.base_position.copy()
    for _ in range(n_steps):
        W_f = np.concatenate([f[..., :3], np.cross(d.base_position[:, None], f[..., :3])], -1)
        d = oracle.step(model, d, link_forces=W_f)
    t = n_steps * model.time_step
    np.testing.assert_allclose(d.base_position, p0 + 0.5 * f[0, 0, :3] / model.total_mass() * t**2, atol=1e-3)
    # semi-implicit Euler closed form: p_k = p0 + dt^2 a k (k+1)/2
    a = f[0, 0, :3] / model.total_mass()
    np.testing.assert_allclose(
        d.base_position[0], p0[0] + model.time_step**2 * a * n_steps * (n_steps + 1) / 2, rtol=RTOL, atol=ATOL
    )


def test_free_fall_closed_form(models):
    model = helpers.with_params(models("box"), kin_dyn_parameters=_strip_points(models("box")))
    d = oracle.OracleData.build(model, base_position=[0, 0, 10.0], base_linear_velocity=[0.5, 0, 0])
    k = 200
    for _ in range(k):
        d = oracle.step(model, d)
    dt = model.time_step
    np.testing.assert_allclose(d.base_position[0, 2], 10.0 + model.gravity * dt * dt * k * (k + 1) / 2, rtol=RTOL)
    np.testing.assert_allclose(d.base_position[0, 0], 0.5 * k * dt, rtol=RTOL)


# reference: tests/test_actuation.py:11-48
def test_tn_curve(models):
    model = helpers.with_params(
        models("pendulum"), actuation_params=ja.ActuationParams(torque_max=10.0, omega_th=1.0, omega_max=2.0)
    )
    def torque(w, tau_ref=30.0):
        d = oracle.OracleData.build(model, joint_velocities=[w])
        return oracle.compute_resultant_torques(model, d, joint_force_references=np.array([[tau_ref]]))[0, 0]

    assert torque(0.5) == pytest.approx(10.0)
    assert 0.0 < torque(1.5) < 30.0 and torque(1.5) == pytest.approx(5.0)
    assert torque(2.5) == pytest.approx(0.0)
    assert torque(-1.5, -30.0) == pytest.approx(-5.0)


def test_joint_limit_and_friction_torques(models):
    model = models("pendulum")
    kdp = dataclasses.replace(
        model.kin_dyn_parameters,
        position_limits_min=np.array([-1.0]), position_limits_max=np.array([1.0]),
        position_limit_spring=np.array([75.0]), position_limit_damper=np.array([0.1]),
        friction_static=np.array([0.2]), friction_viscous=np.array([0.3]),
    )  # fmt: skip
    model = helpers.with_params(model, kin_dyn_parameters=kdp)
    d = oracle.OracleData.build(model, joint_positions=[1.2], joint_velocities=[0.5])
    tau = oracle.compute_resultant_torques(model, d)[0, 0]
    tau_pl = -75.0 * 0.2
    tau_pl = tau_pl - tau_pl * 0.1 * 0.5  # jnp.positive is the identity (quirk 1)
    assert tau == pytest.approx(tau_pl - (0.2 + 0.3 * 0.5))


# reference: tests/test_api_model.py:551-577 (RNEA(ABA(tau, f)) = tau, base wrench 0)
@pytest.mark.parametrize("name", ["cartpole", "chain5", "chain9f", "anymal", "icub"])
@pytest.mark.parametrize("rep", [VelRepr.Inertial, VelRepr.Body, VelRepr.Mixed])
def test_aba_rnea_round_trip(models, name, rep):
    model = models(name)
    N = 4
    d = oracle.random_model_data(model, batch_size=N, seed=3, velocity_representation=rep)
    rng = np.random.default_rng(7)
    tau = 10 * rng.uniform(size=(N, model.dofs()))
    f = rng.uniform(size=(N, model.number_of_links(), 6))
    if not model.floating_base():
        f[:, 0] = 0  # a wrench on a fixed base is ignored by ABA (quirk 9)
    vd, sdd = oracle.forward_dynamics_aba(model, d, joint_forces=tau, link_forces=f)
    fB, tau_id = oracle.inverse_dynamics(model, d, joint_accelerations=sdd, base_acceleration=vd, link_forces=f)
    np.testing.assert_allclose(tau_id, tau, rtol=RTOL, atol=1e-8)
    if model.floating_base():
        np.testing.assert_allclose(fB, 0, atol=1e-8)


def _mass_matrix_from_rnea(model, d):
    """Columns of M in the data's representation: M e_i = ID(nu_dot = e_i) - ID(0)."""
    n = model.dofs()
    h = np.concatenate(oracle.inverse_dynamics(model, d), -1)
    cols = []
    for i in range(6 + n):
        e = np.zeros((d.batch_size, 6 + n))
        e[:, i] = 1
        cols.append(np.concatenate(oracle.inverse_dynamics(model, d, base_acceleration=e[:, :6], joint_accelerations=e[:, 6:]), -1) - h)
    return np.stack(cols, -1), h


# reference: tests/test_api_model.py:532-549 (ABA == CRB solve  M nu_dot = S tau - h + J^T f)
@pytest.mark.parametrize("name", ["chain9f", "icub"])
def test_aba_equals_dense_solve(models, name):
    model = models(name)
    N = 3
    d = oracle.random_model_data(model, batch_size=N, seed=5, velocity_representation=VelRepr.Inertial)
    rng = np.random.default_rng(11)
    tau = 10 * rng.uniform(size=(N, model.dofs()))
    M, h = _mass_matrix_from_rnea(model, d)
    rhs = np.concatenate([np.zeros((N, 6)), tau], -1) - h
    nu_dot = np.linalg.solve(M, rhs[..., None])[..., 0]
    vd, sdd = oracle.forward_dynamics_aba(model, d, joint_forces=tau)
    np.testing.assert_allclose(np.concatenate([vd, sdd], -1), nu_dot, rtol=1e-6, atol=1e-7)
    assert np.allclose(M, np.swapaxes(M, -1, -2), atol=1e-9)
    assert np.all(np.linalg.eigvalsh(M) > 0)


@pytest.mark.parametrize("name", ["chain9f", "anymal"])
def test_crba_matches_rnea_columns(models, name):
    model = models(name)
    d = oracle.random_model_data(model, batch_size=2, seed=9, velocity_representation=VelRepr.Body)
    M, _ = _mass_matrix_from_rnea(model, d)
    np.testing.assert_allclose(oracle.free_floating_mass_matrix(model, d), M, rtol=1e-7, atol=1e-8)


# reference idea: tests/test_api_contact.py:59-102 -- Jacobian-free variant
def test_contact_point_velocity_is_time_derivative_of_position(models):
    model = helpers.with_params(models("icub"), gravity=0.0)
    d = models.random_data("icub", 3, seed=2, in_contact=False)
    p0, v0 = oracle.collidable_points_pos_vel(model, link_transforms=d.link_transforms, link_velocities=d.link_velocities)
    # advance the configuration with the *current* velocities over a tiny dt (no dynamics)
    h = 1e-7
    W_v = d.generalized_velocity(VelRepr.Inertial)
    pdot = W_v[:, :3] + np.cross(W_v[:, 3:6], d.base_position)
    qdot = oracle.refmath.quaternion_derivative(d.base_orientation, W_v[:, 3:6])
    d2 = dataclasses.replace(
        d,
        base_position=d.base_position + h * pdot,
        base_quaternion=d.base_orientation + h * qdot,
        joint_positions=d.joint_positions + h * d.joint_velocities,
    ).update_caches(model)
    p1, _ = oracle.collidable_points_pos_vel(model, link_transforms=d2.link_transforms, link_velocities=d2.link_velocities)
    np.testing.assert_allclose((p1 - p0) / h, v0, rtol=1e-5, atol=1e-6)


# reference: tests/test_simulations.py:347-401 (joint-limit spring holds the pendulum near the limit)
def test_joint_limit_spring_holds_pendulum(models):
    model = models("pendulum")
    kdp = dataclasses.replace(
        model.kin_dyn_parameters,
        position_limits_min=np.array([-1.5708]), position_limits_max=np.array([1.5708]),
        position_limit_spring=np.array([75.0]), position_limit_damper=np.array([0.1]),
    )  # fmt: skip
    model = helpers.with_params(model, kin_dyn_parameters=kdp)
    theta = 10 * np.pi / 180
    d = oracle.OracleData.build(model, joint_positions=[1.5708 + theta])
    worst = 0.0
    for _ in range(3000):
        d = oracle.step(model, d)
        worst = max(worst, float(d.joint_positions[0, 0]))
    assert worst <= 1.5708 + theta * 1.1
    assert d.joint_positions[0, 0] < 1.5708 + theta * 1.1


# build's own sanity test: energy of the frictionless cartpole
def test_cartpole_energy_drift_is_bounded(models):
    model = models("cartpole")
    d = oracle.OracleData.build(model, joint_positions=[0.1, 0.7], joint_velocities=[0.3, -0.2])

    def energy(d):
        M = oracle.crba(model, joint_positions=d.joint_positions)[0, 6:, 6:]
        sd = d.joint_velocities[0]
        com = oracle.com_position(model, d)[0]
        return 0.5 * sd @ M @ sd - model.total_mass() * model.gravity * com[2]

    e0 = energy(d)
    for _ in range(1000):
        d = oracle.step(model, d)
    assert abs(energy(d) - e0) < 2e-3 * max(1.0, abs(e0))


def _strip_points(model):
    kdp = model.kin_dyn_parameters
    return dataclasses.replace(
        kdp, contact_body=np.zeros(0, dtype=np.int64), contact_point=np.zeros((0, 3)), contact_enabled=np.zeros(0, dtype=bool)
    )
