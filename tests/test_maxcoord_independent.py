"""[round 6] Forward dynamics against an independent statement of the physics: tests/maxcoord.py solves every URDF in
MAXIMAL coordinates (one free rigid body per massive link, every joint -- fixed ones too -- as acceleration constraints,
one dense KKT solve) from the URDF text alone.  The oracle, the kernel core (host emulation) and the HIP kernels
(``-m gpu``) must reproduce its joint accelerations and its base acceleration (mixed representation) for floating and fixed
bases, revolute and prismatic joints, joint efforts, external link wrenches, eight children on one link, and links that the
product's parser LUMPS (massive bodies on fixed joints: ``robots.lumped_tree_urdf``) -- the rule of SURVEY 8(a) row T that no
other test can see, because every other parity test hands the oracle the product's own tables.

Quirk 12 of the reference (the world -> base offset of a fixed-base model moves the cached link frames but not
the ABA) makes the mixed-representation arm of an external link wrench non-physical on such models: they run without link
wrenches here (``pendulum``, ``chain5``)."""
import ast
import pathlib

import numpy as np
import pytest

import helpers
import jaxsim_amd as ja
import maxcoord
import oracle
from jaxsim_amd import robots
from oracle import VelRepr

TEXTS = {
    "cartpole": lambda: robots.cartpole_urdf(),
    "pendulum": lambda: robots.single_pendulum_urdf(),
    "double_pendulum": lambda: robots.double_pendulum_urdf(),
    "chain5": lambda: robots.chain_urdf(5, fixed_base=True, seed=1),
    "chain9f": lambda: robots.chain_urdf(9, fixed_base=False, seed=2),
    "box": lambda: robots.box_urdf(),
    "anymal": lambda: robots.anymal12_urdf(),
    "icub": lambda: robots.icub23_urdf(),
    "octopod": lambda: robots.hub_urdf(8, 2, foot_boxes=4, seed=1),
    "lumped7f": lambda: robots.lumped_tree_urdf(7, seed=0),
    "lumped12f": lambda: robots.lumped_tree_urdf(12, seed=2),
    "lumped5": lambda: robots.lumped_tree_urdf(5, seed=1, fixed_base=True),
}
OFFSET_BASE = ("pendulum", "chain5", "double_pendulum")  # fixed bases placed off the world origin: no link wrenches (quirk 12)
_MODELS = {}


def case(name, N, seed, dtype=np.float64):
    if name not in _MODELS:
        text = TEXTS[name]()
        _MODELS[name] = (text, ja.JaxSimModel.build_from_model_description(text))
    text, model = _MODELS[name]
    d = oracle.random_model_data(model, batch_size=N, seed=seed, dtype=dtype, velocity_representation=VelRepr.Mixed)
    if not model.floating_base():  # ABA keeps the base of a fixed-base model at rest (rbda/aba.py:109-121)
        d.base_linear_velocity[:] = 0
        d.base_angular_velocity[:] = 0
    rng = np.random.default_rng(seed + 1)
    tau = rng.uniform(-3, 3, size=(N, model.dofs())).astype(dtype)
    f = rng.uniform(-5, 5, size=(N, model.number_of_links(), 6)).astype(dtype)
    if name in OFFSET_BASE:
        f[:] = 0
    return text, model, d, tau, f


def truth(text, model, d, tau, f):
    """(base acceleration in MIXED representation [N, 6], joint accelerations [N, n] in the model's joint order) from
    tests/maxcoord.py; the inputs are read in float64 whatever the dtype of the state."""
    jn, lnm = model.joint_names(), model.link_names()
    N = d.base_position.shape[0]
    vd, sdd = np.zeros((N, 6)), np.zeros((N, len(jn)))
    bv = d.base_velocity(VelRepr.Mixed).astype(np.float64)  # (the data object STORES the inertial-fixed velocity, api/data.py:151-156)
    for e in range(N):
        ba, acc = maxcoord.forward_dynamics(
            text, base_position=d.base_position[e].astype(np.float64), base_quaternion=d.base_quaternion[e].astype(np.float64),
            base_linear_velocity=bv[e, :3], base_angular_velocity=bv[e, 3:],
            joint_positions=dict(zip(jn, d.joint_positions[e].astype(np.float64))), joint_velocities=dict(zip(jn, d.joint_velocities[e].astype(np.float64))),
            joint_forces=dict(zip(jn, tau[e].astype(np.float64))),
            link_wrenches={n: (f[e, i, :3].astype(np.float64), f[e, i, 3:].astype(np.float64)) for i, n in enumerate(lnm)}, gravity=model.gravity)  # fmt: skip
        vd[e] = ba
        sdd[e] = [acc[n] for n in jn]
    return vd, sdd


def mixed_state_is_stored(d):
    # (OracleData keeps the base velocity as given in `velocity_representation`; the maximal-coordinate solver wants MIXED)
    assert d.velocity_representation == VelRepr.Mixed


def rel(a, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max())) if ref.size else 0.0


def test_the_solver_is_independent_of_product_and_oracle():
    """tests/maxcoord.py imports NumPy and xml.etree -- nothing of the product's parser or tables, nothing of oracle/."""
    tree = ast.parse((pathlib.Path(__file__).parent / "maxcoord.py").read_text())
    mods = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            mods |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            mods.add((node.module or "").split(".")[0])
    assert mods <= {"numpy", "xml", "__future__"}, mods


def test_free_rigid_body_known_answer():
    """The solver against a closed form it does not contain: a torque-free rigid body whose centre of mass is OFF the link
    origin -- alpha = -I^-1 (omega x I omega), and the origin accelerates by g + alpha x r + omega x (omega x r), r = origin - com."""
    I = np.diag([0.02, 0.05, 0.03])
    c = np.array([0.05, -0.02, 0.04])
    text = ('<robot name="b"><link name="b"><inertial><origin xyz="0.05 -0.02 0.04" rpy="0 0 0"/><mass value="2.0"/>'
            '<inertia ixx="0.02" ixy="0" ixz="0" iyy="0.05" iyz="0" izz="0.03"/></inertial></link></robot>')
    q = np.array([0.9, 0.1, -0.3, 0.2])
    R = maxcoord.quat_matrix(q)
    w = np.array([0.7, -1.1, 0.4])
    ba, _ = maxcoord.forward_dynamics(text, base_position=[0.3, 0.2, 1.0], base_quaternion=q, base_linear_velocity=[0.1, 0.2, 0.3],
                                      base_angular_velocity=w, joint_positions={}, joint_velocities={})
    Iw = R @ I @ R.T
    alpha = -np.linalg.solve(Iw, np.cross(w, Iw @ w))
    r = -(R @ c)
    np.testing.assert_allclose(ba[3:], alpha, atol=1e-13)
    np.testing.assert_allclose(ba[:3], np.array([0, 0, -9.81]) + np.cross(alpha, r) + np.cross(w, np.cross(w, r)), atol=1e-13)


@pytest.mark.parametrize("name", list(TEXTS))
def test_oracle_forward_dynamics_equals_maximal_coordinates(name):
    text, model, d, tau, f = case(name, 4, seed=3)
    mixed_state_is_stored(d)
    vd, sdd = truth(text, model, d, tau, f)
    ovd, osdd = oracle.forward_dynamics_aba(model, d, joint_forces=tau, link_forces=f)
    assert rel(osdd, sdd) < 1e-10
    if model.floating_base():
        assert rel(ovd, vd) < 1e-10


def test_lumping_is_what_the_comparison_sees():
    """The lumped trees really exercise the rule: the product's table has 7 links where the URDF has 7 + payloads, and the
    lumped mass is the sum."""
    text, model = TEXTS["lumped7f"](), None
    U = maxcoord.Urdf(text)
    model = ja.JaxSimModel.build_from_model_description(text)
    n_bodies = sum(1 for rec in U.links.values() if rec["mass"] > 0)
    assert model.number_of_links() == 7 and n_bodies >= 14
    np.testing.assert_allclose(np.sum(model.kin_dyn_parameters.link_mass), sum(rec["mass"] for rec in U.links.values()), rtol=1e-12)


@pytest.mark.parametrize("name", list(TEXTS))
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-4)])
def test_kernel_core_equals_maximal_coordinates(name, dtype, tol):
    """The kernel core (host emulation of the HIP kernels' source) through its raw interface: inertial-fixed base
    acceleration out, converted to mixed by hand here -- pddot_B = vdot_O + wdot x p_B + w x pdot_B."""
    import emul_binding as eb

    text, model, d, tau, f = case(name, 4, seed=3, dtype=dtype)
    vd, sdd = truth(text, model, d, tau, f)
    out = eb.run(model, eb.MODE_FD, helpers.odata_to_block(model, d), tau=tau.T, link_forces=f.reshape(4, -1).T, force_repr=2).T.astype(np.float64)
    assert rel(out[:, 6:], sdd) < tol
    if model.floating_base():
        bv = d.base_velocity(VelRepr.Mixed).astype(np.float64)
        p, w, pd = d.base_position.astype(np.float64), bv[:, 3:], bv[:, :3]
        lin = out[:, :3] + np.cross(out[:, 3:6], p) + np.cross(w, pd)
        assert rel(np.concatenate([lin, out[:, 3:6]], -1), vd) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TEXTS))
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-4)])
def test_hip_kernels_equal_maximal_coordinates_gpu(name, dtype, tol):
    """``js.model.forward_dynamics_aba`` on the device (C-ABI ``jxs_forward_dynamics_aba``), mixed representation in and out."""
    import jaxsim_amd.api as js

    N = 37
    text, model, d, tau, f = case(name, N, seed=4, dtype=dtype)
    vd, sdd = truth(text, model, d, tau, f)
    g = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), ja.VelRepr.Mixed)
    gvd, gsdd = js.model.forward_dynamics_aba(model, g, joint_forces=tau, link_forces=f)
    assert rel(gsdd, sdd) < tol
    if model.floating_base():
        assert rel(gvd, vd) < tol


# ---- [round 6] rows E / I (js.ode.system_acceleration, js.contact.link_contact_forces) against the same solver ------------
CONTACT_CASES = [("box", "soft", None), ("anymal", "soft", None), ("icub", "soft", None), ("chain9f", "soft", None), ("octopod", "soft", None),
                 ("anymal", "rigid", helpers.ANYMAL_FEET_4), ("box", "rigid", [0, 1, 2, 3]), ("anymal", "relaxed", helpers.ANYMAL_FEET_16),
                 ("icub16", "relaxed", list(range(16))), ("octopod", "relaxed", [8 * f_ + c_ for f_ in range(4) for c_ in range(4)])]


def _contact_case(models, name, kind, idx, N):
    text = models._urdf[name]()
    model = models(name)
    if kind == "rigid":
        model = helpers.rigid_model(model, idx, K=1e4, D=1e2, build=dict(solver_options={"solver_tol": 1e-9}))
    elif kind == "relaxed":
        model = helpers.relaxed_model(model, idx, mu=0.5)
    d = models.random_data(name, N, seed=9, rep=VelRepr.Inertial)
    rng = np.random.default_rng(10)
    return text, model, d, rng.uniform(-3, 3, size=(N, model.dofs())), rng.uniform(-5, 5, size=(N, model.number_of_links(), 6))


def _worst_against_maxcoord(text, model, d, tau, f, W, vd_inertial, sdd_dev):
    """Worst relative distance of the device's / emulation's (inertial base acceleration, joint accelerations) to the
    maximal-coordinate solver under the external wrenches `f` plus the reported contact wrenches `W` (both inertial)."""
    jn, lnm = model.joint_names(), model.link_names()
    bvm = d.base_velocity(VelRepr.Mixed)
    worst = 0.0
    for e in range(d.base_position.shape[0]):
        ba, acc = maxcoord.forward_dynamics(
            text, base_position=d.base_position[e], base_quaternion=d.base_quaternion[e], base_linear_velocity=bvm[e, :3], base_angular_velocity=bvm[e, 3:],
            joint_positions=dict(zip(jn, d.joint_positions[e])), joint_velocities=dict(zip(jn, d.joint_velocities[e])), joint_forces=dict(zip(jn, tau[e])),
            world_wrenches={n: (f[e, i, :3] + W[e, i, :3], f[e, i, 3:] + W[e, i, 3:]) for i, n in enumerate(lnm)}, gravity=model.gravity)  # fmt: skip
        sdd = np.array([acc[n] for n in jn])
        p, w, pd = d.base_position[e], bvm[e, 3:], bvm[e, :3]
        ref_vd = np.concatenate([ba[:3] - np.cross(ba[3:], p) - np.cross(w, pd), ba[3:]])  # vdot_O = pddot_B - wdot x p_B - w x pdot_B
        scale = max(1.0, float(np.abs(sdd).max()) if sdd.size else 0.0, float(np.abs(ref_vd).max()))
        if sdd.size:
            worst = max(worst, float(np.abs(np.asarray(sdd_dev)[e] - sdd).max()) / scale)
        worst = max(worst, float(np.abs(np.asarray(vd_inertial)[e] - ref_vd).max()) / scale)
    return worst


@pytest.mark.parametrize("name,kind,idx", CONTACT_CASES)
def test_kernel_core_contact_accelerations_follow_from_the_reported_wrenches(models, name, kind, idx):
    """The CPU twin of the GPU test below: MODE_DYN / MODE_DYN_RIGID of the kernel sources in the host emulation -- the link
    wrenches the kernel writes, fed to the maximal-coordinate solver, give the accelerations the same launch writes."""
    import emul_binding as eb
    from jaxsim_amd import state as st

    N = 6
    text, model, d, tau, f = _contact_case(models, name, kind, idx, N)
    xdot, W = eb.run(model, eb.MODE_DYN, helpers.odata_to_block(model, d), tau=tau.T, link_forces=f.reshape(N, -1).T, force_repr=0)
    fields = st.unpack_state(st.StateLayout.of(model), np.asarray(xdot, dtype=np.float64))
    W = np.asarray(W, dtype=np.float64).T.reshape(N, model.number_of_links(), 6)
    assert np.abs(W).max() > 1.0
    vd = np.concatenate([fields["base_linear_velocity"], fields["base_angular_velocity"]], -1)
    assert _worst_against_maxcoord(text, model, d, tau, f, W, vd, fields["joint_velocities"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,idx", CONTACT_CASES)
def test_contact_accelerations_follow_from_the_reported_wrenches_gpu(models, name, kind, idx):
    """The device's ``link_contact_forces`` (inertial wrenches per link) fed to the maximal-coordinate solver together with
    the same joint torques and external link wrenches must give the device's ``system_acceleration``: the forces the contact
    model reports are the forces the dynamics used, and the dynamics under them are Newton-Euler -- for SoftContacts, RigidContacts
    and RelaxedRigidContacts, without the oracle.  Inertial representation throughout (with Body / Mixed data the reference
    adds inertial contact wrenches to link forces of the data's representation, api/ode.py:77-118: reproduced, not physics)."""
    import jaxsim_amd.api as js

    N = 12
    text, model, d, tau, f = _contact_case(models, name, kind, idx, N)
    g = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), ja.VelRepr.Inertial)
    W, _aux = js.contact.link_contact_forces(model, g, link_forces=f, joint_torques=tau)
    W = np.asarray(W, dtype=np.float64)
    assert np.abs(W).max() > 1.0  # contacts act in this sample
    gvd, gsdd, _cs = js.ode.system_acceleration(model, g, link_forces=f, joint_torques=tau)
    assert _worst_against_maxcoord(text, model, d, tau, f, W, gvd, gsdd) < 1e-9
