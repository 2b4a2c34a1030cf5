"""Host-side logic: URDF reader rules, static tables, state layout, boundary conversions."""

import numpy as np
import pytest

import helpers
import jaxsim_amd as ja
import oracle
from jaxsim_amd import _hostmath as hm
from jaxsim_amd import data as jdata
from jaxsim_amd import robots
from jaxsim_amd import state as st
from jaxsim_amd.parsers import urdf


def test_bfs_indexing_children_sorted_by_name(models):
    kdp = models("icub").kin_dyn_parameters
    # reference rule: BFS from the base, children sorted by name (kinematic_graph.py:133-134,701)
    assert kdp.link_names[:4] == ("root_link", "l_hip_1", "r_hip_1", "torso_1")
    assert kdp.parent_array[0] == -1 and all(kdp.parent_array[i] < i for i in range(1, 24))
    assert kdp.joint_names[0] == "l_hip_pitch"  # joint index = child link index
    assert kdp.number_of_links() == 24 and kdp.number_of_joints() == 23
    assert kdp.number_of_collidable_points() == 32
    assert kdp.tree_depths().max() == 7
    assert models("icub16").kin_dyn_parameters.number_of_collidable_points() == 16


def test_cartpole_tables(models):
    m = models("cartpole")
    kdp = m.kin_dyn_parameters
    assert not m.floating_base()
    assert kdp.link_names == ("rail", "cart", "pole")
    assert kdp.frame_names == ("rail_frame", "cart_frame") or set(kdp.frame_names) == {"rail_frame", "cart_frame"}
    assert list(kdp.joint_types) == [2, 1]  # prismatic-y then continuous-x
    np.testing.assert_allclose(kdp.motion_subspaces[1], [0, 1, 0, 0, 0, 0])
    np.testing.assert_allclose(kdp.motion_subspaces[2], [0, 0, 0, 1, 0, 0])
    np.testing.assert_allclose(kdp.lambda_H_pre[1][:3, 3], [0, 0, 1.2])
    assert kdp.position_limits_min[0] == -2.4 and kdp.position_limits_max[1] == np.finfo(float).max
    np.testing.assert_allclose(m.total_mass(), 6.5)
    assert kdp.number_of_collidable_points() == 0  # cylinders are skipped, config C2 has no boxes


def test_four_bar_opened_tables():
    """[round 5] Row T on the reference's third shipped asset (``tests/assets/4_bar_opened.urdf``, re-typed as
    ``robots.four_bar_opened_urdf``), expectations stated by hand from the rules of SURVEY 8(a) row T: root = the link
    that is nobody's child, BFS indices with children sorted by NAME, joint index = child link index, massless
    fixed-joint children become frames of their parent, a box is eight corner points (bottom first)."""
    from jaxsim_amd import robots

    m = ja.JaxSimModel.build_from_model_description(robots.four_bar_opened_urdf())
    kdp = m.kin_dyn_parameters
    assert m.floating_base()
    assert kdp.link_names == ("AB", "BC1", "DA", "CD", "BC2")  # AB; its children BC1 < DA; then DA's CD; then CD's BC2
    assert list(kdp.parent_array) == [-1, 0, 0, 2, 3]
    assert kdp.joint_names == ("B", "A", "D", "C")  # joint i drives link i + 1
    assert set(kdp.frame_names) == {"BC1_frame", "BC2_frame"}
    frames = dict(zip(kdp.frame_names, kdp.frame_body))
    assert frames["BC1_frame"] == 1 and frames["BC2_frame"] == 4
    np.testing.assert_allclose(kdp.frame_transform[kdp.frame_names.index("BC1_frame")][:3, 3], [0, 0.25, 0])
    Rz = kdp.frame_transform[kdp.frame_names.index("BC2_frame")][:3, :3]
    np.testing.assert_allclose(Rz, [[np.cos(3.1416), -np.sin(3.1416), 0], [np.sin(3.1416), np.cos(3.1416), 0], [0, 0, 1]], atol=1e-12)
    assert list(kdp.joint_types) == [1, 1, 1, 1]
    for i in range(1, 5):
        np.testing.assert_allclose(kdp.motion_subspaces[i], [0, 0, 0, 0, 0, 1])
    np.testing.assert_allclose(kdp.lambda_H_pre[1][:3, 3], [0, -0.25, 0])  # joint B
    np.testing.assert_allclose(kdp.lambda_H_pre[2][:3, 3], [0, 0.25, 0])   # joint A
    np.testing.assert_allclose(kdp.lambda_H_pre[3][:3, 3], [0, 0.5, 0])    # joint D
    np.testing.assert_allclose(kdp.lambda_H_pre[3][:3, :3], [[np.cos(1.57), -np.sin(1.57), 0], [np.sin(1.57), np.cos(1.57), 0], [0, 0, 1]], atol=1e-12)
    np.testing.assert_allclose(kdp.link_mass, [1.0, 0.5, 1.0, 1.0, 0.5])
    np.testing.assert_allclose(m.total_mass(), 4.0)
    np.testing.assert_allclose(kdp.position_limits_min, -1.57)
    np.testing.assert_allclose(kdp.position_limits_max, 1.57)
    assert kdp.number_of_collidable_points() == 8 and set(np.asarray(kdp.contact_body)) == {3}  # the box on CD
    pts = np.asarray(kdp.contact_point)
    np.testing.assert_allclose(pts[:4, 2], -0.05)
    np.testing.assert_allclose(sorted(set(np.round(pts[:, 1], 6))), [0.0, 0.5])  # the box is centred 0.25 m along CD
    assert kdp.tree_depths().max() == 3


def test_fixed_joint_lumping_conserves_mass_and_inertia():
    # child rigidly attached 0.5 m above the parent: lumped inertia = parallel-axis sum
    u = (
        '<robot name="l"><link name="a"><inertial><origin xyz="0 0 0"/><mass value="2"/>'
        '<inertia ixx="0.1" iyy="0.2" izz="0.3" ixy="0" ixz="0" iyz="0"/></inertial></link>'
        '<link name="b"><inertial><origin xyz="0 0 0"/><mass value="3"/>'
        '<inertia ixx="0.01" iyy="0.01" izz="0.01" ixy="0" ixz="0" iyz="0"/></inertial>'
        '<collision><origin xyz="0 0 0"/><geometry><box size="0.2 0.2 0.2"/></geometry></collision></link>'
        '<joint name="j" type="fixed"><origin xyz="0 0 0.5" rpy="0 0 1.5707963267948966"/><parent link="a"/><child link="b"/></joint></robot>'
    )
    d = urdf.parse_urdf(u)
    assert [l.name for l in d.links] == ["a"] and [f.name for f in d.frames] == ["b"]
    m, c, I = hm.inertia_to_params(d.links[0].inertia)
    assert m == pytest.approx(5.0)
    np.testing.assert_allclose(c, [0, 0, 0.3], atol=1e-12)
    # inertia about the combined CoM
    Ixx = 0.1 + 2 * 0.3**2 + 0.01 + 3 * 0.2**2
    np.testing.assert_allclose(np.diag(I), [Ixx, 0.2 + 2 * 0.09 + 0.01 + 3 * 0.04, 0.31], atol=1e-12)
    # the collision box moved with the lumped link: z in {0.4, 0.6}
    z = sorted({round(float(p.position[2]), 6) for p in d.collidable_points})
    assert z == [0.4, 0.6] and len(d.collidable_points) == 8


def test_box_points_order_and_sphere_count(models):
    pts = models("box").kin_dyn_parameters.contact_point
    assert pts.shape == (8, 3)
    np.testing.assert_allclose(pts[:4, 2], -0.05)  # bottom corners first (rod/utils.py:116-151)
    np.testing.assert_allclose(pts[4:, 2], 0.05)
    np.testing.assert_allclose(pts[0], [-0.15, -0.1, -0.05])
    sp = models("sphere").kin_dyn_parameters.contact_point
    assert sp.shape == (50, 3)
    np.testing.assert_allclose(np.linalg.norm(sp, axis=1), 0.1)


def test_link_inertia_round_trip(models):
    kdp = models("chain9f").kin_dyn_parameters
    for i in range(kdp.number_of_links()):
        M = hm.inertia_to_sixd(kdp.link_mass[i], kdp.link_com[i], kdp.link_inertia_com[i])
        np.testing.assert_allclose(M, kdp.link_spatial_inertia[i], atol=1e-12)
        assert np.all(np.linalg.eigvalsh(M) > 0)


def test_fixed_base_world_joint_offset_goes_to_base_pose(models):
    kdp = models("pendulum").kin_dyn_parameters
    np.testing.assert_allclose(kdp.suc_H_i[0][:3, 3], [0, 0, 1.0])
    assert not models("pendulum").floating_base()


def test_gravity_sign_and_defaults(models):
    m = models("icub")
    assert m.gravity == pytest.approx(-9.81) and m.time_step == pytest.approx(1e-3)
    assert (m.contact_params.K, m.contact_params.D, m.contact_params.mu) == (1e6, 2000.0, 0.5)
    assert (m.actuation_params.torque_max, m.actuation_params.omega_th, m.actuation_params.omega_max) == (3000.0, 30.0, 100.0)
    with m.editable(validate=False) as m2:
        m2.time_step = 5e-4
    assert m.time_step == pytest.approx(1e-3) and m2.time_step == pytest.approx(5e-4)


def test_state_layout_and_pack_round_trip(models):
    m = models("icub")
    L = st.StateLayout.of(m)
    assert L.n_rows == 155  # 13 + 2*23 + 3*32
    assert (L.row_s, L.row_vlin, L.row_sd, L.row_m) == (7, 30, 36, 59)
    d = models.random_data("icub", 5, seed=1)
    blk = helpers.odata_to_block(m, d)
    assert blk.shape == (155, 5) and blk.flags.c_contiguous
    f = st.unpack_state(L, blk)
    np.testing.assert_array_equal(f["joint_velocities"], d.joint_velocities)
    np.testing.assert_array_equal(f["tangential_deformation"], d.tangential_deformation)
    np.testing.assert_array_equal(f["base_quaternion"], d.base_quaternion)


@pytest.mark.parametrize("rep_o,rep_p", [(oracle.VelRepr.Body, ja.VelRepr.Body), (oracle.VelRepr.Mixed, ja.VelRepr.Mixed),
                                         (oracle.VelRepr.Inertial, ja.VelRepr.Inertial)])  # fmt: skip
@pytest.mark.parametrize("is_force", [False, True])
def test_boundary_conversions_match_reference_formulas(rep_o, rep_p, is_force):
    rng = np.random.default_rng(0)
    N = 7
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    H = np.zeros((N, 4, 4))
    H[:, :3, :3] = hm.quaternion_to_rotation(q)
    H[:, :3, 3] = rng.normal(size=(N, 3))
    H[:, 3, 3] = 1
    x = rng.normal(size=(N, 6))
    np.testing.assert_allclose(jdata._other_to_inertial(x, rep_p, H, is_force),
                               oracle.other_representation_to_inertial(x, rep_o, H, is_force=is_force), atol=1e-12)  # fmt: skip
    np.testing.assert_allclose(jdata._inertial_to_other(x, rep_p, H, is_force),
                               oracle.inertial_to_other_representation(x, rep_o, H, is_force=is_force), atol=1e-12)  # fmt: skip
    back = jdata._inertial_to_other(jdata._other_to_inertial(x, rep_p, H, is_force), rep_p, H, is_force)
    np.testing.assert_allclose(back, x, atol=1e-12)


def test_estimate_good_contact_parameters_matches_oracle(models):
    import jaxsim_amd.api as js

    for name, nc in (("box", 4), ("icub", 16), ("anymal", 4)):
        m = models(name)
        p = js.contact.estimate_good_contact_parameters(m, number_of_active_collidable_points_steady_state=nc)
        o = oracle.estimate_good_contact_parameters(m, number_of_active_collidable_points_steady_state=nc)
        assert (p.K, p.D, p.mu) == pytest.approx((o["K"], o["D"], o["mu"]))


def test_quaternion_helpers():
    rpy = np.array([[0.3, -0.2, 1.1]])
    q = hm.rpy_to_quaternion(rpy)
    Rx = hm.rpy_to_rotation([0.3, 0, 0])
    Ry = hm.rpy_to_rotation([0, -0.2, 0])
    Rz = hm.rpy_to_rotation([0, 0, 1.1])
    np.testing.assert_allclose(hm.quaternion_to_rotation(q)[0], Rx @ Ry @ Rz, atol=1e-12)  # intrinsic XYZ
    np.testing.assert_allclose(oracle.refmath.so3_from_quaternion(2.5 * q)[0], Rx @ Ry @ Rz, atol=1e-12)


def test_product_never_imports_the_oracle():
    import pathlib
    import re

    root = pathlib.Path(ja.__file__).resolve().parent
    for py in root.rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), py


@pytest.mark.parametrize("tile", [1, 2, 4, 16])
@pytest.mark.parametrize("N", [1, 5, 16, 33])
def test_tile_interleaved_layout_round_trip(tile, N):
    rows = 7
    blk = np.arange(rows * N, dtype=np.float32).reshape(rows, N)
    flat = st.tile_block(blk, tile)
    nt = -(-N // tile)
    assert flat.shape == (nt * rows * tile,)
    # element (row, env) lives at ((env // T) * rows + row) * T + env % T   (include/jaxsim_amd.h)
    for row, env in ((0, 0), (3, N - 1), (rows - 1, N // 2)):
        assert flat[((env // tile) * rows + row) * tile + env % tile] == blk[row, env]
    np.testing.assert_array_equal(st.untile_block(flat, rows, N, tile), blk)


def test_model_reduction_locks_joints_and_lumps_links(models):
    """``js.model.reduce`` / ``considered_joints`` (reference api/model.py:807-878): the reduced model is
    the full one with the removed joints held at their locked positions -- same total mass, and its
    mass matrix is the full one without the locked rows / columns."""
    import jaxsim_amd.api as js

    full = models("anymal")
    names = list(full.joint_names())
    locked = {"LF_KFE": -0.7, "RH_HAA": 0.3, "RH_HFE": 0.0}
    keep = [n for n in names if n not in locked]
    red = js.model.reduce(full, considered_joints=keep, locked_joint_positions={k: v for k, v in locked.items() if v != 0.0})
    assert red.number_of_joints() == len(keep) and set(red.joint_names()) == set(keep)
    assert red.number_of_links() == full.number_of_links() - len(locked)
    assert red.total_mass() == pytest.approx(full.total_mass(), rel=1e-12)
    assert red.time_step == full.time_step and red.gravity == full.gravity
    rng = np.random.default_rng(0)
    s_red = rng.uniform(-0.8, 0.8, size=(3, len(keep)))
    s_full = np.zeros((3, len(names)))
    for k, n in enumerate(names):
        s_full[:, k] = locked[n] if n in locked else s_red[:, list(red.joint_names()).index(n)]
    M_full = oracle.crba(full, joint_positions=s_full)
    M_red = oracle.crba(red, joint_positions=s_red)
    idx = [0, 1, 2, 3, 4, 5] + [6 + names.index(n) for n in red.joint_names()]
    np.testing.assert_allclose(M_red, M_full[:, idx][:, :, idx], rtol=1e-10, atol=1e-10)
    # the collidable points follow their lumped links
    assert red.kin_dyn_parameters.number_of_collidable_points() == full.kin_dyn_parameters.number_of_collidable_points()
    # all joints considered: identical tables
    same = js.model.reduce(full, considered_joints=names)
    np.testing.assert_array_equal(same.kin_dyn_parameters.parent_array, full.kin_dyn_parameters.parent_array)
    with pytest.raises(ValueError, match="not existing"):
        js.model.reduce(full, considered_joints=names + ["nope"])


def test_relaxed_rigid_contacts_host_objects(models):
    """``RelaxedRigidContacts.build`` / ``RelaxedRigidContactsParams`` mirror the reference
    (relaxed_rigid.py:29-75,203-251): default L-BFGS option keys are kept, user options are merged,
    unhashable option values are rejected; parameter validity follows ``valid()`` (:184-200)."""
    import jaxsim_amd as ja
    import jaxsim_amd.api as js

    cm = ja.RelaxedRigidContacts.build()
    assert cm.solver_options == {"tol": 1e-6, "maxiter": 50, "memory_size": 10, "scale_init_precond": False}
    assert ja.RelaxedRigidContacts.build(solver_options={"tol": 1e-3}).solver_options["tol"] == 1e-3
    with pytest.raises(ValueError, match="hashable"):
        ja.RelaxedRigidContacts.build(solver_options={"tol": [1e-3]})
    p = cm._parameters_class()
    assert (p.time_constant, p.damping_coefficient, p.d_min, p.d_max, p.width, p.midpoint, p.power, p.mu) == (
        0.02, 1.0, 0.9, 0.95, 0.001, 0.5, 2.0, 0.005)  # fmt: skip
    assert p.valid() and not ja.RelaxedRigidContactsParams.build(d_min=0.99, d_max=0.95).valid()
    assert not ja.RelaxedRigidContactsParams.build(damping_coefficient=0.0).valid()
    # estimate_good_contact_parameters builds the parameter class of the active contact model
    # (api/contact.py:203-211): stiffness / damping / friction go to K, D, mu
    base = models("box")
    soft = js.contact.estimate_good_contact_parameters(base)
    for cm_, cls in ((ja.RelaxedRigidContacts.build(), ja.RelaxedRigidContactsParams), (ja.RigidContacts.build(), ja.RigidContactsParams)):
        with base.editable(validate=False) as m:
            m.contact_model = cm_
        got = js.contact.estimate_good_contact_parameters(m)
        assert type(got) is cls and got.mu == 0.5 and got.K == soft.K and got.D == soft.D


class _FakeData:
    """Stands in for JaxSimModelData where only the cached link transforms matter (no GPU)."""

    def __init__(self, W_H_L, rep, batched=True):
        self._link_transforms, self.velocity_representation, self._batched = W_H_L, rep, batched
        self.batch_size = W_H_L.shape[0] if batched else 1

    def valid(self, model=None):
        return True


def _random_transforms(rng, N, nL):
    from oracle import refmath

    H = np.zeros((N, nL, 4, 4))
    q = rng.normal(size=(N * nL, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    H[:, :, :3, :3] = refmath.so3_from_quaternion(q).reshape(N, nL, 3, 3)
    H[:, :, :3, 3] = rng.normal(size=(N, nL, 3))
    H[:, :, 3, 3] = 1
    return H


@pytest.mark.parametrize("rep", ["inertial", "mixed", "body"])
def test_references_store_inertial_and_convert_back(models, rep):
    """``JaxSimModelReferences`` (references.py:23-449): link forces are stored inertial-fixed and read
    back in the active representation; conversion = ``other_representation_to_inertial`` with the link
    transforms (api/common.py:160-222), checked against the oracle's restatement."""
    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from oracle import refstep

    model = models("anymal")
    nL, n, N = model.number_of_links(), model.dofs(), 5
    rng = np.random.default_rng(3)
    W_H_L = _random_transforms(rng, N, nL)
    vr = {"inertial": ja.VelRepr.Inertial, "mixed": ja.VelRepr.Mixed, "body": ja.VelRepr.Body}[rep]
    data = _FakeData(W_H_L, vr)
    f = rng.normal(size=(N, nL, 6))
    tau = rng.normal(size=(N, n))
    refs = js.references.JaxSimModelReferences.build(model, joint_force_references=tau, link_forces=f, data=data)
    assert refs.velocity_representation == vr and refs.valid(model)
    expect = refstep.other_representation_to_inertial(f, rep, W_H_L, is_force=True)
    np.testing.assert_allclose(refs._link_forces, expect, atol=1e-12)
    np.testing.assert_allclose(refs.link_forces(model, data), f, atol=1e-12)
    np.testing.assert_array_equal(refs.joint_force_references(model), tau)
    # by name, additive, and a different read-out representation
    names = (model.link_names()[3], model.link_names()[0])
    extra = rng.normal(size=(N, 2, 6))
    r2 = refs.apply_link_forces(extra, model=model, data=data, link_names=names, additive=True)
    got = r2.link_forces(model, data, link_names=names)
    np.testing.assert_allclose(got, f[:, [3, 0]] + extra, atol=1e-12)
    r3 = refs.apply_link_forces(extra, model=model, data=data, link_names=names, additive=False)
    np.testing.assert_allclose(r3.link_forces(model, data, link_names=names), extra, atol=1e-12)
    untouched = [i for i in range(nL) if i not in (0, 3)]
    np.testing.assert_allclose(r3.link_forces(model, data)[:, untouched], f[:, untouched], atol=1e-12)
    as_inertial = r3.switch_velocity_representation(ja.VelRepr.Inertial).link_forces(model)
    np.testing.assert_allclose(as_inertial, r3._link_forces, atol=0)
    jn = (model.joint_names()[4], model.joint_names()[1])
    r4 = refs.set_joint_force_references(np.ones((N, 2)), model=model, joint_names=jn)
    np.testing.assert_array_equal(r4.joint_force_references(model, joint_names=jn), np.ones((N, 2)))


def test_references_error_behaviour(models):
    """Same errors as the reference (references.py:205-215, 371-395)."""
    import jaxsim_amd as ja
    import jaxsim_amd.api as js

    model = models("box")
    R = js.references.JaxSimModelReferences
    z = R.zero(model)
    assert z._link_forces.shape == (1, 6) and z._joint_force_references.shape == (0,)
    with pytest.raises(ValueError, match="without a model"):
        z.link_forces(link_names=("box",))
    mixed = z.switch_velocity_representation(ja.VelRepr.Mixed)
    with pytest.raises(ValueError, match="Missing model "):
        mixed.link_forces()
    with pytest.raises(ValueError, match="Missing model data"):
        mixed.link_forces(model)
    with pytest.raises(ValueError, match="must match"):
        z.apply_link_forces(np.zeros((2, 6)), model=model, link_names=model.link_names())
    with pytest.raises(ValueError, match="unknown link"):
        z.apply_link_forces(np.zeros((1, 6)), model=model, link_names=("nope",))
    with pytest.raises(ValueError, match="expected joint forces"):
        R.build(model, link_forces=np.zeros((3, 6)))
    # no-model path: inertial forces for all links, additive or not
    a = z.apply_link_forces(np.ones((1, 6)))
    np.testing.assert_array_equal(a.apply_link_forces(np.ones((1, 6)), additive=True)._link_forces, 2 * np.ones((1, 6)))


def _tables(model):
    k = model.kin_dyn_parameters
    return dict(parent=k.parent_array, jt=k.joint_types, ax=k.joint_axis, lam=k.lambda_H_pre, suc=k.suc_H_i, m=k.link_mass,
                com=k.link_com, I=k.link_inertia_com, M6=k.link_spatial_inertia, kv=k.friction_viscous, smin=k.position_limits_min, smax=k.position_limits_max,
                body=k.contact_body, pt=k.contact_point)  # fmt: skip


def test_sdf_front_end_matches_the_equivalent_urdf():
    """SDF input (the reference reads its ``tests/assets/double_pendulum.sdf`` through `rod`,
    ``parsers/rod/parser.py:26-120``): pose graph with ``relative_to``, joint poses defaulting to the child
    link, world joint -> fixed base, ``<axis>`` children, explicit frames; converted to the URDF frame
    convention.  The same model written as URDF gives the same tables."""
    import jaxsim_amd as ja
    from jaxsim_amd import robots

    for coll in (False, True):
        a = ja.JaxSimModel.build_from_model_description(robots.double_pendulum_sdf(with_base_collision=coll))
        b = ja.JaxSimModel.build_from_model_description(robots.double_pendulum_urdf(with_base_collision=coll))
        assert a.link_names() == b.link_names() and a.joint_names() == b.joint_names()
        assert not a.floating_base() and a.number_of_links() == 3 and a.dofs() == 2
        ta, tb = _tables(a), _tables(b)
        for k in ta:
            np.testing.assert_allclose(np.asarray(ta[k], dtype=float), np.asarray(tb[k], dtype=float), atol=1e-12, err_msg=k)
    assert {"right_link_extremity_frame", "left_link_extremity_frame"} <= set(a.kin_dyn_parameters.frame_names)


def test_sdf_link_frame_offset_and_axis_expressed_in():
    """A child link frame away from its joint frame is folded into the inertial / collision poses (URDF
    frame convention, ``rod/parser.py:76-84``): the dynamics tables do not change.  ``expressed_in``
    re-expresses the joint axis."""
    import jaxsim_amd as ja
    from jaxsim_amd import robots
    from jaxsim_amd.parsers import urdf as up

    a = ja.JaxSimModel.build_from_model_description(robots.double_pendulum_sdf())
    c = ja.JaxSimModel.build_from_model_description(robots.double_pendulum_sdf(link_offset=(0.0, 0.0, 0.3)))
    ta, tc = _tables(a), _tables(c)
    for k in ta:
        np.testing.assert_allclose(np.asarray(ta[k], dtype=float), np.asarray(tc[k], dtype=float), atol=1e-12, err_msg=k)
    # joint frame rolled by -3.1415 about x: an axis given in the model frame as (0, 1, 0) is (0, -1, ~0) in it
    sdf = robots.double_pendulum_sdf().replace("<xyz>1 0 0</xyz>", '<xyz expressed_in="__model__">0 1 0</xyz>', 1)
    d = up.parse_urdf(sdf)
    j = next(j for j in d.joints if j.name == "right_joint")
    np.testing.assert_allclose(j.axis, [0.0, np.cos(-3.1415), -np.sin(-3.1415)], atol=1e-12)
    with pytest.raises(ValueError, match="unknown frame"):
        up.parse_urdf(robots.double_pendulum_sdf().replace('relative_to="right_joint"', 'relative_to="nope"'))
    with pytest.raises(ValueError, match="URDF <robot> or SDF"):
        up.parse_urdf("<foo/>")


@pytest.mark.parametrize("rep", ["inertial", "mixed", "body"])
def test_references_apply_frame_forces(rep):
    """``apply_frame_forces`` (references.py:451-560): a force at a frame equals the same physical wrench
    applied to the parent link; frames of lumped links (cartpole's ``*_frame`` links) carry their pose."""
    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import robots
    from oracle import refstep

    model = ja.JaxSimModel.build_from_model_description(robots.cartpole_urdf())
    kdp = model.kin_dyn_parameters
    assert set(kdp.frame_names) == {"cart_frame", "rail_frame"} and len(kdp.frame_body) == 2
    nL, N = model.number_of_links(), 4
    rng = np.random.default_rng(5)
    W_H_L = _random_transforms(rng, N, nL)
    vr = {"inertial": ja.VelRepr.Inertial, "mixed": ja.VelRepr.Mixed, "body": ja.VelRepr.Body}[rep]
    data = _FakeData(W_H_L, vr)
    refs = js.references.JaxSimModelReferences.zero(model, data=data, velocity_representation=vr)
    refs = refs.apply_link_forces(rng.normal(size=(N, nL, 6)), model=model, data=data)
    f = rng.normal(size=(N, 1, 6))
    k = kdp.frame_names.index("rail_frame")
    out = refs.apply_frame_forces(f, model=model, data=data, frame_names=("rail_frame",))
    W_H_F = W_H_L[:, kdp.frame_body[k]] @ kdp.frame_transform[k]
    expect = np.zeros((N, nL, 6))  # not additive: every other link is reset like in the reference
    expect[:, kdp.frame_body[k]] = refstep.other_representation_to_inertial(f[:, 0], rep, W_H_F, is_force=True)
    np.testing.assert_allclose(out._link_forces, expect, atol=1e-12)
    assert out.velocity_representation == vr
    add = refs.apply_frame_forces(f, model=model, data=data, frame_names="rail_frame", additive=True)
    np.testing.assert_allclose(add._link_forces, refs._link_forces + expect, atol=1e-12)
    with pytest.raises(ValueError, match="must match"):
        refs.apply_frame_forces(np.zeros((N, 2, 6)), model=model, data=data, frame_names=("rail_frame",))
    with pytest.raises(ValueError, match="unknown frame"):
        refs.apply_frame_forces(f, model=model, data=data, frame_names=("nope",))


def test_step_host_path_caches_follow_the_environment(models, monkeypatch):
    """[round 5] The host path of `js.model.step` keeps two things out of the per-call cost -- the value checks of
    `JAXSIM_ENABLE_EXCEPTIONS` (read from the raw environment mapping) and the device copy of the model (a record kept next
    to the device copies, valid while no model field was assigned and the kernel policy is the one it was made under).
    Both must follow a change immediately: `monkeypatch.setenv` / `os.environ[...] = ...`, `model.<field> = ...`."""
    from jaxsim_amd.api import model as M

    monkeypatch.delenv("JAXSIM_ENABLE_EXCEPTIONS", raising=False)
    assert not M._exceptions_enabled()
    for v, want in (("True", True), ("1", True), ("on", True), ("0", False), ("no", False)):
        monkeypatch.setenv("JAXSIM_ENABLE_EXCEPTIONS", v)
        assert M._exceptions_enabled() is want
    calls = []
    monkeypatch.setattr(M.runtime, "device_model", lambda model, dtype: calls.append(1) or object())
    with models("cartpole").editable(validate=False) as m:
        pass
    dt = np.dtype(np.float32)
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "cached")
    a = M._device_model_fast(m, dt)
    assert M._device_model_fast(m, dt) is a and len(calls) == 1          # the hot loop: two dictionary look-ups
    assert M._device_model_fast(m, np.dtype(np.float64)) is not a and len(calls) == 2  # per precision
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")                       # another kernel policy: resolved again
    assert M._device_model_fast(m, dt) is not a and len(calls) == 3
    m.time_step = 2e-3                                                     # a model constant changed: the device copies are dropped
    M._device_model_fast(m, dt)
    assert len(calls) == 4


def test_fp32_relaxed_contacts_with_a_negligible_regulariser_are_refused(models, monkeypatch):
    """[round 6, VERDICT r5 weak 5] RelaxedRigidContacts at the reference's DEFAULT mu = 0.005 has no float32 tolerance
    (HISTORY.md 4e): the product says so with a ValueError instead of returning numbers without a correct digit.  float64,
    mu = 0.5 (estimate_good_contact_parameters), a single point, and the explicit opt-out are accepted."""
    import helpers
    from jaxsim_amd import runtime

    monkeypatch.delenv("JAXSIM_AMD_FP32_RELAXED_UNCHECKED", raising=False)
    box = helpers.relaxed_model(models("box"), [0, 1, 2, 3])  # four points on one link, default parameters
    assert box.contact_params.mu == 0.005
    with pytest.raises(ValueError, match="float32.*mu = 0.005"):
        runtime.fp32_relaxed_defaults_guard(box, np.float32)
    runtime.fp32_relaxed_defaults_guard(box, np.float64)                                                  # the reference's default precision
    runtime.fp32_relaxed_defaults_guard(helpers.relaxed_model(models("box"), [0, 1, 2, 3], mu=0.5), np.float32)
    runtime.fp32_relaxed_defaults_guard(helpers.relaxed_model(models("box"), [0]), np.float32)            # one point: full rank
    runtime.fp32_relaxed_defaults_guard(models("box"), np.float32)                                        # SoftContacts
    with pytest.raises(ValueError):
        runtime.fp32_relaxed_defaults_guard(helpers.relaxed_model(models("anymal"), helpers.ANYMAL_FEET_16), np.float32)
    monkeypatch.setenv("JAXSIM_AMD_FP32_RELAXED_UNCHECKED", "1")
    runtime.fp32_relaxed_defaults_guard(box, np.float32)
