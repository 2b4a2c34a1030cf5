"""URDF generators for the benchmark / test models.

The reference pulls iCub/ErgoCub/ANYmal URDFs from pip packages that are not
available offline (``tests/conftest.py:277-316``), so the humanoid and the
quadruped here are **synthetic**: same joint list / topology as the public
robots (``README.md:50-55`` for the 23 iCub joints) with
stated, made-up masses, inertias and sole boxes.  The small models re-express
the reference's own fixtures: box / sphere (``tests/conftest.py:207-274``),
single pendulum (``:370-476``), the cartpole example and the double-pendulum
test asset (numeric values only; emitted here as URDF text by our own code).
"""

from __future__ import annotations

import numpy as np


def _inertial(mass, com=(0, 0, 0), I=(1e-3, 1e-3, 1e-3), rpy=(0, 0, 0)) -> str:
    ixx, iyy, izz = I
    return (
        f'<inertial><origin xyz="{com[0]} {com[1]} {com[2]}" rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/>'
        f'<mass value="{mass}"/>'
        f'<inertia ixx="{ixx}" ixy="0" ixz="0" iyy="{iyy}" iyz="0" izz="{izz}"/></inertial>'
    )


def _box_inertia(m, x, y, z):
    return (m / 12 * (y * y + z * z), m / 12 * (x * x + z * z), m / 12 * (x * x + y * y))


def _box_collision(size, xyz=(0, 0, 0), rpy=(0, 0, 0)) -> str:
    return (
        f'<collision><origin xyz="{xyz[0]} {xyz[1]} {xyz[2]}" rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/>'
        f'<geometry><box size="{size[0]} {size[1]} {size[2]}"/></geometry></collision>'
    )


def _joint(name, jtype, parent, child, xyz, axis, rpy=(0, 0, 0), lower=None, upper=None, damping=0.0, friction=0.0):
    lim = ""
    if jtype != "fixed":
        if lower is not None:
            lim = f'<limit effort="1000" velocity="100" lower="{lower}" upper="{upper}"/>'
        else:
            lim = '<limit effort="1000" velocity="100"/>'
    dyn = f'<dynamics damping="{damping}" friction="{friction}"/>' if (damping or friction) else ""
    ax = f'<axis xyz="{axis[0]} {axis[1]} {axis[2]}"/>' if jtype != "fixed" else ""
    return (
        f'<joint name="{name}" type="{jtype}"><origin xyz="{xyz[0]} {xyz[1]} {xyz[2]}" '
        f'rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/><parent link="{parent}"/><child link="{child}"/>{ax}{lim}{dyn}</joint>'
    )


def box_urdf(size=(0.3, 0.2, 0.1), mass=1.0) -> str:
    """Single-link floating box, 8 corner points (``tests/conftest.py:207-243``)."""
    x, y, z = size
    return (
        '<robot name="box"><link name="box_link">'
        + _inertial(mass, I=_box_inertia(mass, x, y, z))
        + _box_collision(size)
        + "</link></robot>"
    )


def sphere_urdf(radius=0.1, mass=1.0) -> str:
    """Single-link floating sphere, 50 Fibonacci points (``tests/conftest.py:246-274``)."""
    I = 2.0 / 5.0 * mass * radius * radius
    return (
        '<robot name="sphere"><link name="sphere_link">'
        + _inertial(mass, I=(I, I, I))
        + f'<collision><origin xyz="0 0 0" rpy="0 0 0"/><geometry><sphere radius="{radius}"/></geometry></collision>'
        + "</link></robot>"
    )


def single_pendulum_urdf(length=0.5, mass=1.0, lower=None, upper=None) -> str:
    """Fixed-base 2-link pendulum about the x axis (``tests/conftest.py:370-476``)."""
    I = _box_inertia(mass, 0.05, 0.05, length)
    return (
        '<robot name="single_pendulum"><link name="world"/>'
        '<link name="base">' + _inertial(1.0, I=(0.01, 0.01, 0.01)) + "</link>"
        '<link name="link">' + _inertial(mass, com=(0, 0, -length / 2), I=I) + "</link>"
        + _joint("world_to_base", "fixed", "world", "base", (0, 0, 1.0), (0, 0, 0))
        + _joint("pivot", "revolute" if lower is not None else "continuous", "base", "link", (0, 0, 0), (1, 0, 0),
                 lower=lower, upper=upper)
        + "</robot>"
    )


def double_pendulum_urdf(with_base_collision: bool = False) -> str:
    """Fixed base + two independent revolute-x links, joint damping 1.0.

    Same numbers as the reference's ``tests/assets/double_pendulum.sdf`` (base 100 kg,
    links 1 kg with unit inertia and CoM at z=0.5, joints at (+-0.2, 0, 2) rolled by
    -3.1415); BASELINE config C1 ("no contacts") strips the base collision box.
    """
    coll = _box_collision((0.2, 0.2, 2.15), xyz=(0, 0, 1)) if with_base_collision else ""
    link = lambda n: f'<link name="{n}">' + _inertial(1.0, com=(0, 0, 0.5), I=(1.0, 1.0, 1.0)) + "</link>"  # noqa: E731
    return (
        '<robot name="double_pendulum"><link name="world"/>'
        '<link name="base_link">' + _inertial(100.0, I=(1.0, 1.0, 1.0)) + coll + "</link>"
        + link("right_link") + link("left_link")
        + _joint("fixed_base", "fixed", "world", "base_link", (0, 0, 0), (0, 0, 0))
        + _joint("right_joint", "revolute", "base_link", "right_link", (0.2, 0, 2), (1, 0, 0),
                 rpy=(-3.1415, 0, 0), lower=-100, upper=100, damping=1.0)
        + _joint("left_joint", "revolute", "base_link", "left_link", (-0.2, 0, 2), (1, 0, 0),
                 rpy=(-3.1415, 0, 0), lower=-100, upper=100, damping=1.0)
        + "</robot>"
    )


def serial_double_pendulum_urdf(m1=1.3, m2=0.7, L1=0.45, c1=0.2, c2=0.3, I1=0.021, I2=0.013, damping=0.0, friction=0.0) -> str:
    """Fixed base + a SERIAL planar double pendulum about the x axis (links along their local +z): the
    classical coupled two-link arm whose equations of motion are known in closed form.  Link i has mass
    m_i, its CoM c_i along the link from its joint, inertia I_i about the x axis through the CoM; joint 2
    sits L1 along link 1.  Not a reference fixture: it exists for the oracle-independent analytic pins
    (tests/test_oracle_independent.py)."""
    return (
        '<robot name="serial_double_pendulum"><link name="world"/>'
        '<link name="base">' + _inertial(2.0, I=(0.01, 0.01, 0.01)) + "</link>"
        '<link name="upper">' + _inertial(m1, com=(0, 0, c1), I=(I1, 0.5 * I1, 0.7 * I1)) + "</link>"
        '<link name="lower">' + _inertial(m2, com=(0, 0, c2), I=(I2, 0.6 * I2, 0.4 * I2)) + "</link>"
        + _joint("world_to_base", "fixed", "world", "base", (0, 0, 1.5), (0, 0, 0))
        + _joint("shoulder", "continuous", "base", "upper", (0, 0, 0), (1, 0, 0), damping=damping, friction=friction)
        + _joint("elbow", "continuous", "upper", "lower", (0, 0, L1), (1, 0, 0), damping=damping, friction=friction)
        + "</robot>"
    )


def double_pendulum_sdf(with_base_collision: bool = False, link_offset=(0.0, 0.0, 0.0)) -> str:
    """The model of ``double_pendulum_urdf`` written as SDF 1.7 the way the reference's fixture is
    structured (``tests/assets/double_pendulum.sdf``): a fixed world joint, joint poses given
    ``relative_to`` the base link, child link poses ``relative_to`` their joint, explicit ``<frame>``
    elements.  ``link_offset`` moves the child link frames away from their joint frames (identity in the
    reference's file) to exercise the conversion to the URDF frame convention."""
    ox, oy, oz = link_offset
    inertial = lambda m, z: (  # noqa: E731
        f"<inertial><pose>0 0 {z} 0 0 0</pose><mass>{m}</mass>"
        "<inertia><ixx>1.0</ixx><ixy>0</ixy><ixz>0</ixz><iyy>1.0</iyy><iyz>0</iyz><izz>1.0</izz></inertia></inertial>"
    )
    coll = ('<collision name="c"><pose>0 0 1 0 0 0</pose><geometry><box><size>0.2 0.2 2.15</size></box></geometry></collision>'
            if with_base_collision else "")  # fmt: skip

    def arm(side, x):
        return (
            f'<joint name="{side}_joint" type="revolute"><pose relative_to="base_link">{x} 0 2 -3.1415 0 0</pose>'
            f"<parent>base_link</parent><child>{side}_link</child>"
            "<axis><xyz>1 0 0</xyz><limit><lower>-100</lower><upper>100</upper></limit>"
            "<dynamics><damping>1.0</damping></dynamics></axis></joint>"
            f'<link name="{side}_link"><pose relative_to="{side}_joint">{ox} {oy} {oz} 0 0 0</pose>'
            + inertial(1.0, 0.5 - oz if (ox, oy) == (0.0, 0.0) else 0.5)
            + "</link>"
            f'<frame name="{side}_link_extremity_frame" attached_to="{side}_link">'
            f'<pose relative_to="{side}_link">{-x} 0 1 3.14 0 0</pose></frame>'
        )

    return (
        '<?xml version="1.0"?><sdf version="1.7"><model name="double_pendulum">'
        '<joint name="fixed_base" type="fixed"><parent>world</parent><child>base_link</child></joint>'
        '<link name="base_link">' + inertial(100.0, 0) + coll + "</link>"
        + arm("right", 0.2) + arm("left", -0.2)
        + "</model></sdf>"
    )


def cartpole_urdf(with_collisions: bool = False) -> str:
    """rail(0) - prismatic-y -> cart(1) - continuous-x -> pole(2), fixed base.

    Numbers follow the reference example ``examples/assets/cartpole.urdf``; the two
    massless ``*_frame`` links exercise the frame/lumping path.  BASELINE config C2 is
    "ABA + integrator only": no collision shapes unless ``with_collisions``.
    """
    cart_coll = _box_collision((0.1, 0.2, 0.05)) if with_collisions else ""
    return (
        '<robot name="cartpole"><link name="world"/>'
        '<link name="rail">'
        + _inertial(5.0, com=(0, 0, 1.2), rpy=(1.5707963267948963, 0, 0),
                    I=(10.416697916666665, 10.416697916666665, 6.25e-05))
        + "</link>"
        '<link name="cart">'
        + _inertial(1.0, I=(0.0035416666666666674, 0.0010416666666666669, 0.0041666666666666675))
        + cart_coll + "</link>"
        '<link name="pole">'
        + _inertial(0.5, com=(0, 0, 0.5), I=(0.04166979166666667, 0.04166979166666667, 6.25e-06))
        + "</link>"
        '<link name="cart_frame"/><link name="rail_frame"/>'
        + _joint("cart_frame_joint", "fixed", "cart", "cart_frame", (0, 0, 0), (0, 0, 0))
        + _joint("rail_frame_joint", "fixed", "rail", "rail_frame", (0, 0, 1.2), (0, 0, 0))
        + _joint("world_to_rail", "fixed", "world", "rail", (0, 0, 0), (0, 0, 0))
        + _joint("linear", "prismatic", "rail", "cart", (0, 0, 1.2), (0, 1, 0), lower=-2.4, upper=2.4)
        + _joint("pivot", "continuous", "cart", "pole", (0, 0, 0), (1, 0, 0))
        + "</robot>"
    )


# ---------------------------------------------------------------------------------------------
# Synthetic humanoid with the iCub 23-DoF joint list.  SYNTHETIC inertial data.
# ---------------------------------------------------------------------------------------------

ICUB_JOINTS = (
    "torso_pitch", "torso_roll", "torso_yaw",
    "l_shoulder_pitch", "l_shoulder_roll", "l_shoulder_yaw", "l_elbow",
    "r_shoulder_pitch", "r_shoulder_roll", "r_shoulder_yaw", "r_elbow",
    "l_hip_pitch", "l_hip_roll", "l_hip_yaw", "l_knee", "l_ankle_pitch", "l_ankle_roll",
    "r_hip_pitch", "r_hip_roll", "r_hip_yaw", "r_knee", "r_ankle_pitch", "r_ankle_roll",
)  # fmt: skip


def icub23_urdf(sole_boxes_per_foot: int = 2, joint_limit: float = 1.0, joint_damping: float = 1.0,
                joint_friction: float = 0.2) -> str:
    """Synthetic 24-link / 23-DoF floating-base humanoid, total mass ~33 kg.

    ``sole_boxes_per_foot`` boxes per foot, 8 corner points each: 2 -> n_cp = 32 (the
    stated C3/C4 default), 1 -> n_cp = 16.
    Standing height of the root link above the soles is ~0.60 m.  Every joint carries
    viscous damping / Coulomb friction (URDF ``<dynamics>``), like the real robot's URDF.
    """
    out = ['<robot name="icub23_synthetic">']

    def link(name, mass, com, dims, extra=""):
        out.append(f'<link name="{name}">' + _inertial(mass, com=com, I=_box_inertia(mass, *dims)) + extra + "</link>")

    jl = joint_limit
    link("root_link", 5.0, (0, 0, 0.0), (0.15, 0.2, 0.12))
    # torso chain
    link("torso_1", 1.0, (0, 0, 0.02), (0.08, 0.08, 0.06))
    link("torso_2", 1.0, (0, 0, 0.02), (0.08, 0.08, 0.06))
    link("chest", 8.0, (0, 0, 0.12), (0.18, 0.26, 0.25))
    out.append(_joint("torso_pitch", "revolute", "root_link", "torso_1", (0, 0, 0.08), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
    out.append(_joint("torso_roll", "revolute", "torso_1", "torso_2", (0, 0, 0.04), (1, 0, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
    out.append(_joint("torso_yaw", "revolute", "torso_2", "chest", (0, 0, 0.04), (0, 0, 1), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
    for s, sy in (("l", 1.0), ("r", -1.0)):
        link(f"{s}_shoulder_1", 0.5, (0, 0.02 * sy, 0), (0.06, 0.06, 0.06))
        link(f"{s}_shoulder_2", 0.5, (0, 0, -0.02), (0.06, 0.06, 0.06))
        link(f"{s}_shoulder_3", 1.2, (0, 0, -0.08), (0.06, 0.06, 0.16))
        link(f"{s}_forearm", 1.0, (0, 0, -0.08), (0.05, 0.05, 0.18))
        out.append(_joint(f"{s}_shoulder_pitch", "revolute", "chest", f"{s}_shoulder_1", (0, 0.12 * sy, 0.2), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_shoulder_roll", "revolute", f"{s}_shoulder_1", f"{s}_shoulder_2", (0, 0.05 * sy, 0), (1, 0, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_shoulder_yaw", "revolute", f"{s}_shoulder_2", f"{s}_shoulder_3", (0, 0, -0.04), (0, 0, 1), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_elbow", "revolute", f"{s}_shoulder_3", f"{s}_forearm", (0, 0, -0.16), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        # legs
        sole = ""
        if sole_boxes_per_foot >= 1:
            if sole_boxes_per_foot == 1:
                sole = _box_collision((0.18, 0.08, 0.02), xyz=(0.03, 0, -0.05))
            else:
                w = 0.18 / sole_boxes_per_foot
                for k in range(sole_boxes_per_foot):
                    cx = 0.03 - 0.09 + w * (k + 0.5)
                    sole += _box_collision((w, 0.08, 0.02), xyz=(cx, 0, -0.05))
        link(f"{s}_hip_1", 0.8, (0, 0, 0), (0.07, 0.07, 0.07))
        link(f"{s}_hip_2", 0.8, (0, 0, -0.02), (0.07, 0.07, 0.07))
        link(f"{s}_upper_leg", 2.5, (0, 0, -0.11), (0.09, 0.09, 0.24))
        link(f"{s}_lower_leg", 2.0, (0, 0, -0.1), (0.07, 0.07, 0.22))
        link(f"{s}_ankle_1", 0.5, (0, 0, 0), (0.05, 0.05, 0.05))
        link(f"{s}_ankle_2", 0.8, (0.03, 0, -0.04), (0.18, 0.08, 0.04), extra=sole)
        out.append(_joint(f"{s}_hip_pitch", "revolute", "root_link", f"{s}_hip_1", (0, 0.07 * sy, -0.06), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_hip_roll", "revolute", f"{s}_hip_1", f"{s}_hip_2", (0, 0, -0.02), (1, 0, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_hip_yaw", "revolute", f"{s}_hip_2", f"{s}_upper_leg", (0, 0, -0.04), (0, 0, 1), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_knee", "revolute", f"{s}_upper_leg", f"{s}_lower_leg", (0, 0, -0.22), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_ankle_pitch", "revolute", f"{s}_lower_leg", f"{s}_ankle_1", (0, 0, -0.2), (0, 1, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
        out.append(_joint(f"{s}_ankle_roll", "revolute", f"{s}_ankle_1", f"{s}_ankle_2", (0, 0, 0), (1, 0, 0), lower=-jl, upper=jl, damping=joint_damping, friction=joint_friction))
    out.append("</robot>")
    return "".join(out)


def anymal12_urdf(points_per_foot_box: bool = True, joint_limit: float = 1.0, foot_shape: str = "box") -> str:
    """Synthetic 13-link / 12-DoF quadruped (4 x HAA-x / HFE-y / KFE-y), ~50 kg.  ``foot_shape="sphere"``: a sphere
    collision shape of radius 3 cm at every shank tip instead of the box -- 50 Fibonacci points each
    (``parsers/rod/utils.py:200-204``), 200 collidable points like the real robot's URDF."""
    out = ['<robot name="anymal12_synthetic">']
    out.append('<link name="base">' + _inertial(30.0, I=_box_inertia(30.0, 0.6, 0.3, 0.2)) + "</link>")
    jl = joint_limit
    for leg, sx, sy in (("LF", 1, 1), ("RF", 1, -1), ("LH", -1, 1), ("RH", -1, -1)):
        foot = _box_collision((0.04, 0.04, 0.04), xyz=(0, 0, -0.3)) if points_per_foot_box else ""
        if points_per_foot_box and foot_shape == "sphere":
            foot = '<collision><origin xyz="0 0 -0.3" rpy="0 0 0"/><geometry><sphere radius="0.03"/></geometry></collision>'
        out.append(f'<link name="{leg}_HIP">' + _inertial(1.5, com=(0, 0.02 * sy, 0), I=_box_inertia(1.5, 0.1, 0.1, 0.1)) + "</link>")
        out.append(f'<link name="{leg}_THIGH">' + _inertial(2.0, com=(0, 0, -0.14), I=_box_inertia(2.0, 0.06, 0.06, 0.3)) + "</link>")
        out.append(f'<link name="{leg}_SHANK">' + _inertial(1.5, com=(0, 0, -0.14), I=_box_inertia(1.5, 0.05, 0.05, 0.3)) + foot + "</link>")
        out.append(_joint(f"{leg}_HAA", "revolute", "base", f"{leg}_HIP", (0.3 * sx, 0.12 * sy, 0), (1, 0, 0), lower=-jl, upper=jl))
        out.append(_joint(f"{leg}_HFE", "revolute", f"{leg}_HIP", f"{leg}_THIGH", (0, 0.06 * sy, 0), (0, 1, 0), lower=-jl, upper=jl))
        out.append(_joint(f"{leg}_KFE", "revolute", f"{leg}_THIGH", f"{leg}_SHANK", (0, 0, -0.3), (0, 1, 0), lower=-jl, upper=jl))
    out.append("</robot>")
    return "".join(out)


def four_bar_opened_urdf() -> str:
    """The reference's third shipped test asset, ``tests/assets/4_bar_opened.urdf`` (numeric values only, emitted by our
    own code): a planar four-bar linkage cut open at C -- AB (root, floating) -> BC1 (joint B) and AB -> DA (joint A) ->
    CD (joint D) -> BC2 (joint C), every joint revolute about z and turned by 1.57 rad, two massless frame links at the
    cut (``BC1_frame``, ``BC2_frame``: the reference closes the loop there with kinematic constraints, out of scope
    here), one collision box on CD."""
    def bar(name, m, ly, com_y, I, coll=False):
        c = _box_collision((0.1, ly, 0.1), xyz=(0, com_y, 0)) if coll else ""
        return f'<link name="{name}">' + _inertial(m, com=(0, com_y, 0), I=I) + c + "</link>"

    long_I, short_I = (0.02167, 0.00167, 0.02167), (0.010835, 0.000835, 0.010835)
    out = ['<robot name="4_bar_opened">', bar("AB", 1.0, 0.5, 0.0, long_I), bar("BC1", 0.5, 0.25, 0.125, short_I), '<link name="BC1_frame"/>',
           _joint("BC1_frame_joint", "fixed", "BC1", "BC1_frame", (0, 0.25, 0), (0, 0, 0)),
           bar("BC2", 0.5, 0.25, 0.125, short_I), '<link name="BC2_frame"/>',
           _joint("BC2_frame_joint", "fixed", "BC2", "BC2_frame", (0, 0.25, 0), (0, 0, 0), rpy=(0, 0, 3.1416)),
           bar("CD", 1.0, 0.5, 0.25, long_I, coll=True), bar("DA", 1.0, 0.5, 0.25, long_I)]  # fmt: skip
    for name, parent, child, y in (("B", "AB", "BC1", -0.25), ("C", "CD", "BC2", 0.5), ("D", "DA", "CD", 0.5), ("A", "AB", "DA", 0.25)):
        out.append(_joint(name, "revolute", parent, child, (0, y, 0), (0, 0, 1), rpy=(0, 0, 1.57), lower=-1.57, upper=1.57))
    out.append("</robot>")
    return "".join(out)


def planar_biped_urdf(sole=(0.2, 0.08, 0.04)) -> str:
    """Walker2d-style planar biped with a floating base: torso (10 kg) + 2 x (thigh, shank, foot), all six joints
    revolute about y (hip, knee, ankle), a sole box on each foot.  Every joint axis between the two feet is PARALLEL:
    the relative twist of the feet spans 3 dimensions in every configuration, so the 12 x 12 inverse operational-space
    inertia of the two contact links has rank 9 however many joints lie between them -- the model class the round-4
    review used to break the link-space contact solve."""
    out = ['<robot name="planar_biped">']
    out.append('<link name="torso">' + _inertial(10.0, com=(0, 0, 0.15), I=_box_inertia(10.0, 0.2, 0.3, 0.5)) + "</link>")
    for s, sy in (("l", 1), ("r", -1)):
        out.append(f'<link name="{s}_thigh">' + _inertial(3.0, com=(0, 0, -0.2), I=_box_inertia(3.0, 0.1, 0.1, 0.4)) + "</link>")
        out.append(f'<link name="{s}_shank">' + _inertial(2.0, com=(0, 0, -0.2), I=_box_inertia(2.0, 0.08, 0.08, 0.4)) + "</link>")
        out.append(f'<link name="{s}_foot">' + _inertial(1.0, com=(0.04, 0, -0.03), I=_box_inertia(1.0, *sole))
                   + _box_collision(sole, xyz=(0.04, 0, -0.04)) + "</link>")
        out.append(_joint(f"{s}_hip", "revolute", "torso", f"{s}_thigh", (0, 0.1 * sy, -0.1), (0, 1, 0), lower=-1.5, upper=1.5))
        out.append(_joint(f"{s}_knee", "revolute", f"{s}_thigh", f"{s}_shank", (0, 0, -0.4), (0, 1, 0), lower=-1.5, upper=1.5))
        out.append(_joint(f"{s}_ankle", "revolute", f"{s}_shank", f"{s}_foot", (0, 0, -0.4), (0, 1, 0), lower=-1.5, upper=1.5))
    out.append("</robot>")
    return "".join(out)


def hub_urdf(n_legs: int = 8, links_per_leg: int = 2, foot_boxes: int = 4, seed: int = 0) -> str:
    """A floating hub with ``n_legs`` legs of ``links_per_leg`` links each -- an octopod for the default: MORE THAN SIX
    children on one link (kMaxChildren of csrc/jxs_params.h was 6 through round 5; VERDICT r5 missing 3).  The first
    ``foot_boxes`` legs end in a collision box.  Joint axes alternate between the leg's tangent and the vertical."""
    rng = np.random.default_rng(seed)
    out = ['<robot name="hub">']
    out.append('<link name="body">' + _inertial(8.0, com=(0.0, 0.0, 0.02), I=_box_inertia(8.0, 0.4, 0.4, 0.12)) + "</link>")
    for leg in range(n_legs):
        ang = 2.0 * np.pi * leg / n_legs
        cx, sy = float(np.cos(ang)), float(np.sin(ang))
        parent = "body"
        for k in range(links_per_leg):
            name = f"leg{leg:02d}_{k}"
            m = float(rng.uniform(0.4, 1.2))
            ln_ = float(rng.uniform(0.18, 0.3))
            last = k == links_per_leg - 1
            coll = _box_collision((0.06, 0.06, 0.04), xyz=(0.0, 0.0, -ln_)) if (last and leg < foot_boxes) else ""
            out.append(f'<link name="{name}">' + _inertial(m, com=(0.0, 0.0, -0.5 * ln_), I=_box_inertia(m, 0.05, 0.05, ln_)) + coll + "</link>")
            xyz = (0.25 * cx, 0.25 * sy, -0.03) if k == 0 else (0.0, 0.0, -prev_len)
            axis = (-sy, cx, 0.0) if k % 2 == 0 else (0.0, 0.0, 1.0)
            if k >= 2:
                axis = (cx, sy, 0.0)
            out.append(_joint(f"j_{name}", "revolute", parent, name, xyz, axis, rpy=(0.0, 0.0, float(rng.uniform(-0.3, 0.3))),
                              lower=-1.2, upper=1.2, damping=float(rng.uniform(0, 0.1)), friction=float(rng.uniform(0, 0.05))))
            parent, prev_len = name, ln_
    out.append("</robot>")
    return "".join(out)


def lumped_tree_urdf(n_links: int = 7, seed: int = 0, fixed_base: bool = False) -> str:
    """Random tree whose every moving link carries one or two MASSIVE bodies on FIXED joints (sensor boxes, covers: offset,
    rotated joint frame, rotated inertial frame) and a massless leaf frame: the lumping rule of the parser
    (``kinematic_graph.py:379-611``: I += X^T I_removed X) under test -- tests/maxcoord.py treats the same fixed joints as
    six constraints between separate bodies and never lumps."""
    rng = np.random.default_rng(seed)
    out = ['<robot name="lumped_tree">']
    if fixed_base:
        out.append('<link name="world"/>')

    def body(name):
        m = float(rng.uniform(0.3, 2.0))
        com = tuple(float(v) for v in rng.uniform(-0.08, 0.08, 3))
        dims = tuple(float(v) for v in rng.uniform(0.05, 0.25, 3))
        rpy = tuple(float(v) for v in rng.uniform(-0.6, 0.6, 3))
        return f'<link name="{name}">' + _inertial(m, com=com, I=_box_inertia(m, *dims), rpy=rpy) + "</link>"

    for i in range(n_links):
        out.append(body(f"link{i:02d}"))
        for k in range(int(rng.integers(1, 3))):
            out.append(body(f"link{i:02d}_payload{k}"))
            out.append(_joint(f"link{i:02d}_mount{k}", "fixed", f"link{i:02d}", f"link{i:02d}_payload{k}",
                              tuple(float(v) for v in rng.uniform(-0.15, 0.15, 3)), (0, 0, 0), rpy=tuple(float(v) for v in rng.uniform(-1.0, 1.0, 3))))
        out.append(f'<link name="link{i:02d}_frame"/>')
        out.append(_joint(f"link{i:02d}_frame_joint", "fixed", f"link{i:02d}", f"link{i:02d}_frame", (0.05, 0.0, 0.1), (0, 0, 0), rpy=(0.1, 0.2, 0.3)))
    if fixed_base:
        out.append(_joint("world_to_base", "fixed", "world", "link00", (0, 0, 0), (0, 0, 0)))
    for i in range(1, n_links):
        parent = int(rng.integers(max(0, i - 3), i))
        jt = "prismatic" if rng.uniform() < 0.25 else "revolute"
        axis = rng.normal(size=3)
        axis = tuple(float(v) for v in axis / np.linalg.norm(axis))
        out.append(_joint(f"joint{i:02d}", jt, f"link{parent:02d}", f"link{i:02d}", tuple(float(v) for v in rng.uniform(-0.3, 0.3, 3)), axis,
                          rpy=tuple(float(v) for v in rng.uniform(-1.0, 1.0, 3)), lower=-1.5, upper=1.5))
    out.append("</robot>")
    return "".join(out)


def chain_urdf(n_links: int = 5, fixed_base: bool = True, seed: int = 0, max_back: int = 3, collision_links=None,
               parallel_axes: str | None = None, base_offset=(0.1, -0.2, 0.5)) -> str:
    """Random serial/branching chain with mixed revolute/prismatic joints, skewed axes and
    rotated joint frames: a stress model for parity tests (cf. the reference's scalable
    "garpez" fixture, ``tests/conftest.py:479-707``).  The parent of link i is one of the ``max_back``
    previous links (1: a serial chain of depth n_links - 1).  ``collision_links``: the links that carry a collision
    box (default: the first and the last link of a floating chain, none of a fixed one).
    ``parallel_axes``: ``"all"`` -- every joint revolute about x with unrotated joint frames (a planar mechanism: the
    relative twist of any two links spans 3 dimensions); ``"aligned"`` -- revolute joints about one of the coordinate
    axes, unrotated frames (runs of parallel axes, as in real robots); ``None`` -- axes in general position.
    ``base_offset``: where the fixed joint places the base link of a fixed-base chain in the world (the rigid contact
    models refuse a base-link offset: their fuzz campaigns pass zeros)."""
    rng = np.random.default_rng(seed)
    out = ['<robot name="chain">']
    if collision_links is None:
        collision_links = () if fixed_base else (0, n_links - 1)
    if fixed_base:
        out.append('<link name="world"/>')
    for i in range(n_links):
        m = float(rng.uniform(0.5, 2.0))
        com = tuple(float(v) for v in rng.uniform(-0.1, 0.1, 3))
        dims = tuple(float(v) for v in rng.uniform(0.05, 0.3, 3))
        rpy = tuple(float(v) for v in rng.uniform(-0.5, 0.5, 3))
        coll = _box_collision(dims, xyz=com) if i in collision_links else ""
        out.append(f'<link name="link{i:02d}">' + _inertial(m, com=com, I=_box_inertia(m, *dims), rpy=rpy) + coll + "</link>")
    if fixed_base:
        out.append(_joint("world_to_base", "fixed", "world", "link00", tuple(base_offset), (0, 0, 0)))
    for i in range(1, n_links):
        parent = int(rng.integers(max(0, i - max_back), i))
        jt = "prismatic" if rng.uniform() < 0.25 else "revolute"
        axis = rng.normal(size=3)
        axis = tuple(float(v) for v in axis / np.linalg.norm(axis))
        xyz = tuple(float(v) for v in rng.uniform(-0.3, 0.3, 3))
        rpy = tuple(float(v) for v in rng.uniform(-1.0, 1.0, 3))
        if parallel_axes == "all":
            jt, axis, rpy = "revolute", (1.0, 0.0, 0.0), (0.0, 0.0, 0.0)
        elif parallel_axes == "aligned":
            jt, rpy = "revolute", (0.0, 0.0, 0.0)
            axis = tuple(float(v) for v in np.eye(3)[int(rng.integers(0, 3))])
        out.append(_joint(f"joint{i:02d}", jt, f"link{parent:02d}", f"link{i:02d}", xyz, axis, rpy=rpy,
                          lower=-1.5, upper=1.5, damping=float(rng.uniform(0, 0.2)), friction=float(rng.uniform(0, 0.1))))
    out.append("</robot>")
    return "".join(out)
