"""Own URDF reader -> reduced model description (build-time, host only).

The reference parses URDF/SDF through the third-party ``rod`` package plus the
``gz sdf`` CLI (``src/jaxsim/parsers/rod/parser.py:36-420``); neither exists
here, so this module reads URDF with ``xml.etree`` and applies the *rules* the
reference applies when it turns a description into its static tables
(SURVEY.md section 8(a) row T):

* links with mass <= 0 are dropped (``rod/parser.py:110-119``) and become
  frames when a fixed joint attaches them to a real link;
* fixed joints are removed by lumping the child link into its parent:
  ``M_parent += X^T M_child X`` with ``X = Ad(parent_H_child)^-1``
  (``parsers/descriptions/link.py:86-115``,
  ``parsers/kinematic_graph.py:379-611``); joints / collision points of the
  removed link are re-expressed in the surviving link
  (``parsers/descriptions/model.py:86-138``);
* a fixed joint whose parent is the ``world`` link marks a fixed-base model and
  its origin is folded into the base link pose (``rod/parser.py:145-199``);
* link index = BFS order from the base link with children sorted by name
  (``parsers/kinematic_graph.py:133-134,668-709``), joint index = child index;
* joint axis is normalised (``parsers/descriptions/joint.py:90-98``); limits
  default to +-finfo.max, ``friction_static <- dynamics/@friction``,
  ``friction_viscous <- dynamics/@damping`` (``rod/parser.py:234-277``);
* collision boxes -> 8 corner points (4 bottom then 4 top), spheres -> 50-point
  Fibonacci lattice, cylinders skipped, meshes -> vertices when JAXSIM_COLLISION_MESH_ENABLED is set
  (``parsers/meshes.py``), skipped otherwise like the reference default
  (``parsers/rod/utils.py:102-225``, ``rod/parser.py:327-357``); points are
  listed link by link in file order, shapes in order.
"""

from __future__ import annotations

import dataclasses
import os
import xml.etree.ElementTree as ET

import numpy as np

from .. import _hostmath as hm
from . import meshes

FIXED, REVOLUTE, PRISMATIC = 0, 1, 2  # src/jaxsim/parsers/descriptions/joint.py JointType


@dataclasses.dataclass
class LinkDescription:
    name: str
    mass: float
    inertia: np.ndarray  # 6x6 spatial inertia at the link frame
    pose: np.ndarray = dataclasses.field(default_factory=lambda: np.eye(4))
    parent_name: str | None = None
    index: int = -1


@dataclasses.dataclass
class JointDescription:
    name: str
    parent: str
    child: str
    jtype: int
    axis: np.ndarray
    pose: np.ndarray  # parent link frame -> joint frame (URDF <origin>)
    position_limit: tuple[float, float]
    friction_static: float = 0.0
    friction_viscous: float = 0.0
    position_limit_damper: float = 0.0
    position_limit_spring: float = 0.0
    index: int = -1


@dataclasses.dataclass
class CollidablePoint:
    parent_link: str
    position: np.ndarray
    enabled: bool = True


@dataclasses.dataclass
class FrameDescription:
    name: str
    parent_name: str
    pose: np.ndarray


@dataclasses.dataclass
class ModelDescription:
    name: str
    fixed_base: bool
    links: list[LinkDescription]  # BFS order, index assigned
    joints: list[JointDescription]  # sorted by index (== child link index)
    collidable_points: list[CollidablePoint]
    frames: list[FrameDescription]

    @property
    def base_link(self) -> LinkDescription:
        return self.links[0]


def _floats(text: str | None, n: int, default: float = 0.0) -> np.ndarray:
    if text is None:
        return np.full(n, default, dtype=float)
    vals = [float(t) for t in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} floats, got {text!r}")
    return np.array(vals, dtype=float)


def _origin(elem) -> np.ndarray:
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(4)
    return hm.transform_from_xyz_rpy(_floats(o.get("xyz"), 3), _floats(o.get("rpy"), 3))


def _link_inertia(inertial) -> tuple[float, np.ndarray]:
    """6D inertia at the link frame from a URDF <inertial> (``rod/utils.py:21-66``)."""
    if inertial is None:
        return 0.0, np.zeros((6, 6))
    m = float(inertial.find("mass").get("value"))
    ie = inertial.find("inertia")
    g = lambda k: float(ie.get(k, 0.0)) if ie is not None else 0.0  # noqa: E731
    I_com = np.array(
        [[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]]
    )
    M_com = hm.inertia_to_sixd(m, np.zeros(3), I_com)
    L_H_CoM = _origin(inertial)
    CoM_X_L = hm.adjoint(L_H_CoM, inverse=True)
    return m, CoM_X_L.T @ M_com @ CoM_X_L


def _box_points(size, H) -> np.ndarray:
    x, y, z = size
    center = np.array([x / 2, y / 2, z / 2])
    bottom = np.array([[0, 0, 0], [x, 0, 0], [x, y, 0], [0, y, 0]], dtype=float)
    use_top = os.environ.get("JAXSIM_COLLISION_USE_BOTTOM_ONLY", "0").lower() in {"false", "0"}
    top = np.array([[0, 0, z], [x, 0, z], [x, y, z], [0, y, z]], dtype=float) if use_top else np.zeros((0, 3))
    corners = np.vstack([bottom, top]) - center
    return (H[:3, :3] @ corners.T).T + H[:3, 3]


def _sphere_points(radius, H) -> np.ndarray:
    samples = int(os.getenv("JAXSIM_COLLISION_SPHERE_POINTS", "50"))
    phi = np.pi * (3.0 - np.sqrt(5.0))
    pts = []
    for i in range(samples):
        y = 1 - 2 * i / (samples - 1)
        r = np.sqrt(1 - y * y)
        pts.append([np.cos(phi * i) * r, y, np.sin(phi * i) * r])
    pts = np.array(pts)
    if os.environ.get("JAXSIM_COLLISION_USE_BOTTOM_ONLY", "0").lower() in {"true", "1"}:
        pts = pts[pts[:, 2] <= 0]
    pts = radius * pts
    return (H[:3, :3] @ pts.T).T + H[:3, 3]


def _joint_motion(jtype: int, axis: np.ndarray, q: float) -> np.ndarray:
    """``pre_H_suc`` of a 1-DoF joint at position ``q`` (``math/joint_model.py:146-200``)."""
    H = np.eye(4)
    if jtype == REVOLUTE:
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        H[:3, :3] = np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
    elif jtype == PRISMATIC:
        H[:3, 3] = q * axis
    return H


def parse_urdf(
    urdf: str,
    *,
    is_path: bool | None = None,
    considered_joints: list[str] | tuple[str, ...] | None = None,
    locked_joint_positions: dict[str, float] | None = None,
    mesh_method=None,
) -> ModelDescription:
    """Parse a URDF (or SDF, see ``parse_sdf``) string or path into a reduced (fixed joints lumped)
    description.

    ``considered_joints`` (``api/model.py:128-223,807-878``): the 1-DoF joints to keep; every other
    joint is locked at ``locked_joint_positions.get(name, 0.0)`` and removed by lumping its child into
    the parent, exactly like a fixed joint (``parsers/kinematic_graph.py:379-611``)."""
    if is_path is None:
        is_path = not urdf.lstrip().startswith("<")
    root = ET.parse(urdf).getroot() if is_path else ET.fromstring(urdf)
    # mesh collisions (JAXSIM_COLLISION_MESH_ENABLED, parsers/meshes.py): relative mesh paths are resolved
    # against the directory of the description file; `mesh_method` selects the points (default: all vertices)
    base_dir = os.path.dirname(os.path.abspath(urdf)) if is_path else os.getcwd()
    if root.tag == "sdf":
        raw = _read_sdf(root, base_dir=base_dir, mesh_method=mesh_method)
    elif root.tag == "robot":
        raw = _read_urdf(root, base_dir=base_dir, mesh_method=mesh_method)
    else:
        raise ValueError("not a URDF <robot> or SDF <sdf> document")
    return _assemble(*raw, considered_joints=considered_joints, locked_joint_positions=locked_joint_positions)


def _read_urdf(root, base_dir=None, mesh_method=None):
    """Raw links / joints / collision points of a URDF ``<robot>``."""
    name = root.get("name", "model")
    # ---- raw links / joints / collisions --------------------------------------------
    raw_links: dict[str, LinkDescription] = {}
    link_order: list[str] = []
    raw_points: list[CollidablePoint] = []
    for le in root.findall("link"):
        lname = le.get("name")
        link_order.append(lname)
        m, M = _link_inertia(le.find("inertial"))
        raw_links[lname] = LinkDescription(name=lname, mass=m, inertia=M)
        for ce in le.findall("collision"):
            geom = ce.find("geometry")
            if geom is None:
                continue
            H = _origin(ce)
            if geom.find("box") is not None:
                pts = _box_points(_floats(geom.find("box").get("size"), 3), H)
            elif geom.find("sphere") is not None:
                pts = _sphere_points(float(geom.find("sphere").get("radius")), H)
            elif geom.find("mesh") is not None and meshes.mesh_collisions_enabled():
                me_ = geom.find("mesh")
                scale = _floats(me_.get("scale"), 3) if me_.get("scale") else np.ones(3)
                pts = meshes.mesh_collision_points(me_.get("filename"), scale, H, base_dir=base_dir, method=mesh_method)
            else:
                continue  # cylinder: skipped; mesh: skipped unless JAXSIM_COLLISION_MESH_ENABLED (rod/parser.py:333-348)
            raw_points += [CollidablePoint(parent_link=lname, position=p) for p in pts]

    fmax = float(np.finfo(float).max)
    raw_joints: list[JointDescription] = []
    for je in root.findall("joint"):
        jt = je.get("type")
        if jt not in {"revolute", "continuous", "prismatic", "fixed"}:
            raise ValueError(f"unsupported joint type {jt!r} (joint {je.get('name')!r})")
        jtype = {"revolute": REVOLUTE, "continuous": REVOLUTE, "prismatic": PRISMATIC, "fixed": FIXED}[jt]
        ax = je.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0.0, 0.0])
        if jtype != FIXED:
            nrm = np.linalg.norm(axis)
            if nrm == 0:
                raise ValueError(f"zero axis in joint {je.get('name')!r}")
            axis = axis / nrm
        lim, dyn = je.find("limit"), je.find("dynamics")
        lo = float(lim.get("lower")) if lim is not None and lim.get("lower") is not None else -fmax
        up = float(lim.get("upper")) if lim is not None and lim.get("upper") is not None else fmax
        if jt == "continuous":
            lo, up = -fmax, fmax
        raw_joints.append(
            JointDescription(
                name=je.get("name"),
                parent=je.find("parent").get("link"),
                child=je.find("child").get("link"),
                jtype=jtype,
                axis=axis,
                pose=_origin(je),
                position_limit=(lo, up),
                friction_static=float(dyn.get("friction", 0.0)) if dyn is not None else 0.0,
                friction_viscous=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
                position_limit_damper=float(os.environ.get("JAXSIM_JOINT_POSITION_LIMIT_DAMPER", 0.0)),
                position_limit_spring=float(os.environ.get("JAXSIM_JOINT_POSITION_LIMIT_SPRING", 0.0)),
            )
        )

    return name, raw_links, link_order, raw_points, raw_joints, []


def _assemble(name, raw_links, link_order, raw_points, raw_joints, extra_frames, *, considered_joints, locked_joint_positions):
    """Reduction, fixed-base folding, lumping of fixed joints, BFS indexing (shared by URDF and SDF).
    ``extra_frames``: explicit frames ``(name, link, link_H_frame)`` of the source document."""
    # ---- model reduction: lock the joints that are not considered --------------------------
    if considered_joints is not None:
        movable = {j.name for j in raw_joints if j.jtype != FIXED}
        unknown = set(considered_joints) - {j.name for j in raw_joints}
        if unknown:
            raise ValueError(f"considered joints not existing in the model: {sorted(unknown)}")
        locked = locked_joint_positions or {}
        if not set(locked).issubset({j.name for j in raw_joints}):
            raise ValueError(f"Passed joints not existing in the model: {sorted(set(locked) - movable)}")
        for j in raw_joints:
            if j.jtype != FIXED and j.name not in set(considered_joints):
                j.pose = j.pose @ _joint_motion(j.jtype, j.axis, float(locked.get(j.name, 0.0)))
                j.jtype = FIXED

    # ---- fixed base: fold the world->base fixed joint into the base pose ---------
    fixed_base = False
    base_name = None
    world_joints = [j for j in raw_joints if j.parent == "world" and j.jtype == FIXED]
    if "world" in raw_links and world_joints:
        if len(world_joints) != 1:
            raise ValueError("Found more/less than one joint connecting a fixed-base model to the world")
        fixed_base = True
        base_name = world_joints[0].child
        raw_links[base_name].pose = world_joints[0].pose @ raw_links[base_name].pose
        raw_joints = [j for j in raw_joints if j is not world_joints[0]]
    raw_links.pop("world", None)

    children_of = {}
    for j in raw_joints:
        children_of.setdefault(j.parent, []).append(j)
    has_parent = {j.child for j in raw_joints}
    if base_name is None:
        roots = [n for n in link_order if n in raw_links and n not in has_parent]
        if len(roots) != 1:
            raise ValueError(f"cannot determine the base link (candidates: {roots})")
        base_name = roots[0]

    # ---- walk the tree from the base; lump fixed joints / massless links ------------
    kept_links: dict[str, LinkDescription] = {}
    kept_joints: list[JointDescription] = []
    frames: list[FrameDescription] = []
    owner: dict[str, tuple[str, np.ndarray]] = {}  # raw link -> (surviving link, survivor_H_raw)

    if raw_links[base_name].mass <= 0:
        raise ValueError("the base link must have a positive mass")

    def visit(lname: str, survivor: str, S_H_l: np.ndarray) -> None:
        owner[lname] = (survivor, S_H_l)
        for j in children_of.get(lname, []):
            child = raw_links[j.child]
            S_H_j = S_H_l @ j.pose  # survivor -> joint (== child link) frame
            if j.jtype == FIXED:
                if child.mass > 0:
                    X = hm.adjoint(S_H_j, inverse=True)
                    kept_links[survivor].mass += child.mass
                    kept_links[survivor].inertia = kept_links[survivor].inertia + X.T @ child.inertia @ X
                frames.append(FrameDescription(name=child.name, parent_name=survivor, pose=S_H_j))
                visit(child.name, survivor, S_H_j)
            else:
                if child.mass <= 0:
                    # The reference drops massless links, which disconnects their subtree.
                    continue
                kept_links[child.name] = LinkDescription(
                    name=child.name, mass=child.mass, inertia=child.inertia.copy(), parent_name=survivor
                )
                kept_joints.append(dataclasses.replace(j, parent=survivor, pose=S_H_j))
                visit(child.name, child.name, np.eye(4))

    kept_links[base_name] = LinkDescription(
        name=base_name,
        mass=raw_links[base_name].mass,
        inertia=raw_links[base_name].inertia.copy(),
        pose=raw_links[base_name].pose,
    )
    visit(base_name, base_name, np.eye(4))

    # ---- BFS indexing, children sorted by name ------------------------------------------
    kids: dict[str, list[str]] = {}
    for j in kept_joints:
        kids.setdefault(j.parent, []).append(j.child)
    order, queue = [base_name], [base_name]
    while queue:
        cur = queue.pop(0)
        for ch in sorted(kids.get(cur, [])):
            order.append(ch)
            queue.append(ch)
    links = []
    for idx, lname in enumerate(order):
        kept_links[lname].index = idx
        links.append(kept_links[lname])
    for j in kept_joints:
        j.index = kept_links[j.child].index
    kept_joints.sort(key=lambda j: j.index)

    # ---- collidable points, re-expressed in the surviving link ---------------------------
    points = []
    for cp in raw_points:
        if cp.parent_link not in owner:
            continue
        survivor, S_H_l = owner[cp.parent_link]
        points.append(
            CollidablePoint(parent_link=survivor, position=S_H_l[:3, :3] @ cp.position + S_H_l[:3, 3])
        )

    for fname, lname, L_H_F in extra_frames:
        if lname in owner:
            survivor, S_H_l = owner[lname]
            frames.append(FrameDescription(name=fname, parent_name=survivor, pose=S_H_l @ L_H_F))

    return ModelDescription(
        name=name, fixed_base=fixed_base, links=links, joints=kept_joints, collidable_points=points, frames=frames
    )


# ---- SDF (the reference reads it through `rod`, src/jaxsim/parsers/rod/parser.py:26-120) -------------------
def _sdf_pose(elem):
    """``<pose relative_to=...>x y z roll pitch yaw</pose>`` of an SDF element -> (4x4, relative_to)."""
    pe = elem.find("pose") if elem is not None else None
    if pe is None:
        return np.eye(4), None
    v = _floats(pe.text, 6)
    return hm.transform_from_xyz_rpy(v[:3], v[3:]), (pe.get("relative_to") or None)


def _read_sdf(root, model_name: str | None = None, base_dir=None, mesh_method=None):
    """Raw links / joints / collision points of the first (or the named) ``<model>`` of an SDF document,
    converted to the URDF frame convention like the reference does
    (``sdf_model.switch_frame_convention(rod.FrameConvention.Urdf)``, ``rod/parser.py:76-84``): the
    frame of a non-root link is the frame of its parent joint, inertial / collision / child-joint poses
    are re-expressed in it.  Pose graph: ``relative_to`` names a link, joint or ``<frame>`` of the model;
    defaults are the model frame for links, the child link for joints, ``attached_to`` for frames.
    Joint axes are expressed in the joint frame unless ``expressed_in`` says otherwise (SDF >= 1.7)."""
    models = list(root.iter("model"))
    if model_name is not None:
        models = [m for m in models if m.get("name") == model_name]
    if not models:
        raise ValueError("no <model> in the SDF document")
    me = models[0]
    name = me.get("name", "model")
    W_H_M, _ = _sdf_pose(me)

    elems = {}  # frame name -> (pose, relative_to default resolved)
    link_elems = {le.get("name"): le for le in me.findall("link")}
    joint_elems = {je.get("name"): je for je in me.findall("joint")}
    for n, le in link_elems.items():
        H, rel = _sdf_pose(le)
        elems[n] = (H, rel or "__model__")
    for n, je in joint_elems.items():
        H, rel = _sdf_pose(je)
        elems[n] = (H, rel or je.find("child").text.strip())
    for fe in me.findall("frame"):
        H, rel = _sdf_pose(fe)
        elems[fe.get("name")] = (H, rel or fe.get("attached_to") or "__model__")

    cache: dict[str, np.ndarray] = {"__model__": np.eye(4), "world": np.linalg.inv(W_H_M)}

    def M_H(frame: str, stack=()) -> np.ndarray:
        if frame not in cache:
            if frame not in elems:
                raise ValueError(f"unknown frame {frame!r} in the SDF pose graph")
            if frame in stack:
                raise ValueError(f"cyclic relative_to chain through {frame!r}")
            H, rel = elems[frame]
            cache[frame] = M_H(rel, stack + (frame,)) @ H
        return cache[frame]

    parent_joint = {je.find("child").text.strip(): n for n, je in joint_elems.items()}

    def frame_of(link: str) -> np.ndarray:  # URDF convention: the parent joint's frame, the link's own for a root
        return M_H(parent_joint[link]) if link in parent_joint else M_H(link)

    raw_links, link_order, raw_points, extra_frames = {}, [], [], []
    for lname, le in link_elems.items():
        link_order.append(lname)
        F_H_L = np.linalg.inv(frame_of(lname)) @ M_H(lname)
        ine = le.find("inertial")
        m, M = 0.0, np.zeros((6, 6))
        if ine is not None and ine.find("mass") is not None:
            m = float(ine.find("mass").text)
            ie = ine.find("inertia")
            g = lambda k: float(ie.find(k).text) if ie is not None and ie.find(k) is not None else 0.0  # noqa: E731
            I_com = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
            F_H_CoM = F_H_L @ _sdf_pose(ine)[0]
            X = hm.adjoint(F_H_CoM, inverse=True)
            M = X.T @ hm.inertia_to_sixd(m, np.zeros(3), I_com) @ X
        raw_links[lname] = LinkDescription(name=lname, mass=m, inertia=M)
        for ce in le.findall("collision"):
            geom = ce.find("geometry")
            if geom is None:
                continue
            H = F_H_L @ _sdf_pose(ce)[0]
            if geom.find("box") is not None:
                pts = _box_points(_floats(geom.find("box").find("size").text, 3), H)
            elif geom.find("sphere") is not None:
                pts = _sphere_points(float(geom.find("sphere").find("radius").text), H)
            elif geom.find("mesh") is not None and meshes.mesh_collisions_enabled():
                me_ = geom.find("mesh")
                sc = me_.find("scale")
                scale = _floats(sc.text, 3) if sc is not None else np.ones(3)
                pts = meshes.mesh_collision_points(me_.find("uri").text.strip(), scale, H, base_dir=base_dir, method=mesh_method)
            else:
                continue  # cylinder: skipped; mesh: skipped unless JAXSIM_COLLISION_MESH_ENABLED (rod/parser.py:333-348)
            raw_points += [CollidablePoint(parent_link=lname, position=p) for p in pts]
    for fe in me.findall("frame"):
        att = fe.get("attached_to")
        while att in elems and att not in link_elems:  # a frame attached to a frame / joint: follow to the link
            att = joint_elems[att].find("child").text.strip() if att in joint_elems else elems[att][1]
        if att in link_elems:
            extra_frames.append((fe.get("name"), att, np.linalg.inv(frame_of(att)) @ M_H(fe.get("name"))))

    fmax = float(np.finfo(float).max)
    raw_joints = []
    if any(je.find("parent").text.strip() == "world" for je in joint_elems.values()):
        raw_links["world"] = LinkDescription(name="world", mass=0.0, inertia=np.zeros((6, 6)))
    for jname, je in joint_elems.items():
        jt = je.get("type")
        if jt not in {"revolute", "continuous", "prismatic", "fixed"}:
            raise ValueError(f"unsupported joint type {jt!r} (joint {jname!r})")
        jtype = {"revolute": REVOLUTE, "continuous": REVOLUTE, "prismatic": PRISMATIC, "fixed": FIXED}[jt]
        parent, child = je.find("parent").text.strip(), je.find("child").text.strip()
        ax = je.find("axis")
        axis = np.array([1.0, 0.0, 0.0])
        lo, up, damping, friction = -fmax, fmax, 0.0, 0.0
        if ax is not None:
            xe = ax.find("xyz")
            if xe is not None:
                axis = _floats(xe.text, 3)
                if xe.get("expressed_in"):
                    R = (np.linalg.inv(M_H(jname)) @ M_H(xe.get("expressed_in")))[:3, :3]
                    axis = R @ axis
            lim, dyn = ax.find("limit"), ax.find("dynamics")
            if lim is not None and lim.find("lower") is not None:
                lo = float(lim.find("lower").text)
            if lim is not None and lim.find("upper") is not None:
                up = float(lim.find("upper").text)
            if dyn is not None and dyn.find("damping") is not None:
                damping = float(dyn.find("damping").text)
            if dyn is not None and dyn.find("friction") is not None:
                friction = float(dyn.find("friction").text)
        if jtype != FIXED:
            nrm = np.linalg.norm(axis)
            if nrm == 0:
                raise ValueError(f"zero axis in joint {jname!r}")
            axis = axis / nrm
        if jt == "continuous":
            lo, up = -fmax, fmax
        P_frame = np.linalg.inv(W_H_M) if parent == "world" else frame_of(parent)
        raw_joints.append(
            JointDescription(
                name=jname, parent=parent, child=child, jtype=jtype, axis=axis,
                pose=np.linalg.inv(P_frame) @ M_H(jname), position_limit=(lo, up),
                friction_static=friction, friction_viscous=damping,
                position_limit_damper=float(os.environ.get("JAXSIM_JOINT_POSITION_LIMIT_DAMPER", 0.0)),
                position_limit_spring=float(os.environ.get("JAXSIM_JOINT_POSITION_LIMIT_SPRING", 0.0)),
            )  # fmt: skip
        )
    return name, raw_links, link_order, raw_points, raw_joints, extra_frames


def parse_sdf(sdf: str, **kwargs) -> ModelDescription:
    """SDF string or path -> reduced description (``parse_urdf`` dispatches on the root tag)."""
    return parse_urdf(sdf, **kwargs)
