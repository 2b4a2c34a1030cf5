"""Oracle: batched NumPy restatement of the reference ``js.model.step()`` path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``; parity unpinned by execution,
pinned analytically).  The code is deliberately structured like the reference --
dense 6x6 matrices, body-frame recursions, the same order of operations -- so it can
be audited line by line against the cited sources.  Every function is batched over a
leading axis ``N`` (the reference gets that axis from ``jax.vmap``,
``tests/test_benchmark.py:24-35``).

``model`` is duck-typed: any object exposing ``kin_dyn_parameters`` (the NumPy tables
of ``jaxsim_amd.kin_dyn_parameters.KinDynParameters``), ``time_step``, ``gravity``
(signed z acceleration, -9.81 by default), ``contact_params`` (K, D, mu, p, q),
``actuation_params`` (torque_max, omega_th, omega_max, enable_friction),
``terrain`` (``height(x, y)`` / ``normal(x, y)``) and ``floating_base()``.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from . import refmath as rm

__all__ = [
    "OracleData",
    "VelRepr",
    "aba",
    "rnea",
    "crba",
    "forward_kinematics_model",
    "joint_transforms",
    "collidable_points_pos_vel",
    "hunt_crossley_contact_model",
    "compute_contact_forces",
    "link_contact_forces",
    "compute_resultant_torques",
    "tn_curve_fn",
    "system_acceleration",
    "semi_implicit_euler_integration",
    "step",
    "forward_dynamics_aba",
    "inverse_dynamics",
    "free_floating_bias_forces",
    "free_floating_gravity_forces",
    "free_floating_mass_matrix",
    "inertial_to_other_representation",
    "other_representation_to_inertial",
    "build_default_contact_params",
    "estimate_good_contact_parameters",
    "random_model_data",
    "com_position",
]


class VelRepr:  # src/jaxsim/api/common.py:39-47
    Inertial = "inertial"
    Body = "body"
    Mixed = "mixed"


# =============================================================================================
# Representation conversions (src/jaxsim/api/common.py:100-222)
# =============================================================================================


def inertial_to_other_representation(array, other_representation, transform, *, is_force):
    W_array, W_H_O = array, transform
    if other_representation == VelRepr.Inertial:
        return W_array
    if other_representation == VelRepr.Mixed:
        W_H_O = W_H_O.copy()
        W_H_O[..., :3, :3] = np.eye(3)
    elif other_representation != VelRepr.Body:
        raise ValueError(other_representation)
    if not is_force:
        O_X_W = rm.adjoint_from_transform(W_H_O, inverse=True)
        return rm.mv(O_X_W, W_array)
    O_Xf_W = np.swapaxes(rm.adjoint_from_transform(W_H_O), -1, -2)
    return rm.mv(O_Xf_W, W_array)


def other_representation_to_inertial(array, other_representation, transform, *, is_force):
    O_array, W_H_O = array, transform
    if other_representation == VelRepr.Inertial:
        return O_array
    if other_representation == VelRepr.Mixed:
        W_H_O = W_H_O.copy()
        W_H_O[..., :3, :3] = np.eye(3)
    elif other_representation != VelRepr.Body:
        raise ValueError(other_representation)
    if not is_force:
        return rm.mv(rm.adjoint_from_transform(W_H_O), O_array)
    W_Xf_O = np.swapaxes(rm.adjoint_from_transform(W_H_O, inverse=True), -1, -2)
    return rm.mv(W_Xf_O, O_array)


# =============================================================================================
# Joint transforms (src/jaxsim/api/kin_dyn_parameters.py:396-451,
#                   src/jaxsim/math/joint_model.py:146-200)
# =============================================================================================


def _supported_joint_motion(joint_types, s, axes, dtype):
    """pre_H_suc of every joint: Revolute = Rodrigues, Prismatic = translation."""
    N, n = s.shape
    H = np.zeros((N, n, 4, 4), dtype=dtype)
    H[..., 3, 3] = 1
    R_rev = rm.rotation_from_axis_angle(s[..., None] * axes[None].astype(dtype))  # [N,n,3,3]
    is_rev = (joint_types == 1)[None, :, None, None]
    H[..., :3, :3] = np.where(is_rev, R_rev, np.eye(3, dtype=dtype))
    is_pri = (joint_types == 2)[None, :, None]
    H[..., :3, 3] = np.where(is_pri, s[..., None] * axes[None].astype(dtype), 0)
    return H


def joint_transforms(model, joint_positions, base_transform):
    """``i_X_lambda(i)`` for i = 0..nL-1, entry 0 = ``B_X_W`` (incl. ``suc_H_i[0]``)."""
    kdp = model.kin_dyn_parameters
    dtype = joint_positions.dtype
    N = joint_positions.shape[0]
    lam_H_pre = np.broadcast_to(kdp.lambda_H_pre.astype(dtype), (N,) + kdp.lambda_H_pre.shape)
    suc_H_i = np.broadcast_to(kdp.suc_H_i.astype(dtype), (N,) + kdp.suc_H_i.shape)
    pre_H_suc_J = _supported_joint_motion(kdp.joint_types, joint_positions, kdp.joint_axis, dtype)
    pre_H_suc = np.concatenate([base_transform[:, None], pre_H_suc_J], axis=1)
    H = lam_H_pre @ pre_H_suc @ suc_H_i
    return rm.adjoint_from_transform(H, inverse=True)


def _process_inputs(model, dtype, N, standard_gravity):
    """The part of ``process_inputs`` that matters here (``src/jaxsim/rbda/utils.py:148-153``)."""
    W_g = np.zeros((N, 6), dtype=dtype)
    W_g[:, 2] = standard_gravity
    return W_g


def _link_spatial_inertia(model, dtype):
    """``link_spatial_inertia_matrices`` (``src/jaxsim/api/model.py:902-917``)."""
    kdp = model.kin_dyn_parameters
    return rm.inertia_to_sixd(
        kdp.link_mass.astype(dtype), kdp.link_com.astype(dtype), kdp.link_inertia_com.astype(dtype)
    )


# =============================================================================================
# ABA (src/jaxsim/rbda/aba.py:12-292)
# =============================================================================================


def aba(
    model,
    *,
    base_position,
    base_quaternion,
    joint_positions,
    base_linear_velocity,
    base_angular_velocity,
    joint_velocities,
    joint_forces=None,
    link_forces=None,
    standard_gravity=rm.STANDARD_GRAVITY,
):
    kdp = model.kin_dyn_parameters
    s, sd = joint_positions, joint_velocities
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    tau = joint_forces if joint_forces is not None else np.zeros_like(s)
    W_f = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    W_g = _process_inputs(model, dtype, N, standard_gravity)
    W_v_WB = np.concatenate([base_linear_velocity, base_angular_velocity], axis=-1)

    M = _link_spatial_inertia(model, dtype)  # [nL,6,6]
    lam = kdp.parent_array

    W_H_B = rm.transform_from_quaternion_translation(base_quaternion, base_position)
    W_X_B = rm.adjoint_from_transform(W_H_B)  # aba.py:85
    B_X_W = rm.adjoint_from_transform(W_H_B, inverse=True)  # aba.py:86
    i_X_lam = joint_transforms(model, s, W_H_B)  # aba.py:91-93
    S = kdp.motion_subspaces.astype(dtype)  # [nL,6]

    v = np.zeros((N, nL, 6), dtype=dtype)
    c = np.zeros((N, nL, 6), dtype=dtype)
    pA = np.zeros((N, nL, 6), dtype=dtype)
    MA = np.zeros((N, nL, 6, 6), dtype=dtype)
    i_X_0 = np.zeros((N, nL, 6, 6), dtype=dtype)
    i_X_0[:, 0] = np.eye(6, dtype=dtype)

    if model.floating_base():  # aba.py:109-121
        v[:, 0] = rm.mv(B_X_W, W_v_WB)
        MA[:, 0] = M[0]
        pA[:, 0] = rm.mv(rm.vx_star(v[:, 0]) @ MA[:, 0], v[:, 0]) - rm.mv(np.swapaxes(W_X_B, -1, -2), W_f[:, 0])

    # Pass 1 (aba.py:131-161)
    for i in range(1, nL):
        ii = i - 1
        vJ = S[i] * sd[:, ii, None]
        v[:, i] = rm.mv(i_X_lam[:, i], v[:, lam[i]]) + vJ
        c[:, i] = rm.mv(rm.vx(v[:, i]), vJ)
        MA[:, i] = M[i]
        i_X_0[:, i] = i_X_lam[:, i] @ i_X_0[:, lam[i]]
        i_Xf_W = np.swapaxes(rm.adjoint_inverse(i_X_0[:, i] @ B_X_W), -1, -2)
        pA[:, i] = rm.mv(rm.vx_star(v[:, i]) @ M[i], v[:, i]) - rm.mv(i_Xf_W, W_f[:, i])

    # Pass 2 (aba.py:184-224)
    U = np.zeros((N, nL, 6), dtype=dtype)
    d = np.zeros((N, nL), dtype=dtype)
    u = np.zeros((N, nL), dtype=dtype)
    for i in range(nL - 1, 0, -1):
        ii = i - 1
        U[:, i] = rm.mv(MA[:, i], S[i])
        d[:, i] = U[:, i] @ S[i]
        u[:, i] = tau[:, ii] - pA[:, i] @ S[i]
        Ma = MA[:, i] - (U[:, i] / d[:, i, None])[:, :, None] * U[:, i][:, None, :]
        pa = pA[:, i] + rm.mv(Ma, c[:, i]) + U[:, i] * (u[:, i] / d[:, i])[:, None]
        if lam[i] != 0 or model.floating_base():
            Xt = np.swapaxes(i_X_lam[:, i], -1, -2)
            MA[:, lam[i]] = MA[:, lam[i]] + Xt @ Ma @ i_X_lam[:, i]
            pA[:, lam[i]] = pA[:, lam[i]] + rm.mv(Xt, pa)

    # Pass 3 (aba.py:240-267)
    if model.floating_base():
        a0 = np.linalg.solve(-MA[:, 0], pA[:, 0][..., None])[..., 0]
    else:
        a0 = -rm.mv(B_X_W, W_g)
    sdd = np.zeros_like(s)
    a = np.zeros((N, nL, 6), dtype=dtype)
    a[:, 0] = a0
    for i in range(1, nL):
        ii = i - 1
        a_i = rm.mv(i_X_lam[:, i], a[:, lam[i]]) + c[:, i]
        sdd[:, ii] = (u[:, i] - np.einsum("nj,nj->n", U[:, i], a_i)) / d[:, i]
        a[:, i] = a_i + S[i] * sdd[:, ii, None]

    if model.floating_base():  # aba.py:284-292
        W_a_WB = rm.mv(W_X_B, a[:, 0]) + W_g
    else:
        W_a_WB = np.zeros((N, 6), dtype=dtype)
    return W_a_WB.astype(dtype), sdd.astype(dtype)


# =============================================================================================
# RNEA (src/jaxsim/rbda/rnea.py:12-238)
# =============================================================================================


def rnea(
    model,
    *,
    base_position,
    base_quaternion,
    joint_positions,
    base_linear_velocity,
    base_angular_velocity,
    joint_velocities,
    base_linear_acceleration=None,
    base_angular_acceleration=None,
    joint_accelerations=None,
    link_forces=None,
    standard_gravity=rm.STANDARD_GRAVITY,
):
    kdp = model.kin_dyn_parameters
    s, sd = joint_positions, joint_velocities
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    sdd = joint_accelerations if joint_accelerations is not None else np.zeros_like(s)
    W_f = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    z3 = np.zeros((N, 3), dtype=dtype)
    W_vd_WB = np.concatenate(
        [
            base_linear_acceleration if base_linear_acceleration is not None else z3,
            base_angular_acceleration if base_angular_acceleration is not None else z3,
        ],
        axis=-1,
    )
    W_g = _process_inputs(model, dtype, N, standard_gravity)
    W_v_WB = np.concatenate([base_linear_velocity, base_angular_velocity], axis=-1)

    M = _link_spatial_inertia(model, dtype)
    lam = kdp.parent_array
    W_H_B = rm.transform_from_quaternion_translation(base_quaternion, base_position)
    W_X_B = rm.adjoint_from_transform(W_H_B)
    B_X_W = rm.adjoint_from_transform(W_H_B, inverse=True)
    i_X_lam = joint_transforms(model, s, W_H_B)
    S = kdp.motion_subspaces.astype(dtype)

    v = np.zeros((N, nL, 6), dtype=dtype)
    a = np.zeros((N, nL, 6), dtype=dtype)
    f = np.zeros((N, nL, 6), dtype=dtype)
    i_X_0 = np.zeros((N, nL, 6, 6), dtype=dtype)
    i_X_0[:, 0] = np.eye(6, dtype=dtype)

    a[:, 0] = -rm.mv(B_X_W, W_g)  # rnea.py:111-112
    if model.floating_base():  # rnea.py:114-131
        v[:, 0] = rm.mv(B_X_W, W_v_WB)
        a[:, 0] = rm.mv(B_X_W, W_vd_WB - W_g)
        f[:, 0] = (
            rm.mv(M[0], a[:, 0])
            + rm.mv(rm.vx_star(v[:, 0]) @ M[0], v[:, 0])
            - rm.mv(np.swapaxes(W_X_B, -1, -2), W_f[:, 0])
        )

    for i in range(1, nL):  # rnea.py:139-172
        ii = i - 1
        vJ = S[i] * sd[:, ii, None]
        v[:, i] = rm.mv(i_X_lam[:, i], v[:, lam[i]]) + vJ
        a[:, i] = rm.mv(i_X_lam[:, i], a[:, lam[i]]) + S[i] * sdd[:, ii, None] + rm.mv(rm.vx(v[:, i]), vJ)
        i_X_0[:, i] = i_X_lam[:, i] @ i_X_0[:, lam[i]]
        i_Xf_W = np.swapaxes(rm.adjoint_inverse(i_X_0[:, i] @ B_X_W), -1, -2)
        f[:, i] = rm.mv(M[i], a[:, i]) + rm.mv(rm.vx_star(v[:, i]) @ M[i], v[:, i]) - rm.mv(i_Xf_W, W_f[:, i])

    tau = np.zeros_like(s)
    for i in range(nL - 1, 0, -1):  # rnea.py:193-219
        ii = i - 1
        tau[:, ii] = f[:, i] @ S[i]
        if lam[i] != 0 or model.floating_base():
            f[:, lam[i]] = f[:, lam[i]] + rm.mv(np.swapaxes(i_X_lam[:, i], -1, -2), f[:, i])

    W_f0 = rm.mv(np.swapaxes(B_X_W, -1, -2), f[:, 0])  # rnea.py:236
    return W_f0.astype(dtype), tau.astype(dtype)


# =============================================================================================
# CRBA (src/jaxsim/rbda/crba.py:10-170) -- used by the ABA == CRB-solve self-consistency test
# =============================================================================================


def crba(model, *, joint_positions):
    """Body-fixed free-floating mass matrix ``M(s)`` of shape ``[N, 6+n, 6+n]``."""
    kdp = model.kin_dyn_parameters
    s = joint_positions
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    n = nL - 1
    lam = kdp.parent_array
    S = kdp.motion_subspaces.astype(dtype)
    Mlink = _link_spatial_inertia(model, dtype)
    eye4 = np.broadcast_to(np.eye(4, dtype=dtype), (N, 4, 4))
    i_X_lam = joint_transforms(model, s, eye4)  # crba.py:32-35 (identity base transform)

    Mc = np.broadcast_to(Mlink, (N, nL, 6, 6)).copy()
    for i in range(nL - 1, 0, -1):  # composite rigid body inertias
        Xt = np.swapaxes(i_X_lam[:, i], -1, -2)
        Mc[:, lam[i]] = Mc[:, lam[i]] + Xt @ Mc[:, i] @ i_X_lam[:, i]

    Mm = np.zeros((N, 6 + n, 6 + n), dtype=dtype)
    Mm[:, :6, :6] = Mc[:, 0]
    for i in range(1, nL):
        ii = i - 1
        Fi = rm.mv(Mc[:, i], S[i])
        Mm[:, 6 + ii, 6 + ii] = Fi @ S[i]
        j = i
        while lam[j] > 0:
            Fi = rm.mv(np.swapaxes(i_X_lam[:, j], -1, -2), Fi)
            j = lam[j]
            jj = j - 1
            Mm[:, 6 + ii, 6 + jj] = Fi @ S[j]
            Mm[:, 6 + jj, 6 + ii] = Mm[:, 6 + ii, 6 + jj]
        Fi = rm.mv(np.swapaxes(i_X_lam[:, j], -1, -2), Fi)
        Mm[:, :6, 6 + ii] = Fi
        Mm[:, 6 + ii, :6] = Fi
    return Mm


# =============================================================================================
# Forward kinematics (src/jaxsim/rbda/forward_kinematics.py:12-113)
# =============================================================================================


def forward_kinematics_model(
    model,
    *,
    base_position,
    base_quaternion,
    joint_positions,
    base_linear_velocity_inertial,
    base_angular_velocity_inertial,
    joint_velocities,
):
    kdp = model.kin_dyn_parameters
    s, sd = joint_positions, joint_velocities
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    lam = kdp.parent_array
    S = kdp.motion_subspaces.astype(dtype)
    W_H_B = rm.transform_from_quaternion_translation(base_quaternion, base_position)
    i_X_lam = joint_transforms(model, s, W_H_B)

    W_X_i = np.zeros((N, nL, 6, 6), dtype=dtype)
    W_X_i[:, 0] = rm.adjoint_inverse(i_X_lam[:, 0])
    W_v_Wi = np.zeros((N, nL, 6), dtype=dtype)
    W_v_Wi[:, 0] = np.concatenate([base_linear_velocity_inertial, base_angular_velocity_inertial], axis=-1)
    for i in range(1, nL):
        ii = i - 1
        W_X_i[:, i] = W_X_i[:, lam[i]] @ rm.adjoint_inverse(i_X_lam[:, i])
        W_v_Wi[:, i] = W_v_Wi[:, lam[i]] + rm.mv(W_X_i[:, i], S[i] * sd[:, ii, None])
    return rm.adjoint_to_transform(W_X_i), W_v_Wi


# =============================================================================================
# State container (src/jaxsim/api/data.py:46-523)
# =============================================================================================


@dataclasses.dataclass
class OracleData:
    """Batched restatement of ``JaxSimModelData``: base velocity stored inertial-fixed."""

    base_position: np.ndarray  # [N,3]
    base_quaternion: np.ndarray  # [N,4] wxyz
    joint_positions: np.ndarray  # [N,n]
    base_linear_velocity: np.ndarray  # [N,3] inertial-fixed
    base_angular_velocity: np.ndarray  # [N,3]
    joint_velocities: np.ndarray  # [N,n]
    tangential_deformation: np.ndarray  # [N,n_cp,3]
    link_transforms: np.ndarray = None  # cache [N,nL,4,4]
    link_velocities: np.ndarray = None  # cache [N,nL,6] inertial-fixed
    velocity_representation: str = VelRepr.Mixed

    @property
    def dtype(self):
        return self.joint_positions.dtype

    @property
    def batch_size(self):
        return self.base_position.shape[0]

    @property
    def base_orientation(self):  # data.py:267-286
        q = self.base_quaternion
        norm = rm.safe_norm(q, keepdims=True)
        return q / (norm + np.finfo(self.dtype).eps * (norm == 0))

    @property
    def base_transform(self):
        return rm.transform_from_quaternion_translation(self.base_quaternion, self.base_position)

    def base_velocity(self, representation=None):  # data.py:288-312
        rep = representation or self.velocity_representation
        W_v = np.concatenate([self.base_linear_velocity, self.base_angular_velocity], axis=-1)
        return inertial_to_other_representation(W_v, rep, self.base_transform, is_force=False)

    def generalized_velocity(self, representation=None):  # data.py:326-341
        return np.concatenate([self.base_velocity(representation), self.joint_velocities], axis=-1)

    @staticmethod
    def build(
        model,
        *,
        base_position=None,
        base_quaternion=None,
        joint_positions=None,
        base_linear_velocity=None,
        base_angular_velocity=None,
        joint_velocities=None,
        tangential_deformation=None,
        velocity_representation=VelRepr.Mixed,
        batch_size=1,
        dtype=np.float64,
    ) -> "OracleData":
        """``JaxSimModelData.build`` (data.py:65-202): input base velocity is in the given
        representation and is stored inertial-fixed."""
        kdp = model.kin_dyn_parameters
        n, n_cp = kdp.number_of_joints(), kdp.number_of_collidable_points()
        N = batch_size
        for arr in (base_position, base_quaternion, joint_positions, joint_velocities):
            if arr is not None and np.ndim(arr) == 2:
                N = np.shape(arr)[0]

        def prep(x, shape, default=0.0):
            if x is None:
                out = np.full((N,) + shape, default, dtype=dtype)
                return out
            return np.broadcast_to(np.asarray(x, dtype=dtype), (N,) + shape).copy()

        p = prep(base_position, (3,))
        if base_quaternion is None:
            q = np.zeros((N, 4), dtype=dtype)
            q[:, 0] = 1
        else:
            q = prep(base_quaternion, (4,))
        s = prep(joint_positions, (n,))
        sd = prep(joint_velocities, (n,))
        vl = prep(base_linear_velocity, (3,))
        va = prep(base_angular_velocity, (3,))
        W_H_B = rm.transform_from_quaternion_translation(q, p)
        W_v = other_representation_to_inertial(
            np.concatenate([vl, va], axis=-1), velocity_representation, W_H_B, is_force=False
        ).astype(dtype)
        m = prep(tangential_deformation, (n_cp, 3))
        data = OracleData(p, q, s, W_v[:, :3].copy(), W_v[:, 3:].copy(), sd, m,
                          velocity_representation=velocity_representation)
        return data.update_caches(model)

    def update_caches(self, model) -> "OracleData":
        """The cache refresh of ``JaxSimModelData.replace`` (data.py:405-523)."""
        q = self.base_quaternion
        norm = rm.safe_norm(q, keepdims=True)
        q = q / np.where(norm == 0, 1.0, norm).astype(self.dtype)  # data.py:434-440
        H, V = forward_kinematics_model(
            model,
            base_position=self.base_position,
            base_quaternion=q,
            joint_positions=self.joint_positions,
            base_linear_velocity_inertial=self.base_linear_velocity,
            base_angular_velocity_inertial=self.base_angular_velocity,
            joint_velocities=self.joint_velocities,
        )
        out = dataclasses.replace(self, base_quaternion=q, link_transforms=H, link_velocities=V)
        out._model = model  # (not a field: lets tests/helpers.py::upcast refresh the caches in the new precision)
        return out


# =============================================================================================
# Contacts (src/jaxsim/rbda/collidable_points.py:9-65, rbda/contacts/common.py:25-63,
#           rbda/contacts/soft.py:195-444, api/contact.py:514-603)
# =============================================================================================


def collidable_points_pos_vel(model, *, link_transforms, link_velocities):
    kdp = model.kin_dyn_parameters
    idx = kdp.indices_of_enabled_collidable_points
    body = kdp.contact_body[idx]
    dtype = link_transforms.dtype
    L_p_C = kdp.contact_point[idx].astype(dtype)  # [nc,3]
    H = link_transforms[:, body]  # [N,nc,4,4]
    W_p_C = np.einsum("ncij,cj->nci", H[..., :3, :3], L_p_C) + H[..., :3, 3]
    V = link_velocities[:, body]  # [N,nc,6]
    # [I, -S(p)] @ v = v_lin + w x p   (collidable_points.py:50-53)
    CW_vl_WC = V[..., :3] + np.cross(V[..., 3:], W_p_C)
    return W_p_C, CW_vl_WC


def compute_penetration_data(model, p, v):
    terrain = model.terrain
    dtype = p.dtype
    n_hat = terrain.normal(p[..., 0], p[..., 1]).astype(dtype)
    h = np.zeros_like(p)
    h[..., 2] = terrain.height(p[..., 0], p[..., 1]).astype(dtype) - p[..., 2]
    delta = np.maximum(0.0, np.sum(h * n_hat, axis=-1)).astype(dtype)
    delta_dot = -np.sum(v * n_hat, axis=-1)
    delta_dot = np.where(delta > 0, delta_dot, 0.0).astype(dtype)
    return delta, delta_dot, n_hat


def hunt_crossley_contact_model(model, position, velocity, tangential_deformation, K, D, mu, p=0.5, q=0.5):
    """``SoftContacts.hunt_crossley_contact_model`` (soft.py:195-339), batched over points."""
    W_p_C, W_pd_C, m = position, velocity, tangential_deformation
    dtype = W_p_C.dtype
    K, D, mu, p, q = (dtype.type(x) for x in (K, D, mu, p, q))
    delta, delta_dot, n_hat = compute_penetration_data(model, W_p_C, W_pd_C)
    eps = np.finfo(dtype).eps  # dtype dependent by design (soft.py:246)
    dp = np.power(delta + eps, p)
    dq = np.power(delta + eps, q)

    force_normal_mag = (K * dp) * delta + (D * dq) * delta_dot
    force_normal_mag = np.maximum(dtype.type(0.0), force_normal_mag)
    f_normal = force_normal_mag[..., None] * n_hat

    dot = lambda a, b: np.sum(a * b, axis=-1, keepdims=True)  # noqa: E731
    v_tangential = W_pd_C - dot(W_pd_C, n_hat) * n_hat
    m_normal = dot(m, n_hat) * n_hat
    m_tangential = m - dot(m, n_hat) * n_hat
    f_tangential = -((K * dp)[..., None] * m_tangential + (D * dq)[..., None] * v_tangential)

    sticking = np.logical_or(delta <= 0, dot(f_tangential, f_tangential)[..., 0] <= (mu * force_normal_mag) ** 2)
    norm = rm.safe_norm(f_tangential)
    direction = f_tangential / (norm + eps * (norm == 0))[..., None]
    f_tangential = np.where(
        sticking[..., None], f_tangential, np.minimum(mu * force_normal_mag, norm)[..., None] * direction
    )
    f_tangential = np.where((delta <= 0)[..., None], 0.0, f_tangential).astype(dtype)

    md_no_contact = -(K / D) * m
    md_sticking = v_tangential - (K / D) * m_normal
    md_slipping = -(f_tangential + (K * dp)[..., None] * m_tangential) / (D * dq)[..., None]
    status = sticking.astype(int) + (delta <= 0).astype(int)  # 0 slipping, 1 sticking, 2 no contact
    md = np.where((status == 0)[..., None], md_slipping, np.where((status == 1)[..., None], md_sticking, md_no_contact))
    return (f_normal + f_tangential).astype(dtype), md.astype(dtype)


def compute_contact_forces(model, data: OracleData):
    """``SoftContacts.compute_contact_forces`` (soft.py:390-444): per-point inertial wrenches + m_dot."""
    kdp = model.kin_dyn_parameters
    idx = kdp.indices_of_enabled_collidable_points
    cp = model.contact_params
    W_p_C, W_pd_C = collidable_points_pos_vel(
        model, link_transforms=data.link_transforms, link_velocities=data.link_velocities
    )
    m = data.tangential_deformation
    CW_fl, md_enabled = hunt_crossley_contact_model(
        model, W_p_C, W_pd_C, m[:, idx], K=cp.K, D=cp.D, mu=cp.mu, p=cp.p, q=cp.q
    )
    # W_f = [f; p x f]   (soft.py:377-388)
    W_f = np.concatenate([CW_fl, np.cross(W_p_C, CW_fl)], axis=-1)
    md = np.zeros_like(m)
    md[:, idx] = md_enabled
    return W_f, md


def link_contact_forces(model, data: OracleData):
    """``link_contact_forces`` + ``link_forces_from_contact_forces`` (contact.py:514-603)."""
    kdp = model.kin_dyn_parameters
    W_f_C, md = compute_contact_forces(model, data)
    body = kdp.contact_body[kdp.indices_of_enabled_collidable_points]
    mask = (body[:, None] == np.arange(kdp.number_of_links())[None, :]).astype(W_f_C.dtype)  # [nc,nL]
    W_f_L = np.einsum("cl,ncj->nlj", mask, W_f_C)
    return W_f_L, md


# =============================================================================================
# Actuation (src/jaxsim/api/actuation_model.py:7-126)
# =============================================================================================


def tn_curve_fn(model, joint_velocities):
    ap = model.actuation_params
    dtype = joint_velocities.dtype
    tau_max, w_th, w_max = dtype.type(ap.torque_max), dtype.type(ap.omega_th), dtype.type(ap.omega_max)
    abs_vel = np.abs(joint_velocities)
    return np.where(
        abs_vel <= w_th,
        tau_max,
        np.where(abs_vel <= w_max, tau_max * (1 - (abs_vel - w_th) / (w_max - w_th)), 0.0),
    ).astype(dtype)


def compute_resultant_torques(model, data: OracleData, joint_force_references=None):
    kdp = model.kin_dyn_parameters
    s, sd = data.joint_positions, data.joint_velocities
    dtype = s.dtype
    tau_ref = joint_force_references if joint_force_references is not None else np.zeros_like(s)
    tau_pl = np.zeros_like(s)
    tau_fr = np.zeros_like(s)
    if kdp.number_of_joints() > 0:
        k_j = kdp.position_limit_spring.astype(dtype)
        d_j = kdp.position_limit_damper.astype(dtype)
        lower = np.minimum(s - kdp.position_limits_min.astype(dtype), 0.0)  # clip(max=0)
        upper = np.maximum(s - kdp.position_limits_max.astype(dtype), 0.0)  # clip(min=0)
        tau_pl = tau_pl - k_j * (lower + upper)
        # jnp.positive is unary plus (identity): tau_pl -= tau_pl * d_j * sd  (actuation_model.py:64-66)
        tau_pl = tau_pl - (+tau_pl) * (d_j * sd)
        if model.actuation_params.enable_friction:
            kc = kdp.friction_static.astype(dtype)
            kv = kdp.friction_viscous.astype(dtype)
            tau_fr = -(kc * np.sign(sd) + kv * sd)
    tau_total = tau_ref + tau_fr + tau_pl
    tau_lim = tn_curve_fn(model, sd)
    return np.clip(tau_total, -tau_lim, tau_lim).astype(dtype)


# =============================================================================================
# ODE + integrator + step (src/jaxsim/api/ode.py:16-131, api/integrators.py:14-88,
#                           api/model.py:2601-2681, 1269-1406)
# =============================================================================================


def forward_dynamics_aba(model, data: OracleData, *, joint_forces=None, link_forces=None):
    """``forward_dynamics_aba`` (model.py:1269-1406): link forces and the returned base
    acceleration are in ``data.velocity_representation``."""
    N, nL = data.batch_size, model.kin_dyn_parameters.number_of_links()
    dtype = data.dtype
    f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    W_f_L = other_representation_to_inertial(f_L, data.velocity_representation, data.link_transforms, is_force=True)
    W_v_WB = data.base_velocity(VelRepr.Inertial)
    W_vd_WB, sdd = aba(
        model,
        base_position=data.base_position,
        base_quaternion=data.base_orientation,
        joint_positions=data.joint_positions,
        base_linear_velocity=W_v_WB[:, :3],
        base_angular_velocity=W_v_WB[:, 3:],
        joint_velocities=data.joint_velocities,
        joint_forces=joint_forces,
        link_forces=W_f_L,
        standard_gravity=model.gravity,
    )
    rep = data.velocity_representation
    if rep == VelRepr.Inertial:
        W_H_C = np.broadcast_to(np.eye(4, dtype=dtype), (N, 4, 4))
        W_v_WC = np.zeros((N, 6), dtype=dtype)
    elif rep == VelRepr.Body:
        W_H_C, W_v_WC = data.base_transform, W_v_WB
    else:
        W_H_C = data.base_transform.copy()
        W_H_C[:, :3, :3] = np.eye(3)
        W_v_WC = np.zeros((N, 6), dtype=dtype)
        W_v_WC[:, :3] = data.base_velocity(VelRepr.Mixed)[:, :3]
    C_X_W = rm.adjoint_from_transform(W_H_C, inverse=True)
    C_vd_WB = rm.mv(C_X_W, W_vd_WB - rm.mv(rm.vx(W_v_WC), W_v_WB))
    if not model.floating_base():
        C_vd_WB = np.zeros((N, 6), dtype=dtype)
    return C_vd_WB.astype(dtype), sdd


def is_rigid_contact_model(model) -> bool:
    return type(getattr(model, "contact_model", None)).__name__ == "RigidContacts"


def is_relaxed_rigid_contact_model(model) -> bool:
    return type(getattr(model, "contact_model", None)).__name__ == "RelaxedRigidContacts"


def system_acceleration(model, data: OracleData, *, link_forces=None, joint_torques=None):
    """``system_acceleration`` (ode.py:16-131) evaluated in inertial representation, as the
    semi-implicit Euler integrator does (integrators.py:22)."""
    kdp = model.kin_dyn_parameters
    N, nL = data.batch_size, kdp.number_of_links()
    dtype = data.dtype
    f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    W_f_L_terrain = np.zeros_like(f_L)
    md = np.zeros_like(data.tangential_deformation)
    if kdp.number_of_collidable_points() > 0:  # ode.py:57
        if is_rigid_contact_model(model) or is_relaxed_rigid_contact_model(model):
            # contact.py:538-546: every model but SoftContacts sees the applied forces
            from . import refrelaxed, refrigid

            impl = refrigid if is_rigid_contact_model(model) else refrelaxed
            data_in = dataclasses.replace(data, velocity_representation=VelRepr.Inertial)  # integrators.py:22
            W_f_L_terrain, _ = impl.link_contact_forces(model, data_in, link_forces=f_L, joint_torques=joint_torques)
        else:
            W_f_L_terrain, md = link_contact_forces(model, data)
    W_f_L_total = f_L + W_f_L_terrain
    data_in = dataclasses.replace(data, velocity_representation=VelRepr.Inertial)
    W_vd_WB, sdd = forward_dynamics_aba(model, data_in, joint_forces=joint_torques, link_forces=W_f_L_total)
    return W_vd_WB, sdd, md


def system_acceleration_active(model, data: OracleData, *, link_forces=None, joint_torques=None):
    """``system_acceleration`` (ode.py:16-131) AS WRITTEN for data in any velocity representation: ``link_forces`` are
    in the representation of ``data`` (ode.py:30-33), the contact wrenches come back inertial (contact.py:527-530), the
    two are added as they are (ode.py:77) and the sum is handed to ABA through a references object built in the data's
    representation (ode.py:103-122) -- so with Body / Mixed data the inertial contact wrenches are READ as wrenches of that
    representation.  Identical to ``system_acceleration`` above for inertial data, which is how the integrators call it
    (integrators.py:22, ode.py:204).  Returns ``(vdot_WB in the data's representation, sddot, mdot)``."""
    kdp = model.kin_dyn_parameters
    N, nL = data.batch_size, kdp.number_of_links()
    dtype = data.dtype
    f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    W_f_L_terrain = np.zeros_like(f_L)
    md = np.zeros_like(data.tangential_deformation)
    if kdp.number_of_collidable_points() > 0:  # ode.py:57
        if is_rigid_contact_model(model) or is_relaxed_rigid_contact_model(model):
            from . import refrelaxed, refrigid

            impl = refrigid if is_rigid_contact_model(model) else refrelaxed
            W_f_L_terrain, _ = impl.link_contact_forces(model, data, link_forces=f_L, joint_torques=joint_torques)
        else:
            W_f_L_terrain, md = link_contact_forces(model, data)
    vd, sdd = forward_dynamics_aba(model, data, joint_forces=joint_torques, link_forces=f_L + W_f_L_terrain)
    return vd, sdd, md


def semi_implicit_euler_integration(model, data: OracleData, link_forces, joint_torques) -> OracleData:
    dtype = data.dtype
    W_vd_WB, sdd, md = system_acceleration(model, data, link_forces=link_forces, joint_torques=joint_torques)
    dt = dtype.type(model.time_step)
    new_acc = np.concatenate([W_vd_WB, sdd], axis=-1)
    new_vel = data.generalized_velocity(VelRepr.Inertial) + dt * new_acc
    W_v_B, sd = new_vel[:, :6], new_vel[:, 6:]
    W_w_WB = new_vel[:, 3:6]
    W_pd_B = new_vel[:, :3] + np.cross(W_w_WB, data.base_position)  # integrators.py:50
    W_Qd_B = rm.quaternion_derivative(data.base_orientation, W_w_WB, omega_in_body_fixed=False)
    W_p_B = data.base_position + dt * W_pd_B
    W_Q_B = data.base_orientation + dt * W_Qd_B
    nrm = rm.safe_norm(W_Q_B, keepdims=True)
    W_Q_B = W_Q_B / np.where(nrm == 0, 1.0, nrm).astype(dtype)
    s = data.joint_positions + dt * sd
    m = data.tangential_deformation + dt * md  # integrators.py:67-71
    new = dataclasses.replace(
        data,
        base_quaternion=W_Q_B.astype(dtype),
        base_position=W_p_B.astype(dtype),
        joint_positions=s.astype(dtype),
        joint_velocities=sd.astype(dtype),
        base_linear_velocity=W_v_B[:, :3].astype(dtype),
        base_angular_velocity=W_w_WB.astype(dtype),
        tangential_deformation=m.astype(dtype),
    )
    return new.update_caches(model)


def system_position_dynamics(data: OracleData, baumgarte_quaternion_regularization=1.0):
    """``system_position_dynamics`` (ode.py:134-171), inertial-fixed representation."""
    W_w_WB = data.base_angular_velocity
    W_pd_B = data.base_linear_velocity + np.cross(W_w_WB, data.base_position)
    W_Qd_B = rm.quaternion_derivative(
        data.base_orientation, W_w_WB, omega_in_body_fixed=False, K=baumgarte_quaternion_regularization
    )
    return W_pd_B, W_Qd_B, data.joint_velocities


def system_dynamics(model, data: OracleData, *, link_forces=None, joint_torques=None) -> dict:
    """``system_dynamics`` (ode.py:174-225): the state derivative as a dict keyed like the
    integrator's state (quaternion Baumgarte gain 1.0)."""
    W_vd_WB, sdd, md = system_acceleration(model, data, link_forces=link_forces, joint_torques=joint_torques)
    W_pd_B, W_Qd_B, sd = system_position_dynamics(data, 1.0)
    return dict(
        base_position=W_pd_B,
        base_quaternion=W_Qd_B,
        joint_positions=sd,
        base_linear_velocity=W_vd_WB[:, :3],
        base_angular_velocity=W_vd_WB[:, 3:],
        joint_velocities=sdd,
        tangential_deformation=md,
    )


def rk4_integration(model, data: OracleData, link_forces, joint_torques) -> OracleData:
    """``rk4_integration`` (integrators.py:91-167): classic RK4 on the dict state; every stage
    state goes through ``data.replace`` (quaternion normalisation + cache refresh), the external
    inertial link wrenches and the joint torques are those of the initial state."""
    dtype = data.dtype
    dt = dtype.type(model.time_step)

    def f(x):
        data_ti = dataclasses.replace(data, **x).update_caches(model)
        return system_dynamics(model, data_ti, link_forces=link_forces, joint_torques=joint_torques)

    nrm = rm.safe_norm(data.base_quaternion, keepdims=True)
    x_t0 = dict(
        base_position=data.base_position,
        base_quaternion=data.base_quaternion / np.where(nrm == 0, 1.0, nrm).astype(dtype),
        joint_positions=data.joint_positions,
        base_linear_velocity=data.base_linear_velocity,
        base_angular_velocity=data.base_angular_velocity,
        joint_velocities=data.joint_velocities,
        tangential_deformation=data.tangential_deformation,
    )

    def advance(x, dxdt, h):
        return {k: (x[k] + h * dxdt[k]).astype(dtype) for k in x}

    k1 = f(x_t0)
    k2 = f(advance(x_t0, k1, dtype.type(0.5) * dt))
    k3 = f(advance(x_t0, k2, dtype.type(0.5) * dt))
    k4 = f(advance(x_t0, k3, dt))
    dxdt = {k: (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) / 6 for k in x_t0}
    x_tf = advance(x_t0, dxdt, dt)
    return dataclasses.replace(data, **x_tf).update_caches(model)


def rk4fast_integration(model, data: OracleData, link_forces, joint_torques) -> OracleData:
    """``rk4fast_integration`` (integrators.py:170-276), as written:

    * the contact forces are computed once from the initial state and added to the external link
      forces as a constant inertial wrench (``:175-187``);
    * the position derivatives of every stage are those of the **initial** data --
      ``system_position_dynamics(data=data, ...)`` is called with ``data``, not the stage data
      (``:200-203``) -- so positions advance with the initial velocities, only the accelerations are
      re-evaluated (ABA at the stage state with the frozen wrenches);
    * with SoftContacts the derivative of the tangential deformation is stored *as the state*
      (``contact_state = update_contact_state(contact_state_derivative)``, ``:189-191,225``) and the
      state itself is returned as its derivative (``:214``): the deformation after the step is
      meaningless.  Without collidable points ``W_f_L_terrain`` is undefined (``:175-187``, NameError).
      Both cases are refused here; the function is restated for the contact models without contact
      state (RigidContacts, RelaxedRigidContacts), where it is well defined.

    One **deliberate deviation**: the reference calls ``link_contact_forces`` with the data in its own
    representation (``:175-187``, outside the Inertial switch) and hands it the *already inertial* external
    wrenches; its rigid / relaxed models rebuild their references with
    ``velocity_representation=data.velocity_representation``, so with Mixed or Body data and non-zero
    ``link_forces`` the inertial wrenches are re-read in that representation (converted twice).  Here the
    contact solve sees the external wrenches as what they are -- inertial -- whatever the representation of
    the data: the physics of the step does not depend on the representation the caller happens to use
    (``tests/test_oracle_rigid.py::test_rk4fast_link_forces_do_not_depend_on_the_data_representation``
    pins this; with Inertial data or without link forces the two readings coincide)."""
    kdp = model.kin_dyn_parameters
    if not (is_rigid_contact_model(model) or is_relaxed_rigid_contact_model(model)):
        raise NotImplementedError("rk4fast_integration corrupts the tangential deformation of SoftContacts in the reference")
    if kdp.number_of_collidable_points() == 0:
        raise NotImplementedError("rk4fast_integration fails in the reference without collidable points (W_f_L_terrain undefined)")
    from . import refrelaxed, refrigid

    dtype = data.dtype
    dt = dtype.type(model.time_step)
    impl = refrigid if is_rigid_contact_model(model) else refrelaxed
    data_in = dataclasses.replace(data, velocity_representation=VelRepr.Inertial)
    W_f_L_terrain, _ = impl.link_contact_forces(model, data_in, link_forces=link_forces, joint_torques=joint_torques)
    W_f_L_total = link_forces + W_f_L_terrain
    W_pd_B, W_Qd_B, sd = system_position_dynamics(data, 1.0)  # of the initial data, at every stage

    def f(x):
        data_ti = dataclasses.replace(data, velocity_representation=VelRepr.Inertial, **x).update_caches(model)
        W_vd_WB, sdd = forward_dynamics_aba(model, data_ti, joint_forces=joint_torques, link_forces=W_f_L_total)
        return dict(base_position=W_pd_B, base_quaternion=W_Qd_B, joint_positions=sd, base_linear_velocity=W_vd_WB[:, :3],
                    base_angular_velocity=W_vd_WB[:, 3:], joint_velocities=sdd)  # fmt: skip

    nrm = rm.safe_norm(data.base_quaternion, keepdims=True)
    x_t0 = dict(
        base_position=data.base_position,
        base_quaternion=data.base_quaternion / np.where(nrm == 0, 1.0, nrm).astype(dtype),
        joint_positions=data.joint_positions,
        base_linear_velocity=data.base_linear_velocity,
        base_angular_velocity=data.base_angular_velocity,
        joint_velocities=data.joint_velocities,
    )

    def advance(x, dxdt, h):
        return {k: (x[k] + h * dxdt[k]).astype(dtype) for k in x}

    k1 = f(x_t0)
    k2 = f(advance(x_t0, k1, dtype.type(0.5) * dt))
    k3 = f(advance(x_t0, k2, dtype.type(0.5) * dt))
    k4 = f(advance(x_t0, k3, dt))
    dxdt = {k: (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) / 6 for k in x_t0}
    x_tf = advance(x_t0, dxdt, dt)
    return dataclasses.replace(data, **x_tf).update_caches(model)


_INTEGRATORS = {0: semi_implicit_euler_integration, 1: rk4_integration, 2: rk4fast_integration}  # integrators.py:279-283


def step(model, data: OracleData, *, link_forces=None, joint_force_references=None) -> OracleData:
    """``js.model.step`` (model.py:2601-2681) for SoftContacts; the integrator is
    ``model.integrator`` (SemiImplicitEuler or RungeKutta4)."""
    N, nL = data.batch_size, model.kin_dyn_parameters.number_of_links()
    dtype = data.dtype
    O_f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    W_f_L = other_representation_to_inertial(
        np.asarray(O_f_L, dtype=dtype), data.velocity_representation, data.link_transforms, is_force=True
    )
    tau_ref = (
        np.asarray(joint_force_references, dtype=dtype)
        if joint_force_references is not None
        else np.zeros_like(data.joint_positions)
    )
    tau_total = compute_resultant_torques(model, data, joint_force_references=tau_ref)
    data_tf = _INTEGRATORS[int(getattr(model, "integrator", 0))](model, data, W_f_L, tau_total)
    if is_rigid_contact_model(model):  # model.py:2677-2679
        from . import refrigid

        data_tf = refrigid.update_velocity_after_impact(model, data_tf)
    return data_tf


# =============================================================================================
# Inverse dynamics wrappers (src/jaxsim/api/model.py:1746-1978)
# =============================================================================================


def inverse_dynamics(model, data: OracleData, *, joint_accelerations=None, base_acceleration=None, link_forces=None):
    N, nL = data.batch_size, model.kin_dyn_parameters.number_of_links()
    dtype = data.dtype
    sdd = joint_accelerations if joint_accelerations is not None else np.zeros_like(data.joint_positions)
    vd_WB = base_acceleration if base_acceleration is not None else np.zeros((N, 6), dtype=dtype)
    f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    rep = data.velocity_representation
    W_v_WB = data.base_velocity(VelRepr.Inertial)
    if rep == VelRepr.Inertial:
        W_H_C = np.broadcast_to(np.eye(4, dtype=dtype), (N, 4, 4))
        W_v_WC = np.zeros((N, 6), dtype=dtype)
    elif rep == VelRepr.Body:
        W_H_C, W_v_WC = data.base_transform, W_v_WB
    else:
        W_H_C = data.base_transform.copy()
        W_H_C[:, :3, :3] = np.eye(3)
        W_v_WC = np.zeros((N, 6), dtype=dtype)
        W_v_WC[:, :3] = data.base_velocity(VelRepr.Mixed)[:, :3]
    W_X_C = rm.adjoint_from_transform(W_H_C)
    C_X_W = rm.adjoint_from_transform(W_H_C, inverse=True)
    C_v_WC = rm.mv(C_X_W, W_v_WC)
    W_vd_WB = rm.mv(W_X_C, vd_WB + rm.mv(rm.vx(C_v_WC), data.base_velocity(rep)))  # model.py:1801-1842
    W_f_L = other_representation_to_inertial(f_L, rep, data.link_transforms, is_force=True)
    W_f_B, tau = rnea(
        model,
        base_position=data.base_position,
        base_quaternion=data.base_quaternion,  # raw quaternion (model.py:1856)
        joint_positions=data.joint_positions,
        base_linear_velocity=W_v_WB[:, :3],
        base_angular_velocity=W_v_WB[:, 3:],
        joint_velocities=data.joint_velocities,
        base_linear_acceleration=W_vd_WB[:, :3],
        base_angular_acceleration=W_vd_WB[:, 3:],
        joint_accelerations=sdd,
        link_forces=W_f_L,
        standard_gravity=model.gravity,
    )
    f_B = inertial_to_other_representation(W_f_B, rep, data.base_transform, is_force=True)
    return f_B.astype(dtype), tau.astype(dtype)


def free_floating_gravity_forces(model, data: OracleData):
    """``g(q)`` (model.py:1897-1931): RNEA at zero velocity/acceleration/forces."""
    z = dataclasses.replace(
        data,
        base_linear_velocity=np.zeros_like(data.base_linear_velocity),
        base_angular_velocity=np.zeros_like(data.base_angular_velocity),
        joint_velocities=np.zeros_like(data.joint_velocities),
    ).update_caches(model)
    return np.concatenate(inverse_dynamics(model, z), axis=-1)


def free_floating_bias_forces(model, data: OracleData):
    """``h(q, nu)`` (model.py:1934-1978); fixed-base models drop the base velocity."""
    d = data
    if not model.floating_base():
        d = dataclasses.replace(
            data,
            base_linear_velocity=np.zeros_like(data.base_linear_velocity),
            base_angular_velocity=np.zeros_like(data.base_angular_velocity),
        ).update_caches(model)
    return np.concatenate(inverse_dynamics(model, d), axis=-1)


def free_floating_mass_matrix(model, data: OracleData):
    """``free_floating_mass_matrix`` (``model.py:1529-1590``): CRBA in body-fixed representation, moved to the
    representation of ``data`` by ``_transform_M_block`` with ``B_X_W`` (Inertial) or ``B_X_BW`` (Mixed)."""
    M_body = crba(model, joint_positions=data.joint_positions)
    rep = data.velocity_representation
    if rep == VelRepr.Body:
        return M_body
    H = data.base_transform.copy()
    if rep == VelRepr.Mixed:
        H[:, :3, 3] = 0.0  # BW_H_B
    X = rm.adjoint_from_transform(H, inverse=True)
    Xt = np.swapaxes(X, -1, -2)
    M = M_body.copy()
    M[:, :6, :6] = Xt @ M_body[:, :6, :6] @ X
    M[:, :6, 6:] = Xt @ M_body[:, :6, 6:]
    M[:, 6:, :6] = M_body[:, 6:, :6] @ X
    return M


# =============================================================================================
# Contact-parameter estimator (rbda/contacts/common.py:88-168, api/contact.py:160-211)
# =============================================================================================


def build_default_contact_params(
    model,
    *,
    stiffness=None,
    damping=None,
    standard_gravity=rm.STANDARD_GRAVITY,
    static_friction_coefficient=0.5,
    max_penetration=0.001,
    number_of_active_collidable_points_steady_state=1,
    damping_ratio=1.0,
    p=0.5,
    q=0.5,
):
    m = float(np.sum(model.kin_dyn_parameters.link_mass))
    if stiffness is None:
        f_average = m * standard_gravity / number_of_active_collidable_points_steady_state
        stiffness = float(np.clip(f_average / np.power(max_penetration, 1 + p), 0, 1e6))
    critical_damping = 2 * np.sqrt(stiffness * m)
    if damping is None:
        damping = float(np.clip(damping_ratio * critical_damping, 0, 1e4))
    return dict(K=float(stiffness), D=float(damping), mu=float(static_friction_coefficient), p=float(p), q=float(q))


def com_position(model, data: OracleData):
    """World CoM position (``src/jaxsim/api/com.py:13-60``)."""
    kdp = model.kin_dyn_parameters
    H = data.link_transforms
    c = np.einsum("nlij,lj->nli", H[..., :3, :3], kdp.link_com) + H[..., :3, 3]
    return np.einsum("l,nli->ni", kdp.link_mass, c) / np.sum(kdp.link_mass)


def estimate_good_contact_parameters(
    model,
    *,
    standard_gravity=rm.STANDARD_GRAVITY,
    static_friction_coefficient=0.5,
    number_of_active_collidable_points_steady_state=1,
    damping_ratio=1.0,
    max_penetration=None,
):
    if max_penetration is None:
        zero = OracleData.build(model)
        W_pz_CoM = com_position(model, zero)[0, 2]
        if model.floating_base():
            W_p_C, _ = collidable_points_pos_vel(
                model, link_transforms=zero.link_transforms, link_velocities=zero.link_velocities
            )
            W_pz_CoM = W_pz_CoM - W_p_C[0, :, 2].min()
        max_penetration = 0.01 * W_pz_CoM
    return build_default_contact_params(
        model,
        standard_gravity=standard_gravity,
        static_friction_coefficient=static_friction_coefficient,
        max_penetration=max_penetration,
        number_of_active_collidable_points_steady_state=number_of_active_collidable_points_steady_state,
        damping_ratio=damping_ratio,
    )


# =============================================================================================
# Random states (src/jaxsim/api/data.py:552-682; distribution only -- JAX's threefry stream is
# not reproduced)
# =============================================================================================


def random_model_data(
    model,
    *,
    batch_size=1,
    seed=0,
    dtype=np.float64,
    velocity_representation=VelRepr.Mixed,
    base_pos_bounds=((-1, -1, 0.5), (1, 1, 1)),
    base_rpy_bounds=((-np.pi,) * 3, (np.pi,) * 3),
    base_vel_lin_bounds=((-1,) * 3, (1,) * 3),
    base_vel_ang_bounds=((-1,) * 3, (1,) * 3),
    joint_vel_bounds=(-1.0, 1.0),
    joint_pos_clip=10.0,
) -> OracleData:
    kdp = model.kin_dyn_parameters
    rng = np.random.default_rng(seed)
    N, n = batch_size, kdp.number_of_joints()
    p = rng.uniform(*np.array(base_pos_bounds, dtype=float), size=(N, 3))
    rpy = rng.uniform(*np.array(base_rpy_bounds, dtype=float), size=(N, 3))
    q = rm.quaternion_from_euler_xyz(rpy)
    lo = np.maximum(kdp.position_limits_min, -joint_pos_clip)
    hi = np.minimum(kdp.position_limits_max, joint_pos_clip)
    s = rng.uniform(lo, hi, size=(N, n)) if n > 0 else np.zeros((N, 0))
    sd = rng.uniform(*joint_vel_bounds, size=(N, n)) if n > 0 else np.zeros((N, 0))
    vl = rng.uniform(*np.array(base_vel_lin_bounds, dtype=float), size=(N, 3))
    va = rng.uniform(*np.array(base_vel_ang_bounds, dtype=float), size=(N, 3))
    if not model.floating_base():
        vl, va = np.zeros_like(vl), np.zeros_like(va)
    return OracleData.build(
        model,
        base_position=p,
        base_quaternion=q,
        joint_positions=s,
        base_linear_velocity=vl,
        base_angular_velocity=va,
        joint_velocities=sd,
        velocity_representation=velocity_representation,
        dtype=dtype,
    )
