"""``jaxsim.api.contact`` mirror: parameter estimation (host, build-time), the read-only queries on the cached
kinematics, and -- [round 6] -- ``link_contact_forces`` (one launch of ``jxs_system_dynamics``)."""

from __future__ import annotations

import numpy as np

from .. import _hostmath as hm
from ..model import STANDARD_GRAVITY, JaxSimModel, SoftContactsParams


def _zero_pose_link_transforms(model: JaxSimModel) -> np.ndarray:
    """World link transforms at the zero configuration (host NumPy, build-time only)."""
    kdp = model.kin_dyn_parameters
    nL = kdp.number_of_links()
    H = np.zeros((nL, 4, 4))
    H[0] = kdp.suc_H_i[0]
    for i in range(1, nL):
        H[i] = H[kdp.parent_array[i]] @ kdp.lambda_H_pre[i] @ kdp.suc_H_i[i]
    return H


def estimate_good_contact_parameters(
    model: JaxSimModel,
    *,
    standard_gravity: float = STANDARD_GRAVITY,
    static_friction_coefficient: float = 0.5,
    number_of_active_collidable_points_steady_state: int = 1,
    damping_ratio: float = 1.0,
    max_penetration: float | None = None,
) -> SoftContactsParams:
    """``estimate_good_contact_parameters`` (``src/jaxsim/api/contact.py:160-211``): when
    ``max_penetration`` is not given it is 1 % of the zero-pose CoM height above the lowest
    collidable point (floating base) or above the world origin (fixed base)."""
    kdp = model.kin_dyn_parameters
    if max_penetration is None:
        H = _zero_pose_link_transforms(model)
        com = np.einsum("lij,lj->li", H[:, :3, :3], kdp.link_com) + H[:, :3, 3]
        z_com = float(np.sum(kdp.link_mass * com[:, 2]) / np.sum(kdp.link_mass))
        if model.floating_base() and kdp.number_of_collidable_points() > 0:
            idx = kdp.indices_of_enabled_collidable_points
            body = kdp.contact_body[idx]
            pz = np.einsum("cij,cj->ci", H[body][:, :3, :3], kdp.contact_point[idx])[:, 2] + H[body][:, 2, 3]
            z_com -= float(pz.min())
        max_penetration = 0.01 * z_com
    soft = SoftContactsParams.build_default_from_jaxsim_model(
        model,
        standard_gravity=standard_gravity,
        static_friction_coefficient=static_friction_coefficient,
        max_penetration=max_penetration,
        number_of_active_collidable_points_steady_state=number_of_active_collidable_points_steady_state,
        damping_ratio=damping_ratio,
    )
    # the reference builds the parameter class of the active contact model from the same estimate
    # (`model.contact_model._parameters_class().build_default_from_jaxsim_model`, contact.py:203-211,
    # rbda/contacts/common.py:88-168): stiffness, damping and friction go to K, D, mu
    cls = getattr(model.contact_model, "_parameters_class", SoftContactsParams)
    if cls is SoftContactsParams:
        return soft
    return cls.build(K=soft.K, D=soft.D, mu=soft.mu)


# ---- queries on the cached kinematics (src/jaxsim/api/contact.py:18-145,214-350) ------------------
# The kernels of the step keep the contact phase on the device; these are the reference's read-only
# helpers, evaluated on the host from the cached link transforms / velocities of the data object
# (``jxs_refresh_kinematics``).  Enabled collidable points only, like the reference.


def _enabled(model: JaxSimModel):
    kdp = model.kin_dyn_parameters
    idx = kdp.indices_of_enabled_collidable_points
    return kdp.contact_body[idx], kdp.contact_point[idx]


def collidable_point_kinematics(model: JaxSimModel, data):
    """Position and velocity of the enabled collidable points in the world frame
    (``api/contact.py:18-45``, ``rbda/collidable_points.py:9-65``): ``p = W_H_L L_p``,
    ``pdot = v_lin + w x p`` of the inertial-fixed link velocity."""
    body, L_p = _enabled(model)
    H = np.asarray(data._kinematics()[0], dtype=np.float64)[:, body]  # [N, n_cp, 4, 4]
    V = np.asarray(data._kinematics()[1], dtype=np.float64)[:, body]
    p = np.einsum("ncij,cj->nci", H[..., :3, :3], L_p) + H[..., :3, 3]
    pd = V[..., :3] + np.cross(V[..., 3:], p)
    return data._out(p.astype(data.dtype)), data._out(pd.astype(data.dtype))


def collidable_point_positions(model: JaxSimModel, data):
    return collidable_point_kinematics(model, data)[0]


def collidable_point_velocities(model: JaxSimModel, data):
    return collidable_point_kinematics(model, data)[1]


def in_contact(model: JaxSimModel, data, *, link_names=None):
    """Boolean per link: some enabled collidable point of the link is at or below the terrain
    (``api/contact.py:92-145``)."""
    names = model.link_names()
    if link_names is not None and set(link_names).difference(names):
        raise ValueError("One or more link names are not part of the model")
    body, _ = _enabled(model)
    p, _ = collidable_point_kinematics(model, data)
    p = np.asarray(p, dtype=np.float64).reshape(-1, len(body), 3)
    below = p[..., 2] <= model.terrain.height(p[..., 0], p[..., 1])
    idxs = [names.index(n) for n in link_names] if link_names is not None else list(range(model.number_of_links()))
    out = np.stack([(below & (body == i)[None, :]).any(axis=1) for i in idxs], axis=1)
    return data._out(out)


def transforms(model: JaxSimModel, data):
    """``W_H_C`` of the implicit frames ``C = (W_p_C, [L])`` of the enabled points (``api/contact.py:214-257``)."""
    body, L_p = _enabled(model)
    H = np.asarray(data._kinematics()[0], dtype=np.float64)[:, body].copy()
    H[..., :3, 3] += np.einsum("ncij,cj->nci", H[..., :3, :3], L_p)
    return data._out(H.astype(data.dtype))


def jacobian(model: JaxSimModel, data, *, output_vel_repr=None):
    """Free-floating Jacobians of the frames of the enabled collidable points, ``[n_cp, 6, 6+n]``
    (``api/contact.py:260-350``): the Jacobian of the parent link in inertial-fixed output representation,
    re-expressed in the frame ``C`` (Body) or ``C[W]`` (Mixed)."""
    from ..data import _inertial_to_other
    from ..model import VelRepr
    from . import model as _m

    out_rep = data.velocity_representation if output_vel_repr is None else VelRepr(output_vel_repr)
    body, _ = _enabled(model)
    W_J = np.asarray(_m.generalized_free_floating_jacobian(model, data, output_vel_repr=VelRepr.Inertial), np.float64)
    W_J = W_J.reshape((-1,) + W_J.shape[-3:])[:, body]  # [N, n_cp, 6, nv]
    if out_rep == VelRepr.Inertial:
        return data._out(W_J.astype(data.dtype))
    W_H_C = np.asarray(transforms(model, data), np.float64).reshape((-1, len(body), 4, 4))
    nv = W_J.shape[-1]
    cols = np.moveaxis(W_J, -1, 2)  # [N, n_cp, nv, 6]
    H = np.broadcast_to(W_H_C[:, :, None], cols.shape[:3] + (4, 4))
    O = _inertial_to_other(cols.reshape(-1, 6), out_rep, H.reshape(-1, 4, 4), False).reshape(cols.shape)
    return data._out(np.moveaxis(O, 2, -1).astype(data.dtype).reshape(W_J.shape[:2] + (6, nv)))


def jacobian_derivative(model: JaxSimModel, data, *, output_vel_repr=None):
    """Derivative of the free-floating Jacobians of the enabled collidable points, ``[n_cp, 6, 6+n]``
    (``src/jaxsim/api/contact.py:353-511``): ``O_Jdot_WC_I = O_Xdot_W W_J T + O_X_W W_Jdot T + O_X_W W_J Tdot`` with
    the inertial / inertial link Jacobians and their derivatives (one launch of the Jacobian kernel each)."""
    from ..model import VelRepr
    from . import model as _m

    out_rep = data.velocity_representation if output_vel_repr is None else VelRepr(output_vel_repr)
    body, L_p = _enabled(model)
    N, n = data.batch_size, model.dofs()
    W_H_B = data._base_transform_batched()
    rep = data.velocity_representation
    # ---- input representation (:401-436)
    if rep == VelRepr.Inertial:
        X, Xd = np.broadcast_to(np.eye(6), (N, 6, 6)), np.zeros((N, 6, 6))
    elif rep == VelRepr.Body:
        X = _m._adjoint(W_H_B)
        Xd = X @ _m._vx_matrix(np.asarray(data._base_velocity_batched(VelRepr.Body), np.float64))
    else:
        W_H_BW = W_H_B.copy()
        W_H_BW[:, :3, :3] = np.eye(3)
        X = _m._adjoint(W_H_BW)
        v = np.asarray(data._base_velocity_batched(VelRepr.Mixed), np.float64).copy()
        v[:, 3:] = 0.0
        Xd = X @ _m._vx_matrix(v)
    T, Td = _m._block_T(X, n)[:, None], _m._block_T(Xd, n, identity=False)[:, None]
    # ---- link Jacobians and derivatives, inertial in / inertial out (:438-449)
    with data.switch_velocity_representation(VelRepr.Inertial):
        W_J = np.asarray(_m.generalized_free_floating_jacobian(model, data), np.float64)
        W_Jd = np.asarray(_m.generalized_free_floating_jacobian_derivative(model, data), np.float64)
    W_J = W_J.reshape((N,) + W_J.shape[-3:])[:, body]
    W_Jd = W_Jd.reshape((N,) + W_Jd.shape[-3:])[:, body]
    # ---- output representation (:455-490)
    nc = len(body)
    if out_rep == VelRepr.Inertial:
        O_X_W, O_Xd_W = np.broadcast_to(np.eye(6), (N, nc, 6, 6)), np.zeros((N, nc, 6, 6))
    else:
        W_H_C = np.asarray(transforms(model, data), np.float64).reshape(N, nc, 4, 4)
        W_v_WL = np.asarray(data._kinematics()[1], np.float64).reshape(N, -1, 6)[:, body]
        if out_rep == VelRepr.Body:
            O_X_W = _m._adjoint(W_H_C, inverse=True)
            O_Xd_W = -O_X_W @ _m._vx_matrix(W_v_WL)
        else:
            W_H_CW = W_H_C.copy()
            W_H_CW[..., :3, :3] = np.eye(3)
            O_X_W = _m._adjoint(W_H_CW, inverse=True)
            CW_v = np.einsum("ncij,ncj->nci", O_X_W, W_v_WL)
            W_v_W_CW = np.zeros_like(CW_v)
            W_v_W_CW[..., :3] = CW_v[..., :3]
            O_Xd_W = -O_X_W @ _m._vx_matrix(W_v_W_CW)
    Jd = O_Xd_W @ W_J @ T + O_X_W @ W_Jd @ T + O_X_W @ W_J @ Td
    return data._out(Jd.astype(data.dtype))


# ---- contact forces (src/jaxsim/api/contact.py:514-603) ----------------------------------------------------------


def link_contact_forces(model: JaxSimModel, data, *, link_forces=None, joint_torques=None):
    """``link_contact_forces`` (``src/jaxsim/api/contact.py:514-555``): the ``[nL, 6]`` contact wrenches of the links in
    inertial representation and the contact model's auxiliary dictionary -- ``{"m_dot": ...}`` (rate of the tangential
    deformation of every collidable point, zero for disabled ones) for SoftContacts, ``{}`` for RigidContacts and
    RelaxedRigidContacts, whose forces depend on ``link_forces`` (read in the representation of ``data``, like
    ``rbda/contacts/rigid.py:268-276``) and ``joint_torques``.  One launch; SoftContacts stops before ABA."""
    from ..state import StateLayout, unpack_state
    from . import ode as _ode

    N, nL = data.batch_size, model.number_of_links()
    soft = _ode._is_soft(model)
    n_cp = model.kin_dyn_parameters.number_of_collidable_points()
    xdot, W_f = _ode.system_dynamics_device(
        model, data, link_forces=None if soft else link_forces, joint_torques=None if soft else joint_torques,
        force_repr=data.velocity_representation, want_derivative=soft and n_cp > 0, want_link_contact_forces=True,
    )  # fmt: skip
    W_f_L = W_f.to_host().T.reshape(N, nL, 6)
    aux = {}
    if soft:
        md = unpack_state(StateLayout.of(model), xdot.to_host())["tangential_deformation"] if n_cp > 0 else np.zeros((N, 0, 3), dtype=data.dtype)
        aux = {"m_dot": data._out(md)}
    return data._out(W_f_L), aux


def link_forces_from_contact_forces(model: JaxSimModel, *, contact_forces):
    """``link_forces_from_contact_forces`` (``src/jaxsim/api/contact.py:558-603``): the wrenches ``[n_cp, 6]`` of the
    enabled collidable points (world coordinates) summed per parent link, ``[nL, 6]``; batched input ``[N, n_cp, 6]``
    gives ``[N, nL, 6]``.  (Host NumPy: the kernels do this sum in registers, ``jxs_core.h link_wrench_sums``; this is the
    reference's stand-alone helper for forces that come from somewhere else.)"""
    body, _ = _enabled(model)
    W_f_C = np.asarray(contact_forces, dtype=float)
    W_f_C = W_f_C.reshape((-1, 6)) if W_f_C.ndim <= 2 else W_f_C
    if W_f_C.shape[-2] != len(body):
        raise ValueError((W_f_C.shape, (len(body), 6)))
    mask = (body[:, None] == np.arange(model.number_of_links())[None, :]).astype(W_f_C.dtype)
    return np.einsum("cl,...cj->...lj", mask, W_f_C)
