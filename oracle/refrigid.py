"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's ``RigidContacts`` model
(BASELINE.json config 5).

Follows, function by function:

* ``src/jaxsim/rbda/jacobian.py:128-339`` (full doubly-left Jacobian and its derivative),
* ``src/jaxsim/api/model.py:925-1228`` (``generalized_free_floating_jacobian(_derivative)``),
* ``src/jaxsim/api/contact.py:214-511`` (``transforms``, ``jacobian``, ``jacobian_derivative``),
* ``src/jaxsim/api/model.py:1529-1631`` (mass matrix / inverse in the active representation),
* ``src/jaxsim/rbda/contacts/rigid.py:176-539`` (contact forces through a QP, impact velocity).

Third-party arithmetic that is NOT under ``/root/reference``:

* ``qpax`` (unpinned in ``pyproject.toml:47-59``): ``qpax.solve_qp`` is a primal-dual interior
  point method with Mehrotra predictor-corrector steps after Mattingley & Boyd, "CVXGEN: a code
  generator for embedded convex optimization" (2012), section 5.  ``solve_qp_pdip`` below restates
  that published algorithm (initialisation 5.2, iteration 5.3, 0.99 step-to-boundary).  The
  package itself is absent, so the iterates are **parity unpinned**; the QP is strictly convex
  (``Q = G + 1e-6 I``), hence the point both converge to is unique, and the reference pins it only
  through ``tests/test_simulations.py:245-292`` (box settles at z = 0.05), re-expressed in
  ``tests/test_oracle_known_answers.py``.
* ``jnp.linalg.lstsq`` (minimum-norm least squares through an SVD): restated with
  ``numpy.linalg.lstsq`` (same LAPACK driver family, same ``rcond = eps * max(M, N)``).

``mass_inverse`` (``src/jaxsim/rbda/mass_inverse.py:11-233``) is an O(n^2) articulated-body
recursion whose result is, by its own contract, ``inv(M)``; it is restated as the dense inverse of
the CRBA matrix, not re-derived.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from . import refmath as rm
from . import refstep as rs
from .refstep import VelRepr


# =============================================================================================
# Jacobians (rbda/jacobian.py:128-339)
# =============================================================================================


def jacobian_full_doubly_left(model, joint_positions):
    """``B_J_full_WX_B`` [N,6,6+n] and ``B_H_L`` [N,nL,4,4] (jacobian.py:128-222)."""
    kdp = model.kin_dyn_parameters
    s = joint_positions
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    lam = kdp.parent_array
    S = kdp.motion_subspaces.astype(dtype)
    eye4 = np.broadcast_to(np.eye(4, dtype=dtype), (N, 4, 4))
    i_X_lam = rs.joint_transforms(model, s, eye4)
    B_X_i = np.zeros((N, nL, 6, 6), dtype=dtype)
    B_X_i[:, 0] = np.eye(6)
    J = np.zeros((N, 6, 6 + nL - 1), dtype=dtype)
    J[:, :6, :6] = np.eye(6)
    for i in range(1, nL):
        B_X_i[:, i] = B_X_i[:, lam[i]] @ rm.adjoint_inverse(i_X_lam[:, i])
        J[:, :, 6 + i - 1] = rm.mv(B_X_i[:, i], S[i])
    return J, rm.adjoint_to_transform(B_X_i)


def jacobian_derivative_full_doubly_left(model, joint_positions, joint_velocities):
    """``B_Jdot_full_WX_B`` [N,6,6+n] (jacobian.py:225-339)."""
    kdp = model.kin_dyn_parameters
    s, sd = joint_positions, joint_velocities
    dtype = s.dtype
    N, nL = s.shape[0], kdp.number_of_links()
    lam = kdp.parent_array
    S = kdp.motion_subspaces.astype(dtype)
    eye4 = np.broadcast_to(np.eye(4, dtype=dtype), (N, 4, 4))
    i_X_lam = rs.joint_transforms(model, s, eye4)
    B_X_i = np.zeros((N, nL, 6, 6), dtype=dtype)
    B_X_i[:, 0] = np.eye(6)
    B_Xd_i = np.zeros((N, nL, 6, 6), dtype=dtype)
    B_v_Bi = np.zeros((N, nL, 6), dtype=dtype)
    Jd = np.zeros((N, 6, 6 + nL - 1), dtype=dtype)
    for i in range(1, nL):
        ii = i - 1
        B_X_i[:, i] = B_X_i[:, lam[i]] @ rm.adjoint_inverse(i_X_lam[:, i])
        B_v_Bi[:, i] = B_v_Bi[:, lam[i]] + rm.mv(B_X_i[:, i], S[i]) * sd[:, ii, None]
        i_X_B = rm.adjoint_inverse(B_X_i[:, i])
        B_Xd_i[:, i] = B_X_i[:, i] @ rm.vx(rm.mv(i_X_B, B_v_Bi[:, i]))  # A_Xd_B = A_X_B vx(B_v_AB)
        Jd[:, :, 6 + ii] = rm.mv(B_Xd_i[:, i], S[i])
    return Jd


def _support_mask(model, dtype):
    """``hstack([ones(5), kappa_bool])`` per link (model.py:975-981): [nL, 6+n] column mask."""
    kdp = model.kin_dyn_parameters
    nL = kdp.number_of_links()
    lam = kdp.parent_array
    mask = np.zeros((nL, 6 + nL - 1), dtype=dtype)
    mask[:, :6] = 1
    for L in range(nL):
        j = L
        while j > 0:  # joint j moves link j and is on the path base -> L
            mask[L, 6 + j - 1] = 1
            j = lam[j]
    return mask


def _block_diag_T(X, n):
    """``block_diag(X, I_n)`` batched."""
    N = X.shape[0]
    T = np.zeros((N, 6 + n, 6 + n), dtype=X.dtype)
    T[:, :6, :6] = X
    T[:, 6:, 6:] = np.eye(n, dtype=X.dtype)
    return T


def _block_diag_Td(Xd, n):
    N = Xd.shape[0]
    T = np.zeros((N, 6 + n, 6 + n), dtype=Xd.dtype)
    T[:, :6, :6] = Xd
    return T


def _mixed_base_frame(data):
    """``BW_H_B``-style transform with the rotation of the base and no translation... and the
    other way round: returns (W_H_BW: translation only, BW_H_B: rotation only)."""
    W_H_B = data.base_transform
    W_H_BW = W_H_B.copy()
    W_H_BW[:, :3, :3] = np.eye(3)
    BW_H_B = W_H_B.copy()
    BW_H_B[:, :3, 3] = 0
    return W_H_BW, BW_H_B


def generalized_free_floating_jacobian(model, data: rs.OracleData, input_repr, output_repr):
    """``generalized_free_floating_jacobian`` (model.py:925-1045): the generalized velocity in ``input_repr``,
    the link velocities in ``output_repr``: ``O_J_WL_I`` [N,nL,6,6+n]."""
    n = model.kin_dyn_parameters.number_of_joints()
    dtype = data.dtype
    B_J_full, B_H_L = jacobian_full_doubly_left(model, data.joint_positions)
    W_H_B = data.base_transform
    if input_repr == VelRepr.Inertial:
        B_X_I = rm.adjoint_from_transform(W_H_B, inverse=True)
    elif input_repr == VelRepr.Body:
        B_X_I = np.broadcast_to(np.eye(6, dtype=dtype), (data.batch_size, 6, 6))
    else:
        _, BW_H_B = _mixed_base_frame(data)  # model.py:964-970: rotation only
        B_X_I = rm.adjoint_from_transform(BW_H_B, inverse=True)
    B_J_full_I = B_J_full @ _block_diag_T(B_X_I, n)
    B_J_WL_I = _support_mask(model, dtype)[None, :, None, :] * B_J_full_I[:, None]
    if output_repr == VelRepr.Inertial:  # :999-1006
        return rm.adjoint_from_transform(W_H_B)[:, None] @ B_J_WL_I
    if output_repr == VelRepr.Body:  # :1008-1015
        return rm.adjoint_from_transform(B_H_L, inverse=True) @ B_J_WL_I
    LW_H_L = W_H_B[:, None] @ B_H_L  # :1017-1037
    LW_H_L[..., :3, 3] = 0
    LW_H_B = LW_H_L @ rm.transform_inverse(B_H_L)
    return rm.adjoint_from_transform(LW_H_B) @ B_J_WL_I


def generalized_free_floating_jacobian_inertial_output(model, data: rs.OracleData, input_repr):
    """``generalized_free_floating_jacobian(..., output_vel_repr=Inertial)``: ``W_J_WL_I`` [N,nL,6,6+n]."""
    return generalized_free_floating_jacobian(model, data, input_repr, VelRepr.Inertial)


def generalized_free_floating_jacobian_derivative_inertial(model, data: rs.OracleData):
    """``generalized_free_floating_jacobian_derivative`` with inertial-fixed input and output
    representation (model.py:1048-1228, the Inertial/Inertial branches): ``W_Jdot_WL_W``."""
    n = model.kin_dyn_parameters.number_of_joints()
    dtype = data.dtype
    B_Jd_full = jacobian_derivative_full_doubly_left(model, data.joint_positions, data.joint_velocities)
    B_J_full, _ = jacobian_full_doubly_left(model, data.joint_positions)
    mask = _support_mask(model, dtype)[None, :, None, :]
    B_Jd_WL_B = mask * B_Jd_full[:, None]
    B_J_WL_B = mask * B_J_full[:, None]
    W_H_B = data.base_transform
    B_X_W = rm.adjoint_from_transform(W_H_B, inverse=True)
    W_v_WB = data.base_velocity(VelRepr.Inertial)
    B_Xd_W = -B_X_W @ rm.vx(W_v_WB)
    T = _block_diag_T(B_X_W, n)[:, None]
    Td = _block_diag_Td(B_Xd_W, n)[:, None]
    W_X_B = rm.adjoint_from_transform(W_H_B)
    B_v_WB = data.base_velocity(VelRepr.Body)
    W_Xd_B = W_X_B @ rm.vx(B_v_WB)
    O_X_B, O_Xd_B = W_X_B[:, None], W_Xd_B[:, None]
    return O_Xd_B @ B_J_WL_B @ T + O_X_B @ B_Jd_WL_B @ T + O_X_B @ B_J_WL_B @ Td


# =============================================================================================
# Contact frames (api/contact.py:214-511)
# =============================================================================================


def _enabled(model):
    kdp = model.kin_dyn_parameters
    idx = kdp.indices_of_enabled_collidable_points
    return kdp.contact_body[idx], kdp.contact_point[idx]


def contact_transforms(model, data: rs.OracleData):
    """``W_H_C`` of the implicit frames ``C = (W_p_C, [L])`` (contact.py:214-257)."""
    body, L_p_C = _enabled(model)
    W_H_L = data.link_transforms[:, body]
    L_H_C = np.broadcast_to(np.eye(4, dtype=data.dtype), (len(body), 4, 4)).copy()
    L_H_C[:, :3, 3] = L_p_C
    return W_H_L @ L_H_C[None]


def contact_jacobian_mixed(model, data: rs.OracleData):
    """``contact.jacobian`` with mixed input and output representation (contact.py:260-350):
    ``CW_J_WC_BW`` [N,n_cp,6,6+n]."""
    body, _ = _enabled(model)
    W_J_WL = generalized_free_floating_jacobian_inertial_output(model, data, VelRepr.Mixed)
    W_J_WC = W_J_WL[:, body]
    W_H_CW = contact_transforms(model, data)
    W_H_CW[..., :3, :3] = np.eye(3)
    CW_X_W = rm.adjoint_from_transform(W_H_CW, inverse=True)
    return CW_X_W @ W_J_WC


def contact_jacobian_derivative_mixed(model, data: rs.OracleData):
    """``contact.jacobian_derivative`` with mixed input and output representation
    (contact.py:353-511): ``CW_Jdot_WC_BW`` [N,n_cp,6,6+n]."""
    n = model.kin_dyn_parameters.number_of_joints()
    body, _ = _enabled(model)
    # input representation: T = diag(W_X_BW, I), Tdot = diag(W_X_BW vx([v_lin; 0]), 0)  (:418-429)
    W_H_BW, _ = _mixed_base_frame(data)
    W_X_BW = rm.adjoint_from_transform(W_H_BW)
    BW_v_W_BW = data.base_velocity(VelRepr.Mixed).copy()
    BW_v_W_BW[:, 3:] = 0
    W_Xd_BW = W_X_BW @ rm.vx(BW_v_W_BW)
    T = _block_diag_T(W_X_BW, n)[:, None]
    Td = _block_diag_Td(W_Xd_BW, n)[:, None]
    # link Jacobians and derivatives, inertial in / inertial out (:438-449)
    W_J_WL_W = generalized_free_floating_jacobian_inertial_output(model, data, VelRepr.Inertial)[:, body]
    W_Jd_WL_W = generalized_free_floating_jacobian_derivative_inertial(model, data)[:, body]
    # output representation (:471-481)
    W_H_CW = contact_transforms(model, data)
    W_H_CW[..., :3, :3] = np.eye(3)
    CW_X_W = rm.adjoint_from_transform(rm.transform_inverse(W_H_CW))
    CW_v_WC = rm.mv(CW_X_W, data.link_velocities[:, body])
    W_v_W_CW = np.zeros_like(CW_v_WC)
    W_v_W_CW[..., :3] = CW_v_WC[..., :3]
    CW_Xd_W = -CW_X_W @ rm.vx(W_v_W_CW)
    return CW_Xd_W @ W_J_WL_W @ T + CW_X_W @ W_Jd_WL_W @ T + CW_X_W @ W_J_WL_W @ Td


# =============================================================================================
# Mass matrix in the mixed representation (api/model.py:1529-1631)
# =============================================================================================


def _transform_M_block(M_body, X):
    """``invT^T M invT`` with ``invT = diag(X, I)`` (model.py:1529-1550)."""
    out = M_body.copy()
    Xt = np.swapaxes(X, -1, -2)
    out[:, :6, :6] = Xt @ M_body[:, :6, :6] @ X
    out[:, :6, 6:] = Xt @ M_body[:, :6, 6:]
    out[:, 6:, :6] = M_body[:, 6:, :6] @ X
    return out


def free_floating_mass_matrix_mixed(model, data: rs.OracleData):
    M_body = rs.crba(model, joint_positions=data.joint_positions)
    _, BW_H_B = _mixed_base_frame(data)
    return _transform_M_block(M_body, rm.adjoint_from_transform(BW_H_B, inverse=True))


def free_floating_mass_matrix_inverse_mixed(model, data: rs.OracleData):
    M_inv_body = np.linalg.inv(rs.crba(model, joint_positions=data.joint_positions))
    _, BW_H_B = _mixed_base_frame(data)  # model.py:1625-1628 (named B_H_BW there)
    BW_X_B = rm.adjoint_from_transform(BW_H_B)
    return _transform_M_block(M_inv_body, np.swapaxes(BW_X_B, -1, -2))


# =============================================================================================
# QP: primal-dual interior point (qpax.solve_qp; CVXGEN section 5)
# =============================================================================================

QP_MAX_ITER = 30


def _step_to_boundary(v, dv):
    """Largest alpha >= 0 with v + alpha dv >= 0 (inf if dv >= 0 everywhere)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(dv < 0, -v / dv, np.inf)
    return float(np.min(r)) if r.size else np.inf


def points_alone_in_base_subtrees(model) -> bool:
    """The structural condition under which the HIP kernel warm-starts its interior-point iteration (csrc/jxs_pack.h
    ``KParams::qp_warm``): a floating base, one to four enabled collidable points, each on a link below a DIFFERENT child
    of the base (a legged robot with one point per foot) -- the Delassus matrix is then well conditioned."""
    kdp = model.kin_dyn_parameters
    if not model.floating_base():
        return False
    parent = np.asarray(kdp.parent_array)
    bodies = [int(b) for b, e in zip(np.asarray(kdp.contact_body), np.asarray(kdp.contact_enabled)) if e]
    if not 1 <= len(bodies) <= 4:
        return False
    tops = []
    for b in bodies:
        if b == 0:
            return False
        while parent[b] != 0:
            b = int(parent[b])
        tops.append(b)
    return len(set(tops)) == len(tops)


def solve_qp_pdip(Q, q, G, h, solver_tol=1e-3, max_iter=QP_MAX_ITER, stall_guard=False, warm_start=False):
    """min 1/2 x'Qx + q'x  s.t.  Gx <= h  (no equality constraints on this path,
    rigid.py:347-349).  Returns ``(x, s, z, iterations, converged)``.

    ``warm_start`` (the HIP kernel's certificate for well-conditioned problems, not part of CVXGEN): if the unconstrained
    minimiser ``-Q^-1 q`` is feasible (``G x <= h``) it is the minimiser and is returned with 0 iterations; otherwise the
    iteration starts from CVXGEN's point as always.

    ``stall_guard`` (the HIP kernel's termination rule, not part of CVXGEN): also stop when the
    complementarity gap is below the tolerance and the residual no longer halves -- in fp32 with
    forces of 1e3 N the residual floor is above ``solver_tol`` -- or when the gap is six orders
    below the tolerance."""
    Q, q, G, h = (np.asarray(a, dtype=np.result_type(Q, np.float32)) for a in (Q, q, G, h))
    nz = G.shape[0]
    if warm_start:
        xu = np.linalg.solve(Q, -q)
        gu = G @ xu - h
        if np.max(gu) <= 0:
            return xu, -gu, np.zeros_like(gu), 0, True  # the unconstrained minimiser is feasible: it is the solution
    # ---- initialisation (CVXGEN 5.2): [Q G'; G -I][x; z] = [-q; h], then shift into the cone
    x = np.linalg.solve(Q + G.T @ G, -q + G.T @ h)
    z = G @ x - h
    alpha_p = np.max(z)  # = -min(-z)
    s = -z if alpha_p < 0 else -z + (1 + alpha_p)
    alpha_d = -np.min(z)
    if alpha_d >= 0:
        z = z + (1 + alpha_d)
    it, converged, res_prev = 0, False, np.inf
    for it in range(max_iter + 1):
        r1 = Q @ x + q + G.T @ z  # dual residual
        r3 = G @ x + s - h  # primal residual
        mu = float(s @ z) / nz
        res = max(np.max(np.abs(r1)), np.max(np.abs(r3)))
        if res < solver_tol and mu < solver_tol:
            converged = True
            break
        if stall_guard and (mu < solver_tol * 1e-6 or (mu < solver_tol and res > 0.5 * res_prev)):
            break
        res_prev = res
        if it == max_iter:
            break
        W = z / s
        H = Q + G.T @ (W[:, None] * G)
        L = np.linalg.cholesky(H)

        def newton(r2):
            # Q dx + G'dz = -r1 ; G dx + ds = -r3 ; z ds + s dz = -r2
            rhs = -r1 - G.T @ ((-r2 + z * r3) / s)
            dx = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            ds = -r3 - G @ dx
            dz = (-r2 - z * ds) / s
            return dx, ds, dz

        # predictor (affine scaling) step
        _, ds_a, dz_a = newton(s * z)
        a_aff = min(1.0, _step_to_boundary(s, ds_a), _step_to_boundary(z, dz_a))
        sigma = (float((s + a_aff * ds_a) @ (z + a_aff * dz_a)) / float(s @ z)) ** 3
        # centering-corrector step
        dx, ds, dz = newton(s * z + ds_a * dz_a - sigma * mu)
        a = min(1.0, 0.99 * min(_step_to_boundary(s, ds), _step_to_boundary(z, dz)))
        x, s, z = x + a * dx, s + a * ds, z + a * dz
    return x, s, z, it, converged


# =============================================================================================
# RigidContacts (rbda/contacts/rigid.py)
# =============================================================================================


def ineq_constraint_matrix(inactive, mu, dtype):
    """Block-diagonal friction pyramid in WORLD axes (rigid.py:476-500): per point
    ``[[1,0,-mu],[0,1,-mu],[-1,0,-mu],[0,-1,-mu],[0,0,-1],[0,0,inactive]]``."""
    n_cp = len(inactive)
    G = np.zeros((6 * n_cp, 3 * n_cp), dtype=dtype)
    blk = np.array([[1, 0, -mu], [0, 1, -mu], [-1, 0, -mu], [0, -1, -mu], [0, 0, -1], [0, 0, 0]], dtype=dtype)
    for c in range(n_cp):
        b = blk.copy()
        b[5, 2] = 1.0 if inactive[c] else 0.0
        G[6 * c : 6 * c + 6, 3 * c : 3 * c + 3] = b
    return G


def rigid_problem(model, data: rs.OracleData, *, link_forces=None, joint_torques=None):
    """Everything ``RigidContacts.compute_contact_forces`` assembles before the QP
    (rigid.py:228-340): positions, penetration data, Delassus matrix, free contact acceleration.
    ``link_forces`` are in the representation of ``data``."""
    kdp = model.kin_dyn_parameters
    N, nL = data.batch_size, kdp.number_of_links()
    dtype = data.dtype
    f_L = link_forces if link_forces is not None else np.zeros((N, nL, 6), dtype=dtype)
    tau = joint_torques if joint_torques is not None else np.zeros_like(data.joint_positions)
    W_p_C, W_pd_C = rs.collidable_points_pos_vel(
        model, link_transforms=data.link_transforms, link_velocities=data.link_velocities
    )
    delta, delta_dot, n_hat = rs.compute_penetration_data(model, W_p_C, W_pd_C)
    # references object: the wrenches go to inertial once and come back in mixed (rigid.py:263-309)
    W_f_L = rs.other_representation_to_inertial(f_L, data.velocity_representation, data.link_transforms, is_force=True)
    data_mx = dataclasses.replace(data, velocity_representation=VelRepr.Mixed)
    LW_f_L = rs.inertial_to_other_representation(W_f_L, VelRepr.Mixed, data.link_transforms, is_force=True)
    BW_nu = data_mx.generalized_velocity(VelRepr.Mixed)
    M_inv = free_floating_mass_matrix_inverse_mixed(model, data_mx)
    J_WC = contact_jacobian_mixed(model, data_mx)
    Jd_WC = contact_jacobian_derivative_mixed(model, data_mx)
    BW_vd, sdd = rs.forward_dynamics_aba(model, data_mx, joint_forces=tau, link_forces=LW_f_L)
    BW_nud_free = np.concatenate([BW_vd, sdd], axis=-1)
    # W_pdd_C of the free motion (rigid.py:503-521)
    CW_a = np.einsum("ncij,nj->nci", Jd_WC, BW_nu) + np.einsum("ncij,nj->nci", J_WC, BW_nud_free)
    a_free = CW_a[..., :3].reshape(N, -1)
    cp = model.contact_params
    inactive = delta <= 0
    baum = np.where(inactive[..., None], 0.0, (cp.K * delta + cp.D * delta_dot)[..., None] * n_hat).reshape(N, -1)
    Jl = J_WC[:, :, :3, :].reshape(N, -1, J_WC.shape[-1])
    delassus = Jl @ M_inv @ np.swapaxes(Jl, -1, -2)
    return dict(position=W_p_C, inactive=inactive, delassus=delassus, a_free=a_free, baumgarte=baum,
                J_lin=Jl, M_inv=M_inv, nud_free=BW_nud_free)  # fmt: skip


#: ``True`` selects the reduced statement of the same QP that the HIP kernel solves
#: (jaxsim_amd/csrc/jxs_rigid.inc): an inactive point is squeezed to f = 0 by its rows
#: f_z <= 0, -f_z <= 0 and the pyramid, so it is removed from the problem instead; the constant row
#: 0 <= 0 of an active point is dropped.  Same minimiser (the QP is strictly convex), different
#: interior-point iterates: the two forms agree to O(solver_tol), see tests/test_oracle_rigid.py.
REDUCED_QP = False


def compute_contact_forces(model, data: rs.OracleData, *, link_forces=None, joint_torques=None, reduced=None):
    """``RigidContacts.compute_contact_forces`` (rigid.py:228-389): inertial 6D wrenches of the
    enabled points, [N,n_cp,6], plus the QP diagnostics."""
    pb = rigid_problem(model, data, link_forces=link_forces, joint_torques=joint_torques)
    cm = model.contact_model
    N, n_cp = pb["inactive"].shape
    dtype = data.dtype
    mu = model.contact_params.mu
    f = np.zeros((N, n_cp, 3), dtype=dtype)
    info = []
    reduced = REDUCED_QP if reduced is None else reduced
    warm = reduced and points_alone_in_base_subtrees(model)  # (the kernel's initial point; the reference's statement keeps CVXGEN's)
    for e in range(N):
        Q = pb["delassus"][e] + cm.regularization_delassus * np.eye(3 * n_cp, dtype=dtype)
        q = pb["a_free"][e] - pb["baumgarte"][e]
        if not reduced:
            G = ineq_constraint_matrix(pb["inactive"][e], mu, dtype)
            x, s, z, it, ok = solve_qp_pdip(Q, q, G, np.zeros(6 * n_cp, dtype=dtype), solver_tol=cm.solver_tol)
            f[e] = x.reshape(n_cp, 3)
        else:
            act = np.flatnonzero(~pb["inactive"][e])
            it, ok = 0, True
            if act.size:
                rows = (3 * act[:, None] + np.arange(3)[None, :]).reshape(-1)
                G = ineq_constraint_matrix(np.zeros(act.size, dtype=bool), mu, dtype)
                G = G[np.arange(6 * act.size) % 6 != 5]  # drop the 0 <= 0 rows
                x, s, z, it, ok = solve_qp_pdip(Q[np.ix_(rows, rows)], q[rows], G, np.zeros(5 * act.size, dtype=dtype),
                                                solver_tol=cm.solver_tol, stall_guard=True, warm_start=warm)  # fmt: skip
                f[e, act] = x.reshape(-1, 3)
        info.append((it, ok))
    # mixed force at the point -> inertial wrench [f; p x f]   (rigid.py:368-387)
    W_f_C = np.concatenate([f, np.cross(pb["position"], f)], axis=-1)
    return W_f_C.astype(dtype), dict(problem=pb, qp=info, forces=f)


def link_contact_forces(model, data: rs.OracleData, *, link_forces=None, joint_torques=None):
    kdp = model.kin_dyn_parameters
    W_f_C, aux = compute_contact_forces(model, data, link_forces=link_forces, joint_torques=joint_torques)
    body = kdp.contact_body[kdp.indices_of_enabled_collidable_points]
    mask = (body[:, None] == np.arange(kdp.number_of_links())[None, :]).astype(W_f_C.dtype)
    return np.einsum("cl,ncj->nlj", mask, W_f_C), aux


def compute_impact_velocity(inactive, M, J_WC, nu):
    """``RigidContacts.compute_impact_velocity`` (rigid.py:176-220) for one environment."""
    Jl = J_WC[:, :3, :].copy()
    Jl[inactive] = 0
    Jl = Jl.reshape(-1, M.shape[0])
    k = Jl.shape[0]
    A = np.block([[M, -Jl.T], [Jl, np.zeros((k, k), dtype=M.dtype)]])
    b = np.concatenate([M @ nu, np.zeros(k, dtype=M.dtype)])
    return np.linalg.lstsq(A, b, rcond=None)[0][: M.shape[0]]


def update_velocity_after_impact(model, data: rs.OracleData) -> rs.OracleData:
    """``RigidContacts.update_velocity_after_impact`` (rigid.py:391-446)."""
    W_p_C, _ = rs.collidable_points_pos_vel(
        model, link_transforms=data.link_transforms, link_velocities=data.link_velocities
    )
    delta, _, _ = rs.compute_penetration_data(model, W_p_C, np.zeros_like(W_p_C))
    data_mx = dataclasses.replace(data, velocity_representation=VelRepr.Mixed)
    J_WC = contact_jacobian_mixed(model, data_mx)
    M = free_floating_mass_matrix_mixed(model, data_mx)
    nu = data_mx.generalized_velocity(VelRepr.Mixed)
    post = np.stack([compute_impact_velocity(delta[e] <= 0, M[e], J_WC[e], nu[e]) for e in range(data.batch_size)])
    W_H_BW, _ = _mixed_base_frame(data)
    W_v = rs.other_representation_to_inertial(post[:, :6], VelRepr.Mixed, W_H_BW, is_force=False)
    new = dataclasses.replace(
        data,
        base_linear_velocity=W_v[:, :3].astype(data.dtype),
        base_angular_velocity=W_v[:, 3:].astype(data.dtype),
        joint_velocities=post[:, 6:].astype(data.dtype),
    )
    return new.update_caches(model)
