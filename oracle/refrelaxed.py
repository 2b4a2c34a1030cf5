"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's ``RelaxedRigidContacts`` model
(the contact model of the reference's own ``test_simulation_step``
benchmark, ``tests/test_benchmark.py:142-152``).

Follows ``src/jaxsim/rbda/contacts/relaxed_rigid.py``:

* ``RelaxedRigidContactsParams`` (``:29-75``, defaults),
* ``_regularizers`` (``:490-653``): impedance ``xi``, reference acceleration ``a_ref`` and the diagonal
  regulariser ``r`` -- **component-wise** in the world axes, because the "position in the constraint
  frame" is the 3-vector ``-delta n`` (``:343``).  ``parameters.K`` / ``parameters.D`` are read and
  then shadowed by the stiffness / damping derived from the time constant (``:567-568``): they have no
  effect, reproduced as such,
* ``compute_contact_forces`` (``:284-488``): ``A = J_l M^-1 J_l^T + diag(r)``, ``b = a_free - a_ref``,
  contact forces = minimiser of ``|A x + b|^2``.

Third-party arithmetic that is NOT under ``/root/reference``: ``optax.lbfgs`` (unpinned,
``pyproject.toml:47-59``).  The reference runs it from a Hunt/Crossley initial guess until
``|grad| < tol`` (1e-6) or 50 iterations inside ``jax.lax.custom_linear_solve(A, -b)`` (``:459-465``),
i.e. it asks for the solution of ``A x = -b``.  The package is absent, so the iterates are **parity
unpinned**; this file returns the point they converge to, ``x = -A^-1 b`` over the rows of the active
points (``A`` is positive definite there when ``mu > 0`` and ``xi < 1``; rows of inactive points are
identically zero in ``A`` and ``b`` and keep their zero initial guess).

How far an early-stopped L-BFGS can be from that point depends on the conditioning.  With
``estimate_good_contact_parameters`` (``mu = 0.5``, the reference's benchmark idiom,
``tests/test_benchmark.py:142-152``) ``r`` is ~1e-1 of the Delassus entries and the system is benign.
With the bare defaults (``mu = 0.005``) ``r`` is ~1e-6 of them: where ``J_l M^-1 J_l^T`` is singular
(several points on one rigid body) the force components in its null space are decided by ``r`` alone,
they are large (``b_null / r``) and -- because ``r`` differs between the axes and the points -- feed
back into the range-space components at O(1).  The reference's 50 L-BFGS iterations on the squared
residual cannot be expected to resolve those directions, so for such states its output is solver-
dependent and not reproducible here; the converged solution is the documented target.  The
reference's only test of this model (``tests/test_simulations.py:295-346``: the box comes to rest at
z = 0.05 +- 1e-4 with the bare defaults, a symmetric state with ``b_null = 0``) is re-expressed in
``tests/test_oracle_relaxed.py`` and holds for the converged solution.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from . import refrigid
from . import refstep as rs


@dataclasses.dataclass
class RelaxedRigidContactsParams:
    """Defaults of ``RelaxedRigidContactsParams`` (relaxed_rigid.py:29-75)."""

    time_constant: float = 0.02
    damping_coefficient: float = 1.0
    d_min: float = 0.9
    d_max: float = 0.95
    width: float = 0.001
    midpoint: float = 0.5
    power: float = 2.0
    K: float = 0.0
    D: float = 0.0
    mu: float = 0.005


def regularizers(model, position_constraint, velocity_constraint, parameters):
    """``RelaxedRigidContacts._regularizers`` (relaxed_rigid.py:490-653).

    ``position_constraint`` / ``velocity_constraint``: [N, n_cp, 3].  Returns ``a_ref`` and ``r`` as
    [N, 3 n_cp] (the reference concatenates the per-point 3-vectors)."""
    cp = parameters
    kdp = model.kin_dyn_parameters
    dtype = position_constraint.dtype
    Om, zeta, xi_min, xi_max = cp.time_constant, cp.damping_coefficient, cp.d_min, cp.d_max
    width, mid, p, mu = cp.width, cp.midpoint, cp.power, cp.mu
    body = np.asarray(kdp.contact_body)[kdp.indices_of_enabled_collidable_points]
    # inv(M_L[link, :3, :3]) = I / m of the parent link (api/model.py link_spatial_inertia_matrices)
    inv_m = (1.0 / np.asarray(kdp.link_mass, dtype=float)[body]).astype(dtype)

    pos, vel = position_constraint, velocity_constraint
    with np.errstate(invalid="ignore", divide="ignore"):
        imp_x = np.abs(pos) / width
        imp_a = (1.0 / np.power(mid, p - 1)) * np.power(imp_x, p)
        imp_b = 1 - (1.0 / np.power(1 - mid, p - 1)) * np.power(1 - imp_x, p)
        imp_y = np.where(imp_x < mid, imp_a, imp_b)
        xi = xi_min + imp_y * (xi_max - xi_min)
        xi = np.clip(xi, xi_min, xi_max)
        xi = np.where(imp_x > 1.0, xi_max, xi)
    K = 1 / (xi_max * Om * zeta) ** 2  # :567 (shadows parameters.K)
    D = 2 / (xi_max * Om)  # :568
    a_ref = -(D * vel + K * xi * pos)
    R = (2 * mu**2 * (1 - xi) / (xi + 1e-12)) * (1 + mu**2) * inv_m[None, :, None]
    is_active = (np.einsum("nck,nck->nc", pos, pos) > 0).astype(dtype)[..., None]
    N = pos.shape[0]
    return (a_ref * is_active).reshape(N, -1).astype(dtype), (R * is_active).reshape(N, -1).astype(dtype)


def relaxed_problem(model, data: rs.OracleData, *, link_forces=None, joint_torques=None):
    """``A`` and ``b`` of relaxed_rigid.py:331-397."""
    pb = refrigid.rigid_problem(model, data, link_forces=link_forces, joint_torques=joint_torques)
    W_p_C, W_pd_C = rs.collidable_points_pos_vel(
        model, link_transforms=data.link_transforms, link_velocities=data.link_velocities
    )
    delta, _, n_hat = rs.compute_penetration_data(model, W_p_C, W_pd_C)
    position_constraint = -delta[..., None] * n_hat
    a_ref, r = regularizers(model, position_constraint, W_pd_C, model.contact_params)
    act = np.repeat(delta > 0, 3, axis=-1)  # rows of J_l, Jdot_l masked by (delta > 0) (:369-381)
    G = pb["delassus"] * act[:, :, None] * act[:, None, :]
    a_free = pb["a_free"] * act
    A = G + np.einsum("ni,ij->nij", r, np.eye(r.shape[-1], dtype=r.dtype))
    return dict(A=A, b=a_free - a_ref, active=delta > 0, position=W_p_C, a_ref=a_ref, r=r, J_lin=pb["J_lin"])


def compute_contact_forces(model, data: rs.OracleData, *, link_forces=None, joint_torques=None):
    """``RelaxedRigidContacts.compute_contact_forces`` (relaxed_rigid.py:284-488): inertial 6D
    wrenches of the enabled points, [N, n_cp, 6]."""
    pb = relaxed_problem(model, data, link_forces=link_forces, joint_torques=joint_torques)
    N, n_cp = pb["active"].shape
    dtype = data.dtype
    f = np.zeros((N, n_cp, 3), dtype=dtype)
    for e in range(N):
        act = np.flatnonzero(pb["active"][e])
        if act.size == 0:
            continue
        rows = (3 * act[:, None] + np.arange(3)[None, :]).reshape(-1)
        x = np.linalg.solve(pb["A"][e][np.ix_(rows, rows)], -pb["b"][e][rows])
        f[e, act] = x.reshape(-1, 3)
    W_f_C = np.concatenate([f, np.cross(pb["position"], f)], axis=-1)
    return W_f_C.astype(dtype), dict(problem=pb, forces=f)


def link_contact_forces(model, data: rs.OracleData, *, link_forces=None, joint_torques=None):
    kdp = model.kin_dyn_parameters
    W_f_C, aux = compute_contact_forces(model, data, link_forces=link_forces, joint_torques=joint_torques)
    body = np.asarray(kdp.contact_body)[kdp.indices_of_enabled_collidable_points]
    mask = (body[:, None] == np.arange(kdp.number_of_links())[None, :]).astype(W_f_C.dtype)
    return np.einsum("cl,ncj->nlj", mask, W_f_C), aux
