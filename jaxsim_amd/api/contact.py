"""``jaxsim.api.contact`` mirror: parameter estimation helper (host, build-time)."""

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
    return SoftContactsParams.build_default_from_jaxsim_model(
        model,
        standard_gravity=standard_gravity,
        static_friction_coefficient=static_friction_coefficient,
        max_penetration=max_penetration,
        number_of_active_collidable_points_steady_state=number_of_active_collidable_points_steady_state,
        damping_ratio=damping_ratio,
    )
