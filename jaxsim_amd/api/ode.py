"""``jaxsim.api.ode`` mirror: the system dynamics ``x -> dx/dt`` the reference's integrators call and its contact-model
benchmarks time (``src/jaxsim/api/ode.py:16-225``, ``tests/test_benchmark.py:103-139``).

One launch of the ``MODE_DYN`` kernels (``jxs_system_dynamics``): contact forces of the model's contact model, summed per
link, plus the external link forces through ABA -- the step without the actuation model and without the integrator.
There is no CPU fallback; the small representation changes of the returned base acceleration are NumPy on ``[N, 6]``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib, runtime
from ..data import JaxSimModelData
from ..model import JaxSimModel, VelRepr
from ..runtime import DeviceArray
from ..state import StateLayout, unpack_state
from . import model as _m


def _is_soft(model: JaxSimModel) -> bool:
    return type(model.contact_model).__name__ not in ("RigidContacts", "RelaxedRigidContacts")


def system_dynamics_device(
    model: JaxSimModel,
    data: JaxSimModelData,
    *,
    link_forces=None,
    joint_torques=None,
    force_repr: VelRepr = VelRepr.Inertial,
    baumgarte_quaternion_regularization: float = 1.0,
    want_derivative: bool = True,
    want_link_contact_forces: bool = False,
    out: DeviceArray | None = None,
    out_forces: DeviceArray | None = None,
):
    """The launch behind every function of this module (extension: the results stay on the device).

    Returns ``(xdot, W_f_L)``: ``xdot`` is a ``DeviceArray`` in the layout of the state block -- every row the time
    derivative of the state row it stands for, inertial-fixed like the state -- and ``W_f_L`` the ``[nL * 6][N]`` block of
    the inertial link contact wrenches; either is ``None`` when not asked for.  ``link_forces`` are read in
    ``force_repr``."""
    st = data._state
    dtype, N = st.dtype, st.cols
    dm = _m._device_model_fast(model, dtype)
    if "_dyn_mode" not in dm.__dict__:  # first call: the model-specialised kernel of this mode, following specialize.policy()
        from .. import specialize as _sp

        dm.__dict__["_dyn_mode"] = _sp.dyn_mode_of(model)
        _sp.ensure_mode(dm, model, dm.__dict__["_dyn_mode"])
    if _m._exceptions_enabled():
        _m._check_quaternion(model, data, normalized=False)  # ABA receives data.base_orientation (normalised)
    nL, n = model.number_of_links(), model.dofs()
    f = _m._as_device(link_forces, nL * 6, N, dtype, (nL, 6), st.tile)
    tau = _m._as_device(joint_torques, n, N, dtype, (n,), st.tile)
    xdot = (out if out is not None else DeviceArray.like(st)) if want_derivative else None
    W_f = (out_forces if out_forces is not None else DeviceArray(nL * 6, N, dtype, tile=st.tile)) if want_link_contact_forces else None
    _lib.check(
        _lib.load().jxs_system_dynamics(
            dm.handle, C.c_void_p(st.ptr), _m._ptr(tau), _m._ptr(f), int(force_repr), float(baumgarte_quaternion_regularization),
            _m._ptr(xdot), _m._ptr(W_f), N, runtime._sp(),
        ),
        "jxs_system_dynamics",
    )  # fmt: skip
    if _m._exceptions_enabled():
        _m._check_solver_faults(model, data)
    return xdot, W_f


def _contact_state_of(model: JaxSimModel, data: JaxSimModelData, fields: dict | None) -> dict:
    """What ``system_acceleration`` returns as its third item (``api/ode.py:53-71``): the contact model's
    ``update_contact_state`` of the contact-state derivative -- ``{"tangential_deformation": m_dot}`` for SoftContacts
    (the key names the state, the value is its RATE: ``rbda/contacts/soft.py:164-177``), ``{}`` for the rigid models --
    and ``data.contact_state`` itself for a model without collidable points."""
    if model.kin_dyn_parameters.number_of_collidable_points() == 0:
        return data.contact_state
    if not _is_soft(model):
        return {}
    return {"tangential_deformation": data._out(fields["tangential_deformation"])}


def system_acceleration(model: JaxSimModel, data: JaxSimModelData, *, link_forces=None, joint_torques=None):
    """``system_acceleration`` (``src/jaxsim/api/ode.py:16-131``): ``(base acceleration in the active representation,
    joint accelerations, contact state)``.

    With inertial data -- how the integrators and ``system_dynamics`` call it -- this is one launch.  With Body / Mixed
    data the reference adds the INERTIAL contact wrenches to ``link_forces`` and hands the sum to ABA as wrenches of the
    data's representation (``:77-118``); that is reproduced literally: contact wrenches from one launch
    (``link_forces`` read in the data's representation, like the contact models do), the sum through
    ``forward_dynamics_aba``."""
    rep = data.velocity_representation
    N, nL = data.batch_size, model.number_of_links()
    has_points = model.kin_dyn_parameters.number_of_collidable_points() > 0
    if rep == VelRepr.Inertial or not has_points:
        if not has_points:  # api/ode.py:57: no contact block at all
            vd, sdd = _m.forward_dynamics_aba(model, data, joint_forces=joint_torques, link_forces=link_forces)
            return vd, sdd, _contact_state_of(model, data, None)
        xdot, _ = system_dynamics_device(model, data, link_forces=link_forces, joint_torques=joint_torques, force_repr=VelRepr.Inertial)
        fields = unpack_state(StateLayout.of(model), xdot.to_host())
        W_vd = np.concatenate([fields["base_linear_velocity"], fields["base_angular_velocity"]], -1)
        return data._out(W_vd), data._out(fields["joint_velocities"]), _contact_state_of(model, data, fields)
    # (the derivative block is only needed for the deformation rates of SoftContacts; without it that model stops before ABA)
    xdot, W_f = system_dynamics_device(model, data, link_forces=link_forces, joint_torques=joint_torques, force_repr=rep,
                                       want_derivative=_is_soft(model), want_link_contact_forces=True)  # fmt: skip
    fields = unpack_state(StateLayout.of(model), xdot.to_host()) if xdot is not None else None
    W_f_L = W_f.to_host().T.reshape(N, nL, 6).astype(np.float64)
    f_L = np.zeros((N, nL, 6)) if link_forces is None else np.broadcast_to(np.asarray(_host_link_forces(link_forces, N, nL), np.float64), (N, nL, 6))
    vd, sdd = _m.forward_dynamics_aba(model, data, joint_forces=joint_torques, link_forces=f_L + W_f_L)
    return vd, sdd, _contact_state_of(model, data, fields)


def _host_link_forces(x, N: int, nL: int) -> np.ndarray:
    """``link_forces`` in any accepted form as a host ``[N, nL, 6]`` (or broadcastable) array."""
    if isinstance(x, DeviceArray):
        return x.to_host().T.reshape(N, nL, 6)
    a = np.asarray(x, dtype=np.float64)
    return a.reshape(nL, 6) if a.size == nL * 6 else a.reshape(N, nL, 6)


def system_position_dynamics(data: JaxSimModelData, baumgarte_quaternion_regularization: float = 1.0):
    """``system_position_dynamics`` (``src/jaxsim/api/ode.py:134-171``): ``(pdot_B, Qdot, sdot)``.  Like the reference it
    reads ``data.base_velocity`` in the ACTIVE representation and treats it as inertial-fixed -- meaningful under
    ``switch_velocity_representation(VelRepr.Inertial)``, which is how ``system_dynamics`` calls it."""
    from .. import _hostmath as hm

    f = data._fields()
    v = np.asarray(data._base_velocity_batched(), np.float64)
    w = v[:, 3:6]
    pd = v[:, 0:3] + np.cross(w, f["base_position"].astype(np.float64))
    q = f["base_quaternion"].astype(np.float64)
    nrm = np.linalg.norm(q, axis=-1, keepdims=True)
    qn = q / (nrm + np.finfo(np.float64).eps * (nrm == 0))  # data.base_orientation
    Qd = hm.quaternion_derivative(qn, w, K=float(baumgarte_quaternion_regularization))
    return data._out(pd.astype(data.dtype)), data._out(Qd.astype(data.dtype)), data.joint_velocities


def system_dynamics(
    model: JaxSimModel,
    data: JaxSimModelData,
    *,
    link_forces=None,
    joint_torques=None,
    baumgarte_quaternion_regularization: float = 1.0,
) -> dict:
    """``system_dynamics`` (``src/jaxsim/api/ode.py:174-225``): the derivative of the state as a dictionary keyed like
    the integrators' state -- ONE launch (``jxs_system_dynamics``).  Evaluated in inertial-fixed representation whatever
    the representation of ``data`` (``:204``), so ``link_forces`` are read as inertial wrenches there, like the
    reference does."""
    xdot, _ = system_dynamics_device(
        model, data, link_forces=link_forces, joint_torques=joint_torques, force_repr=VelRepr.Inertial,
        baumgarte_quaternion_regularization=baumgarte_quaternion_regularization,
    )  # fmt: skip
    f = unpack_state(StateLayout.of(model), xdot.to_host())
    o = data._out
    return dict(
        base_position=o(f["base_position"]),
        base_quaternion=o(f["base_quaternion"]),
        joint_positions=o(f["joint_positions"]),
        base_linear_velocity=o(f["base_linear_velocity"]),
        base_angular_velocity=o(f["base_angular_velocity"]),
        joint_velocities=o(f["joint_velocities"]),
        contact_state=_contact_state_of(model, data, f),
    )
