"""Batched ``JaxSimModelData`` mirror whose state lives in HBM as one ``[rows][N]`` block.

Mirrors the reference container (``src/jaxsim/api/data.py:46-549``): same field and property
names, base velocity stored inertial-fixed regardless of ``velocity_representation``
(default Mixed, ``:75,151-156,187-188``), normalised ``base_orientation``, cached link
kinematics computed on demand (``_link_transforms`` / ``_link_velocities``, refreshed by a
kernel instead of at every ``replace``).  The object is immutable by convention: ``step``
returns a new one.  Batched inputs have a leading ``N`` axis like the output of
``jax.vmap``; unbatched inputs are treated as ``N = 1`` and squeezed on the way out.
"""

from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _hostmath as hm
from . import _lib, runtime
from .model import VelRepr
from .state import StateLayout, pack_state, unpack_state


def _inertial_to_other(W_array, rep, W_H_O, is_force):
    """``inertial_to_other_representation`` (``src/jaxsim/api/common.py:100-158``), batched."""
    if rep == VelRepr.Inertial:
        return W_array
    R, p = W_H_O[..., :3, :3], W_H_O[..., :3, 3]
    if rep == VelRepr.Mixed:
        R = np.broadcast_to(np.eye(3), R.shape)
    lin, ang = W_array[..., :3], W_array[..., 3:]
    Rt = np.swapaxes(R, -1, -2)
    if not is_force:  # O_X_W v = [R^T (v - p x w); R^T w]
        return np.concatenate(
            [np.einsum("...ij,...j->...i", Rt, lin - np.cross(p, ang)), np.einsum("...ij,...j->...i", Rt, ang)], -1
        )
    # W_X_O^T f = [R^T f; R^T (mu - p x f)]
    return np.concatenate(
        [np.einsum("...ij,...j->...i", Rt, lin), np.einsum("...ij,...j->...i", Rt, ang - np.cross(p, lin))], -1
    )


def _other_to_inertial(O_array, rep, W_H_O, is_force):
    """``other_representation_to_inertial`` (``src/jaxsim/api/common.py:160-222``), batched."""
    if rep == VelRepr.Inertial:
        return O_array
    R, p = W_H_O[..., :3, :3], W_H_O[..., :3, 3]
    if rep == VelRepr.Mixed:
        R = np.broadcast_to(np.eye(3), R.shape)
    lin = np.einsum("...ij,...j->...i", R, O_array[..., :3])
    ang = np.einsum("...ij,...j->...i", R, O_array[..., 3:])
    if not is_force:  # W_X_O v = [R v + p x R w; R w]
        return np.concatenate([lin + np.cross(p, ang), ang], -1)
    # O_X_W^T f = [R f; R mu + p x R f]
    return np.concatenate([lin, ang + np.cross(p, lin)], -1)


class JaxSimModelData:
    """State of N independent instances of one model, resident on the GPU."""

    def __init__(self, model, state: runtime.DeviceArray, velocity_representation: VelRepr, batched: bool):
        self._model_ref = model
        self._state = state
        self.velocity_representation = velocity_representation if type(velocity_representation) is VelRepr else VelRepr(velocity_representation)
        self._batched = bool(batched)
        self._host = None  # lazily downloaded dict of [N,...] arrays
        self._kin = None  # lazily computed (link_transforms, link_velocities)

    # -- construction -----------------------------------------------------------------------
    @staticmethod
    def build(
        model,
        base_position=None,
        base_quaternion=None,
        joint_positions=None,
        base_linear_velocity=None,
        base_angular_velocity=None,
        joint_velocities=None,
        contact_state: dict | None = None,
        velocity_representation: VelRepr = VelRepr.Mixed,
        *,
        batch_size: int | None = None,
        dtype=np.float64,
    ) -> "JaxSimModelData":
        """``JaxSimModelData.build`` (``src/jaxsim/api/data.py:65-202``).

        The base velocity arguments are expressed in ``velocity_representation`` and stored
        inertial-fixed.  ``dtype`` defaults to float64 like the reference (x64 enabled,
        ``src/jaxsim/__init__.py:17-35``); the benchmark configurations use float32.
        """
        lay = StateLayout.of(model)
        n, n_cp = lay.n_joints, lay.n_points
        given = dict(
            base_position=(base_position, 3),
            base_quaternion=(base_quaternion, 4),
            joint_positions=(joint_positions, n),
            base_linear_velocity=(base_linear_velocity, 3),
            base_angular_velocity=(base_angular_velocity, 3),
            joint_velocities=(joint_velocities, n),
        )
        N, batched = (batch_size, True) if batch_size is not None else (1, False)
        for name, (val, width) in given.items():
            if val is None:
                continue
            arr = np.asarray(val, dtype=np.float64)
            if arr.ndim == 2:
                N, batched = arr.shape[0], True
        td = (contact_state or {}).get("tangential_deformation")
        if td is not None and np.ndim(td) == 3:
            N, batched = np.shape(td)[0], True

        def prep(val, width, default=None):
            if val is None:
                out = np.zeros((N, width))
                if default is not None:
                    out[:] = default
                return out
            arr = np.asarray(val, dtype=np.float64)
            arr = np.atleast_1d(arr.squeeze()) if arr.ndim != 2 else arr
            if arr.ndim == 1:
                if arr.shape != (width,):
                    raise ValueError((arr.shape, (width,)))  # like rbda/utils.py:102-133
                arr = np.broadcast_to(arr, (N, width))
            elif arr.shape != (N, width):
                raise ValueError((arr.shape, (N, width)))
            return np.array(arr, dtype=np.float64)

        p = prep(base_position, 3)
        q = prep(base_quaternion, 4, default=[1.0, 0, 0, 0])
        s = prep(joint_positions, n)
        sd = prep(joint_velocities, n)
        vl = prep(base_linear_velocity, 3)
        va = prep(base_angular_velocity, 3)
        W_H_B = np.zeros((N, 4, 4))
        qn = q / np.linalg.norm(q, axis=-1, keepdims=True)
        W_H_B[:, :3, :3] = hm.quaternion_to_rotation(qn)
        W_H_B[:, :3, 3] = p
        W_H_B[:, 3, 3] = 1
        W_v = _other_to_inertial(np.concatenate([vl, va], -1), VelRepr(velocity_representation), W_H_B, False)
        m = np.zeros((N, n_cp, 3))
        if td is not None:
            m[:] = np.asarray(td, dtype=np.float64).reshape((-1, n_cp, 3))
        block = pack_state(
            lay,
            base_position=p,
            base_quaternion=q,
            joint_positions=s,
            base_linear_velocity=W_v[:, :3],
            base_angular_velocity=W_v[:, 3:],
            joint_velocities=sd,
            tangential_deformation=m,
            dtype=np.dtype(dtype),
        )
        tile = runtime.device_model(model, dtype).layout.tile
        return JaxSimModelData(model, runtime.DeviceArray.from_host(block, tile=tile), velocity_representation, batched)

    @staticmethod
    def zero(model, velocity_representation: VelRepr = VelRepr.Mixed, **kwargs) -> "JaxSimModelData":
        """``JaxSimModelData.zero`` (``src/jaxsim/api/data.py:204-222``)."""
        return JaxSimModelData.build(model, velocity_representation=velocity_representation, **kwargs)

    @staticmethod
    def from_state_block(model, block: np.ndarray, velocity_representation=VelRepr.Mixed) -> "JaxSimModelData":
        """Wrap a host ``[rows, N]`` block (inertial-fixed base velocity) -- no conversion."""
        lay = StateLayout.of(model)
        if block.shape[0] != lay.n_rows:
            raise ValueError((block.shape, lay.n_rows))
        tile = runtime.device_model(model, block.dtype).layout.tile
        return JaxSimModelData(model, runtime.DeviceArray.from_host(block, tile=tile), velocity_representation, True)

    # -- host views ---------------------------------------------------------------------------
    @property
    def dtype(self):
        return self._state.dtype

    @property
    def batch_size(self) -> int:
        return self._state.cols

    def state_block(self) -> np.ndarray:
        """Download the raw ``[rows, N]`` block."""
        return self._state.to_host()

    def _invalidate_caches(self) -> None:
        """The device buffer was overwritten (``step(..., inplace=True)``): forget the host copies."""
        self._host = None
        self._kin = None

    def _fields(self) -> dict:
        if self._host is None:
            self._host = unpack_state(StateLayout.of(self._model_ref), self._state.to_host())
        return self._host

    def _out(self, a: np.ndarray) -> np.ndarray:
        return a if self._batched else a[0]

    @property
    def joint_positions(self):
        return self._out(self._fields()["joint_positions"])

    @property
    def joint_velocities(self):
        return self._out(self._fields()["joint_velocities"])

    @property
    def base_position(self):
        return self._out(self._fields()["base_position"])

    @property
    def base_quaternion(self):
        return self._out(self._fields()["base_quaternion"])

    @property
    def base_orientation(self):
        """Normalised quaternion (``src/jaxsim/api/data.py:267-286``)."""
        q = self._fields()["base_quaternion"]
        norm = np.linalg.norm(q, axis=-1, keepdims=True)
        return self._out(q / (norm + np.finfo(q.dtype).eps * (norm == 0)))

    @property
    def _base_linear_velocity(self):
        return self._out(self._fields()["base_linear_velocity"])

    @property
    def _base_angular_velocity(self):
        return self._out(self._fields()["base_angular_velocity"])

    @property
    def contact_state(self) -> dict:
        return {"tangential_deformation": self._out(self._fields()["tangential_deformation"])}

    def _base_transform_batched(self) -> np.ndarray:
        f = self._fields()
        q = f["base_quaternion"].astype(np.float64)
        H = np.zeros((q.shape[0], 4, 4))
        H[:, :3, :3] = hm.quaternion_to_rotation(q / np.linalg.norm(q, axis=-1, keepdims=True))
        H[:, :3, 3] = f["base_position"]
        H[:, 3, 3] = 1
        return H

    @property
    def base_transform(self):
        return self._out(self._base_transform_batched().astype(self.dtype))

    @property
    def _base_transform(self):
        return self.base_transform

    def _base_velocity_batched(self, rep=None) -> np.ndarray:
        f = self._fields()
        W_v = np.concatenate([f["base_linear_velocity"], f["base_angular_velocity"]], -1).astype(np.float64)
        rep = self.velocity_representation if rep is None else rep
        return _inertial_to_other(W_v, rep, self._base_transform_batched(), False)

    @property
    def base_velocity(self):
        """Base 6D velocity in the active representation (``src/jaxsim/api/data.py:288-312``)."""
        return self._out(self._base_velocity_batched().astype(self.dtype))

    @property
    def generalized_velocity(self):
        v = np.concatenate([self._base_velocity_batched(), self._fields()["joint_velocities"]], -1)
        return self._out(v.astype(self.dtype))

    @property
    def generalized_position(self):
        return self.base_transform, self.joint_positions

    @contextlib.contextmanager
    def switch_velocity_representation(self, velocity_representation: VelRepr):
        """``switch_velocity_representation`` (``src/jaxsim/api/common.py:60-98``)."""
        old = self.velocity_representation
        self.velocity_representation = VelRepr(velocity_representation)
        try:
            yield self
        finally:
            self.velocity_representation = old

    # -- cached kinematics (computed by the MODE_KIN kernel on first use) -----------------------
    def _kinematics(self):
        if self._kin is None:
            model = self._model_ref
            dm = runtime.device_model(model, self.dtype)
            nL, N = model.number_of_links(), self.batch_size
            H = runtime.DeviceArray(nL * 12, N, self.dtype, tile=self._state.tile)
            V = runtime.DeviceArray(nL * 6, N, self.dtype, tile=self._state.tile)
            _lib.check(
                _lib.load().jxs_refresh_kinematics(
                    dm.handle, C.c_void_p(self._state.ptr), C.c_void_p(H.ptr), C.c_void_p(V.ptr), N, runtime._sp()
                ),
                "jxs_refresh_kinematics",
            )
            Hh = H.to_host().T.reshape(N, nL, 3, 4)
            full = np.zeros((N, nL, 4, 4), dtype=self.dtype)
            full[:, :, :3, :] = Hh
            full[:, :, 3, 3] = 1
            self._kin = (full, V.to_host().T.reshape(N, nL, 6).copy())
        return self._kin

    @property
    def _link_transforms(self):
        return self._out(self._kinematics()[0])

    @property
    def _link_velocities(self):
        return self._out(self._kinematics()[1])

    # -- functional update ----------------------------------------------------------------------
    def replace(
        self,
        model,
        joint_positions=None,
        joint_velocities=None,
        base_quaternion=None,
        base_linear_velocity=None,
        base_angular_velocity=None,
        base_position=None,
        *,
        contact_state: dict | None = None,
        validate: bool = False,
    ) -> "JaxSimModelData":
        """``JaxSimModelData.replace`` (``src/jaxsim/api/data.py:405-523``): the quaternion is
        re-normalised; base velocities, when given, are in the active representation."""
        f = self._fields()
        N = self.batch_size

        def pick(new, old):
            if new is None:
                return old
            a = np.asarray(new, dtype=np.float64)
            if old.size == 0:  # (a model without joints: the empty joint arrays of the reference)
                return old
            return np.broadcast_to(a.reshape((-1,) + old.shape[1:]), old.shape).copy()

        q = pick(base_quaternion, f["base_quaternion"]).astype(np.float64)
        nrm = np.linalg.norm(q, axis=-1, keepdims=True)
        q = q / np.where(nrm == 0, 1.0, nrm)
        p = pick(base_position, f["base_position"])
        vl, va = f["base_linear_velocity"], f["base_angular_velocity"]
        if base_linear_velocity is not None or base_angular_velocity is not None:
            act = self._base_velocity_batched()
            lin = pick(base_linear_velocity, act[:, :3])
            ang = pick(base_angular_velocity, act[:, 3:])
            H = np.zeros((N, 4, 4))
            H[:, :3, :3] = hm.quaternion_to_rotation(q)
            H[:, :3, 3] = p
            H[:, 3, 3] = 1
            W_v = _other_to_inertial(np.concatenate([lin, ang], -1), self.velocity_representation, H, False)
            vl, va = W_v[:, :3], W_v[:, 3:]
        m = f["tangential_deformation"]
        if contact_state is not None and "tangential_deformation" in contact_state:
            m = pick(contact_state["tangential_deformation"], m)
        block = pack_state(
            StateLayout.of(model),
            base_position=p,
            base_quaternion=q,
            joint_positions=pick(joint_positions, f["joint_positions"]),
            base_linear_velocity=vl,
            base_angular_velocity=va,
            joint_velocities=pick(joint_velocities, f["joint_velocities"]),
            tangential_deformation=m,
            dtype=self.dtype,
        )
        return JaxSimModelData(
            model, runtime.DeviceArray.from_host(block, tile=self._state.tile), self.velocity_representation, self._batched
        )

    def valid(self, model) -> bool:
        """Shape compatibility check (``src/jaxsim/api/data.py:525-549``)."""
        return self._state.rows == StateLayout.of(model).n_rows

    def copy(self) -> "JaxSimModelData":
        return JaxSimModelData(self._model_ref, self._state.copy(), self.velocity_representation, self._batched)


def random_model_data(
    model,
    *,
    batch_size: int = 1,
    seed: int = 0,
    velocity_representation: VelRepr = VelRepr.Mixed,
    base_pos_bounds=((-1, -1, 0.5), (1, 1, 1)),
    base_rpy_bounds=((-np.pi,) * 3, (np.pi,) * 3),
    base_vel_lin_bounds=((-1,) * 3, (1,) * 3),
    base_vel_ang_bounds=((-1,) * 3, (1,) * 3),
    joint_vel_bounds=(-1.0, 1.0),
    dtype=np.float64,
) -> JaxSimModelData:
    """Same distribution as the reference's ``random_model_data``
    (``src/jaxsim/api/data.py:552-682``) drawn from NumPy's PCG64 -- JAX's threefry stream
    cannot be reproduced bit-exactly and need not be."""
    rng = np.random.default_rng(seed)
    kdp = model.kin_dyn_parameters
    N, n = batch_size, kdp.number_of_joints()
    p = rng.uniform(*np.array(base_pos_bounds, dtype=float), size=(N, 3))
    q = hm.rpy_to_quaternion(rng.uniform(*np.array(base_rpy_bounds, dtype=float), size=(N, 3)))
    lo = np.maximum(kdp.position_limits_min, -10.0)
    hi = np.minimum(kdp.position_limits_max, 10.0)
    s = rng.uniform(lo, hi, size=(N, n)) if n else np.zeros((N, 0))
    sd = rng.uniform(*joint_vel_bounds, size=(N, n)) if n else np.zeros((N, 0))
    vl = rng.uniform(*np.array(base_vel_lin_bounds, dtype=float), size=(N, 3))
    va = rng.uniform(*np.array(base_vel_ang_bounds, dtype=float), size=(N, 3))
    if not model.floating_base():
        vl, va = np.zeros_like(vl), np.zeros_like(va)
    return JaxSimModelData.build(
        model,
        base_position=p,
        base_quaternion=q,
        joint_positions=s,
        base_linear_velocity=vl,
        base_angular_velocity=va,
        joint_velocities=sd,
        velocity_representation=velocity_representation,
        batch_size=N,
        dtype=dtype,
    )
