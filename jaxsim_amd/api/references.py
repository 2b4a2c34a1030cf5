"""``jaxsim.api.references`` for the step path: ``JaxSimModelReferences``
(``src/jaxsim/api/references.py:23-449``), the container user code fills with joint force references
and external link forces before calling ``js.model.step`` (``README``-style loops pass
``references.link_forces(model, data)`` / ``references.joint_force_references(model)``).

Like the reference, the link forces are **stored inertial-fixed** and converted from / to the active
velocity representation with the link transforms of the data object (``:205-247``, ``:431-449``);
unlike it, every array carries a leading batch axis when the data object is batched (``[N, nL, 6]`` /
``[N, n]``), matching ``jaxsim_amd.api.model.step``.  Host-side NumPy: the conversion is a handful of
3x3 products per link, done once per user update, not per step.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from ..data import JaxSimModelData, _inertial_to_other, _other_to_inertial
from ..model import JaxSimModel, VelRepr


def _idxs(names, all_names, what):
    if names is None:
        return np.arange(len(all_names))
    if isinstance(names, str):
        names = (names,)
    missing = [n for n in names if n not in all_names]
    if missing:
        raise ValueError(f"unknown {what} names: {missing}")
    return np.array([all_names.index(n) for n in names], dtype=int)


@dataclasses.dataclass(frozen=True)
class JaxSimModelReferences:
    """References of a model: ``_link_forces`` (inertial-fixed 6D forces, ``[(N,) nL, 6]``) and
    ``_joint_force_references`` (``[(N,) n]``)."""

    _link_forces: np.ndarray
    _joint_force_references: np.ndarray
    velocity_representation: VelRepr = VelRepr.Inertial

    # -- construction (references.py:36-132) ----------------------------------------------------
    @staticmethod
    def zero(model: JaxSimModel, data: JaxSimModelData | None = None, velocity_representation: VelRepr = VelRepr.Inertial):
        return JaxSimModelReferences.build(model=model, data=data, velocity_representation=velocity_representation)

    @staticmethod
    def build(
        model: JaxSimModel,
        joint_force_references=None,
        link_forces=None,
        data: JaxSimModelData | None = None,
        velocity_representation: VelRepr | None = None,
    ) -> "JaxSimModelReferences":
        nL, n = model.number_of_links(), model.dofs()
        batch = (data.batch_size,) if data is not None and data._batched else ()
        tau = np.zeros(batch + (n,)) if joint_force_references is None else np.asarray(joint_force_references, dtype=float)
        f_L = np.zeros(batch + (nL, 6)) if link_forces is None else np.asarray(link_forces, dtype=float)
        if tau.shape[-1:] != (n,) or f_L.shape[-2:] != (nL, 6):
            raise ValueError(f"expected joint forces [..., {n}] and link forces [..., {nL}, 6], got {tau.shape} and {f_L.shape}")
        if velocity_representation is None:
            velocity_representation = getattr(data, "velocity_representation", VelRepr.Inertial)
        refs = JaxSimModelReferences(
            _link_forces=np.zeros_like(f_L), _joint_force_references=tau.copy(), velocity_representation=velocity_representation
        )
        if velocity_representation == VelRepr.Inertial:
            return dataclasses.replace(refs, _link_forces=f_L.copy())
        return refs.apply_link_forces(forces=f_L, model=model, data=data, link_names=model.link_names(), additive=False)

    def valid(self, model: JaxSimModel | None = None) -> bool:
        """Shapes compatible with the model (references.py:134-162)."""
        if model is None:
            return True
        return self._joint_force_references.shape[-1:] == (model.dofs(),) and self._link_forces.shape[-2:] == (
            model.number_of_links(), 6)  # fmt: skip

    def switch_velocity_representation(self, velocity_representation: VelRepr) -> "JaxSimModelReferences":
        """The stored forces are inertial-fixed: switching only changes how they are read and written
        (the reference offers this as a context manager on a mutable object, api/common.py:58-96)."""
        return dataclasses.replace(self, velocity_representation=velocity_representation)

    # -- extraction (references.py:168-300) -----------------------------------------------------
    def _transforms(self, model, data, idxs):
        if data is None:
            raise ValueError(f"Missing model data to use a representation different from {VelRepr.Inertial.name}")
        if not data.valid(model=model):
            raise ValueError("The provided data is not valid for the model")
        W_H_L = np.asarray(data._link_transforms)
        return W_H_L[..., idxs, :, :]

    def link_forces(self, model: JaxSimModel | None = None, data: JaxSimModelData | None = None, link_names=None):
        """Link forces in the active representation (``L_f_L`` body, ``LW_f_L`` mixed, ``W_f_L``
        inertial)."""
        W_f_L = self._link_forces
        if model is None:
            if self.velocity_representation != VelRepr.Inertial:
                raise ValueError(f"Missing model to use a representation different from {VelRepr.Inertial.name}")
            if link_names is not None:
                raise ValueError("Link names cannot be provided without a model")
            return W_f_L
        idxs = _idxs(link_names, model.link_names(), "link")
        if self.velocity_representation == VelRepr.Inertial:
            return W_f_L[..., idxs, :]
        return _inertial_to_other(W_f_L[..., idxs, :], self.velocity_representation, self._transforms(model, data, idxs), True)

    def joint_force_references(self, model: JaxSimModel | None = None, joint_names=None):
        if model is None:
            if joint_names is not None:
                raise ValueError("Joint names cannot be provided without a model")
            return self._joint_force_references
        if not self.valid(model=model):
            raise ValueError("The actuation object is not compatible with the provided model")
        return self._joint_force_references[..., _idxs(joint_names, model.joint_names(), "joint")]

    # -- storing (references.py:306-449) --------------------------------------------------------
    def set_joint_force_references(self, forces, model: JaxSimModel | None = None, joint_names=None) -> "JaxSimModelReferences":
        forces = np.asarray(forces, dtype=float)
        if model is None:
            return dataclasses.replace(self, _joint_force_references=np.atleast_1d(forces).copy())
        if not self.valid(model=model):
            raise ValueError("The references object is not compatible with the provided model")
        out = self._joint_force_references.copy()
        out[..., _idxs(joint_names, model.joint_names(), "joint")] = forces
        return dataclasses.replace(self, _joint_force_references=out)

    def apply_link_forces(
        self,
        forces,
        model: JaxSimModel | None = None,
        data: JaxSimModelData | None = None,
        link_names=None,
        additive: bool = False,
    ) -> "JaxSimModelReferences":
        """Set (or add to) the forces of the given links; ``forces`` are in the active representation."""
        f_L = np.asarray(forces, dtype=float)
        if f_L.ndim == 1:
            f_L = f_L[None, :]
        if model is None:
            if self.velocity_representation != VelRepr.Inertial:
                raise ValueError(f"Missing model to use a representation different from {VelRepr.Inertial.name}")
            if link_names is not None:
                raise ValueError("Link names cannot be provided without a model")
            base = self._link_forces if additive else np.zeros_like(self._link_forces)
            return dataclasses.replace(self, _link_forces=base + f_L)
        if isinstance(link_names, str):
            link_names = (link_names,)
        if link_names is not None and len(link_names) != f_L.shape[-2]:
            raise ValueError(f"The number of link names ({len(link_names)}) must match the number of forces ({f_L.shape[-2]})")
        idxs = _idxs(link_names, model.link_names(), "link")
        if self.velocity_representation == VelRepr.Inertial:
            W_f_L = f_L
        else:
            W_f_L = _other_to_inertial(f_L, self.velocity_representation, self._transforms(model, data, idxs), True)
        out = np.array(np.broadcast_to(self._link_forces, np.broadcast_shapes(self._link_forces.shape, W_f_L.shape[:-2] + self._link_forces.shape[-2:])))
        out[..., idxs, :] = (out[..., idxs, :] if additive else 0.0) + W_f_L
        return dataclasses.replace(self, _link_forces=out)

    def apply_frame_forces(self, forces, model: JaxSimModel, data: JaxSimModelData, frame_names=None, additive: bool = False):
        """``apply_frame_forces`` (references.py:451-560): 6D forces given at frames (rigidly attached to a
        link) in the active representation with the *frame* as body frame; they are converted to inertial
        with ``W_H_F = W_H_L L_H_F``, summed per parent link, and set as the forces of **all** links
        (links without a listed frame get zero unless ``additive``), like the reference."""
        f_F = np.asarray(forces, dtype=float)
        if f_F.ndim == 1:
            f_F = f_F[None, :]
        kdp = model.kin_dyn_parameters
        if isinstance(frame_names, str):
            frame_names = (frame_names,)
        idxs = _idxs(frame_names, kdp.frame_names, "frame")
        if len(idxs) != f_F.shape[-2]:
            raise ValueError(f"The number of frame names ({len(idxs)}) must match the number of forces ({f_F.shape[-2]})")
        body = np.asarray(kdp.frame_body)[idxs]
        if self.velocity_representation == VelRepr.Inertial:
            W_f_F = f_F
        else:
            W_H_L = self._transforms(model, data, body)
            W_H_F = W_H_L @ np.asarray(kdp.frame_transform)[idxs]
            W_f_F = _other_to_inertial(f_F, self.velocity_representation, W_H_F, True)
        mask = (body[:, None] == np.arange(model.number_of_links())[None, :]).astype(float)
        W_f_L = np.einsum("fl,...fk->...lk", mask, W_f_F)
        inertial = dataclasses.replace(self, velocity_representation=VelRepr.Inertial)
        out = inertial.apply_link_forces(W_f_L, model=model, data=data, additive=additive)
        return dataclasses.replace(out, velocity_representation=self.velocity_representation)
