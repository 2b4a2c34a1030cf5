"""Static per-model tables (host, NumPy): the constants every kernel consumes.

Mirror of the reference's ``KinDynParameters`` (``src/jaxsim/api/kin_dyn_parameters.py:86-284``)
without the JAX pytree plumbing: parent array, motion subspaces, joint model
(``lambda_H_pre``, ``suc_H_i``, types, axes), link inertial parameters, joint
parameters and collidable points.  Everything is ``float64`` NumPy; the C-ABI
library converts to the model dtype when the device copy is created.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from . import _hostmath as hm
from .parsers.urdf import FIXED, PRISMATIC, REVOLUTE, ModelDescription


@dataclasses.dataclass(frozen=True)
class KinDynParameters:
    link_names: tuple[str, ...]
    joint_names: tuple[str, ...]
    frame_names: tuple[str, ...]
    #: lambda(i); parent_array[0] == -1 (``kin_dyn_parameters.py:193-198``)
    parent_array: np.ndarray  # [nL] int
    #: joint types of joints 1..n (1 revolute, 2 prismatic) (``math/joint_model.py:192-199``)
    joint_types: np.ndarray  # [n] int
    joint_axis: np.ndarray  # [n,3] unit axes
    #: S_i = [0; axis] revolute, [axis; 0] prismatic, zeros at index 0 (``kin_dyn_parameters.py:239-261``)
    motion_subspaces: np.ndarray  # [nL,6]
    #: parent link -> joint predecessor frame; entry 0 is identity (``math/joint_model.py:70-98``)
    lambda_H_pre: np.ndarray  # [nL,4,4]
    #: joint successor frame -> child link; entry 0 = base pose in the model frame
    suc_H_i: np.ndarray  # [nL,4,4]
    #: link spatial inertias at the link origin, rebuilt like ``Inertia.to_sixd``
    link_spatial_inertia: np.ndarray  # [nL,6,6]
    link_mass: np.ndarray  # [nL]
    link_com: np.ndarray  # [nL,3]
    link_inertia_com: np.ndarray  # [nL,3,3]
    # joint parameters (``kin_dyn_parameters.py:502-571``)
    friction_static: np.ndarray  # [n]
    friction_viscous: np.ndarray  # [n]
    position_limits_min: np.ndarray  # [n]
    position_limits_max: np.ndarray  # [n]
    position_limit_spring: np.ndarray  # [n]
    position_limit_damper: np.ndarray  # [n]
    # contact parameters (``kin_dyn_parameters.py:765-840``)
    contact_body: np.ndarray  # [n_cp] int
    contact_point: np.ndarray  # [n_cp,3]
    contact_enabled: np.ndarray  # [n_cp] bool
    # frame parameters (``kin_dyn_parameters.py:843-893``): parent link index and pose in it
    frame_body: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, dtype=np.int64))  # [n_frames]
    frame_transform: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((0, 4, 4)))  # [n_frames,4,4]

    def number_of_links(self) -> int:
        return int(self.parent_array.shape[0])

    def number_of_joints(self) -> int:
        return self.number_of_links() - 1

    def number_of_collidable_points(self) -> int:
        return int(self.contact_body.shape[0])

    @property
    def indices_of_enabled_collidable_points(self) -> np.ndarray:
        return np.where(self.contact_enabled)[0]

    def tree_depths(self) -> np.ndarray:
        """Depth of every link in the kinematic tree (base = 0)."""
        depth = np.zeros(self.number_of_links(), dtype=int)
        for i in range(1, self.number_of_links()):
            depth[i] = depth[self.parent_array[i]] + 1
        return depth

    @staticmethod
    def build(description: ModelDescription) -> "KinDynParameters":
        links, joints = description.links, description.joints
        nL = len(links)
        if len(joints) != nL - 1:
            raise ValueError("every non-base link must have exactly one parent joint")
        name_to_index = {l.name: l.index for l in links}

        parent = np.full(nL, -1, dtype=np.int64)
        lam_H_pre = np.tile(np.eye(4), (nL, 1, 1))
        suc_H_i = np.tile(np.eye(4), (nL, 1, 1))
        suc_H_i[0] = links[0].pose
        S = np.zeros((nL, 6))
        jtypes = np.zeros(nL - 1, dtype=np.int64)
        axes = np.zeros((nL - 1, 3))
        for j in joints:
            i = j.index
            if i != name_to_index[j.child]:
                raise ValueError("joint index must equal its child link index")
            parent[i] = name_to_index[j.parent]
            if parent[i] >= i:
                raise ValueError("BFS indexing violated: parent index must be lower than child index")
            lam_H_pre[i] = j.pose
            jtypes[i - 1] = j.jtype
            axes[i - 1] = j.axis
            if j.jtype == REVOLUTE:
                S[i, 3:] = j.axis
            elif j.jtype == PRISMATIC:
                S[i, :3] = j.axis
            elif j.jtype == FIXED:
                raise ValueError("fixed joints must have been lumped by the parser")

        # Link parameters: store (m, com, I_CoM) and rebuild the 6x6 the way the reference does
        # (``LinkParameters.build_from_spatial_inertia`` -> ``Inertia.to_sixd``,
        # ``kin_dyn_parameters.py:600-624``, ``math/inertia.py:14-41``).
        mass = np.zeros(nL)
        com = np.zeros((nL, 3))
        I_com = np.zeros((nL, 3, 3))
        M6 = np.zeros((nL, 6, 6))
        for l in links:
            m, c, I = hm.inertia_to_params(l.inertia)
            I = 0.5 * (I + I.T)
            mass[l.index], com[l.index], I_com[l.index] = m, c, I
            M6[l.index] = hm.inertia_to_sixd(m, c, I)

        n = nL - 1
        pts = description.collidable_points
        return KinDynParameters(
            link_names=tuple(l.name for l in links),
            joint_names=tuple(j.name for j in joints),
            frame_names=tuple(f.name for f in description.frames),
            frame_body=np.array([name_to_index[f.parent_name] for f in description.frames], dtype=np.int64),
            frame_transform=np.array([f.pose for f in description.frames], dtype=float).reshape(-1, 4, 4),
            parent_array=parent,
            joint_types=jtypes,
            joint_axis=axes,
            motion_subspaces=S,
            lambda_H_pre=lam_H_pre,
            suc_H_i=suc_H_i,
            link_spatial_inertia=M6,
            link_mass=mass,
            link_com=com,
            link_inertia_com=I_com,
            friction_static=np.array([j.friction_static for j in joints], dtype=float).reshape(n),
            friction_viscous=np.array([j.friction_viscous for j in joints], dtype=float).reshape(n),
            position_limits_min=np.array([j.position_limit[0] for j in joints], dtype=float).reshape(n),
            position_limits_max=np.array([j.position_limit[1] for j in joints], dtype=float).reshape(n),
            position_limit_spring=np.array([j.position_limit_spring for j in joints], dtype=float).reshape(n),
            position_limit_damper=np.array([j.position_limit_damper for j in joints], dtype=float).reshape(n),
            contact_body=np.array([name_to_index[p.parent_link] for p in pts], dtype=np.int64),
            contact_point=np.array([p.position for p in pts], dtype=float).reshape(len(pts), 3),
            contact_enabled=np.array([p.enabled for p in pts], dtype=bool),
        )
