"""State-block layout: the 7 arrays of ``JaxSimModelData`` as one ``[rows][N]`` SoA block.

Rows (reference fields ``src/jaxsim/api/data.py:46-63``)::

    base_position[3] | base_quaternion[4] wxyz | joint_positions[n] |
    base_linear_velocity[3] (inertial-fixed) | base_angular_velocity[3] | joint_velocities[n] |
    tangential_deformation[n_cp][3]

The batch index is the fastest axis so that consecutive environments are consecutive in
HBM.  The reference's ``vmap`` layout has N slowest (``[N, n]``, ...); the transposition
happens here, at upload/download time, outside the timed loop.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass(frozen=True)
class StateLayout:
    n_links: int
    n_joints: int
    n_points: int

    @property
    def row_pos(self) -> int:
        return 0

    @property
    def row_quat(self) -> int:
        return 3

    @property
    def row_s(self) -> int:
        return 7

    @property
    def row_vlin(self) -> int:
        return 7 + self.n_joints

    @property
    def row_vang(self) -> int:
        return 10 + self.n_joints

    @property
    def row_sd(self) -> int:
        return 13 + self.n_joints

    @property
    def row_m(self) -> int:
        return 13 + 2 * self.n_joints

    @property
    def n_rows(self) -> int:
        return 13 + 2 * self.n_joints + 3 * self.n_points

    @staticmethod
    def of(model) -> "StateLayout":
        kdp = model.kin_dyn_parameters
        return StateLayout(kdp.number_of_links(), kdp.number_of_joints(), kdp.number_of_collidable_points())


def pack_state(
    layout: StateLayout,
    *,
    base_position,
    base_quaternion,
    joint_positions,
    base_linear_velocity,
    base_angular_velocity,
    joint_velocities,
    tangential_deformation,
    dtype,
) -> np.ndarray:
    """``[N, ...]`` host arrays (reference layout) -> ``[rows, N]`` block."""
    N = np.shape(base_position)[0]
    L = layout
    out = np.zeros((L.n_rows, N), dtype=dtype)
    out[L.row_pos : L.row_pos + 3] = np.asarray(base_position).T
    out[L.row_quat : L.row_quat + 4] = np.asarray(base_quaternion).T
    out[L.row_s : L.row_s + L.n_joints] = np.asarray(joint_positions).reshape(N, L.n_joints).T
    out[L.row_vlin : L.row_vlin + 3] = np.asarray(base_linear_velocity).T
    out[L.row_vang : L.row_vang + 3] = np.asarray(base_angular_velocity).T
    out[L.row_sd : L.row_sd + L.n_joints] = np.asarray(joint_velocities).reshape(N, L.n_joints).T
    if L.n_points:
        out[L.row_m :] = np.asarray(tangential_deformation).reshape(N, 3 * L.n_points).T
    return out


def unpack_state(layout: StateLayout, block: np.ndarray) -> dict[str, np.ndarray]:
    """``[rows, N]`` block -> dict of ``[N, ...]`` arrays (reference layout)."""
    L = layout
    N = block.shape[1]
    return dict(
        base_position=block[L.row_pos : L.row_pos + 3].T.copy(),
        base_quaternion=block[L.row_quat : L.row_quat + 4].T.copy(),
        joint_positions=block[L.row_s : L.row_s + L.n_joints].T.copy(),
        base_linear_velocity=block[L.row_vlin : L.row_vlin + 3].T.copy(),
        base_angular_velocity=block[L.row_vang : L.row_vang + 3].T.copy(),
        joint_velocities=block[L.row_sd : L.row_sd + L.n_joints].T.copy(),
        tangential_deformation=block[L.row_m :].T.reshape(N, L.n_points, 3).copy(),
    )


def tile_block(block: np.ndarray, tile: int) -> np.ndarray:
    """Host ``[rows, N]`` array -> the device storage order ``[ceil(N/T), rows, T]`` (flat).

    ``T`` environments (those one wavefront processes, ``jxs_layout.tile``) are interleaved per row
    so that a wave's rows are contiguous in HBM; the last tile is zero-padded."""
    rows, N = block.shape
    nt = -(-N // tile)
    out = np.zeros((nt, rows, tile), dtype=block.dtype)
    pad = np.zeros((rows, nt * tile), dtype=block.dtype)
    pad[:, :N] = block
    out[:] = pad.reshape(rows, nt, tile).transpose(1, 0, 2)
    return out.reshape(-1)


def untile_block(flat: np.ndarray, rows: int, N: int, tile: int) -> np.ndarray:
    """Inverse of :func:`tile_block`."""
    nt = -(-N // tile)
    return np.ascontiguousarray(flat.reshape(nt, rows, tile).transpose(1, 0, 2).reshape(rows, nt * tile)[:, :N])
