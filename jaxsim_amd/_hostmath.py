"""Small NumPy SE(3)/spatial helpers used by the *host-side* model compiler.

These are build-time helpers only (URDF -> constant tables, state upload
conversions).  Nothing here runs inside ``step``: the hot path lives in
``csrc/`` as HIP kernels.

Conventions follow the reference (``src/jaxsim/math``): 6D vectors are
``[linear; angular]``, velocity adjoint ``X = [[R, S(p) R], [0, R]]``
(``src/jaxsim/math/adjoint.py:92-107``), quaternions are ``wxyz``.
"""

from __future__ import annotations

import numpy as np


def skew(v: np.ndarray) -> np.ndarray:
    """``S(v)`` such that ``S(v) @ w = v x w`` (``src/jaxsim/math/skew.py:12-40``)."""
    x, y, z = np.asarray(v, dtype=float).reshape(3)
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def rpy_to_rotation(rpy) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw -> rotation matrix ``Rz(y) Ry(p) Rx(r)``."""
    r, p, y = (float(a) for a in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def transform(R: np.ndarray, p) -> np.ndarray:
    H = np.eye(4)
    H[:3, :3] = R
    H[:3, 3] = np.asarray(p, dtype=float).reshape(3)
    return H


def transform_from_xyz_rpy(xyz, rpy) -> np.ndarray:
    return transform(rpy_to_rotation(rpy), xyz)


def transform_inverse(H: np.ndarray) -> np.ndarray:
    R, p = H[:3, :3], H[:3, 3]
    return transform(R.T, -R.T @ p)


def quaternion_to_rotation(q) -> np.ndarray:
    """Unit-quaternion (wxyz) -> DCM.  Batched over leading axes."""
    q = np.asarray(q, dtype=float)
    w, x, y, z = np.moveaxis(q, -1, 0)
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def rpy_to_quaternion(rpy) -> np.ndarray:
    """Intrinsic X-Y-Z Euler angles -> wxyz quaternion (batched).

    Mirrors ``Rotation.from_euler("XYZ", a)`` used by the reference's random
    state generator (``src/jaxsim/api/data.py:624-631``): R = Rx(a0) Ry(a1) Rz(a2).
    """
    a = np.asarray(rpy, dtype=float)
    hx, hy, hz = a[..., 0] / 2, a[..., 1] / 2, a[..., 2] / 2
    cx, sx, cy, sy, cz, sz = np.cos(hx), np.sin(hx), np.cos(hy), np.sin(hy), np.cos(hz), np.sin(hz)
    # q = qx * qy * qz
    w = cx * cy * cz - sx * sy * sz
    x = sx * cy * cz + cx * sy * sz
    y = cx * sy * cz - sx * cy * sz
    z = cx * cy * sz + sx * sy * cz
    return np.stack([w, x, y, z], axis=-1)


def adjoint(H: np.ndarray, inverse: bool = False) -> np.ndarray:
    """Velocity adjoint of a homogeneous transform (``math/adjoint.py:66-107``)."""
    R, p = H[:3, :3], H[:3, 3]
    X = np.zeros((6, 6))
    if not inverse:
        X[:3, :3] = R
        X[:3, 3:] = skew(p) @ R
        X[3:, 3:] = R
    else:
        X[:3, :3] = R.T
        X[:3, 3:] = -R.T @ skew(p)
        X[3:, 3:] = R.T
    return X


def inertia_to_sixd(mass: float, com, I: np.ndarray) -> np.ndarray:
    """``Inertia.to_sixd`` (``src/jaxsim/math/inertia.py:14-41``)."""
    c = skew(com)
    M = np.zeros((6, 6))
    M[:3, :3] = mass * np.eye(3)
    M[:3, 3:] = mass * c.T
    M[3:, :3] = mass * c
    M[3:, 3:] = np.asarray(I, dtype=float) + mass * c @ c.T
    return M


def inertia_to_params(M: np.ndarray):
    """``Inertia.to_params`` (``src/jaxsim/math/inertia.py:43-63``)."""
    m = np.trace(M[:3, :3]) / 3.0
    mC = M[3:, :3]
    c = np.array([mC[2, 1], mC[0, 2], mC[1, 0]]) / m
    I = M[3:, 3:] - (mC @ mC.T / m)
    return m, c, I


def quaternion_derivative(q: np.ndarray, omega: np.ndarray, K: float = 0.1) -> np.ndarray:
    """``Quaternion.derivative`` with the angular velocity in the inertial frame (``src/jaxsim/math/quaternion.py:68-132``),
    batched: ``Qdot = 1/2 Q_inertial(q) [K |w| (1 - |q|); w]``, ``q`` = wxyz."""
    q = np.asarray(q, dtype=float)
    w = np.asarray(omega, dtype=float)
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    h0 = K * np.linalg.norm(w, axis=-1) * (1.0 - np.linalg.norm(q, axis=-1))
    wx, wy, wz = w[..., 0], w[..., 1], w[..., 2]
    return 0.5 * np.stack(
        [
            qw * h0 - qx * wx - qy * wy - qz * wz,
            qx * h0 + qw * wx + qz * wy - qy * wz,
            qy * h0 - qz * wx + qw * wy + qx * wz,
            qz * h0 + qy * wx - qx * wy + qw * wz,
        ],
        axis=-1,
    )
