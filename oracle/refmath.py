"""Oracle math layer: batched NumPy restatement of ``src/jaxsim/math/*``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function broadcasts
over leading batch axes.  6D vectors are ``[linear; angular]``.
"""

from __future__ import annotations

import numpy as np

STANDARD_GRAVITY = 9.81  # src/jaxsim/math/__init__.py:14


def safe_norm(a: np.ndarray, axis=-1, keepdims=False) -> np.ndarray:
    """``safe_norm`` (``src/jaxsim/math/utils.py:8-58``): norm with 0 for zero/NaN-free input."""
    return np.sqrt(np.sum(a * a, axis=axis, keepdims=keepdims))


def wedge(v: np.ndarray) -> np.ndarray:
    """``Skew.wedge`` (``src/jaxsim/math/skew.py:12-40``)."""
    v = np.asarray(v)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    O = np.zeros_like(x)
    return np.stack(
        [np.stack([O, -z, y], -1), np.stack([z, O, -x], -1), np.stack([-y, x, O], -1)], -2
    )


def vee(m: np.ndarray) -> np.ndarray:
    """``Skew.vee`` (``src/jaxsim/math/skew.py:42-58``)."""
    return np.stack([m[..., 2, 1], m[..., 0, 2], m[..., 1, 0]], -1)


def so3_from_quaternion(q: np.ndarray) -> np.ndarray:
    """``jaxlie.SO3(wxyz=q).as_matrix()`` -- DCM of ``q/|q|``.

    jaxlie scales by ``sqrt(2/|q|^2)`` before forming the outer product, i.e. it
    implicitly normalises; restated here with the same scaling.
    """
    q = np.asarray(q)
    nsq = np.sum(q * q, axis=-1, keepdims=True)
    s = q * np.sqrt(2.0 / nsq).astype(q.dtype)
    w, x, y, z = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    one = np.ones_like(w)
    R = np.stack(
        [
            np.stack([one - y * y - z * z, x * y - z * w, x * z + y * w], -1),
            np.stack([x * y + z * w, one - x * x - z * z, y * z - x * w], -1),
            np.stack([x * z - y * w, y * z + x * w, one - x * x - y * y], -1),
        ],
        -2,
    )
    return R


def transform_from_rotation_translation(R: np.ndarray, p: np.ndarray) -> np.ndarray:
    H = np.zeros(R.shape[:-2] + (4, 4), dtype=R.dtype)
    H[..., :3, :3] = R
    H[..., :3, 3] = p
    H[..., 3, 3] = 1
    return H


def transform_from_quaternion_translation(q: np.ndarray, p: np.ndarray) -> np.ndarray:
    """``Transform.from_quaternion_and_translation`` (``src/jaxsim/math/transform.py:14-50``)."""
    return transform_from_rotation_translation(so3_from_quaternion(q), p)


def transform_inverse(H: np.ndarray) -> np.ndarray:
    """``Transform.inverse`` (``src/jaxsim/math/transform.py:97-126``)."""
    R, p = H[..., :3, :3], H[..., :3, 3]
    Rt = np.swapaxes(R, -1, -2)
    return transform_from_rotation_translation(Rt, -np.einsum("...ij,...j->...i", Rt, p))


def adjoint_from_rotation_translation(R: np.ndarray, p: np.ndarray, inverse: bool = False) -> np.ndarray:
    """``Adjoint.from_rotation_and_translation`` (``src/jaxsim/math/adjoint.py:66-107``)."""
    X = np.zeros(R.shape[:-2] + (6, 6), dtype=R.dtype)
    if not inverse:
        X[..., :3, :3] = R
        X[..., :3, 3:] = wedge(p) @ R
        X[..., 3:, 3:] = R
    else:
        Rt = np.swapaxes(R, -1, -2)
        X[..., :3, :3] = Rt
        X[..., :3, 3:] = -Rt @ wedge(p)
        X[..., 3:, 3:] = Rt
    return X


def adjoint_from_transform(H: np.ndarray, inverse: bool = False) -> np.ndarray:
    """``Adjoint.from_transform`` (``src/jaxsim/math/adjoint.py:46-64``).

    The reference round-trips through ``jaxlie.SE3.from_matrix`` (matrix ->
    quaternion -> matrix), a no-op up to rounding for a valid rotation.
    """
    return adjoint_from_rotation_translation(H[..., :3, :3], H[..., :3, 3], inverse=inverse)


def adjoint_inverse(X: np.ndarray) -> np.ndarray:
    """``Adjoint.inverse`` (``src/jaxsim/math/adjoint.py:135-160``)."""
    Rt = np.swapaxes(X[..., :3, :3], -1, -2)
    T = X[..., :3, 3:]
    out = np.zeros_like(X)
    out[..., :3, :3] = Rt
    out[..., :3, 3:] = -Rt @ T @ Rt
    out[..., 3:, 3:] = Rt
    return out


def adjoint_to_transform(X: np.ndarray) -> np.ndarray:
    """``Adjoint.to_transform`` (``src/jaxsim/math/adjoint.py:109-133``): ``p = vee(X12 R^T)``."""
    R = X[..., :3, :3]
    oxR = X[..., :3, 3:]
    return transform_from_rotation_translation(R, vee(oxR @ np.swapaxes(R, -1, -2)))


def vx(v6: np.ndarray) -> np.ndarray:
    """``Cross.vx`` (``src/jaxsim/math/cross.py:14-43``): ``[[S(w), S(v)],[0, S(w)]]``."""
    v, w = v6[..., :3], v6[..., 3:]
    X = np.zeros(v6.shape[:-1] + (6, 6), dtype=v6.dtype)
    X[..., :3, :3] = wedge(w)
    X[..., :3, 3:] = wedge(v)
    X[..., 3:, 3:] = wedge(w)
    return X


def vx_star(v6: np.ndarray) -> np.ndarray:
    """``Cross.vx_star`` (``src/jaxsim/math/cross.py:45-58``): ``-vx^T``."""
    return -np.swapaxes(vx(v6), -1, -2)


def inertia_to_sixd(mass: np.ndarray, com: np.ndarray, I: np.ndarray) -> np.ndarray:
    """``Inertia.to_sixd`` (``src/jaxsim/math/inertia.py:14-41``)."""
    c = wedge(com)
    ct = np.swapaxes(c, -1, -2)
    m = np.asarray(mass)[..., None, None]
    M = np.zeros(c.shape[:-2] + (6, 6), dtype=c.dtype)
    M[..., :3, :3] = m * np.eye(3, dtype=c.dtype)
    M[..., :3, 3:] = m * ct
    M[..., 3:, :3] = m * c
    M[..., 3:, 3:] = I + m * (c @ ct)
    return M


def rotation_from_axis_angle(vector: np.ndarray) -> np.ndarray:
    """``Rotation.from_axis_angle`` (``src/jaxsim/math/rotation.py:58-84``).

    ``R = (c I - s S(u) + c1 u u^T)^T`` with ``c1 = 2 sin^2(theta/2)``; ``theta == 0``
    divides by 1, giving ``u = 0`` and ``R = I``.
    """
    theta = safe_norm(vector)
    s, c = np.sin(theta), np.cos(theta)
    c1 = 2 * np.sin(theta / 2.0) ** 2
    safe_theta = np.where(theta == 0, np.ones_like(theta), theta)
    u = vector / safe_theta[..., None]
    eye = np.eye(3, dtype=vector.dtype)
    R = (
        c[..., None, None] * eye
        - s[..., None, None] * wedge(u)
        + c1[..., None, None] * (u[..., :, None] * u[..., None, :])
    )
    return np.swapaxes(R, -1, -2)


def quaternion_derivative(q: np.ndarray, omega: np.ndarray, omega_in_body_fixed: bool = False, K: float = 0.1):
    """``Quaternion.derivative`` (``src/jaxsim/math/quaternion.py:68-132``)."""
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    if omega_in_body_fixed:
        Q = np.stack(
            [
                np.stack([qw, -qx, -qy, -qz], -1),
                np.stack([qx, qw, -qz, qy], -1),
                np.stack([qy, qz, qw, -qx], -1),
                np.stack([qz, -qy, qx, qw], -1),
            ],
            -2,
        )
    else:
        Q = np.stack(
            [
                np.stack([qw, -qx, -qy, -qz], -1),
                np.stack([qx, qw, qz, -qy], -1),
                np.stack([qy, -qz, qw, qx], -1),
                np.stack([qz, qy, -qx, qw], -1),
            ],
            -2,
        )
    norm_w = safe_norm(omega)
    head = (K * norm_w * (1 - safe_norm(q))).astype(q.dtype)
    vec = np.concatenate([head[..., None], omega], axis=-1)
    return (0.5 * np.einsum("...ij,...j->...i", Q, vec)).astype(q.dtype)


def mv(A: np.ndarray, x: np.ndarray) -> np.ndarray:
    return np.einsum("...ij,...j->...i", A, x)


def quaternion_from_euler_xyz(angles: np.ndarray) -> np.ndarray:
    """Intrinsic X-Y-Z Euler angles -> wxyz quaternion.

    What ``jax.scipy...Rotation.from_euler("XYZ", a)`` computes for the reference's random
    state generator (``src/jaxsim/api/data.py:624-631``): ``R = Rx(a0) Ry(a1) Rz(a2)``.
    """
    a = np.asarray(angles, dtype=float)
    hx, hy, hz = a[..., 0] / 2, a[..., 1] / 2, a[..., 2] / 2
    cx, sx, cy, sy, cz, sz = np.cos(hx), np.sin(hx), np.cos(hy), np.sin(hy), np.cos(hz), np.sin(hz)
    w = cx * cy * cz - sx * sy * sz
    x = sx * cy * cz + cx * sy * sz
    y = cx * sy * cz - sx * cy * sz
    z = cx * cy * sz + sx * sy * cz
    return np.stack([w, x, y, z], axis=-1)
