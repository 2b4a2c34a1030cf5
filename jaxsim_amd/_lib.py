"""ctypes binding of the C-ABI library (``include/jaxsim_amd.h``).

The product path has no CPU fallback: if ``libjaxsim_amd.so`` cannot be loaded, or no HIP
device answers, every entry point raises ``JaxsimAmdError`` loudly.
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libjaxsim_amd.so"

JXS_F32, JXS_F64 = 0, 1


class JaxsimAmdError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    """``jxs_model_desc`` (include/jaxsim_amd.h)."""

    _fields_ = [
        ("n_links", C.c_int32),
        ("floating_base", C.c_int32),
        ("dtype", C.c_int32),
        ("parent", C.POINTER(C.c_int32)),
        ("joint_type", C.POINTER(C.c_int32)),
        ("joint_axis", C.POINTER(C.c_double)),
        ("lambda_H_pre", C.POINTER(C.c_double)),
        ("suc_H_i", C.POINTER(C.c_double)),
        ("link_mass", C.POINTER(C.c_double)),
        ("link_com", C.POINTER(C.c_double)),
        ("link_inertia", C.POINTER(C.c_double)),
        ("friction_static", C.POINTER(C.c_double)),
        ("friction_viscous", C.POINTER(C.c_double)),
        ("position_limit_min", C.POINTER(C.c_double)),
        ("position_limit_max", C.POINTER(C.c_double)),
        ("position_limit_spring", C.POINTER(C.c_double)),
        ("position_limit_damper", C.POINTER(C.c_double)),
        ("n_points", C.c_int32),
        ("point_body", C.POINTER(C.c_int32)),
        ("point_position", C.POINTER(C.c_double)),
        ("point_enabled", C.POINTER(C.c_uint8)),
        ("time_step", C.c_double),
        ("gravity", C.c_double),
        ("K", C.c_double),
        ("D", C.c_double),
        ("mu", C.c_double),
        ("p", C.c_double),
        ("q", C.c_double),
        ("terrain_height", C.c_double),
        ("torque_max", C.c_double),
        ("omega_th", C.c_double),
        ("omega_max", C.c_double),
        ("enable_friction", C.c_int32),
        ("terrain_normal", C.c_double * 3),
        ("integrator", C.c_int32),
        ("contact_model", C.c_int32),
        ("regularization_delassus", C.c_double),
        ("solver_tol", C.c_double),
        ("rr_time_constant", C.c_double),
        ("rr_damping_coefficient", C.c_double),
        ("rr_d_min", C.c_double),
        ("rr_d_max", C.c_double),
        ("rr_width", C.c_double),
        ("rr_midpoint", C.c_double),
        ("rr_power", C.c_double),
        ("terrain_grid", C.POINTER(C.c_double)),
        ("terrain_nx", C.c_int32),
        ("terrain_ny", C.c_int32),
        ("terrain_origin", C.c_double * 2),
        ("terrain_spacing", C.c_double * 2),
        ("terrain_delta", C.c_double),
    ]


class Layout(C.Structure):
    """``jxs_layout`` (include/jaxsim_amd.h)."""

    _fields_ = [(n, C.c_int32) for n in (
        "n_links", "n_joints", "n_points", "n_rows", "row_pos", "row_quat", "row_s", "row_vlin",
        "row_vang", "row_sd", "row_m", "group", "tile", "dtype", "row_mode")]  # fmt: skip


def dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt == np.float32:
        return JXS_F32
    if dt == np.float64:
        return JXS_F64
    raise ValueError(f"unsupported dtype {dt}; use float32 or float64")


def make_desc(model, dtype) -> tuple[ModelDesc, list]:
    """Flatten a host ``JaxSimModel`` into a ``jxs_model_desc``.

    Returns the struct and the list of NumPy arrays that back its pointers (keep it alive).
    """
    kdp = model.kin_dyn_parameters
    nL = kdp.number_of_links()
    keep = []

    def dptr(a, shape):
        arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
        keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(C.c_double))

    def iptr(a, n):
        arr = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(n))
        keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(C.c_int32))

    def per_joint(a, fill=0.0):
        return np.concatenate([[fill], np.asarray(a, dtype=np.float64).reshape(nL - 1)])

    fmax = np.finfo(np.float64).max
    d = ModelDesc()
    d.n_links = nL
    d.floating_base = int(model.floating_base())
    d.dtype = dtype_code(dtype)
    d.parent = iptr(kdp.parent_array, nL)
    d.joint_type = iptr(np.concatenate([[0], kdp.joint_types]), nL)
    d.joint_axis = dptr(np.vstack([np.zeros((1, 3)), kdp.joint_axis.reshape(nL - 1, 3)]), (nL, 3))
    d.lambda_H_pre = dptr(kdp.lambda_H_pre, (nL, 16))
    d.suc_H_i = dptr(kdp.suc_H_i, (nL, 16))
    d.link_mass = dptr(kdp.link_mass, (nL,))
    d.link_com = dptr(kdp.link_com, (nL, 3))
    d.link_inertia = dptr(kdp.link_inertia_com, (nL, 9))
    d.friction_static = dptr(per_joint(kdp.friction_static), (nL,))
    d.friction_viscous = dptr(per_joint(kdp.friction_viscous), (nL,))
    d.position_limit_min = dptr(per_joint(kdp.position_limits_min, -fmax), (nL,))
    d.position_limit_max = dptr(per_joint(kdp.position_limits_max, fmax), (nL,))
    d.position_limit_spring = dptr(per_joint(kdp.position_limit_spring), (nL,))
    d.position_limit_damper = dptr(per_joint(kdp.position_limit_damper), (nL,))
    n_cp = kdp.number_of_collidable_points()
    d.n_points = n_cp
    d.point_body = iptr(kdp.contact_body if n_cp else np.zeros(1), max(n_cp, 1))
    d.point_position = dptr(kdp.contact_point if n_cp else np.zeros((1, 3)), (max(n_cp, 1), 3))
    en = np.ascontiguousarray(np.asarray(kdp.contact_enabled if n_cp else np.zeros(1), dtype=np.uint8))
    keep.append(en)
    d.point_enabled = en.ctypes.data_as(C.POINTER(C.c_uint8))
    d.time_step = float(model.time_step)
    d.gravity = float(model.gravity)
    cp = model.contact_params
    d.K, d.D, d.mu = float(cp.K), float(cp.D), float(cp.mu)
    d.p, d.q = float(getattr(cp, "p", 0.5)), float(getattr(cp, "q", 0.5))  # RigidContactsParams has no exponents
    d.terrain_height = float(model.terrain._height)
    grid = getattr(model.terrain, "_heights", None)
    if grid is not None:  # HeightFieldTerrain: heights[ix, iy], x outer
        d.terrain_grid = dptr(grid, (grid.size,))
        d.terrain_nx, d.terrain_ny = int(grid.shape[0]), int(grid.shape[1])
        d.terrain_origin = (C.c_double * 2)(*[float(v) for v in model.terrain._origin])
        d.terrain_spacing = (C.c_double * 2)(*[float(v) for v in model.terrain._spacing])
        d.terrain_delta = float(model.terrain.delta)
    ap = model.actuation_params
    d.torque_max, d.omega_th, d.omega_max = float(ap.torque_max), float(ap.omega_th), float(ap.omega_max)
    d.enable_friction = int(bool(ap.enable_friction))
    nrm = getattr(model.terrain, "_normal", (0.0, 0.0, 1.0))
    d.terrain_normal = (C.c_double * 3)(*[float(x) for x in nrm])
    d.integrator = int(model.integrator)
    cm = model.contact_model
    d.contact_model = {"RigidContacts": 1, "RelaxedRigidContacts": 2}.get(type(cm).__name__, 0)
    if d.contact_model == 2:
        if not cp.valid():
            raise ValueError("invalid RelaxedRigidContactsParams")
        d.rr_time_constant, d.rr_damping_coefficient = float(cp.time_constant), float(cp.damping_coefficient)
        d.rr_d_min, d.rr_d_max, d.rr_width = float(cp.d_min), float(cp.d_max), float(cp.width)
        d.rr_midpoint, d.rr_power = float(cp.midpoint), float(cp.power)
    d.regularization_delassus = float(getattr(cm, "regularization_delassus", 1e-6))
    d.solver_tol = float(getattr(cm, "solver_tol", 1e-3))
    return d, keep


def model_signature(model, dtype) -> tuple:
    """Everything the device copy depends on; a change rebuilds the device tables."""
    kdp = model.kin_dyn_parameters
    cp, ap = model.contact_params, model.actuation_params
    return (
        id(kdp), np.dtype(dtype).str, model.time_step, model.gravity, model.floating_base(),
        cp.K, cp.D, cp.mu, getattr(cp, "p", 0.5), getattr(cp, "q", 0.5), model.terrain._height, tuple(getattr(model.terrain, "_normal", (0.0, 0.0, 1.0))),
        ap.torque_max, ap.omega_th, ap.omega_max, ap.enable_friction, int(model.integrator),
        type(model.contact_model).__name__, getattr(model.contact_model, "regularization_delassus", None),
        getattr(model.contact_model, "solver_tol", None),
        tuple(getattr(cp, k, None) for k in ("time_constant", "damping_coefficient", "d_min", "d_max", "width", "midpoint", "power")),
        # height-field terrain: the (read-only) sample array by identity, its placement by value
        (id(getattr(model.terrain, "_heights", None)), getattr(model.terrain, "_origin", None), getattr(model.terrain, "_spacing", None),
         getattr(model.terrain, "delta", None)) if getattr(model.terrain, "_heights", None) is not None else None,
    )  # fmt: skip


_lib = None


def _declare(lib):
    vp, cp_ = C.c_void_p, C.c_char_p
    lib.jxs_last_error.restype = cp_
    lib.jxs_last_error.argtypes = []
    sig = {
        "jxs_device_count": [C.POINTER(C.c_int)],
        "jxs_set_device": [C.c_int],
        "jxs_malloc": [C.POINTER(vp), C.c_uint64],
        "jxs_free": [vp],
        "jxs_memcpy_h2d": [vp, vp, C.c_uint64, vp],
        "jxs_memcpy_d2h": [vp, vp, C.c_uint64, vp],
        "jxs_memcpy_d2d": [vp, vp, C.c_uint64, vp],
        "jxs_memset": [vp, C.c_int, C.c_uint64, vp],
        "jxs_stream_create": [C.POINTER(vp)],
        "jxs_stream_destroy": [vp],
        "jxs_stream_synchronize": [vp],
        "jxs_stream_wait_spin": [vp],
        "jxs_device_synchronize": [],
        "jxs_event_create": [C.POINTER(vp)],
        "jxs_event_destroy": [vp],
        "jxs_event_record": [vp, vp],
        "jxs_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
        "jxs_model_create": [C.POINTER(ModelDesc), C.POINTER(vp)],
        "jxs_kernel_spec": [C.POINTER(ModelDesc), C.c_int, C.c_char_p, C.c_int],
        "jxs_model_attach_specialized": [vp, C.c_int, C.c_char_p],
        "jxs_model_specialized_modes": [vp, C.POINTER(C.c_uint)],
        "jxs_model_destroy": [vp],
        "jxs_model_layout": [vp, C.POINTER(Layout)],
        "jxs_step": [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp],
        "jxs_step_gravity_compensated": [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp],
        "jxs_rollout": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp],
        "jxs_rollout_controlled": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp],
        "jxs_rollout_recorded": [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp],
        "jxs_forward_dynamics_aba": [vp, vp, vp, vp, C.c_int, vp, C.c_int, vp],
        "jxs_inverse_dynamics": [vp, vp, vp, vp, C.c_int, vp, C.c_int, vp],
        "jxs_system_dynamics": [vp, vp, vp, vp, C.c_int, C.c_double, vp, vp, C.c_int, vp],
        "jxs_link_contact_forces": [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp],
        "jxs_gravity_torques": [vp, vp, vp, C.c_int, vp],
        "jxs_tile_from_env_major": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp],
        "jxs_tile_to_env_major": [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp],
        "jxs_validate_state": [vp, vp, C.c_int, C.POINTER(C.c_int), vp],
        "jxs_step_repeat": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp],
        "jxs_step_repeat_timed": [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double)],
        "jxs_refresh_kinematics": [vp, vp, vp, vp, C.c_int, vp],
        "jxs_mass_matrix": [vp, vp, vp, C.c_int, vp],
        "jxs_solver_fault_counts": [vp, C.POINTER(C.c_int), C.c_int, vp],
        "jxs_debug_reload_env": [],
        "jxs_jacobian_full": [vp, vp, vp, vp, C.c_int, vp],
        "jxs_mass_matrix_inverse": [vp, vp, vp, C.c_int, vp],
        "jxs_comm_unique_id": [C.c_char * 128],
        "jxs_comm_version": [C.POINTER(C.c_int)],
        "jxs_device_pci_bus_id": [C.c_char_p, C.c_int],
        "jxs_comm_init": [C.POINTER(vp), C.c_char * 128, C.c_int, C.c_int],
        "jxs_comm_destroy": [vp],
        "jxs_allgather": [vp, vp, vp, C.c_uint64, C.c_int, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    return sig


EXPORTED_SYMBOLS = None


def load(path: os.PathLike | None = None):
    """Load ``libjaxsim_amd.so`` (built by ``__graft_entry__.build()`` / ``csrc/build.sh``)."""
    global _lib, EXPORTED_SYMBOLS
    if _lib is not None and path is None:
        return _lib
    override = os.environ.get("JAXSIM_AMD_LIB")  # developer knob: A/B-test another build of the library
    p = pathlib.Path(path) if path is not None else (pathlib.Path(override) if override else LIB_PATH)
    if not p.exists():
        raise JaxsimAmdError(
            f"HIP extension not built: {p} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or jaxsim_amd/csrc/build.sh). There is no CPU fallback."
        )
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # missing ROCm runtime, wrong arch, ...
        raise JaxsimAmdError(f"cannot load {p}: {e}") from e
    EXPORTED_SYMBOLS = tuple(_declare(lib).keys()) + ("jxs_last_error",)
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().jxs_last_error()
        raise JaxsimAmdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
