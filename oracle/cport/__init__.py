"""ctypes binding of the oracle's C port (TEST INFRASTRUCTURE / CPU baseline only)."""

from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SO = _HERE / "liboracle_cport.so"
_lib = None


def build(force: bool = False) -> pathlib.Path:
    src = _HERE / "step_ref.c"
    if force or not _SO.exists() or src.stat().st_mtime > _SO.stat().st_mtime:
        subprocess.run(["bash", str(_HERE / "build.sh")], check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        from jaxsim_amd._lib import ModelDesc  # the C struct of include/jaxsim_amd.h

        _lib = C.CDLL(str(build()))
        for name in ("oracle_step_f64", "oracle_step_f32"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return _lib


def step(model, state: np.ndarray, *, tau=None, link_forces_inertial=None, n_steps: int = 1, n_threads: int = 1):
    """``n_steps`` reference-style steps on a host ``[rows, N]`` block; returns the new block."""
    from jaxsim_amd._lib import make_desc

    dtype = state.dtype
    desc, keep = make_desc(model, dtype)
    fn = lib().oracle_step_f64 if dtype == np.float64 else lib().oracle_step_f32
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    state, tau, f = c(state), c(tau), c(link_forces_inertial)
    out = np.empty_like(state)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = fn(C.byref(desc), p(state), p(out), p(tau), p(f), state.shape[1], int(n_steps), int(n_threads))
    if rc != 0:
        raise RuntimeError("oracle C port: unsupported model")
    return out
