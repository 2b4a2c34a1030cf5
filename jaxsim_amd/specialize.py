"""Model-specialised kernels (include/jaxsim_amd.h, section "model-specialised kernels").

The reference compiles ``js.model.step`` per model: the kinematic tree is static under ``jax.jit``
(src/jaxsim/api/model.py:36-120) and XLA folds it into the program.  The generic kernels of
``libjaxsim_amd.so`` read the same facts as wave-uniform flags at run time; here the SAME hand-written kernel
source (jaxsim_amd/csrc/jxs_spec.hip -> jxs_kernels.h -> jxs_core.h) is compiled once per
(dtype, lanes per environment, mode, integer model flags) with those flags as constants.  Physical parameters
stay run-time data: changing masses, gains, the time step or the contact parameters never needs a rebuild.

* ``spec(model, dtype, mode)``      canonical description (text) of what a kernel would be specialised on;
* ``cached(...)`` / ``compile(...)``  the shared object in ``jaxsim_amd/csrc/spec_cache`` (file name = hash of
  the description, the kernel sources and the compiler flags), built with hipcc (gfx950; ~20 s);
* ``attach(device_model, model)``   route the launches of the mode through it (checked by the library).

``runtime.device_model`` attaches a cached object when there is one and, when hipcc is present, **builds the
missing ones on first use** (the ``jax.jit`` experience: the first step of a new model compiles, ~20 s; later
processes find the object in the cache).  ``JAXSIM_AMD_SPECIALIZE=cached`` never compiles, ``=0`` disables the
lookup, ``=1`` insists on building.  ``__graft_entry__.build()`` pre-builds the objects of the benchmark
configurations.  The test-suite pins ``cached`` (tests/conftest.py) so that both the specialised and the
library's own kernels stay covered and no test waits for a compiler.
"""
from __future__ import annotations

import ctypes as C
import functools
import hashlib
import os
import pathlib
import subprocess

import numpy as np

from . import _lib

MODE_STEP = 0
MODE_STEP_RIGID = 6
_CSRC = pathlib.Path(__file__).resolve().parent / "csrc"
CACHE = _CSRC / "spec_cache"
_HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fno-slp-vectorize",
          "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wno-cuda-compat", "-Wno-pass-failed", "-Djxs_launch=jxs_launch_spec"]  # fmt: skip


def _src_dir() -> pathlib.Path:
    """The directory the kernel sources are compiled from: csrc/, or the developer override (tools/gpu/*.sh A/B runs)."""
    return pathlib.Path(os.environ.get("JAXSIM_AMD_SPEC_CSRC") or _CSRC)


@functools.lru_cache(maxsize=4)
def _source_sha_of(directory: str) -> str:
    h = hashlib.sha256()
    d = pathlib.Path(directory)
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".inc")) or name == "jxs_spec.hip":
            h.update(name.encode() + b"\0" + (d / name).read_bytes())
    h.update((pathlib.Path(__file__).resolve().parent.parent / "include" / "jaxsim_amd.h").read_bytes())
    return h.hexdigest()[:16]


def source_sha() -> str:
    """Hash of the kernel sources a specialised object is compiled from -- of the directory that is actually
    compiled (once per process and directory)."""
    return _source_sha_of(str(_src_dir()))


MODE_ROLLOUT, MODE_STEP_RK4, MODE_STEP_RK4_RIGID = 4, 5, 7


def modes_of(model) -> list[int]:
    """The kernel modes ``js.model.step`` / ``js.model.rollout`` launch for this model (csrc/jxs_params.h Mode)."""
    desc, _keep = _lib.make_desc(model, np.float32)
    enabled = desc.n_points > 0 and any(desc.point_enabled[k] for k in range(desc.n_points))
    rigid = desc.contact_model != 0 and enabled
    if desc.integrator != 0:  # Runge-Kutta: one launch per step, four dynamics evaluations inside
        return [MODE_STEP_RK4_RIGID if rigid else MODE_STEP_RK4]
    return [MODE_STEP_RIGID] if rigid else [MODE_STEP, MODE_ROLLOUT]


MODE_DYN, MODE_DYN_RIGID = 12, 13


def dyn_mode_of(model) -> int:
    """The kernel mode behind ``js.ode.system_dynamics`` / ``js.contact.link_contact_forces`` for this model [round 6]."""
    desc, _keep = _lib.make_desc(model, np.float32)
    enabled = desc.n_points > 0 and any(desc.point_enabled[k] for k in range(desc.n_points))
    return MODE_DYN_RIGID if (desc.contact_model != 0 and enabled) else MODE_DYN


def mode_of(model) -> int:
    """The mode of a single ``js.model.step``."""
    return modes_of(model)[0]


def spec(model, dtype, mode: int = MODE_STEP) -> str:
    lib = _lib.load()
    desc, _keep = _lib.make_desc(model, dtype)
    buf = C.create_string_buffer(4096)
    n = lib.jxs_kernel_spec(C.byref(desc), mode, buf, len(buf))
    if n < 0:
        _lib.check(n, "jxs_kernel_spec")
    return buf.value.decode()


def _flags() -> list[str]:
    return _FLAGS + os.environ.get("JAXSIM_AMD_SPEC_EXTRA_FLAGS", "").split()  # developer aid, e.g. -DJXS_PHASE_TIMING


def path_of(text: str) -> pathlib.Path:
    key = hashlib.sha256((text + "|" + source_sha() + "|" + " ".join(_flags())).encode()).hexdigest()[:20]
    return CACHE / f"libjxs_spec_{key}.so"


def _record(text: str) -> None:
    """``JAXSIM_AMD_SPEC_RECORD=<file>``: append the description of every kernel a process asks for (hit or miss).
    tests/spec_manifest.txt is such a record of the GPU suite; ``__graft_entry__.build()`` pre-builds it."""
    path = os.environ.get("JAXSIM_AMD_SPEC_RECORD")
    if path:
        with open(path, "a") as f:
            f.write(text + "\n")


def cached(model, dtype, mode: int = MODE_STEP) -> pathlib.Path | None:
    text = spec(model, dtype, mode)
    _record(text)
    p = path_of(text)
    return p if p.exists() else None


def compile(model, dtype, mode: int = MODE_STEP, *, force: bool = False) -> pathlib.Path:  # noqa: A001
    """Build (or find) the specialised kernel object; needs hipcc, not a GPU."""
    return compile_text(spec(model, dtype, mode), force=force)


def compile_text(text: str, *, force: bool = False) -> pathlib.Path:
    """``compile`` from the description alone (``spec``): everything the build needs is in the text."""
    out = path_of(text)
    if out.exists() and not force:
        return out
    head, assign = text.rsplit(";", 1)
    fields = dict(kv.split("=") for kv in head.split(";"))
    CACHE.mkdir(exist_ok=True)
    # one builder per object: the ranks of a multi-process run (and the threads of attach()) that miss the same object
    # queue on its lock file; whoever gets it second finds the object built and returns it
    import fcntl

    lock_path = out.with_suffix(".lock")
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if out.exists() and not force:
                return out
            return _compile_locked(text, fields, assign, out)
        finally:
            # [round 5] the lock file goes when its build ends (round 4 left 2528 of them in the cache, and they travelled
            # to every GPU box).  A process still queued on the unlinked file gets its lock, finds the object and returns;
            # one that arrives later makes a new file -- at worst two builders, each renaming a complete object into place.
            # (whether the build succeeded or not: a failed or interrupted build left its lock behind until now)
            try:
                lock_path.unlink()
            except OSError:
                pass
            fcntl.flock(lock, fcntl.LOCK_UN)


def _compile_locked(text: str, fields: dict, assign: str, out: pathlib.Path) -> pathlib.Path:
    import threading

    # (unique per builder: two threads of one process may build the same object -- the lock file of a finished build is unlinked)
    tmp = out.with_suffix(f".tmp{os.getpid()}_{threading.get_ident()}.so")
    cmd = [_HIPCC, *_flags(), f"-DJXS_SPEC_T={fields['T']}", f"-DJXS_SPEC_G={fields['G']}", f"-DJXS_SPEC_MODE={fields['MODE']}",
           f"-DJXS_SPEC_ASSIGN={assign}", f'-DJXS_SPEC_STRING="{text}"', "jxs_spec.hip", "-o", str(tmp)]  # fmt: skip
    # (JAXSIM_AMD_SPEC_CSRC, developer aid of tools/gpu/*.sh: the kernel sources of another directory -- a copy of an earlier
    # csrc/ with the same parameter block -- compiled against this library, for A/B runs on one box)
    r = subprocess.run(cmd, cwd=_src_dir(), capture_output=True, text=True)
    if r.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed for the specialised kernel ({text}):\n{r.stderr[-4000:]}")
    # [round 4] wait-state lint of the object (isa_lint.py): a hazard inside a hand-written DPP block fails the build
    # here, before any launch could run it (the library's own kernels are linted by csrc/build.sh)
    try:
        from . import isa_lint

        isa_lint.check(str(tmp))
    except Exception as exc:  # [ADVICE r4] whatever the lint dies of, the temporary object goes and the caller sees ONE
        tmp.unlink(missing_ok=True)  # exception type it already handles (attach / ensure_mode fall back to the library's kernel)
        if isinstance(exc, RuntimeError):
            raise
        raise RuntimeError(f"ISA lint of the specialised kernel failed to run: {exc!r}") from exc
    os.replace(tmp, out)  # atomic: concurrent ranks may build the same object
    return out


# forward dynamics, inverse dynamics, cached kinematics, mass matrix, Jacobians, mass-matrix inverse, gravity torques: on request
QUERY_MODES = (1, 2, 3, 8, 9, 10, 11)


def attach(dm, model, mode: int | None = None, *, build: bool = False, require: bool = False) -> bool:
    """Attach the specialised kernels of the model (all modes of ``modes_of``, or one ``mode``) to the device
    model ``dm``.  Without ``build`` only objects found in the cache are used; with it the missing ones are
    compiled now, concurrently (one hipcc process per mode).  True if any was attached."""
    lib = _lib.load()
    todo = modes_of(model) if mode is None else ([mode] if isinstance(mode, int) else list(mode))
    paths: dict[int, pathlib.Path | None] = {m: cached(model, dm.dtype, m) for m in todo}
    missing = [m for m in todo if paths[m] is None]
    if require and missing and not build:
        raise _lib.JaxsimAmdError("JAXSIM_AMD_SPECIALIZE=require: no pre-built model-specialised kernel for " +
                                  "; ".join(spec(model, dm.dtype, m) for m in missing))
    if build and missing:
        # one hipcc process per missing mode; a mode whose build fails falls back to the library's own kernel ALONE
        # (the objects that did build are attached), and the failure is reported once all are done
        from concurrent.futures import ThreadPoolExecutor

        def one(mm):
            try:
                return compile(model, dm.dtype, mm), None
            except (RuntimeError, OSError) as exc:
                return None, exc

        with ThreadPoolExecutor(len(missing)) as ex:
            results = list(ex.map(one, missing))
        failed = []
        for m, (p, exc) in zip(missing, results):
            paths[m] = p
            if exc is not None:
                failed.append((m, exc))
        if failed and len(failed) == len(missing) and all(paths[m] is None for m in todo):
            raise failed[0][1]
        for m, exc in failed:
            import warnings

            warnings.warn(f"jaxsim_amd: no model-specialised kernel for mode {m} ({exc}); using the generic one", RuntimeWarning, stacklevel=2)
    done = False
    for m in todo:
        p = paths[m]
        if p is not None:
            _lib.check(lib.jxs_model_attach_specialized(dm.handle, m, str(p).encode()), "jxs_model_attach_specialized")
            dm.__dict__.setdefault("_spec_files", {})[m] = p.name
            done = True
    return done


MODE_GRAV = 11


def ensure_mode(dm, model, mode: int) -> bool:
    """Attach the specialised kernel of a query mode on its first use, following ``policy()`` (cached object, or
    built now when the compiler is there).  Called by the API functions whose kernels are not part of ``modes_of``
    (e.g. the gravity-torque kernel behind ``gravity_compensation_torques``).  True if the mode runs specialised."""
    tried = dm.__dict__.setdefault("_spec_tried", set())
    if mode in tried:
        return mode in attached_files(dm)
    tried.add(mode)
    how = policy()
    if how == "off":
        return False
    if how == "require":
        return attach(dm, model, mode, require=True)
    try:
        return attach(dm, model, mode, build=(how == "build"))
    except (RuntimeError, OSError) as exc:
        import warnings

        warnings.warn(f"jaxsim_amd: no model-specialised kernel for mode {mode} ({exc}); using the generic one", RuntimeWarning, stacklevel=2)
        return False


def modes(dm) -> list[int]:
    """Modes of the device model that run a specialised kernel."""
    mask = C.c_uint()
    _lib.check(_lib.load().jxs_model_specialized_modes(dm.handle, C.byref(mask)), "jxs_model_specialized_modes")
    return [k for k in range(16) if mask.value >> k & 1]


def hipcc_available() -> bool:
    """The compiler AND the disassembler of the ISA lint are there: a build that cannot be linted is not attempted."""
    from . import isa_lint

    return os.path.isfile(_HIPCC) and os.access(_HIPCC, os.X_OK) and isa_lint.available()


def policy() -> str:
    """'off' (never), 'cached' (use an object that exists, never compile), 'build' (compile on first use).

    Default [round 3]: **build on first use whenever hipcc is present** -- what ``jax.jit`` does for the reference
    (the first ``js.model.step`` of a model compiles, src/jaxsim/api/model.py:2599-2601) -- and 'cached' on a machine
    without the compiler.  ``JAXSIM_AMD_SPECIALIZE=0`` / ``cached`` / ``1`` force a choice.  A failed build is a
    logged warning and the library's own ahead-of-time kernel runs (``KV_COMMON`` / generic): never a CPU path."""
    v = os.environ.get("JAXSIM_AMD_SPECIALIZE", "")
    if v == "0":
        return "off"
    if v == "1":
        return "build"
    if v == "cached":
        return "cached"
    if v == "require":  # the test-suite's specialised pass: like 'cached', but a missing object is an error, not a fallback
        return "require"
    return "build" if hipcc_available() else "cached"


def attached_files(dm) -> dict[int, str]:
    """``{mode: file name}`` of the specialised objects attached to ``dm`` by this module."""
    return dict(getattr(dm, "_spec_files", {}))
