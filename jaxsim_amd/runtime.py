"""Thin device runtime over the C-ABI: buffers, a size-bucketed free list, streams, events.

No PyTorch / no HIP Python bindings: memory comes from ``jxs_malloc`` and is handed to the
kernels as raw device pointers.  Freed buffers go back to a per-size free list so that the
functional ``step`` (a fresh state block per call, like the reference's immutable pytrees)
does not pay a ``hipMalloc`` per step.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _lib

_pool_lock = threading.Lock()
# bucket size -> [(pointer, stream the buffer was last used on)]: kernels are asynchronous, so a pooled
# buffer may still be read by work in flight on that stream.  A new owner on the SAME stream is ordered
# behind it by the stream itself; an owner on another stream first waits for the old one (DeviceArray.__init__).
_pool: dict[int, list[tuple[int, object]]] = {}
_current_stream = None  # None = default (null) stream
_NULL = "null-stream"


def current_stream():
    return _current_stream


def set_stream(stream) -> None:
    """Select the HIP stream (``Stream`` or ``None``) used by subsequent calls."""
    global _current_stream
    _current_stream = stream


def _sp(stream=None):
    s = stream if stream is not None else _current_stream
    return None if s is None else s.handle


def device_count() -> int:
    lib = _lib.load()
    n = C.c_int(0)
    rc = lib.jxs_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_device() -> None:
    if device_count() == 0:
        raise _lib.JaxsimAmdError("no HIP device available; jaxsim_amd has no CPU fallback")


def set_device(index: int) -> None:
    _lib.check(_lib.load().jxs_set_device(int(index)), "jxs_set_device")


def synchronize(stream=None) -> None:
    lib = _lib.load()
    if stream is None and _current_stream is None:
        _lib.check(lib.jxs_device_synchronize(), "jxs_device_synchronize")
    else:
        _lib.check(lib.jxs_stream_synchronize(_sp(stream)), "jxs_stream_synchronize")


class Stream:
    def __init__(self):
        h = C.c_void_p()
        _lib.check(_lib.load().jxs_stream_create(C.byref(h)), "jxs_stream_create")
        self.handle = h

    def synchronize(self):
        _lib.check(_lib.load().jxs_stream_synchronize(self.handle), "jxs_stream_synchronize")

    def __del__(self):
        try:
            _lib.load().jxs_stream_destroy(self.handle)
        except Exception:
            pass


class Event:
    def __init__(self):
        h = C.c_void_p()
        _lib.check(_lib.load().jxs_event_create(C.byref(h)), "jxs_event_create")
        self.handle = h

    def record(self, stream=None):
        _lib.check(_lib.load().jxs_event_record(self.handle, _sp(stream)), "jxs_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        _lib.check(_lib.load().jxs_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)), "jxs_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            _lib.load().jxs_event_destroy(self.handle)
        except Exception:
            pass


class DeviceArray:
    """A logical ``[rows][N]`` device array of float32/float64 (batch index fastest), stored
    tile-interleaved as ``[ceil(N/T)][rows][T]`` -- see ``state.tile_block``."""

    def __init__(self, rows: int, cols: int, dtype, *, tile: int, zero: bool = False):
        self.rows, self.cols, self.tile = int(rows), int(cols), int(tile)
        self.dtype = np.dtype(dtype)
        self.n_tiles = -(-self.cols // self.tile)
        self.nbytes = self.n_tiles * self.tile * self.rows * self.dtype.itemsize
        self._bucket = max(self.nbytes, 1)
        lib = _lib.load()
        ptr, last = None, None
        with _pool_lock:
            free = _pool.get(self._bucket)
            if free:
                ptr, last = free.pop()
        if ptr is None:
            p = C.c_void_p()
            _lib.check(lib.jxs_malloc(C.byref(p), self._bucket), "jxs_malloc")
            ptr = p.value
        elif last is not (_current_stream if _current_stream is not None else _NULL):
            # recycled from another stream: its last kernels may still be reading the buffer
            if last is _NULL:
                _lib.check(lib.jxs_device_synchronize(), "jxs_device_synchronize")
            else:
                last.synchronize()
        self._ptr = ptr
        self._streams = {_current_stream if _current_stream is not None else _NULL}
        if zero:
            _lib.check(lib.jxs_memset(C.c_void_p(self.ptr), 0, self.nbytes, _sp()), "jxs_memset")

    @classmethod
    def like(cls, other: "DeviceArray") -> "DeviceArray":
        """A new (uninitialised) array with the shape, dtype and tiling of ``other`` -- the output block of a functional
        ``step``: the metadata is copied, the buffer comes from the free list of that size (the block the previous
        ``data`` object just gave back, in the reference's loop ``data = js.model.step(model, data)``)."""
        self = cls.__new__(cls)
        self.rows, self.cols, self.tile, self.dtype = other.rows, other.cols, other.tile, other.dtype
        self.n_tiles, self.nbytes, self._bucket = other.n_tiles, other.nbytes, other._bucket
        cur = _current_stream if _current_stream is not None else _NULL
        ptr = None
        free = _pool.get(self._bucket)
        if free:
            try:
                ptr, last = free.pop()  # (list.pop is atomic under the GIL)
            except IndexError:
                ptr = None
        if ptr is None:
            p = C.c_void_p()
            _lib.check(_lib.load().jxs_malloc(C.byref(p), self._bucket), "jxs_malloc")
            ptr = p.value
        elif last is not cur:  # recycled from another stream: its last kernels may still be reading the buffer
            if last is _NULL:
                _lib.check(_lib.load().jxs_device_synchronize(), "jxs_device_synchronize")
            else:
                last.synchronize()
        self._ptr = ptr
        self._streams = {cur}
        return self

    @property
    def ptr(self):
        """Raw device pointer.  Every consumer passes it to work on the CURRENT stream, so reading it
        records that stream as a user of the buffer (see ``__del__``)."""
        self._streams.add(_current_stream if _current_stream is not None else _NULL)
        return self._ptr

    @property
    def shape(self):
        return (self.rows, self.cols)

    def __del__(self):
        ptr = getattr(self, "_ptr", None)
        if ptr is None:
            return
        self._ptr = None
        try:
            if len(self._streams) == 1:  # the usual case: one stream ever touched the buffer -- it goes back tagged with it
                (tag,) = self._streams
                free = _pool.get(self._bucket)
                if free is None:
                    with _pool_lock:
                        free = _pool.setdefault(self._bucket, [])
                free.append((ptr, tag))  # (list.append is atomic under the GIL)
                return
            streams = list(self._streams)
            # the buffer goes back to the pool tagged with ONE stream; work still in flight on any other
            # stream that touched it is waited for here (rare: set_stream() between uses)
            tag = streams[-1] if len(streams) == 1 else (_current_stream if _current_stream is not None else _NULL)
            for st in streams:
                if st is not tag:
                    if st is _NULL:
                        _lib.check(_lib.load().jxs_device_synchronize(), "jxs_device_synchronize")
                    else:
                        st.synchronize()
            with _pool_lock:
                _pool.setdefault(self._bucket, []).append((ptr, tag))
        except Exception:  # interpreter shutdown: the driver reclaims the memory with the process
            pass

    @staticmethod
    def from_host(a: np.ndarray, *, tile: int, dtype=None) -> "DeviceArray":
        """Upload a host ``[rows, N]`` array (tiling happens here, outside any timed loop)."""
        from .state import tile_block

        a = np.ascontiguousarray(a, dtype=dtype or a.dtype)
        if a.ndim != 2:
            raise ValueError("expected a [rows, N] array")
        out = DeviceArray(a.shape[0], a.shape[1], a.dtype, tile=tile)
        flat = tile_block(a, tile)
        _lib.check(
            _lib.load().jxs_memcpy_h2d(C.c_void_p(out.ptr), flat.ctypes.data_as(C.c_void_p), out.nbytes, _sp()),
            "jxs_memcpy_h2d",
        )
        return out

    @staticmethod
    def from_device_env_major(ptr: int, n_envs: int, rows: int, dtype, *, tile: int) -> "DeviceArray":
        """Tile a device buffer ``[N][rows]`` (row-major, e.g. the ``data_ptr`` of another framework's
        contiguous ``(N, rows)`` array) into a new ``DeviceArray``; stays on the device."""
        out = DeviceArray(rows, n_envs, dtype, tile=tile, zero=n_envs % tile != 0)
        _lib.check(
            _lib.load().jxs_tile_from_env_major(
                C.c_void_p(int(ptr)), C.c_void_p(out.ptr), out.rows, out.cols, out.tile, _lib.dtype_code(out.dtype), _sp()
            ),
            "jxs_tile_from_env_major",
        )
        return out

    def to_device_env_major(self, ptr: int) -> None:
        """Write this array as ``[N][rows]`` (row-major) into the device buffer at ``ptr``."""
        _lib.check(
            _lib.load().jxs_tile_to_env_major(
                C.c_void_p(self.ptr), C.c_void_p(int(ptr)), self.rows, self.cols, self.tile, _lib.dtype_code(self.dtype), _sp()
            ),
            "jxs_tile_to_env_major",
        )

    @property
    def __cuda_array_interface__(self) -> dict:
        """Raw view of the tiled storage (1-D) for frameworks that speak the CUDA array interface."""
        return {
            "shape": (self.n_tiles * self.rows * self.tile,),
            "typestr": self.dtype.str,
            "data": (int(self.ptr), False),
            "version": 3,
        }

    def to_host_raw(self) -> np.ndarray:
        """Download the storage as is: ``[n_tiles, rows, tile]``."""
        out = np.empty((self.n_tiles, self.rows, self.tile), dtype=self.dtype)
        _lib.check(
            _lib.load().jxs_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.nbytes, _sp()),
            "jxs_memcpy_d2h",
        )
        return out

    def to_host(self) -> np.ndarray:
        """Download as a host ``[rows, N]`` array."""
        from .state import untile_block

        return untile_block(self.to_host_raw().reshape(-1), self.rows, self.cols, self.tile)

    def copy(self) -> "DeviceArray":
        out = DeviceArray(self.rows, self.cols, self.dtype, tile=self.tile)
        _lib.check(
            _lib.load().jxs_memcpy_d2d(C.c_void_p(out.ptr), C.c_void_p(self.ptr), self.nbytes, _sp()),
            "jxs_memcpy_d2d",
        )
        return out


def trim_pool() -> None:
    """Return every pooled buffer to the driver."""
    lib = _lib.load()
    with _pool_lock:
        lists = list(_pool.values())
        _pool.clear()
    # (drained by pop: `DeviceArray.like` and `__del__` take and give buffers without the lock -- list.pop / append are
    # atomic -- so no buffer can be both handed out and freed)
    for ptrs in lists:
        while ptrs:
            try:
                p, _ = ptrs.pop()
            except IndexError:
                break
            lib.jxs_free(C.c_void_p(p))


class DeviceModel:
    """Owner of one ``jxs_model`` handle (device copy of the constant tables)."""

    def __init__(self, model, dtype):
        require_device()
        lib = _lib.load()
        desc, keep = _lib.make_desc(model, dtype)
        h = C.c_void_p()
        _lib.check(lib.jxs_model_create(C.byref(desc), C.byref(h)), "jxs_model_create")
        self.handle = h
        self.dtype = np.dtype(dtype)
        lay = _lib.Layout()
        _lib.check(lib.jxs_model_layout(h, C.byref(lay)), "jxs_model_layout")
        self.layout = lay

    def __del__(self):
        try:
            _lib.load().jxs_model_destroy(self.handle)
        except Exception:
            pass


def fp32_relaxed_defaults_guard(model, dtype) -> None:
    """[round 6] RelaxedRigidContacts in float32 with a NEGLIGIBLE regulariser is refused instead of answered with no
    correct digit.  The regulariser of the relaxed system ``(J M^-1 J^T + R) f = a_ref - a_free`` scales with
    ``2 mu^2 (1 + mu^2)`` relative to the Delassus entries (``relaxed_rigid.py:540-568``); with the reference's DEFAULT
    ``mu = 0.005`` it is 5e-5 of them, below what float32 resolves once the Delassus matrix is rank deficient (two or more
    points on one link, or more contact rows than degrees of freedom): measured errors of 6e-2 (median) to 1e2 over random
    trees, HISTORY.md section 4e -- the reference's own formulation evaluated in float32 fails its Cholesky there.
    float64 (the reference's default precision) is exact at any ``mu``; ``mu = 0.5`` (the reference's
    ``estimate_good_contact_parameters`` idiom) is within 2e-4 in float32.  The threshold is the one the packer uses to
    choose the solver (``csrc/jxs_pack.h``: ``2 mu^2 (1 + mu^2) >= 0.02``).  ``JAXSIM_AMD_FP32_RELAXED_UNCHECKED=1`` runs
    such a model anyway (finite results, no accuracy claim: ``test_relaxed_defaults_in_fp32_stay_finite``)."""
    if np.dtype(dtype) != np.float32 or type(model.contact_model).__name__ != "RelaxedRigidContacts":
        return
    kdp = model.kin_dyn_parameters
    en = np.asarray(kdp.contact_enabled, dtype=bool)
    if not en.any():
        return
    mu = float(model.contact_params.mu)
    if 2.0 * mu * mu * (1.0 + mu * mu) >= 0.02:
        return
    per_link = np.bincount(np.asarray(kdp.contact_body)[en], minlength=kdp.number_of_links())
    deficient = per_link.max() >= 2 or 3 * int(en.sum()) > 6 + kdp.number_of_joints()
    if not deficient or os.environ.get("JAXSIM_AMD_FP32_RELAXED_UNCHECKED"):
        return
    raise ValueError(
        f"RelaxedRigidContacts in float32 with mu = {mu:g}: the regulariser (2 mu^2 (1 + mu^2) = {2.0 * mu * mu * (1.0 + mu * mu):.1e} of the "
        "Delassus entries) is below float32 resolution for a rank-deficient contact set (several points on one link): no "
        "accuracy can be stated.  Use float64 (the reference's default precision), or contact parameters from "
        "estimate_good_contact_parameters (mu = 0.5), or set JAXSIM_AMD_FP32_RELAXED_UNCHECKED=1 to run without a tolerance.")


def device_model(model, dtype) -> DeviceModel:
    """Device tables of ``model`` for ``dtype``; rebuilt when a model constant changed."""
    from . import specialize  # model-specialised step kernel: a cached object, or built now if asked for

    # (the kernel policy is part of the key: a process that changes JAXSIM_AMD_SPECIALIZE gets the kernels it asked for)
    sig = (_lib.model_signature(model, dtype), specialize.policy())
    cache = model.__dict__.setdefault("_device", {})
    hit = cache.get(np.dtype(dtype).str)
    if hit is not None and hit[0] == sig:
        return hit[1]
    fp32_relaxed_defaults_guard(model, dtype)
    dm = DeviceModel(model, dtype)
    how = specialize.policy()
    if how == "require":
        specialize.attach(dm, model, require=True)  # (raises: the suite's specialised pass must not fall back silently)
    elif how != "off":
        try:
            specialize.attach(dm, model, build=(how == "build"))
        except (RuntimeError, OSError) as exc:  # no hipcc / failed build: the generic kernel of the library runs
            import warnings

            warnings.warn(f"jaxsim_amd: no model-specialised kernel ({exc}); using the generic one", RuntimeWarning, stacklevel=2)
    cache[np.dtype(dtype).str] = (sig, dm)
    return dm
