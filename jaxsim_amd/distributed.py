"""Multi-GPU: one process per GPU, the batch sharded in contiguous slices, no per-step
communication; ONE RCCL all-gather over xGMI concatenates the final state shards.

The reference has no multi-device code (single-process JAX, ``jax.vmap`` only; SURVEY.md
headline fact 1): environments never interact, so the batch shards trivially.  The process
group of the *launcher* (``torch.distributed``, started by ``torch.distributed.run``) is used
only for bootstrap -- broadcasting the 128-byte RCCL unique id, barriers, and the timing
reduction of ``bench.py`` -- while the data path goes through ``jxs_allgather`` (RCCL called
from the C-ABI library on raw device pointers).  On a machine without GPUs (the CPU tests,
``gloo`` backend) the same host logic gathers through ``torch.distributed`` instead.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, runtime


def shard_bounds(n_total: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous slice ``[lo, hi)`` of the batch owned by ``rank`` (remainder spread over the
    first ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError((rank, world_size))
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_block(block: np.ndarray, rank: int, world_size: int) -> np.ndarray:
    """Slice a host ``[rows, N]`` state block for ``rank``."""
    lo, hi = shard_bounds(block.shape[1], rank, world_size)
    return np.ascontiguousarray(block[:, lo:hi])


def concat_shards(gathered: np.ndarray) -> np.ndarray:
    """``[world, rows, n_local]`` (the all-gather result) -> ``[rows, world * n_local]``."""
    w, rows, n = gathered.shape
    return np.ascontiguousarray(np.transpose(gathered, (1, 0, 2)).reshape(rows, w * n))


class Communicator:
    """RCCL communicator owned by the C-ABI library (``jxs_comm_*``)."""

    def __init__(self, unique_id: bytes, rank: int, world_size: int):
        if len(unique_id) != 128:
            raise ValueError("RCCL unique id must be 128 bytes")
        self.rank, self.world_size = int(rank), int(world_size)
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(_lib.load().jxs_comm_init(C.byref(h), buf, self.rank, self.world_size), "jxs_comm_init")
        self.handle = h

    @staticmethod
    def create_unique_id() -> bytes:
        buf = (C.c_char * 128)()
        _lib.check(_lib.load().jxs_comm_unique_id(buf), "jxs_comm_unique_id")
        return bytes(buf.raw)

    def all_gather(self, shard: runtime.DeviceArray) -> runtime.DeviceArray:
        """Gather equal-sized shards; the result stores rank r's tiles at ``[r]`` of ``[world][...]``."""
        out = runtime.DeviceArray(shard.rows * self.world_size, shard.cols, shard.dtype, tile=shard.tile)
        _lib.check(
            _lib.load().jxs_allgather(
                self.handle, C.c_void_p(shard.ptr), C.c_void_p(out.ptr), shard.n_tiles * shard.rows * shard.tile,
                _lib.dtype_code(shard.dtype), runtime._sp(),
            ),
            "jxs_allgather",
        )  # fmt: skip
        return out

    def all_gather_scalars(self, value: float) -> np.ndarray:
        """One float64 per rank, gathered through RCCL; returns the ``[world]`` host array."""
        send = runtime.DeviceArray.from_host(np.array([[float(value)]], dtype=np.float64), tile=1)
        out = self.all_gather(send)
        runtime.synchronize()
        return out.to_host_raw().reshape(-1)[: self.world_size].copy()

    def barrier(self) -> None:
        """Device-side barrier: a one-element all-gather followed by a stream synchronisation."""
        self.all_gather_scalars(0.0)

    def __del__(self):
        try:
            _lib.load().jxs_comm_destroy(self.handle)
        except Exception:
            pass


def _rendezvous_dir() -> str:
    """Per-user directory (mode 0700) under the temp dir for the bootstrap files: other users of the host
    can neither read the RCCL id nor plant a file under a predictable name."""
    import os
    import tempfile

    d = os.path.join(tempfile.gettempdir(), f"jaxsim_amd_{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid():
        raise _lib.JaxsimAmdError(f"{d} is owned by another user")
    if st.st_mode & 0o077:
        os.chmod(d, 0o700)
    return d


def file_rendezvous(rank: int, world_size: int, key: str, timeout_s: float = 120.0, make_id=None) -> bytes:
    """Single-node bootstrap without any framework: rank 0 creates the RCCL unique id and
    publishes it through an atomically renamed file in a per-user directory under the temp dir; the
    other ranks poll for it.  ``key`` must be the same on all ranks of one job and unique per job (the
    launcher's MASTER_PORT / run id / pid, ``job_key``).  ``make_id`` replaces the RCCL call (dry runs
    of the bootstrap on a machine without GPUs)."""
    import os
    import time

    path = os.path.join(_rendezvous_dir(), f"rdzv_{key}.bin")
    # A file left behind by a crashed earlier job with the same key is ignored.  The window is wide:
    # ranks of one job can start minutes apart (the first import on a fresh box pages the image in).
    fresh_after = time.time() - 900.0
    if rank == 0:
        try:
            os.remove(path)  # a leftover of an earlier job with the same key must not be picked up
        except OSError:
            pass
        uid = (make_id or Communicator.create_unique_id)()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    deadline = time.monotonic() + timeout_s
    while time.monotonic() < deadline:
        try:
            if os.path.getmtime(path) >= fresh_after:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid
        except OSError:
            pass
        time.sleep(0.02)
    raise _lib.JaxsimAmdError(f"rendezvous timed out waiting for {path}")


def job_key() -> str:
    """Identifier shared by the ranks of ONE launch and by no other: the launcher's rendezvous port and
    run id plus the pid of the launcher process itself (every rank is its child)."""
    import os

    parts = [str(os.environ.get(k, "")) for k in ("MASTER_PORT", "TORCHELASTIC_RUN_ID")]
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        parts.append(str(os.getppid()))
    return "_".join(parts)


def communicator_from_env(tag: str = "") -> Communicator:
    """Communicator for a job started by ``torch.distributed.run`` (or any launcher exporting
    RANK / WORLD_SIZE / MASTER_PORT): file rendezvous + ``ncclCommInitRank``.  No torch import."""
    import os

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    key = job_key() + tag
    comm = Communicator(file_rendezvous(rank, world, key), rank, world)
    comm.barrier()
    if rank == 0:  # everyone has read the id once the first collective completed
        try:
            os.remove(os.path.join(_rendezvous_dir(), f"rdzv_{key}.bin"))
        except OSError:
            pass
    return comm


class FileCollective:
    """Last-resort host collective for a single node (used by bench.py only if the RCCL
    communicator cannot be created): barriers and scalar gathers through files in the temp dir."""

    def __init__(self, rank: int, world_size: int, key: str, device_sync: bool = True):
        import os

        self.rank, self.world_size = int(rank), int(world_size)
        self.device_sync = bool(device_sync)
        self.dir = os.path.join(_rendezvous_dir(), f"fc_{key}")
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        self.seq = 0

    def all_gather_scalars(self, value: float, timeout_s: float = 300.0) -> np.ndarray:
        import os
        import time

        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}_{self.rank}")
        with open(mine + ".tmp", "w") as f:
            f.write(repr(float(value)))
        os.replace(mine + ".tmp", mine)
        out = np.zeros(self.world_size)
        deadline = time.monotonic() + timeout_s
        for r in range(self.world_size):
            path = os.path.join(self.dir, f"{self.seq}_{r}")
            while True:
                try:
                    with open(path) as f:
                        out[r] = float(f.read())
                    break
                except (OSError, ValueError):
                    if time.monotonic() > deadline:
                        raise _lib.JaxsimAmdError(f"file collective timed out on {path}") from None
                    time.sleep(0.0002)
        return out

    def barrier(self) -> None:
        if self.device_sync:
            runtime.synchronize()
        self.all_gather_scalars(0.0)


def communicator_from_torch() -> Communicator:
    """Bootstrap a ``Communicator`` from an initialised ``torch.distributed`` process group:
    rank 0 creates the RCCL unique id, the launcher's group broadcasts it."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    box = [Communicator.create_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Communicator(box[0], rank, world)


def all_gather_state_blocks_host(local_block: np.ndarray) -> np.ndarray:
    """CPU path of the final concat (``gloo``): gather host ``[rows, n_local]`` blocks of equal
    size through the launcher's process group and return ``[rows, N_total]``."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(local_block))
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return concat_shards(np.stack([o.numpy() for o in outs], axis=0))


def untile_gathered_on_device(gathered: runtime.DeviceArray, rows: int, n_total: int, tile: int) -> np.ndarray:
    """The storage of an all-gather of whole-tile shards, read as the tiled ``[rows][n_total]`` array it is: untiled by
    the device kernel behind ``jxs_tile_to_env_major`` and returned as a host ``[rows, n_total]`` array."""
    env_major = runtime.DeviceArray(1, rows * n_total, gathered.dtype, tile=1)
    _lib.check(
        _lib.load().jxs_tile_to_env_major(C.c_void_p(gathered.ptr), C.c_void_p(env_major.ptr), int(rows), int(n_total), int(tile),
                                          _lib.dtype_code(gathered.dtype), runtime._sp()),
        "jxs_tile_to_env_major",
    )  # fmt: skip
    return np.ascontiguousarray(env_major.to_host_raw().reshape(n_total, rows).T)


def all_gather_state(comm: Communicator, data) -> np.ndarray:
    """GPU path of the final concat: RCCL all-gather of the device state, returned on the
    host as one ``[rows, N_total]`` block (every rank gets the full batch)."""
    from .state import untile_block

    st = data._state
    # ncclAllGather takes ONE per-rank count: unequal shards (shard_bounds spreads a remainder over the
    # first ranks) would hang or corrupt the gather -- check before touching the data path
    cols = comm.all_gather_scalars(float(st.cols))
    if not np.all(cols == cols[0]):
        raise _lib.JaxsimAmdError(f"all_gather_state needs equal shards, got {cols.astype(int).tolist()} environments per rank; "
                                  "pad the batch to a multiple of the world size")
    gathered = comm.all_gather(st)  # storage: [world][n_tiles][rows][tile]
    if st.cols % st.tile == 0:
        # [round 6] whole tiles per rank: the gathered storage IS the tiled storage of a [rows][world * cols] array, rank
        # r's environments at columns r * cols ... -- untiled ON THE DEVICE (jxs_tile_to_env_major) into [N_total][rows]
        # and downloaded once; the host only takes a transposed view (round 5 untiled every rank's block on the host)
        return untile_gathered_on_device(gathered, st.rows, st.cols * comm.world_size, st.tile)
    raw = gathered.to_host_raw().reshape(comm.world_size, -1)
    return concat_shards(np.stack([untile_block(raw[r], st.rows, st.cols, st.tile) for r in range(comm.world_size)]))
