"""N>1 path on CPU: world_size-2 ``gloo`` job (127.0.0.1 rendezvous) + host sharding logic."""

import os
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest

from jaxsim_amd import distributed

HERE = pathlib.Path(__file__).resolve().parent


def test_shard_bounds_cover_the_batch():
    for n, w in ((8192, 8), (1024, 1), (10, 3), (7, 8)):
        cuts = [distributed.shard_bounds(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1
    assert distributed.shard_bounds(8192, 3, 8) == (3072, 4096)
    with pytest.raises(ValueError):
        distributed.shard_bounds(8, 8, 8)


def test_concat_shards_inverts_sharding():
    blk = np.arange(5 * 12, dtype=np.float32).reshape(5, 12)
    shards = np.stack([distributed.shard_block(blk, r, 3) for r in range(3)], axis=0)
    np.testing.assert_array_equal(distributed.concat_shards(shards), blk)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_2_gloo_shard_step_gather():
    import emul_binding

    emul_binding.build()  # build once, before the two ranks race for it
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(HERE / "dist_worker.py"),
    ]  # fmt: skip
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "DIST_OK" in res.stdout


def test_file_rendezvous_ignores_stale_files(tmp_path, monkeypatch):
    import os
    import tempfile
    import time

    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    rd = pathlib.Path(distributed._rendezvous_dir())
    assert rd.parent == tmp_path and (rd.stat().st_mode & 0o777) == 0o700
    stale = rd / "rdzv_k1.bin"
    stale.write_bytes(b"x" * 128)
    old = time.time() - 3600
    os.utime(stale, (old, old))
    from jaxsim_amd import _lib

    with pytest.raises(_lib.JaxsimAmdError):
        distributed.file_rendezvous(1, 2, "k1", timeout_s=0.2)
    fresh = rd / "rdzv_k2.bin"
    fresh.write_bytes(bytes(range(128)))
    assert distributed.file_rendezvous(1, 2, "k2", timeout_s=1.0) == bytes(range(128))
    # rank 0 replaces whatever an earlier job left under the same key
    assert distributed.file_rendezvous(0, 2, "k2", make_id=lambda: b"z" * 128) == b"z" * 128
    assert distributed.file_rendezvous(1, 2, "k2", timeout_s=1.0) == b"z" * 128


@pytest.mark.parametrize("world", [2, 8])
def test_bench_bootstrap_dry_run(world):
    """bench.py's own multi-rank bootstrap (job key, id file, host collective, shard bounds of 1024 x world
    environments) launched exactly like the driver launches the multi-GPU bench, without a device."""
    import json

    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(HERE.parent / "bench.py"),
        "--gpus", str(world), "--steps", "20", "--warmup", "5", "--dry-run-bootstrap",
    ]  # fmt: skip
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("JAXSIM_AMD_LIB", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, res.stdout
    out = json.loads(line[0])
    assert out["ok"] and out["n_gpus"] == world and out["global_batch"] == 1024 * world
    assert out["shards"][0] == [0, 1024] and out["shards"][-1] == [1024 * (world - 1), 1024 * world]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_is_its_own_launcher(world):
    """[round 4] `python bench.py --gpus N` as ONE command, without torch.distributed.run: bench.py starts one rank per
    GPU itself (bench.py::spawn_ranks), rank 0's JSON line is the only thing on stdout, exit code 0."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "JAXSIM_AMD_LIB")}
    res = subprocess.run([sys.executable, str(HERE.parent / "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5", "--dry-run-bootstrap"],
                         capture_output=True, text=True, timeout=300, env=dict(env, OMP_NUM_THREADS="1"))  # fmt: skip
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = res.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout
    out = json.loads(lines[0])
    assert out["ok"] and out["n_gpus"] == world and out["global_batch"] == 1024 * world and out["same_id_on_all_ranks"]
    assert out["shards"][-1] == [1024 * (world - 1), 1024 * world]
    # [round 5] the self-launcher gives every rank its own LOCAL_RANK, i.e. its own device of the node
    assert out["device_of_rank"] == list(range(world)) and out["distinct_devices"]


def test_bench_launcher_propagates_a_failing_rank():
    """A rank that cannot run (here: no second HIP device -- on a CPU box none at all) makes the one-command launch
    exit non-zero without a result line, and no rank is left waiting in the rendezvous."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "JAXSIM_AMD_LIB")}
    res = subprocess.run([sys.executable, str(HERE.parent / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--saturated-envs", "0", "--no-other-contact-models"],
                         capture_output=True, text=True, timeout=300, env=dict(env, OMP_NUM_THREADS="1", ROCR_VISIBLE_DEVICES="0"))  # fmt: skip
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")], res.stdout
    assert "stopping the other ranks" in res.stderr or "exited with" in res.stderr


def test_all_gather_state_refuses_unequal_shards(monkeypatch):
    class FakeComm:
        world_size = 2

        def all_gather_scalars(self, v):
            return np.array([v, v + 1.0])

        def all_gather(self, st):
            raise AssertionError("the data path must not be reached")

    class St:
        cols, rows, tile = 5, 3, 1

    class Data:
        _state = St()

    from jaxsim_amd import _lib

    with pytest.raises(_lib.JaxsimAmdError, match="equal shards"):
        distributed.all_gather_state(FakeComm(), Data())


def test_file_collective_two_ranks(tmp_path, monkeypatch):
    import tempfile
    import threading

    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    monkeypatch.setattr(distributed.runtime, "synchronize", lambda *a, **k: None)
    out = {}

    def worker(r):
        fc = distributed.FileCollective(r, 2, "t")
        fc.barrier()
        out[r] = fc.all_gather_scalars(10.0 + r)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    np.testing.assert_array_equal(out[0], [10.0, 11.0])
    np.testing.assert_array_equal(out[1], [10.0, 11.0])
