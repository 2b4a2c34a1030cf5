"""bench.py on the GPU box: the one-JSON-line contract at N = 1, and the N > 1 path started by ONE command
(`python bench.py --gpus 2`: bench.py spawns its ranks itself) with the two ranks sharing the box's single device
(`--share-device`: RCCL refuses two ranks on one device, the host-side file collective takes over and the line says
so).  The kernel policy is bench.py's own (model-specialised, pre-built by `__graft_entry__.build()`)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
FAST = ["--steps", "20", "--warmup", "5", "--saturated-envs", "0", "--no-other-contact-models"]


def run_bench(*flags, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "JAXSIM_AMD_SPECIALIZE")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"bench.py must print ONE line on stdout, got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract_gpu():
    d = run_bench("--gpus", "1", *FAST, "--cpu-baseline-seconds", "2")
    assert d["metric"].startswith("env-steps/sec") and d["unit"] == "env-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["value"] > 5e7 and abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert "model-specialised" in r["kernel"]  # the default experience: the pre-built specialised step kernel ran
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert d["config"]["workload"]


@pytest.mark.gpu
def test_bench_two_ranks_from_one_command_gpu():
    d = run_bench("--gpus", "2", "--share-device", *FAST, "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["comm"]["ranks"] == 2 and len(d["comm"]["ms_per_step_per_rank"]) == 2
    # the max over the ranks is the job's time, the value counts both shards
    assert d["ms_per_step"] >= max(d["comm"]["ms_per_step_per_rank"]) - 1e-9
    assert abs(d["value"] - 2 * 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # two ranks on one device: RCCL says no and the line says which collective ran instead
    assert d["comm"]["kind"] in ("file_collective", "rccl") and d["comm"]["class"] in ("FileCollective", "Communicator")
    if d["comm"]["kind"] == "file_collective":
        assert "ncclCommInitRank" in d["comm"]["error"] and d["comm"]["nccl_version"] is None
    # [round 5] the line names the device of every rank; here both ranks share the box's one GPU, and it says so
    assert isinstance(d["device_pci_bus_ids"], list) and len(d["device_pci_bus_ids"]) == 2 and d["comm"]["distinct_devices"] is False


@pytest.mark.gpu
def test_bench_require_rccl_is_fatal_without_rccl_gpu():
    """[round 5] `--require-rccl`: a rank whose communicator is not RCCL exits non-zero -- two ranks on ONE device is the
    case a 1-GPU box can produce (ncclCommInitRank refuses it) -- so a scaling record cannot be a file collective by accident."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "JAXSIM_AMD_LIB")}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-device", "--require-rccl", *FAST, "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)  # fmt: skip
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")], res.stdout
    assert "--require-rccl" in res.stderr


@pytest.mark.gpu
def test_bench_global_batch_is_strong_scaling_gpu():
    """`--global-batch B` splits a fixed batch over the GPUs (north_star: batch 8192 over 1 / 2 / 4 / 8 GPUs) and the
    line says so."""
    d = run_bench("--gpus", "1", "--global-batch", "2048", *FAST, "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 2048 and d["config"]["envs_per_gpu"] == 2048
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
