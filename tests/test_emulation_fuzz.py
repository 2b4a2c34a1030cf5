"""A small slice of the emulation fuzz campaigns (tools/fuzz/*.py: the kernel sources compiled for the CPU against the
oracle on random trees) in the CPU suite: the campaigns of the round's end ran hundreds of trees
(profiles/r04_experiments.md) and found the one real defect of the round -- this keeps a dozen trees of each alive."""
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("script,seed,trials", [("fuzz_step.py", 21, 40), ("fuzz_rigid.py", 22, 30), ("fuzz_query.py", 23, 25),
                                                ("fuzz_rollout.py", 24, 30), ("fuzz_contact_tree.py", 25, 30), ("fuzz_contact_tree2.py", 26, 40)])  # fmt: skip
def test_fuzz_slice(script, seed, trials):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(ROOT), str(ROOT / "tests"), os.environ.get("PYTHONPATH", "")]))
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz" / script), str(seed), str(trials)], capture_output=True, text=True,
                       env=env, timeout=900)  # fmt: skip
    assert p.returncode == 0, p.stderr[-2000:]
    last = [ln for ln in p.stdout.splitlines() if ln.startswith("fails")]
    assert last and last[-1].startswith("fails 0 "), p.stdout[-2000:]
    # [ADVICE r4] a campaign that refuses or skips most of its trials proves little: the ones that count what they
    # compared must have compared at least three quarters of what they drew
    words = last[-1].split()
    if "compared" in words:
        compared, refused, failed = (int(words[words.index(k) + 1]) for k in ("compared", "refused", "oracle_failed"))
        assert compared >= 0.75 * (compared + refused + failed), last[-1]


def test_device_campaign_tool_dry_run(tmp_path):
    """[round 5] tools/fuzz/gpu_campaign.py (the fuzz campaign through the PRODUCT on an MI355X: `-m gpu` runs a slice,
    tests/test_gpu_parity.py; the full record is profiles/r05_gpu_fuzz_campaign.txt) -- its case generation, gates and
    table, here with the emulation's result standing in for the device's."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(ROOT), str(ROOT / "tests"), os.environ.get("PYTHONPATH", "")]), GPU_CAMPAIGN_DRY="1")
    tool, cases = str(ROOT / "tools" / "fuzz" / "gpu_campaign.py"), str(tmp_path / "cases.pkl")
    p = subprocess.run([sys.executable, tool, "prepare", cases, "41", "16"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "prepared" in p.stdout, p.stderr[-2000:]
    prepared = int(p.stdout.split("prepared")[1].split()[0])
    assert prepared >= 20  # (16 trees, two precisions each except RigidContacts)
    p = subprocess.run([sys.executable, tool, "run", cases], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and f"{prepared} cases compared" in p.stdout and "fails 0;" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
