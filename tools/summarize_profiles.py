#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (gpurun_out/<run>/...) into profiles/<tag>_*.{csv,md}."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

run = pathlib.Path(sys.argv[1])  # e.g. gpurun_out/r8
tag = sys.argv[2]  # e.g. r01
out = pathlib.Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True)
lines = [f"# rocprofv3 summary {tag} (source: {run})", ""]
ks = glob.glob(str(run / "stats" / "*" / "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], out / f"{tag}_kernel_stats.csv")
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 500 --warmup 20 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models`", "",
              "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks[0])):
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
    lines.append("")
summary = {}
for d in sorted(run.glob("pmc_*")):
    f = glob.glob(str(d / "*" / "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    waves = None
    for r in csv.DictReader(open(f[0])):
        if "jxs_kernel<float, 32, 0," in r["Kernel_Name"]:  # the step kernel only (not the fused rollout / kinematics)
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            waves = int(r["Grid_Size"]) // 64
    for k, v in agg.items():
        summary[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v), "waves_per_launch": waves}
if summary:
    lines += ["## PMC counters of `jxs_kernel<float,32,MODE_STEP>` (separate `--pmc` passes, mean per launch)", "",
              "| counter | mean per launch | per wave | launches |", "|---|---|---|---|"]
    for k, v in summary.items():
        lines.append(f"| {k} | {v['mean_per_launch']:.1f} | {v['mean_per_launch'] / v['waves_per_launch']:.1f} | {v['launches']} |")
    lines.append("")
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        fetch = summary["FETCH_SIZE"]["mean_per_launch"] * 1024
        write = summary["WRITE_SIZE"]["mean_per_launch"] * 1024
        lines += [f"HBM-side traffic per launch: FETCH_SIZE {fetch / 1e6:.2f} MB raw (x2 gfx950 correction of "
                  f"MI355X_MICROARCH.md section HBM = {2 * fetch / 1e6:.2f} MB), WRITE_SIZE {write / 1e6:.2f} MB.", ""]
        summary["traffic_bytes_per_launch"] = 2 * fetch + write
    # bench.py only quotes the traffic for the configuration and the kernel sources it was measured on
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import bench

    summary["config"] = {"model": "icub23", "envs": 1024, "dtype": "float32"}
    summary["kernel_source_sha"] = bench.kernel_source_sha()
    (out / f"{tag}_pmc.json").write_text(json.dumps(summary, indent=1))
(out / f"{tag}_summary.md").write_text("\n".join(lines))
print("\n".join(lines))
