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
out = pathlib.Path(sys.argv[3]) if len(sys.argv) > 3 else pathlib.Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True, parents=True)
lines = [f"# rocprofv3 summary {tag} (source: {run})", ""]
ks = glob.glob(str(run / "stats" / "*" / "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], out / f"{tag}_kernel_stats.csv")
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 500 --warmup 20 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models --no-python-loop`", "",
              "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks[0])):
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
    lines.append("")
summary, generic, headline_kernel = {}, {}, "jxs_kernel<float,32,MODE_STEP>"
for d in sorted(run.glob("pmc_*")):
    f = glob.glob(str(d / "*" / "*counter_collection.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    # the step kernel only (not the fused rollout / kinematics): the model-specialised build that the headline runs
    # (namespace jxs_launch_spec) when there is one, else the generic kernel of the library; the generic one separately
    spec = any("jxs_launch_spec::jxs_kernel<float, 32, 0," in r["Kernel_Name"] for r in rows)
    for name, dst in (("jxs_launch_spec::jxs_kernel<float, 32, 0," if spec else "jxs_launch::jxs_kernel<float, 32, 0,", summary),
                      ("jxs_launch::jxs_kernel<float, 32, 0,", generic)):
        agg = collections.defaultdict(list)
        waves = None
        for r in rows:
            if name in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                waves = int(r["Grid_Size"]) // 64
        for k, v in agg.items():
            dst[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v), "waves_per_launch": waves}
    headline_kernel = "jxs_launch_spec::jxs_kernel<float,32,MODE_STEP> (model-specialised)" if spec else "jxs_launch::jxs_kernel<float,32,MODE_STEP>"
if summary:
    for title, tab in ((headline_kernel, summary), ("jxs_launch::jxs_kernel<float,32,MODE_STEP> (the library's kernel without a per-model build -- variant 2 = KV_COMMON; bench.py's `generic_kernel` secondary)", generic)):
        if not tab or (tab is generic and "spec" not in headline_kernel):
            continue
        lines += [f"## PMC counters of `{title}` (separate `--pmc` passes, mean per launch)", "",
                  "| counter | mean per launch | per wave | launches |", "|---|---|---|---|"]
        for k, v in tab.items():
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

    # config 5 (quadruped, RigidContacts): HBM-side traffic of its step kernel from the c5_pmc_* passes
    c5 = {}
    for d in sorted(run.glob("c5_pmc_*")):
        f = glob.glob(str(d / "*" / "*counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "jxs_kernel<float, 16, 6," in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            c5[k] = sum(v) / len(v)
    if "FETCH_SIZE" in c5 and "WRITE_SIZE" in c5:
        summary["config5_traffic_bytes_per_launch"] = 2 * c5["FETCH_SIZE"] * 1024 + c5["WRITE_SIZE"] * 1024
        lines_c5 = (f"Config 5 step kernel (`jxs_kernel<float,16,MODE_STEP_RIGID>`, 4096 environments): FETCH_SIZE {c5['FETCH_SIZE'] * 1024 / 1e6:.2f} MB raw "
                    f"(x2 = {2 * c5['FETCH_SIZE'] * 1024 / 1e6:.2f} MB), WRITE_SIZE {c5['WRITE_SIZE'] * 1024 / 1e6:.2f} MB per launch.")
        lines += [lines_c5, ""]
    summary["config"] = {"model": "icub23", "envs": 1024, "dtype": "float32"}
    summary["kernel_source_sha"] = bench.kernel_source_sha()
    summary["kernel"] = headline_kernel
    (out / f"{tag}_pmc.json").write_text(json.dumps(summary, indent=1))
(out / f"{tag}_summary.md").write_text("\n".join(lines))
print("\n".join(lines))

# the other artefacts of tools/profile_round.sh, copied under the round's tag
for src, dst in (("c5", "c5_kernel_stats.csv"), ("c5_relaxed", "c5_relaxed_kernel_stats.csv"), ("relaxed_humanoid", "relaxed_humanoid_kernel_stats.csv")):
    ks = glob.glob(str(run / src / "*" / "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], out / f"{tag}_{dst}")
for src, dst in (("phases.log", "phase_cycles.txt"), ("phases_generic.log", "phase_cycles_generic_kernel.txt"),
                 ("phases_contact_models.log", "phase_cycles_contact_models.txt"),
                 ("phases_contact_models_generic.log", "phase_cycles_contact_models_generic_kernel.txt"),
                 ("bench_N1.json", "bench_N1.json"), ("bench_steps20.json", "bench_steps20.json"),
                 ("fp32_error_gpu.log", "fp32_error_gpu.txt"), ("issue_rate.log", "issue_rate_ubench.txt"), ("c5.log", "c5_bench.txt"),
                 ("phases_two_wave.log", "phase_cycles_two_wave.txt"), ("cu_share.log", "cu_share_ubench.txt"), ("sweep.log", "sweep_batch_sizes.txt")):  # fmt: skip
    if (run / src).exists():
        shutil.copy(run / src, out / f"{tag}_{dst}")
