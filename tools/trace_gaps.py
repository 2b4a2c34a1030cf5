#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 --kernel-trace CSV, print duration / gap statistics of the step kernel,
overall and for bursts of exactly K back-to-back launches (the timed regions of `bench.py --steps K`).
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [K]"""
import csv
import glob
import sys

import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [r for r in csv.DictReader(open(f)) if "jxs_kernel" in r["Kernel_Name"] and ", 0>" in r["Kernel_Name"]]
t0 = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
t1 = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
o = np.argsort(t0)
t0, t1 = t0[o], t1[o]
dur, gap = (t1 - t0) / 1e3, (t0[1:] - t1[:-1]) / 1e3
print(f"{len(rows)} step-kernel dispatches: duration median {np.median(dur):.2f} us, gap median {np.median(gap):.2f} us")
# bursts = maximal runs with gaps < 5 us
cuts = np.where(gap > 5.0)[0]
starts = np.concatenate([[0], cuts + 1])
ends = np.concatenate([cuts, [len(t0) - 1]])
for s, e in zip(starts, ends):
    n = e - s + 1
    if n != K:
        continue
    d, g = dur[s : e + 1], gap[s:e]
    print(f"burst of {n}: span {(t1[e] - t0[s]) / 1e3:8.2f} us = {(t1[e] - t0[s]) / 1e3 / n:.3f}/launch | first 3 dur {d[:3].round(2)} last {d[-1]:.2f} | "
          f"median dur {np.median(d):.2f} gaps: first 3 {g[:3].round(2)} median {np.median(g):.2f} max {g.max():.2f}")
