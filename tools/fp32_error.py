#!/usr/bin/env python3
"""fp32 error budget of one step: per-environment error against the fp64 oracle of
(a) the kernel core in IEEE emulation with the anchored ABA, (b) the same with one reference point for the
whole tree (round-1 formulation, JXS_DISABLE_ANCHORS=1), both ABA layouts, (c) the reference formulation
evaluated in fp32 (the oracle run with float32 arrays).  CPU only (test infrastructure: oracle + emulation).   python tools/fp32_error.py [N] [model]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import emul_binding as eb  # noqa: E402
import helpers  # noqa: E402
import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "icub"
zoo = helpers.ModelZoo()
model = zoo(name)
if len(sys.argv) > 3 and sys.argv[3] == "bench":
    import bench

    model = bench.build_model("icub23")
d = zoo.random_data(name, N, seed=4, dtype=np.float32)
truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d)))
blk = helpers.odata_to_block(model, d)


def per_env(out):
    return np.max(np.abs(out.astype(np.float64) - truth) / np.maximum(1.0, np.abs(truth)), axis=0)


def report(tag, e):
    print(f"{tag:42s} median {np.median(e):.2e}  p90 {np.percentile(e, 90):.2e}  p99 {np.percentile(e, 99):.2e}  worst {e.max():.2e}")


for rows in ("", "1"):
    tag = "link-per-lane sweeps" if rows else "row-distributed passes"
    if rows:
        os.environ["JXS_DISABLE_ROW_MODE"] = "1"
    os.environ.pop("JXS_DISABLE_ANCHORS", None)
    report(f"kernel core, {tag}, anchored chains", per_env(eb.run(model, eb.MODE_STEP, blk)))
    os.environ["JXS_DISABLE_ANCHORS"] = "1"
    report(f"kernel core, {tag}, one origin (round 1)", per_env(eb.run(model, eb.MODE_STEP, blk)))
    os.environ.pop("JXS_DISABLE_ANCHORS", None)
    os.environ.pop("JXS_DISABLE_ROW_MODE", None)
report("reference formulation in fp32 (oracle)", per_env(helpers.odata_to_block(model, oracle.step(model, d))))

if os.environ.get("JXS_ERR_ROWS"):
    for tag, env in (("anchored", None), ("one origin", "1")):
        if env:
            os.environ["JXS_DISABLE_ANCHORS"] = env
        out = eb.run(model, eb.MODE_STEP, blk)
        os.environ.pop("JXS_DISABLE_ANCHORS", None)
        err = np.abs(out.astype(np.float64) - truth) / np.maximum(1.0, np.abs(truth))
        med = np.median(err, axis=1)
        n = model.dofs()
        names = ["p"] * 3 + ["q"] * 4 + [f"s:{j}" for j in model.joint_names()] + ["v"] * 3 + ["w"] * 3 + [f"sd:{j}" for j in model.joint_names()]
        top = np.argsort(-med)[:12]
        print(tag, [(names[i] if i < len(names) else f"m{i}", f"{med[i]:.1e}", f"ref~{np.median(np.abs(truth[i])):.1e}") for i in top])
