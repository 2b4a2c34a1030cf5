#!/usr/bin/env python3
"""Developer tool [round 6, VERDICT r5 next 3]: the ISSUE FLOOR of every phase of the model-specialised step kernel next
to the cycles measured for it.

A wave that is alone on its SIMD (batch 1024: 512 waves on 1024 SIMDs) issues one vector instruction every ~5.4 ticks
whether or not it depends on the one before (profiles/r05_issue_rate_ubench.txt: dependent v_fma 5.35 ticks, eight
independent chains 5.43), so the time of a phase is bounded from below by

    floor = sum over the instructions between its two cycle stamps of the instruction's issue cost   (COST below: the
            "+ <instruction>" rows of the same micro-benchmark = what one more instruction of that kind adds to a wave)

and what a phase takes beyond its floor is exposed latency (LDS / memory round trips the wave waits for at an s_waitcnt
with nothing left to issue).  The tool disassembles the -DJXS_PHASE_TIMING build of the kernel (hipcc cross-compiles:
no GPU needed for the static half), cuts the instruction stream at the stamps (s_memtime + the store of slot i), prices
every piece, and prints it beside the measured cycles: from a log of tools/phase_timing.py (--measured FILE), or measured
here when a device is present.

  JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING python tools/issue_floor.py [--measured gpurun_out/.../phases_1024.log]
"""
import argparse
import collections
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JAXSIM_AMD_SPEC_EXTRA_FLAGS", "-DJXS_PHASE_TIMING")
import bench  # noqa: E402
from jaxsim_amd import isa_lint, specialize  # noqa: E402

# ticks one more instruction of the class costs a wave that is alone on its SIMD (profiles/r05_issue_rate_ubench.txt)
COST = {"valu": 5.35, "cndmask": 6.44, "pk": 4.05, "dpp": 4.10, "trans": 7.12, "vmov": 3.20, "salu": 3.30, "snop": 3.14,
        "waitcnt": 4.46, "branch": 17.17, "bperm": 18.42, "ds_read": 14.9, "ds_write": 14.9, "vmem": 16.0, "smem": 3.30}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")
PHASES = {1: "loads arrive", 2: "actuation+local xform", 3: "FK (pointer jumping)", 4: "velocities", 5: "contacts", 6: "inertia+bias",
          7: "pass 2 (rest)", 8: "base solve", 9: "pass 3", 10: "integrate+stores"}


def classify(i) -> str:
    op = i.op
    if op == "s_nop":
        return "snop"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith(("s_load", "s_memtime", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
        return "bperm"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if "dpp" in op or any(("row_" in a or "quad_perm" in a or "wave_" in a or "row_bcast" in a) for a in i.args):
            return "dpp"
        if op.startswith("v_pk_"):
            return "pk"
        if op.startswith(TRANS):
            return "trans"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "vmov"
        return "valu"
    return "salu"


def segments(insts):
    """[(slot of the stamp that ENDS the piece, [instructions])]: the stream cut at the stamps; the stamp's own
    instructions (s_and_saveexec .. s_memtime .. global_store .. s_or exec) are dropped."""
    out, cur, k = [], [], 0
    n = len(insts)
    while k < n:
        i = insts[k]
        if i.op == "s_memtime":
            # the store of this stamp: the next global_store_dwordx2; its offset / 8 is the slot
            j = k + 1
            while j < n and not insts[j].op.startswith("global_store_dwordx2"):
                j += 1
            m = re.search(r"offset:(\d+)", insts[j].text) if j < n else None
            slot = int(m.group(1)) // 8 if m else 0
            # drop the mask set-up in front of the stamp (s_and_saveexec + branch) if it is the tail of `cur`
            while cur and (cur[-1].op.startswith(("s_and_saveexec", "s_cbranch_execz")) or cur[-1].op == "s_mov_b64" and "exec" in cur[-1].text):
                cur.pop()
            out.append((slot, cur))
            cur = []
            k = j + 1
            if k < n and insts[k].op == "s_or_b64" and "exec" in insts[k].text:
                k += 1
            continue
        cur.append(i)
        k += 1
    out.append((-1, cur))
    return out


def measured_from_log(path):
    got = {}
    for ln in open(path):
        m = re.match(r"\s+(.+?)\s{2,}(\d+)\s+\(max", ln)
        if m:
            got[m.group(1).strip()] = float(m.group(2))
        m = re.match(r"\s+pass 2 per level \(deepest first\): (.*)", ln)
        if m:
            got["levels"] = [float(x) for x in m.group(1).split()]
        m = re.match(r"\s+total\s+(\d+)", ln)
        if m:
            got["total"] = float(m.group(1))
    return got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--measured", help="a log of tools/phase_timing.py (same kernel, same flags)")
    ap.add_argument("--model", default="icub23", help="icub23 (the headline) | quadruped_rigid (BASELINE config 5: the stage loop is unrolled, the stamps of stage 0 and of the impact stage follow each other)")
    ap.add_argument("--dump", action="store_true", help="print the instructions of every piece")
    args = ap.parse_args()
    model = bench.build_quadruped_rigid() if args.model == "quadruped_rigid" else bench.build_model(args.model)
    path = specialize.compile(model, np.float32, specialize.mode_of(model))
    insts = None
    for elf in isa_lint.code_objects(str(path)):
        for sym, ii in isa_lint.parse(isa_lint.disassemble(elf)).items():
            if "jxs_kernel" in sym and "duo" not in sym:
                insts = ii
    if insts is None:
        raise SystemExit("no step kernel in " + str(path))
    if not any(i.op == "s_memtime" for i in insts):
        raise SystemExit("the object carries no stamps: set JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING")
    meas = measured_from_log(args.measured) if args.measured else {}
    segs = segments(insts)
    print(f"{path.name}: {len(insts)} instructions, {sum(i.op.startswith('v_') for i in insts)} vector; issue cost per class (ticks): "
          + " ".join(f"{k}={v}" for k, v in COST.items()))
    print(f"{'piece (ends at stamp)':34s} {'instr':>5s} {'valu':>5s} {'dpp':>4s} {'pk':>4s} {'trans':>5s} {'cnd':>4s} {'mov':>4s} {'lds':>4s} {'bprm':>4s} {'vmem':>4s} {'salu':>4s} {'wait':>4s} {'br':>3s} | {'floor':>6s} {'meas':>6s} {'ratio':>5s}")
    levels = list(meas.get("levels", []))
    # measured per piece: the arrival stamps 20..23 sit inside "loads arrive"; the per-level stamps 24.. inside "pass 2"
    tot_floor = tot_meas = 0.0
    acc_name, acc = None, None
    rows = []
    n_level = 0
    for slot, piece in segs:
        c = collections.Counter(classify(i) for i in piece)
        floor = sum(COST[k] * v for k, v in c.items())
        if slot in PHASES:
            name = PHASES[slot]
        elif 20 <= slot <= 23:
            name = f"(arrival stamp {slot}, inside 'loads arrive')"
        elif 24 <= slot <= 31:
            name = f"pass 2 level {slot - 24}"
        elif slot == 0:
            name = "(before the first stamp)"
        elif slot == 11:
            name = "(hw id stamp)"
        else:
            name = "(after the last stamp)" if slot < 0 else f"(stamp {slot})"
        m = None
        if slot in PHASES and slot != 7 and PHASES[slot] in meas:
            m = meas[PHASES[slot]]
        if 24 <= slot <= 31 and levels:
            m = levels[n_level] if n_level < len(levels) else None
            n_level += 1
        rows.append((name, len(piece), c, floor, m, piece))
    # "loads arrive" measured covers the pieces of stamps 20..23 and 1 together
    for name, n, c, floor, m, piece in rows:
        tot_floor += floor
        if m is not None:
            tot_meas += m
        lds = c["ds_read"] + c["ds_write"]
        print(f"{name:34s} {n:5d} {c['valu']:5d} {c['dpp']:4d} {c['pk']:4d} {c['trans']:5d} {c['cndmask']:4d} {c['vmov']:4d} {lds:4d} {c['bperm']:4d} {c['vmem']:4d} {c['salu'] + c['smem'] + c['snop']:4d} {c['waitcnt']:4d} {c['branch']:3d} | {floor:6.0f} "
              + (f"{m:6.0f} {m / floor if floor else 0:5.2f}" if m is not None else f"{'':6s} {'':5s}"))
        if args.dump:
            for i in piece:
                print(f"      {classify(i):8s} {i.text}")
    print(f"{'sum of the floors':34s} {'':77s}| {tot_floor:6.0f}" + (f"   measured total {meas['total']:.0f}" if "total" in meas else ""))
    if meas:
        # the pieces whose measured time the log gives only as a sum
        la = sum(f for nme, _, _, f, _, _ in rows if "arrival" in nme or nme in ("loads arrive", "(before the first stamp)", "(hw id stamp)"))
        print(f"  loads arrive: floor of its pieces {la:.0f}, measured {meas.get('loads arrive', 0):.0f} (memory latency: the wave waits for its first bytes)")


if __name__ == "__main__":
    main()
