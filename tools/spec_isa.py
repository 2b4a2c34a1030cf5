#!/usr/bin/env python3
"""Assembly + instruction mix of a model-specialised kernel (developer aid; needs hipcc, not a GPU).
    python tools/spec_isa.py icub23|anymal12|quadruped_rigid|humanoid_relaxed float32 [mode] [extra hipcc flags...]   -> /tmp/spec_<model>_<dtype>_<mode>.s"""
import collections
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from jaxsim_amd import specialize as sp  # noqa: E402

name, dtype = sys.argv[1], sys.argv[2]
mode = int(sys.argv[3]) if len(sys.argv) > 3 else None
extra = sys.argv[4:]
model = {"quadruped_rigid": bench.build_quadruped_rigid, "humanoid_relaxed": bench.build_humanoid_relaxed}.get(name, lambda: bench.build_model(name))()
mode = sp.mode_of(model) if mode is None else mode
text = sp.spec(model, np.dtype(dtype), mode)
head, assign = text.rsplit(";", 1)
fields = dict(kv.split("=") for kv in head.split(";"))
out = pathlib.Path(f"/tmp/spec_{name}_{dtype}_{mode}.s")
flags = [f for f in sp._flags() if f not in ("-shared", "-fPIC")]
cmd = [sp._HIPCC, *flags, *extra, f"-DJXS_SPEC_T={fields['T']}", f"-DJXS_SPEC_G={fields['G']}", f"-DJXS_SPEC_MODE={fields['MODE']}",
       f"-DJXS_SPEC_ASSIGN={assign}", f'-DJXS_SPEC_STRING="{text}"', "--cuda-device-only", "-S", "jxs_spec.hip", "-o", str(out)]
subprocess.run(cmd, cwd=sp._CSRC, check=True)
s = out.read_text()
for m in re.finditer(r"; -- Begin function (\S+)\n(.*?)(?=; -- Begin function|\Z)", s, re.S):
    fn, body = m.group(1), m.group(2)
    if "s_endpgm" not in body:
        continue
    code = body.split("s_endpgm")[0]
    ops = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)", code, re.M))
    meta = {k: re.search(rf"; {k}: (\d+)", body).group(1) for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize") if re.search(rf"; {k}: (\d+)", body)}
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    print(fn, meta)
    print("  VALU", valu, " pk", sum(v for k, v in ops.items() if k.startswith("v_pk_")), " SALU", sum(v for k, v in ops.items() if k.startswith("s_")),
          " ds", sum(v for k, v in ops.items() if k.startswith("ds_")), " global", sum(v for k, v in ops.items() if k.startswith("global_")))
    print("  " + "  ".join(f"{k} {v}" for k, v in ops.most_common(40)))
