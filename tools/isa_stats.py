#!/usr/bin/env python3
"""Instruction statistics of the gfx950 kernels of one translation unit (developer aid).
    python tools/isa_stats.py float 0 [G]      -> compiles jxs_inst.hip for (dtype, mode) to assembly and counts"""
import os
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
dtype, mode = sys.argv[1], sys.argv[2]
only_g = sys.argv[3] if len(sys.argv) > 3 else None
out = pathlib.Path(f"/tmp/inst_{dtype}_{mode}.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *os.environ.get("JXS_ISA_FLAGS", "-fno-slp-vectorize").split(), "-mllvm",
                "-amdgpu-kernarg-preload-count=16", "-Wno-cuda-compat", "-Wno-pass-failed", f"-DJXS_INST_T={dtype}", f"-DJXS_INST_MODE={mode}",
                "--cuda-device-only", "-S", "jxs_inst.hip", "-o", str(out)], cwd=ROOT / "jaxsim_amd" / "csrc", check=True,
               stderr=subprocess.DEVNULL)
s = out.read_text()
for m in re.finditer(r"; -- Begin function (\S+)\n(.*?)(?=; -- Begin function|\Z)", s, re.S):
    name, body = m.group(1), m.group(2)
    g = re.search(r"I[fd]Li(\d+)ELi(\d+)E", name)
    if not g or (only_g and g.group(1) != only_g):
        continue
    code = body.split("s_endpgm")[0]
    cnt = lambda pat: len(re.findall(pat, code))  # noqa: E731
    meta = {k: re.search(rf"; {k}: (\d+)", body).group(1) for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize") if re.search(rf"; {k}: (\d+)", body)}
    pats = dict(valu=r"\n\s+v_", dpp=r"_dpp ", pk=r"v_pk_", salu=r"\n\s+s_", gl1=r"global_load_dword ", gl2=r"global_load_dwordx2", gl3=r"global_load_dwordx3",
                gl4=r"global_load_dwordx4", gst=r"global_store", dr1=r"ds_read_b32", dr2=r"ds_read_b64|ds_read2_b32", dr4=r"ds_read_b128|ds_read2_b64",
                dw1=r"ds_write_b32", dw2=r"ds_write_b64|ds_write2_b32", dw4=r"ds_write_b128|ds_write2_b64", bperm=r"ds_bpermute", scratch=r"scratch_", wait=r"s_waitcnt")
    c = {k: cnt(v) for k, v in pats.items()}
    print("G=%s mode=%s: VALU %d (dpp %d, pk %d) SALU %d gload %d/%d/%d/%d gstore %d ds_read %d/%d/%d ds_write %d/%d/%d bperm %d scratch %d waitcnt %d %s" % (
        g.group(1), g.group(2), c["valu"], c["dpp"], c["pk"], c["salu"], c["gl1"], c["gl2"], c["gl3"], c["gl4"], c["gst"], c["dr1"], c["dr2"], c["dr4"],
        c["dw1"], c["dw2"], c["dw4"], c["bperm"], c["scratch"], c["wait"], meta))
