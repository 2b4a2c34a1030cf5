"""round 5 (CPU side): the -DJXS_PHASE_TIMING builds of the specialised kernels tools/profile_round.sh stamps, so that the
GPU box does not spend its minutes compiling them."""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "{root}/tests")
import helpers, bench
from jaxsim_amd import specialize
zoo = helpers.ModelZoo()
todo = [(bench.build_model("icub23"), 0)]
for pts, kind in ((4, "rigid"), (16, "rigid"), (16, "relaxed"), (32, "relaxed")):
    robot = "icub" if pts == 32 else "anymal"
    idx = helpers.ANYMAL_FEET_4 if pts == 4 else helpers.ANYMAL_FEET_16 if pts == 16 else list(range(32))
    todo.append(((helpers.rigid_model(zoo(robot), idx, K=1e4, D=2e2) if kind == "rigid" else helpers.relaxed_model(zoo(robot), idx, mu=0.5)), 6))
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(8) as ex:
    print(list(ex.map(lambda mm: specialize.compile(mm[0], np.float32, mm[1]).name, todo)))
'''
subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=dict(os.environ, JAXSIM_AMD_SPEC_EXTRA_FLAGS="-DJXS_PHASE_TIMING"), check=True)
