"""round 5, call F (CPU side): the specialised kernels of tools/gpu/r05_f.sh (config 5 with and without the block skip)."""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "{root}/tests")
import helpers
from jaxsim_amd import specialize
zoo = helpers.ModelZoo()
model = helpers.rigid_model(zoo("anymal"), helpers.ANYMAL_FEET_4, K=1e4, D=2e2)
for mode in (specialize.MODE_STEP_RIGID, specialize.MODE_GRAV):
    print(specialize.compile(model, np.float32, mode).name)
'''
procs = [subprocess.Popen([sys.executable, "-c", CHILD.format(root=ROOT)], env=dict(os.environ, JAXSIM_AMD_SPEC_EXTRA_FLAGS=f)) for f in ("", "-DJXS_NO_QP_BLOCK_SKIP")]
print([p.wait() for p in procs])
