"""round 5, call O (CPU side): config 5 with and without the warm start of the interior-point iteration"""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "{root}/tests")
import helpers, bench
from jaxsim_amd import specialize
zoo = helpers.ModelZoo()
for model in (helpers.rigid_model(zoo("anymal"), helpers.ANYMAL_FEET_4, K=1e4, D=2e2), bench.build_quadruped_rigid()):
    for mode in (specialize.MODE_STEP_RIGID, specialize.MODE_GRAV):
        print(specialize.compile(model, np.float32, mode).name)
'''
procs = [subprocess.Popen([sys.executable, "-c", CHILD.format(root=ROOT)], env=dict(os.environ, **e)) for e in ({}, {"JXS_DISABLE_QP_WARM": "1"})]
print([p.wait() for p in procs])
