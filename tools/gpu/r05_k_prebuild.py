"""round 5, call K (CPU side): kernels of tools/gpu/r05_k.sh"""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "{root}/tests")
import helpers, jaxsim_amd as ja, jaxsim_amd.api as js
from jaxsim_amd import specialize, robots
zoo = helpers.ModelZoo()
which = "{which}"
if which == "rigid32":
    model, dt = helpers.rigid_model(zoo("icub"), list(range(32)), K=1e4, D=2e2), np.float64
else:
    base = ja.JaxSimModel.build_from_model_description(robots.anymal12_urdf(foot_shape="sphere"))
    model = helpers.relaxed_model(base, list(range(200)))
    model, dt = helpers.with_params(model, contact_params=js.contact.estimate_good_contact_parameters(model)), np.float32
for mode in (specialize.MODE_STEP_RIGID, specialize.MODE_GRAV):
    print(which, specialize.compile(model, dt, mode).name)
'''
procs = [subprocess.Popen([sys.executable, "-c", CHILD.format(root=ROOT, which=w)], env=dict(os.environ, **e)) for w, e in (("rigid32", {}), ("q200", {}), ("q200", {"JXS_MIN_LANES": "64"}))]
print([p.wait() for p in procs])
