"""round 5, call B (CPU side): pre-build the model-specialised kernels of tools/gpu/r05_b.sh so that the GPU box spends
its minutes measuring.  One process per knob set (the packer reads the developer knobs from the environment)."""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
CASES = [  # (points, contact, extra env)
    (32, "relaxed", {}), (32, "relaxed", {"JXS_PREFER_LINKSPACE": "1"}), (32, "relaxed", {"JXS_DISABLE_CT_TREE": "1", "JXS_DISABLE_LINKSPACE": "1"}),
    (16, "relaxed", {}), (16, "relaxed", {"JXS_DISABLE_CT_TREE": "1"}),
    (4, "relaxed", {}), (4, "relaxed", {"JXS_DISABLE_CT_TREE": "1"}),
]
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "{root}/tests")
import helpers, jaxsim_amd.api as js
from jaxsim_amd import specialize
points = {points}
zoo = helpers.ModelZoo()
robot = "icub" if points == 32 else "anymal"
idx = helpers.ANYMAL_FEET_4 if points == 4 else helpers.ANYMAL_FEET_16 if points == 16 else list(range(32))
model = helpers.relaxed_model(zoo(robot), idx)
model = helpers.with_params(model, contact_params=js.contact.estimate_good_contact_parameters(model))
for mode in (specialize.MODE_STEP_RIGID, specialize.MODE_GRAV):
    print(specialize.compile(model, np.float32, mode).name, specialize.spec(model, np.float32, mode)[-60:])
'''
procs = []
for points, contact, env in CASES:
    e = dict(os.environ, **env)
    procs.append(subprocess.Popen([sys.executable, "-c", CHILD.format(root=ROOT, points=points)], env=e))
rc = [p.wait() for p in procs]
print("rc", rc)
