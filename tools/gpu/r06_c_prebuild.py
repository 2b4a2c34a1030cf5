"""round 6 (CPU side): the specialised kernels bench.py asks for (headline humanoid in both precisions, rollout, the
secondary contact models, the quadruped's gravity torques), so that the GPU box does not spend its minutes compiling them."""
import pathlib, sys
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
from jaxsim_amd import specialize
todo = [(bench.build_model("icub23"), np.float32), (bench.build_model("icub23"), np.float64)]
if "--all" in sys.argv:
    todo += [(m, dt) for m, dt in bench.secondary_models()]
texts = {specialize.spec(model, dt, mode) for model, dt in todo for mode in specialize.modes_of(model)}
if "--all" in sys.argv:
    texts.add(specialize.spec(bench.build_quadruped_rigid(), np.float32, specialize.MODE_GRAV))
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(8) as ex:
    for p in ex.map(specialize.compile_text, sorted(texts)):
        print(p.name)
