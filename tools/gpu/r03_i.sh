#!/usr/bin/env bash
# round 3, GPU call I: gravity-torque kernel (MODE_GRAV), config 5 loop, parity suite
set -u
R=$PWD
OUT=$R/gpurun_out/r03_j
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
for a in "" "--standing" "--envs 16384"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | sed "s/^/$a: /" | tee -a "$OUT/c5.txt"
done
timeout 300 python - <<'PY' 2>&1 | tee -a "$OUT/c5.txt"
import ctypes as C, sys, numpy as np
sys.path.insert(0, "tests")
import helpers, jaxsim_amd.api as js
from jaxsim_amd import _lib, runtime, specialize
zoo = helpers.ModelZoo()
model = helpers.rigid_model(zoo("anymal"), helpers.ANYMAL_FEET_4, K=1e4, D=2e2)
N = 4096
d = zoo.random_data("anymal", N, seed=0, dtype=np.float32)
data = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), 2)
lib = _lib.load(); stream = runtime.Stream(); runtime.set_stream(stream)
for spec in (False, True):
    model.__dict__.pop("_device", None)
    dm = runtime.device_model(model, np.float32)
    if spec: specialize.ensure_mode(dm, model, specialize.MODE_GRAV)
    tau = runtime.DeviceArray(model.dofs(), N, np.float32, tile=data._state.tile, zero=True)
    st, tp = C.c_void_p(data._state.ptr), C.c_void_p(tau.ptr)
    def run(k):
        for _ in range(k): _lib.check(lib.jxs_gravity_torques(dm.handle, st, tp, N, stream.handle), "g")
    run(50); stream.synchronize()
    e0, e1 = runtime.Event(), runtime.Event(); e0.record(stream); run(500); e1.record(stream); stream.synchronize()
    print(f"gravity torques N={N} specialised={spec} modes={specialize.modes(dm)}: {e0.elapsed_ms(e1)/500*1e3:.2f} us per launch")
PY
