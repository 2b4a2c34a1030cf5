"""[round 4] Bitwise determinism of every step kernel: the same input gives the same bits on every call.

Round 3 shipped a kernel (RungeKutta4 + RigidContacts, model-specialised) that returned different results from call to
call, and blamed a hardware hazard.  The micro-benchmark tools/ubench/exec_dpp.hip showed the hazard does not exist, and
the bisect profiles/r04_masked_lds_write_bisect.md that the cause was the compiler's freedom around exec-masked LDS writes
(csrc/jxs_lanes_device.h lds_publish).  This sweep is the regression net: every zoo model x {semi-implicit Euler, RungeKutta4,
RigidContacts, RelaxedRigidContacts, RungeKutta4 with either} x {fp32, fp64} x {the library's kernels, its run-time-flag
kernel, the model-specialised kernel} -- eight calls each, with another model's kernel in between (what is left in
the LDS and the registers of a CU by the previous launch must not matter), compared bit for bit.
"""
import os

import numpy as np
import pytest

import helpers
import jaxsim_amd as ja
import jaxsim_amd.api as js
from test_gpu_parity import RELAXED_CASES, RIGID_CASES, _rk4, to_gpu

pytestmark = pytest.mark.gpu

ZOO = ["box", "sphere", "pendulum", "double_pendulum", "cartpole", "chain5", "chain9f", "anymal", "icub", "icub16", "planar_biped"]
RK4_SOFT = ["box", "sphere", "cartpole", "chain9f", "anymal", "icub16", "icub"]  # (the models test_rk4_step_matches_oracle_gpu steps)
RK4_RIGID = ["box4", "anymal4", "icub8"]
RK4_RELAXED = ["box8", "anymal16", "chain9f6", "icub16", "planar_biped"]


def _cases():
    out = [(f"euler-{n}", "euler", n) for n in ZOO]
    out += [(f"rk4-{n}", "rk4", n) for n in RK4_SOFT]
    out += [(f"rigid-{k}", "rigid", k) for k in RIGID_CASES]
    out += [(f"relaxed-{k}", "relaxed", k) for k in RELAXED_CASES]
    out += [(f"rk4_rigid-{k}", "rk4_rigid", k) for k in RK4_RIGID]
    out += [(f"rk4_relaxed-{k}", "rk4_relaxed", k) for k in RK4_RELAXED]
    return out


def _build(models, kind, key):
    """(model, zoo name of the model: the random states come from its generator)"""
    if kind == "euler":
        return models(key), key
    if kind == "rk4":
        return _rk4(models(key)), key
    table, make = (RIGID_CASES, helpers.rigid_model) if "rigid" in kind else (RELAXED_CASES, helpers.relaxed_model)
    name, idx, params = table[key]
    model = make(models(name), idx, **params)
    if kind.startswith("rk4"):
        model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4)
    return model, name


@pytest.fixture(scope="module")
def disturbance(models):
    """A launch of ANOTHER kernel between two calls: the soft-contact step of the humanoid over 1024 environments
    occupies every CU and leaves its own data in their LDS and registers."""
    model = models("icub")
    blk = helpers.odata_to_block(model, models.random_data("icub", 1024, seed=1, dtype=np.float32))

    def run():
        x = js.data.JaxSimModelData.from_state_block(model, blk.copy())
        for _ in range(2):
            x = js.model.step(model, x)
        return x

    return run


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["fp32", "fp64"])
@pytest.mark.parametrize("case", _cases(), ids=[c[0] for c in _cases()])
def test_step_is_bitwise_deterministic(models, disturbance, kernel_policy, knobs, case, dtype):
    _, kind, key = case
    model, name = _build(models, kind, key)
    N = 21 if kind != "euler" else 70
    d = models.random_data(name, N, seed=5, dtype=dtype)
    tau, f = helpers.random_inputs(model, N, 7, dtype)
    # the third kernel variant of the soft-contact step: every model flag read at run time (KV_GENERIC) instead of the
    # common-feature variant the library picks for floating-base row-layout models
    variants = [None] + (["run_time_flags"] if (kernel_policy == "library" and kind == "euler") else [])
    for variant in variants:
        if variant is not None:
            knobs("JXS_DISABLE_COMMON_VARIANT", 1)
        first = None
        for rep in range(8):
            if rep % 2 == 1:
                disturbance()
            out = js.model.step(model, to_gpu(model, d), link_forces=f, joint_force_references=tau).state_block()
            if first is None:
                first = out
                assert out.dtype == dtype
            else:
                np.testing.assert_array_equal(out, first, err_msg=f"{case[0]} {np.dtype(dtype).name} {variant or kernel_policy}: call {rep} differs from call 0")
    if kernel_policy == "specialised" and not os.environ.get("JAXSIM_AMD_TEST_RECORD"):
        from jaxsim_amd import runtime, specialize

        assert set(specialize.modes_of(model)) <= set(specialize.modes(runtime.device_model(model, dtype)))
