"""Model-specialised kernels (jaxsim_amd/specialize.py): description, cache, build, and on the GPU the
specialised kernel against the generic one of the same library."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

import helpers
import jaxsim_amd.api as js
from jaxsim_amd import _lib, runtime, specialize

HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


@pytest.fixture(scope="module")
def models():
    return helpers.ModelZoo()


def test_description_names_topology_not_physics(models):
    m = models("icub")
    a = specialize.spec(m, np.float32)
    assert a.startswith("T=float;G=32;MODE=0;") and "P.floating=1" in a and "P.row_mode=1" in a
    # physical parameters are run-time data: another time step / contact stiffness / mass is the same kernel
    b = specialize.spec(helpers.with_params(m, K=2.0 * float(m.contact_params.K)), np.float32)
    assert a == b
    assert specialize.spec(m, np.float64).startswith("T=double;")
    # another tree is another kernel
    assert specialize.spec(models("anymal"), np.float32) != a
    assert specialize.path_of(a) != specialize.path_of(specialize.spec(m, np.float64))


def test_mode_follows_the_contact_model_and_integrator(models):
    assert specialize.mode_of(models("icub")) == specialize.MODE_STEP
    assert specialize.modes_of(models("icub")) == [specialize.MODE_STEP, specialize.MODE_ROLLOUT]
    rigid = helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_4)
    assert specialize.mode_of(rigid) == specialize.MODE_STEP_RIGID
    import jaxsim_amd as ja

    rk4 = helpers.with_params(models("icub"), integrator=ja.IntegratorType.RungeKutta4)
    assert specialize.modes_of(rk4) == [specialize.MODE_STEP_RK4]
    assert specialize.spec(rigid, np.float32, specialize.MODE_STEP_RIGID).count("P.rg_merge=1") == 1


def _flags(model, dtype=np.float32):
    text = specialize.spec(model, dtype, specialize.mode_of(model))
    return {k[2:]: int(v.rstrip("ul"), 0) for k, v in (kv.split("=") for kv in text.rsplit(";", 1)[1].split(","))}


def test_description_carries_what_the_round_3_kernels_branch_on(models):
    """Packer-derived constants of the round-3 step kernels (jxs_pack.h): DPP-reachable children, lane-ordered rows,
    and the features whose terms a model-specialised kernel leaves out."""
    icub, anymal, cartpole, pendulum = (_flags(models(n)) for n in ("icub", "anymal", "cartpole", "pendulum"))
    # no prismatic joint in the humanoid / quadruped, one in the cartpole
    assert (icub["any_pri"], anymal["any_pri"], cartpole["any_pri"]) == (0, 0, 1)
    # floating-base URDF models carry no base-link offset; the fixed-base pendulum of the zoo does
    assert (icub["has_base_off"], anymal["has_base_off"], pendulum["has_base_off"]) == (0, 0, 1)
    # the quadruped's base finds its second, third and fourth leg 4, 7 and 10 lanes up (DFS lane order, three links per leg)
    assert anymal["child_off"] == 0xA74
    # the humanoid: the root's third child sits 13 lanes up in its own 16-lane row; its second children differ between
    # the root and the chest, so k = 1 stays a shuffle
    assert icub["child_off"] == 0xD0
    # row layout of the humanoid: two of the three extra children sit in the slot next to their parent's
    assert bin(icub["row_pull_dpp"]).count("1") == 2 and icub["row_mode"] == 1
    # points of the humanoid are numbered in slot order, its joints are not in lane order; a chain is
    assert (icub["prow_seq"], icub["jrow_seq"], cartpole["jrow_seq"]) == (1, 0, 1)
    # the knobs switch the features off (A/B on a GPU box: tools/gpu/r03_env_ab.sh)
    for knob, key in (("JXS_DISABLE_CHILD_DPP", "child_off"), ("JXS_DISABLE_PULL_DPP", "row_pull_dpp")):
        os.environ[knob] = "1"
        try:
            assert _flags(models("anymal" if key == "child_off" else "icub"))[key] == 0
        finally:
            del os.environ[knob]

@pytest.mark.skipif(HIPCC is None, reason="hipcc not installed")
def test_build_produces_the_two_entry_points(models, tmp_path, monkeypatch):
    monkeypatch.setattr(specialize, "CACHE", tmp_path)
    m = models("cartpole")
    assert specialize.cached(m, np.float32) is None
    so = specialize.compile(m, np.float32)
    assert so.parent == tmp_path and specialize.cached(m, np.float32) == so
    lib = C.CDLL(str(so))
    lib.jxs_spec_string.restype = C.c_char_p
    assert lib.jxs_spec_string().decode() == specialize.spec(m, np.float32)
    assert hasattr(lib, "jxs_spec_launch")
    # the kernel lives in its own namespace: the generic instantiation of libjaxsim_amd.so (same template
    # arguments) is another symbol and cannot be bound across the two objects
    import subprocess

    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    assert "jxs_launch_spec" in syms and "_ZN10jxs_launch10jxs_kernel" not in syms


def _step_n(model, data, n):
    for _ in range(n):
        data = js.model.step(model, data)
    return data.state_block()


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("icub", np.float32), ("icub", np.float64), ("anymal", np.float32), ("cartpole", np.float32),
                                        ("double_pendulum", np.float64)])  # fmt: skip
def test_specialised_step_equals_the_generic_kernel(models, name, dtype, monkeypatch):
    model = models(name)
    d = models.random_data(name, 37, seed=3, dtype=dtype)
    block = helpers.odata_to_block(model, d)
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    model.__dict__.pop("_device", None)
    ref = _step_n(model, js.data.JaxSimModelData.from_state_block(model, block, 2), 3)
    assert specialize.modes(runtime.device_model(model, dtype)) == []
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "1")
    model.__dict__.pop("_device", None)
    out = _step_n(model, js.data.JaxSimModelData.from_state_block(model, block, 2), 3)
    assert specialize.modes(runtime.device_model(model, dtype)) == [specialize.MODE_STEP, specialize.MODE_ROLLOUT]
    model.__dict__.pop("_device", None)
    # the same arithmetic with the branches folded: identical up to the contraction choices of the compiler (three steps
    # of contact-rich fp32 states: measured 2e-5 .. 1e-4 depending on the build; the per-step gate against the oracle is 1.5e-3)
    tol = 1e-12 if dtype == np.float64 else 2e-4
    assert helpers.rel_err(out, ref) < tol


@pytest.mark.gpu
def test_specialised_rigid_step_equals_the_generic_kernel(models, monkeypatch):
    model = helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_4)
    d = helpers.standing_data(model, 24, seed=1, dtype=np.float64)
    block = helpers.odata_to_block(model, d)
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    model.__dict__.pop("_device", None)
    ref = _step_n(model, js.data.JaxSimModelData.from_state_block(model, block, 2), 2)
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "1")
    model.__dict__.pop("_device", None)
    out = _step_n(model, js.data.JaxSimModelData.from_state_block(model, block, 2), 2)
    assert specialize.modes(runtime.device_model(model, np.float64)) == [specialize.MODE_STEP_RIGID]
    model.__dict__.pop("_device", None)
    assert helpers.rel_err(out, ref) < 1e-9


@pytest.mark.gpu
def test_attach_refuses_a_kernel_built_for_another_model(models, monkeypatch):
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    icub, anymal = models("icub"), models("anymal")
    so = specialize.compile(anymal, np.float32)
    icub.__dict__.pop("_device", None)
    dm = runtime.device_model(icub, np.float32)
    rc = _lib.load().jxs_model_attach_specialized(dm.handle, specialize.MODE_STEP, str(so).encode())
    assert rc != 0 and specialize.modes(dm) == []
    icub.__dict__.pop("_device", None)


@pytest.mark.gpu
def test_specialised_query_kernels_equal_the_generic_ones(models, monkeypatch):
    """forward / inverse dynamics, mass matrix and its inverse through kernels specialised on the model."""
    model = models("anymal")
    d = models.random_data("anymal", 19, seed=2, dtype=np.float64)
    block = helpers.odata_to_block(model, d)

    def queries():
        data = js.data.JaxSimModelData.from_state_block(model, block, 2)
        a, sdd = js.model.forward_dynamics_aba(model, data)
        tau = js.model.inverse_dynamics(model, data, joint_accelerations=np.asarray(sdd), base_acceleration=np.asarray(a))
        return [np.asarray(x) for x in (a, sdd, tau[0], tau[1], js.model.free_floating_mass_matrix(model, data),
                                        js.model.free_floating_mass_matrix_inverse(model, data))]  # fmt: skip

    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    model.__dict__.pop("_device", None)
    ref = queries()
    assert js.model.specialize(model, np.float64, queries=True)
    assert set(specialize.QUERY_MODES) <= set(specialize.modes(runtime.device_model(model, np.float64)))
    out = queries()
    model.__dict__.pop("_device", None)
    for x, y in zip(out, ref):
        assert helpers.rel_err(x, y) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n_links,fixed,max_back,seed", [(3, True, 1, 11), (7, False, 1, 12), (12, False, 3, 13), (20, True, 2, 14),
                                                         (33, False, 4, 15), (9, False, 8, 16)])  # fmt: skip
def test_specialised_random_trees_equal_the_generic_kernel(n_links, fixed, max_back, seed, monkeypatch):
    """Random trees (serial chains, bushy trees, fixed and floating bases, with and without contact points, both
    ABA layouts): the constant folding of a specialised build must not change what the kernel computes."""
    import jaxsim_amd as ja
    from jaxsim_amd import robots

    model = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(n_links, fixed_base=fixed, seed=seed, max_back=max_back))
    data0 = js.data.random_model_data(model, batch_size=13, seed=seed, dtype=np.float32)
    block = data0.state_block()
    rng = np.random.default_rng(seed)
    tau = rng.uniform(-1, 1, (13, model.dofs())).astype(np.float32)

    def run():
        # one step: these random trees in random states are violent (a state of 1e7 after three steps of the
        # 33-link tree), further steps only measure how chaos amplifies the last bit
        data = js.data.JaxSimModelData.from_state_block(model, block, 2)
        a, sdd = js.model.forward_dynamics_aba(model, data, joint_forces=tau)
        data = js.model.step(model, data, joint_force_references=tau)
        return [data.state_block(), np.asarray(a), np.asarray(sdd)]

    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    model.__dict__.pop("_device", None)
    ref = run()
    assert js.model.specialize(model, np.float32, queries=True)
    out = run()
    model.__dict__.pop("_device", None)
    for x, y in zip(out, ref):
        assert np.isfinite(y).all()
        # fp32 rounding with other contraction choices; deep random chains amplify it (they are 2e-2 from the fp64
        # oracle, tests/helpers.py) -- a folded branch gone wrong would be O(1)
        assert helpers.rel_err(x, y) < 5e-4


def test_default_policy_builds_on_first_use_when_hipcc_is_present(monkeypatch):
    """[round 3] What a drop-in user gets: without any environment variable the first step of a model compiles its
    specialised kernel when the compiler is there (jax.jit semantics), and only looks into the cache otherwise."""
    monkeypatch.delenv("JAXSIM_AMD_SPECIALIZE", raising=False)
    assert specialize.policy() == ("build" if specialize.hipcc_available() else "cached")
    monkeypatch.setattr(specialize, "_HIPCC", "/nonexistent/hipcc")
    assert specialize.policy() == "cached"
    for value, want in (("0", "off"), ("1", "build"), ("cached", "cached")):
        monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", value)
        assert specialize.policy() == want


@pytest.mark.gpu
def test_specialised_object_is_mapped_while_and_after_a_model_runs(models, monkeypatch):
    """The specialised kernel object the step runs through is visible in /proc/self/maps -- while the device model is
    alive and afterwards (objects are never unloaded: launches may still be in flight, jxs_api.hip ~ModelT) -- so a
    driver that records which native code a process loaded sees spec_cache/libjxs_spec_*.so, not only the library."""
    import gc

    def mapped():
        return {ln.split()[-1] for ln in open("/proc/self/maps") if "libjxs_spec_" in ln}

    model = models("icub")
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "1")
    model.__dict__.pop("_device", None)
    d = models.random_data("icub", 8, seed=1, dtype=np.float32)
    out = js.model.step(model, js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), 2))
    dm = runtime.device_model(model, np.float32)
    files = specialize.attached_files(dm)
    assert specialize.MODE_STEP in specialize.modes(dm) and files[specialize.MODE_STEP].startswith("libjxs_spec_")
    want = str((specialize.CACHE / files[specialize.MODE_STEP]).resolve())
    assert want in mapped(), (want, mapped())
    assert np.isfinite(out.state_block()).all()
    del dm, out
    model.__dict__.pop("_device", None)
    gc.collect()
    assert want in mapped()


@pytest.mark.gpu
def test_specialised_rk4_rigid_kernel_is_deterministic_gpu(models, monkeypatch):
    """RungeKutta4 + RigidContacts on the quadruped AS A MODEL-SPECIALISED KERNEL (the default experience with hipcc;
    the rest of the suite pins the ahead-of-time kernels for most models): finite, the same bits on every call, and
    the library's own kernel within fp32 rounding.  This kernel exposed a hardware hazard of round 3's DPP child
    gathers -- a DPP operand right behind the scalar instruction that re-enables lanes still saw them switched off
    (jxs_lanes_device.h exec_settle): results were non-finite in some environments on some calls."""
    import jaxsim_amd as ja

    model = helpers.with_params(helpers.rigid_model(models("anymal"), helpers.ANYMAL_FEET_4, K=1e4, D=2e2), integrator=ja.IntegratorType.RungeKutta4)
    d = models.random_data("anymal", 21, seed=5, dtype=np.float32)
    blk = helpers.odata_to_block(model, d)

    def step():
        return js.model.step(model, js.data.JaxSimModelData.from_state_block(model, blk.copy())).state_block()

    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "0")
    model.__dict__.pop("_device", None)
    generic = step()
    monkeypatch.setenv("JAXSIM_AMD_SPECIALIZE", "1" if HIPCC else "cached")  # (pre-built by __graft_entry__.build())
    model.__dict__.pop("_device", None)
    try:
        if specialize.modes(runtime.device_model(model, np.float32)) != [specialize.MODE_STEP_RK4_RIGID]:
            pytest.skip("no specialised RungeKutta4 + RigidContacts kernel in the cache and no hipcc")
        first = step()
        assert np.isfinite(first).all()
        for _ in range(8):
            np.testing.assert_array_equal(step(), first)
        assert helpers.rel_err(first, generic) < 5e-4
    finally:
        model.__dict__.pop("_device", None)
