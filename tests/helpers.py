"""Shared helpers of the test-suite (TEST INFRASTRUCTURE)."""

from __future__ import annotations

import dataclasses

import numpy as np

import jaxsim_amd as ja
import oracle
from jaxsim_amd import robots
from jaxsim_amd import state as st

# Stated tolerances (north_star: "within a stated fp32/fp64 tolerance"), metric = rel_err below
# (max |a - ref| / max(1, |ref|), element-wise, worst element of the whole batch).  The truth is
# ALWAYS the fp64 oracle evaluated on the same (already rounded) inputs:
#   fp64 kernels: 1e-10 (measured 1e-13);
#   fp32 kernels, one step / one evaluation: 1e-3 WORST element of the worst environment, with the
#   distribution checked at full size (test_full_size_step_*: median < 3e-6, 99th percentile < 3e-4 of the
#   per-environment error).  Measured on MI355X with the anchored ABA (tools/fp32_error_gpu.py, 512 humanoid
#   states with the reference's default K = 1e6 and centimetres of penetration, i.e. contact accelerations
#   of 1e4..1e5 m/s^2): median 1.4e-6, p99 1.4e-4, worst 1.4e-4 .. 5.7e-4 depending on the seed; the
#   reference formulation itself, run in fp32 (the oracle with float32 arrays), has median 2.8e-7, p99 7e-5,
#   worst 6e-5 .. 7e-4 on the same states -- the worst case is the conditioning of the stiff-contact states,
#   not of either formulation.  Round 1 (one reference point for the whole tree): median 2.1e-5, worst 1.2e-3,
#   tolerance 3e-3.
#   `chain9f` (a random 9-link floating chain with collidable points on light links): the reference
#   formulation in fp32 is itself 3e-3 .. 7e-3 away from its fp64 result in the worst environment (median
#   9e-6) -- a noise-limited model kept as a stress case: 2e-2.
# The reference calls its own 32-bit mode "still experimental" (src/jaxsim/__init__.py:37-41); the fp64
# kernels are exact to rounding.
# [round 4] The fp32 gates are PER MODEL: three times the worst element error measured on MI355X over 2 x 512 random
# states per model (profiles/r04_fp32_error_gpu.txt; tools/fp32_error_gpu.py regenerates the table) against the fp64
# oracle on the same STATE -- caches recomputed in fp64, see `upcast` below; round 3 compared against a truth that carried
# the fp32 rounding of the cached kinematics and had gates 3 .. 9 times wider for the contact models:
#   model            measured worst    gate      reference formulation in fp32 (worst), same truth
#   pendulum         1.2e-7            5e-7      7.8e-8
#   double_pendulum  1.2e-7            3e-7 (*)  7.8e-8
#   cartpole         1.2e-7            3e-7 (*)  7.8e-8
#   chain5           3.2e-7            1.5e-6    1.2e-7
#   sphere           1.7e-4            5.5e-4    2.7e-4
#   box              1.8e-4            5.5e-4    7.6e-4
#   anymal           3.6e-5 (1.2e-4)   3.5e-4    3.7e-4    (1.2e-4: the actuation-limit states of test_actuation_limits_...; an
#                                                          environment on an edge of the contact model reaches 2.0e-4 at
#                                                          N = 1024, where the fp64 oracle itself moves by as much under
#                                                          one ulp of input noise: helpers.oracle_sensitivity)
#   icub / icub16    2.1e-4 / 1.2e-4   6.5e-4 / 4e-4   6.7e-4 / 4.9e-4
#   chain9f          3.6e-3            1.1e-2    1.4e-3   (a noise-limited random chain kept as the stress case)
# (*) kept from round 3 (2.5 x).  FP32_TOL (1e-3) remains the gate of models without an entry and the ceiling of all.
FP64_TOL = 1e-10
FP32_TOL = 1e-3
FP32_TOL_BY_MODEL = {
    "pendulum": 5e-7, "double_pendulum": 3e-7, "cartpole": 3e-7, "chain5": 1.5e-6, "sphere": 5.5e-4, "box": 5.5e-4,
    "anymal": 3.5e-4, "icub": 6.5e-4, "icub16": 4e-4, "chain9f": 1.1e-2,
}  # fmt: skip


class ModelZoo:
    """Lazily built host models used across the tests."""

    _urdf = {
        "box": lambda: robots.box_urdf(),
        "sphere": lambda: robots.sphere_urdf(),
        "pendulum": lambda: robots.single_pendulum_urdf(),
        "double_pendulum": lambda: robots.double_pendulum_urdf(),
        "cartpole": lambda: robots.cartpole_urdf(),
        "chain5": lambda: robots.chain_urdf(5, fixed_base=True, seed=1),
        "chain9f": lambda: robots.chain_urdf(9, fixed_base=False, seed=2),
        # a serial floating chain: its two collision boxes (first and last link) are eleven joints apart
        "serial12f": lambda: robots.chain_urdf(12, fixed_base=False, seed=4, max_back=1),
        # [round 5] every joint axis parallel: the relative twist of any two links spans three dimensions, whatever the number
        # of joints between them (a Walker2d-style planar biped with a floating base; a planar serial chain)
        "planar_biped": lambda: robots.planar_biped_urdf(),
        "planar10f": lambda: robots.chain_urdf(10, fixed_base=False, seed=6, max_back=1, parallel_axes="all"),
        # [round 6] more than six children on one link: an octopod (8 legs of 2 links, four feet) and a 12-spoke hub
        "octopod": lambda: robots.hub_urdf(8, 2, foot_boxes=4, seed=1),
        "hub12": lambda: robots.hub_urdf(12, 1, foot_boxes=2, seed=2),
        "anymal": lambda: robots.anymal12_urdf(),
        "icub": lambda: robots.icub23_urdf(),
        "icub16": lambda: robots.icub23_urdf(sole_boxes_per_foot=1),
    }
    # base height range putting some collidable points in contact
    contact_z = {"box": (0.0, 0.1), "sphere": (0.09, 0.12), "chain9f": (0.0, 0.3), "serial12f": (0.0, 0.3), "anymal": (0.58, 0.70), "octopod": (0.40, 0.55), "hub12": (0.2, 0.32),
                 "planar_biped": (0.78, 0.95), "planar10f": (0.0, 0.3),
                 "icub": (0.56, 0.68), "icub16": (0.56, 0.68)}  # fmt: skip

    def __init__(self):
        self._cache = {}

    def __call__(self, name: str) -> ja.JaxSimModel:
        if name not in self._cache:
            self._cache[name] = ja.JaxSimModel.build_from_model_description(self._urdf[name]())
        return self._cache[name]

    def random_data(self, name, N, seed=0, dtype=np.float64, rep=oracle.VelRepr.Mixed, in_contact=True):
        m = self(name)
        kw = {}
        if in_contact and name in self.contact_z:
            z0, z1 = self.contact_z[name]
            kw = dict(base_pos_bounds=((-1, -1, z0), (1, 1, z1)), base_rpy_bounds=((-0.3, -0.3, -3), (0.3, 0.3, 3)))
        d = oracle.random_model_data(m, batch_size=N, seed=seed, dtype=dtype, velocity_representation=rep, **kw)
        rng = np.random.default_rng(seed + 1000)
        d.tangential_deformation[:] = (1e-3 * rng.normal(size=d.tangential_deformation.shape)).astype(dtype)
        return d


def standing_data(model, N, seed=0, dtype=np.float64, sink=2e-3, noise=0.03):
    """States with (nearly) every bottom collidable point in contact: zero pose plus small noise, the
    base lowered so that the lowest enabled points sit `sink` under the ground, small velocities."""
    rng = np.random.default_rng(seed)
    kdp = model.kin_dyn_parameters
    n = kdp.number_of_joints()
    s = noise * rng.uniform(-1, 1, size=(N, n))
    rpy = 0.3 * noise * rng.uniform(-1, 1, size=(N, 3))
    q = oracle.refmath.quaternion_from_euler_xyz(rpy)
    d0 = oracle.OracleData.build(model, base_quaternion=q, joint_positions=s)
    p, _ = oracle.refstep.collidable_points_pos_vel(model, link_transforms=d0.link_transforms, link_velocities=d0.link_velocities)
    z = -p[:, :, 2].min(axis=1) - sink
    pos = np.stack([rng.uniform(-1, 1, N), rng.uniform(-1, 1, N), z], axis=1)
    return oracle.OracleData.build(
        model, base_position=pos, base_quaternion=q, joint_positions=s, dtype=dtype,
        base_linear_velocity=0.1 * rng.uniform(-1, 1, (N, 3)), base_angular_velocity=0.1 * rng.uniform(-1, 1, (N, 3)),
        joint_velocities=0.2 * rng.uniform(-1, 1, (N, n)),
    )


def odata_to_block(model, d: oracle.OracleData, dtype=None) -> np.ndarray:
    L = st.StateLayout.of(model)
    return st.pack_state(
        L,
        base_position=d.base_position,
        base_quaternion=d.base_quaternion,
        joint_positions=d.joint_positions,
        base_linear_velocity=d.base_linear_velocity,
        base_angular_velocity=d.base_angular_velocity,
        joint_velocities=d.joint_velocities,
        tangential_deformation=d.tangential_deformation,
        dtype=dtype or d.dtype,
    )


def block_to_odata(model, block: np.ndarray, rep=oracle.VelRepr.Mixed) -> oracle.OracleData:
    f = st.unpack_state(st.StateLayout.of(model), block)
    d = oracle.OracleData(
        f["base_position"], f["base_quaternion"], f["joint_positions"], f["base_linear_velocity"],
        f["base_angular_velocity"], f["joint_velocities"], f["tangential_deformation"], velocity_representation=rep,
    )  # fmt: skip
    return d.update_caches(model)


def rel_err(a: np.ndarray, ref: np.ndarray) -> float:
    """max |a-ref| / max(1, |ref|) -- the tolerance metric used by every parity test."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref)))) if a.size else 0.0


def note(key: str, value: float) -> None:
    """Developer aid: append a measured error to ``$JXS_ERR_LOG`` (the per-case gates are set from what the GPU
    run records there: measured worst x 3)."""
    import os

    path = os.environ.get("JXS_ERR_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{key} {value:.3e}\n")


# Single evaluations (forward / inverse dynamics, cached kinematics): the accelerations and torques of the small
# contact-free models are O(10..100) and carry the plain fp32 rounding of the recursion (measured 8e-7 in IEEE
# emulation), which a step multiplies by dt before it reaches the state: their step gates (3e-7) do not apply.
FP32_EVAL_TOL_BY_MODEL = {"pendulum": 1e-5, "double_pendulum": 1e-5, "cartpole": 1e-5, "chain5": 1e-5}


def tol_of(dtype, name: str | None = None, evaluation: bool = False) -> float:
    """Stated tolerance of a step (default) or of a single evaluation (forward / inverse dynamics, kinematics)."""
    if np.dtype(dtype) == np.float64:
        return FP64_TOL
    if evaluation and name in FP32_EVAL_TOL_BY_MODEL:
        return FP32_EVAL_TOL_BY_MODEL[name]
    return FP32_TOL_BY_MODEL.get(name, FP32_TOL)


def random_inputs(model, N, seed, dtype):
    rng = np.random.default_rng(seed)
    n, nL = model.dofs(), model.number_of_links()
    tau = rng.uniform(-5, 5, size=(N, n)).astype(dtype)
    f = rng.uniform(-1, 1, size=(N, nL, 6)).astype(dtype)
    return tau, f


def with_params(model, **changes):
    """Shallow copy of a host model with some model-level fields replaced."""
    with model.editable(validate=False) as m:
        for k, v in changes.items():
            setattr(m, k, v)
    return m


def enable_points(model, idx):
    kdp = model.kin_dyn_parameters
    en = np.zeros(kdp.number_of_collidable_points(), dtype=bool)
    en[list(idx)] = True
    return with_params(model, kin_dyn_parameters=dataclasses.replace(kdp, contact_enabled=en))


def upcast(d: oracle.OracleData, model=None) -> oracle.OracleData:
    """fp64 copy of an oracle state (same STATE values): the truth for fp32 parity checks.

    [round 4] The cached link kinematics are RECOMPUTED in fp64 from the state.  They used to be the fp32 caches cast to
    fp64, and the contact kinematics read the caches: the "truth" then carried the rounding of an
    fp32 forward-kinematics pass (6e-8 m in the height of a foot = 1e-4 in the velocity a 1e6 N/m^1.5 contact gives it in
    one step), the oracle run in fp32 shared exactly that rounding and looked ten times better than it is, and the
    kernel -- which derives the kinematics from the state, as a run of the reference in fp64 would -- looked ten times
    worse (tools/fp32_error.py, profiles/r04_fp32_error.txt)."""
    kw = {}
    for fld in dataclasses.fields(d):
        v = getattr(d, fld.name)
        kw[fld.name] = v.astype(np.float64) if isinstance(v, np.ndarray) else v
    out = oracle.OracleData(**kw)
    model = model if model is not None else getattr(d, "_model", None)
    if d.link_transforms is not None:
        if model is None:
            raise ValueError("helpers.upcast: pass the model (the caches of the copy are recomputed in fp64)")
        out = out.update_caches(model)
    return out


def oracle_sensitivity(model, d32: oracle.OracleData, trials: int = 3, seed: int = 0, **step_kw) -> np.ndarray:
    """Per environment: how far ONE fp64 oracle step moves (the parity metric, max over the state rows) when every
    entry of the fp32 input state is perturbed by one ulp (random signs, `trials` draws).  The contact models are
    discontinuous (a point entering contact, stick / slip, max(0, .)): an environment that sits on such an edge answers
    a 6e-8 m change of a foot height with 6e-4 in a joint velocity (tools/fp32_error.py, profiles/r04_fp32_error.txt),
    and no fp32 evaluation -- the reference's own formulation included -- can be expected closer to the fp64 result
    than that.  Full-size distribution tests bound the error of such environments by a multiple of this instead of
    widening the gate for all."""
    blk32 = odata_to_block(model, d32)
    blk = blk32.astype(np.float64)
    base = odata_to_block(model, oracle.step(model, block_to_odata(model, blk, d32.velocity_representation), **step_kw))
    rng = np.random.default_rng(seed)
    ulp = np.spacing(np.abs(blk32)).astype(np.float64)
    sens = np.zeros(blk.shape[1])
    for _ in range(trials):
        b2 = blk + rng.choice([-1.0, 1.0], size=blk.shape) * ulp
        o = odata_to_block(model, oracle.step(model, block_to_odata(model, b2, d32.velocity_representation), **step_kw))
        sens = np.maximum(sens, (np.abs(o - base) / np.maximum(1.0, np.abs(base))).max(axis=0))
    return sens


def rigid_model(model, idx, *, build=None, **params):
    """Host model with the RigidContacts model on the enabled points ``idx`` (reference idiom:
    ``tests/test_simulations.py:245-270``)."""
    cm = ja.RigidContacts.build(**(build or {}))
    return enable_points(with_params(model, contact_model=cm, contact_params=ja.RigidContactsParams(**params)), idx)


def relaxed_model(model, idx, *, build=None, **params):
    """Host model with the RelaxedRigidContacts model on the enabled points ``idx`` (reference idiom:
    ``tests/test_simulations.py:295-318``)."""
    cm = ja.RelaxedRigidContacts.build(**(build or {}))
    return enable_points(with_params(model, contact_model=cm, contact_params=ja.RelaxedRigidContactsParams.build(**params)), idx)


#: bottom corners of the four foot boxes of the synthetic quadruped (16 points), one corner per foot (4)
ANYMAL_FEET_16 = [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27]
ANYMAL_FEET_4 = [0, 8, 16, 24]


# ---- row B: joint-limit spring / damper and the torque-speed curve ---------------------------------
def actuation_variant(model, seed):
    """Tight position limits with spring AND damper, a low torque limit and a narrow torque-speed curve,
    so that random states sit beyond the limits and in all three regions of the curve
    (api/actuation_model.py:55-66 and :95-126; the damper term is the reference's `jnp.positive` quirk)."""
    rng = np.random.default_rng(seed)
    kdp = model.kin_dyn_parameters
    n = kdp.number_of_joints()
    kdp2 = dataclasses.replace(
        kdp, position_limits_min=np.full(n, -0.3), position_limits_max=np.full(n, 0.3),
        position_limit_spring=rng.uniform(50.0, 150.0, n), position_limit_damper=rng.uniform(0.05, 0.5, n),
        friction_static=rng.uniform(0.0, 0.3, n), friction_viscous=rng.uniform(0.0, 0.5, n),
    )  # fmt: skip
    return with_params(model, kin_dyn_parameters=kdp2,
                               actuation_params=ja.ActuationParams(torque_max=8.0, omega_th=0.3, omega_max=0.8))  # fmt: skip


def actuation_state(models, name, model, N, seed, dtype):
    d = models.random_data(name, N, seed=seed, dtype=dtype)
    rng = np.random.default_rng(seed + 77)
    n = model.dofs()
    d.joint_positions[:] = rng.uniform(-0.8, 0.8, (N, n)).astype(dtype)
    d.joint_velocities[:] = rng.uniform(-1.2, 1.2, (N, n)).astype(dtype)
    d = d.update_caches(model)
    s, w = np.asarray(d.joint_positions, float), np.abs(np.asarray(d.joint_velocities, float))
    # the states really exercise every branch: below / above the limits, and the three speed regions
    assert (s < -0.3).any() and (s > 0.3).any() and (np.abs(s) <= 0.3).any()
    assert (w <= 0.3).any() and ((w > 0.3) & (w <= 0.8)).any() and (w > 0.8).any()
    return d




# ---- fixed-base models with the rigid contact models [round 3] ---------------------------------------
def fixed_cart_model(kind="rigid", **params):
    """The cartpole with collision shapes (fixed base: a rail, a cart on a prismatic joint, a pole) with the four
    BOTTOM corners of the cart as rigid / relaxed contact points: a fixed-base mechanism touching the ground."""
    m = ja.JaxSimModel.build_from_model_description(robots.cartpole_urdf(with_collisions=True))
    kdp = m.kin_dyn_parameters
    idx = [int(i) for i in np.flatnonzero((np.asarray(kdp.contact_body) == 1) & (np.asarray(kdp.contact_point)[:, 2] < 0))][:4]
    assert len(idx) == 4 and not m.floating_base()
    return (rigid_model if kind == "rigid" else relaxed_model)(m, idx, **params)


def fixed_cart_data(model, N, seed=0, dtype=np.float64, base_velocity=0.0):
    """States of ``fixed_cart_model`` whose cart corners straddle the ground (the rail is 1 m above the base origin,
    so the base sits about 1 m below the terrain), with an optional stored base velocity -- which the reference's
    impact writes into fixed-base states anyway (rbda/contacts/rigid.py:391-446)."""
    d = oracle.random_model_data(model, batch_size=N, seed=seed, dtype=dtype,
                                 base_pos_bounds=((-1, -1, -1.0), (1, 1, -1.0)), base_rpy_bounds=((-0.15, -0.15, -3), (0.15, 0.15, 3)))  # fmt: skip
    # lower every environment until its lowest enabled point is 0 .. 8 mm under the ground (half of them stay above)
    rng0 = np.random.default_rng(seed + 500)
    pz, _ = oracle.collidable_points_pos_vel(model, link_transforms=d.link_transforms, link_velocities=d.link_velocities)
    en = np.flatnonzero(model.kin_dyn_parameters.contact_enabled)
    shift = -pz[:, en, 2].min(axis=1) - rng0.uniform(0.0, 0.008, N)
    pos = np.array(d.base_position, dtype=np.float64)
    pos[:, 2] += shift
    d = dataclasses.replace(d, base_position=pos.astype(dtype)).update_caches(model)
    if base_velocity:
        rng = np.random.default_rng(seed + 1000)
        d = dataclasses.replace(d, base_linear_velocity=(base_velocity * rng.uniform(-1, 1, (N, 3))).astype(dtype),
                                base_angular_velocity=(base_velocity * rng.uniform(-1, 1, (N, 3))).astype(dtype)).update_caches(model)
    return d
