"""Developer tool: time BASELINE.json config 5 on one GPU -- synthetic ANYmal-like quadruped
(13 links, 12 DoF), RigidContacts, tau = RNEA gravity term recomputed every step on the device,
fp32, batch 4096.  Not the headline benchmark (bench.py measures config 3); the numbers feed
HISTORY.md section 6.

    python tools/bench_c5.py [--points 4|16] [--envs 4096] [--steps 200] [--contact rigid|relaxed]

`--contact relaxed` swaps in RelaxedRigidContacts with `estimate_good_contact_parameters` (the contact
model of the reference's own `test_simulation_step` benchmark, tests/test_benchmark.py:142-152).
"""

import argparse
import ctypes as C
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4, choices=[4, 16, 32, 200], help="200: the quadruped with a 50-point sphere on every foot (RelaxedRigidContacts only: more points than lanes)")
    ap.add_argument("--standing", action="store_true", help="standing states: every sole point in contact")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--contact", default="rigid", choices=["rigid", "relaxed"])
    ap.add_argument("--default-params", action="store_true", help="relaxed: the reference's default RelaxedRigidContactsParams (mu = 0.005) instead of estimate_good_contact_parameters")
    args = ap.parse_args()

    import helpers  # tests/helpers.py: model zoo + rigid_model
    import jaxsim_amd.api as js
    from jaxsim_amd import _lib, runtime

    runtime.require_device()
    zoo = helpers.ModelZoo()
    # --points 32: the 24-link humanoid of config 3 with all its 32 collidable points enabled
    robot = "icub" if args.points == 32 else "anymal"
    idx = helpers.ANYMAL_FEET_4 if args.points == 4 else helpers.ANYMAL_FEET_16 if args.points == 16 else list(range(args.points))
    base = zoo(robot)
    if args.points == 200:
        import jaxsim_amd as ja
        from jaxsim_amd import robots

        base = ja.JaxSimModel.build_from_model_description(robots.anymal12_urdf(foot_shape="sphere"))
    if args.contact == "rigid":
        model = helpers.rigid_model(base, idx, K=1e4, D=2e2)
    else:
        model = helpers.relaxed_model(base, idx)
        if not args.default_params:
            model = helpers.with_params(model, contact_params=js.contact.estimate_good_contact_parameters(model))
    dtype = np.dtype(args.dtype)
    if args.standing:
        d = helpers.standing_data(model, args.envs, seed=0, dtype=dtype, noise=0.003)
    else:
        d = zoo.random_data(robot, args.envs, seed=0, dtype=dtype)
        if args.points == 200:  # (the states of the zoo's quadruped, whose caches belong to its point set: rebuilt for this one)
            import oracle

            d = oracle.random_model_data(model, batch_size=args.envs, seed=0, dtype=dtype, base_pos_bounds=((-1, -1, 0.58), (1, 1, 0.70)), base_rpy_bounds=((-0.3, -0.3, -3), (0.3, 0.3, 3)))
    data = js.data.JaxSimModelData.from_state_block(model, helpers.odata_to_block(model, d), 2)
    dm = runtime.device_model(model, dtype)
    from jaxsim_amd import specialize

    specialize.ensure_mode(dm, model, specialize.MODE_GRAV)
    lib = _lib.load()
    stream = runtime.Stream()
    runtime.set_stream(stream)
    tau = runtime.DeviceArray(model.dofs(), args.envs, dtype, tile=data._state.tile, zero=True)
    st, tp = C.c_void_p(data._state.ptr), C.c_void_p(tau.ptr)

    def run(k):
        for _ in range(k):
            _lib.check(lib.jxs_gravity_torques(dm.handle, st, tp, args.envs, stream.handle), "jxs_gravity_torques")
            _lib.check(lib.jxs_step(dm.handle, st, st, tp, None, 2, args.envs, stream.handle), "jxs_step")

    run(args.warmup)
    stream.synchronize()
    e0, e1 = runtime.Event(), runtime.Event()
    e0.record(stream)
    run(args.steps)
    e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_ms(e1) / args.steps
    blk = data.state_block()
    cname = "RigidContacts" if args.contact == "rigid" else "RelaxedRigidContacts"
    print(json.dumps({
        "workload": f"config 5: {robot} synthetic{' (standing)' if args.standing else ''}, {cname} ({args.points} points), gravity compensation, {args.dtype}",
        "envs": args.envs, "steps": args.steps, "ms_per_step": ms, "env_steps_per_s": args.envs / (ms * 1e-3),
        "finite_envs": float(np.isfinite(blk).all(axis=0).mean()), "lanes_per_env": int(dm.layout.group),
    }))  # fmt: skip


if __name__ == "__main__":
    main()
