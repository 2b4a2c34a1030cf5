#!/usr/bin/env python3
"""[round 4] The query kernels of SURVEY section 8 rows F / R / N / f4 timed the way the reference times them
(`tests/test_benchmark.py:39-152`: one function per benchmark, batch of states, `jax.block_until_ready`): forward
dynamics (ABA), bias forces (RNEA at zero acceleration), inverse dynamics, gravity torques, forward kinematics,
mass matrix (CRBA), mass-matrix inverse, the full Jacobian + its derivative -- and the step kernels beside them.
State resident in HBM, HIP events on the launch stream around `--reps` launches, the 24-link humanoid of the
headline (`bench.build_model`), both precisions.  Per launch: time, batch / time, and the HBM bytes the launch has
to move (state in + result out, algorithmic) against 8 TB/s.

    python tools/bench_queries.py [--envs 1024] [--reps 300] > profiles/r04_query_kernels.txt
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--model", default="icub23")
ap.add_argument("--specialised", action="store_true", help="model-specialised builds of the query kernels too (js.model.specialize(model, queries=True): hipcc, seconds)")
args = ap.parse_args()

runtime.require_device()
lib = _lib.load()
stream = runtime.Stream()
runtime.set_stream(stream)
model = bench.build_model(args.model)
N = args.envs
what = "model-specialised builds" if args.specialised else "the library kernels built ahead of time; the step is the specialised kernel of the default policy"
print(f"# query kernels ({what}), {args.model} synthetic humanoid, N = {N}, {args.reps} launches per figure, HIP events on the launch stream")
print(f"# {'kernel':44s} {'dtype':5s} {'us/launch':>10s} {'M env/s':>9s} {'alg. KB/env':>11s} {'GB/s':>8s} {'% HBM':>6s}")
for dtype in (np.float32, np.float64):
    data = bench.synthetic_state(model, N, seed=0, dtype=dtype)
    if args.specialised:
        import jaxsim_amd.api as js

        js.model.specialize(model, dtype, queries=True)
    dm = runtime.device_model(model, dtype)
    lay = dm.layout
    tile = lay.tile
    Np = (N + tile - 1) // tile * tile
    nL, n = lay.n_links, lay.n_joints
    nv = 6 + n
    sz = np.dtype(dtype).itemsize
    rows_state = data.state_block().shape[0]

    def buf(rows):
        p = C.c_void_p()
        _lib.check(lib.jxs_malloc(C.byref(p), rows * Np * sz), "jxs_malloc")
        _lib.check(lib.jxs_memset(p, 0, rows * Np * sz, stream.handle), "jxs_memset")
        return p

    sp = C.c_void_p(data._state.ptr)
    acc, frc, tau, M, Mi, J, HT, HV = buf(nv), buf(nv), buf(n), buf(nv * nv), buf(nv * nv), buf(2 * 6 * nv), buf(nL * 12), buf(nL * 6)
    scratch = buf(rows_state)
    cases = [
        ("forward_dynamics_aba (MODE_FD)", lambda: lib.jxs_forward_dynamics_aba(dm.handle, sp, None, None, 2, acc, N, stream.handle), rows_state + nv),
        ("free_floating_bias_forces (MODE_ID, zero acc.)", lambda: lib.jxs_inverse_dynamics(dm.handle, sp, None, None, 2, frc, N, stream.handle), rows_state + nv),
        ("inverse_dynamics (MODE_ID)", lambda: lib.jxs_inverse_dynamics(dm.handle, sp, acc, None, 2, frc, N, stream.handle), rows_state + 2 * nv),
        ("gravity torques g(q) (MODE_GRAV)", lambda: lib.jxs_gravity_torques(dm.handle, sp, tau, N, stream.handle), rows_state + n),
        ("forward kinematics + link velocities (MODE_KIN)", lambda: lib.jxs_refresh_kinematics(dm.handle, sp, HT, HV, N, stream.handle), rows_state + nL * 18),
        ("free_floating_mass_matrix (MODE_CRBA)", lambda: lib.jxs_mass_matrix(dm.handle, sp, M, N, stream.handle), rows_state + nv * nv),
        ("free_floating_mass_matrix_inverse (MODE_MINV)", lambda: lib.jxs_mass_matrix_inverse(dm.handle, sp, Mi, N, stream.handle), rows_state + nv * nv),
        ("jacobian_full + derivative (MODE_JAC)", lambda: lib.jxs_jacobian_full(dm.handle, sp, J, HT, N, stream.handle), rows_state + 2 * 6 * nv + nL * 12),
        ("step, out of place (MODE_STEP)", lambda: lib.jxs_step(dm.handle, sp, scratch, None, None, 2, N, stream.handle), 2 * rows_state),
    ]
    for name, call, rows_moved in cases:
        for _ in range(20):
            _lib.check(call(), name)
        stream.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record(stream)
            for _ in range(args.reps):
                _lib.check(call(), name)
            e1.record(stream)
            stream.synchronize()
            best.append(e0.elapsed_ms(e1) / args.reps * 1e3)
        us = float(np.median(best))
        kb = rows_moved * sz / 1e3
        gbs = rows_moved * sz * N / (us * 1e-6) / 1e9
        print(f"  {name:44s} {np.dtype(dtype).name[-2:]:>5s} {us:10.2f} {N / us:9.1f} {kb:11.2f} {gbs:8.1f} {100 * gbs / 8000:6.2f}")

# [round 6] the reference's three contact-model benchmarks (tests/test_benchmark.py:103-139): js.ode.system_dynamics on the humanoid
# with SoftContacts / RigidContacts / RelaxedRigidContacts and estimate_good_contact_parameters -- ONE launch each here
# (jxs_system_dynamics: contact forces, per-link sums, ABA, position derivatives), and js.contact.link_contact_forces alone
import jaxsim_amd as ja  # noqa: E402
import jaxsim_amd.api as js  # noqa: E402

print(f"# system_dynamics / link_contact_forces (tests/test_benchmark.py:103-139), {args.model} synthetic humanoid with all its sole points, estimate_good_contact_parameters, N = {N}")
for cname, cm in (("SoftContacts", ja.SoftContacts()), ("RigidContacts", ja.RigidContacts.build()), ("RelaxedRigidContacts", ja.RelaxedRigidContacts.build())):
    for dtype in (np.float32, np.float64):
        m = bench.build_model(args.model)
        m.contact_model = cm
        m.contact_params = js.contact.estimate_good_contact_parameters(m)
        data = bench.synthetic_state(m, N, seed=0, dtype=dtype)
        try:
            dm = runtime.device_model(m, dtype)
        except Exception as exc:  # (e.g. RigidContacts with 32 points in fp64 beyond the LDS budget: reported, not hidden)
            print(f"  {'system_dynamics, ' + cname:44s} {np.dtype(dtype).name[-2:]:>5s}  refused: {str(exc)[:90]}")
            continue
        from jaxsim_amd import specialize

        spec = specialize.ensure_mode(dm, m, specialize.dyn_mode_of(m))  # what js.ode.system_dynamics does on its first call
        lay = dm.layout
        Np = (N + lay.tile - 1) // lay.tile * lay.tile
        sz = np.dtype(dtype).itemsize
        rows_state = data.state_block().shape[0]
        xdot, W = C.c_void_p(), C.c_void_p()
        _lib.check(lib.jxs_malloc(C.byref(xdot), rows_state * Np * sz), "jxs_malloc")
        _lib.check(lib.jxs_malloc(C.byref(W), lay.n_links * 6 * Np * sz), "jxs_malloc")
        sp = C.c_void_p(data._state.ptr)
        for name, call, rows_moved in (
            ("system_dynamics, " + cname + ("" if spec else " (library kernel)"), lambda: lib.jxs_system_dynamics(dm.handle, sp, None, None, 2, C.c_double(1.0), xdot, None, N, stream.handle), 2 * rows_state),
            ("link_contact_forces, " + cname, lambda: lib.jxs_link_contact_forces(dm.handle, sp, None, None, 2, W, None, N, stream.handle), rows_state + 6 * lay.n_links),
        ):
            for _ in range(20):
                _lib.check(call(), name)
            stream.synchronize()
            best = []
            for _ in range(5):
                e0, e1 = runtime.Event(), runtime.Event()
                e0.record(stream)
                for _ in range(args.reps):
                    _lib.check(call(), name)
                e1.record(stream)
                stream.synchronize()
                best.append(e0.elapsed_ms(e1) / args.reps * 1e3)
            us = float(np.median(best))
            gbs = rows_moved * sz * N / (us * 1e-6) / 1e9
            print(f"  {name:44s} {np.dtype(dtype).name[-2:]:>5s} {us:10.2f} {N / us:9.1f} {rows_moved * sz / 1e3:11.2f} {gbs:8.1f} {100 * gbs / 8000:6.2f}")
        lib.jxs_free(xdot), lib.jxs_free(W)
