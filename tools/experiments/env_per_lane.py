#!/usr/bin/env python3
"""EXPERIMENT driver: the soft-contact step with one environment per lane
(tools/experiments/env_per_lane.hip) against the product kernel (link per lane, G lanes per environment).
Checks that both produce the same step, then times them over a range of batch sizes.

    python tools/experiments/env_per_lane.py [--model icub23|anymal12] [--sizes 1024,16384,65536,262144]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from jaxsim_amd import _lib, runtime  # noqa: E402

KMAXL, KMAXP = 32, 64


class ELink(C.Structure):
    _fields_ = [("parent", C.c_int), ("jtype", C.c_int), ("axis", C.c_float * 3), ("Rpre", C.c_float * 9), ("ppre", C.c_float * 3),
                ("mass", C.c_float), ("com", C.c_float * 3), ("I", C.c_float * 6), ("kc", C.c_float), ("kv", C.c_float),
                ("smin", C.c_float), ("smax", C.c_float), ("klim", C.c_float), ("dlim", C.c_float), ("p0", C.c_int), ("p1", C.c_int)]  # fmt: skip


class EModel(C.Structure):
    _fields_ = [("nL", C.c_int), ("n", C.c_int), ("n_points", C.c_int), ("n_rows", C.c_int), ("floating", C.c_int)] + [
        (k, C.c_float) for k in ("dt", "g", "K", "D", "mu", "eps", "K_over_D", "quat_K", "tau_max", "w_th", "w_max", "inv_w_range", "terrain_h")
    ] + [("link", ELink * KMAXL), ("ppos", (C.c_float * 3) * KMAXP), ("prow", C.c_int * KMAXP)]  # fmt: skip


def build_emodel(model) -> EModel:
    kdp = model.kin_dyn_parameters
    nL, n = kdp.number_of_links(), kdp.number_of_joints()
    m = EModel()
    m.nL, m.n, m.n_points, m.floating = nL, n, kdp.number_of_collidable_points(), int(model.floating_base())
    m.n_rows = 13 + 2 * n + 3 * m.n_points
    cp, ap = model.contact_params, model.actuation_params
    m.dt, m.g, m.K, m.D, m.mu = model.time_step, model.gravity, cp.K, cp.D, cp.mu
    m.eps, m.K_over_D, m.quat_K = float(np.finfo(np.float32).eps), cp.K / cp.D, 0.1
    m.tau_max, m.w_th, m.w_max, m.inv_w_range = ap.torque_max, ap.omega_th, ap.omega_max, 1.0 / (ap.omega_max - ap.omega_th)
    m.terrain_h = float(model.terrain._height)
    fmax = float(np.finfo(np.float32).max)
    en = [k for k in range(m.n_points) if kdp.contact_enabled[k]]
    order = sorted(en, key=lambda k: int(kdp.contact_body[k]))
    assert len(order) <= KMAXP and nL <= KMAXL
    for slot, k in enumerate(order):
        m.ppos[slot] = (C.c_float * 3)(*[float(x) for x in kdp.contact_point[k]])
        m.prow[slot] = k
    bodies = [int(kdp.contact_body[k]) for k in order]
    for i in range(nL):
        L = m.link[i]
        L.parent, L.jtype = int(kdp.parent_array[i]), 0 if i == 0 else int(kdp.joint_types[i - 1])
        pre, suc = np.asarray(kdp.lambda_H_pre[i]), np.asarray(kdp.suc_H_i[i])
        assert np.allclose(suc, np.eye(4)), "the experiment kernel assumes suc_H_i = I (URDF models)"
        L.Rpre = (C.c_float * 9)(*pre[:3, :3].reshape(-1))
        L.ppre = (C.c_float * 3)(*pre[:3, 3])
        if i > 0:
            L.axis = (C.c_float * 3)(*kdp.joint_axis[i - 1])
            j = i - 1
            L.kc, L.kv = float(kdp.friction_static[j]), float(kdp.friction_viscous[j])
            L.smin, L.smax = max(-fmax, float(kdp.position_limits_min[j])), min(fmax, float(kdp.position_limits_max[j]))
            L.klim, L.dlim = float(kdp.position_limit_spring[j]), float(kdp.position_limit_damper[j])
        L.mass = float(kdp.link_mass[i])
        L.com = (C.c_float * 3)(*kdp.link_com[i])
        I = np.asarray(kdp.link_inertia_com[i]).reshape(3, 3)
        L.I = (C.c_float * 6)(I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2])
        idx = [s for s, b in enumerate(bodies) if b == i]
        L.p0, L.p1 = (idx[0], idx[-1] + 1) if idx else (0, 0)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="icub23")
    ap.add_argument("--sizes", default="1024,4096,16384,65536,262144")
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    so = os.path.join(HERE, "libenvlane.so")
    if not os.path.exists(so):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                        os.path.join(HERE, "env_per_lane.hip"), "-o", so], check=True)  # fmt: skip
    model = bench.build_model(args.model)
    lib = _lib.load()
    epl = C.CDLL(so)
    em = build_emodel(model)
    assert epl.epl_model_bytes() == C.sizeof(EModel), (epl.epl_model_bytes(), C.sizeof(EModel))
    dmodel = C.c_void_p()
    assert epl.epl_upload_model(C.byref(em), C.byref(dmodel)) == 0
    stream = runtime.Stream()
    runtime.set_stream(stream)
    dm = runtime.device_model(model, np.float32)
    for N in [int(x) for x in args.sizes.split(",")]:
        data = bench.synthetic_state(model, min(N, 4096), seed=0, dtype=np.float32)
        blk = np.tile(data.state_block(), (1, -(-N // min(N, 4096))))[:, :N].astype(np.float32)
        # ---- product kernel
        import jaxsim_amd.api as js

        g = js.data.JaxSimModelData.from_state_block(model, blk)
        ptr = C.c_void_p(g._state.ptr)
        _lib.check(lib.jxs_step(dm.handle, ptr, ptr, None, None, 2, N, stream.handle), "jxs_step")
        stream.synchronize()
        ref = g.state_block()
        # ---- experiment kernel: plain [rows][N]
        plain = runtime.DeviceArray.from_host(blk, tile=N)
        us = C.c_float(0)
        assert epl.epl_run(dmodel, C.c_void_p(plain.ptr), N, 1, C.byref(us)) == 0
        out = plain.to_host()
        err = float(np.max(np.abs(out - ref) / np.maximum(1.0, np.abs(ref))))
        # ---- timing
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, args.steps, stream.handle), "jxs_step_repeat")
        stream.synchronize()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record(stream)
        _lib.check(lib.jxs_step_repeat(dm.handle, ptr, None, None, 2, N, args.steps, stream.handle), "jxs_step_repeat")
        e1.record(stream)
        stream.synchronize()
        us_prod = e0.elapsed_ms(e1) / args.steps * 1e3
        assert epl.epl_run(dmodel, C.c_void_p(plain.ptr), N, 20, C.byref(us)) == 0
        assert epl.epl_run(dmodel, C.c_void_p(plain.ptr), N, args.steps, C.byref(us)) == 0
        print(f"{args.model} N={N:7d}: one-step difference {err:.2e} | link-per-lane {us_prod:9.2f} us ({N / us_prod:8.1f} M env-steps/s) | "
              f"env-per-lane {us.value:9.2f} us ({N / us.value:8.1f} M env-steps/s) | ratio {us_prod / us.value:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
