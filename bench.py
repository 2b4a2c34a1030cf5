#!/usr/bin/env python3
"""Headline benchmark: simulated env-steps/s of ``js.model.step`` for the synthetic iCub
23-DoF floating-base humanoid with soft ground contacts (BASELINE.json configs[2] on one GPU,
configs[3] = the same workload sharded over the GPUs of a node).

    python bench.py --gpus 1 --steps 2000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 2000 --warmup 50

A "step" is one ``jxs_step`` launch over the rank's batch (1024 environments per GPU, weak
scaling: 8192 environments on 8 GPUs).  The state is resident in HBM, there is no host
round-trip inside the timed region and no per-step communication; after the timed region the
final state shards are concatenated with ONE RCCL all-gather (timed separately).  Rank 0
prints one JSON line.  ``torch.distributed.run`` is only the launcher: the ranks rendezvous through a
temp file and use RCCL through the C-ABI library for barriers, the max-over-ranks and the gather.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3  # vector FP32 peak (same guide)
FLOPS_PER_ENV_STEP = 30e3  # structure-exploiting flop model, SURVEY.md section 8(d) / A.5
# HBM-side bytes per launch come from the rocprofv3 PMC passes committed under profiles/ (separate --pmc
# FETCH_SIZE / WRITE_SIZE runs of this very command; FETCH_SIZE doubled as the microarch guide prescribes
# for gfx950).  Counters cannot be read from inside this process, so the figure is tied to the kernel
# sources it was measured on: the profile records a hash of jaxsim_amd/csrc/*, and a run of a different
# kernel reports `traffic: null` instead of a stale number.
def kernel_source_sha():
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "jaxsim_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".inc", ".hip", ".sh")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _pmc_tags():
    """The rounds that committed a PMC record (profiles/rNN_pmc.json), newest first."""
    import glob
    import re

    tags = [m.group(1) for m in (re.match(r"(r\d+)_pmc\.json$", os.path.basename(p)) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))) if m]
    return sorted(tags, reverse=True)


def profiled_traffic(model_name, n_envs, dtype_name):
    """(bytes per launch | None, note) for the configuration that was profiled."""
    for tag in _pmc_tags():
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")) as f:
                prof = json.load(f)
        except (OSError, ValueError):
            continue
        cfg = prof.get("config", {"model": "icub23", "envs": 1024, "dtype": "float32"})
        if (cfg.get("model"), cfg.get("envs"), cfg.get("dtype")) != (model_name, n_envs, dtype_name):
            return None, f"profiles/{tag}_pmc.json holds another configuration"
        sha = prof.get("kernel_source_sha")
        if sha is not None and sha != kernel_source_sha():
            return None, f"profiles/{tag}_pmc.json was measured on other kernel sources ({sha}); re-run tools/profile_round.sh"
        return prof.get("traffic_bytes_per_launch"), f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, mean per launch)"
    return None, "no PMC profile committed"


def profiled_config5_traffic():
    """HBM-side bytes per launch of the config-5 step kernel (tools/profile_round.sh c5_pmc_* passes), or None when the
    committed profile was taken on other kernel sources."""
    for tag in _pmc_tags():
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")) as f:
                prof = json.load(f)
        except (OSError, ValueError):
            continue
        if prof.get("kernel_source_sha") != kernel_source_sha():
            return None, f"profiles/{tag}_pmc.json was measured on other kernel sources; re-run tools/profile_round.sh"
        return prof.get("config5_traffic_bytes_per_launch"), f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of jxs_kernel<float,16,MODE_STEP_RIGID>, mean per launch)"
    return None, "no PMC profile committed"


SHADER_CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md); 157.3 TFLOP/s fp32 = 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz
N_SIMDS = 1024            # 256 CUs x 4
VALU_ISSUE_CYCLES = 2.0   # a wave64 fp32 instruction occupies a SIMD for two cycles at that peak


def kernel_valu_count(dm, mode=0):
    """Static VALU instruction count of the step kernel a device model launches (the model-specialised object that is
    attached, disassembled with llvm-objdump; None when there is none or the tool is missing).  The kernel is almost
    straight-line code, so this is the instruction count of one wave to a few per cent."""
    try:
        from jaxsim_amd import isa_lint, specialize

        name = specialize.attached_files(dm).get(mode)
        if name is None:
            return None
        best = None
        for elf in isa_lint.code_objects(str(specialize.CACHE / name)):
            for sym, insts in isa_lint.parse(isa_lint.disassemble(elf)).items():
                if "jxs_kernel" in sym and "duo" not in sym:
                    n = sum(i.op.startswith("v_") for i in insts)
                    best = n if best is None else min(best, n)
        return best
    except Exception:
        return None


def issue_figures(valu_per_wave, envs_per_wave, n_envs, us_per_step):
    """Two numbers that say where the step kernel's time goes once the HBM roofline is out of reach:
    `valu_issue_util` = the cycles the SIMDs spend issuing this launch's vector instructions / the cycles they had;
    `lane_slot_efficiency` = modelled lane-operations (FLOPS_PER_ENV_STEP / 2 multiply-adds) / lane slots issued."""
    if not valu_per_wave:
        return {}
    waves = -(-n_envs // envs_per_wave)
    cycles_available = N_SIMDS * us_per_step * 1e-6 * SHADER_CLOCK_GHZ * 1e9
    return {"valu_per_wave_static": int(valu_per_wave), "waves": int(waves),
            "valu_issue_util": waves * valu_per_wave * VALU_ISSUE_CYCLES / cycles_available,
            "lane_slot_efficiency": (FLOPS_PER_ENV_STEP / 2) / (valu_per_wave * 64 / envs_per_wave)}


def rccl_required(args, device_count: int, world: int) -> bool:
    """[round 6] Whether a rank whose RCCL communicator cannot be created must exit non-zero.  Asked for explicitly
    (--require-rccl / JAXSIM_AMD_REQUIRE_RCCL=1), or BY DEFAULT whenever the box has a device per rank: there the file
    collective can only be an accident, and a scaling record must not be one.  The file collective remains for
    --share-device (fewer GPUs than ranks, where RCCL refuses two ranks on one device) and --allow-file-collective."""
    if args.require_rccl:
        return True
    if args.share_device or args.allow_file_collective:
        return False
    return device_count >= world


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=1024)
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many environments in total, split evenly over the GPUs (north_star: batch 8192 over 1 / 2 / 4 / 8 GPUs); overrides --envs-per-gpu and the line says \"scaling\": \"strong\"")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--model", default="icub23", choices=["icub23", "icub23_16", "anymal12", "cartpole"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="developer: run the multi-rank code path even with WORLD_SIZE=1")
    ap.add_argument("--require-rccl", action="store_true", default=os.environ.get("JAXSIM_AMD_REQUIRE_RCCL", "") not in ("", "0"),
                    help="multi-rank runs: a rank whose communicator is not RCCL (ncclCommInitRank failed and the host-side file collective "
                    "would take over) exits non-zero instead -- a scaling record cannot be a file collective by accident "
                    "(also JAXSIM_AMD_REQUIRE_RCCL=1)")
    ap.add_argument("--allow-file-collective", action="store_true", help="multi-rank runs on a box with a device per rank: let the host-side file collective "
                    "take over when RCCL fails instead of exiting non-zero (the default there since round 6 is --require-rccl)")
    ap.add_argument("--share-device", action="store_true", help="developer: ranks take device LOCAL_RANK %% device_count -- the N > 1 path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device: the host-side file collective takes over, `comm.kind` says so); the figure is not a scaling measurement")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--saturated-envs", type=int, default=65536, help="secondary figure: batch that saturates one GPU (0 = skip)")
    ap.add_argument("--no-other-contact-models", action="store_true", help="skip the secondary RigidContacts / RelaxedRigidContacts figures")
    ap.add_argument("--no-python-loop", action="store_true", help="skip the secondary `python_step_loop` figure (the rocprofv3 passes of tools/profile_round.sh: "
                    "its 12 000 plain launches would be averaged into the per-dispatch statistics of the step kernel)")
    ap.add_argument("--dry-run-bootstrap", action="store_true",
                    help="no GPU work: run only the multi-rank bootstrap of this script (job key, id exchange through the "
                    "rendezvous file, host collective, shard bounds) and print what a rank-0 line would say about it")
    args = ap.parse_args()
    args.scaling = "weak"
    if args.global_batch > 0:
        if args.global_batch % max(args.gpus, 1) != 0:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of --gpus {args.gpus}")
        args.envs_per_gpu = args.global_batch // max(args.gpus, 1)
        args.scaling = "strong"
    return args


def build_model(name):
    """Benchmark model.  Contact parameters come from the reference's own recipe,
    ``estimate_good_contact_parameters`` (``src/jaxsim/api/contact.py:160-211``) with 16 active
    points and damping ratio 0.2; joint-limit springs (k = 100 N m/rad, the reference reads them
    from ``JAXSIM_JOINT_POSITION_LIMIT_SPRING``) keep the unactuated joints inside +-1 rad.  With
    the reference's *default* K = 1e6 / D = 2000 the explicit contact forces on the light foot
    links diverge within ~50 steps at dt = 1e-3 -- in the fp64 oracle too -- so those defaults
    would benchmark NaNs (HISTORY.md section 7)."""
    import dataclasses

    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import robots

    urdf = {
        "icub23": lambda: robots.icub23_urdf(sole_boxes_per_foot=2),
        "icub23_16": lambda: robots.icub23_urdf(sole_boxes_per_foot=1),
        "anymal12": robots.anymal12_urdf,
        "cartpole": robots.cartpole_urdf,
    }[name]()
    model = ja.JaxSimModel.build_from_model_description(urdf)
    kdp = model.kin_dyn_parameters
    n = kdp.number_of_joints()
    model.kin_dyn_parameters = dataclasses.replace(kdp, position_limit_spring=np.full(n, 100.0))
    if kdp.number_of_collidable_points() > 0:
        model.contact_params = js.contact.estimate_good_contact_parameters(
            model, number_of_active_collidable_points_steady_state=16, damping_ratio=0.2
        )
    return model


def synthetic_state(model, n_envs, seed, dtype):
    """Synthetic inputs after SURVEY.md section 8(d) ('standing set'): base xy U(-1,1), |roll|,
    |pitch| <= 0.3, yaw U(-pi, pi), joints U(limits = +-1 rad), all velocities U(-1, 1), zero
    tangential deformation, tau_ref = 0, link_forces = None; the base height is then shifted so
    that the lowest collidable point of every environment starts 5 mm above the ground (link
    transforms from the GPU kinematics kernel)."""
    import jaxsim_amd.api as js

    floating = model.floating_base()
    data = js.data.random_model_data(
        model,
        batch_size=n_envs,
        seed=seed,
        dtype=dtype,
        base_pos_bounds=((-1, -1, 0.55), (1, 1, 0.75)) if floating else ((0, 0, 0), (0, 0, 0)),
        base_rpy_bounds=((-0.3, -0.3, -np.pi), (0.3, 0.3, np.pi)) if floating else ((0, 0, 0), (0, 0, 0)),
    )
    kdp = model.kin_dyn_parameters
    if floating and kdp.number_of_collidable_points() > 0:
        H = data._link_transforms[:, kdp.contact_body]  # [N, n_cp, 4, 4]
        pz = np.einsum("ncj,cj->nc", H[:, :, 2, :3], kdp.contact_point) + H[:, :, 2, 3]
        p = np.array(data.base_position, dtype=np.float64)
        p[:, 2] += 0.005 - pz.min(axis=1)
        data = data.replace(model, base_position=p)
    return data


def other_contact_models(dtype, stream, steps=200, warmup=20):
    """Secondary figures (never `value`): the two other contact models of the step path on one GPU.

    * BASELINE.json configs[4]: quadruped, RigidContacts (one point per foot), tau = RNEA gravity term
      recomputed on the device every step, batch 4096;
    * the reference's own `test_simulation_step` benchmark idiom (tests/test_benchmark.py:142-152):
      RelaxedRigidContacts + `estimate_good_contact_parameters` on the humanoid with all 32 points, batch 1024.
    """
    import ctypes as C
    import dataclasses

    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import _lib, robots, runtime

    lib = _lib.load()
    out = {}
    tname = "float" if np.dtype(dtype) == np.float32 else "double"

    def timed(model, n_envs, seed, with_tau, kernel):
        data = synthetic_state(model, n_envs, seed=seed, dtype=dtype)
        start = data._state.copy()  # (every measured loop below starts from the same state: the work depends on the contact state)
        dm = runtime.device_model(model, dtype)
        st = C.c_void_p(data._state.ptr)
        tau = runtime.DeviceArray(model.dofs(), n_envs, dtype, tile=data._state.tile, zero=True)
        tp = C.c_void_p(tau.ptr)
        if with_tau:
            from jaxsim_amd import specialize as _sp

            _sp.ensure_mode(dm, model, _sp.MODE_GRAV)  # what js.model.gravity_compensation_torques does on first use

        def run(k, gravity=with_tau):
            for _ in range(k):
                if gravity:
                    _lib.check(lib.jxs_gravity_torques(dm.handle, st, tp, n_envs, stream.handle), "jxs_gravity_torques")
                _lib.check(lib.jxs_step(dm.handle, st, st, tp if with_tau else None, None, 2, n_envs, stream.handle), "jxs_step")

        def events(k, **kw):
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record(stream)
            run(k, **kw)
            e1.record(stream)
            stream.synchronize()
            return e0.elapsed_ms(e1) / k * 1e3

        run(warmup)
        stream.synchronize()
        us = events(steps)
        us_step = us
        if with_tau:
            # the step kernel inside the same controller loop (its run time depends on the contact state, so a
            # loop of step launches alone is a different workload): loop period minus the period of a loop of
            # gravity-torque launches alone (a kernel whose run time does not depend on the state)
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record(stream)
            for _ in range(steps):
                _lib.check(lib.jxs_gravity_torques(dm.handle, st, tp, n_envs, stream.handle), "jxs_gravity_torques")
            e1.record(stream)
            stream.synchronize()
            us_step = us - e0.elapsed_ms(e1) / steps * 1e3
        fused_us = None
        if with_tau:
            # [round 4] the same controller loop as ONE launch per step (jxs_step_gravity_compensated: g(q) formed inside
            # the step kernel from the kinematics it has in registers anyway)
            def run_fused(k):
                for _ in range(k):
                    _lib.check(lib.jxs_step_gravity_compensated(dm.handle, st, st, None, None, 2, n_envs, stream.handle), "jxs_step_gravity_compensated")

            _lib.check(lib.jxs_memcpy_d2d(st, C.c_void_p(start.ptr), start.nbytes, stream.handle), "jxs_memcpy_d2d")
            run_fused(warmup)
            stream.synchronize()
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record(stream)
            run_fused(steps)
            e1.record(stream)
            stream.synchronize()
            fused_us = e0.elapsed_ms(e1) / steps * 1e3
        finite = float(np.isfinite(data.state_block()).all(axis=0).mean())
        lay = dm.layout
        # read state + read tau + write state; the rigid contact models carry no
        # tangential deformation (rbda/contacts/rigid.py:445-458), so the 3 n_cp term drops
        alg = (2 * (13 + 2 * lay.n_joints) + lay.n_joints) * np.dtype(dtype).itemsize
        gbs = alg * n_envs / (us_step * 1e-6) / 1e9
        from jaxsim_amd import specialize

        fused = {} if fused_us is None else {"one_launch_per_step": {
            "us_per_step": fused_us, "env_steps_per_s": n_envs / (fused_us * 1e-6),
            "note": "jxs_step_gravity_compensated: tau = g(q) formed inside the step kernel; same controller loop, one launch instead of two"}}
        return {"envs": n_envs, "steps": steps, "us_per_step": us, "env_steps_per_s": n_envs / (us * 1e-6), "finite_envs": finite, **fused,
                "model_specialised_kernel": bool(specialize.modes(dm)),
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "traffic": None, "algorithmic_bytes_per_env_step": alg, "kernel": kernel,
                             "kernel_avg_launch_us": us_step}}  # fmt: skip

    try:
        quad = build_quadruped_rigid()
        out["config5_rigid_contacts"] = timed(quad, 4096, 100, True, f"jxs_kernel<{tname},16,MODE_STEP_RIGID>") | {
            "workload": "anymal12 synthetic, RigidContacts (4 points), tau = RNEA gravity term every step (jxs_gravity_torques + jxs_step)"}  # fmt: skip
        if np.dtype(dtype) == np.float32:
            tr, note = profiled_config5_traffic()
            out["config5_rigid_contacts"]["roofline"]["traffic"] = tr
            out["config5_rigid_contacts"]["roofline"]["traffic_source"] = note
    except Exception as e:
        out["config5_rigid_contacts"] = {"error": repr(e)}
    try:
        hum = build_humanoid_relaxed()
        out["relaxed_rigid_contacts"] = timed(hum, 1024, 200, False, f"jxs_kernel<{tname},32,MODE_STEP_RIGID>") | {
            "workload": "icub23 synthetic, all 32 collidable points, RelaxedRigidContacts with estimate_good_contact_parameters (jxs_step)"}  # fmt: skip
    except Exception as e:
        out["relaxed_rigid_contacts"] = {"error": repr(e)}
    out["note"] = "secondary figures, not `value`; DESIGN.md sections 4d, 4e, 6"
    return out


def build_quadruped_rigid():
    """BASELINE.json configs[4]: quadruped with RigidContacts, one point per foot."""
    import dataclasses

    import jaxsim_amd as ja
    from jaxsim_amd import robots

    quad = ja.JaxSimModel.build_from_model_description(robots.anymal12_urdf())
    kdp = quad.kin_dyn_parameters
    en = np.zeros(kdp.number_of_collidable_points(), dtype=bool)
    en[[0, 8, 16, 24]] = True  # one bottom corner of every foot box
    quad.kin_dyn_parameters = dataclasses.replace(kdp, contact_enabled=en)
    quad.contact_model = ja.RigidContacts.build()
    quad.contact_params = ja.RigidContactsParams(K=1e4, D=2e2)
    return quad


def build_humanoid_relaxed():
    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import robots

    hum = ja.JaxSimModel.build_from_model_description(robots.icub23_urdf(sole_boxes_per_foot=2))
    hum.contact_model = ja.RelaxedRigidContacts.build()
    hum.contact_params = js.contact.estimate_good_contact_parameters(hum)
    return hum


def secondary_models():
    """(model, dtype) of the secondary figures: __graft_entry__.build() pre-builds their specialised kernels."""
    import jaxsim_amd as ja
    from jaxsim_amd import robots

    return [(build_quadruped_rigid(), np.float32), (build_humanoid_relaxed(), np.float32),
            (ja.JaxSimModel.build_from_model_description(robots.double_pendulum_urdf()), np.float64),
            (ja.JaxSimModel.build_from_model_description(robots.cartpole_urdf()), np.float32)]  # fmt: skip


def cpu_baseline(model, block, budget_s):
    """Oracle C port (reference-structured dense 6x6 ABA, OpenMP over envs) on the host cores,
    the same workload (the very state block the GPU starts from), bounded sample."""
    from oracle import cport

    cores = os.cpu_count() or 1
    n_envs, dtype = block.shape[1], block.dtype
    cport.build(force=True)  # -march=native: rebuild on the machine that runs it
    best = (0.0, 1)
    for nt in sorted({max(1, min(cores, x)) for x in (cores, cores // 2, 128, 64, 32, 16, 8, 1)}):
        cport.step(model, block, n_steps=4, n_threads=nt)  # warm up this team size
        t0 = time.perf_counter()
        cport.step(model, block, n_steps=24, n_threads=nt)
        rate = 24 * n_envs / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, nt)
    rate, nt = best
    # time-boxed measurement at the chosen team size: chunks grow until one takes >= 1 s, then
    # chunks are repeated until the budget is spent (the first, warm-up, chunk is not counted)
    chunk = 500
    while True:
        t0 = time.perf_counter()
        cport.step(model, block, n_steps=chunk, n_threads=nt)
        if time.perf_counter() - t0 >= 1.0 or chunk >= 1 << 20:
            break
        chunk *= 2
    n_steps, dt = 0, 0.0
    while dt < budget_s:
        t0 = time.perf_counter()
        cport.step(model, block, n_steps=chunk, n_threads=nt)
        dt += time.perf_counter() - t0
        n_steps += chunk
    return {
        "value": n_steps * n_envs / dt,
        "unit": "env-steps/s",
        "cores": nt,
        "kind": "port",
        "sample": f"{n_steps} steps x {n_envs} envs of the same workload and initial state, oracle C port (dense 6x6 "
        f"reference formulation, gcc -O3 -march=native, OpenMP, best of several team sizes = {nt} threads on a "
        f"{cores}-CPU host), {np.dtype(dtype).name}, {dt:.1f} s",
    }


def timed_repetitions(timed_region, run_steps, steps, reps, stream, barrier, lib):
    """`reps` timed regions of EXACTLY `steps` launches each.  Every region is bracketed by a barrier and a
    stream synchronisation on both sides and timed with the host wall clock; the synchronisation waits poll
    (a blocking wait adds its wake-up latency, comparable to the whole region when `steps` is small), and
    the bracket itself runs inside ONE C call (`jxs_step_repeat_timed`: sync, clock, launches, sync, clock) so
    that no interpreter overhead sits inside the region; [round 4] the closing synchronisation polls a word in pinned
    host memory that the stream writes behind the last launch (hipStreamWriteValue32) instead of calling
    hipStreamQuery in a loop: the host sees the end ~2 us sooner (12 instead of 14 us of bracket per region).  Alternate repetitions take HIP events on the launch
    stream instead (the two event records would otherwise sit inside the wall-clock region)."""
    from jaxsim_amd import _lib, runtime

    wall, evs = [], []
    for r in range(2 * reps):
        barrier()
        if r % 2 == 0:
            wall.append(timed_region(steps))
        else:
            ev0, ev1 = runtime.Event(), runtime.Event()
            _lib.check(lib.jxs_stream_wait_spin(stream.handle), "jxs_stream_wait_spin")
            ev0.record(stream)
            run_steps(steps)
            ev1.record(stream)
            _lib.check(lib.jxs_stream_wait_spin(stream.handle), "jxs_stream_wait_spin")
            evs.append(ev0.elapsed_ms(ev1) * 1e-3)
        barrier()
    return wall, evs


def other_configs(stream, steps=200):
    """Secondary figures for BASELINE.json configs[0] and configs[1] (parity-test configurations, measured once
    so that every config has a number): the 2-link pendulum at batch 1 in fp64 (a latency figure: one launch
    per step, and the fused rollout) and the cartpole at batch 1024 in fp32."""
    import ctypes as C

    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import _lib, robots, runtime

    lib = _lib.load()
    out = {}
    for key, urdf, n_envs, dtype in (("config1_pendulum_batch1_fp64", robots.double_pendulum_urdf(), 1, np.float64),
                                     ("config2_cartpole_batch1024_fp32", robots.cartpole_urdf(), 1024, np.float32)):  # fmt: skip
        try:
            model = ja.JaxSimModel.build_from_model_description(urdf)
            data = js.data.random_model_data(model, batch_size=n_envs, seed=0, dtype=dtype)
            dm = runtime.device_model(model, dtype)
            sp = C.c_void_p(data._state.ptr)
            run = lambda k: _lib.check(lib.jxs_step_repeat(dm.handle, sp, None, None, 2, n_envs, k, stream.handle), "jxs_step_repeat")  # noqa: E731
            run(steps)
            stream.synchronize()
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record(stream)
            run(steps)
            e1.record(stream)
            stream.synchronize()
            us = e0.elapsed_ms(e1) / steps * 1e3
            e2, e3 = runtime.Event(), runtime.Event()
            e2.record(stream)
            _lib.check(lib.jxs_rollout(dm.handle, sp, None, None, 2, n_envs, 1000, stream.handle), "jxs_rollout")
            e3.record(stream)
            stream.synchronize()
            lay = dm.layout
            alg = (2 * (13 + 2 * lay.n_joints + 3 * lay.n_points) + lay.n_joints) * np.dtype(dtype).itemsize
            out[key] = {"envs": n_envs, "dtype": np.dtype(dtype).name, "us_per_step": us, "env_steps_per_s": n_envs / (us * 1e-6),
                        "fused_rollout_us_per_step": e2.elapsed_ms(e3) / 1000 * 1e3, "algorithmic_bytes_per_env_step": alg,
                        "hbm_frac": alg * n_envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "finite": bool(np.isfinite(data.state_block()).all())}  # fmt: skip
        except Exception as e:
            out[key] = {"error": repr(e)}
    out["note"] = "secondary figures, not `value`: one launch per step through hipGraph replays; batch 1 is launch latency"
    return out


def default_contact_params_variant(model_name, n_local, dtype, stream, window=40, windows=50):
    """Secondary figure: the SECOND input set SURVEY.md section 8(d) names -- the reference's DEFAULT soft-contact
    parameters (K = 1e6, D = 2000, src/jaxsim/rbda/contacts/soft.py:28-46), no joint damping / limit springs beyond
    the URDF's.  With dt = 1e-3 these sit at the stability limit of the explicit integration (HISTORY.md section 7:
    they diverge in the fp64 oracle too within ~50-1000 steps), so the figure is taken over short windows: `windows`
    windows of `window` (<= 40) steps, each from a FRESH copy of the synthetic state, kernel time by HIP events over
    the launches of one window.  Same kernel, same instruction stream as the headline; reported so that both input
    sets the survey names have a number."""
    import ctypes as C

    import jaxsim_amd as ja
    from jaxsim_amd import _lib, robots, runtime

    lib = _lib.load()
    model = ja.JaxSimModel.build_from_model_description(robots.icub23_urdf(sole_boxes_per_foot=2) if model_name == "icub23" else None)
    assert float(model.contact_params.K) == 1e6 and float(model.contact_params.D) == 2000.0  # the reference defaults
    data0 = synthetic_state(model, n_local, seed=0, dtype=dtype)
    dm = runtime.device_model(model, dtype)
    block0 = data0._state.copy()
    work = data0._state
    sp = C.c_void_p(work.ptr)
    us, finite = [], []
    for w in range(windows + 2):
        _lib.check(lib.jxs_memcpy_d2d(sp, C.c_void_p(block0.ptr), block0.nbytes, stream.handle), "jxs_memcpy_d2d")
        stream.synchronize()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record(stream)
        _lib.check(lib.jxs_step_repeat(dm.handle, sp, None, None, 2, n_local, window, stream.handle), "jxs_step_repeat")
        e1.record(stream)
        stream.synchronize()
        if w >= 2:  # (the first windows capture the replay graph)
            us.append(e0.elapsed_ms(e1) / window * 1e3)
    fin = float(np.isfinite(data0.state_block()).all(axis=0).mean())
    u = float(np.median(us))
    lay = dm.layout
    alg = (2 * (13 + 2 * lay.n_joints + 3 * lay.n_points) + lay.n_joints) * np.dtype(dtype).itemsize
    return {"contact_params": {"K": 1e6, "D": 2000.0, "mu": float(model.contact_params.mu)}, "window_steps": window, "windows": windows,
            "us_per_step": u, "env_steps_per_s": n_local / (u * 1e-6), "finite_envs_after_a_window": fin,
            "roofline": {"bound": "hbm", "achieved": alg * n_local / (u * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg * n_local / (u * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_env_step": alg},
            "note": "SURVEY 8(d) variant `default_contact_params`: reference default K / D, windows of <= 40 steps from a fresh state "
                    "(longer runs leave the finite range, in the fp64 oracle too); HIP events per window, median; secondary figure, not `value`"}  # fmt: skip


def secondary_dtype(model_name, n_local, dtype, stream, steps=200):
    """Secondary figure: the same workload and launch pattern in the other precision (the reference's own
    default arithmetic is fp64, src/jaxsim/__init__.py:17-35; the headline follows SURVEY.md section 8 = fp32)."""
    import ctypes as C

    from jaxsim_amd import _lib, runtime

    lib = _lib.load()
    model = build_model(model_name)
    data = synthetic_state(model, n_local, seed=0, dtype=dtype)
    dm = runtime.device_model(model, dtype)
    sp = C.c_void_p(data._state.ptr)
    _lib.check(lib.jxs_step_repeat(dm.handle, sp, None, None, 2, n_local, 50, stream.handle), "jxs_step_repeat")
    _lib.check(lib.jxs_step_repeat(dm.handle, sp, None, None, 2, n_local, steps, stream.handle), "jxs_step_repeat")
    stream.synchronize()
    e0, e1 = runtime.Event(), runtime.Event()
    e0.record(stream)
    _lib.check(lib.jxs_step_repeat(dm.handle, sp, None, None, 2, n_local, steps, stream.handle), "jxs_step_repeat")
    e1.record(stream)
    stream.synchronize()
    us = e0.elapsed_ms(e1) / steps * 1e3
    lay = dm.layout
    alg = (2 * (13 + 2 * lay.n_joints + 3 * lay.n_points) + lay.n_joints) * np.dtype(dtype).itemsize
    gbs = alg * n_local / (us * 1e-6) / 1e9
    tname = "float" if np.dtype(dtype) == np.float32 else "double"
    return {"dtype": "f32" if np.dtype(dtype) == np.float32 else "f64", "envs": n_local, "steps": steps, "us_per_step": us,
            "env_steps_per_s": n_local / (us * 1e-6), "nonfinite_envs": int((~np.isfinite(data.state_block()).all(axis=0)).sum()),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_env_step": alg, "kernel": f"jxs_kernel<{tname},{lay.group},MODE_STEP>", "kernel_avg_launch_us": us},
            "note": "same workload, launch pattern and initial state distribution in the other precision; secondary figure, not `value`"}  # fmt: skip


def dry_run_bootstrap(args, rank, world, result_out):
    """The host side of a multi-rank launch without any device: what must work before the first RCCL call
    of a real 8-GPU run.  The 128-byte id is random bytes instead of ncclGetUniqueId; the collective is the
    file collective that bench.py falls back to when RCCL is unavailable."""
    from jaxsim_amd import distributed

    key = distributed.job_key() + "_dry"
    uid = distributed.file_rendezvous(rank, world, key, timeout_s=60.0, make_id=lambda: os.urandom(128))
    fc = distributed.FileCollective(rank, world, key, device_sync=False)
    fc.barrier()
    n_total = args.envs_per_gpu * world
    lo, hi = distributed.shard_bounds(n_total, rank, world)
    los = fc.all_gather_scalars(float(lo))
    his = fc.all_gather_scalars(float(hi))
    ids = fc.all_gather_scalars(float(int.from_bytes(uid[:6], "little")))
    t = [float(fc.all_gather_scalars(0.001 * (r + 1)).max()) for r in range(3)]  # the max-over-ranks reduction
    # the device each rank would select (main(): LOCAL_RANK, or LOCAL_RANK % device_count with --share-device)
    devs = [int(v) for v in fc.all_gather_scalars(float(int(os.environ.get("LOCAL_RANK", "0"))))]
    ok = (los[0] == 0 and his[-1] == n_total and all(his[r] == los[r + 1] for r in range(world - 1))
          and all(his[r] - los[r] == args.envs_per_gpu for r in range(world)) and len(set(ids.tolist())) == 1)  # fmt: skip
    fc.barrier()
    if rank == 0:
        print(json.dumps({"dry_run_bootstrap": True, "ok": bool(ok), "n_gpus": world, "global_batch": n_total,
                          "shards": [[int(a), int(b)] for a, b in zip(los, his)], "same_id_on_all_ranks": len(set(ids.tolist())) == 1,
                          "max_over_ranks": t, "job_key": key, "device_of_rank": devs, "distinct_devices": len(set(devs)) == world,
                          "require_rccl": bool(args.require_rccl)}), file=result_out, flush=True)  # fmt: skip
    if not ok:
        raise SystemExit(f"rank {rank}: bootstrap dry run failed: {los} {his} {ids}")


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start one process per GPU of this node with the variables
    torch.distributed.run would export (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the bootstrap of
    jaxsim_amd/distributed.py is file-based and torch-free, its job key uses MASTER_PORT and the pid of this parent),
    forward rank 0's result line, and return the first non-zero exit code of any rank (0 if all succeeded).  A rank
    that dies takes the others down with it instead of leaving them in a rendezvous."""
    import socket
    import subprocess
    import time

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))  # fmt: skip
        # rank 0 inherits stdout (its JSON line is the result); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if os.environ.get("JAXSIM_AMD_LIB"):
        # the developer knob of jaxsim_amd/_lib.py would let any library stand in for the product
        raise SystemExit("bench.py measures the in-tree library only: unset JAXSIM_AMD_LIB")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one command: `python bench.py --gpus N` is its own launcher (one child process per GPU, same arguments)
        raise SystemExit(spawn_ranks(args.gpus))
    # Model-specialised step kernels (jaxsim_amd/specialize.py): what `jax.jit` is to the reference -- the same
    # kernel source compiled with the model's integer flags as constants.  Built once per model (seconds, hipcc;
    # __graft_entry__.build() pre-builds the configurations of this file), outside every timed region.
    os.environ.setdefault("JAXSIM_AMD_SPECIALIZE", "1")
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL prints "Librccl path : ..." through
    # C stdio, flushed at exit) share fd 1: keep a private handle for the result line and point fd 1 at
    # stderr for everything else.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if args.steps < 1:
        raise SystemExit("--steps must be >= 1")

    multi = world > 1 or args.force_dist
    if args.dry_run_bootstrap:
        return dry_run_bootstrap(args, rank, world, result_out)
    import jaxsim_amd.api as js
    from jaxsim_amd import _lib, distributed, runtime

    runtime.require_device()
    runtime.set_device(local_rank % runtime.device_count() if args.share_device else local_rank)
    args.require_rccl = rccl_required(args, runtime.device_count(), world)
    # Multi-rank runs: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_PORT come from torch.distributed.run; the
    # ranks rendezvous through a temp file and talk RCCL through the C-ABI library only.  (torch is
    # NOT imported: its wheel bundles a second ROCm runtime with the same sonames, and two HIP
    # runtimes in one process do not both see the GPU -- measured on the MI355X box.)
    comm, comm_error = None, None
    if multi:
        # one node by contract: RCCL's out-of-band bootstrap can always use the loopback interface (a
        # container without any other interface would otherwise fail to pick one)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        try:
            comm = distributed.communicator_from_env()
        except Exception as e:  # keep the bench line: fall back to a host-side file collective
            comm_error = repr(e)
            if args.require_rccl:
                # [round 5] asked to be fatal: the launcher (torch.distributed.run or bench.py itself) takes the other ranks down
                print(f"bench.py: rank {rank}: --require-rccl and the RCCL communicator could not be created: {comm_error}", file=sys.stderr, flush=True)
                raise SystemExit(3)
            comm = distributed.FileCollective(rank, world, distributed.job_key())
    dtype = np.dtype(args.dtype)
    model = build_model(args.model)
    n_local = args.envs_per_gpu
    lo, hi = distributed.shard_bounds(n_local * world, rank, world)
    assert hi - lo == n_local
    data = synthetic_state(model, n_local, seed=rank, dtype=dtype)
    initial_block = data.state_block() if rank == 0 else None
    stream = runtime.Stream()
    runtime.set_stream(stream)
    dm = runtime.device_model(model, dtype)
    lib = _lib.load()
    import ctypes as C

    state_ptr = C.c_void_p(data._state.ptr)

    def run_steps(k):
        # one step kernel launch per step, enqueued from C (jxs_step_repeat is the host loop over jxs_step
        # without the per-call cost of the interpreter -- eight ranks share the host): blocks of 250 and 50
        # launches and the remainder are each captured once into a hipGraph and replayed
        if k > 0:
            _lib.check(lib.jxs_step_repeat(dm.handle, state_ptr, None, None, 2, n_local, k, stream.handle), "jxs_step_repeat")

    def timed_region(k):
        sec = C.c_double(0.0)
        _lib.check(lib.jxs_step_repeat_timed(dm.handle, state_ptr, None, None, 2, n_local, k, stream.handle, C.byref(sec)), "jxs_step_repeat_timed")
        return float(sec.value)

    def barrier():
        if comm is not None:
            comm.barrier()

    run_steps(args.warmup)
    # one untimed pass with the chunking of the timed regions: jxs_step_repeat captures its replay graphs on
    # first use, so that no capture falls inside a timed region
    run_steps(args.steps)
    stream.synchronize()

    # Timed regions: EXACTLY --steps launches each, >= 5 repetitions, the MEDIAN region is reported (the
    # region of a short request, e.g. --steps 20 = 0.2 ms, is otherwise at the mercy of one host hiccup).
    reps = 5 if args.steps >= 500 else 9
    wall, evs = timed_repetitions(timed_region, run_steps, args.steps, reps, stream, barrier, lib)
    per_rank_ms = None
    if comm is not None:  # max over ranks, repetition by repetition
        gathered = [np.asarray(comm.all_gather_scalars(w), dtype=np.float64) for w in wall]
        wall = [float(g.max()) for g in gathered]
        per_rank_ms = [float(x) / args.steps * 1e3 for x in np.median(np.stack(gathered), axis=0)]  # each rank's own median
    elapsed = float(np.median(wall))
    kernel_ms = float(np.median(evs)) * 1e3 / args.steps  # HIP events on the launch stream

    # secondary figure: the launch path at steady state (2000 launches, HIP events) when the timed regions are short
    steady = None
    if args.steps < 1000:
        run_steps(2000)
        stream.synchronize()
        es0, es1 = runtime.Event(), runtime.Event()
        es0.record(stream)
        run_steps(2000)
        es1.record(stream)
        stream.synchronize()
        us_ss = es0.elapsed_ms(es1) / 2000 * 1e3
        steady = {"launches": 2000, "us_per_step": us_ss, "env_steps_per_s_rank0": n_local / (us_ss * 1e-6),
                  "note": "same kernel and launch path (hipGraph replays of single-step launches), HIP events over 2000 launches; secondary figure"}

    # secondary figure: the generic kernel of the library (model flags read at run time) on the same workload
    generic = None
    from jaxsim_amd import specialize

    spec_modes = specialize.modes(dm)
    if rank == 0 and spec_modes:
        try:
            saved = os.environ.get("JAXSIM_AMD_SPECIALIZE")
            os.environ["JAXSIM_AMD_SPECIALIZE"] = "0"
            try:
                gmodel = build_model(args.model)
                gdata = synthetic_state(gmodel, n_local, seed=rank, dtype=dtype)
                gdm = runtime.device_model(gmodel, dtype)
            finally:
                os.environ["JAXSIM_AMD_SPECIALIZE"] = saved if saved is not None else "1"
            gp = C.c_void_p(gdata._state.ptr)
            _lib.check(lib.jxs_step_repeat(gdm.handle, gp, None, None, 2, n_local, 2000, stream.handle), "jxs_step_repeat")
            stream.synchronize()
            eg0, eg1 = runtime.Event(), runtime.Event()
            eg0.record(stream)
            _lib.check(lib.jxs_step_repeat(gdm.handle, gp, None, None, 2, n_local, 2000, stream.handle), "jxs_step_repeat")
            eg1.record(stream)
            stream.synchronize()
            us_g = eg0.elapsed_ms(eg1) / 2000 * 1e3
            generic = {"launches": 2000, "us_per_step": us_g, "env_steps_per_s_rank0": n_local / (us_g * 1e-6),
                       "note": "libjaxsim_amd.so's own kernel without a model-specialised build (JAXSIM_AMD_SPECIALIZE=0: the ahead-of-time variant with the common "
                               "feature switches as constants, 9.7 us with all flags read at run time), same launch path, HIP events over 2000 launches; secondary figure"}
            del gdata, gdm
        except Exception as e:  # secondary: never lose the headline for it
            generic = {"error": repr(e)}

    # secondary figure: the same K steps as ONE fused jxs_rollout launch (state in registers
    # between steps; what jax.lax.fori_loop over step is to the reference).  Not the headline.
    k_roll = max(args.steps, 200)
    rc = lib.jxs_rollout(dm.handle, state_ptr, None, None, 2, n_local, 10, stream.handle)
    runtime.synchronize(stream)
    ev2, ev3 = runtime.Event(), runtime.Event()
    ev2.record(stream)
    rc = rc or lib.jxs_rollout(dm.handle, state_ptr, None, None, 2, n_local, k_roll, stream.handle)
    ev3.record(stream)
    runtime.synchronize(stream)
    _lib.check(rc, "jxs_rollout")
    rollout_ms_per_step = ev2.elapsed_ms(ev3) / k_roll
    # ... and with a SEQUENCE of joint torques, one block per step (jxs_rollout_controlled: jax.lax.scan over step with
    # precomputed joint_force_references -- open-loop rollouts): one more load per step inside the same fused launch
    controlled_us, recorded_us = None, None
    try:
        n_j = model.dofs()
        seq_host = np.random.default_rng(3).uniform(-1.0, 1.0, size=(k_roll * n_j, n_local))
        seq = runtime.DeviceArray.from_host(seq_host, tile=data._state.tile, dtype=dtype)
        _lib.check(lib.jxs_rollout_controlled(dm.handle, state_ptr, C.c_void_p(seq.ptr), None, 2, n_local, k_roll, stream.handle), "jxs_rollout_controlled")
        runtime.synchronize(stream)
        ev4, ev5 = runtime.Event(), runtime.Event()
        ev4.record(stream)
        _lib.check(lib.jxs_rollout_controlled(dm.handle, state_ptr, C.c_void_p(seq.ptr), None, 2, n_local, k_roll, stream.handle), "jxs_rollout_controlled")
        ev5.record(stream)
        runtime.synchronize(stream)
        controlled_us = ev4.elapsed_ms(ev5) / k_roll * 1e3
        # ... and RECORDED: the state block after every step stored from registers (jxs_rollout_recorded: the stacked
        # outputs of a scan), with the torque sequence -- controls in, trajectory out, one launch
        traj = runtime.DeviceArray(k_roll * data._state.shape[0], n_local, dtype, tile=data._state.tile)
        tp = C.c_void_p(traj.ptr)
        _lib.check(lib.jxs_rollout_recorded(dm.handle, state_ptr, C.c_void_p(seq.ptr), 1, None, 2, n_local, k_roll, tp, stream.handle), "jxs_rollout_recorded")
        runtime.synchronize(stream)
        ev6, ev7 = runtime.Event(), runtime.Event()
        ev6.record(stream)
        _lib.check(lib.jxs_rollout_recorded(dm.handle, state_ptr, C.c_void_p(seq.ptr), 1, None, 2, n_local, k_roll, tp, stream.handle), "jxs_rollout_recorded")
        ev7.record(stream)
        runtime.synchronize(stream)
        recorded_us = ev6.elapsed_ms(ev7) / k_roll * 1e3
        del seq, traj
    except Exception as e:  # secondary: never lose the headline for it
        controlled_us = controlled_us if controlled_us is not None else repr(e)
        recorded_us = repr(e)

    # [round 5] What a USER of the reference gets: the literal loop `for _ in range(K): data = js.model.step(model, data)`
    # (README.md:80-83, tests/test_simulations.py:170-191) -- one Python call, one ctypes call, one plain launch per step,
    # a fresh data object per step (functional, like the reference's pytrees) or the same buffer (`inplace=True`, an
    # extension).  `value` above is the same launches enqueued from C (`jxs_step_repeat`); this is the interpreter's share.
    python_loop = None
    if rank == 0 and not args.no_python_loop:
        try:
            import time as _time

            K_py = 2000
            python_loop = {"steps": K_py, "idiom": "for _ in range(K): data = js.model.step(model, data)"}
            for label, kw in (("functional", {}), ("inplace", {"inplace": True})):
                d_py = js.data.JaxSimModelData.from_state_block(model, initial_block.astype(dtype), data.velocity_representation)
                for _ in range(50):
                    d_py = js.model.step(model, d_py, **kw)
                runtime.synchronize(stream)
                best = None
                for _rep in range(3):
                    t0 = _time.perf_counter()
                    for _ in range(K_py):
                        d_py = js.model.step(model, d_py, **kw)
                    t_enq = _time.perf_counter() - t0  # the interpreter is done enqueueing
                    runtime.synchronize(stream)
                    t_all = _time.perf_counter() - t0
                    if best is None or t_all < best[1]:
                        best = (t_enq, t_all)
                python_loop[label] = {"us_per_call": best[1] / K_py * 1e6, "host_enqueue_us_per_call": best[0] / K_py * 1e6,
                                      "env_steps_per_s": n_local * K_py / best[1]}
                del d_py
        except Exception as e:  # secondary: never lose the headline for it
            python_loop = {"error": repr(e)}

    # secondary figures: the same kernel at larger batches on THIS GPU -- 8192 environments (the one-GPU point of
    # north_star's 1 / 2 / 4 / 8 curve) and the chip saturated (64 Ki environments: what the step
    # costs once enough waves hide each other's latencies).  Not the headline configuration.
    def one_gpu_batch(n_big):
        reps_big = -(-n_big // n_local)
        big = js.data.JaxSimModelData.from_state_block(
            model, np.tile(initial_block, (1, reps_big))[:, :n_big].astype(dtype), data.velocity_representation
        )
        bp = C.c_void_p(big._state.ptr)
        # [round 6] 300 untimed steps, then the median of three timed regions of 300 (tools/sweep.py's protocol).  Rounds 4 - 5
        # timed ONE region of 200 steps behind 20: at batch 8192 that is 0.4 ms of warm-up in front of 3.6 ms on a device that
        # has idled through the host-side phases before it -- 18.4 us where the sweep of the same kernel measures 15.7.
        _lib.check(lib.jxs_step_repeat(dm.handle, bp, None, None, 2, n_big, 300, stream.handle), "jxs_step_repeat")
        regions = []
        for _ in range(3):
            ev4, ev5 = runtime.Event(), runtime.Event()
            ev4.record(stream)
            _lib.check(lib.jxs_step_repeat(dm.handle, bp, None, None, 2, n_big, 300, stream.handle), "jxs_step_repeat")
            ev5.record(stream)
            runtime.synchronize(stream)
            regions.append(ev4.elapsed_ms(ev5) / 300 * 1e3)
        us = float(np.median(regions))
        res = {"envs": n_big, "us_per_step": us, "env_steps_per_s": n_big / (us * 1e-6), "us_per_step_regions": regions,
               "finite_envs": float(np.isfinite(big.state_block()).all(axis=0).mean()), "steps_taken": 1200}
        del big
        return res

    saturated = None
    if world == 1 and args.saturated_envs > 0:
        try:
            saturated = one_gpu_batch(args.saturated_envs)
            saturated["finite_note"] = ("the rate at which environments leave is the explicit contact model's, the same in the fp64 oracle: "
                                        "profiles/r04_divergence_audit.txt (65536 distinct states x 1000 steps: HIP fp32 47, HIP fp64 51, C port fp32 47, C port fp64 51 gone)")
        except Exception as e:  # secondary: never lose the headline for it
            saturated = {"error": repr(e)}
    batch_8192 = None
    if world == 1 and args.saturated_envs > 0 and n_local != 8192:
        try:
            batch_8192 = one_gpu_batch(8192)
        except Exception as e:
            batch_8192 = {"error": repr(e)}
    saturated_4x = None  # [round 6] four times the saturating batch: the tail of the last round of waves weighs less (>= 750 M)
    if world == 1 and args.saturated_envs > 0:
        try:
            saturated_4x = one_gpu_batch(4 * args.saturated_envs)
        except Exception as e:
            saturated_4x = {"error": repr(e)}

    # final state concat: ONE RCCL all-gather over xGMI, outside the timed region
    allgather_ms = None
    final = data.state_block()
    allgather_error = None
    comm_ranks = None
    if comm is not None:
        comm_ranks = int(comm.world_size)
        try:
            if comm_error is not None:
                raise RuntimeError(f"RCCL communicator unavailable: {comm_error}")
            runtime.synchronize(stream)
            barrier()
            t1 = time.perf_counter()
            full = distributed.all_gather_state(comm, data)
            allgather_ms = (time.perf_counter() - t1) * 1e3
            lo = rank * n_local
            if full.shape != (final.shape[0], n_local * world) or not np.array_equal(full[:, lo : lo + n_local], final, equal_nan=True):
                allgather_error = "gathered state does not contain this rank's shard"
        except Exception as e:  # the gather is outside the timed region: report, do not lose the line
            allgather_error = repr(e)
    nonfinite_envs = int((~np.isfinite(final).all(axis=0)).sum())
    # [round 5] what the communicator was: library version, and the device every rank ran on (PCI bus ids, gathered as
    # integers domain << 16 | bus << 8 | device << 3 | function through the communicator itself)
    import ctypes as _C

    nccl_version, pci_ids = None, None
    # [ADVICE r5] the local code is computed OUTSIDE the try block around the gather: a rank that failed to read its
    # bus id still enters the collective (with code -1) instead of leaving the other ranks waiting in it
    my_pci, code = None, -1.0
    try:
        buf = _C.create_string_buffer(32)
        _lib.check(lib.jxs_device_pci_bus_id(buf, 32), "jxs_device_pci_bus_id")
        my_pci = buf.value.decode()
        dom, bus, devfn = my_pci.split(":")
        dev_, fn_ = devfn.split(".")
        code = float((int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev_, 16) << 3) | int(fn_, 16))
    except Exception as e:  # reporting only
        my_pci = repr(e)
    try:
        if comm is not None:
            codes = [int(c) for c in comm.all_gather_scalars(code)]
            pci_ids = [f"{c >> 16:04x}:{(c >> 8) & 0xff:02x}:{(c >> 3) & 0x1f:02x}.{c & 7:x}" if c >= 0 else None for c in codes]
        else:
            pci_ids = [my_pci]
        if comm is not None and comm_error is None:
            v = _C.c_int(0)
            _lib.check(lib.jxs_comm_version(_C.byref(v)), "jxs_comm_version")
            nccl_version = int(v.value)
    except Exception as e:  # reporting only
        pci_ids = pci_ids if pci_ids is not None else repr(e)

    if rank == 0:
        lay = dm.layout
        n_total = n_local * world
        value = n_total * args.steps / elapsed
        n, n_cp = lay.n_joints, lay.n_points
        alg_bytes_per_env = (2 * (13 + 2 * n + 3 * n_cp) + n) * dtype.itemsize
        achieved_gbs = alg_bytes_per_env * n_local / (kernel_ms * 1e-3) / 1e9
        valu_count = kernel_valu_count(dm) if dtype == np.float32 else None
        traffic, traffic_note = profiled_traffic(args.model, n_local, dtype.name)
        tname = "float" if dtype == np.float32 else "double"
        out = {
            "metric": "env-steps/sec (whole node), iCub 23-DoF soft-contact, batch 1024/8192",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if dtype == np.float32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} synthetic floating-base humanoid, soft contacts (K={model.contact_params.K:.4g}, D={model.contact_params.D:.4g}, mu=0.5), "
                f"semi-implicit Euler dt=1e-3, nL={lay.n_links} n={n} n_cp={n_cp}, "
                f"{n_local} envs per GPU x {world} GPU(s) = {n_total} envs, one step kernel launch per step (jxs_step_repeat: hipGraph replays of blocks of 250 / 50 launches captured during warm-up, fewer than 50 launched plainly)",
                "note_on_value": "`value` is the wall clock of the median timed region of exactly --steps launches, launch and "
                "synchronisation latency of the region included (~12 us per region: 9 % at --steps 20, 0.1 % at 2000); "
                "`steady_state` is the same launch path measured over >= 2000 launches",
                "envs_per_gpu": n_local,
                "global_batch": n_total,
                "lanes_per_env": int(lay.group),
                "aba_layout": "row-distributed (8 lanes per active link)" if lay.row_mode else "link per lane",
                "kernel_variant": ("model-specialised: the step kernel compiled with this model's integer flags (tree shape, level masks, feature switches) as "
                                   "constants, as jax.jit does for the reference; physical parameters are run-time data; `generic_kernel` is the library's run-time-flag kernel"
                                   if spec_modes else "generic (model flags read at run time)"),
                "parallelism": f"batch-sharded x{world}, no per-step communication",
                # which native object the timed launches ran through (jaxsim_amd/specialize.py; built on first use)
                "specialised_object": specialize.attached_files(dm).get(specialize.MODE_STEP),
                "specialised_for": specialize.spec(model, dtype, specialize.MODE_STEP) if spec_modes else None,
            },
            "timing": {
                "repetitions": reps,
                "statistic": "median over repetitions of one timed region of exactly `steps` launches (barrier + stream sync on both sides of every region; max over ranks per repetition)",
                "wall_us_per_step_each": [w / args.steps * 1e6 for w in wall],
                "event_us_per_launch_each": [e / args.steps * 1e6 for e in evs],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_env_step": alg_bytes_per_env,
                "kernel": (f"jxs_launch_spec::jxs_kernel<{tname},{lay.group},MODE_STEP> (model-specialised build of the step kernel, jaxsim_amd/specialize.py)"
                           if spec_modes else f"jxs_kernel<{tname},{lay.group},MODE_STEP>"),
                "kernel_avg_launch_us": kernel_ms * 1e3,
                "fp32_flop_model_per_env_step": FLOPS_PER_ENV_STEP,
                "fp32_frac_of_vector_peak": FLOPS_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                **issue_figures(valu_count, 64 // lay.group, n_local, kernel_ms * 1e3),
            },
            "nonfinite_envs_rank0": nonfinite_envs,
            "nonfinite_note": "fp32 + explicit contacts at their stability limit: single environments can leave the finite range after some "
                              "hundred steps (index 723 of seed 0 does, after 864 steps); one-step parity with the oracle holds along "
                              "that trajectory, HISTORY.md section 7",
            "allgather_ms": allgather_ms,
            "allgather_error": allgather_error,
            "comm": None if comm is None else {"kind": "rccl" if comm_error is None else "file_collective", "class": type(comm).__name__, "ranks": comm_ranks,
                                               "error": comm_error, "nccl_version": nccl_version, "require_rccl": bool(args.require_rccl),
                                               "distinct_devices": (len(set(pci_ids)) == len(pci_ids)) if isinstance(pci_ids, list) else None,
                                               "ms_per_step_per_rank": per_rank_ms},
            "device_pci_bus_ids": pci_ids,
            "steady_state": steady,
            "python_step_loop": python_loop,
            "generic_kernel": generic,
            "fused_rollout": {"us_per_step": rollout_ms_per_step * 1e3, "env_steps_per_s_rank0": n_local / (rollout_ms_per_step * 1e-3),
                              "controlled_us_per_step": controlled_us,
                              "controlled_and_recorded_us_per_step": recorded_us,
                              "note": "same steps as one jxs_rollout launch; `controlled`: with one block of joint torques per step "
                                      "(jxs_rollout_controlled); `recorded`: and the state block after every step stored (jxs_rollout_recorded); secondary figures, not `value`"},
        }
        if saturated is not None:
            if "env_steps_per_s" in saturated:
                saturated["hbm_frac"] = alg_bytes_per_env * saturated["env_steps_per_s"] / 1e9 / HBM_PEAK_GBS
                saturated["fp32_frac_of_vector_peak"] = FLOPS_PER_ENV_STEP * saturated["env_steps_per_s"] / 1e12 / FP32_PEAK_TFLOPS
                saturated["note"] = "same step kernel, one GPU filled; secondary figure, not `value`"
                saturated.update(issue_figures(valu_count, 64 // lay.group, saturated["envs"], saturated["us_per_step"]))
            out["saturated"] = saturated
        if saturated_4x is not None:
            if "env_steps_per_s" in saturated_4x:
                saturated_4x["hbm_frac"] = alg_bytes_per_env * saturated_4x["env_steps_per_s"] / 1e9 / HBM_PEAK_GBS
                saturated_4x["note"] = "four times the saturating batch on one GPU; secondary figure, not `value`"
                saturated_4x.update(issue_figures(valu_count, 64 // lay.group, saturated_4x["envs"], saturated_4x["us_per_step"]))
            out["saturated_4x"] = saturated_4x
        if batch_8192 is not None:
            if "env_steps_per_s" in batch_8192:
                batch_8192["hbm_frac"] = alg_bytes_per_env * batch_8192["env_steps_per_s"] / 1e9 / HBM_PEAK_GBS
                batch_8192["note"] = "north_star's batch 8192 on ONE GPU (the first point of the 1 / 2 / 4 / 8 strong-scaling curve); same step kernel; secondary figure, not `value`"
                batch_8192.update(issue_figures(valu_count, 64 // lay.group, batch_8192["envs"], batch_8192["us_per_step"]))
            out["global_batch_8192_one_gpu"] = batch_8192
        if world == 1 and not args.no_other_contact_models:
            try:
                out["default_contact_params"] = default_contact_params_variant(args.model, n_local, dtype, stream)
            except Exception as e:  # secondary: never lose the headline for it
                out["default_contact_params"] = {"error": repr(e)}
            out["other_contact_models"] = other_contact_models(dtype, stream)
            try:
                out["other_configs"] = other_configs(stream)
            except Exception as e:
                out["other_configs"] = {"error": repr(e)}
            try:
                other = np.dtype(np.float64 if dtype == np.float32 else np.float32)
                out["other_precision"] = secondary_dtype(args.model, n_local, other, stream)
            except Exception as e:
                out["other_precision"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, initial_block, args.cpu_baseline_seconds)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}  # fmt: skip
        print(json.dumps(out), file=result_out, flush=True)

    if comm is not None:
        comm.barrier()


if __name__ == "__main__":
    main()
