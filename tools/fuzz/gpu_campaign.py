#!/usr/bin/env python3
"""[round 5] The fuzz campaigns ON THE DEVICE.  The campaigns of this directory run the kernel SOURCES in the host
emulation; this tool runs the same kind of random trees through the PRODUCT (libjaxsim_amd.so on an MI355X, through
`js.model.step`) and compares with the oracle's truth AND with the emulation's result of the same case, so that the
statement "what the emulation shows is what the device does" is itself measured over a campaign, not on the zoo only.

    python tools/fuzz/gpu_campaign.py prepare tools/fuzz/_cases.pkl [seed] [trials]     # here (CPU): cases, truth, emulation
    python tools/fuzz/gpu_campaign.py prebuild tools/fuzz/_cases.pkl K                  # here: model-specialised kernels of K sampled cases (hipcc)
    python tools/fuzz/gpu_campaign.py run     tools/fuzz/_cases.pkl [out.txt]            # on the GPU box

`run` goes through the library's kernels (JAXSIM_AMD_SPECIALIZE=0: a model-specialised build per random tree would be a
compiler run per case) and then repeats the K prebuilt cases through their model-specialised kernels (`=require`).

Cases: random trees of 1 to 40 links (robots.chain_urdf: serial to bushy, revolute / prismatic, every third with
parallel or axis-aligned joints, fixed and floating bases), collision boxes on one to eight random links, a random
subset of the points enabled; SoftContacts / RelaxedRigidContacts / RigidContacts; semi-implicit Euler / RungeKutta4;
fp64 and fp32; a random cap of the lane group for the chunked tree solve; the three velocity representations; half of
the cases with joint torques and link forces; a quarter with position limits / joint friction / the torque-speed curve;
a quarter on a tilted ground plane.  `prepare` drops the cases the model
constructor refuses and the ones the ORACLE cannot solve (LinAlgError), and records both counts.
TEST INFRASTRUCTURE (uses oracle/ and tests/emul): not part of the product."""
import os, pickle, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

TOL = {("soft", "float64"): 1e-9, ("soft", "float32"): 2e-4, ("relaxed", "float64"): 1e-8, ("relaxed", "float32"): 3e-3, ("rigid", "float64"): 1e-5}
# (device against emulation: the same arithmetic up to the contraction of multiply-adds and the hardware reciprocals)
TOL_EMUL = {"float64": 1e-9, "float32": 3e-3}
RIGID_TOL_EMUL = 1e-5  # RigidContacts at solver_tol 1e-9 .. 1e-3: the iterates agree to where the iteration stops


def make_model(case):
    import helpers
    import jaxsim_amd as ja
    from jaxsim_amd import robots

    if "hub" in case["tree"]:  # [round 6] a hub with seven to twelve legs: more than six children on one link
        base = ja.JaxSimModel.build_from_model_description(robots.hub_urdf(**case["tree"]["hub"]))
    else:
        base = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(**case["tree"]))
    kind, idx = case["kind"], case["idx"]
    if kind == "relaxed":
        model = helpers.relaxed_model(base, idx, mu=case["mu"])
    elif kind == "rigid":
        model = helpers.rigid_model(base, idx, K=1e4, D=1e2, build=dict(solver_options={"solver_tol": case["solver_tol"]}))
    else:
        model = helpers.enable_points(base, idx)
    if case["rk4"]:
        model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4)
    if case.get("actuation") is not None:  # position limits with spring and damper, joint friction, torque-speed curve
        model = helpers.actuation_variant(model, case["actuation"])
    if case.get("terrain") is not None and case["terrain"][0] == "hf":  # [round 6] a height field: a * sin(kx x + p) cos(ky y) on a 0.1 m grid
        _, a, kx, ky, ph = case["terrain"]
        model = helpers.with_params(model, terrain=ja.HeightFieldTerrain.from_function(lambda x, y: a * np.sin(kx * x + ph) * np.cos(ky * y),
                                                                                        x_range=(-2.5, 2.5), y_range=(-2.5, 2.5), spacing=0.1))
    elif case.get("terrain") is not None:  # a tilted ground plane
        model = helpers.with_params(model, terrain=ja.PlaneTerrain.build(height=case["terrain"][0], normal=list(case["terrain"][1:])))
    return model


def set_knobs(case):
    os.environ.pop("JXS_CT_CHUNK_LANES", None)
    if case["cap"]:
        os.environ["JXS_CT_CHUNK_LANES"] = case["cap"]


def prepare(path, seed, trials):
    import emul_binding as eb, helpers, oracle

    REPR_CODE = {oracle.VelRepr.Inertial: 0, oracle.VelRepr.Body: 1, oracle.VelRepr.Mixed: 2}
    rng = np.random.default_rng(seed)
    cases, refused, oracle_failed = [], 0, 0
    for trial in range(trials):
        n_links = int(rng.integers(1, 41))
        fixed = bool(rng.integers(0, 5) == 0) and n_links > 1
        ncl = int(rng.integers(1, min(8, n_links) + 1))
        cl = tuple(sorted(set(int(v) for v in rng.choice(np.arange(0, n_links), size=ncl, replace=False))))
        if fixed and cl == (0,):
            cl = (n_links - 1,)
        tree = dict(n_links=n_links, fixed_base=fixed, seed=20000 + trial, max_back=int(rng.integers(1, 4)), collision_links=cl,
                    parallel_axes=[None, "all", "aligned", None][trial % 4], base_offset=(0.0, 0.0, 0.0))
        if trial % 10 == 7:  # [round 6] every tenth tree a hub with 7 .. 12 legs (kMaxChildren = 12)
            legs, per = int(rng.integers(7, 13)), int(rng.integers(1, 3))
            feet = int(rng.integers(1, min(4, legs) + 1))
            tree = dict(hub=dict(n_legs=legs, links_per_leg=per, foot_boxes=feet, seed=20000 + trial), seed=20000 + trial)
            cl = tuple(range(feet))
        npts = 8 * len(cl)
        k = int(rng.integers(1, npts + 1)) if trial % 2 else npts
        idx = sorted(int(v) for v in rng.choice(npts, size=k, replace=False))
        kind = ["soft", "relaxed", "relaxed", "rigid"][int(rng.integers(0, 4))]
        if kind == "rigid" and len(idx) > 12:
            idx = idx[:12]  # (the dense interior-point emulation of a 64-lane group takes minutes beyond that)
        rk4 = bool(rng.integers(0, 4) == 0)
        cap = [None, None, "8", "16"][int(rng.integers(0, 4))] if (kind == "relaxed" and not rk4) else None
        mu = float(rng.choice([0.3, 0.5, 0.8]))
        solver_tol = float(rng.choice([1e-9, 1e-6]))
        rep = [oracle.VelRepr.Inertial, oracle.VelRepr.Body, oracle.VelRepr.Mixed][int(rng.integers(0, 3))]
        with_inputs = bool(rng.integers(0, 2))  # joint torques and link forces (given in the data's velocity representation)
        actuation = int(20000 + trial) if rng.integers(0, 4) == 0 else None
        terrain = (float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), 1.0) if rng.integers(0, 4) == 0 else None
        if terrain is None and rng.integers(0, 3) == 0:  # [round 6] a quarter of the cases on a height field
            terrain = ("hf", float(rng.uniform(0.02, 0.15)), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0, 6.28)))
        for dtype in (np.float64, np.float32):
            if kind == "rigid" and dtype == np.float32:
                continue  # (fp32 RigidContacts: the gate is the model's own sensitivity, measured elsewhere: tools/fp32_error_gpu.py)
            case = dict(trial=trial, tree=tree, kind=kind, idx=idx, rk4=rk4, cap=cap, mu=mu, solver_tol=solver_tol, dtype=np.dtype(dtype).name,
                        actuation=actuation, terrain=terrain, tau=None, f=None)
            set_knobs(case)
            try:
                model = make_model(case)
                d = oracle.random_model_data(model, batch_size=4, seed=tree["seed"], dtype=dtype, velocity_representation=rep,
                                             base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.3)), base_rpy_bounds=((-0.4, -0.4, -3), (0.4, 0.4, 3)))
                kw_o, kw_e = {}, {}
                if with_inputs:
                    case["tau"], case["f"] = helpers.random_inputs(model, 4, tree["seed"], dtype)
                    kw_o = dict(link_forces=case["f"].astype(np.float64), joint_force_references=case["tau"].astype(np.float64))
                    kw_e = dict(tau=case["tau"].T, link_forces=case["f"].reshape(4, -1).T, force_repr=REPR_CODE[rep])
                truth = helpers.odata_to_block(model, oracle.step(model, helpers.upcast(d, model) if dtype == np.float32 else d, **kw_o))
                state = helpers.odata_to_block(model, d)
                emul = eb.run(model, eb.MODE_STEP, state, **kw_e)
            except RuntimeError as ex:
                refused += 1
                print("refused", trial, kind, case["dtype"], str(ex)[:90])
                continue
            except np.linalg.LinAlgError:
                oracle_failed += 1
                continue
            ref32_err, sens, scatter = float("nan"), None, None
            if dtype == np.float32:  # the REFERENCE'S formulation evaluated in fp32 (the oracle on float32 arrays): what fp32 costs this model anyway
                try:
                    with np.errstate(all="ignore"):
                        ref32_err = float(helpers.rel_err(helpers.odata_to_block(model, oracle.step(model, d, **({} if not with_inputs else dict(link_forces=case["f"], joint_force_references=case["tau"])))), truth))
                except np.linalg.LinAlgError:
                    pass
                try:  # ... and how far one ulp of input noise moves the fp64 oracle's step, per environment (contact edges)
                    with np.errstate(all="ignore"):
                        sens = helpers.oracle_sensitivity(model, d, trials=3, **kw_o)
                except np.linalg.LinAlgError:
                    pass
                # ... and the scatter of the fp32 evaluation itself: the EMULATION of these sources on six inputs one ulp away
                # (random signs) against its result on the input -- rounding anywhere inside the step is amplified like this
                prng, ulp, base = np.random.default_rng(trial), np.spacing(np.abs(state)), emul.astype(np.float64)
                scatter = np.zeros(state.shape[1])
                for _ in range(6):
                    o = eb.run(model, eb.MODE_STEP, (state + prng.choice([-1.0, 1.0], size=state.shape).astype(np.float32) * ulp).astype(np.float32), **kw_e).astype(np.float64)
                    scatter = np.maximum(scatter, (np.abs(o - base) / np.maximum(1.0, np.abs(base))).max(axis=0))
            if dtype == np.float64 and trial % 3 == 0:  # fp64 extras: a 3-step rollout (with the state of every step) and the gravity-compensated step
                try:
                    dk, traj = d, []
                    for _ in range(3):
                        dk = oracle.step(model, dk, **kw_o)
                        traj.append(helpers.odata_to_block(model, dk))
                    if np.isfinite(traj[-1]).all() and np.abs(traj[-1]).max() < 1e4:  # (a trajectory the ORACLE loses or that explodes -- the default stiffness on 0.5 kg links -- is no truth)
                        case["rollout3"] = np.stack(traj)
                        case["rollout3_emul_err"] = float(helpers.rel_err(eb.run(model, eb.MODE_STEP, state, n_steps=3, **kw_e), traj[-1]))
                    if not with_inputs and model.dofs() > 0:
                        g = oracle.free_floating_gravity_forces(model, d)[:, 6:]
                        case["gravcomp"] = helpers.odata_to_block(model, oracle.step(model, d, joint_force_references=g))
                except np.linalg.LinAlgError:
                    pass
            case.update(state=state, truth=truth, emul=emul, rep=str(d.velocity_representation), ref32_err=ref32_err, sens=sens, scatter=scatter,
                        group=int(eb.layout(model, dtype).group), emul_err=float(helpers.rel_err(emul, truth)))
            cases.append(case)
    os.environ.pop("JXS_CT_CHUNK_LANES", None)
    with open(path, "wb") as f:
        pickle.dump(dict(seed=seed, trials=trials, refused=refused, oracle_failed=oracle_failed, cases=cases), f)
    print("prepared", len(cases), "cases; refused", refused, "oracle failed", oracle_failed)


def sample_of(blob, k):
    """Every (len / k)-th case: the same sample in `prebuild` and in `run`."""
    n = len(blob["cases"])
    return list(range(0, n, max(1, n // max(1, k))))[:k]


def _prebuild_one(args):
    path, i = args
    from jaxsim_amd import specialize

    with open(path, "rb") as f:
        case = pickle.load(f)["cases"][i]
    set_knobs(case)
    model = make_model(case)
    modes = list(specialize.modes_of(model))  # (`require` wants every mode of the model ...
    if case.get("gravcomp") is not None:      # ... and the gravity-torque kernel behind js.model.gravity_compensation_torques of the fp64 extras)
        modes.append(specialize.MODE_GRAV)
    return [specialize.compile(model, np.dtype(case["dtype"]), mode).name for mode in modes]


def prebuild(path, k):
    from concurrent.futures import ProcessPoolExecutor

    with open(path, "rb") as f:
        blob = pickle.load(f)
    blob["specialised"] = sample_of(blob, k)
    with open(path, "wb") as f:
        pickle.dump(blob, f)
    with ProcessPoolExecutor(min(32, os.cpu_count() or 4)) as ex:
        names = list(ex.map(_prebuild_one, [(path, i) for i in blob["specialised"]]))
    print("prebuilt", len(set(n for ns in names for n in ns)), "objects for", len(names), "cases")


def run(path, out_path):
    import helpers, oracle
    import jaxsim_amd as ja
    import jaxsim_amd.api as js
    from jaxsim_amd import _lib, specialize

    with open(path, "rb") as f:
        blob = pickle.load(f)
    lines, worst, worst_emul, count, nfail, stats, widened, widened64, extras = [], {}, {}, {}, 0, {}, 0, 0, {}
    REP = {oracle.VelRepr.Inertial: ja.VelRepr.Inertial, oracle.VelRepr.Body: ja.VelRepr.Body, oracle.VelRepr.Mixed: ja.VelRepr.Mixed}
    todo = [(c, False) for c in blob["cases"]] + [(blob["cases"][i], True) for i in blob.get("specialised", [])]
    for case, specialised in todo:
        os.environ["JAXSIM_AMD_SPECIALIZE"] = "require" if specialised else "0"
        set_knobs(case)
        if not os.environ.get("GPU_CAMPAIGN_DRY"):
            _lib.check(_lib.load().jxs_debug_reload_env(), "jxs_debug_reload_env")
        if os.environ.get("GPU_CAMPAIGN_DRY"):  # self-test of this tool without a device: the emulation's result stands in
            out = case["emul"]
        else:
            model = make_model(case)
            data = js.data.JaxSimModelData.from_state_block(model, case["state"], REP[case["rep"]])
            out = js.model.step(model, data, **({} if case.get("tau") is None else dict(link_forces=case["f"], joint_force_references=case["tau"]))).state_block()
        e, ee = float(helpers.rel_err(out, case["truth"])), float(helpers.rel_err(out, case["emul"]))
        if not os.environ.get("GPU_CAMPAIGN_DRY") and case.get("rollout3") is not None:
            kw = {} if case.get("tau") is None else dict(link_forces=case["f"], joint_force_references=case["tau"])
            fin, states = js.model.rollout(model, data, 3, return_trajectory=True, **kw)
            er = max(float(helpers.rel_err(states[k], case["rollout3"][k])) for k in range(3))
            er = max(er, float(helpers.rel_err(fin.state_block(), case["rollout3"][2])), float(helpers.rel_err(js.model.rollout(model, data, 3, **kw).state_block(), case["rollout3"][2])))
            extras.setdefault((case["kind"], "rollout of 3 steps, recorded and not"), []).append((er, case["rollout3_emul_err"]))
            if case.get("gravcomp") is not None:
                eg = float(helpers.rel_err(js.model.step(model, data, gravity_compensation=True).state_block(), case["gravcomp"]))
                eg = max(eg, float(helpers.rel_err(js.model.step(model, data, joint_force_references=js.model.gravity_compensation_torques(model, data)).state_block(), case["gravcomp"])))
                extras.setdefault((case["kind"], "gravity-compensated step, one launch and two"), []).append((eg, 0.0))
        chunked = case["cap"] is not None and len(case["idx"]) > int(case["cap"])
        key = (case["kind"] + ("/chunked" if chunked else "") + (" [specialised]" if specialised else ""), "rk4" if case["rk4"] else "euler", case["dtype"])
        if specialised and not os.environ.get("GPU_CAMPAIGN_DRY"):
            from jaxsim_amd import runtime

            assert specialize.attached_files(runtime.device_model(model, np.dtype(case["dtype"]))), "the model-specialised kernel was not attached"
        count[key] = count.get(key, 0) + 1
        worst[key] = max(worst.get(key, 0.0), e)
        worst_emul[key] = max(worst_emul.get(key, 0.0), ee)
        tol = TOL[(case["kind"], case["dtype"])]
        tol_e = RIGID_TOL_EMUL if case["kind"] == "rigid" else TOL_EMUL[case["dtype"]]
        if case["dtype"] == "float32":
            # fp32 on random trees of light links under the reference's default (stiff) contact parameters: the step
            # amplifies rounding by orders of magnitude in some states, in EVERY fp32 evaluation -- the gate of such a case is
            # what the reference's own formulation loses in fp32, what the emulation of these sources loses, and -- per
            # environment -- ten times what ONE ULP of input noise does to the fp64 oracle (an environment on an edge of the
            # discontinuous contact model lands on either side) and three times what it does to the fp32 emulation, not a constant
            r32 = case["ref32_err"] if np.isfinite(case["ref32_err"]) else 0.0
            base = max(tol, 3.0 * max(r32, case["emul_err"]))
            bound = np.maximum(base, 10.0 * (case["sens"] if case["sens"] is not None else 0.0))
            bound = np.maximum(bound, 3.0 * case["scatter"])
            per_env = lambda a, b: (np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))).max(axis=0)  # noqa: E731
            ok = bool((per_env(out, case["truth"]) < bound).all() and (per_env(out, case["emul"].astype(np.float64)) < bound).all())
            widened += int(ok and not ((e < base) and (ee < base)))
        else:
            # (fp64: a 40-link tree under RungeKutta4 at the default contact stiffness can amplify 1e-16 to 1e-8 -- in the device AND in
            # the emulation; such a case is held to three times the emulation's distance to the truth, and counted)
            ok = (e < max(tol, 3.0 * case["emul_err"])) and (ee < max(tol_e, 3.0 * case["emul_err"]))
            widened64 += int(ok and not ((e < tol) and (ee < tol_e)))
        stats.setdefault(key, []).append((e, case["emul_err"], case["ref32_err"]))
        if not ok:
            nfail += 1
            lines.append("FAIL trial %d %s nL %d fixed %s coll %s points %d: device-truth %.2e (emulation-truth %.2e, reference formulation in fp32 %.2e) device-emulation %.2e"
                         % (case["trial"], key, case["tree"]["n_links"], case["tree"]["fixed_base"], case["tree"]["collision_links"], len(case["idx"]), e, case["emul_err"], case["ref32_err"], ee))
    os.environ.pop("JXS_CT_CHUNK_LANES", None)
    # the fp64 extras: three steps compound the one-step distance (RigidContacts: where the iteration stops, three times)
    extra_lines = []
    XTOL = {"soft": 1e-8, "relaxed": 1e-7, "rigid": 1e-4}
    for key in sorted(extras):
        v = np.array(extras[key])
        # (a diverging trajectory amplifies 1e-16 like anything else: such a case is held to three times the emulation's own distance)
        bad = int(np.sum(v[:, 0] >= np.maximum(XTOL[key[0]], 3.0 * v[:, 1])))
        nfail += bad
        extra_lines.append("%-10s %-46s float64 %6d | device vs truth: worst %.2e median %.2e, above %.0e: %d of which beyond 3 x the emulation's distance: %d"
                           % (key[0], key[1], len(v), v[:, 0].max(), np.median(v[:, 0]), XTOL[key[0]], int(np.sum(v[:, 0] >= XTOL[key[0]])), bad))
    lines.append("campaign seed %d, %d trees: %d cases compared on the device (prepare: %d refused by the model constructor, %d the oracle could not solve); fails %d; "
                 "fp32 cases that pass by their own measured sensitivity only (above the class tolerance and above 3 x the emulation / the reference formulation in fp32): %d; fp64 cases above the class tolerance that pass by 3 x the emulation's own distance to the truth: %d"
                 % (blob["seed"], blob["trials"], sum(count.values()), blob["refused"], blob["oracle_failed"], nfail, widened, widened64))
    lines.append("distance to the fp64 truth (rel., max over the state rows and the 4 environments of a case): worst / median over the cases; 'above' = cases above the class tolerance")
    lines.append("%-30s %-6s %-8s %6s | %-28s | %-28s | %-28s | %s" % ("contact model", "integr", "dtype", "cases", "device: worst median above", "emulation: worst median above",
                                                                 "reference form. fp32: same", "device vs emulation: worst"))
    for key in sorted(count):
        a = np.array(stats[key], dtype=np.float64)
        tol = TOL[(key[0].split("/")[0].split(" ")[0], key[2])]
        col = lambda v: "%.2e %.2e %4d" % (np.nanmax(v), np.nanmedian(v), int(np.sum(v >= tol))) if np.isfinite(v).any() else "-"  # noqa: E731
        lines.append("%-30s %-6s %-8s %6d | %-28s | %-28s | %-28s | %.2e" % (key[0], key[1], key[2], count[key], col(a[:, 0]), col(a[:, 1]), col(a[:, 2]), worst_emul[key]))
    lines.extend(extra_lines)
    text = "\n".join(lines)
    print(text)
    if out_path:
        with open(out_path, "w") as f:
            f.write(text + "\n")
    return nfail


if __name__ == "__main__":
    cmd, path = sys.argv[1], sys.argv[2]
    if cmd == "prebuild":
        prebuild(path, int(sys.argv[3]) if len(sys.argv) > 3 else 32)
    elif cmd == "prepare":
        prepare(path, int(sys.argv[3]) if len(sys.argv) > 3 else 31, int(sys.argv[4]) if len(sys.argv) > 4 else 200)
    else:
        sys.exit(1 if run(path, sys.argv[3] if len(sys.argv) > 3 else None) else 0)
