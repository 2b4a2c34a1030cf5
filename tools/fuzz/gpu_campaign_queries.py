#!/usr/bin/env python3
"""[round 5] The QUERY kernels and the FUSED ROLLOUTS on the device, over random trees (companion of gpu_campaign.py,
which covers `step`).  Truth is prepared on the host by the oracle; the device answers through the product's Python
API (`js.model.*`, the reference's names):

    forward_dynamics_aba, inverse_dynamics, free_floating_bias_forces, free_floating_gravity_forces,
    the cached link transforms / velocities, free_floating_mass_matrix (+ inverse: M Minv = 1),
    jacobian_full_doubly_left (+ derivative), rollout(k) against k oracle steps, rollout with a torque sequence

    python tools/fuzz/gpu_campaign_queries.py prepare tools/fuzz/_cases_q.pkl [seed] [trials]   # here (CPU)
    python tools/fuzz/gpu_campaign_queries.py run     tools/fuzz/_cases_q.pkl [out.txt]          # on the GPU box

fp64 gates are constants; an fp32 quantity is held to max(constant, 3 x what the REFERENCE'S formulation loses when the
oracle evaluates it on float32 arrays) -- on random trees of light links the conditioning of a quantity is the model's.
Known and listed by `run` (HISTORY.md section 5 / 9): forward dynamics in fp32 on random trees of 15 to 40 links is up to
100 x less accurate than the reference's link-coordinate formulation (4.9e-4 against 5e-6 in the worst of 600 trees): a
first-child chain has ONE reference point (its leaf, section 4f), up to metres away from the joint axes near its head.
The gate of FD in fp32 is the general fp32 tolerance (1e-3) for that reason; the cases are printed, not hidden.
TEST INFRASTRUCTURE (uses oracle/): not part of the product."""
import os, pickle, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

TOL64 = dict(DYN=1e-7, LCF=1e-7, FD=1e-8, ID=1e-9, BIAS=1e-9, GRAV=1e-10, KIN_H=1e-10, KIN_V=1e-10, CRBA=1e-10, MINV=1e-6, JAC=1e-10, JACD=1e-10, ROLLOUT=1e-7, CONTROLLED=1e-7)
TOL32 = dict(DYN=1e-3, LCF=1e-3, FD=1e-3, ID=2e-4, BIAS=2e-4, GRAV=2e-5, KIN_H=2e-5, KIN_V=2e-5, CRBA=2e-5, MINV=3e-2, JAC=2e-5, JACD=2e-5, ROLLOUT=3e-3, CONTROLLED=3e-3)


def make_model(case):
    import helpers
    import jaxsim_amd as ja
    from jaxsim_amd import robots

    if "hub" in case["tree"]:  # [round 6] a hub with 7 .. 12 legs: more than six children on one link
        model = ja.JaxSimModel.build_from_model_description(robots.hub_urdf(**case["tree"]["hub"]))
    else:
        model = ja.JaxSimModel.build_from_model_description(robots.chain_urdf(**case["tree"]))
    if case["soft_contact"] is not None:  # (the default K = 1e6 on 0.5 kg links is at the stability limit of the step: rollouts use a softer ground)
        K, D, mu = case["soft_contact"]
        model = helpers.with_params(model, contact_params=ja.SoftContactsParams.build(K=K, D=D, mu=mu))
    if case["rk4"]:
        model = helpers.with_params(model, integrator=ja.IntegratorType.RungeKutta4)
    return model


def make_dyn_model(case):
    """[round 6] The model of the DYN / LCF quantities (js.ode.system_dynamics, js.contact.link_contact_forces): the tree of the
    case with one of the three contact models on all its points, a quarter of them on a height-field terrain."""
    import helpers
    import jaxsim_amd as ja

    model = make_model(dict(case, rk4=False))
    kind = case.get("dyn_kind", "soft")
    npts = model.kin_dyn_parameters.number_of_collidable_points()
    if kind == "relaxed" and npts:
        model = helpers.relaxed_model(model, list(range(npts)), mu=0.5)
    elif kind == "rigid" and npts:
        model = helpers.rigid_model(model, list(range(npts)), K=1e4, D=1e2, build=dict(solver_options={"solver_tol": 1e-9}))
    if case.get("dyn_hf") is not None:
        a, kx, ky, ph = case["dyn_hf"]
        model = helpers.with_params(model, terrain=ja.HeightFieldTerrain.from_function(lambda x, y: a * np.sin(kx * x + ph) * np.cos(ky * y),
                                                                                        x_range=(-2.5, 2.5), y_range=(-2.5, 2.5), spacing=0.1))
    return model


DYN_KEYS = ("base_position", "base_quaternion", "joint_positions", "base_linear_velocity", "base_angular_velocity", "joint_velocities")


def oracle_lcf(model, d, tau, f, oracle):
    if oracle.refstep.is_rigid_contact_model(model):
        from oracle import refrigid

        return refrigid.link_contact_forces(model, d, link_forces=f, joint_torques=tau)[0]
    if oracle.refstep.is_relaxed_rigid_contact_model(model):
        from oracle import refrelaxed

        return refrelaxed.link_contact_forces(model, d, link_forces=f, joint_torques=tau)[0]
    if model.kin_dyn_parameters.number_of_collidable_points() == 0:
        return np.zeros((d.batch_size, model.number_of_links(), 6))
    return oracle.refstep.link_contact_forces(model, d)[0]


def scaled(a, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))


def truths(model, d, case, oracle, refrigid, helpers):
    """Every quantity of one case from the oracle (arrays of the data's dtype in, that dtype's arithmetic)."""
    N = d.joint_positions.shape[0]
    tau, f, acc, tau_seq = case["tau"], case["f"], case["acc"], case["tau_seq"]
    c = lambda a: np.asarray(a).astype(d.joint_positions.dtype)  # noqa: E731
    out = {}
    vd, sdd = oracle.forward_dynamics_aba(model, d, joint_forces=c(tau), link_forces=c(f))
    out["FD"] = np.concatenate([vd, sdd], -1)
    fB, tq = oracle.inverse_dynamics(model, d, joint_accelerations=c(acc[:, 6:]), base_acceleration=c(acc[:, :6]), link_forces=c(f))
    out["ID"] = np.concatenate([fB if model.floating_base() else np.zeros_like(fB), tq], -1)
    out["BIAS"] = oracle.free_floating_bias_forces(model, d)
    out["GRAV"] = oracle.free_floating_gravity_forces(model, d)
    dc = d.update_caches(model)
    out["KIN_H"], out["KIN_V"] = dc.link_transforms, dc.link_velocities
    out["CRBA"] = oracle.free_floating_mass_matrix(model, d)
    J, _ = refrigid.jacobian_full_doubly_left(model, d.joint_positions)
    out["JAC"] = J
    out["JACD"] = refrigid.jacobian_derivative_full_doubly_left(model, d.joint_positions, d.joint_velocities)
    dk = d
    for _ in range(case["k"]):
        dk = oracle.step(model, dk)
    out["ROLLOUT"] = helpers.odata_to_block(model, dk)
    dk = d
    for s in range(case["k"]):
        dk = oracle.step(model, dk, joint_force_references=c(tau_seq[s]))
    out["CONTROLLED"] = helpers.odata_to_block(model, dk)
    # [round 6] system_dynamics (inertial representation whatever the data's, api/ode.py:204) and link_contact_forces
    import dataclasses

    dm = make_dyn_model(case)
    dd = dataclasses.replace(d, velocity_representation=oracle.VelRepr.Inertial)
    ref = oracle.refstep.system_dynamics(dm, dd, link_forces=c(f), joint_torques=c(tau))
    out["DYN"] = np.concatenate([np.asarray(ref[k]).reshape(N, -1) for k in DYN_KEYS], -1)
    out["LCF"] = oracle_lcf(dm, d, c(tau), c(f), oracle)
    return out


def prepare(path, seed, trials):
    import helpers, oracle
    from oracle import refrigid

    rng = np.random.default_rng(seed)
    cases, oracle_failed = [], 0
    reps = [oracle.VelRepr.Inertial, oracle.VelRepr.Body, oracle.VelRepr.Mixed]
    for trial in range(trials):
        n_links = int(rng.integers(1, 41))
        fixed = bool(rng.integers(0, 3) == 0) and n_links > 1
        ncl = int(rng.integers(0, 4))
        cl = tuple(sorted(set(int(v) for v in rng.integers(0, n_links, size=ncl))))
        tree = dict(n_links=n_links, fixed_base=fixed, seed=40000 + trial, max_back=int(rng.integers(1, 5)), collision_links=cl,
                    parallel_axes=[None, "all", "aligned", None][trial % 4])
        if trial % 10 == 7:  # [round 6] every tenth tree a hub with 7 .. 12 legs (kMaxChildren = 12)
            legs = int(rng.integers(7, 13))
            feet = int(rng.integers(0, min(3, legs) + 1))
            tree = dict(hub=dict(n_legs=legs, links_per_leg=int(rng.integers(1, 3)), foot_boxes=feet, seed=40000 + trial), seed=40000 + trial,
                        n_links=-1, fixed_base=False)
            cl = tuple(range(feet))
        rep = reps[int(rng.integers(0, 3))]
        N, k = 4, int(rng.integers(2, 7))
        base = dict(trial=trial, tree=tree, rep=rep, k=k, rk4=bool(rng.integers(0, 3) == 0), soft_contact=(2e4, 60.0, 0.6) if cl else None,
                    # (the rigid contact models refuse the world -> base offset of these fixed-base chains, DESIGN.md section 1: SoftContacts there)
                    dyn_kind=["soft", "relaxed", "rigid"][int(rng.integers(0, 3))] if not tree["fixed_base"] else ["soft"][int(rng.integers(0, 3)) * 0],
                    dyn_hf=(float(rng.uniform(0.02, 0.15)), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0, 6.28))) if (cl and rng.integers(0, 4) == 0) else None)
        for dtype in (np.float64, np.float32):
            case = dict(base, dtype=np.dtype(dtype).name)
            model = make_model(case)
            n = model.dofs()
            d = oracle.random_model_data(model, batch_size=N, seed=tree["seed"], dtype=dtype, velocity_representation=rep,
                                         base_pos_bounds=((-1, -1, 0.0), (1, 1, 0.4)), base_rpy_bounds=((-0.5, -0.5, -3), (0.5, 0.5, 3)))
            prng = np.random.default_rng(tree["seed"])
            tau, f = helpers.random_inputs(model, N, tree["seed"], dtype)
            case.update(tau=tau, f=f, acc=prng.uniform(-2, 2, size=(N, 6 + n)).astype(dtype), tau_seq=prng.uniform(-3, 3, size=(k, N, n)).astype(dtype),
                        state=helpers.odata_to_block(model, d))
            try:
                with np.errstate(all="ignore"):
                    case["truth"] = truths(model, helpers.upcast(d, model) if dtype == np.float32 else d, case, oracle, refrigid, helpers)
                    if dtype == np.float32:  # the reference's formulation in fp32: the oracle on the float32 arrays
                        r32 = truths(model, d, case, oracle, refrigid, helpers)
                        case["ref32_err"] = {q: scaled(r32[q], case["truth"][q]) for q in r32}
            except np.linalg.LinAlgError:
                oracle_failed += 1
                continue
            cases.append(case)
    with open(path, "wb") as f_:
        pickle.dump(dict(seed=seed, trials=trials, oracle_failed=oracle_failed, cases=cases), f_)
    print("prepared", len(cases), "cases; oracle failed", oracle_failed)


def run(path, out_path):
    import helpers, oracle
    import jaxsim_amd as ja
    import jaxsim_amd.api as js

    with open(path, "rb") as f_:
        blob = pickle.load(f_)
    REP = {oracle.VelRepr.Inertial: ja.VelRepr.Inertial, oracle.VelRepr.Body: ja.VelRepr.Body, oracle.VelRepr.Mixed: ja.VelRepr.Mixed}
    os.environ["JAXSIM_AMD_SPECIALIZE"] = "0"  # the library's kernels (a compiler run per random tree otherwise)
    stats, lines, nfail, widened, outliers, exploded = {}, [], 0, 0, [], 0
    for case in blob["cases"]:
        model = make_model(case)
        g = js.data.JaxSimModelData.from_state_block(model, case["state"], REP[case["rep"]])
        tau, f, acc = case["tau"], case["f"], case["acc"]
        got = {}
        vd, sdd = js.model.forward_dynamics_aba(model, g, joint_forces=tau, link_forces=f)
        got["FD"] = np.concatenate([vd, sdd], -1)
        fB, tq = js.model.inverse_dynamics(model, g, joint_accelerations=acc[:, 6:], base_acceleration=acc[:, :6], link_forces=f)
        got["ID"] = np.concatenate([fB if model.floating_base() else np.zeros_like(fB), tq], -1)
        got["BIAS"] = js.model.free_floating_bias_forces(model, g)
        got["GRAV"] = js.model.free_floating_gravity_forces(model, g)
        got["KIN_H"], got["KIN_V"] = g._link_transforms, g._link_velocities
        got["CRBA"] = js.model.free_floating_mass_matrix(model, g)
        J, Jd, _ = js.model.jacobian_full_doubly_left(model, g)
        got["JAC"], got["JACD"] = J, Jd
        got["ROLLOUT"] = js.model.rollout(model, g, case["k"]).state_block()
        got["CONTROLLED"] = js.model.rollout(model, g, case["k"], joint_force_references=case["tau_seq"]).state_block()
        dmod = make_dyn_model(case)
        gd = js.data.JaxSimModelData.from_state_block(dmod, case["state"], REP[case["rep"]])
        dyn = js.ode.system_dynamics(dmod, gd, link_forces=f, joint_torques=tau)
        got["DYN"] = np.concatenate([np.asarray(dyn[k_]).reshape(4, -1) for k_ in DYN_KEYS], -1)
        got["LCF"] = np.asarray(js.contact.link_contact_forces(dmod, gd, link_forces=f, joint_torques=tau)[0])
        errs = {q: scaled(got[q], np.asarray(case["truth"][q], dtype=np.float64)) for q in got}
        Mi = np.asarray(js.model.free_floating_mass_matrix_inverse(model, g), dtype=np.float64)
        M = np.asarray(case["truth"]["CRBA"], dtype=np.float64)
        errs["MINV"] = float(np.abs(M @ Mi - np.eye(M.shape[-1])).max())
        f32 = case["dtype"] == "float32"
        for q, e in errs.items():
            if q in ("ROLLOUT", "CONTROLLED") and not float(np.abs(np.asarray(case["truth"][q], dtype=np.float64)).max()) < 1e4:
                exploded += 1  # (a trajectory that explodes within k steps -- the explicit integrator on a tree of light links -- is no truth:
                continue       #  tools/fuzz/gpu_campaign.py drops them the same way)
            key = (q + ("/rk4" if case["rk4"] and q in ("ROLLOUT", "CONTROLLED") else "") + ("/" + case.get("dyn_kind", "soft") if q in ("DYN", "LCF") else ""), case["dtype"])
            r32 = case.get("ref32_err", {}).get(q, float("nan")) if f32 else float("nan")
            stats.setdefault(key, []).append((e, r32))
            tol = (TOL32 if f32 else TOL64)[q]
            if q in ("DYN", "LCF") and case.get("dyn_kind") == "rigid":
                tol = 3e-3 if f32 else 1e-5  # (RigidContacts: where the interior-point iteration stops; fp32: the rigid models' own gate)
            bound = max(tol, 3.0 * r32) if (f32 and np.isfinite(r32)) else tol
            widened += int(e < bound and not e < tol)
            if f32 and np.isfinite(r32) and e > 1e-4 and e > 30.0 * r32:
                outliers.append("  trial %d %s nL %d %s %s: %.2e, reference formulation in fp32 %.2e" % (case["trial"], key[0], case["tree"]["n_links"], "fixed" if case["tree"]["fixed_base"] else "floating", case["rep"], e, r32))
            if not (e < bound):
                nfail += 1
                lines.append("FAIL trial %d %s nL %d fixed %s %s: %.2e (reference formulation in fp32 %.2e)" % (case["trial"], key, case["tree"]["n_links"], case["tree"]["fixed_base"], case["rep"], e, r32))
    lines.append("query / rollout campaign seed %d, %d trees: %d cases on the device (%d the oracle could not evaluate); fails %d; fp32 quantities above their constant gate that pass by "
                 "3 x the reference formulation's own fp32 error: %d" % (blob["seed"], blob["trials"], len(blob["cases"]), blob["oracle_failed"], nfail, widened))
    lines.append("rollouts whose fp64 truth exceeds 1e4 within its k steps (exploding trajectories: excluded): %d" % exploded)
    lines.append("fp32 quantities above 1e-4 AND more than 30 x what the reference's formulation loses in fp32 (the kernel's formulation, not the model): %d" % len(outliers))
    lines.extend(outliers)
    lines.append("%-16s %-8s %6s | %-30s | %s" % ("quantity", "dtype", "cases", "device: worst median above-gate", "reference formulation in fp32: worst median"))
    for key in sorted(stats):
        a = np.array(stats[key], dtype=np.float64)
        tol = (TOL32 if key[1] == "float32" else TOL64)[key[0].split("/")[0]]
        ref = "%.2e %.2e" % (np.nanmax(a[:, 1]), np.nanmedian(a[:, 1])) if np.isfinite(a[:, 1]).any() else "-"
        lines.append("%-16s %-8s %6d | %-30s | %s" % (key[0], key[1], len(a), "%.2e %.2e %4d" % (a[:, 0].max(), np.median(a[:, 0]), int(np.sum(a[:, 0] >= tol))), ref))
    text = "\n".join(lines)
    print(text)
    if out_path:
        with open(out_path, "w") as f_:
            f_.write(text + "\n")
    return nfail


if __name__ == "__main__":
    cmd, path = sys.argv[1], sys.argv[2]
    if cmd == "prepare":
        prepare(path, int(sys.argv[3]) if len(sys.argv) > 3 else 47, int(sys.argv[4]) if len(sys.argv) > 4 else 200)
    else:
        sys.exit(1 if run(path, sys.argv[3] if len(sys.argv) > 3 else None) else 0)
