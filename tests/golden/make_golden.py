#!/usr/bin/env python3
"""Generate the committed golden vectors (tests/golden/*.npz).

The reference cannot run here (no JAX), so these vectors come from the NumPy ORACLE: they are
regression pins of the oracle and fixed seeded cases for the kernels -- NOT reference outputs.
Inputs and expected outputs only; run from the repo root:  python tests/golden/make_golden.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import helpers  # noqa: E402
import oracle  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
CASES = ["double_pendulum", "cartpole", "box", "chain9f", "anymal", "icub"]


def main():
    zoo = helpers.ModelZoo()
    for name in CASES:
        model = zoo(name)
        N = 4
        d = zoo.random_data(name, N, seed=2024, rep=oracle.VelRepr.Inertial)
        tau, f = helpers.random_inputs(model, N, 2025, np.float64)
        acc = np.random.default_rng(2026).uniform(-2, 2, size=(N, 6 + model.dofs()))
        nxt = oracle.step(model, d, link_forces=f, joint_force_references=tau)
        vd, sdd = oracle.forward_dynamics_aba(model, d, joint_forces=tau, link_forces=f)
        fB, tid = oracle.inverse_dynamics(model, d, joint_accelerations=acc[:, 6:], base_acceleration=acc[:, :6], link_forces=f)
        np.savez_compressed(
            OUT / f"{name}.npz",
            state=helpers.odata_to_block(model, d), tau=tau, link_forces=f, acc=acc,
            step=helpers.odata_to_block(model, nxt), fd=np.concatenate([vd, sdd], -1),
            id=np.concatenate([fB, tid], -1), link_transforms=d.link_transforms, link_velocities=d.link_velocities,
        )  # fmt: skip
        print("wrote", name)


if __name__ == "__main__":
    main()
