"""The oracle's C port (CPU baseline of bench.py) against the NumPy oracle."""

import numpy as np
import pytest

import helpers
import oracle
from oracle import cport


@pytest.mark.parametrize("name", ["box", "cartpole", "chain5", "chain9f", "anymal", "icub"])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-4)])
def test_cport_step_matches_numpy_oracle(models, name, dtype, tol):
    model = models(name)
    N = 6
    d = models.random_data(name, N, seed=4, dtype=dtype, rep=oracle.VelRepr.Inertial)
    tau, f = helpers.random_inputs(model, N, 5, dtype)
    ref = oracle.step(model, d, link_forces=f, joint_force_references=tau)
    out = cport.step(model, helpers.odata_to_block(model, d), tau=tau.T, link_forces_inertial=f.reshape(N, -1).T)
    assert helpers.rel_err(out, helpers.odata_to_block(model, ref)) < tol


def test_cport_threads_and_multi_step_agree(models):
    model = models("icub")
    d = models.random_data("icub", 64, seed=2)
    blk = helpers.odata_to_block(model, d)
    a = cport.step(model, blk, n_steps=10, n_threads=1)
    b = cport.step(model, blk, n_steps=10, n_threads=4)
    np.testing.assert_array_equal(a, b)
    for _ in range(10):
        d = oracle.step(model, d)
    assert helpers.rel_err(a, helpers.odata_to_block(model, d)) < 1e-9
