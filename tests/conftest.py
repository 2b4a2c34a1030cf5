import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The product's default is to compile a model-specialised kernel on first use when hipcc is present
# (jaxsim_amd/specialize.py::policy -- the jax.jit experience).  The suite pins 'cached': the zoo models whose objects
# __graft_entry__.build() pre-built run specialised, every other model runs the library's own kernels, no test waits
# ~20 s per new model for a compiler, and a kernel never changes between two calls a test compares bitwise.
# tests/test_specialize.py covers build-on-first-use explicitly.
os.environ.setdefault("JAXSIM_AMD_SPECIALIZE", "cached")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the developer knob of jaxsim_amd/_lib.py would let any library stand in for the product
    if os.environ.get("JAXSIM_AMD_LIB"):
        raise pytest.UsageError("unset JAXSIM_AMD_LIB: the tests check the in-tree libjaxsim_amd.so only")


def _device_count() -> int:
    try:
        from jaxsim_amd import runtime

        return runtime.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a machine without a HIP device or without the built
    library, so a plain `pytest` on a CPU box is green; on the GPU box nothing is skipped."""
    if not any("gpu" in it.keywords for it in items):
        return
    if (config.getoption("markexpr", "") or "").strip() == "gpu":
        return  # explicitly selected (the GPU box): a missing device or library must FAIL, not skip
    if _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device / libjaxsim_amd.so not built: GPU parity tests need an MI355X")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def models():
    import helpers

    return helpers.ModelZoo()
