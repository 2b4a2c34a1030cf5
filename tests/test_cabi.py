"""C-ABI library: builds for gfx950, loads on a CPU-only machine, exports every symbol
``include/jaxsim_amd.h`` declares, and fails loudly (no CPU fallback) without a GPU."""

import ctypes as C
import pathlib
import re

import numpy as np
import pytest

import jaxsim_amd as ja
from jaxsim_amd import _lib, runtime

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    if not _lib.LIB_PATH.exists():
        import __graft_entry__

        __graft_entry__.build()
    return _lib.load()


def declared_symbols():
    text = (ROOT / "include" / "jaxsim_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jxs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ("jxs_model_create", "jxs_step", "jxs_rollout", "jxs_rollout_controlled", "jxs_rollout_recorded", "jxs_forward_dynamics_aba", "jxs_inverse_dynamics",
                 "jxs_refresh_kinematics", "jxs_allgather", "jxs_comm_init", "jxs_comm_version", "jxs_device_pci_bus_id", "jxs_last_error"):  # fmt: skip
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/jaxsim_amd.h but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == set(declared_symbols())


def test_struct_layout_matches_header():
    text = (ROOT / "include" / "jaxsim_amd.h").read_text()
    body = re.search(r"typedef struct jxs_model_desc \{(.*?)\} jxs_model_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        # "double K, D, mu, p, q" style declarations
        first = names.index(next(n for n in names if n not in ("const", "int32_t", "double", "uint8_t")))
        fields += [n.strip(",").split("[")[0] for n in names[first:]]
    assert fields == [f[0] for f in _lib.ModelDesc._fields_]


@pytest.mark.skipif(runtime.device_count() > 0, reason="CPU-only behaviour")
def test_no_gpu_means_loud_failure(lib):
    from jaxsim_amd import robots

    model = ja.JaxSimModel.build_from_model_description(robots.cartpole_urdf())
    desc, keep = _lib.make_desc(model, np.float32)
    h = C.c_void_p()
    rc = lib.jxs_model_create(C.byref(desc), C.byref(h))
    assert rc == -2 and b"no HIP device" in lib.jxs_last_error()
    import jaxsim_amd.api as js

    with pytest.raises(_lib.JaxsimAmdError):
        js.data.JaxSimModelData.build(model)


def test_invalid_arguments_return_codes(lib):
    assert lib.jxs_step(None, None, None, None, None, 0, 1, None) == -1
    assert b"null" in lib.jxs_last_error()
    assert lib.jxs_model_create(None, None) == -1
