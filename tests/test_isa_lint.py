"""jaxsim_amd/isa_lint.py: the wait-state / join-block lint that csrc/build.sh and specialize.compile run over every
code object.  The rules are pinned on hand-written gfx950 snippets (assembled here, no GPU) whose hardware behaviour
tools/ubench/exec_dpp.hip measured (profiles/r04_exec_dpp_ubench.txt), and the shipped library must pass."""
import pathlib
import shutil
import subprocess

import pytest

from jaxsim_amd import isa_lint

CLANG = "/opt/rocm/lib/llvm/bin/clang"
pytestmark = pytest.mark.skipif(not (pathlib.Path(CLANG).exists() and pathlib.Path(isa_lint.OBJDUMP).exists()), reason="ROCm LLVM tools not installed")

HEAD = ".text\n.globl k\n.p2align 8\n.type k,@function\nk:\n"
DPP = "v_mov_b32_dpp v2, v1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"


def lint(tmp_path, body):
    src = tmp_path / "k.s"
    src.write_text(HEAD + body + "s_endpgm\n")
    obj = tmp_path / "k.o"
    subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)], check=True)
    _counts, hits = isa_lint.lint_file(str(obj))
    return sorted(h["rule"] for _n, h in hits)


@pytest.mark.parametrize("nops,want", [("", ["valu_vgpr"]), ("s_nop 0\n", ["valu_vgpr"]), ("s_nop 1\n", []), ("v_mov_b32 v9, v8\nv_mov_b32 v10, v8\n", [])])
def test_valu_write_of_the_permuted_operand_needs_two_wait_states(tmp_path, nops, want):
    assert lint(tmp_path, "v_mov_b32 v1, v0\n" + nops + DPP) == want


def test_plain_operands_of_a_dpp_instruction_are_not_flagged(tmp_path):
    # src1 and the accumulator are forwarded like those of any VALU instruction (ubench T10 / T11)
    assert lint(tmp_path, "v_mov_b32 v3, v0\nv_fmac_f32_dpp v2, v1, v3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") == []
    assert lint(tmp_path, "v_mov_b32 v2, v0\nv_fmac_f32_dpp v2, v1, v3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") == []


@pytest.mark.parametrize("nops,want", [("", ["valu_exec"]), ("s_nop 3\n", ["valu_exec"]), ("s_nop 4\n", [])])
def test_valu_write_of_exec_needs_five_wait_states(tmp_path, nops, want):
    assert lint(tmp_path, "v_cmpx_ne_u32_e32 vcc, 0, v5\n" + nops + DPP) == want


def test_scalar_write_of_exec_is_interlocked(tmp_path):
    """Round 3's suspect: measured clean at 0 .. 6 wait states (profiles/r04_exec_dpp_ubench.txt), so not a rule --
    but still countable with the threshold the round-3 review used."""
    body = "s_or_b64 exec, exec, s[0:1]\n" + DPP
    assert lint(tmp_path, body) == []
    assert [h["rule"] for h in isa_lint.lint_kernel(next(iter(isa_lint.parse(isa_lint.disassemble(isa_lint.code_objects(str(tmp_path / "k.o"))[0])).values())),
                                                    dict(isa_lint.RULES, salu_exec=5))] == ["salu_exec"]  # fmt: skip


def test_hazard_through_a_branch_is_found_on_every_path(tmp_path):
    # the producer sits in front of a conditional branch, the DPP at its target
    body = "v_mov_b32 v1, v0\ns_cbranch_scc1 L1\ns_nop 7\nL1:\n" + DPP
    assert lint(tmp_path, body) == ["valu_vgpr"]


def test_vector_work_in_front_of_the_exec_restore_of_a_join_block(tmp_path):
    """The compiler bug behind round 3's non-deterministic kernel: live-range-split copies placed at the head of the
    join block of an `if`, in front of `s_or_b64 exec` (they run for the lanes of the body only, or for none)."""
    good = "s_and_saveexec_b64 s[0:1], vcc\ns_cbranch_execz L1\nv_mov_b32 v3, v4\nL1:\ns_mov_b32 s14, s8\ns_or_b64 exec, exec, s[0:1]\nv_mov_b32 v5, v6\n"
    bad = "s_and_saveexec_b64 s[0:1], vcc\ns_cbranch_execz L1\nv_mov_b32 v3, v4\nL1:\nv_accvgpr_write_b32 a3, v7\ns_mov_b32 s14, s8\ns_or_b64 exec, exec, s[0:1]\n"
    orelse = "s_and_saveexec_b64 s[0:1], vcc\ns_xor_b64 s[0:1], exec, s[0:1]\ns_cbranch_execz L1\nv_mov_b32 v3, v4\nL1:\ns_andn2_saveexec_b64 s[0:1], s[0:1]\nv_mov_b32 v3, v5\ns_or_b64 exec, exec, s[0:1]\n"
    assert lint(tmp_path, good) == []
    assert lint(tmp_path, bad) == ["masked_join"]
    assert lint(tmp_path, orelse) == []  # (the else side of an if / else is a masked body, not a join)


def test_the_shipped_library_passes(tmp_path):
    from jaxsim_amd import _lib

    if not pathlib.Path(_lib.LIB_PATH).exists():
        pytest.skip("library not built")
    counts, hits = isa_lint.lint_file(str(_lib.LIB_PATH))
    assert len(counts) >= 150 and sum(counts.values()) > 50000  # (the lint saw the kernels and their DPP instructions)
    assert hits == []


def test_check_raises_and_names_the_site(tmp_path):
    src = tmp_path / "k.s"
    src.write_text(HEAD + "v_mov_b32 v1, v0\n" + DPP + "s_endpgm\n")
    obj = tmp_path / "k.o"
    subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)], check=True)
    with pytest.raises(RuntimeError, match="valu_vgpr.*v_mov_b32_dpp"):
        isa_lint.check(str(obj))
