"""The build guard against the toolchain's misplaced live-range copies (crispr-bean_amd/isa_check.py): the
detector finds the pattern in a minimal hand-assembled code object, passes its clean twin, and the three
libraries of this checkout have no finding.  No GPU needed (hipcc cross-assembles)."""
import os
import subprocess

import pytest

import bean_amd  # noqa: F401
from bean_amd import _lib, isa_check

CLANG = "/opt/rocm/lib/llvm/bin/clang"

# the join of round 3's fault, reduced: a copy of a value that is live across the `if` (v3 -> v5) stands
# at the branch target, in front of the s_or_b64 that brings the other lanes back
BAD = """
    .text
    .globl k
    .p2align 8
    .type k,@function
k:
    v_cmp_gt_i32_e32 vcc, 5, v0
    s_and_saveexec_b64 s[0:1], vcc
    s_cbranch_execz .Ljoin
    v_add_u32_e32 v1, 1, v1
.Ljoin:
    {pre}
    s_or_b64 exec, exec, s[0:1]
    {post}
    v_mov_b32_e32 v3, 0
    v_mov_b32_e32 v3, v5
    s_endpgm
"""


def _assemble(tmp_path, name, pre, post):
    src = tmp_path / f"{name}.s"
    obj = tmp_path / f"{name}.o"
    src.write_text(BAD.format(pre=pre, post=post))
    subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                    "-o", str(obj)], check=True)
    return str(obj)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not installed")
def test_detector_on_a_minimal_code_object(tmp_path):
    bad = _assemble(tmp_path, "bad", "v_mov_b32_e32 v5, v3", "")
    good = _assemble(tmp_path, "good", "", "v_mov_b32_e32 v5, v3")
    lanes = _assemble(tmp_path, "lanes", "v_writelane_b32 v7, s4, 3", "v_mov_b32_e32 v5, v3")
    found = isa_check.findings_of(bad)
    assert len(found) == 1 and found[0][0] == "k" and found[0][2] == "copy"
    assert "v_mov_b32_e32 v5, v3" in found[0][3][0][1]
    assert isa_check.findings_of(good) == []
    assert isa_check.findings_of(lanes) == []  # v_writelane ignores exec: an SGPR spill may stand there
    assert isa_check.main([bad]) == 1 and isa_check.main([good]) == 0


@pytest.mark.parametrize("amax", _lib.ALL_BUILDS)
def test_built_libraries_have_no_finding(amax):
    path = _lib.build_library(amax=amax)  # (fails by itself on a finding when it has to rebuild)
    assert isa_check.findings_of(path) == []
