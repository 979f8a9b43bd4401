"""Inline-assembly hazard lint (tools/isa_lint.py) over the kernels that use asm vector instructions next to MFMAs.

The compiler pads matrix-pipe hazards only for instructions it knows; an asm `v_pk_add_f32` whose result an MFMA reads one
instruction later gives wrong results that depend on register allocation (seen once in round 2).  The lint compiles the
sources to gfx950 assembly (no GPU needed) and fails on any such site, so a scheduling change cannot bring it back unnoticed."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

SNIPPET = """
kern:
	v_mfma_f32_16x16x4_f32 v[170:173], v27, v79, v[170:173]
	;;#ASMSTART
	v_pk_add_f32 v[188:189], v[74:75], v[90:91] neg_lo:[0,1] neg_hi:[0,1]
	;;#ASMEND
	%s
	v_mfma_f32_16x16x4_f32 v[74:77], v12, v188, v[170:173]
.Lfunc_end0:
"""


def test_lint_flags_an_asm_result_consumed_by_the_next_mfma():
    bad = [f for f in isa_lint.lint_text(SNIPPET % "v_mov_b32_e32 v1, v2") if f[2] == "A"]
    assert len(bad) == 1 and "v[188:189]" in bad[0][3]
    assert not [f for f in isa_lint.lint_text(SNIPPET % "s_nop 1") if f[2] == "A"]           # two wait states: fine
    assert not [f for f in isa_lint.lint_text(SNIPPET.replace("v12, v188", "v12, v190") % "") if f[2] == "A"]


LOOP = """
kern:
.LBB0_1:
	v_mfma_f32_16x16x4_f32 v[74:77], v12, v188, v[170:173]
	v_mov_b32_e32 v1, v2
	v_mov_b32_e32 v3, v4
	;;#ASMSTART
	v_pk_add_f32 v[188:189], v[74:75], v[90:91]
	;;#ASMEND
	%s
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_lint_follows_loop_back_edges():
    """The asm result is consumed by the MFMA at the TOP of the loop, one branch later: invisible in textual order."""
    bad = [f for f in isa_lint.lint_text(LOOP % "") if f[2] == "A"]
    assert len(bad) == 1 and "v[188:189]" in bad[0][3]
    assert not [f for f in isa_lint.lint_text(LOOP % "s_nop 0") if f[2] == "A"]      # s_nop 0 + the branch: two wait states


@pytest.mark.parametrize("name", ["conv_mfma.hip", "conv_wino8.hip", "conv_wgrad_wino.hip"])
def test_no_asm_to_mfma_hazard_in_the_kernels(name):
    src = glob.glob(os.path.join(ROOT, "*_amd", "csrc", name))[0]
    text = isa_lint.compile_to_asm(src)
    funcs = isa_lint.parse(text)
    n_asm = sum(1 for fn in funcs.values() for i in fn if i["asm"] and i["op"].startswith("v_"))
    assert n_asm > 100, "the lint did not see the inline-asm instructions (assembly format changed?)"
    errs = [f for f in isa_lint.lint_text(text) if f[2] == "A"]
    assert not errs, errs[:5]
