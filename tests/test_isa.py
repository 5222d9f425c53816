"""Static check of the built ISA (no GPU): tools/check_asm_mfma_reads.py on the files that hold asm MFMAs, and the checker
itself on two small texts (the round-3 hoisted-conversion bug and its fix)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_mfma_reads as chk  # noqa: E402

BAD = """
_Z6kernelv:
	;;#ASMSTART
	v_mfma_f32_32x32x16_f16 v[2:17], v[18:21], v[22:25], v[2:17]
	;;#ASMEND
	v_cvt_pk_f16_f32 v30, v2, v3
	;;#ASMSTART
	s_nop 15
	s_nop 7
	;;#ASMEND
	v_cvt_pk_f16_f32 v31, v4, v5
	s_endpgm
"""
GOOD = BAD.replace("	v_cvt_pk_f16_f32 v30, v2, v3\n", "").replace("v31, v4, v5", "v31, v2, v3")


def test_checker_flags_a_read_hoisted_above_the_wait_states(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    found = chk.check(str(bad))
    assert len(found) == 1 and "reads v2 0 wait states" in found[0]
    assert chk.check(str(good)) == []


def test_checker_ignores_reads_inside_asm_and_compiler_mfmas(tmp_path):
    text = BAD.replace("	v_cvt_pk_f16_f32 v30, v2, v3\n", "	;;#ASMSTART\n	v_cvt_pk_f16_f32 v30, v2, v3\n	;;#ASMEND\n")
    f = tmp_path / "a.s"
    f.write_text(text)
    assert chk.check(str(f)) == []
    g = tmp_path / "b.s"
    g.write_text(BAD.replace("	;;#ASMSTART\n	v_mfma", "	v_mfma", 1).replace("v[2:17]\n	;;#ASMEND", "v[2:17]", 1))
    assert chk.check(str(g)) == []          # hipcc pads the MFMAs it emits itself


AGPR_BAD = """
_Z7kernel2v:
	;;#ASMSTART
	v_mfma_f32_32x32x16_f16 a[0:15], v[18:21], v[22:25], a[0:15]
	;;#ASMEND
	v_accvgpr_read_b32 v40, a3
	s_nop 15
	s_nop 7
	v_accvgpr_read_b32 v41, a4
	v_add_f32_e32 v0, v0, v1
	s_endpgm
"""


def test_checker_tracks_accumulators_in_agprs(tmp_path):
    """the weight-gradient accumulators of the fused backward kernels live in AGPRs (asm "+a"); a compiler-generated
    v_accvgpr_read inside the MFMA's wait states is the same hazard as a VGPR read.  a3 is read at once (flagged), a4 behind the
    wait states (fine); v0 / v1 are a different register file than a0 / a1 (not flagged)."""
    f = tmp_path / "agpr.s"
    f.write_text(AGPR_BAD)
    found = chk.check(str(f))
    assert len(found) == 1 and "reads a3 0 wait states" in found[0], found
    g = tmp_path / "agpr_ok.s"
    g.write_text(AGPR_BAD.replace("	v_accvgpr_read_b32 v40, a3\n", ""))
    assert chk.check(str(g)) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not present")
def test_built_isa_keeps_compiler_reads_behind_asm_mfma_wait_states():
    out = subprocess.run(["make", "-C", ROOT, "-s", "-j4", "check-isa"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


SGPR_BAD = """
_Z6kernelv:
	v_readlane_b32 s4, v248, 15
	v_readlane_b32 s5, v248, 16
	;;#ASMSTART
	global_load_dwordx4 a[0:3], v204, s[4:5]
	;;#ASMEND
	s_endpgm
"""


def test_sgpr_checker_flags_an_asm_load_right_behind_a_restored_spill(tmp_path):
    """Round 4: an asm vector-memory instruction reading an SGPR pair that v_readlane restored less than 5 wait states earlier
    (hipcc pads only its own instructions) -- the fault of the first register-staged weight loads; `s_nop 4` in the statement."""
    import check_asm_sgpr_hazard as sg
    f = tmp_path / "bad.s"
    f.write_text(SGPR_BAD)
    found = sg.check(str(f))
    assert len(found) == 1 and "reads s4 1 wait states" in found[0], found
    g = tmp_path / "good.s"
    g.write_text(SGPR_BAD.replace("	global_load", "	s_nop 4\n	global_load"))
    assert sg.check(str(g)) == []
    h = tmp_path / "salu.s"          # an SALU-written pair needs no padding
    h.write_text(SGPR_BAD.replace("v_readlane_b32 s4, v248, 15", "s_mov_b32 s4, s8").replace("v_readlane_b32 s5, v248, 16", "s_mov_b32 s5, s9"))
    assert sg.check(str(h)) == []
