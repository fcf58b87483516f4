"""tools/check_device_asm.py: the build guard against hipcc relaxing a long branch through the return-address
register of a leaf device function.  Synthetic assembly for both verdicts, plus the real build."""
import glob
import os
import subprocess
import sys

from blsutil import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_device_asm as chk  # noqa: E402

BAD = """
	.amdhsa_kernel my_kernel
my_kernel:
	s_getpc_b64 s[30:31]
	s_setpc_b64 s[30:31]
	s_endpgm
_Z4leafv:                               ; @_Z4leafv
	v_mov_b32_e32 v0, 0
	s_cbranch_scc0 .LBB1_2
	s_getpc_b64 s[30:31]
.Lpost_getpc0:
	s_add_u32 s30, s30, (.LBB1_3-.Lpost_getpc0)&4294967295
	s_addc_u32 s31, s31, (.LBB1_3-.Lpost_getpc0)>>32
	s_setpc_b64 s[30:31]
.LBB1_2:
	v_mov_b32_e32 v1, 1
.LBB1_3:
	s_setpc_b64 s[30:31]
"""
GOOD = BAD.replace("_Z4leafv:                               ; @_Z4leafv\n",
                   "_Z4leafv:                               ; @_Z4leafv\n\tv_writelane_b32 v40, s30, 0\n\tv_writelane_b32 v40, s31, 1\n")


def test_checker_flags_leaf_function_only(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    assert chk.check(str(bad)) == ["_Z4leafv"]          # the kernel with the same pattern is not reported
    assert chk.check(str(good)) == []
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_device_asm.py"), str(bad)]) == 1
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_device_asm.py"), str(good)]) == 0


def test_current_build_is_clean():
    files = glob.glob(os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "build", "*", "*gfx950.s"))
    if not files:
        import pytest
        pytest.skip("no build directory (library built elsewhere)")
    for f in files:
        assert chk.check(f) == [], f


REV_DPP = """
	.amdhsa_kernel k2
k2:
	v_mov_b32_dpp v1, v0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_add_u32_dpp v2, v0, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_sub_u32_dpp v2, v0, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_subrev_f32_dpp v2, v0, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_subrev_u32_dpp v2, v48, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_subrev_co_u32_dpp v2, vcc, v48, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_lshlrev_b32_dpp v2, v3, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	s_endpgm
"""


def test_checker_flags_reversed_operand_integer_dpp(tmp_path):
    """gfx950 executes v_subrev_u32_dpp / v_lshlrev_b32_dpp with the lane permutation on the other operand than LLVM's DPP
    combiner assumes (tools/microbench/dpp_sub_probe.hip): any such instruction fails the build.  The
    commutative and the non-reversed forms, and the float subrev, are fine."""
    f = tmp_path / "rev.s"
    f.write_text(REV_DPP)
    assert chk.rev_dpp_uses(str(f)) == [("v_lshlrev_b32_dpp", 1), ("v_subrev_co_u32_dpp", 1), ("v_subrev_u32_dpp", 1)]
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_device_asm.py"), str(f)]) == 1
    ok = tmp_path / "ok.s"
    ok.write_text("\n".join(l for l in REV_DPP.split("\n") if "rev_u32" not in l and "rev_co" not in l and "lshlrev" not in l))
    assert chk.rev_dpp_uses(str(ok)) == []
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_device_asm.py"), str(ok)]) == 0


def test_current_build_has_no_reversed_operand_dpp_and_disables_the_combiner():
    files = glob.glob(os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "build", "*", "*gfx950.s"))
    for f in files:
        assert chk.rev_dpp_uses(f) == [], f
    mk = open(os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "Makefile")).read()
    assert "-amdgpu-dpp-combine=false" in mk


def test_design_kernel_table_matches_the_build():
    """DESIGN.md 3.4's resource table (registers, private segment, LDS, occupancy per kernel) is generated from the device assembly
    of the current build: a kernel change that is not followed by `python tools/kernel_table.py --update-design` fails here, so the
    document cannot drift from the binaries."""
    import pytest
    files = glob.glob(os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "build", "*", "*gfx950.s"))
    if not files:
        pytest.skip("no build directory (library built elsewhere)")
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), "--check-design"]) == 0
