"""The build guard of tools/check_spill_exec.py: VGPR spill code ahead of the instruction that re-activates lanes in a block
(hipcc, ROCm 7.2, gfx950: found as the cause of the round-1 "UKF chol(P) codegen hazard")."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

BAD = """
_Z1kv:                                  ; @_Z1kv
	s_mov_b64 s[0:1], exec
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
	v_add_u32_e32 v3, 0x300, v3
	v_cmp_le_i32_e32 vcc, s16, v3
	s_or_b64 s[4:5], vcc, s[4:5]
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execnz .LBB0_1
.LBB0_2:
	s_movk_i32 s17, 0xff80
	s_waitcnt vmcnt(0)
	scratch_store_dwordx2 off, v[68:69], off offset:88 ; 8-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
"""
GOOD = BAD.replace("	scratch_store_dwordx2 off, v[68:69], off offset:88 ; 8-byte Folded Spill\n	s_or_b64 exec, exec, s[0:1]\n",
                   "	s_or_b64 exec, exec, s[0:1]\n	scratch_store_dwordx2 off, v[68:69], off offset:88 ; 8-byte Folded Spill\n")


def test_detector_on_the_pattern_found(tmp_path):
    import check_spill_exec as g

    for name, text, want in (("bad.s", BAD, 1), ("good.s", GOOD, 0)):
        p = tmp_path / name
        p.write_text(text)
        f = g.scan(str(p))
        assert len(f) == want, f
    assert "EXEC == 0" in g.scan(str(tmp_path / "bad.s"))[0][2]


def test_product_assembly_is_clean():
    """the device assembly of the product library, compiled with the product flags (hipcc cross-compiles without a GPU)"""
    csrc = os.path.join(ROOT, "awesomeslam_amd", "csrc")
    r = subprocess.run(["make", "-s", "-C", csrc, "check-spills"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert b"check_spill_exec: clean" in r.stdout and b"check_agpr_strip: clean" in r.stdout and b"check_vmcnt_protocol: clean" in r.stdout


STRIP = """
_ZN5aslam15large_trsm_pipeILi17ELi0EEEv:  ; @_ZN5aslam15large_trsm_pipeILi17ELi0EEEv
	s_load_dword s0, s[4:5], 0x0
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_BEGIN
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_16x16x4_f32 v[0:3], v4, a12, v[0:3]
	;;#ASMEND
%s
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_END
	;;#ASMEND
	s_endpgm
_ZN5aslam19large_chol_residentILi17ELi0EEEv:  ; @_ZN5aslam19large_chol_residentILi17ELi0EEEv
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_BEGIN
	;;#ASMEND
	v_add_f32_e32 v1, v2, v3
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_END
	;;#ASMEND
	v_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[6:7], 0
	s_endpgm
_ZN5aslam15large_chol_bf16ILi17EEEv:  ; @_ZN5aslam15large_chol_bf16ILi17EEEv
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=1 buffer_store_dwordx4=2
	;;#ASMEND
.LBB2_1:
	global_store_dwordx4 v[0:1], v[2:5], off
	buffer_store_dwordx4 v[2:5], v0, s[0:3], 0 offen
	buffer_store_dwordx4 v[2:5], v0, s[0:3], 0 offen
	s_cbranch_scc1 .LBB2_1
%s
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_END
	;;#ASMEND
.LBB2_2:
	s_endpgm
_ZN5aslam15large_trsm_bf16ILi17ELi0EEEv:  ; @_ZN5aslam15large_trsm_bf16ILi17ELi0EEEv
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=4
	;;#ASMEND
	global_store_dwordx4 v[0:1], v[2:5], off
	global_store_dwordx4 v[0:1], v[2:5], off
	global_store_dwordx4 v[0:1], v[2:5], off
	global_store_dwordx4 v[0:1], v[2:5], off
	;;#ASMSTART
	buffer_load_dwordx4 v0, s[0:3], 0 offen lds
	s_waitcnt vmcnt(12)
	;;#ASMEND
	;;#ASMSTART
	; ASLAM_STRIP_LIVE_END
	;;#ASMEND
	s_endpgm
"""


def test_agpr_strip_detector(tmp_path):
    """compiler-generated AGPR use while the hand-allocated strip of the large-state kernels is live: flagged inside
    the markers, accepted outside them (the fp64 MFMAs of the diagonal-block factorisation run between two sweeps)"""
    import check_agpr_strip as g

    for name, filler, want in (("ok.s", "\tv_add_f32_e32 v1, v2, v3", 0), ("spill.s", "\tv_accvgpr_write_b32 a3, v5 ;  Reload Reuse", 1),
                               ("mfma.s", "\tv_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[6:7], 0", 1)):
        p = tmp_path / name
        p.write_text(STRIP % (filler, ""))
        f = g.scan(str(p))
        assert len(f) == want, (name, f)
    # a branch that leaves a live range (a cold block laid out behind the END marker) makes the linear-order reading unsound
    p = tmp_path / "cold.s"
    p.write_text(STRIP % ("\tv_add_f32_e32 v1, v2, v3", "\ts_cbranch_scc0 .LBB2_2"))
    f = g.scan(str(p))
    assert len(f) == 1 and "out of" in f[0][2], f
    # a strip kernel without markers: the guard would be vacuous
    p = tmp_path / "nomarkers.s"
    p.write_text((STRIP % ("\tv_add_f32_e32 v1, v2, v3", "")).replace("; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=4", "; nothing").replace(
        "\t; ASLAM_STRIP_LIVE_END\n\t;;#ASMEND\n\ts_endpgm\n\"\"\"", ""))
    assert any("vacuous" in x[2] for x in g.scan(str(p)))


def test_vmcnt_protocol_detector(tmp_path):
    """the compiler-generated vector-memory instructions inside the hand-counted sweeps must be exactly the stores the s_waitcnt vmcnt counts
    assume: one extra store, a scratch reload or a compiler-generated vmcnt wait fails the build"""
    import check_vmcnt_protocol as g

    ok = tmp_path / "ok.s"
    ok.write_text(STRIP % ("\tv_add_f32_e32 v1, v2, v3", ""))
    f, checked = g.scan(str(ok))
    assert not f and len(checked) == 2, (f, checked)
    for name, extra, frag in (("store.s", "\tglobal_store_dwordx4 v[0:1], v[2:5], off", "differ"),
                              ("reload.s", "\tscratch_load_dword v7, off, off offset:16 ; 4-byte Folded Reload", "unexpected"),
                              ("wait.s", "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "s_waitcnt vmcnt")):
        p = tmp_path / name
        p.write_text(STRIP % ("\tv_add_f32_e32 v1, v2, v3", extra))
        f, _ = g.scan(str(p))
        assert f and any(frag in x[2] for x in f), (name, f)


def test_dpp_hazard_detector(tmp_path):
    """A DPP instruction must not read a VGPR that a VALU instruction wrote less than two wait states earlier (gfx90a+): the DPP fmacs of the
    diagonal-tile factorisations are inline assembly, which hipcc's hazard recogniser does not look into (round 4)."""
    import check_dpp_hazard as g

    dpp = "\tv_fmac_f64_dpp v[8:9], -v[92:93], v[92:93] row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_mul_f64 v[92:93], v[14:15], s[70:71]\n\ts_nop 1\n" + dpp + dpp + "\tv_mov_b32_e32 v7, v1\n\tv_add_f32_e32 v2, v3, v4\n" + dpp)
    assert g.check(str(ok)) == []
    for name, body in (("one_wait_state.s", "k:\n\tv_mul_f64 v[92:93], v[14:15], s[70:71]\n\ts_nop 0\n" + dpp),
                       ("copy_in_front.s", "k:\n\tv_mul_f64 v[92:93], v[14:15], s[70:71]\n\ts_nop 1\n" + dpp + "\tv_mov_b32_e32 v93, v1\n" + dpp),
                       ("half_of_the_pair.s", "k:\n\tv_mov_b32_e32 v92, v1\n\tv_add_f32_e32 v2, v3, v4\n" + dpp)):
        p = tmp_path / name
        p.write_text(body)
        assert len(g.check(str(p))) == 1, name
