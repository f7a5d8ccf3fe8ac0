"""The build-time check of csrc/conv_pws.hip's generated code (tools/check_inflight_regs.py, run by the Makefile rule of
conv_pws.o): it must flag a copy of registers an inline-asm load is still filling, and pass code that waits first."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_inflight_regs.py")

HEAD = "_ZN12_GLOBAL__N_115conv_pws_kernelINS_6PwsCfgILi8ELi2ELi1EEELb1EEEv8ConvArgs:\n"
LOADS = """\t;;#ASMSTART
\tglobal_load_dwordx4 v[56:59], v33, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[60:63], v34, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[64:67], v35, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[68:71], v36, s[4:5]
\t;;#ASMEND
\tv_mfma_f32_32x32x16_f16 v[0:15], v[100:103], v[104:107], v[0:15]
"""
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(6)\n\t;;#ASMEND\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def run(text, tmp_path):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, TOOL, str(f)], capture_output=True, text=True)


def test_copy_in_front_of_the_wait_is_flagged(tmp_path):
    # what hipcc produced when the wait carried the registers as tied operands: a copy BEFORE the wait
    bad = HEAD + LOADS + "\tv_mov_b64_e32 v[52:53], v[56:57]\n" + WAIT + "\tv_pk_add_f32 v[94:95], v[52:53], v[88:89]\n" + TAIL
    r = run(bad, tmp_path)
    assert r.returncode == 1 and "still in flight" in r.stdout


def test_use_behind_the_wait_passes_and_later_units_stay_guarded(tmp_path):
    good = HEAD + LOADS + WAIT + "\tv_pk_add_f32 v[94:95], v[56:57], v[88:89]\n\tv_pk_add_f32 v[94:95], v[60:61], v[88:89]\n" + WAIT + \
        "\tv_pk_add_f32 v[94:95], v[64:65], v[88:89]\n" + TAIL
    r = run(good, tmp_path)
    assert r.returncode == 0, r.stdout
    # loads 2 and 3 belong to the second unit: one wait does not retire them
    early = HEAD + LOADS + WAIT + "\tv_pk_add_f32 v[94:95], v[64:65], v[88:89]\n" + WAIT + TAIL
    r = run(early, tmp_path)
    assert r.returncode == 1 and "v[64" in r.stdout.replace("[64, 65]", "v[64")


def test_a_file_without_the_kernels_is_an_error(tmp_path):
    assert run("nothing here\n", tmp_path).returncode == 1


def test_control_flow_a_linear_scan_cannot_follow_is_rejected(tmp_path):
    """The scan follows file order (documented in the tool): a backward branch while loads are in flight, and a wait that a
    forward branch could bypass, are refused; a forward branch that lands in front of the wait is fine."""
    loop = HEAD + ".LBB0_1:\n" + LOADS + "\ts_cbranch_scc1 .LBB0_1\n" + WAIT + WAIT + TAIL
    r = run(loop, tmp_path)
    assert r.returncode == 1 and "BACKWARD" in r.stdout
    bypass = HEAD + LOADS + "\ts_cbranch_vccnz .LBB0_9\n" + WAIT + ".LBB0_9:\n\tv_pk_add_f32 v[94:95], v[56:57], v[88:89]\n" + WAIT + TAIL
    r = run(bypass, tmp_path)
    assert r.returncode == 1 and "bypass" in r.stdout
    fine = HEAD + LOADS + "\ts_cbranch_vccnz .LBB0_9\n\tv_mov_b32 v1, v2\n.LBB0_9:\n" + WAIT + \
        "\tv_pk_add_f32 v[94:95], v[56:57], v[88:89]\n" + WAIT + TAIL
    assert run(fine, tmp_path).returncode == 0
