"""CPU tier: the generated gfx950 code of the streaming kernels never touches a register that an untracked prefetch
load (iqd_mfma.h: gload16_untracked) still owns - tools/isa_lint.py compiles the file and follows the control flow.
(The hazard is silent on a quiet machine: the bytes usually have arrived long before.  It was found by this lint.)
Round 5: and no inline assembly is the first to touch a matrix instruction's result (the compiler counts those wait
states only for instructions it knows) - the same runs check it, the last test shows the check on two listings."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_untracked_loads_of_the_wbfm_stream_kernel():
    """Round 3: the WBFM streaming kernel keeps four pieces of input in flight with the same untracked loads."""
    src = os.path.join(ROOT, "rtlsdrdiags_amd", "csrc", "iqd_stream.hip")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 finding(s)" in last and " 25 kernels" in last, last
    # vector registers spilled to scratch: only in the instantiations with the gain-epoch lookup (third template flag),
    # which run for the few calls after a gain change - never in the kernels of the steady state
    spilled = [ln.split()[1].rstrip(":") for ln in r.stdout.splitlines() if ln.startswith("scratch: ")]
    import re
    flags = [re.search(r"wbfm_stream_kernelILi?n?\d+ELb([01])ELb([01])ELb([01])EEE", k).groups() for k in spilled]   # MAG, EPOCHS, GATED
    assert all(f[1] == "1" for f in flags), spilled
    assert int(last.split(" global loads")[0].split()[-1]) > 100


def test_untracked_loads_are_not_touched_before_they_arrive():
    src = os.path.join(ROOT, "rtlsdrdiags_amd", "csrc", "iqd_stream2.hip")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 finding(s)" in last and " 9 kernels" in last and "0 kernel(s) with scratch" in last, last
    n_loads = int(last.split(" global loads")[0].split()[-1])
    assert n_loads > 100          # the lint saw the loads it is about


def test_untracked_loads_of_the_mixed_launch():
    """The launch that runs several families' pipelines side by side compiles the same workgroup bodies a second time."""
    src = os.path.join(ROOT, "rtlsdrdiags_amd", "csrc", "iqd_stream_mixed.hip")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 finding(s)" in last and " 3 kernels" in last and "0 kernel(s) with scratch" in last, last
    assert int(last.split(" global loads")[0].split()[-1]) > 100


def test_lint_flags_inline_assembly_as_first_reader_of_a_matrix_result():
    """Round 5: the compiler counts the wait states between a matrix instruction and a VALU read of its result only for
    instructions it knows; an inline-assembly first reader gets none (seen on the device: wrong table rows).  The lint's
    second check on two synthetic listings."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    bad = """
	v_mfma_i32_16x16x64_i8 v[54:57], v[30:33], v[46:49], v[74:77]
	;;#ASMSTART
	v_msad_u8 v60, v54, s77, 0
	;;#ASMEND
	s_nop 2
""".splitlines()
    good = """
	v_mfma_i32_16x16x64_i8 v[54:57], v[30:33], v[46:49], v[74:77]
	s_nop 6
	v_lshl_add_u32 v50, v57, 8, v54
	v_lshl_add_u32 v51, v56, 8, v55
	;;#ASMSTART
	v_msad_u8 v60, v50, s77, 0
	;;#ASMEND
	v_mfma_i32_16x16x64_i8 v[54:57], v[30:33], v[46:49], v[54:57]
""".splitlines()
    assert len(isa_lint.lint_mfma_readers("k", list(enumerate(bad, 1)))) == 1
    assert isa_lint.lint_mfma_readers("k", list(enumerate(good, 1))) == []


def test_lint_flags_a_vector_read_directly_behind_an_inline_sdwa_partial_write():
    """Round 6: gfx940 / gfx950 want one wait state between a VALU write with a destination select and a VALU read
    of that register; the compiler inserts it behind its own SDWA instructions, not behind inline assembly
    (iqd_prims.h: cast_pack_i16_bounded - two partial writes of one register back to back would be exactly that).  The
    lint's third check on three synthetic listings; the three files' own runs above hold the shipped kernels to it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    back_to_back = """
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v7, v3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD
	;;#ASMEND
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v7, v4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD
	;;#ASMEND
	s_nop 0
	v_dot2c_i32_i16_e32 v9, v7, v8
""".splitlines()
    read_next = """
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v7, v4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD
	;;#ASMEND
	v_dot2c_i32_i16_e32 v9, v7, v8
""".splitlines()
    spaced = """
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v7, v3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD
	;;#ASMEND
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v6, v5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD
	;;#ASMEND
	;;#ASMSTART
	v_cvt_i32_f32_sdwa v7, v4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD
	;;#ASMEND
	v_mul_f32_e32 v10, v11, v12
	v_dot2c_i32_i16_e32 v9, v7, v8
	v_add_u32_sdwa v3, v3, v3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1
	v_add_u32_e32 v65, v3, v65
""".splitlines()
    assert len(isa_lint.lint_sdwa_forwarding("k", list(enumerate(back_to_back, 1)))) == 1
    assert len(isa_lint.lint_sdwa_forwarding("k", list(enumerate(read_next, 1)))) == 1
    assert isa_lint.lint_sdwa_forwarding("k", list(enumerate(spaced, 1))) == []
