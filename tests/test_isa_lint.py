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
