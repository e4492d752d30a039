#!/usr/bin/env python3
"""Lint for the streaming kernels' untracked prefetch loads (rtlsdrdiags_amd/csrc/iqd_mfma.h: gload16_untracked).

Those loads are inline assembly the compiler does not track: nothing may read or overwrite their destination
registers between the load and the matching `s_waitcnt vmcnt(N) ; arrived v[..]` - a register move there (a loop
hand-over the allocator decided on) would copy bytes that have not arrived yet.  This compiles a .hip file to gfx950
assembly and runs a forward data flow over every kernel's control-flow graph (which registers may still be in flight):

    python3 tools/isa_lint.py rtlsdrdiags_amd/csrc/iqd_stream2.hip

Exit status 1 and a listing if a destination register of a pending untracked load is touched before its arrival.

Second check (round 5): the first reader of a matrix instruction's result must not be inline assembly (lint_mfma_readers).
Third check (round 6): no vector instruction reads a register in the instruction right behind an inline-assembly
SDWA write of a part of it (lint_sdwa_forwarding)."""
import re
import subprocess
import sys
import tempfile
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_to_asm(src):
    fd, path = tempfile.mkstemp(suffix=".s")
    os.close(fd)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing",
           "-w", "-I" + os.path.join(ROOT, "rtlsdrdiags_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", path, src] + os.environ.get("ISA_LINT_FLAGS", "").split()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(path).read()
    os.unlink(path)
    return text


def parse(lines):
    """-> list of (line_no, text, kind, regs, target): kind in load / arrived / settle / end / branch / cbranch / other"""
    out, labels, in_asm = [], {}, False
    for no, raw in lines:
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = len(out)
            continue
        if not line or line.startswith(";") or line.startswith("."):
            continue
        code, _, comment = line.partition(";")
        code = code.strip()
        op = code.split()[0] if code else ""
        if in_asm and op.startswith("global_load_dword"):
            out.append((no, line, "load", regs_of(code.split(",")[0]), None))
        elif in_asm and op == "s_waitcnt" and "arrived" in comment:
            out.append((no, line, "arrived", regs_of(comment), None))
        elif op == "s_waitcnt" and "vmcnt(0)" in code:
            out.append((no, line, "settle", set(), None))
        elif op == "s_endpgm":
            out.append((no, line, "end", set(), None))
        elif op == "s_branch":
            out.append((no, line, "branch", set(), code.split()[1]))
        elif op.startswith("s_cbranch"):
            out.append((no, line, "cbranch", set(), code.split()[1]))
        else:
            out.append((no, line, "other", regs_of(code), None))
    return out, labels


def lint_kernel(name, lines):
    """Forward data flow over the kernel's control-flow graph: which registers may belong to an untracked load that has
    not been waited for; any other instruction naming such a register is a finding."""
    ins, labels = parse(lines)
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (no, text, kind, regs, target) in enumerate(ins):
        if kind == "end":
            continue
        if kind in ("branch", "cbranch") and target in labels and labels[target] < n:
            succ[i].append(labels[target])
        if kind != "branch" and i + 1 < n:
            succ[i].append(i + 1)
    pend_in = [dict() for _ in range(n)]            # register -> line of the owning load
    work = [0] if n else []
    seen_once = [False] * n
    while work:
        i = work.pop()
        no, text, kind, regs, target = ins[i]
        cur = dict(pend_in[i])
        if kind == "load":
            for r in regs:
                cur[r] = no
        elif kind == "arrived":
            for r in regs:
                cur.pop(r, None)
        elif kind in ("settle", "end"):
            cur = {}
        for j in succ[i]:
            merged = dict(pend_in[j])
            changed = not seen_once[j]
            for r, owner in cur.items():
                if r not in merged:
                    merged[r] = owner
                    changed = True
            if changed:
                pend_in[j] = merged
                seen_once[j] = True
                work.append(j)
    findings = []
    for i, (no, text, kind, regs, target) in enumerate(ins):
        if kind != "other" and kind != "load":
            continue
        if kind == "other" and pend_in[i] and text.split()[0] in ("s_swappc_b64", "s_setpc_b64", "s_call_b64"):
            # a function call while loads are in flight: the callee knows nothing of the registers they will land in
            findings.append((name, no, text, sorted(pend_in[i]), sorted(set(pend_in[i].values()))))
            continue
        touched = regs & set(pend_in[i]) if kind == "other" else set()
        if kind == "load":      # the address registers of a load may not be pending either; its own destination may
            addr = regs_of(text.split(",", 1)[1]) if "," in text else set()
            touched = addr & set(pend_in[i])
        if touched:
            findings.append((name, no, text, sorted(touched), sorted({pend_in[i][r] for r in touched})))
    return findings


def lint_mfma_readers(name, lines):
    """The first reader of a matrix instruction's destination must be an instruction the compiler knows: it counts the wait
    states a VALU read of an MFMA result needs and fills them with s_nop - but not for inline assembly, whose operands it
    does not look into.  (Round 5: a build whose first reader was an inline `v_msad_u8` read the accumulator too early -
    wrong table rows, every hand-off failed.)  Linear scan in program order; a finding = an ASM-block instruction naming a
    register that an MFMA wrote and nothing else has touched since (the scan forgets at an unconditional branch)."""
    fresh, findings, in_asm = {}, [], False
    for no, raw in lines:
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or re.match(r"^\.?\w+:", line):
            continue
        code = line.partition(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):   # what follows is not reached from here
            fresh = {}
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            ops = code[len(op):].split(",")
            for part in ops[1:]:
                for r in regs_of(part):
                    fresh.pop(r, None)          # MFMA -> MFMA dependencies are the compiler's (and the hardware's) business
            for r in regs_of(ops[0]):
                fresh[r] = no
            continue
        touched = regs_of(code) & set(fresh)
        if not touched:
            continue
        if in_asm:
            findings.append((name, no, line, sorted(touched), sorted({fresh[r] for r in touched})))
        owners = {fresh[r] for r in touched}      # the wait the compiler placed covers the whole result of those matrix instructions
        fresh = {r: o for r, o in fresh.items() if o not in owners}
    return findings


def lint_sdwa_forwarding(name, lines):
    """gfx940 / gfx950 need one wait state between a VALU write with a destination select (SDWA dst_sel BYTE_n / WORD_n, which
    writes a part of the register) and a VALU read of that register.  The compiler's hazard recognizer inserts it for the SDWA
    instructions IT emits - not behind inline assembly (iqd_prims.h: cast_pack_i16_bounded, iqd_mfma.h: the byte negations),
    whose operands it does not look into.  Linear scan: a finding = a vector instruction that names the register directly
    behind such a write (as a source, or as the destination of another partial write, which reads the rest of it).  Anything
    in between - an s_nop, a scalar, LDS or memory instruction, another vector instruction - is the wait state."""
    findings, in_asm, pending = [], False, None     # pending: (register, line) of an inline partial write by the previous instruction
    for no, raw in lines:
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or re.match(r"^\.?\w+:", line):
            continue
        code = line.partition(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if pending is not None and op.startswith("v_") and pending[0] in regs_of(code):
            findings.append((name, no, line, [pending[0]], [pending[1]]))
        pending = None
        if in_asm and "_sdwa" in op:
            m = re.search(r"dst_sel:(\w+)", code)
            if m and m.group(1) != "DWORD":
                dst = sorted(regs_of(code[len(op):].split(",")[0]))
                if dst:
                    pending = (dst[0], no)
    return findings


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rtlsdrdiags_amd", "csrc", "iqd_stream2.hip")
    text = compile_to_asm(src)
    kernels = {}
    cur = None
    for no, line in enumerate(text.splitlines(), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            kernels[cur].append((no, line))
            if line.strip().startswith("s_endpgm"):
                cur = None
    findings = []
    n_loads = 0
    for name, lines in kernels.items():
        n_loads += sum(1 for _, l in lines if "global_load_dword" in l)
        findings += lint_kernel(name, lines)
    mfma_findings = []
    for name, lines in kernels.items():
        mfma_findings += lint_mfma_readers(name, lines)
    sdwa_findings = []
    for name, lines in kernels.items():
        sdwa_findings += lint_sdwa_forwarding(name, lines)
    for name, no, raw, regs, owners in sdwa_findings:
        print("%s: line %d reads v%s directly behind the inline SDWA partial write at line %s (one wait state needed):\n    %s"
              % (name, no, regs, owners, raw))
    mfma_findings = mfma_findings + sdwa_findings
    for name, no, raw, regs, owners in mfma_findings[:len(mfma_findings) - len(sdwa_findings)]:
        print("%s: line %d (inline assembly) is the first reader of v%s, written by the matrix instruction(s) at line(s) %s:\n    %s"
              % (name, no, regs, owners, raw))
    seen = set()
    for name, no, raw, regs, owners in findings:
        key = (name, no)
        if key in seen:
            continue
        seen.add(key)
        print("%s: line %d touches v%s of the untracked load(s) at line(s) %s before arrival:\n    %s"
              % (name, no, regs, owners, raw))
    # Kernels with a private segment (vector registers spilled to scratch), from the code object's metadata.  The launch that
    # holds all four families' pipelines must have none: a build of it with 4 spilled VGPRs beside its ~300 scalars spilled
    # to vector lanes ABORTED on the device (round 4, gpurun_out/dbg3.log); the WBFM kernel's gain-epoch instantiations
    # have always had one and run (tests/test_gpu_gain_epochs.py).  Listed here; tests/test_isa_lint.py says which may.
    scratch = []
    for m in re.finditer(r"\.name:\s+(\S+)\s*\n(?:.*\n)*?\s*\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s*\.vgpr_spill_count:\s+(\d+)", text):
        name, priv, spills = m.group(1), int(m.group(2)), int(m.group(3))
        if name in kernels and any("global_load_dword" in l for _, l in kernels[name]) and (priv or spills):
            scratch.append(name)
            print("scratch: %s: private segment of %d bytes, %d vector registers spilled" % (name, priv, spills))
    print("%s: %d kernels, %d global loads, %d finding(s), %d kernel(s) with scratch" % (os.path.basename(src), len(kernels), n_loads, len(seen) + len(mfma_findings), len(scratch)))
    return 1 if seen or mfma_findings else 0


if __name__ == "__main__":
    sys.exit(main())
