#!/usr/bin/env python3
"""Condenses rocprofv3 CSVs (tools/profile.sh writes them under gpurun_out/<tag>/; bench.py collects a short set of its
own) into one summary: per-launch counter values of the dominant chain kernel, the kernel-trace average duration, HBM
bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 for 16-B-per-lane streams), the kernel's register /
LDS / scratch figures from the code object's own notes (the trace's VGPR column is not the allocation), and a hash of
the kernel sources, so that a summary cannot pass for a measurement of different code.

    python tools/pmc_summary.py gpurun_out/<tag> <kernel-name-substring> <samples per launch> [out.json]
"""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtlsdrdiags_amd", "csrc")


def source_hash():
    """sha256 (first 16 hex digits) over the kernel and engine sources: what a profile belongs to."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp", ".cc")) or name == "Makefile":
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def code_object_notes(kernel_demangled):
    """vgpr / sgpr / spills / scratch / static LDS of a kernel from the code object metadata (compiles the translation unit
    to assembly with the Makefile's flags: a few seconds)."""
    m = re.match(r"(?:void )?(?:iqd::)?(\w+)", kernel_demangled or "")
    if not m:
        return {}
    base = m.group(1)
    tpl = re.search(r"<(.*)>\(", kernel_demangled)
    out = {}
    for unit in ("iqd_stream_mixed.hip", "iqd_stream.hip", "iqd_stream2.hip", "iqd_kernels.hip"):
        src = os.path.join(CSRC, unit)
        if base not in open(src).read():
            continue
        fd, path = tempfile.mkstemp(suffix=".s")
        os.close(fd)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", "-w",
               "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", path, src]
        if subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode != 0:
            os.unlink(path)
            continue
        text = open(path).read()
        os.unlink(path)
        # the metadata block lists kernels as "- .agpr_count ... .name: <mangled> ... .vgpr_count: N"
        for block in text.split("\n  - .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name or base not in name.group(1):
                continue
            if tpl:   # match the template arguments through the mangled name (Li1E = 1, Lb1E = true, Lin1E = -1 ...)
                want = []
                for a in [x.strip() for x in tpl.group(1).split(",")]:
                    if a in ("true", "false"):
                        want.append("Lb%dE" % (a == "true"))
                    elif re.fullmatch(r"-?\d+", a):
                        want.append("Li%s%sE" % ("n" if a.startswith("-") else "", a.lstrip("-")))
                if want and "".join(want) not in name.group(1):
                    continue
            for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                        "group_segment_fixed_size", "agpr_count"):
                mm = re.search(r"\.%s:\s+(\d+)" % key, "  - .agpr_count" + block)
                if mm:
                    out[key] = int(mm.group(1))
            out["code_object_kernel"] = name.group(1)
            return out
    return out


def rows(pattern):
    # gpurun merges every call's output into the same local directories: only the newest file of a pass counts
    paths = glob.glob(pattern, recursive=True)
    if paths:
        with open(max(paths, key=os.path.getmtime), newline="") as f:
            yield from csv.DictReader(f)


def summarize(root, needle, samples, passes=("fetch", "write", "sq1", "sq2", "sq3"), notes=True):
    res = {"source": root, "kernel_match": needle, "samples_per_launch": samples, "sources_sha16": source_hash()}
    durs = []
    for r in rows(os.path.join(root, "kt", "**", "*kernel_trace.csv")):
        if needle in r["Kernel_Name"]:
            durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            res["kernel"] = r["Kernel_Name"][:160]
            wg = int(r.get("Workgroup_Size_X") or 0)
            if wg:
                res["workgroup"] = wg
                res["grid_workgroups"] = int(r["Grid_Size_X"]) // wg
            res["trace_scratch_bytes_per_lane"] = int(r.get("Scratch_Size") or 0)
    if durs:
        durs.sort()
        res["kernel_trace_launches"] = len(durs)
        res["kernel_ms_avg"] = sum(durs) / len(durs) / 1e6
        res["kernel_ms_min"] = durs[0] / 1e6
        res["kernel_ms_median"] = durs[len(durs) // 2] / 1e6
    counters = {}
    for sub in passes:
        acc, disp = {}, {}
        for r in rows(os.path.join(root, sub, "**", "*counter_collection.csv")):
            if needle not in r["Kernel_Name"]:
                continue
            key = r["Counter_Name"]
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
            disp.setdefault(key, set()).add(r["Dispatch_Id"])
            res.setdefault("kernel", r["Kernel_Name"][:160])
        for k in acc:
            # several rows per dispatch (one per XCD / dimension) are summed; dispatches are averaged
            counters[k] = acc[k] / max(len(disp[k]), 1)
    res["counters_per_launch"] = counters
    if notes and res.get("kernel"):
        res["code_object"] = code_object_notes(res["kernel"])
    d = {}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        d["hbm_bytes_per_launch"] = counters["FETCH_SIZE"] * 1024 * 2 + counters["WRITE_SIZE"] * 1024
        d["hbm_correction"] = "FETCH_SIZE x 2 (gfx950 tallies 128-B requests of 16-B-per-lane streams at 64 B), WRITE_SIZE as is"
        d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / (samples * 2.0625)
    if "SQ_INSTS_VALU" in counters:
        d["valu_lane_ops_per_sample"] = counters["SQ_INSTS_VALU"] * 64 / samples
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in counters:
                d[k + "_over_WAVE_CYCLES"] = counters[k] / wc
    if "SQ_LDS_BANK_CONFLICT" in counters and counters.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]
    if counters.get("GRBM_GUI_ACTIVE") and res.get("kernel_ms_avg"):
        # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back); the PMC passes run a little slower than the trace
        # (the counter pass is not the traced pass: when its kernel ran longer the quotient exceeds the 2.4 GHz the
        # part can clock at - then the nominal peak is used, which can only understate the busy fraction)
        raw = counters["GRBM_GUI_ACTIVE"] / 8 / (res["kernel_ms_avg"] * 1e-3) / 1e9
        res["clock_ghz_raw"] = round(raw, 3)
        res["clock_ghz"] = round(min(raw, 2.4), 3)
    if "SQ_ACTIVE_INST_VALU" in counters and res.get("kernel_ms_avg"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs; 1024 SIMDs on the chip
        ghz = res.get("clock_ghz", 2.3)
        d["valu_busy_ms_per_simd"] = counters["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (ghz * 1e9) * 1e3
        d["valu_issue_frac"] = d["valu_busy_ms_per_simd"] / res["kernel_ms_avg"]
    if counters.get("SQ_INSTS_MFMA"):
        d["mfma_per_kilosample"] = counters["SQ_INSTS_MFMA"] / samples * 1000
    res["derived"] = d
    return res


def main():
    root, needle, samples = sys.argv[1], sys.argv[2], float(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    res = summarize(root, needle, samples)
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
