#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs that tools/profile.sh wrote under gpurun_out/<tag>/ into one JSON summary
(profiles/<name>.json): per-launch counter values of the dominant chain kernel, the kernel-trace average
duration, HBM bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 for 16-B-per-lane streams).

    python tools/pmc_summary.py gpurun_out/<tag> <kernel-name-substring> <samples per launch> [out.json]
"""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    # gpurun merges every call's output into the same local directories: only the newest file of a pass counts
    paths = glob.glob(pattern, recursive=True)
    if paths:
        with open(max(paths, key=os.path.getmtime), newline="") as f:
            yield from csv.DictReader(f)


def main():
    root, needle, samples = sys.argv[1], sys.argv[2], float(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    res = {"source": root, "kernel_match": needle, "samples_per_launch": samples}
    durs = []
    for r in rows(os.path.join(root, "kt", "**", "*kernel_trace.csv")):
        if needle in r["Kernel_Name"]:
            durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            res["kernel"] = r["Kernel_Name"][:120]
            res["vgpr"] = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
            res["lds_bytes"] = r.get("LDS_Block_Size")
            res["grid"] = r.get("Grid_Size")
            res["workgroup"] = r.get("Workgroup_Size")
    if durs:
        durs.sort()
        res["kernel_trace_launches"] = len(durs)
        res["kernel_ms_avg"] = sum(durs) / len(durs) / 1e6
        res["kernel_ms_min"] = durs[0] / 1e6
        res["kernel_ms_median"] = durs[len(durs) // 2] / 1e6
    counters = {}
    for sub in ("fetch", "write", "sq1", "sq2", "sq3"):
        acc, n = {}, {}
        for r in rows(os.path.join(root, sub, "**", "*counter_collection.csv")):
            if needle not in r["Kernel_Name"]:
                continue
            key = r["Counter_Name"]
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
            n[key] = n.get(key, 0) + 1
        for k in acc:
            # several rows per dispatch (one per XCD / dimension) are summed; dispatches are averaged
            disp = len({r["Dispatch_Id"] for r in rows(os.path.join(root, sub, "**", "*counter_collection.csv"))
                        if needle in r["Kernel_Name"] and r["Counter_Name"] == k})
            counters[k] = acc[k] / max(disp, 1)
    res["counters_per_launch"] = counters
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
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
