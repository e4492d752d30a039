"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run: python3 tools/timeline.py <dir> [n_last]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Queue_Id", "")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
t0 = rows[-n][0]
for s, e, k, q in rows[-n:]:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
