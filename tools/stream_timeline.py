"""Kernel timeline by stream from a rocprofv3 --kernel-trace CSV dir, for the window [t0 + a ms, t0 + b ms]:
    tools/stream_timeline.py <dir> <a> <b> [min duration us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
a, b = float(sys.argv[2]), float(sys.argv[3])
mind = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
rows = [r for r in csv.DictReader(open(f)) if "nxd::" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e < a or s > b or (e - s) * 1e3 < mind:
        continue
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void nxd::", "").replace("nxd::", "")
    print("stream %3s queue %3s  %9.3f -> %9.3f ms  (%7.3f)  %s" % (r["Stream_Id"], r["Queue_Id"], s, e, e - s, n[:40]))
