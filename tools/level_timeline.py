"""Kernel timeline of one pass from a rocprofv3 --kernel-trace CSV directory, with the idle gaps between dependent kernels:
    tools/level_timeline.py <dir> [pass index]
Prints per kernel: start offset, duration, gap since the latest end seen so far (negative: overlapped), and totals."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "begin_frame" in r["Kernel_Name"]]
s = idx[which]
e = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = int(rows[s]["Start_Timestamp"])
last_end = t0
busy = {}
gap_total = 0.0
for r in rows[s:e]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void nxd::", "").replace("nxd::", "")
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - last_end) / 1e3
    if gap > 0:
        gap_total += gap
    print("%9.1f us  dur %8.1f us  gap %7.1f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, n[:70]))
    busy[n[:40]] = busy.get(n[:40], 0.0) + (en - st) / 1e3
    last_end = max(last_end, en)
print("pass span %.1f us, idle gaps %.1f us" % ((last_end - t0) / 1e3, gap_total))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("  %9.1f us  %s" % (v, k))
