"""Print the kernel timeline of one pass from a rocprofv3 --kernel-trace CSV directory: tools/pass_timeline.py <dir> [pass index]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "begin_frame" in r["Kernel_Name"]]
s, e = idx[which], idx[which + 1]
t0 = int(rows[s]["Start_Timestamp"])
line = []
for r in rows[s:e]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void nxd::", "").replace("nxd::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if n.startswith("trace_kernel<false"):
        line.append("T%.2f" % d)
    elif n.startswith("trace_kernel<true"):
        line.append("S%.2f" % d)
    elif n.startswith("logic"):
        line.append("L%.2f" % d)
print("pass %.2f ms: %s" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e6, " ".join(line)))
