"""From a rocprofv3 --kernel-trace CSV dir: per Queue_Id busy time, union busy time, and pairwise overlap."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "nxd::" in r["Kernel_Name"]]
byq = collections.defaultdict(list)
for r in rows:
    byq[(r["Queue_Id"], r["Stream_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [x for v in byq.values() for x in v]
t0, t1 = min(s for s, _ in allv), max(e for _, e in allv)
print("span %.2f ms, union busy %.2f ms" % ((t1 - t0) / 1e6, union(allv) / 1e6))
for q, v in sorted(byq.items()):
    print("queue/stream", q, "kernels", len(v), "busy (union) %.2f ms" % (union(v) / 1e6), "first start %.2f ms last end %.2f ms" % ((min(s for s, _ in v) - t0) / 1e6, (max(e for _, e in v) - t0) / 1e6))
