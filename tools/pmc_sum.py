"""Sum the PMC counters of a rocprofv3 --pmc output directory per kernel name (short) and print a table."""
import csv, glob, sys, collections, re
d = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        k = k.replace("void nxd::", "").replace("nxd::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k, "dispatches", len(cnt[k]))
    for c in sorted(tot[k]):
        print("   %-40s %.6g  per-dispatch %.6g" % (c, tot[k][c], tot[k][c] / max(1, len(cnt[k]))))
