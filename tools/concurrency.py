"""From a rocprofv3 --kernel-trace CSV dir: how many kernels ran at once over the last `frac` of the trace (the timed region),
time per concurrency level, and kernel time by name.   tools/concurrency.py <dir> [frac]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [r for r in csv.DictReader(open(f)) if "nxd::" in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void nxd::", "")) for r in rows)
t1 = max(e for _, e, _ in iv)
t0 = min(s for s, _, _ in iv)
cut = t1 - (t1 - t0) * frac
iv = [(max(s, cut), e, n) for s, e, n in iv if e > cut]
ev = sorted([(s, 1) for s, _, _ in iv] + [(e, -1) for _, e, _ in iv])
lvl, last, hist = 0, cut, collections.Counter()
for t, d in ev:
    hist[lvl] += t - last
    last = t
    lvl += d
span = t1 - cut
print("window %.2f ms, %d kernels" % (span / 1e6, len(iv)))
for k in sorted(hist):
    print("  %2d kernels running: %6.2f ms  %5.1f %%" % (k, hist[k] / 1e6, 100.0 * hist[k] / span))
by = collections.Counter()
for s, e, n in iv:
    by[n[:44]] += e - s
for n, v in by.most_common():
    print("  %9.2f ms  %s" % (v / 1e6, n))
