import csv, sys, glob, collections
# usage: pmc_summary.py <dir> <kernel-substring>
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[2:] if len(v) > 4 else v     # drop warm-up dispatches
    print("%-32s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
