import csv, sys, glob, collections
# usage: pmc_summary.py <dir> <kernel-substring> [<kernel-substring> ...]
d, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for pat in pats:
            if pat in r.get("Kernel_Name", ""):
                acc[(pat, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (pat, k), v in sorted(acc.items()):
    v = v[len(v) // 10:] if len(v) > 20 else v     # drop warm-up dispatches
    print("%-16s %-28s n=%d mean=%.6g" % (pat, k, len(v), sum(v) / len(v)))
