import csv, sys, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _, c), v in per.items():
            rows[k][c].append(v)
names = sorted({c for k in rows for c in rows[k]})
for k in rows:
    if not any(t in k for t in ("sweep", "gather", "roll", "warp_sample", "project_kernel", "norm_kernel")): continue
    print(k[:70])
    for c in names:
        v = rows[k].get(c)
        if v: print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
