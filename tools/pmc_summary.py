"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (debug tool)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else "ecal"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if filt not in k: continue
    print(k)
    for c, val in sorted(v.items()):
        print("   %-26s per dispatch %16.0f" % (c, val / cnt[(k, c)]))
