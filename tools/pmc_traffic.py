"""profiles/<tag>_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
HBM bytes per launch = 2 * FETCH_SIZE(KB) * 1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B,
MI355X_MICROARCH.md §HBM) + WRITE_SIZE(KB) * 1024.   usage: pmc_traffic.py fetch.csv write.csv out.json events"""
import collections, csv, json, sys


def per_dispatch(path, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "ecal::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    LAUNCHES.update({k: max(LAUNCHES.get(k, 0), cnt[k]) for k in cnt})
    return {k: agg[k] / cnt[k] for k in agg}


LAUNCHES = {}
f = per_dispatch(sys.argv[1], "FETCH_SIZE")
w = per_dispatch(sys.argv[2], "WRITE_SIZE")
out = {"events": int(sys.argv[4]), "note": "bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (gfx950 FETCH_SIZE correction)",
       "kernels": {k: {"fetch_size_kb": f.get(k, 0.0), "write_size_kb": w.get(k, 0.0), "launches": LAUNCHES.get(k, 0),
                       "hbm_bytes_per_launch": 2 * f.get(k, 0.0) * 1024 + w.get(k, 0.0) * 1024} for k in sorted(set(f) | set(w))}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-50s %10.1f MB" % (k, v["hbm_bytes_per_launch"] / 1e6))
