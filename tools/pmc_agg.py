"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum and per-dispatch mean)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:6]:
    print(k, "dispatches", cnt[k])
    for c, v in sorted(agg[k].items()):
        print("   %-28s sum %.4g  per-dispatch %.4g" % (c, v, v / max(1, cnt[k])))
