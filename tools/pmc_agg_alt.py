"""rocprofv3 --pmc counter_collection.csv of a run in which every diagonal was launched TWICE back to back (pmc_slice ... rep 2, -DPM_PROBES library): per kernel, the counters
of the first and of the second launch of each pair (dispatches of one kernel alternate first / second in dispatch order)."""
import collections, csv, sys
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-44:])][r["Counter_Name"]] = float(r["Counter_Value"])
per = collections.defaultdict(list)
for (did, k), c in sorted(rows.items()):
    per[k].append(c)
for k, lst in sorted(per.items(), key=lambda kv: -len(kv[1])):
    if "sweep" not in k or len(lst) < 4:
        continue
    for name in sorted(lst[0]):
        a = [c.get(name, 0.0) for c in lst[0::2]]; b = [c.get(name, 0.0) for c in lst[1::2]]
        n = min(len(a), len(b))
        print("%-46s %-22s pairs %5d  first %.5g  second %.5g  second/first %.3f" % (k, name, n, sum(a[:n]) / n, sum(b[:n]) / n, (sum(b[:n]) / max(1e-9, sum(a[:n])))))
