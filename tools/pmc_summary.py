"""Average a PMC counter per kernel name from a rocprofv3 --pmc counter_collection CSV."""
import csv, sys, collections
f, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != counter: continue
    k = r["Kernel_Name"].replace("dex::", "").split("(")[0][:60]
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{counter:12s} {k:62s} launches={n:6d} avg={v/n:14.1f} total={v:16.1f}")
