"""Sum rocprofv3 --pmc counter_collection.csv values per (kernel, counter)."""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    if "rocclr" in k: continue
    print(f"{k:28s} {c:28s} n={n[(k,c)]:4d} total={v:16.0f} avg={v/n[(k,c)]:16.1f}")
