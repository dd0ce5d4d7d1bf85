"""Per-level launch durations of one GN iteration from a rocprofv3 kernel trace CSV (arg: *_kernel_trace.csv)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last optimize call: take the last iteration = launches after the last k_linearize that is followed by k_assemble
names = [r["Kernel_Name"].split("(")[0].split("<")[0].replace("cgmr::", "").replace("void ", "") for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_assemble"]
a = idx[-1]
# back up to the linearize before it
s = a - 1
e = a
while e + 1 < len(rows) and names[e + 1] != "k_linearize": e += 1
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
tot = collections.defaultdict(float)
for i in range(s, e + 1):
    st, en = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print(f"{names[i]:18s} grid {rows[i]['Grid_Size_X']:>8s} wg {rows[i]['Workgroup_Size_X']:>4s} start {1e-3*(st-t0):9.1f} us  dur {1e-3*(en-st):7.1f} us  gap {1e-3*(st-prev_end):6.1f} us")
    tot[names[i]] += 1e-3 * (en - st)
    prev_end = en
print("iteration span us", 1e-3 * (prev_end - t0), {k: round(v, 1) for k, v in tot.items()})
