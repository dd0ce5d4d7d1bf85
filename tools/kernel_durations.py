"""Per-launch durations of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace --output-format csv directory.
    python tools/kernel_durations.py <dir> <pattern>"""
import csv, glob, os, sys
d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            print(r["Kernel_Name"][:48], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")),
                  round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms")
