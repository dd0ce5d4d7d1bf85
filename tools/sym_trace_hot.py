"""Medians per section of CGMR_SYM_TRACE over the cold optimize(10) calls of tools/gn_breakdown.py (the pool warm, as in the bench):
CGMR_SYM_TRACE=1 python tools/gn_breakdown.py 2> trace.txt; python tools/sym_trace_hot.py trace.txt"""
import sys, re, collections, statistics
d = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"\s+sym\s+(.*?)\s+([0-9.]+) us\s*$", line)
    if m:
        d.setdefault(m.group(1).strip(), []).append(float(m.group(2)))
    m = re.match(r"\s+nd depth (\d) n\s+(\d+) own work ([0-9.]+) us", line)
    if m and int(m.group(1)) <= 1:
        d.setdefault("nd depth %s own work" % m.group(1), []).append(float(m.group(3)))
tot = 0
for k, v in d.items():
    v = v[len(v) // 4:]
    med = statistics.median(v)
    if not k.startswith("nd depth"): tot += med
    print("%-44s %8.1f us (n %d)" % (k, med, len(v)))
print("sum of the sections' medians %.1f us" % tot)
