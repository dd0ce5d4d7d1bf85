"""Turn gpurun_out/prof_round (tools/profile_round.sh) into the committed files under profiles/ for round NN."""
import csv, glob, json, os, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rnd = args[0] if args else "r01"
pmc_only = "--pmc-only" in sys.argv      # on the GPU box, between the counter passes and the profiled bench run: bench.py then
                                         # reads counters that belong to the sources it runs (tools/profile_round.sh)
O = "gpurun_out/prof_round"
P = "profiles"
bench = None
if not pmc_only:
    # bench line
    line = [l for l in open(f"{O}/bench.log") if l.startswith('{"metric"')][-1]
    bench = json.loads(line)
    open(f"{P}/{rnd}_bench_line.json", "w").write(line)
    # kernel stats (shortened names)
    rows = list(csv.DictReader(open(f"{O}/bench_kernel_stats.csv")))
    with open(f"{P}/{rnd}_bench_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)                      # (template arguments carry commas: quoted fields)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        for r in rows:
            name = r["Name"].split("(")[0].replace("cgmr::", "").replace("void ", "")
            if len(name) > 60: name = name[:57] + "..."
            w.writerow([name, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.1f}', f'{float(r["Percentage"]):.3f}', r["MinNs"], r["MaxNs"]])
# PMC summary: avg per launch per (kernel, counter)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
def newest_per_pass(pattern):
    """One counter file per pass directory: the newest (gpurun merges a session's files into what earlier sessions left)."""
    best = {}
    for path in glob.glob(pattern):
        key = path.split(os.sep)[-3]
        if key not in best or os.path.getmtime(path) > os.path.getmtime(best[key]): best[key] = path
    return sorted(best.values())
for path in newest_per_pass(f"{O}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("cgmr::", "").replace("void ", "")
        if "rocclr" in k or "at::" in k: continue
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(f"{P}/{rnd}_pmc_summary.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "avg_per_launch"])
    for (k, c) in sorted(acc):
        w.writerow([k, c, n[(k, c)], f"{acc[(k, c)] / n[(k, c)]:.1f}"])
def avg(k, c):
    # per-launch average over every instance of the kernel (k_front_factor<48>, k_front_factor_leaf, ...)
    keys = [q for q in acc if q[1] == c and (q[0] == k or q[0].startswith(k + "<") or q[0].startswith(k + "_"))]
    return sum(acc[q] for q in keys) / max(sum(n[q] for q in keys), 1), sum(n[q] for q in keys)
traffic = {}
for k in ("k_front_factor", "k_front_update", "k_solve_bwd", "k_top_block", "k_assemble", "k_linearize", "k_match_close_batch"):
    (fe, nl), (wr, _) = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
    fe, wr = fe * 1024, wr * 1024                                          # rocprofv3 reports KB
    traffic[k] = {"fetch_bytes_raw": round(fe), "write_bytes_raw": round(wr),
                  # MI355X_MICROARCH.md, HBM: FETCH_SIZE counts wide (16 B/lane) streaming reads at half their bytes
                  "traffic_bytes_corrected": round(2 * fe + wr), "launches": nl}
    if k == "k_match_close_batch":
        traffic[k]["pairs"] = 4096
        conf, _ = avg(k, "SQ_LDS_BANK_CONFLICT"); act, _ = avg(k, "SQ_LDS_IDX_ACTIVE")
        valu, _ = avg(k, "SQ_ACTIVE_INST_VALU"); wavec, _ = avg(k, "SQ_WAVE_CYCLES")
        if act > 0: traffic[k]["lds_bank_conflict_frac"] = round(conf / act, 4)
        if wavec > 0: traffic[k]["valu_active_frac"] = round(valu / wavec, 4)
        ph = f"{O}/match_phases.json"                                       # tools/gpu_mphase.py on the timing build
        if os.path.exists(ph): traffic[k]["phase_cycles_per_pair"] = json.load(open(ph))
# FETCH_SIZE / WRITE_SIZE calibration (tools/ubench/fetch_calib_ubench.hip: every kernel moves a known byte count once)
calib = {}
for path in newest_per_pass(f"{O}/calib_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_read") or k.startswith("k_write"):
            calib.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"]) * 1024
if calib:
    known = 512 << 20
    traffic["_calibration"] = {k: {c: round(v / known, 4) for c, v in d.items()} for k, d in sorted(calib.items())}
    traffic["_calibration"]["_note"] = "counter bytes / known bytes moved (512 MiB per kernel); k_read8_records = k_assemble's access pattern"
traffic["_note"] = ("per-launch averages from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh); "
                    "corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 half-count of wide streaming reads); matcher launch = 4096 pairs")
sys.path.insert(0, os.getcwd())
import bench as bench_py
traffic["kernel_sources_sha16"] = bench_py.kernel_sources_sha16()              # bench.py flags the counters as stale when the sources change
json.dump(traffic, open(f"{P}/{rnd}_pmc_traffic.json", "w"), indent=1)
if bench is not None:
    print(json.dumps(traffic, indent=1))
    print({k: bench[k] for k in ("value", "ms_per_step")}, bench["roofline"], bench["matcher"]["value"])
