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
    # (the matcher: the instance a batch runs in -- <1> exhaustive, <2> pruned; <0> behind them only takes the few pairs left over)
    if k == "k_match_close_batch" and any(q[0] in (k + "<1>", k + "<2>") for q in keys): keys = [q for q in keys if q[0] != k + "<0>"]
    return sum(acc[q] for q in keys) / max(sum(n[q] for q in keys), 1), sum(n[q] for q in keys)
traffic = {}
for k in ("k_front_level", "k_front_factor", "k_front_update", "k_solve_bwd", "k_top_block", "k_assemble", "k_linearize", "k_match_close_batch"):
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
        wany, _ = avg(k, "SQ_WAIT_ANY"); busy, _ = avg(k, "SQ_BUSY_CYCLES")
        if wavec > 0: traffic[k]["wait_any_frac"] = round(wany / wavec, 4)
        # LDS-array cycles (all CUs) against the CU-cycles of the launch: SQ_WAVE_CYCLES counts every wavefront every 4th cycle,
        # 8 wavefronts per CU are resident for the whole launch
        if wavec > 0 and act > 0: traffic[k]["lds_array_busy_frac"] = round(act / (wavec * 4 / 8), 4)
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

# ---- the C2-only kernel trace (tools/gn_profile_run.py under rocprofv3 --kernel-trace --stats, no counters)
c2_rows = []
c2_path = f"{O}/gn_c2_kernel_stats.csv"
if os.path.exists(c2_path):
    c2_rows = list(csv.DictReader(open(c2_path)))
    with open(f"{P}/{rnd}_gn_c2_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in c2_rows:
            name = r["Name"].split("(")[0].replace("cgmr::", "").replace("void ", "")
            w.writerow([name, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.1f}', f'{float(r["Percentage"]):.3f}', r["MinNs"], r["MaxNs"]])

# ---- profiles/rNN_summary.md: every number below is read from the files written above (nothing is typed in by hand)
def fnum(v, nd=1):
    return f"{v:,.{nd}f}"
L = [f"# Profile summary, round {rnd[1:]} (generated by tools/make_profile_summary.py from gpurun_out/prof_round; do not edit)", ""]
if c2_rows:
    L += ["## Gauss-Newton kernels, the C2 solve alone (rocprofv3 --kernel-trace --stats, tools/gn_profile_run.py: 3 x optimize(10))", "",
          "| kernel | calls | average us | share of kernel time % |", "|---|---|---|---|"]
    for r in sorted(c2_rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        name = r["Name"].split("(")[0].replace("cgmr::", "").replace("void ", "")
        L.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.1f} |")
    L.append("")
L += ["## HBM-side traffic per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes; corrected = 2 x FETCH_SIZE + WRITE_SIZE)", "",
      "| kernel | launches | FETCH_SIZE bytes | WRITE_SIZE bytes | corrected bytes |", "|---|---|---|---|---|"]
for k, t in traffic.items():
    if k.startswith("_") or not isinstance(t, dict) or "launches" not in t or not t["launches"]:
        continue
    L.append(f"| `{k}` | {t['launches']} | {t['fetch_bytes_raw']:,} | {t['write_bytes_raw']:,} | {t['traffic_bytes_corrected']:,} |")
L.append("")
if "_calibration" in traffic:
    L += ["FETCH_SIZE / WRITE_SIZE against known byte counts (tools/ubench/fetch_calib_ubench.hip, counter bytes / bytes moved):", ""]
    for k, d in traffic["_calibration"].items():
        if not k.startswith("_"):
            L.append(f"* `{k}`: " + ", ".join(f"{c} {v}" for c, v in d.items()))
    L.append("")
mk = "k_match_close_batch"
sq = {c: avg(mk, c)[0] for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                  "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT",
                                  "SQ_LDS_UNALIGNED_STALL")}
if sq["SQ_WAVE_CYCLES"] > 0:
    L += [f"## Matcher, `{mk}` (4096 pairs per launch, tools/match_profile_run.py; per-launch averages)", "",
          "| counter | per launch |", "|---|---|"] + [f"| {c} | {v:,.0f} |" for c, v in sq.items()] + [""]
    ratios = []
    if sq["SQ_LDS_IDX_ACTIVE"] > 0:
        ratios.append(f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {sq['SQ_LDS_BANK_CONFLICT'] / sq['SQ_LDS_IDX_ACTIVE']:.3f}")
    ratios.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}")
    ratios.append(f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES']:.3f}")
    ratios.append(f"SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES = {sq['SQ_ACTIVE_INST_LDS'] / sq['SQ_WAVE_CYCLES']:.3f}")
    t = traffic.get(mk, {})
    if t.get("launches"):
        ratios.append(f"HBM-side bytes per pair = {t['traffic_bytes_corrected'] / 4096:,.0f}")
    L += ["Ratios: " + "; ".join(ratios) + ".", ""]
    if "phase_cycles_per_pair" in traffic.get(mk, {}):
        L += ["Phase cycles per pair (timing build, tools/gpu_mphase.py): " + ", ".join(f"{k} {v}" for k, v in traffic[mk]["phase_cycles_per_pair"].items() if k != "note") + ".", ""]
if bench is not None:
    r = bench["roofline"]
    L += ["## The bench line of the same sources (profiles/" + rnd + "_bench_line.json; under the profiler: kernel-trace + stats)", "",
          f"* value {bench['value']} {bench['unit']}, {bench['ms_per_step']} ms per step (host analysis {bench['host_symbolic_ms_per_step']} ms, device {bench['device_ms_per_step']} ms), host load {bench.get('host_loadavg_1min')}",
          f"* `{r['kernel']}`: {r['avg_launch_us']} us per tree level by HIP events ({r.get('merged_levels_per_gn_iter')} of {r.get('launched_levels')} levels as one merged launch of {r.get('merged_level_launch_us')} us), {r['launches_per_gn_iter']} forward-pass launches per GN iteration; algorithmic {r['algorithmic_bytes_per_launch']:,} B per level "
          f"-> {r['achieved']} GB/s = {r['frac']} of 8 TB/s (on the committed rocprofv3 averages: {r.get('frac_on_rocprofv3_avg')}); B_iter_frac {r.get('B_iter_frac')}",
          f"* matcher {bench['matcher']['value']:,.0f} pairs/s (pruned search), {bench['matcher']['exhaustive']['pairs_per_s']:,.0f} exhaustive; LDS-gather roofline fraction {bench['matcher']['roofline']['frac']}",
          f"* C5: one robot alone {bench['exchange']['round_ms_mean_max']} ms per round; eight robots on this GPU {bench['exchange_loopback']['round_ms_per_robot']} ms per robot and round "
          f"(the same robots alone {bench['exchange_loopback']['solo_round_ms_same_robots']['mean']} ms): predicted_weak_scaling_efficiency_8 {bench.get('predicted_weak_scaling_efficiency_8')}", ""]
    rows = list(csv.DictReader(open(f"{O}/bench_kernel_stats.csv")))
    L += ["Kernel statistics of the whole bench run (all legs: C2, key-frame graphs, C5 graphs growing from 50 vertices, matcher):", "",
          "| kernel | calls | average us | share % |", "|---|---|---|---|"]
    for r2 in sorted(rows, key=lambda q: -float(q["TotalDurationNs"]))[:14]:
        name = r2["Name"].split("(")[0].replace("cgmr::", "").replace("void ", "")
        L.append(f"| `{name[:60]}` | {r2['Calls']} | {float(r2['AverageNs']) / 1e3:.2f} | {float(r2['Percentage']):.1f} |")
    L.append("")
open(f"{P}/{rnd}_summary.md", "w").write("\n".join(L))
