#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag>/ into profiles/<tag>/ and derive the HBM traffic figure
bench.py reports (roofline.traffic): FETCH_SIZE and WRITE_SIZE are KiB counts from separate --pmc passes; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so the read side is
reported as a range [raw, 2 x raw] and `traffic` uses the corrected (upper) value."""
import csv, json, os, shutil, subprocess, sys

# usage: summarize_profile.py <tag> [workload@log2]      (default pairing@20, the headline; the key bench.py looks its traffic up under)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "pairing@20"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)

# one bench step = one launch of each ("k_final_exp" also matches k_final_exp_given; only one of the two runs)
KERNELS = ("k_multi_miller", "k_prepared_miller", "k_miller_ref_opt", "k_fexp_norm", "k_fp_batch_inv", "k_final_exp", "k_fexp_stage",
           "k_fexp_glue", "k_fexp_chain")
LEAD = ("k_multi_miller", "k_prepared_miller", "k_miller_ref_opt")      # the first kernel of a step: one dispatch per step
def is_lead(name): return any(kn in name for kn in LEAD)


def counters(path):
    """per-step counter totals: all dispatches of the step's kernels summed, divided by the number of steps (= launches
    of the Miller kernel; the staged final exponentiation launches each of its kernels several times per step)"""
    tot, steps = {}, set()
    for r in csv.DictReader(open(path)):
        if not any(kn in r["Kernel_Name"] for kn in KERNELS):
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if is_lead(r["Kernel_Name"]):
            steps.add(r["Dispatch_Id"])
    n = max(1, len(steps))
    return {c: v / n for c, v in tot.items()}


def step_walls(trace_path):
    """wall time of each bench step from the kernel trace: Miller kernel start to the end of the last kernel before the
    next Miller kernel (kernels of the two final-exponentiation pipelines overlap, so durations do not add up)"""
    rows = [r for r in csv.DictReader(open(trace_path)) if any(kn in r["Kernel_Name"] for kn in KERNELS)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    walls, start, end = [], None, None
    for r in rows:
        if is_lead(r["Kernel_Name"]):
            if start is not None:
                walls.append((end - start) / 1e6)
            start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        elif start is not None:
            end = max(end, int(r["End_Timestamp"]))
    if start is not None:
        walls.append((end - start) / 1e6)
    return walls

for sub, name in (("stats/s_kernel_stats.csv", "kernel_stats.csv"), ("stats/s_kernel_trace.csv", "kernel_trace.csv"),
                  ("fetch/c_counter_collection.csv", "pmc_fetch_size.csv"), ("write/c_counter_collection.csv", "pmc_write_size.csv"),
                  ("sq/c_counter_collection.csv", "pmc_sq.csv"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                  ("bench_unprofiled.json", "bench_unprofiled.json"), ("bench_flags.txt", "bench_flags.txt")):
    p = os.path.join(src, sub)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
fetch = counters(os.path.join(src, "fetch/c_counter_collection.csv"))["FETCH_SIZE"]
write = counters(os.path.join(src, "write/c_counter_collection.csv"))["WRITE_SIZE"]
sq = counters(os.path.join(src, "sq/c_counter_collection.csv"))
stats = list(csv.DictReader(open(os.path.join(src, "stats/s_kernel_stats.csv"))))
ks = [r for r in stats if any(kn in r["Name"] for kn in KERNELS)]
summary = {
    "tag": tag, "workload": key, "git": subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "kernels": {r["Name"].split("(")[0]: {"avg_ms": float(r["AverageNs"]) / 1e6, "calls": int(r["Calls"])} for r in ks},
    "rocprof_step_wall_ms": step_walls(os.path.join(src, "stats/s_kernel_trace.csv")),
    "rocprof_busy_ms_per_step": sum(float(r["TotalDurationNs"]) for r in ks) / 1e6 / max(1, max(int(r["Calls"]) for r in ks if is_lead(r["Name"]))),
    "fetch_size_kib": fetch, "write_size_kib": write,
    "hbm_read_bytes_raw": fetch * 1024, "hbm_read_bytes_corrected_x2": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
    "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "sq": sq,
}
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
# profiles/latest_traffic.json: one entry per workload (bench.py: roofline.traffic), each stamped with its tag and git revision
lt_path = os.path.join(root, "profiles", "latest_traffic.json")
try:
    lt = json.load(open(lt_path))
    if "workloads" not in lt:
        lt = {"workloads": {}}
except Exception:
    lt = {"workloads": {}}
lt["workloads"][key] = {k: summary[k] for k in ("tag", "git", "workload", "traffic_bytes_per_launch", "hbm_read_bytes_raw", "hbm_write_bytes")}
json.dump(lt, open(lt_path, "w"), indent=1)
print(json.dumps(summary, indent=1))
