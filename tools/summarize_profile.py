#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag>/ into profiles/<tag>/ and derive the HBM traffic figure
bench.py reports (roofline.traffic): FETCH_SIZE and WRITE_SIZE are KiB counts from separate --pmc passes; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so the read side is
reported as a range [raw, 2 x raw] and `traffic` uses the corrected (upper) value."""
import csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)

# one bench step = one launch of each ("k_final_exp" also matches k_final_exp_given; only one of the two runs)
KERNELS = ("k_multi_miller", "k_fexp_norm", "k_fp_batch_inv", "k_final_exp")


def counters(path):
    """per-step counter totals: mean over dispatches of each kernel, summed over the kernels of a step"""
    per = {}
    for r in csv.DictReader(open(path)):
        for kn in KERNELS:
            if kn in r["Kernel_Name"]:
                per.setdefault((kn, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    out = {}
    for (kn, c), v in per.items():
        out[c] = out.get(c, 0.0) + sum(v) / len(v)
    return out

for sub, name in (("stats/s_kernel_stats.csv", "kernel_stats.csv"), ("stats/s_kernel_trace.csv", "kernel_trace.csv"),
                  ("fetch/c_counter_collection.csv", "pmc_fetch_size.csv"), ("write/c_counter_collection.csv", "pmc_write_size.csv"),
                  ("sq/c_counter_collection.csv", "pmc_sq.csv"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                  ("bench_unprofiled.json", "bench_unprofiled.json")):
    p = os.path.join(src, sub)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
fetch = counters(os.path.join(src, "fetch/c_counter_collection.csv"))["FETCH_SIZE"]
write = counters(os.path.join(src, "write/c_counter_collection.csv"))["WRITE_SIZE"]
sq = counters(os.path.join(src, "sq/c_counter_collection.csv"))
stats = list(csv.DictReader(open(os.path.join(src, "stats/s_kernel_stats.csv"))))
ks = [r for r in stats if any(kn in r["Name"] for kn in KERNELS)]
summary = {
    "tag": tag, "kernels": {r["Name"].split("(")[0]: {"avg_ms": float(r["AverageNs"]) / 1e6, "calls": int(r["Calls"])} for r in ks},
    "rocprof_avg_ms": sum(float(r["AverageNs"]) for r in ks) / 1e6,
    "fetch_size_kib": fetch, "write_size_kib": write,
    "hbm_read_bytes_raw": fetch * 1024, "hbm_read_bytes_corrected_x2": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
    "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "sq": sq,
}
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(root, "profiles", "latest_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
