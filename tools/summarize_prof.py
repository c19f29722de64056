"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output) into the small tracked files under profiles/.
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of bench.py
  profiles/<tag>_pmc_summary.json     FETCH_SIZE / WRITE_SIZE per launch for the pipeline kernels, with the
                                      gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2 for 16 B/lane
                                      streaming reads) and a calibration on k_split (known byte count)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f in ("bench_under_kt.json",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))


def per_kernel(path, counter):
    acc = defaultdict(list)
    if not os.path.exists(path):
        return acc
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]].append(float(row["Counter_Value"]))
    return acc


fetch = per_kernel(os.path.join(src, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "write", "write_counter_collection.csv"), "WRITE_SIZE")
cal_f = per_kernel(os.path.join(src, "cal_fetch", "cal_counter_collection.csv"), "FETCH_SIZE")
cal_w = per_kernel(os.path.join(src, "cal_write", "cal_counter_collection.csv"), "WRITE_SIZE")

out = {"source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py batch 2^16)",
       "units": "counters are KiB; bytes = value * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half "
                "of a 16 B/lane streaming read)", "kernels": {}}
# the PMC runs use bench.py --steps 2 --warmup 1 : 3 pipeline steps, so per-step sums = total / 3
STEPS = 3
for k in ("k_expand_runs", "k_expand", "k_expand_fb_run", "k_batch_inv", "k_chains", "k_scalar"):
    f = fetch.get(k, [])
    w = write.get(k, [])
    if not f and not w:
        continue
    fb = sum(f) * 1024 / STEPS if f else None
    wb = sum(w) * 1024 / STEPS if w else None
    out["kernels"][k] = {"launches_per_step": len(f) // STEPS, "FETCH_SIZE_bytes_raw_per_step": fb,
                         "WRITE_SIZE_bytes_per_step": wb, "fetch_bytes_corrected_x2_per_step": fb * 2 if fb else None,
                         "hbm_bytes_per_step": (fb * 2 if fb else 0) + (wb or 0)}
# per launch, for bench.py's roofline.traffic (same "per launch" basis as roofline.achieved)
out["hbm_bytes_per_launch"] = {k: v["hbm_bytes_per_step"] / max(1, v["launches_per_step"])
                               for k, v in out["kernels"].items() if k.startswith("k_expand")}
out["hbm_bytes_per_step_all_kernels"] = sum(v["hbm_bytes_per_step"] for v in out["kernels"].values())
cal = []
fl, wl = cal_f.get("k_split", []), cal_w.get("k_split", [])
per = max(1, len(fl) // 3)                      # dispatches per size (warm-up + REPS), sizes in launch order
for i, n in enumerate((1 << 20, 1 << 24, 1 << 26)):
    if (i + 1) * per <= len(fl) and (i + 1) * per <= len(wl):
        fv, wv = fl[(i + 1) * per - 1], wl[(i + 1) * per - 1]
        cal.append({"n": n, "read_bytes_true": 32 * n, "FETCH_SIZE_bytes_raw": fv * 1024,
                    "ratio_raw_over_true": round(fv * 1024 / (32 * n), 3), "written_bytes_true": 72 * n,
                    "WRITE_SIZE_bytes": wv * 1024, "ratio_write": round(wv * 1024 / (72 * n), 3)})
out["calibration_k_split"] = cal
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
