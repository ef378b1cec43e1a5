"""Turn the rocprofv3 outputs merged into gpurun_out/ into the small summaries kept under profiles/:
  profiles/<tag>_kernel_stats.csv   -- `rocprofv3 --kernel-trace --stats` per-kernel summary (as emitted)
  profiles/<tag>_pmc_traffic.json   -- HBM bytes per launch per kernel from two separate --pmc passes
                                       (FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md prescribes
usage: python tools/collect_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> [what]
`what` (optional): a suffix for the two files, e.g. "metrics_kernels" -> profiles/<tag>_metrics_kernels_{kernel_stats.csv,pmc_traffic.json}"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
if len(sys.argv) > 5:
    tag = f"{tag}_{sys.argv[5]}"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def newest(pattern):  # gpurun_out/ accumulates the outputs of earlier calls: take the latest one
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(os.path.join(stats_dir, "*", "*_kernel_stats.csv")), os.path.join(out, f"{tag}_kernel_stats.csv"))


COUNTS = {}   # kernel -> dispatches seen in the FETCH_SIZE pass


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(d, "*", "*_counter_collection.csv")))):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if counter == "FETCH_SIZE":
        COUNTS.update({k: len(v) for k, v in acc.items()})
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
res = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; counters are "
                  "KiB per dispatch averaged over dispatches; bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): on gfx950 FETCH_SIZE "
                  "counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact"}
for k in sorted(set(fetch) | set(write)):
    if "anonymous namespace" not in k:
        continue  # only this library's kernels
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    res[k] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int(1024 * (2 * f + w)),
              "dispatches": COUNTS.get(k, 0)}
# whole training step: the forward recurrence kernel runs exactly twice per step (layer 0, layer 1) -> steps in the pass
n_fwd = sum(c for k, c in COUNTS.items() if "gru_fwd_kernel" in k)
if n_fwd >= 2:
    steps = n_fwd / 2
    total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for k, v in res.items() if isinstance(v, dict))
    res["_steps_in_pass"] = steps
    res["_step_traffic_bytes"] = int(total / steps)
    for k, v in res.items():
        if isinstance(v, dict):
            v["launches_per_step"] = round(v["dispatches"] / steps, 2)
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k[:70]: v for k, v in res.items() if "gru" in k}, indent=1))
