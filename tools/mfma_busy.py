"""MFMA-pipe utilisation per kernel from one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
--output-format csv` run (counters in their own pass: dispatches are serialised, every kernel is measured alone).
mfma_busy_frac = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 XCDs * 1024 SIMDs) over the dispatches of a kernel:
the fraction of SIMD cycles the matrix pipe was busy at the clock the chip actually ran (f32 MFMA 32x32x2: 64 cycles per
instruction per SIMD).  usage: python tools/mfma_busy.py <counter_collection.csv> <out.json> "<what was run>" """
import collections
import csv
import json
import sys

path, out, what = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
res = {"_method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + what +
                  ".  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), summed over the "
                  "dispatches of a kernel (each dispatch runs alone under the counter pass)."}
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if busy <= 0 or act <= 0:
        continue
    res[k] = {"dispatches": cnt[k], "mfma_busy_frac": round(busy / (act / 8 * 1024), 3)}
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
