"""Per-kernel FETCH_SIZE / WRITE_SIZE of one rocprofv3 --pmc pass: python3 tools/pmc_kernel_bytes.py <counter_collection.csv> <COUNTER> [name filter]
Prints launches, mean KiB-units per launch and MB per launch (gfx950: FETCH_SIZE counts 32-byte units x 2 in this image -- the
same correction tools/collect_profiles.py applies; WRITE_SIZE in KiB)."""
import collections
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == counter and flt in r["Kernel_Name"]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{len(v):5d} launches  {sum(v) / len(v) * scale / 1e6:9.2f} MB/launch  {k[:110]}")
