"""Kernel timeline of one training step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline
    python3 tools/step_timeline.py /tmp/kt/*/*_kernel_trace.csv > timeline.txt

Prints stream, start, end, duration (us, relative to the end of the previous step's Adam) and grid of every dispatch of the
last complete step: what overlaps what on the main and the side stream, and where the gaps are."""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step ends with the first Adam launch after its loss kernel; the pipelined engine's second Adam launch (the late
    # slice, inside the NEXT step's forward) and the final flush are not step boundaries
    adam, seen_loss = [], False
    for i, r in enumerate(rows):
        if "euclid_masked_kernel" in r["Kernel_Name"] or "lin_out_kernel" in r["Kernel_Name"] or "lin_out_s6_kernel" in r["Kernel_Name"]:   # the criterion (separate / fused)
            seen_loss = True
        elif "adam_kernel" in r["Kernel_Name"] and seen_loss:
            adam.append(i)
            seen_loss = False
    a, b = adam[-2], adam[-1]
    t0 = int(rows[a]["End_Timestamp"])
    for r in rows[a + 1:b + 1]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("s%s %8.1f %8.1f %7.1f  %-64s grid=%s" % (r.get("Stream_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name,
                                                      r.get("Grid_Size_X", "?")))


if __name__ == "__main__":
    main(sys.argv[1])
